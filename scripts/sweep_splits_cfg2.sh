for s in 0 32 64 96 128 160 208; do python bench.py --config cfg2 --steps 200 --warmup 30 --no-cpu-baseline --splits $s > gpurun_out/sp2_$s.json; done
python - <<PY
import json
for s in (0,32,64,96,128,160,208):
    b=json.load(open("gpurun_out/sp2_%d.json"%s)); print(s, round(b["value"],1), round(b["roofline"]["pass1"]["ms"],4), round(b["roofline"]["pass2"]["ms"],4))
PY
