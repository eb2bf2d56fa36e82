for s in 0 4 6 8 10 12 16 24; do python bench.py --shard-of 8 --steps 60 --warmup 10 --no-cpu-baseline --splits $s > gpurun_out/sp_$s.json; done
python - <<PY
import json
for s in (0,4,6,8,10,12,16,24):
    b=json.load(open("gpurun_out/sp_%d.json"%s)); print(s, round(b["value"],1), round(b["roofline"]["pass1"]["ms"],4), round(b["roofline"]["pass2"]["ms"],4))
PY
