# A/B of the narrow streaming geometry (VBMF_NARROW=0|1) on the small shapes
for n in 0 1; do
  VBMF_NARROW=$n python bench.py --config cfg2 --no-cpu-baseline > gpurun_out/nw_cfg2_$n.json
  for s in 8 4 2; do VBMF_NARROW=$n python bench.py --shard-of $s --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/nw_s${s}_$n.json; done
done
python - <<PY
import json
for k in ("cfg2","s8","s4","s2"):
    for n in (0,1):
        b=json.load(open("gpurun_out/nw_%s_%d.json"%(k,n))); print(k, "narrow" if n else "wide  ", round(b["value"],1), round(b["ms_per_step"],4), round(b["roofline"]["pass1"]["ms"],4), round(b["roofline"]["pass2"]["ms"],4))
PY
