#!/bin/bash
# headline with the Lanczos lambda_max inside its pass launches (VBMF_EXACT_LAMBDA=1, what vbmf_create picks at this size) against the squaring (=0):
# alternating processes, the default and the driver's invocation.   gpurun -- bash scripts/r03_exact_ab.sh <rounds>
rounds=${1:-3}
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
out=$R/gpurun_out/r03_exact_ab; mkdir -p $out
for r in $(seq 1 $rounds); do
  for v in 1 0; do
    VBMF_EXACT_LAMBDA=$v python bench.py --no-cpu-baseline > $out/def_${v}_$r.json 2>/dev/null
    VBMF_EXACT_LAMBDA=$v python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/drv_${v}_$r.json 2>/dev/null
  done
done
python - <<PY
import json,glob
for n in ("def","drv"):
  for v in (1,0):
    for f in sorted(glob.glob("$out/%s_%d_*.json"%(n,v))):
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%s EXACT_LAMBDA=%d  sweeps/s %8.1f  ms %.4f  pass1 %.4f  pass2 %.4f"%(n,v,d["value"],d["ms_per_step"],r["pass1"]["ms"],r["pass2"]["ms"]))
PY
