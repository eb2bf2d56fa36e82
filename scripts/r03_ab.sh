#!/bin/bash
# A/B of the H >= 128 streaming kernels on one box: VBMF_LDS8=1 (new) vs 0 (per-wave kernel), alternating processes
# usage (repo root): gpurun -- bash scripts/r03_ab.sh <tag> <rounds> <bench args...>
tag=$1; rounds=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r03_ab_$tag
mkdir -p $out
cd $R
for r in $(seq 1 $rounds); do
  for v in 1 0; do
    VBMF_LDS8=$v python bench.py "$@" --no-cpu-baseline > $out/b_${v}_$r.json 2> $out/b_${v}_$r.err || { echo "bench failed (LDS8=$v)"; tail -3 $out/b_${v}_$r.err; }
  done
done
python - <<PY
import json,glob
for v in (1,0):
    rows=[]
    for f in sorted(glob.glob("$out/b_%d_*.json"%v)):
        try:
            d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
            rows.append((d["value"],d["ms_per_step"],r["pass1"]["ms"],r["pass2"]["ms"]))
        except Exception as e: print(f,"failed",e)
    for x in rows: print("LDS8=%d  sweeps/s %8.1f  ms %.4f  pass1 %.4f  pass2 %.4f"%((v,)+x))
PY
