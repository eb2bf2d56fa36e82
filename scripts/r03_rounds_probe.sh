#!/bin/bash
# SUPERSEDED by scripts/r03_rounds_alone.py (this version did not print the planner's split count: two of its four rows were split launches).
# Does config 5's Y*A pass (391 workgroups of 8 x tiles on 256 CUs) scale with its ROUNDS or with its WORK?  The same shape with L chosen so
# that the pass is exactly one round (254 workgroups), what config 5 has (391), and two full rounds (508).   gpurun -- bash scripts/r03_rounds_probe.sh
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
out=$R/gpurun_out/r03_rounds; mkdir -p $out
for L in 65024 100000 130048 65536; do
  python bench.py --config cfg5 --L $L --steps 30 --warmup 5 --no-cpu-baseline > $out/L$L.json 2> $out/L$L.err || tail -3 $out/L$L.err
  python - <<PY
import json
d=json.loads(open("$out/L$L.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("L=%6d  x-blocks of Y*A %4d  sweep %.4f ms  pass1 (Y'B) %.4f  pass2 (Y*A) %.4f   pass2 per 1000 rows %.4f us"%($L,($L+255)//256,d["ms_per_step"],r["pass1"]["ms"],r["pass2"]["ms"],1e3*r["pass2"]["ms"]/($L/1000)))
PY
done
