#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
out=$R/gpurun_out/r03_splits; mkdir -p $out
for r in 1 2; do for sp in 0 6 8; do
python bench.py --config cfg5 --steps 40 --warmup 5 --no-cpu-baseline --splits $sp > $out/b_${sp}_$r.json 2> $out/b_${sp}_$r.err || tail -3 $out/b_${sp}_$r.err
python - <<PY
import json
d=json.loads(open("$out/b_${sp}_$r.json").read().strip().splitlines()[-1]); r=d["roofline"]; print("cfg5 splits=$sp: sweeps/s %.1f ms %.4f pass1 %.4f pass2 %.4f"%(d["value"],d["ms_per_step"],r["pass1"]["ms"],r["pass2"]["ms"]))
PY
done; done
