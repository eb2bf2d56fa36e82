#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
out=$R/gpurun_out/r03_full; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_dual.py tests/test_gpu_trial.py -q -m gpu -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
for r in 1 2; do
python bench.py --config cfg5 --H 64 --full-cov --steps 40 --warmup 5 --no-cpu-baseline > $out/b_$r.json 2> $out/b_$r.err || tail -3 $out/b_$r.err
python - <<PY
import json
d=json.loads(open("$out/b_$r.json").read().strip().splitlines()[-1]); print("cfg5 H=64 full_cov: sweeps/s %.1f ms %.4f"%(d["value"],d["ms_per_step"]))
PY
done
