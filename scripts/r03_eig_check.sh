#!/bin/bash
# lambda_max for H > 128 (eig_lanczos_kernel): the H > 128 parity tests, then timelines of the dense and the
# ARD-sparse model at 100k x 10k, H = 256.     gpurun -- bash scripts/r03_eig_check.sh <tag>
tag=${1:-a}
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
out=$R/gpurun_out/r03_eig_$tag; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_sparse.py tests/test_gpu_two_ranks.py -q -m gpu -k "above or 140 or h200 or 256 or config5 or wide_rank or straddle" > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $out/pytest.log
bash scripts/r03_profile.sh dense256_$tag --config cfg3 --H 256 --steps 6 --warmup 2 > $out/dense256.log 2>&1; tail -22 $out/dense256.log
bash scripts/r03_profile.sh cfg5_$tag --config cfg5 --steps 6 --warmup 2 > $out/cfg5.log 2>&1; tail -24 $out/cfg5.log
python bench.py --config cfg5 --steps 40 --warmup 5 --no-cpu-baseline > $out/cfg5_bench.json 2> $out/cfg5_bench.err; python bench.py --config cfg3 --H 256 --steps 30 --warmup 5 --no-cpu-baseline > $out/dense256_bench.json 2> $out/dense256_bench.err
python - <<PY
import json
for n in ("cfg5","dense256"):
    d=json.loads(open("$out/%s_bench.json"%n).read().strip().splitlines()[-1]); r=d["roofline"]
    print("%s: sweeps/s %.1f  ms %.4f  pass1 %.4f pass2 %.4f"%(n,d["value"],d["ms_per_step"],r["pass1"]["ms"],r["pass2"]["ms"]))
PY
