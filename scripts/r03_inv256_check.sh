#!/bin/bash
# inv256_blk inside the library: every H > 128 parity test (dense, ARD-sparse, full_cov, two ranks), then timelines + bench lines of config 5 and of the
# dense H = 256 sweep.     gpurun -- bash scripts/r03_inv256_check.sh <tag>
tag=${1:-a}
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
out=$R/gpurun_out/r03_inv256_$tag; mkdir -p $out
scripts/r03_inv256_probe.bin > $out/probe.txt 2>&1; cat $out/probe.txt
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_sparse.py tests/test_gpu_two_ranks.py tests/test_gpu_dual.py tests/test_gpu_trial.py -q -m gpu > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $out/pytest.log
bash scripts/r03_profile.sh dense256_$tag --config cfg3 --H 256 --steps 30 --warmup 5 > $out/dense256.log 2>&1; grep "cov\|eig\|ctrl_end\|stream_lds8\|sweep:" $out/dense256.log
bash scripts/r03_profile.sh cfg5_$tag --config cfg5 --steps 40 --warmup 5 > $out/cfg5.log 2>&1; grep "cov\|eig\|stream_lds8\|sweep:" $out/cfg5.log
for i in 1 2; do
python bench.py --config cfg5 --steps 40 --warmup 5 --no-cpu-baseline > $out/cfg5_bench.json 2> $out/cfg5_bench.err; python bench.py --config cfg3 --H 256 --steps 30 --warmup 5 --no-cpu-baseline > $out/dense256_bench.json 2> $out/dense256_bench.err
python - <<PY
import json
for n in ("cfg5","dense256"):
    d=json.loads(open("$out/%s_bench.json"%n).read().strip().splitlines()[-1]); r=d["roofline"]
    print("%s: sweeps/s %.1f  ms %.4f  pass1 %.4f pass2 %.4f"%(n,d["value"],d["ms_per_step"],r["pass1"]["ms"],r["pass2"]["ms"]))
PY
done
