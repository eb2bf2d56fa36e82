#!/bin/bash
# round 3: the blocked fp64 inverse in the control chain / sparse_cov_b / full_cov: whole GPU suite, then the benches whose numbers it moves
tag=${1:-a}
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r03_invchk_$tag
mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 $out/pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $out/pytest.log | head -20; }
b() { n=$1; shift; python bench.py "$@" --no-cpu-baseline > $out/bench_$n.json 2> $out/bench_$n.err || { echo "bench $n failed"; tail -3 $out/bench_$n.err; }; python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$n.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$n: sweeps/s %.1f  ms %.4f  pass1 %.4f pass2 %.4f  chain %s"%(d["value"],d["ms_per_step"],r["pass1"]["ms"],r["pass2"]["ms"],d.get("control_chain_us")))
except Exception as e: print("$n failed",e)
PY
}
b cfg3 --steps 100 --warmup 20
b cfg2 --config cfg2
b shard8 --shard-of 8
b cfg5h64 --config cfg5 --H 64 --steps 40 --warmup 5
b cfg5h64full --config cfg5 --H 64 --full-cov --steps 40 --warmup 5
b cfg4s8 --config cfg4 --shard-of 8 --steps 60 --warmup 10
