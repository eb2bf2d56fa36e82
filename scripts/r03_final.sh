#!/bin/bash
# final evidence of a build: whole GPU suite, then the un-profiled bench line of every configuration (one box)
tag=${1:-final}
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
out=$R/gpurun_out/r03_$tag; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -q -m gpu > $out/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $out/pytest.log
b() { n=$1; shift; python bench.py "$@" > $out/${n}_bench.json 2> $out/${n}_bench.err || { echo "bench $n failed"; tail -3 $out/${n}_bench.err; }; python - <<PY
import json
try:
    d=json.loads(open("$out/${n}_bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$n: sweeps/s %.1f  ms %.4f  frac %.3f (%s)  pass1 %.4f pass2 %.4f"%(d["value"],d["ms_per_step"],r["frac"],r["bound"],r["pass1"]["ms"],r["pass2"]["ms"]))
except Exception as e: print("$n failed",e)
PY
}
b cfg3_driver --gpus 1 --steps 20 --warmup 5
b cfg3
b cfg2 --config cfg2 --no-cpu-baseline
b shard8 --shard-of 8 --no-cpu-baseline
b cfg4s8 --config cfg4 --shard-of 8 --steps 60 --warmup 10 --no-cpu-baseline
b cfg5 --config cfg5 --steps 40 --warmup 5 --no-cpu-baseline
b cfg5h64 --config cfg5 --H 64 --steps 40 --warmup 5 --no-cpu-baseline
b cfg5h64full --config cfg5 --H 64 --full-cov --steps 40 --warmup 5 --no-cpu-baseline
b cfg4 --config cfg4 --steps 10 --warmup 2 --no-cpu-baseline
exit $rc
