#!/bin/bash
# whole GPU suite, then a list of benches.  usage: gpurun -- bash scripts/r03_suite.sh <tag> [bench specs "name|args" ...]
tag=${1:-a}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r03_suite_$tag
mkdir -p $out
cd $R
timeout -k 10 1000 python -m pytest tests -q -m gpu --durations=8 > $out/pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -n "FAILED\|Error\|passed\|failed" $out/pytest.log | tail -15
for spec in "$@"; do
  n=${spec%%|*}; args=${spec#*|}
  python bench.py $args --no-cpu-baseline > $out/bench_$n.json 2> $out/bench_$n.err || { echo "bench $n failed"; tail -3 $out/bench_$n.err; }
  python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$n.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$n: sweeps/s %.1f  ms %.4f  pass1 %.4f pass2 %.4f  chain %s"%(d["value"],d["ms_per_step"],r["pass1"]["ms"],r["pass2"]["ms"],d.get("control_chain_us")))
except Exception as e: print("$n failed",e)
PY
done
exit $rc
