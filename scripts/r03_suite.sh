#!/bin/bash
# whole GPU suite + the driver's headline command on one box.  usage: gpurun -- bash scripts/r03_suite.sh [tag]
tag=${1:-a}
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r03_suite_$tag
mkdir -p $out
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -6 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err; tail -c 600 $out/bench_driver.json
