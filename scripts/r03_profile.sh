#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py invocations; per tag: kernel stats CSV, one steady-state sweep as a timeline, the bench line
# usage: bash scripts/r03_profile.sh <tag> <bench.py args...>      (several: call it several times in one gpurun command)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r03_prof/$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/bench.py "$@" --no-cpu-baseline > $out/bench.json 2> $out/bench.err || { echo "profile $tag failed"; tail -3 $out/bench.err; exit 1; }
cd $root
kt=$(ls $out/*/*kernel_trace.csv | head -1)
ks=$(ls $out/*/*kernel_stats.csv | head -1)
cp $ks $out/kernel_stats.csv
python3 scripts/sweep_timeline.py $kt > $out/timeline.txt 2>&1
echo "== $tag"; tail -c 300 $out/bench.json | head -c 300; echo; cat $out/timeline.txt
