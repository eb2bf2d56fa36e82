#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py for the configurations of round 2; summaries land under gpurun_out/r02_prof/<tag>/
# usage: bash scripts/r02_profile.sh <tag> <bench.py args...>
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r02_prof/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/bench.py "$@" --no-cpu-baseline > $out/bench.json 2> $out/bench.err
cd $root
ls $out/*/ | head -5
