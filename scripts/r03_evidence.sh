#!/bin/bash
# round 3 evidence run on one box: whole GPU suite; PMC (MFMA busy / wave states) of config 5 and of config 4's rank share; HBM traffic of the headline passes
tag=${1:-a}
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
out=$R/gpurun_out/r03_evidence_$tag; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -q -m gpu > $out/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 $out/pytest.log
bash scripts/pmc_mfma.sh cfg5 2>&1 | tail -8
cp gpurun_out/pmc_mfma_cfg5.json $out/ 2>/dev/null
bash scripts/pmc_traffic.sh 2>&1 | tail -14
cp gpurun_out/pmc_stream_kernel_cfg3.json $out/ 2>/dev/null
exit $rc
