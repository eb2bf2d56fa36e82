#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/r03_inv_b
$R/scripts/r03_inv_probe.bin > $R/gpurun_out/r03_inv_b/probe.txt 2>&1; cat $R/gpurun_out/r03_inv_b/probe.txt | cut -c1-250
bash $R/scripts/r03_inv_check.sh b
