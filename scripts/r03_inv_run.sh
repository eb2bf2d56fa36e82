#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/r03_inv_c
$R/scripts/r03_inv_probe.bin > $R/gpurun_out/r03_inv_c/probe.txt 2>&1; cat $R/gpurun_out/r03_inv_c/probe.txt | cut -c1-250
cd $R && bash scripts/r03_suite.sh d "cfg3|--steps 100 --warmup 20" "cfg5h64full|--config cfg5 --H 64 --full-cov --steps 40 --warmup 5" "cfg2|--config cfg2"
