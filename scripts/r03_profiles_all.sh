#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r03_parity_rerun.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/r03_parity_rerun.log
bash scripts/r03_profile.sh cfg5 --config cfg5 --steps 40 --warmup 5
bash scripts/r03_profile.sh cfg4s8 --config cfg4 --shard-of 8 --steps 40 --warmup 5
bash scripts/r03_profile.sh shard8 --shard-of 8 --steps 100 --warmup 20
bash scripts/r03_profile.sh cfg2 --config cfg2 --steps 200 --warmup 30
bash scripts/r03_profile.sh cfg3 --steps 60 --warmup 10
