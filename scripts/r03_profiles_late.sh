#!/bin/bash
# the round's last profile set (after post_frag3 and the Lanczos lambda_max): timelines + kernel stats of every bench configuration, the dense
# H = 256 case, the MFMA counters of config 5 and the headline's HBM traffic.   gpurun -- bash scripts/r03_profiles_late.sh
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
bash scripts/r03_profile.sh cfg5 --config cfg5 --steps 40 --warmup 5
bash scripts/r03_profile.sh cfg4s8 --config cfg4 --shard-of 8 --steps 40 --warmup 5
bash scripts/r03_profile.sh shard8 --shard-of 8 --steps 100 --warmup 20
bash scripts/r03_profile.sh cfg2 --config cfg2 --steps 200 --warmup 30
bash scripts/r03_profile.sh cfg3 --steps 60 --warmup 10
bash scripts/r03_profile.sh dense256 --config cfg3 --H 256 --steps 30 --warmup 5
bash scripts/pmc_mfma.sh cfg5 2>&1 | tail -8
bash scripts/pmc_traffic.sh 2>&1 | tail -14
