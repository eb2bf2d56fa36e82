#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/r03_vbls
timeout -k 10 600 python -m pytest tests/test_gpu_vbls.py tests/test_gpu_sparse.py -q -m gpu -x -k "vbls or sessions or heteroscedastic" > gpurun_out/r03_vbls/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_vbls/pytest.log
python scripts/r03_vbls_mil.py > gpurun_out/r03_vbls/fast.txt 2>&1; cat gpurun_out/r03_vbls/fast.txt | tail -7
