#!/bin/bash
# stream-K split of the Y*A pass: the tests that take it, then config 5 with and without it (alternating processes)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
out=$R/gpurun_out/r03_sk; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_vbls.py tests/test_gpu_sparse.py -q -m gpu -x -k "streamk or long_side or config5 or vbls or sparse_run or full_size" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
for r in 1 2; do for v in 1 0; do
  VBMF_STREAMK=$v python bench.py --config cfg5 --steps 40 --warmup 5 --no-cpu-baseline > $out/b_${v}_$r.json 2> $out/b_${v}_$r.err || tail -3 $out/b_${v}_$r.err
done; done
python - <<PY
import json,glob
for v in (1,0):
    for f in sorted(glob.glob("$out/b_%d_*.json"%v)):
        try:
            d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
            print("STREAMK=%d  sweeps/s %8.1f  ms %.4f  pass1 %.4f  pass2 %.4f"%(v,d["value"],d["ms_per_step"],r["pass1"]["ms"],r["pass2"]["ms"]))
        except Exception as e: print(f,"failed",e)
PY
