#!/bin/bash
# round 3: the H >= 128 paths after the streaming-kernel change -- the GPU tests that exercise them, then config 5 and config 4's
# rank share with the new kernel and (VBMF_LDS8=0) with the per-wave kernel, one process each on the same box
# usage (repo root): gpurun -- bash scripts/r03_bigh_check.sh [tag]
tag=${1:-a}
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r03_bigh_$tag
mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "128 or 200 or 256 or fullsize or many_ranks or sparse or hetero or dual or trial" > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $out/pytest.log
for cfg in cfg5 "cfg4 --shard-of 8"; do
  n=$(echo $cfg | tr -d ' -')
  python bench.py --config $cfg --no-cpu-baseline --steps 60 --warmup 10 > $out/bench_${n}_lds8.json 2> $out/bench_${n}_lds8.err
  VBMF_LDS8=0 python bench.py --config $cfg --no-cpu-baseline --steps 60 --warmup 10 > $out/bench_${n}_old.json 2> $out/bench_${n}_old.err
  python - <<PY
import json
for k in ("lds8","old"):
    try:
        d=json.loads(open("$out/bench_${n}_%s.json"%k).read().strip().splitlines()[-1])
        r=d.get("roofline",{})
        print("$cfg",k,"sweeps/s %.1f ms %.4f"%(d["value"],d["ms_per_step"]),"pass1 %.4f pass2 %.4f"%(r.get("pass1",{}).get("ms",0),r.get("pass2",{}).get("ms",0)))
    except Exception as e:
        print("$cfg",k,"FAILED",e)
PY
done
