#!/bin/bash
# split-K of the H >= 128 Y*A pass (VBMF_PASS2_SPLITS; 0 = the planner's un-split launch), alternating processes.
tag=${1:-a}; vals=${2:-"0 2 3"}; rounds=${3:-2}
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
out=$R/gpurun_out/r03_p2split_$tag; mkdir -p $out
for r in $(seq 1 $rounds); do
  for v in $vals; do
    VBMF_PASS2_SPLITS=$v python bench.py --config cfg5 --steps 40 --warmup 5 --no-cpu-baseline > $out/cfg5_${v}_$r.json 2> $out/cfg5_${v}_$r.err || tail -3 $out/cfg5_${v}_$r.err
    VBMF_PASS2_SPLITS=$v python bench.py --config cfg4 --shard-of 8 --steps 60 --warmup 10 --no-cpu-baseline > $out/cfg4s8_${v}_$r.json 2> $out/cfg4s8_${v}_$r.err || tail -3 $out/cfg4s8_${v}_$r.err
  done
done
python - <<PY
import json,glob
for n in ("cfg5","cfg4s8"):
  for v in "$vals".split():
    for f in sorted(glob.glob("$out/%s_%s_*.json"%(n,v))):
        try:
            d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
            print("%s PASS2_SPLITS=%s  sweeps/s %8.1f  ms %.4f  pass1 %.4f  pass2 %.4f"%(n,v,d["value"],d["ms_per_step"],r["pass1"]["ms"],r["pass2"]["ms"]))
        except Exception as e: print(f,"failed",e)
PY
