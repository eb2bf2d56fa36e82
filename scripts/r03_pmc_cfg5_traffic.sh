#!/bin/bash
# Where do config 5's two streaming passes (the same kernel, alternating dispatches) get their bytes from?  FETCH_SIZE and the L2
# hit / miss counts per dispatch, in dispatch order (separate counter passes).   gpurun -- bash scripts/r03_pmc_cfg5_traffic.sh [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03_pmc_cfg5_traffic; mkdir -p $out
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  n=$(echo $c | tr ' ' '_')
  rm -rf $out/$n
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$n -- python3 $R/bench.py --config cfg5 --steps 4 --warmup 2 --no-cpu-baseline --settle-seconds 0 "$@" > $out/$n.log 2>&1 || { echo "rocprofv3 $c failed"; tail -5 $out/$n.log; }
done
python3 - <<PY
import csv, glob, collections
out="$out"
for d in sorted(glob.glob(out+"/*/")):
    f=glob.glob(d+"*/*counter_collection.csv")
    if not f: print("no counters in",d); continue
    per=collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        if "stream_lds8_kernel" not in r["Kernel_Name"]: continue
        k=(int(r["Dispatch_Id"]), r["Counter_Name"])
        per[k]=per.get(k,0.0)+float(r["Counter_Value"])
    names=sorted({k[1] for k in per})
    ids=sorted({k[0] for k in per})
    print("==",d.rstrip("/").split("/")[-1],"per stream_lds8_kernel dispatch, in order (pass 1 = Y'B, pass 2 = Y*A alternate)")
    for n in names:
        print("  %-24s"%n, " ".join("%.4g"%per[(i,n)] for i in ids))
PY
