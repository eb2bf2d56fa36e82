#!/bin/bash
# HBM traffic of the streaming kernel from PMC counters (separate passes, as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE cannot share a pass).  Run on the GPU box from the repo root:
#   bash scripts/pmc_traffic.sh
# Writes gpurun_out/pmc_{fetch,write}/ and prints per-kernel averages.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  d=$R/gpurun_out/pmc_$(echo $c | tr A-Z a-z)
  rm -rf $d
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --settle-seconds 0 > $d.log 2>&1 || { echo "rocprofv3 $c failed"; tail -5 $d.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
R="$R"
for c in ("fetch_size","write_size"):
    f=glob.glob(f"{R}/gpurun_out/pmc_{c}/*/*counter_collection.csv")
    if not f: print("no counter file for",c); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"][:64]].append(float(r["Counter_Value"]))
    print("==",c,"(KB per launch as reported; FETCH_SIZE reads 1/2 of wide streaming loads on gfx950)")
    for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:8]:
        print(f"  {k:64s} n={len(v):4d} avg={sum(v)/len(v):14.1f}")
PY
