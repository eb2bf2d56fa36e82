#!/bin/bash
# round-3 probe on the GPU box: bare MFMA ceilings + streaming variants, then the same binary's streaming part under a PMC pass
# usage (repo root): gpurun -- bash scripts/r03_probe.sh [tag]
set -e
tag=${1:-a}
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r03_probe_$tag
mkdir -p $out
$R/scripts/r03_probe.bin ${2:-all} > $out/probe.txt 2>&1 || { echo "probe failed"; tail -20 $out/probe.txt; exit 1; }
cat $out/probe.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d $out/pmc -- $R/scripts/r03_probe.bin ${3:-stream} > $out/pmc.log 2>&1 || { echo "rocprofv3 failed"; tail -5 $out/pmc.log; exit 1; }
python3 - <<PY
import csv, glob, collections, json, os
out="$out"
fs=sorted(glob.glob(f"{out}/pmc/*/*counter_collection.csv"), key=os.path.getmtime)
per=collections.defaultdict(float)
for r in csv.DictReader(open(fs[-1])):
    per[(r["Kernel_Name"][:90], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for (k,_,cn),v in per.items():
    agg[k][cn].append(v)
res={}
for k,c in agg.items():
    if not c.get("GRBM_GUI_ACTIVE"): continue
    m={name: sum(v)/len(v) for name,v in c.items()}
    if m.get("SQ_VALU_MFMA_BUSY_CYCLES",0) <= 0: continue
    gui=m["GRBM_GUI_ACTIVE"]/8.0
    m["launches"]=len(c["GRBM_GUI_ACTIVE"])
    m["mfma_util"]=m["SQ_VALU_MFMA_BUSY_CYCLES"]/(gui*256*4)
    w=m.get("SQ_WAVE_CYCLES",0)
    if w>0:
        m["wave_wait_any_frac"]=m.get("SQ_WAIT_ANY",0)/w
        m["wave_wait_inst_frac"]=m.get("SQ_WAIT_INST_ANY",0)/w
        m["wave_active_inst_frac"]=m.get("SQ_ACTIVE_INST_ANY",0)/w
    res[k]=m
for k,m in sorted(res.items()):
    print(k, json.dumps({a:(round(b,4) if b<10 else round(b)) for a,b in m.items()}))
json.dump(res, open(f"{out}/pmc_mfma.json","w"), indent=1)
PY
