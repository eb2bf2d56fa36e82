#!/bin/bash
# MFMA utilisation and wave-state split of the streaming kernel from PMC counters (own pass, no trace domains beside
# --kernel-trace).  Run on the GPU box from the repo root:   bash scripts/pmc_mfma.sh cfg3|cfg5
# SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_bf16), summed over the SIMDs;
# GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS note); the SQ_WAVE/WAIT/ACTIVE counters are
# quad-cycles.  utilisation = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs).
set -e
CFG=${1:-cfg3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
d=$R/gpurun_out/pmc_mfma_$CFG
rm -rf $d
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --config $CFG --steps 6 --warmup 2 $PMC_EXTRA --no-cpu-baseline --settle-seconds 0 > $d.log 2>&1 \
  || { echo "rocprofv3 failed"; tail -5 $d.log; exit 1; }
python3 - <<PY
import csv, glob, collections, json, os
R="$R"; cfg="$CFG"
fs=sorted(glob.glob(f"{R}/gpurun_out/pmc_mfma_{cfg}/*/*counter_collection.csv"), key=os.path.getmtime)
per=collections.defaultdict(float)                       # one value per (dispatch, counter): rows of one dispatch are summed
for r in csv.DictReader(open(fs[-1])):
    per[(r["Kernel_Name"][:64], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for (k,_,cn),v in per.items():
    agg[k][cn].append(v)
out={}
for k,c in agg.items():
    if "MFMA" not in "".join(c.keys()) or not c.get("GRBM_GUI_ACTIVE"): continue
    n=len(c["GRBM_GUI_ACTIVE"])
    m={name: sum(v)/len(v) for name,v in c.items()}
    if m.get("SQ_VALU_MFMA_BUSY_CYCLES",0) <= 0: continue
    gui=m["GRBM_GUI_ACTIVE"]/8.0
    m["launches"]=n
    m["mfma_util"]=m["SQ_VALU_MFMA_BUSY_CYCLES"]/(gui*256*4)
    w=m.get("SQ_WAVE_CYCLES",0)
    if w>0:
        m["wave_wait_any_frac"]=m.get("SQ_WAIT_ANY",0)/w
        m["wave_wait_inst_frac"]=m.get("SQ_WAIT_INST_ANY",0)/w
        m["wave_active_inst_frac"]=m.get("SQ_ACTIVE_INST_ANY",0)/w
    out[k]=m
for k,m in sorted(out.items(), key=lambda kv:-kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"])[:6]:
    print(k, json.dumps({a:(round(b,4) if b<10 else round(b)) for a,b in m.items()}))
json.dump(out, open(f"{R}/gpurun_out/pmc_mfma_{cfg}.json","w"), indent=1)
PY
