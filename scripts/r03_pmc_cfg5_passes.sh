#!/bin/bash
# Config 5's two passes are the SAME kernel (stream_lds8_kernel<8, 0>): counters per dispatch, in order, with the dispatch's duration from the
# kernel trace -> clock (GRBM_GUI_ACTIVE / 8 XCDs / duration), matrix-pipe busy fraction, wave-state split of pass 1 (Y'B) and pass 2 (Y*A) apart.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
d=$R/gpurun_out/r03_pmc_cfg5_passes; rm -rf $d
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --config cfg5 --steps 6 --warmup 2 --no-cpu-baseline --settle-seconds 0 "$@" > $d.log 2>&1 \
  || { echo "rocprofv3 failed"; tail -5 $d.log; exit 1; }
python3 - <<PY
import csv, glob, collections
d="$d"
per=collections.defaultdict(dict)
for r in csv.DictReader(open(glob.glob(d+"/*/*counter_collection.csv")[0])):
    if "stream_lds8_kernel" not in r["Kernel_Name"]: continue
    k=int(r["Dispatch_Id"]); per[k][r["Counter_Name"]]=per[k].get(r["Counter_Name"],0.0)+float(r["Counter_Value"])
dur={}
for r in csv.DictReader(open(glob.glob(d+"/*/*kernel_trace.csv")[0])):
    if "stream_lds8_kernel" in r["Kernel_Name"]: dur[int(r["Dispatch_Id"])]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-3
print("dispatch   us      GHz   mfma_util  wait_any  wait_inst  active_inst  wave_cycles/1e6")
for i,k in enumerate(sorted(per)):
    m=per[k]; us=dur.get(k,float("nan")); gui=m["GRBM_GUI_ACTIVE"]/8
    w=m["SQ_WAVE_CYCLES"]
    print("%s %4d  %7.1f  %.3f   %.3f     %.3f    %.3f     %.3f      %.1f"%("pass1" if i%2==0 else "pass2",k,us,gui/us*1e-3,m["SQ_VALU_MFMA_BUSY_CYCLES"]/(gui*1024),m["SQ_WAIT_ANY"]/w,m["SQ_WAIT_INST_ANY"]/w,m["SQ_ACTIVE_INST_ANY"]/w,w/1e6))
PY
