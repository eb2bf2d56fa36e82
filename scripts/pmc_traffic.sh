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
import csv, glob, collections, json
R="$R"
res={}
for c in ("fetch_size","write_size"):
    f=glob.glob(f"{R}/gpurun_out/pmc_{c}/*/*counter_collection.csv")
    if not f: print("no counter file for",c); continue
    per=collections.defaultdict(float)                      # one value per (kernel, dispatch): rows of one dispatch are summed
    for r in csv.DictReader(open(f[0])):
        per[(r["Kernel_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    agg=collections.defaultdict(list)
    for (k,_),v in per.items(): agg[k].append(v)
    print("==",c,"(KB per launch as reported; FETCH_SIZE reads 1/2 of wide streaming loads on gfx950)")
    for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:8]:
        print(f"  {k[:90]:90s} n={len(v):4d} avg={sum(v)/len(v):14.1f}")
        if "stream_gemm_kernel" in k or "stream_lds8_kernel" in k:
            which = "pass2" if k.split(">")[0].rstrip().endswith("1") else "pass1"      # last template argument: EPI
            res.setdefault(which,{})[c]=sum(v)/len(v); res[which]["launches"]=len(v)
# headline workload (cfg3): algorithmic bytes of SURVEY 8(d)
L,M,H=100000,10000,64
alg={"pass1": L*M*2+L*H*4+M*H*4, "pass2": L*M*2+M*H*4+2*L*H*4}
out={"config":{"L":L,"M":M,"H":H,"y_dtype":"bf16","factor_operand":"bf16x2","n_gpus":1},
     "correction":"gfx950: FETCH_SIZE reports 1/2 of wide (16 B/lane) streaming loads (MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE as reported (4 B/lane stores: uncalibrated width)",
     "command":"bash scripts/pmc_traffic.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, --kernel-trace only) -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --settle-seconds 0"}
tot=0; n=0
for w in ("pass1","pass2"):
    if w in res and "fetch_size" in res[w] and "write_size" in res[w]:
        tb=2*res[w]["fetch_size"]*1024+res[w]["write_size"]*1024
        out[w]={"FETCH_SIZE_KB_reported":res[w]["fetch_size"],"WRITE_SIZE_KB_reported":res[w]["write_size"],"traffic_bytes":tb,"algorithmic_bytes":float(alg[w]),"launches":res[w]["launches"]}
        tot+=tb; n+=1
if n==2:
    out["traffic_bytes_per_launch"]=tot/2
    json.dump(out,open(f"{R}/gpurun_out/pmc_stream_kernel_cfg3.json","w"),indent=1)
    print("traffic per launch %.4g B (algorithmic %.4g)"%(tot/2,(alg["pass1"]+alg["pass2"])/2))
PY
