#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
out=$R/gpurun_out/r03_inv256_ab; mkdir -p $out
for r in 1 2; do
 for lib in new prev; do
  if [ $lib = prev ]; then export VBMF_HIP_LIB=$R/variants/libvbmf_prev.so; else unset VBMF_HIP_LIB; fi
  bash scripts/r03_profile.sh cfg5_${lib}_$r --config cfg5 --steps 40 --warmup 5 > $out/cfg5_${lib}_$r.log 2>&1; echo "== cfg5 $lib $r"; grep "cov\|eig\|sweep:" $out/cfg5_${lib}_$r.log
  bash scripts/r03_profile.sh dense_${lib}_$r --config cfg3 --H 256 --steps 30 --warmup 5 > $out/dense_${lib}_$r.log 2>&1; echo "== dense256 $lib $r"; grep "cov\|eig\|ctrl_end\|sweep:" $out/dense_${lib}_$r.log
  python bench.py --config cfg5 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 bench', '$lib', round(d['value'],1), round(d['ms_per_step'],4))"
  python bench.py --config cfg3 --H 256 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dense256 bench', '$lib', round(d['value'],1), round(d['ms_per_step'],4))"
 done
done
