#!/bin/bash
# inv256_blk against the Schur-complement inverse it replaced: the previous commit's library as an A/B variant, alternating on one box.
# Build the variant first (in the build container):
#   mkdir -p /tmp/prevsrc variants && git archive <commit before inv256_blk> vbmatrixfactorization.jl_amd/csrc include | tar -x -C /tmp/prevsrc
#   (cd /tmp/prevsrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared vbmatrixfactorization.jl_amd/csrc/vbmf_hip.hip -o $REPO/variants/libvbmf_prev.so -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib)
# then: gpurun -- bash scripts/r03_inv256_ab.sh      (the capi picks the variant up through VBMF_HIP_LIB)
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
