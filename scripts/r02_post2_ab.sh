#!/bin/bash
# A/B of post_frag2's row tiles per wave on a long side (variants/libvbmf_{a,b}.so built with -DVBMF_POST2_NXT8 / -DVBMF_POST2_NXT4):
#   default: NH=8 -> 2, NH=4 -> 4 (one wave per SIMD);  a: 1 / 2 (two waves per SIMD);  b: 1 / 1
cd $GRAFT_REPO_ROOT
for v in default a b; do
  # (the variant libraries live where r02_epi_ab.sh and scripts/README.md put them)
  if [ $v = default ]; then unset VBMF_HIP_LIB; else export VBMF_HIP_LIB=$GRAFT_REPO_ROOT/vbmatrixfactorization.jl_amd/variants/libvbmf_$v.so; fi
  if [ -n "$VBMF_HIP_LIB" ] && [ ! -f "$VBMF_HIP_LIB" ]; then echo "missing variant library $VBMF_HIP_LIB (build it: scripts/README.md)"; exit 1; fi
  for cfg in "--config cfg5" "--config cfg4 --shard-of 8"; do
    python bench.py $cfg --steps 40 --warmup 5 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', '$cfg', round(d['value'],1), round(d['ms_per_step']*1e3,1), 'us p1', round(r['pass1']['ms']*1e3,1), 'p2', round(r['pass2']['ms']*1e3,1))"
  done
done
