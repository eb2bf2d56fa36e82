#!/bin/bash
# A/B of the register-epilogue Y*A pass at the headline size (and at 70k rows):
#   balance = VBMF_EPI_BALANCE: three x groups per workgroup over the whole chip (1) | four per workgroup (0)
#   build   = default (Y ring 6 deep, previous-factor rows loaded one tile ahead) | dy12 (Y ring 12 deep) | pv0 (rows loaded at the tile head)
for L in 100000 70000; do
for v in default dy12 pv0; do
  if [ $v = default ]; then unset VBMF_HIP_LIB; else export VBMF_HIP_LIB=$PWD/vbmatrixfactorization.jl_amd/variants/libvbmf_$v.so; fi
  for bal in 1 0; do
    VBMF_EPI_BALANCE=$bal python bench.py --L $L --no-cpu-baseline --steps 200 --warmup 30 2>gpurun_out/ab_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['control_chain_us']; print('L=$L $v balance=$bal', round(d['value'],1), 'p1', round(r['pass1']['ms'],4), 'p2', round(r['pass2']['ms'],4), 'tail', c['epilogue_table'], c['epilogue_tiles'], c['epilogue_fold'])" || tail -3 gpurun_out/ab_err.txt
  done
done
done
