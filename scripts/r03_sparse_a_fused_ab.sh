#!/bin/bash
# sparse_update_a_tiles_kernel (A update + operand tiles in one launch) against update kernel + retile: bit-identity of whole runs, the ARD-sparse /
# dual / trial / heteroscedastic test files, then alternating bench processes.   gpurun -- bash scripts/r03_sparse_a_fused_ab.sh <tag> <rounds>
tag=${1:-a}; rounds=${2:-2}
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
out=$R/gpurun_out/r03_safused_$tag; mkdir -p $out
timeout -k 10 300 python - > $out/identity.txt 2>&1 <<'PY'
import os, numpy as np
import __graft_entry__ as G
pkg = G.load_package()
def run(L, M, H, fused, ydt, labels=False, diag_var=False):
    os.environ["VBMF_SPARSE_A_FUSED"] = str(fused)
    pkg.set_defaults(y_dtype=ydt, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    rng = np.random.default_rng(5)
    Y = rng.standard_normal((L, H)) @ rng.standard_normal((H, M)) + 0.1 * rng.standard_normal((L, M))
    kw = {}
    if labels:
        kw = dict(labels=np.arange(1, M + 1, 3), H1=max(1, H // 3))
    p = pkg.vbmf_sparse_init(Y, H, ca=1.0, cb=1.0, sigma=1.0, rng=np.random.default_rng(2), **kw)
    pkg.vbmf_sparse_(Y, p, 4, eps=0.0, diag_var=diag_var)
    pkg.invalidate()
    return p
for (L, M, H, ydt, lab, dv) in [(700, 333, 5, "f32", False, False), (900, 1000, 40, "bf16", True, False), (3000, 2500, 128, "bf16", False, False),
                                (2000, 1500, 200, "bf16", True, False), (800, 300, 20, "f32", False, True), (5000, 77, 64, "bf16", False, False)]:
    yd = pkg.VBMF_Y_F32 if ydt == "f32" else pkg.VBMF_Y_BF16
    a, b = run(L, M, H, 1, yd, lab, dv), run(L, M, H, 0, yd, lab, dv)
    same = all(np.array_equal(getattr(a, f), getattr(b, f)) for f in ("AHat", "BHat", "diagSigmaATVec", "SigmaB", "CA", "CB")) and a.sigmaHat == b.sigmaHat
    print(f"L={L} M={M} H={H} {ydt} labels={lab} diag_var={dv}: fused == update + retile bit for bit: {same}", flush=True)
    assert same
PY
echo "identity rc=$?"; tail -8 $out/identity.txt
timeout -k 10 600 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_dual.py tests/test_gpu_trial.py tests/test_gpu_vbls.py tests/test_gpu_fullsize.py -q -m gpu > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $out/pytest.log
for r in $(seq 1 $rounds); do
  for v in 1 0; do
    VBMF_SPARSE_A_FUSED=$v python bench.py --config cfg5 --steps 40 --warmup 5 --no-cpu-baseline > $out/cfg5_${v}_$r.json 2> $out/cfg5_${v}_$r.err || tail -3 $out/cfg5_${v}_$r.err
    VBMF_SPARSE_A_FUSED=$v python bench.py --config cfg5 --H 64 --steps 40 --warmup 5 --no-cpu-baseline > $out/cfg5h64_${v}_$r.json 2> $out/cfg5h64_${v}_$r.err || tail -3 $out/cfg5h64_${v}_$r.err
  done
done
python - <<PY
import json,glob
for n in ("cfg5","cfg5h64"):
  for v in (1,0):
    for f in sorted(glob.glob("$out/%s_%d_*.json"%(n,v))):
        try:
            d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
            print("%s SPARSE_A_FUSED=%d  sweeps/s %8.1f  ms %.4f  pass1 %.4f  pass2 %.4f"%(n,v,d["value"],d["ms_per_step"],r["pass1"]["ms"],r["pass2"]["ms"]))
        except Exception as e: print(f,"failed",e)
PY
