#!/bin/bash
# post_frag3 (table through LDS) against post_frag2 on one box: bit-identity of three sweeps, the H >= 128 parity tests, then alternating bench
# processes with VBMF_POST3=1 / 0.   usage (repo root): gpurun -- bash scripts/r03_post3_ab.sh <tag> <rounds>
tag=${1:-a}; rounds=${2:-2}
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
out=$R/gpurun_out/r03_post3_$tag; mkdir -p $out
timeout -k 10 300 python - > $out/identity.txt 2>&1 <<'PY'
import os, numpy as np
import __graft_entry__ as G
pkg = G.load_package()
pkg.set_defaults(y_dtype=pkg.VBMF_Y_BF16, factor_dtype=pkg.VBMF_FACTOR_AUTO)
def run(L, M, H, post3, sparse=False):
    os.environ["VBMF_POST3"] = str(post3)
    rng = np.random.default_rng(5)
    Y = rng.standard_normal((L, H)) @ rng.standard_normal((H, M)) + 0.1 * rng.standard_normal((L, M))
    if sparse:
        p = pkg.vbmf_sparse_init(Y, H, ca=1.0, cb=1.0, sigma=1.0, rng=np.random.default_rng(2))
        pkg.vbmf_sparse_(Y, p, 3, eps=0.0)
    else:
        p = pkg.vbmf_init(Y, H, ca=1.0, cb=1.0, sigma2=1.0, rng=np.random.default_rng(2))
        pkg.vbmf_(Y, p, 3, eps=0.0)
    pkg.invalidate()
    return p
for (L, M, H, sp) in [(70000, 300, 128, False), (66001, 260, 140, False), (9000, 2100, 256, False), (40000, 500, 256, True), (3000, 2500, 128, False)]:
    a, b = run(L, M, H, 1, sp), run(L, M, H, 0, sp)
    same = all(np.array_equal(getattr(a, f), getattr(b, f)) for f in ("AHat", "BHat", "SigmaA", "SigmaB"))
    print(f"L={L} M={M} H={H} sparse={sp}: post_frag3 == post_frag2 bit for bit: {same}", flush=True)
    assert same
PY
echo "identity rc=$?"; cat $out/identity.txt | tail -8
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_two_ranks.py -q -m gpu -k "128 or wide or above or config4 or config5 or large_rank or streamk" > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $out/pytest.log
for r in $(seq 1 $rounds); do
  for v in 1 0; do
    VBMF_POST3=$v python bench.py --config cfg5 --steps 40 --warmup 5 --no-cpu-baseline > $out/cfg5_${v}_$r.json 2> $out/cfg5_${v}_$r.err || tail -3 $out/cfg5_${v}_$r.err
    VBMF_POST3=$v python bench.py --config cfg4 --shard-of 8 --steps 60 --warmup 10 --no-cpu-baseline > $out/cfg4s8_${v}_$r.json 2> $out/cfg4s8_${v}_$r.err || tail -3 $out/cfg4s8_${v}_$r.err
  done
done
python - <<PY
import json,glob
for n in ("cfg5","cfg4s8"):
  for v in (1,0):
    for f in sorted(glob.glob("$out/%s_%d_*.json"%(n,v))):
        try:
            d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
            print("%s POST3=%d  sweeps/s %8.1f  ms %.4f  pass1 %.4f  pass2 %.4f"%(n,v,d["value"],d["ms_per_step"],r["pass1"]["ms"],r["pass2"]["ms"]))
        except Exception as e: print(f,"failed",e)
PY
