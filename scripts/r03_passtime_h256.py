"""Config 5's shape (100k x 10k, H = 256): each streaming pass launched alone, back to back (vbmf_debug_time_pass), in both orders --
is the Y*A pass slower than Y'B by itself, or only where it sits in the sweep?   python scripts/r03_passtime_h256.py  (GPU box)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package(); capi = pkg.capi
for (L, M, H) in [(100000, 10000, 256), (65536, 10000, 256), (10000, 100000, 256)]:
    with capi.Context(L, M, H, y_dtype=capi.VBMF_Y_BF16) as c:
        c.set_Y_synthetic(20170101, 16, 0.05)
        rng = np.random.default_rng(1)
        z = np.zeros((H, H))
        c.set_state(rng.standard_normal((M, H)), rng.standard_normal((L, H)), z, z, 0.1 * np.ones(H), 0.1 * np.ones(H), 0.1)
        d = c.dims()
        print(f"L={L} M={M} H={H}: pass1 (Y'B) x tiles {d['XT1']} splits {d['nsplit1']} steps {d['sps1']};  pass2 (Y*A) x tiles {d['XT2']} splits {d['nsplit2']} steps {d['sps2']}", flush=True)
        for rep in range(3):
            a = c.time_pass(1, 20); b = c.time_pass(2, 20); b2 = c.time_pass(2, 20); a2 = c.time_pass(1, 20)
            print("  alone, 20 launches back to back:  pass1 %.4f  pass2 %.4f  pass2 %.4f  pass1 %.4f ms" % (a, b, b2, a2), flush=True)
