import sys, os
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as G
pkg = G.load_package(); capi = pkg.capi
M, H = 10000, 64
for L in (2500, 5000, 10000, 20000, 40000):
    c = capi.Context(L, M, H, y_dtype=capi.VBMF_Y_BF16)
    c.set_Y_synthetic(20170101, H, 0.05)
    rng = np.random.default_rng(1); z = np.zeros((H, H))
    c.set_state(rng.standard_normal((M, H)), rng.standard_normal((L, H)), z, z, 0.1 * np.ones(H), 0.1 * np.ones(H), 0.1)
    for rep in range(2):
        t1 = c.time_pass(1, 100); t2 = c.time_pass(2, 100)
    b = L * M * 2
    print(f"L={L:6d} Y copy {b/1e6:6.0f} MB: pass1 {t1*1e3:7.1f} us ({b/t1/1e9:6.2f} TB/s)  pass2 {t2*1e3:7.1f} us ({b/t2/1e9:6.2f} TB/s)  dims {c.dims() if hasattr(c,'dims') else ''}", flush=True)
    c.close()
