import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package(); capi = pkg.capi
L, M, H = 100000, 10000, 64
for splits in (12,):
    with capi.Context(L, M, H, y_dtype=capi.VBMF_Y_BF16, pass1_splits=splits) as c:
        c.set_Y_synthetic(20170101, H, 0.05)
        rng = np.random.default_rng(1)
        z = np.zeros((H, H))
        c.set_state(rng.standard_normal((M, H)), rng.standard_normal((L, H)), z, z, 0.1 * np.ones(H), 0.1 * np.ones(H), 0.1)
        print("dims", c.dims())
        for rep in range(3):
            print("standalone back-to-back: pass1 %.4f ms  pass2 %.4f ms" % (c.time_pass(1, 10), c.time_pass(2, 10)))
        c.profile_enable(True)
        c.run(20, eps=0.0, est_covs=True, est_var=True)
        p = c.profile_read()
        print("in pipeline:            pass1 %.4f ms  pass2 %.4f ms" % (p["pass1_ms"] / p["pass1_n"], p["pass2_ms"] / p["pass2_n"]))
        for rep in range(2):
            print("standalone after run:   pass1 %.4f ms  pass2 %.4f ms" % (c.time_pass(1, 10), c.time_pass(2, 10)))
