#!/usr/bin/env python3
"""Does a pass's streaming rate depend on WHERE its buffers landed?  Several contexts of the headline shape in one process
(each with its own allocations, all alive), the two plain streaming passes timed on each (vbmf_debug_time_pass, 40 launches)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package(); capi = pkg.capi
L, M, H = 100000, 10000, 64
ctxs = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    c = capi.Context(L, M, H, y_dtype=capi.VBMF_Y_BF16)
    c.set_Y_synthetic(20170101, H, 0.05)
    rng = np.random.default_rng(1)
    z = np.zeros((H, H))
    c.set_state(rng.standard_normal((M, H)), rng.standard_normal((L, H)), z, z, 0.1 * np.ones(H), 0.1 * np.ones(H), 0.1)
    ctxs.append(c)
for rep in range(3):
    for k, c in enumerate(ctxs):
        t1 = c.time_pass(1, 40); t2 = c.time_pass(2, 40)
        print(f"rep {rep} ctx {k}: pass1 {t1*1e3:7.1f} us  pass2 (no epilogue) {t2*1e3:7.1f} us", flush=True)
