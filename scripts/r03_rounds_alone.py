"""The H = 256 Y*A launch alone (vbmf_debug_time_pass, 20 back to back) at row counts that make it exactly one round of workgroups, config 5's
1.53 and two full rounds -- with the planner's split count printed (it must be 1 for the comparison to mean anything)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package(); capi = pkg.capi
M, H = 10000, 256
for L in (65024, 100000, 130048, 195072):
    with capi.Context(L, M, H, y_dtype=capi.VBMF_Y_BF16) as c:
        c.set_Y_synthetic(20170101, 16, 0.05)
        rng = np.random.default_rng(1)
        z = np.zeros((H, H))
        c.set_state(rng.standard_normal((M, H)), rng.standard_normal((L, H)), z, z, 0.1 * np.ones(H), 0.1 * np.ones(H), 0.1)
        d = c.dims()
        t = [c.time_pass(2, 20) for _ in range(3)]
        nb = (d["XT2"] + 7) // 8
        print(f"L={L:6d}: Y*A x blocks {nb:4d} (= {nb / 254:.2f} rounds of 254 CUs), splits {d['nsplit2']}, k-steps {d['sps2']}:  {min(t):.4f} ms alone  ->  {1e3 * min(t) / nb:.3f} us per block,  {min(t) / max(1, -(-nb // 254)):.4f} ms per round", flush=True)
