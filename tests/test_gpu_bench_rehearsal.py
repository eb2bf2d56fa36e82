"""bench.py's own N > 1 code path, rehearsed on ONE GPU (VBMF_BENCH_TRANSPORT=host: every rank on device 0, rendezvous and the
library's all-reduces over gloo through host memory).  The round-end driver launches bench.py with torch.distributed.run on a
multi-GPU node that this pipeline cannot rent for tests; what CAN be checked here is everything in that invocation except
RCCL itself: rank / shard arithmetic, the communicator set-up branch, barrier + max-over-ranks timing, the single JSON line on
rank 0 -- and that the N-rank run ends in the same model state as the 1-rank run of the same command (strong scaling: the SAME
problem).  Exactly the driver's command line, at a small shape."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPE = ["--L", "6000", "--M", "700", "--H", "24", "--steps", "4", "--warmup", "2", "--settle-seconds", "0", "--no-cpu-baseline"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _line(out):
    lines = [l for l in out.strip().splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # ONE JSON line on stdout, from rank 0 only
    return json.loads(lines[0])


def test_bench_two_ranks_rehearsal_matches_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SHAPE, env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = _line(one.stdout)
    env2 = dict(env, VBMF_BENCH_TRANSPORT="host")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SHAPE,
                         env=env2, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert two.returncode == 0, two.stderr[-2000:]
    d2 = _line(two.stdout)
    assert d2["n_gpus"] == 2 and d2["steps"] == 4 and d2["warmup"] == 2 and d2["scaling"] == "strong"
    assert d2["config"]["row_shards"] == 2 and "rehearsal" in d2["config"] and d2["value"] > 0
    assert d2["metric"] == d1["metric"] and d2["unit"] == d1["unit"]
    # the same problem, sharded: the replicated scalars after 6 sweeps agree with the 1-rank run (fp32 partial sums in a different
    # order: 1e-4 is generous)
    assert abs(d2["final"]["sigma2"] - d1["final"]["sigma2"]) <= 1e-4 * abs(d1["final"]["sigma2"]), (d1["final"], d2["final"])
    # d = ||B_old - B_new|| / ||B_old|| is a difference of fp32-stored factors: 5e-3 relative + the 2e-6 noise floor of the parity tests
    assert abs(d2["final"]["d"] - d1["final"]["d"]) <= 5e-3 * abs(d1["final"]["d"]) + 2e-6, (d1["final"], d2["final"])
