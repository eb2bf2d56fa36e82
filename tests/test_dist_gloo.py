"""world_size-2 CPU test (gloo) of the row-sharded sweep: the same two all-reduces the library issues
over RCCL (M x H partial of Y'B; packed B Grams + the row-local part of tr(B'YA)) reproduce the unsharded oracle, and the replicated
quantities stay bit-identical across ranks."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as G
    from oracle import vbmf_oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec_dir = os.path.join(G.PKG_DIR, "dist.py")
        import importlib.util
        spec = importlib.util.spec_from_file_location("vbmf_dist", spec_dir)
        D = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(D)

        L, M, H = 157, 64, 5
        rng = np.random.default_rng(42)
        Y, _, _ = O.toy_matrix(L, M, H, 0.05, rng)
        p = O.vbmf_init(Y, H, ca=0.1, cb=0.1, sigma2=0.1, rng=np.random.default_rng(43), materialize_yhat=False)
        ref = O.copy_params(p)
        for f in ("AHat", "BHat", "SigmaA", "SigmaB", "CA", "CB", "invCA", "invCB"):
            setattr(ref, f, getattr(p, f).copy())
        r0, n = D.row_shard(L, world, rank)
        Yg, Bg = Y[r0:r0 + n], p.BHat[r0:r0 + n].copy()

        def allreduce(x):
            t = torch.from_numpy(np.ascontiguousarray(x))
            dist.all_reduce(t)
            return t.numpy()

        trYY = float(allreduce(np.array([np.sum(Yg * Yg)]))[0])
        GB = allreduce(Bg.T @ Bg)
        SigmaA, SigmaB = p.SigmaA, p.SigmaB
        ca, cb, s2 = np.diag(p.CA).copy(), np.diag(p.CB).copy(), p.sigma2
        ds = []
        for sweep in range(4):
            SigmaA = s2 * np.linalg.inv(GB + L * SigmaB + s2 * np.diag(1 / ca))
            P = allreduce(Yg.T @ Bg)                                   # collective 1
            A = (P @ SigmaA) / s2
            GA = A.T @ A                                               # replicated, no collective
            KB = GA + M * SigmaA + s2 * np.diag(1 / cb)
            SigmaB = s2 * np.linalg.inv(KB)
            Qg = Yg @ A
            Bnew = (Qg @ SigmaB) / s2
            dB = Bg - Bnew
            # collective 2: ONE packed message [B'B | dB'dB | tr(B'YA) = sum (Y_g A) o B_g], as the library sends it
            packed = allreduce(np.concatenate([(Bnew.T @ Bnew).ravel(), (dB.T @ dB).ravel(), [np.sum(Qg * Bnew)]]))
            GBold, GB, GD, tr = GB, packed[:H * H].reshape(H, H), packed[H * H:2 * H * H].reshape(H, H), float(packed[-1])
            Bg = Bnew
            ca = np.diag(GA) / M + np.diag(SigmaA)
            cb = np.diag(GB) / L + np.diag(SigmaB)
            s2 = (trYY - 2 * tr + np.sum((GA + M * SigmaA) * (GB + L * SigmaB))) / (L * M)
            ds.append(np.sqrt(np.linalg.eigvalsh(GD)[-1] / np.linalg.eigvalsh(GBold)[-1]))
        tro = []
        O.vbmf_(Y, ref, 4, eps=0.0, est_covs=True, est_var=True, trace=tro)
        err = dict(A=np.abs(A - ref.AHat).max(), B=np.abs(Bg - ref.BHat[r0:r0 + n]).max(),
                   s2=abs(s2 - ref.sigma2) / ref.sigma2, d=max(abs(a - b[0]) / b[0] for a, b in zip(ds, tro)),
                   SB=np.abs(SigmaB - ref.SigmaB).max())
        # replicas bit-identical
        gathered = [torch.zeros(A.shape, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(A))
        same = all(torch.equal(g, gathered[0]) for g in gathered)
        q.put((rank, err, same, (r0, n)))
    finally:
        dist.destroy_process_group()


def test_row_shard_partition():
    import importlib.util
    spec = importlib.util.spec_from_file_location("vbmf_dist", os.path.join(ROOT, "vbmatrixfactorization.jl_amd", "dist.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    for L, w in [(100000, 8), (10, 3), (157, 2), (1000000, 8), (7, 7)]:
        parts = [D.row_shard(L, w, r) for r in range(w)]
        assert parts[0][0] == 0 and sum(n for _, n in parts) == L
        assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        assert max(n for _, n in parts) - min(n for _, n in parts) <= 1
    with pytest.raises(ValueError):
        D.row_shard(3, 4, 0)


def test_sharded_sweep_matches_unsharded_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, same, shard in res:
        assert same, f"rank {rank}: replicated AHat differs across ranks"
        assert err["A"] < 1e-11 and err["B"] < 1e-11 and err["SB"] < 1e-12, err
        assert err["s2"] < 1e-9 and err["d"] < 1e-7, err
    assert sorted(r[3] for r in res) == [(0, 79), (79, 78)]
