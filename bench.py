#!/usr/bin/env python3
"""bench.py -- VB iterations/sec of the vbmf! sweep on MI355X (BASELINE.json metric).

One "step" = one full sweep of vbmf! (src/vbmf.jl:193-214) with est_covs = est_var = true:
updateA!, updateB!, updateCA!, updateCB!, updateSigma2!, the convergence scalar d and the ELBO, on
synthetic toy_matrix data (examples/toy_data.jl:7-18) that is generated on the device and stays
resident in HBM.  eps = 0, so the loop never exits early.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, Y row-sharded (strong scaling: the SAME 100k x 10k problem), the M x H
partial of Y'B and the packed Grams all-reduced with RCCL inside the library, on its compute stream.

Prints ONE JSON line on rank 0.  `roofline` is for the streaming-contraction kernel (both passes of a
sweep are launches of it): algorithmic bytes per launch / average launch duration, measured with HIP
events recorded on the library's own stream during the timed region.  `roofline.traffic` is the
PMC-measured HBM traffic per launch (committed figure, `traffic_source` names the profile file it comes from);
`roofline.sweep_frac` prices the WHOLE sweep (algorithmic bytes of both passes / ms_per_step) against the same peak.
`cpu_baseline` times the fp64 oracle in the reference's operation order (OpenBLAS, all host cores) on ALL rows of the
workload for `--cpu-sweeps` sweeps (`--cpu-rows N` bounds it to a row sample instead).  `--settle-seconds` runs untimed
pass launches (state not advanced) before the warm-up so the clocks are settled when the timed region starts.
`control_chain_us`: last durations of the parts of the fp64 H x H control chain that rides in the pass launches
(ctrl_end, SigmaA, lambda_max + loop test, SigmaB), stamped on the device.  `--shard-of N` runs rank 0's share of an N-rank
job on one GPU; `--config cfg2|cfg4|cfg5` the other BASELINE shapes; `--full-cov` (with --config cfg5 --H <= 64) the
per-column full-covariance branch of the sparse model.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (L, M, H, y_dtype)  -- BASELINE.json configs[2] is the headline the metric is quoted on
    "cfg3": (100000, 10000, 64, "bf16"),
    "cfg2": (10000, 1000, 32, "f32"),
    "cfg4": (1000000, 10000, 128, "bf16"),
    "cfg5": (100000, 10000, 256, "bf16"),      # vbmf_sparse.jl ARD-sparse (diagonal branch)
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: a sweep is < 1 ms at the headline size, and the device needs a few tens of sweeps to settle its clocks
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--L", type=int, default=0)
    ap.add_argument("--M", type=int, default=0)
    ap.add_argument("--H", type=int, default=0)
    ap.add_argument("--ydtype", default="", choices=["", "bf16", "f32"])
    ap.add_argument("--factor", default="auto", choices=["auto", "bf16", "bf16x2"])
    ap.add_argument("--splits", type=int, default=0, help="split-K of the Y'B pass (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-cov", action="store_true",
                    help="cfg5 only: updateA! with full_cov=true (one H x H inverse per column of Y on the device; H <= 64)")
    ap.add_argument("--event-stride", type=int, default=4,
                    help="HIP events bracket every k-th launch of each pass inside the timed region (each timed launch "
                         "costs two event packets on the stream; 1 = every launch)")
    ap.add_argument("--shard-of", type=int, default=0,
                    help="developer aid on ONE GPU: run rank 0's share of an N-rank strong-scaling job (L/N rows, "
                         "L_global = L, 1-rank communicator => the collective code path); an upper bound for N GPUs")
    ap.add_argument("--cpu-rows", type=int, default=0,
                    help="rows of the CPU baseline (0 = all L rows: the faithful sweep holds an M x M product that does not "
                         "scale with L, so a row sample cannot be extrapolated linearly)")
    ap.add_argument("--cpu-sweeps", type=int, default=2)
    ap.add_argument("--settle-seconds", type=float, default=0.3,
                    help="untimed sweeps run for this long BEFORE the W warm-up sweeps, so that the device's clocks have ramped "
                         "up from idle when the timed region starts (reported as clock_settle in the JSON line)")
    return ap.parse_args()


def cpu_baseline(ctx, L, M, H, rows, sweeps, seed):
    """fp64 oracle, reference operation order ("faithful": three GEMM passes over Y, the Y.^2 and 2Y'
    temporaries and the M x M product of updateSigma2!, src/vbmf.jl:95-157), on the SAME matrix the GPU holds
    (read back from the device).  All L rows by default: 2*M^2*H of the faithful sweep's flops (the M x M product,
    src/vbmf.jl:154) do not scale with L, so timing a row sample and extrapolating linearly over-charges the CPU."""
    from oracle import vbmf_oracle as O          # checker/baseline only -- never on the product path
    rows = int(ctx.L if rows <= 0 else min(rows, ctx.L))
    Y = np.ascontiguousarray(ctx.get_Y(0, rows))
    rng = np.random.default_rng(seed)
    out = {}
    for kind, fused in (("faithful", False), ("fused", True)):
        p = O.vbmf_init(Y, H, ca=0.1, cb=0.1, sigma2=0.1, rng=np.random.default_rng(seed), materialize_yhat=False)
        O.vbmf_(Y, p, 1, eps=0.0, est_covs=True, est_var=True, fused=fused)          # warm-up
        ts = []
        for _ in range(sweeps):
            t0 = time.perf_counter()
            O.vbmf_(Y, p, 1, eps=0.0, est_covs=True, est_var=True, fused=fused)
            ts.append(time.perf_counter() - t0)
        out[kind] = float(np.median(ts))
    try:
        from threadpoolctl import threadpool_info
        threads = max([d.get("num_threads", 1) for d in threadpool_info()] or [os.cpu_count()])
    except Exception:
        threads = os.cpu_count()
    full = rows == L
    scale = rows / float(L)
    return {
        "value": scale / out["faithful"], "unit": "sweeps/s", "cores": int(threads), "kind": "port",
        "sample": (f"all {L} rows x {M} cols" if full else f"first {rows} of {L} rows x {M} cols (extrapolated linearly in L: "
                   "over-charges the L-independent M x M product)") +
                  f", H={H}, fp64 NumPy/OpenBLAS oracle in the reference's operation order (src/vbmf.jl:95-157: three passes "
                  f"over Y, Y.^2, 2Y', the M x M product), median of {sweeps} sweeps after one warm-up = {out['faithful']:.3f} s per sweep",
        "fused_value": scale / out["fused"],
        "sample_seconds_per_sweep": out["faithful"],
        "rows": rows,
    }


def main():
    a = parse()
    # stdout carries exactly ONE line, the JSON result: libraries that print there (RCCL writes a version banner on
    # communicator creation) are sent to stderr for the duration of the run
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    # (multi-process GPU work on this pool needs dmabuf IPC; the image exports it already -- kept here for a bare environment)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world

    # torch first: the library then binds to the HIP/RCCL runtime torch already loaded (one runtime
    # per process); torch is plumbing here (rendezvous, barrier, max-over-ranks), never compute
    import torch
    import torch.distributed as dist
    # VBMF_BENCH_TRANSPORT=host: REHEARSAL of the N > 1 path on a one-GPU box -- every rank on device 0, rendezvous and the
    # library's all-reduces over gloo through host memory (dist.host_staged_transport).  Exercises this file's own multi-rank
    # logic (shard arithmetic, barrier, max over ranks, rank-0 JSON) and the library's collective code path; its number is NOT a
    # measurement of anything (the line says so).  RCCL over xGMI is the only transport of a real run.
    host_transport = world > 1 and os.environ.get("VBMF_BENCH_TRANSPORT", "") == "host"
    if world > 1 and host_transport:
        local_rank = 0
        torch.cuda.set_device(0)
        dist.init_process_group(backend="gloo")
    elif world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as G
    pkg = G.load_package()
    capi = pkg.capi

    L, M, H, ydt = CONFIGS[a.config]
    L, M, H = a.L or L, a.M or M, a.H or H
    ydt = a.ydtype or ydt
    y_dtype = capi.VBMF_Y_F32 if ydt == "f32" else capi.VBMF_Y_BF16
    f_dtype = {"auto": capi.VBMF_FACTOR_AUTO, "bf16": capi.VBMF_FACTOR_BF16, "bf16x2": capi.VBMF_FACTOR_BF16X2}[a.factor]
    if ydt == "f32":
        f_dtype = capi.VBMF_FACTOR_AUTO

    # row shard of this rank (equal counts, remainder to the first shards: SURVEY 8e)
    emu = a.shard_of if (a.shard_of > 1 and world == 1) else 0
    base, rem = divmod(L, emu or world)
    L_loc = base + (1 if rank < rem else 0)
    row0 = rank * base + min(rank, rem)

    sparse = (a.config == "cfg5")
    ctx = capi.Context(L_loc, M, H, y_dtype=y_dtype, factor_dtype=f_dtype, device=local_rank, nranks=world,
                       rank=rank, L_global=(L_loc if emu else L), row_offset=row0, pass1_splits=a.splits,
                       variant=capi.VBMF_VARIANT_SPARSE_DIAG if sparse else capi.VBMF_VARIANT_BASIC)
    if host_transport:
        ctx.comm_set_transport(pkg.dist.host_staged_transport(lambda arr: dist.all_reduce(torch.from_numpy(arr))))
    elif world > 1:
        uid = [capi.Context.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(uid[0])
    elif emu:
        ctx.comm_init(capi.Context.unique_id())

    ctx.set_Y_synthetic(20170101, H, 0.05)
    rng = np.random.default_rng(20170102)
    A0 = rng.standard_normal((M, H))
    B0 = rng.standard_normal((L, H))[row0:row0 + L_loc]
    z = np.zeros((H, H))
    if sparse:      # vbmf_sparse_init defaults (src/vbmf_sparse.jl:101-104) with ca = cb = sigma = 0.1 (examples/toy_data.jl:53)
        hyper = dict(alpha0=1e-10, beta0=1e-10, gamma0=1e-10, delta0=1e-10, eta0=1e-10, zeta0=1e-10)
        ctx.sparse_set_state(A0.reshape(M * H), np.ones(M * H), 0.1 * np.ones(M * H), 1e-10 * np.ones(M * H), B0, z,
                             0.1 * np.ones(H), 1e-10 * np.ones(H), 0.1, 1e-10, hyper)
        if a.full_cov:
            ctx.sparse_set_full_cov(True)
        run = lambda k: ctx.sparse_run(k, eps=0.0, est_cb=True)
    else:
        ctx.set_state(A0, B0, z, z, 0.1 * np.ones(H), 0.1 * np.ones(H), 0.1)
        run = lambda k: ctx.run(k, eps=0.0, est_covs=True, est_var=True)
    del A0, B0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    # clock settle: the device idles while the host prepares the initial state; untimed sweeps until it has been busy for
    # --settle-seconds, then the W warm-up sweeps, then EXACTLY K timed sweeps
    # (the settle phase launches the two streaming passes back to back through the library's tuning hook: the same kernels
    #  on the same data, but the model's state is not advanced, so W + K sweeps are all the sweeps the state has seen)
    settle_launches = 0
    if a.settle_seconds > 0:
        ts = time.perf_counter()
        while time.perf_counter() - ts < a.settle_seconds:
            ctx.time_pass(1, 8)
            ctx.time_pass(2, 8)
            settle_launches += 20                      # (each call launches its pass 2 + 8 times)
    if a.warmup > 0:
        run(a.warmup)
    ctx.profile_enable(max(1, a.event_stride))
    barrier()
    t0 = time.perf_counter()
    it, d, _ = run(a.steps)
    barrier()
    t1 = time.perf_counter()
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    assert it == a.steps, (it, a.steps)

    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if host_transport else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    chain = None if sparse else ctx.chain_us()
    s = {"sigma2": ctx.sparse_get_state(want_B=False)["sigmaHat"]} if sparse else ctx.get_state(want_B=False)
    bytes1, bytes2 = ctx.pass_bytes(1), ctx.pass_bytes(2)
    n = prof["pass1_n"] + prof["pass2_n"]
    avg_ms = (prof["pass1_ms"] + prof["pass2_ms"]) / max(n, 1)
    avg_bytes = (bytes1 * prof["pass1_n"] + bytes2 * prof["pass2_n"]) / max(n, 1)
    achieved = avg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM traffic per launch from PMC counters is collected in separate rocprofv3 passes
    # (scripts/pmc_traffic.sh); report the committed measurement when it is for exactly this workload
    traffic, traffic_source = None, None
    try:
        import glob
        pmf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_stream_kernel_cfg3.json")))[-1]
        pm = json.load(open(pmf))
        k = pm["config"]
        if (k["L"], k["M"], k["H"], k["y_dtype"], k["n_gpus"]) == (L, M, H, ydt, world) and a.factor in ("auto", "bf16x2"):
            traffic = pm["traffic_bytes_per_launch"]
            traffic_source = (f"profiles/{os.path.basename(pmf)} -- separate rocprofv3 --pmc passes of this workload "
                              "(scripts/pmc_traffic.sh), NOT measured in this run")
    except Exception:
        traffic, traffic_source = None, None
    # the other roof: MFMA work of one launch = 2*L*M*Hp flops per factor part (hi + lo in the bf16x2 mode); the pass
    # is priced against whichever roof it sits closer to (H <= 64: HBM; H >= 128 with hi+lo parts: MFMA)
    fop = "f32" if ydt == "f32" else ("bf16" if a.factor == "bf16" else "bf16x2")
    Hp = 32 if H <= 32 else (64 if H <= 64 else (128 if H <= 128 else 256))
    flops = 2.0 * L_loc * M * Hp * (2 if fop == "bf16x2" else 1)
    mfma_peak = 157.3 if ydt == "f32" else 2500.0            # TFLOP/s dense: exact-f32 MFMA / bf16 MFMA (MI355X_MICROARCH.md)
    mfma_achieved = flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    out = {
        "metric": "VB iterations/sec",
        "value": a.steps / elapsed,
        "unit": "sweeps/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / a.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "bf16" if ydt == "bf16" else "f32",
        "data": "synthetic",
        "config": {"workload": f"vbmf! sweep, est_covs=est_var=true, {L}x{M} dense rank-{H}, toy_matrix data "
                               f"(noise 0.05), Y {ydt} resident in HBM", "L": L, "M": M, "H": H,
                   "y_dtype": ydt, "factor_operand": fop,
                   "accumulate": "fp32", "hxh_algebra": "fp64", "row_shards": world},
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
            # whole sweep against the same roof: algorithmic bytes of both passes / wall time of a sweep (everything between
            # the passes included: slab sum, post kernels, reductions, launch gaps)
            "sweep_frac": (bytes1 + bytes2) / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBS,
            "kernel": "stream_gemm_kernel / (H >= 128, bf16x2) stream_lds8_kernel (pass 1: Y'B; pass 2: Y*A, at H <= 64 with the B update "
                      "+ Gram partials as its register epilogue; this rank's shard)",
            "bytes_per_launch": avg_bytes, "avg_launch_ms": avg_ms, "launches_timed": n,
            "launches": 2 * a.steps, "event_stride": max(1, a.event_stride),
            "pass1": {"ms": prof["pass1_ms"] / max(prof["pass1_n"], 1), "bytes": bytes1,
                      "GBps": bytes1 / max(prof["pass1_ms"] / max(prof["pass1_n"], 1), 1e-9) / 1e6},
            "pass2": {"ms": prof["pass2_ms"] / max(prof["pass2_n"], 1), "bytes": bytes2,
                      "GBps": bytes2 / max(prof["pass2_ms"] / max(prof["pass2_n"], 1), 1e-9) / 1e6},
        },
        "final": {"sigma2": s["sigma2"], "d": d},
        "clock_settle": {"seconds": a.settle_seconds, "untimed_pass_launches": settle_launches},
        "control_chain_us": chain,
    }
    # SURVEY 8(d) counts 2*L*M*H flops per contraction -- the factor's lo part (bf16x2) and the padding of H to the rank
    # class are work the precision / tiling choice added, not algorithmic work: both figures are printed
    alg_flops = 2.0 * L_loc * M * H
    alg_achieved = alg_flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    out["roofline"]["other_roof"] = {"bound": "mfma", "achieved": mfma_achieved, "peak": mfma_peak, "unit": "TFLOP/s",
                                     "frac": mfma_achieved / mfma_peak, "flops_per_launch": flops,
                                     "issued_note": "MFMA work issued: 2*L*M*Hp per factor part (hi + lo in the bf16x2 mode)",
                                     "algorithmic_flops_per_launch": alg_flops, "algorithmic_achieved": alg_achieved,
                                     "algorithmic_frac": alg_achieved / mfma_peak,
                                     "sustainable_note": "bare MFMA loops on random operands sustain 1.82 (32x32x16) / 2.06 (16x16x32) "
                                                         "PFLOP/s on this chip under its power management, not 2.5 "
                                                         "(profiles/r03_a_mfma_ceiling_probe.txt)"}
    if mfma_achieved / mfma_peak > achieved / HBM_PEAK_GBS:      # MFMA-bound pass: swap the two roofs
        hb = {k: out["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac")}
        mf = out["roofline"]["other_roof"]
        out["roofline"].update({k: mf[k] for k in ("bound", "achieved", "peak", "unit", "frac")})
        out["roofline"].update({k: mf[k] for k in ("flops_per_launch", "issued_note", "algorithmic_flops_per_launch", "algorithmic_achieved",
                                                   "algorithmic_frac", "sustainable_note")})
        out["roofline"]["other_roof"] = dict(hb, bytes_per_launch=avg_bytes)
    if host_transport:
        out["config"]["rehearsal"] = (f"{world} ranks on ONE GPU, all-reduces staged through host memory over gloo "
                                      "(VBMF_BENCH_TRANSPORT=host): a dry run of the multi-rank code path, NOT a measurement")
    if emu:
        out["config"]["emulation"] = (f"rank 0's share of a {emu}-rank strong-scaling run on one GPU ({L_loc} of {L} rows, "
                                      "1-rank RCCL communicator, L_global = the share): per-rank compute only, no inter-GPU latency")
    if sparse:
        out["metric"] = "VB iterations/sec (vbmf_sparse!, diagonal branch)"
        out["config"]["workload"] = out["config"]["workload"].replace("vbmf! sweep, est_covs=est_var=true", f"vbmf_sparse! sweep (full_cov={'true' if a.full_cov else 'false'}, diag_var=false, est_cb=true)")
    if rank == 0 and not a.no_cpu_baseline and world == 1 and not sparse:
        out["cpu_baseline"] = cpu_baseline(ctx, L, M, H, a.cpu_rows, a.cpu_sweeps, 20170102)
    elif rank == 0:
        out["cpu_baseline"] = None
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
