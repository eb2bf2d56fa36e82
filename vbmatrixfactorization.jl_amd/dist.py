"""Row-sharding helpers for the multi-GPU path (one process per GPU).

Y is partitioned by rows: rank g owns Y_g (L_g x M) and BHat_g (L_g x H) permanently; AHat, SigmaA,
SigmaB, CA, CB, sigma2 are replicated.  Per sweep the library issues exactly two collectives on its
compute stream (RCCL all-reduce, sum):
    1. the M x H partial  P_g = Y_g' BHat_g            (fp32, after the split-K slab sum)
    2. the packed sums    [B_g'B_g | dB_g'dB_g | sum (Y_g A) o B_g]   (fp64, 2*Hp^2 + 1; the last is this rank's part of
       tr(Y'BA') of updateSigma2!, src/vbmf.jl:154)
plus one all-reduce of ||Y_g||^2 at set-up.  L x H data never moves.
"""


def row_shard(L, world, rank):
    """(first_row, n_rows) of `rank`: equal counts, the remainder goes to the first shards."""
    if not (0 <= rank < world) or L < world:
        raise ValueError("bad shard request")
    base, rem = divmod(L, world)
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


def init_comm(ctx, rank, world, broadcast_bytes):
    """Create the RCCL communicator of `ctx`.  `broadcast_bytes(b_or_None) -> bytes` must return rank 0's
    bytes on every rank (e.g. torch.distributed.broadcast_object_list)."""
    from . import capi
    if world == 1:
        return
    uid = capi.Context.unique_id() if rank == 0 else None
    uid = broadcast_bytes(uid)
    ctx.comm_init(uid)


def host_staged_transport(all_reduce_numpy):
    """All-reduce transport that stages through host memory: for bring-up and for tests that run several ranks
    on ONE GPU (RCCL refuses two ranks on a device).  `all_reduce_numpy(a)` must sum the 1-D numpy array `a`
    over the ranks in place (e.g. a gloo torch.distributed.all_reduce on torch.from_numpy(a)).  Returns a
    function for Context.comm_set_transport."""
    import ctypes as C
    import numpy as np
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]

    def fn(buf, count, is_double, stream):
        # Both copies go on the LIBRARY's stream and are waited for there.  (A plain hipMemcpy runs on the null stream, which
        # the library's non-blocking streams are not ordered against, and a host-to-device copy from pageable memory may
        # return before its DMA has landed: the next kernel on `stream` could then read the buffer too early -- seen as a
        # once-in-several-runs mismatch of the replicated factor across ranks.)
        a = np.empty(count, dtype=np.float64 if is_double else np.float32)
        if hip.hipMemcpyAsync(a.ctypes.data, buf, a.nbytes, 2, stream) != 0:      # hipMemcpyDeviceToHost, after the producers
            return 3
        if hip.hipStreamSynchronize(stream) != 0:
            return 2
        all_reduce_numpy(a)
        if hip.hipMemcpyAsync(buf, a.ctypes.data, a.nbytes, 1, stream) != 0:      # hipMemcpyHostToDevice, before the consumers
            return 4
        if hip.hipStreamSynchronize(stream) != 0:                                 # (`a` must outlive the copy)
            return 2
        return 0
    return fn
