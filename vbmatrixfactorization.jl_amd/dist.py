"""Row-sharding helpers for the multi-GPU path (one process per GPU).

Y is partitioned by rows: rank g owns Y_g (L_g x M) and BHat_g (L_g x H) permanently; AHat, SigmaA,
SigmaB, CA, CB, sigma2 are replicated.  Per sweep the library issues exactly two collectives on its
compute stream (RCCL all-reduce, sum):
    1. the M x H partial  P_g = Y_g' BHat_g            (fp32, after the split-K slab sum)
    2. the packed Grams   [B_g'B_g | dB_g'dB_g]        (fp64, 2*Hp^2)
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
