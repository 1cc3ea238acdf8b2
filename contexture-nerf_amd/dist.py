"""One process per GPU (torchrun), views sharded across ranks, RCCL over xGMI for the two exchange steps of the
path: all-reduce(MAX) of the per-face view-weight maxima [F] and all-reduce(SUM) of the atlas
contribution [3+1, T, T].  The reference's only parallelism is nn.DataParallel over the MLP batch
(src/training/trainer.py:134-135); this replaces it with view sharding.  `gloo` works for CPU rehearsal."""
import os
import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*). Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        local = local % max(torch.cuda.device_count(), 1)      # rehearsal: several ranks may share one card
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend or ("nccl" if device.type == "cuda" else "gloo"), rank=rank, world_size=world)
    return rank, world, device


def shard_views(n_views, rank, world):
    """View k -> rank k mod world (6-8 views on up to 8 ranks; extra ranks idle for this mesh)."""
    return [k for k in range(n_views) if k % world == rank]


def all_reduce_max_(t, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


def all_reduce_sum_(t, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def merge_atlas(contrib, group=None, eps=1e-8, frac_bits=32):
    """contrib [C+1,T,T] = per-rank weighted colour sums + weight sum -> atlas [C,T,T], coverage [T,T].
    int64 contrib (units of 2^-frac_bits, what `ConTEXTure.project_back_scatter` accumulates): the ranks' shards are summed as
    INTEGERS and converted once, so the result does not depend on how the views were dealt over ranks — the N-rank atlas is
    bit-identical to the 1-rank atlas.  A float32 contrib is summed as floats (order-dependent in the last bit)."""
    all_reduce_sum_(contrib, group)
    if contrib.dtype == torch.int64:
        # int64 -> double (round to nearest) -> exact power-of-two scale -> float32: the same three steps as ctx_fixed_to_float
        contrib = (contrib.to(torch.float64) * (2.0 ** -frac_bits)).to(torch.float32)
    w = contrib[-1:]
    return contrib[:-1] / w.clamp_min(eps), w[0]
