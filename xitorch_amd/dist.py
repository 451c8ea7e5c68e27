"""Multi-GPU helpers: batch sharding + the tiny per-iteration exchanges (RCCL over xGMI).

The reference is single-device.  Here the batched-operator dimension is sharded, one process per
GPU (torch.distributed, backend "nccl" == RCCL on ROCm).  Operators stay resident on their GPU; the
only traffic is what reproduces the reference's *global* decisions:

  davidson            1 all-reduce(MAX) of {max|resid|, breakdown flag} per iteration (symeig.py:188)
  cg/bicgstab/gmres   1 all-reduce(MAX) of {max residual norm, "someone unconverged"} per iteration
                      (solve.py:157,166,301,310)
  broyden             all-reduce(SUM) of each inner product / norm and of the rank-vector of the
                      multi-dot, because the whole batch is ONE flat system (rootsolver.py:67-76)

All messages are a few bytes to a few hundred bytes: latency-bound, so they are fused into one
call per decision and never sit on the bandwidth path.
"""
import torch

__all__ = ["shard_range", "allreduce_max_", "allreduce_sum_", "is_distributed", "all_ranks_agree_true"]


def is_distributed():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def shard_range(total, world_size, rank):
    """Contiguous block partition of `total` batch members: returns (start, stop) of `rank`.
    The first `total % world_size` ranks get one extra member."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad world_size/rank: %d/%d" % (world_size, rank))
    base, extra = divmod(total, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def _trivial(group):
    return group is None or torch.distributed.get_world_size(group) == 1


def allreduce_max_(t, group=None):
    """In-place MAX over the group (no-op when group is None or has a single rank)."""
    if not _trivial(group):
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=group)
    return t


def allreduce_sum_(t, group=None):
    """In-place SUM over the group (no-op when group is None or has a single rank)."""
    if not _trivial(group):
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=group)
    return t


def all_ranks_agree_true(flag, device, group=None):
    """True only when `flag` is true on EVERY rank of the group (one tiny all-reduce).  Shortcuts that skip a
    solver loop containing collectives must be taken by all ranks or by none: a rank that returns early on
    its own (e.g. its shard of the right-hand side is zero — common in an implicit backward where some batch
    members receive no gradient) would leave the others blocked in the loop's all-reduce."""
    if _trivial(group):
        return bool(flag)
    t = torch.tensor([0.0 if flag else 1.0], dtype=torch.float64, device=device)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=group)
    return bool(t.item() == 0.0)
