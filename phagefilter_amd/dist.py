"""Read sharding and the per-genome count reduction for multi-GPU runs (one process per GPU).

The path shards by reads: a read's classification is independent of every other read (query.rs:113-117 is a pure
per-read filter), the tree is replicated in every GPU's HBM, and the only exchange is one all-reduce (SUM) of the
u64[n_leaves] counters at the end (RCCL over xGMI with the "nccl" backend; "gloo" in the CPU tests)."""
from __future__ import annotations

from typing import Tuple


def shard_bounds(n_reads: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank `rank`: sizes differ by at most one, union = [0, n_reads)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_reads, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_reduce_counts(counts):
    """Sum the per-leaf counters over all ranks in place (torch tensor, int64; the leaf order is identical on
    every rank because every rank holds the same tree).  No-op without an initialised process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts


def reduce_tree_counts(tree, device=None, stream: int = 0):
    """Export a tree's device counters, all-reduce them, import the global counts back (every rank ends with the
    whole job's counts; rank 0 writes CLASSIFICATION.csv)."""
    import torch
    n = int(tree.info().n_leaves)
    buf = torch.zeros(max(n, 1), dtype=torch.int64, device=device if device is not None else f"cuda:{tree.device}")
    tree.export_counts(buf.data_ptr(), stream)
    torch.cuda.current_stream().synchronize()
    all_reduce_counts(buf)
    tree.import_counts(buf.data_ptr(), stream)
    torch.cuda.current_stream().synchronize()
    return buf


def gather_shard_counts(tree, device=None, stream: int = 0):
    """Subtree-sharded trees (BloomTree.load_subtree, one shard per rank, every rank classifies all reads): the
    shards' leaf ranges are disjoint, so the whole tree's counters are one all-reduce of a zero-padded vector in
    which each rank fills its own range [shard_first_leaf, shard_first_leaf + n_leaves)."""
    import torch
    info = tree.info()
    n, first, total = int(info.n_leaves), int(info.shard_first_leaf), int(info.tree_leaves)
    dev = device if device is not None else f"cuda:{tree.device}"
    full = torch.zeros(max(total, 1), dtype=torch.int64, device=dev)
    if n:
        local = torch.zeros(n, dtype=torch.int64, device=dev)
        tree.export_counts(local.data_ptr(), stream)
        torch.cuda.current_stream().synchronize()
        full[first:first + n] = local
    all_reduce_counts(full)
    return full
