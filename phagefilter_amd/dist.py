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
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts


def _sync_stream(stream: int) -> None:
    """Wait for the HIP stream the counters were exported / imported on (0 = the default stream)."""
    import torch
    if stream:
        torch.cuda.ExternalStream(stream).synchronize()
    else:
        torch.cuda.default_stream().synchronize()


def reduce_tree_counts(tree, device=None, stream: int = 0):
    """Export a tree's device counters, all-reduce them, import the global counts back (every rank ends with the
    whole job's counts; rank 0 writes CLASSIFICATION.csv).  `stream` is the raw HIP stream of the query calls."""
    import torch
    n = int(tree.info().n_leaves)
    buf = torch.zeros(max(n, 1), dtype=torch.int64, device=device if device is not None else f"cuda:{tree.device}")
    torch.cuda.current_stream().synchronize()   # the fill (torch's stream) before the export (the raw stream) writes the buffer
    # what THIS rank counted (counters - what the database was opened with): stored counts must not be added once per rank
    tree.export_counts_delta(buf.data_ptr(), stream)
    _sync_stream(stream)          # the copy must have landed before the collective (another stream) reads the buffer
    all_reduce_counts(buf)
    torch.cuda.current_stream().synchronize()
    tree.import_counts_delta(buf.data_ptr(), stream)   # counters = stored + the job's new counts, on every rank
    _sync_stream(stream)
    return buf


def check_disjoint_shards(first: int, n: int, total: int) -> None:
    """Every rank's leaf range [first, first + n) must lie inside [0, total) and the ranges must not overlap: an
    all-gather of the (first, n) pairs, checked on every rank.  No-op without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        if not 0 <= first <= first + n <= total:
            raise ValueError(f"shard range [{first}, {first + n}) outside the tree's {total} leaves")
        return
    # gather first, validate afterwards: every rank sees every span and raises (or not) together — a rank that left
    # before the collective would leave the others waiting in it
    mine = torch.tensor([first, n, total], dtype=torch.int64)
    if dist.get_backend() == "nccl":
        mine = mine.cuda()
    spans = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(spans, mine)
    spans = sorted((int(s[0]), int(s[1]), int(s[2])) for s in spans)
    if any(s[2] != total for s in spans):
        raise ValueError(f"ranks disagree on the tree's leaf count: {spans}")
    for f0, fn, ft in spans:
        if not 0 <= f0 <= f0 + fn <= ft:
            raise ValueError(f"shard range [{f0}, {f0 + fn}) outside the tree's {ft} leaves")
    for (a0, an, _), (b0, _, _) in zip(spans, spans[1:]):
        if an and a0 + an > b0:
            raise ValueError(f"shard leaf ranges overlap: {spans}")


def pad_and_reduce(local, first: int, total: int):
    """Whole-tree counters from one shard's: a zero vector of the tree's `total` leaves with this rank's counters at
    [first, first + len(local)), summed over the ranks by ONE all-reduce (the shards' ranges are disjoint, so the sum is
    the concatenation; a shard held by no rank stays zero)."""
    import torch
    n = int(local.numel())
    check_disjoint_shards(first, n, total)
    full = torch.zeros(max(total, 1), dtype=torch.int64, device=local.device)
    if n:
        full[first:first + n] = local
    all_reduce_counts(full)
    return full


def gather_shard_counts(tree, device=None, stream: int = 0):
    """Subtree-sharded trees (BloomTree.load_subtree / build_balanced_subtree_device, one shard per rank, every rank
    classifies all reads): the whole tree's counters on every rank.  `stream` is the raw HIP stream of the query calls."""
    import torch
    info = tree.info()
    n, first, total = int(info.n_leaves), int(info.shard_first_leaf), int(info.tree_leaves)
    dev = device if device is not None else f"cuda:{tree.device}"
    local = torch.zeros(n, dtype=torch.int64, device=dev)
    if n:
        torch.cuda.current_stream().synchronize()   # the fill before the export on the raw stream
        tree.export_counts(local.data_ptr(), stream)
        _sync_stream(stream)
    return pad_and_reduce(local, first, total)
