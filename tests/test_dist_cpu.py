"""N > 1 path on CPU: world-size-2 gloo run of the read sharding + count reduction (phagefilter_amd/dist.py).
Each rank classifies its shard with the CPU oracle (standing in for its GPU, which this container lacks), the
counters are all-reduced, and the result must equal the single-process run over all reads."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_partition():
    from phagefilter_amd.dist import shard_bounds
    for n in (0, 1, 7, 100, 1001):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import pfq_oracle as orc
    from phagefilter_amd.dist import all_reduce_counts, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    genomes_np = np.stack([np.frombuffer(orc.synth_genome(0x5EED0000 + i, 1500), dtype=np.uint8) for i in range(6)])
    genomes = [g.tobytes() for g in genomes_np]
    ids = [f"G{i:05d}" for i in range(6)]
    tree = orc.build_balanced_tree(genomes, ids, 21, 300007, 7, 5, 10)          # replica on every rank
    n_reads = 1001
    reads = orc.synth_reads(0x5EED1234, 0, n_reads, 150, genomes_np, 1500)
    lo, hi = shard_bounds(n_reads, rank, world)
    orc.query_batch(tree, [r.tobytes() for r in reads[lo:hi]], 1.0, want_hits=False)
    counts = torch.tensor([c for _, c in tree.leaf_counts()], dtype=torch.int64)
    all_reduce_counts(counts)
    if rank == 0:
        np.save(out_path, counts.numpy())
    dist.destroy_process_group()


def test_world2_gloo_counts_equal_single_process(tmp_path):
    import torch.multiprocessing as mp
    from oracle import pfq_oracle as orc
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "counts.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    genomes_np = np.stack([np.frombuffer(orc.synth_genome(0x5EED0000 + i, 1500), dtype=np.uint8) for i in range(6)])
    tree = orc.build_balanced_tree([g.tobytes() for g in genomes_np], [f"G{i:05d}" for i in range(6)], 21, 300007, 7, 5, 10)
    reads = orc.synth_reads(0x5EED1234, 0, 1001, 150, genomes_np, 1500)
    orc.query_batch(tree, [r.tobytes() for r in reads], 1.0, want_hits=False)
    assert list(got) == [c for _, c in tree.leaf_counts()]
    assert got.sum() > 400
