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


# ---------------------------------------------------------------------------------------------------------------
# subtree shards (BASELINE config 5): every rank owns a different shard, classifies ALL reads, one all-reduce
# ---------------------------------------------------------------------------------------------------------------
def _shard_setup():
    from oracle import pfq_oracle as orc
    rng = np.random.default_rng(5)
    genomes = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(rng.integers(500, 900))).astype(np.uint8))
               for _ in range(11)]
    genomes[6] = genomes[2]                                    # a read that hits leaves of two different shards
    ids = [f"G{i:05d}" for i in range(len(genomes))]
    tree = orc.build_balanced_tree(genomes, ids, 21, 200003, 7, 5, 10)
    tree.bits[tree.filter_of[tree.root]][::40] = 0              # a root that is no superset: ancestors must be honoured
    reads = []
    for i in range(600):
        g = genomes[int(rng.integers(0, len(genomes)))]
        o = int(rng.integers(0, len(g) - 150))
        reads.append(g[o:o + 150] if i % 3 else bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 150).astype(np.uint8)))
    return orc, tree, reads


def _shard_worker(rank, world, port, out_path, depth):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from phagefilter_amd.dist import pad_and_reduce
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc, tree, reads = _shard_setup()
    total = len(tree.leaves_dfs())
    shard, first = orc.subtree_shard(tree, depth, rank)        # rank r owns shard r of the depth-1 frontier
    orc.query_batch(shard, reads, 0.6, want_hits=False)        # every rank sees ALL reads
    local = torch.tensor([c for _, c in shard.leaf_counts()], dtype=torch.int64)
    full = pad_and_reduce(local, first, total)
    if rank == 1:                                              # any rank holds the whole job's counters
        np.save(out_path, full.numpy())
    # overlapping shard ranges are refused on every rank
    refused = False
    try:
        pad_and_reduce(local, 0, total)                        # both ranks claim to start at leaf 0
    except ValueError:
        refused = True
    assert refused
    dist.destroy_process_group()


def test_world2_gloo_subtree_shards_equal_whole_tree(tmp_path):
    """dist.pad_and_reduce (the core of gather_shard_counts): two ranks, each with a different subtree shard of the
    same tree and all reads; the reduced zero-padded vector equals the whole tree's counters."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "shard_counts.npy")
    mp.spawn(_shard_worker, args=(2, port, out, 1), nprocs=2, join=True)
    got = np.load(out)
    orc, tree, reads = _shard_setup()
    orc.query_batch(tree, reads, 0.6, want_hits=False)
    want = [c for _, c in tree.leaf_counts()]
    assert list(got) == want
    assert sum(want) > 100


def test_oracle_subtree_shards_concatenate():
    """oracle.subtree_shard: the shards of any depth partition the leaves in order and their hit sets concatenate to the
    whole tree's (the CPU twin of tests/test_gpu_parity.py::test_subtree_shards_concatenate_to_whole_tree)."""
    orc, tree, reads = _shard_setup()
    for thr in (1.0, 0.5):
        for v in range(tree.n_nodes):
            tree.mapped_reads[v] = 0
        orc.query_batch(tree, reads, thr, want_hits=False)
        want = tree.leaf_counts()
        for depth in (0, 1, 2, 3, 5):
            got, i = [], 0
            while True:
                try:
                    sh, first = orc.subtree_shard(tree, depth, i)
                except IndexError:
                    break
                assert first == len(got)
                orc.query_batch(sh, reads, thr, want_hits=False)
                got += sh.leaf_counts()
                i += 1
            assert got == want, (thr, depth)
