"""GPU parity tests of round 2: trees wider than one wave's row (column groups), reference-built trees whose internal
node names collide (guard columns on the bucketed path), subtree shards built without the rest of the tree, one
full-size shard of BASELINE config 5, and the in-process multi-replica count reduction (RCCL).
Through the C ABI, against the CPU oracle; bit-exact."""
import numpy as np
import pytest

from oracle import pfq_format as fmt
from oracle import pfq_oracle as orc
from phagefilter_amd import BloomTree, PfqError, pack_reads
from phagefilter_amd.query import allreduce_counts
from test_gpu_parity import (RNG, _read_plan, check_query, gpu_tree, hits_of, make_reads, oracle_hits, oracle_tree,
                             rand_dna)

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------------------------------------------
# more than 2048 leaf+guard columns: the sliced matrix is cut into column groups, the frontier runs once per group
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_genomes", [2049, 4100])
def test_trees_wider_than_2048_columns(gpu, n_genomes):
    k, nbits, h = 21, 65521, 4
    genomes = [rand_dna(int(RNG.integers(150, 260))) for _ in range(n_genomes)]
    genomes[2050 % n_genomes] = genomes[3]                       # twins in different column groups
    genomes[n_genomes - 1] = genomes[7][:120] + genomes[n_genomes - 1][120:]
    ot, ids = oracle_tree(genomes, k, nbits, h)
    gt = gpu_tree(genomes, ids, k, nbits, h)
    info = gt.info()
    assert (info.n_leaves, info.superset_verified) == (n_genomes, 1)
    reads = make_reads(genomes, 500, 100, 150, k) + make_reads(genomes, 20, 5, 400, k) + [genomes[3][:150], genomes[7][:100]]
    for thr in (1.0, 0.5, 0.0, 0.97):
        assert check_query(gt, ot, reads, thr, path=0).path == 0
    for thr in (1.0, 0.5, 0.97):
        assert check_query(gt, ot, reads, thr, path=1).path == 1
    # pruned to internal "leaves" (still more than one group at depth 12 of the 4100-leaf tree)
    ot.prune(12)
    gt.prune_tree(12)
    assert [t for t, _ in gt.get_leaf_counts()] == [t for t, _ in ot.leaf_counts()]
    check_query(gt, ot, reads, 1.0, path=1)
    check_query(gt, ot, reads, 0.4, path=0)
    gt.close()


# ---------------------------------------------------------------------------------------------------------------
# Internal_Node_<u16> names collide in reference-built trees -> two nodes alias one .bf -> parent ⊇ child
# fails on some edges -> guard columns.  Those trees must stay on the bucketed (fast) path.
# ---------------------------------------------------------------------------------------------------------------
def _h4_tree(n_genomes, collisions, k=21, nbits=100003, h=5):
    genomes = [rand_dna(int(RNG.integers(300, 500))) for _ in range(n_genomes)]
    genomes[11] = genomes[10]
    ot, ids = oracle_tree(genomes, k, nbits, h)
    internal = [v for v in range(ot.n_nodes) if not ot.is_leaf(v)]
    picks = RNG.choice(len(internal), size=2 * collisions, replace=False)
    for a, b in zip(picks[:collisions], picks[collisions:]):
        a, b = internal[int(a)], internal[int(b)]
        # the earlier node's filter is replaced by a fresh one under the same key (bloom_tree.rs:294, cache.rs:83-87):
        # both nodes alias ONE file afterwards, here the later node's
        ot.bf_path[a] = ot.bf_path[b]
        ot.filter_of[a] = ot.filter_of[b]
    return genomes, ot, ids


def test_colliding_internal_names_stay_on_the_bucketed_path(gpu, tmp_path):
    genomes, ot, ids = _h4_tree(320, 32)                         # ~10 % of the 319 internal names collide
    d = str(tmp_path / "db")
    fmt.write_db(ot, d)
    gt = BloomTree.load(d)
    info = gt.info()
    assert info.superset_verified == 0 and info.n_leaves == 320 and info.n_filters < info.n_nodes
    reads = make_reads(genomes, 1500, 300, 150, 21) + make_reads(genomes, 40, 10, 600, 21)
    for thr in (1.0, 0.6, 0.9):
        st = check_query(gt, ot, reads, thr, path=1)
        assert st.path == 1, thr                                  # guards are certified as pairs of their own
        assert check_query(gt, ot, reads, thr, path=0).path == 0
    st = check_query(gt, ot, reads, 1.0, path=1)
    assert st.path == 1 and st.tile_mode == 1
    for key, val in (("PFQ_TILE", "0"), ("PFQ_RECORD_GB", "0"), ("PFQ_TILE_ENTRIES", "20000")):
        gt.set_option(key, val)
        try:
            assert check_query(gt, ot, reads, 1.0, path=1).path == 1
            if key != "PFQ_RECORD_GB":
                assert check_query(gt, ot, reads, 0.6, path=1).path == 1
        finally:
            gt.set_option(key, None)
    # block mode on a tree with guard columns: the guards of the candidates that stand are certified against the sliced matrix
    gt.set_option("PFQ_BLOCK", "1")
    for env in ({}, {"PFQ_TILE_ENTRIES": "20000"}):
        for key, val in env.items():
            gt.set_option(key, val)
        try:
            st = check_query(gt, ot, reads, 1.0, path=1)
            assert st.path == 1 and st.tile_mode == 2
        finally:
            for key in env:
                gt.set_option(key, None)
    gt.close()


def test_guards_and_column_groups_together(gpu, tmp_path):
    """A 2300-leaf tree with colliding names: guard columns live in the last column group, leaves in both."""
    genomes, ot, ids = _h4_tree(2300, 40, nbits=65521, h=4)
    d = str(tmp_path / "db")
    fmt.write_db(ot, d)
    gt = BloomTree.load(d)
    assert gt.info().superset_verified == 0
    reads = make_reads(genomes, 600, 100, 150, 21)
    for thr in (1.0, 0.7):
        assert check_query(gt, ot, reads, thr, path=1).path == 1
        check_query(gt, ot, reads, thr, path=0)
    gt.set_option("PFQ_BLOCK", "1")
    assert check_query(gt, ot, reads, 1.0, path=1).tile_mode == 2
    gt.close()


# ---------------------------------------------------------------------------------------------------------------
# BASELINE config 5: subtree shards built on the device without the rest of the tree
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_genomes,depth", [(13, 2), (32, 3), (21, 1), (5, 4)])
def test_balanced_subtree_build_equals_shard_of_whole_tree(gpu, n_genomes, depth):
    from hipbuf import DeviceBuffer
    k, nbits, h, glen = 21, 50021, 5, 400
    genomes_np = np.stack([np.frombuffer(orc.synth_genome(0x5EED0000 + i, glen), dtype=np.uint8) for i in range(n_genomes)])
    genomes_np[4] = genomes_np[n_genomes - 1]                     # a read hitting leaves of two different shards
    genomes = [g.tobytes() for g in genomes_np]
    ot, ids = oracle_tree(genomes, k, nbits, h)
    dg = DeviceBuffer.from_numpy(genomes_np.reshape(-1))
    reads = make_reads(genomes, 200, 40, 150, k, errors=False)
    seq, off = pack_reads(reads)
    n_shards = 0
    for thr in (1.0, 0.5):
        for v in range(ot.n_nodes):
            ot.mapped_reads[v] = 0
        ohits, _, _ = orc.query_batch(ot, reads, thr)
        want_counts, want_hits = ot.leaf_counts(), oracle_hits(ot, ohits)
        got_counts, got_hits, i = [], [], 0
        while True:
            try:
                sh = BloomTree.build_balanced_subtree_device(dg.ptr, glen, n_genomes, ids, k, nbits, h, 5, 10, depth, i)
            except PfqError as e:
                assert e.code == -1 and i > 0
                break
            info = sh.info()
            osh, first = orc.subtree_shard(ot, depth, i)
            assert (info.shard_first_leaf, info.tree_leaves, info.superset_verified) == (first, n_genomes, 1)
            assert [t for t, _ in sh.get_leaf_counts()] == [t for t, _ in osh.leaf_counts()]
            offs, leaves = sh.query_packed(seq, off, thr, want_hits=True)
            got_counts += sh.get_leaf_counts()
            got_hits += [(r, c + first) for r, c in hits_of(offs, leaves)]
            sh.close()
            i += 1
        n_shards = i
        assert got_counts == want_counts, (depth, thr)
        assert sorted(got_hits) == want_hits, (depth, thr)
    assert n_shards >= 2


def test_config5_one_full_size_shard_2048_of_16384_leaves(gpu):
    """BASELINE config 5 at its real geometry: shard 5 of the depth-3 frontier of the 16 384-leaf SBT (2048 leaves,
    4095 + 3 filters of 71 887 936 bits = 36.8 GB node-major + 18.4 GB sliced), built on the device from the counter-based
    genomes; reads drawn from ALL 16 384 genomes.  Size-independent properties (the oracle cannot hold this tree):
    every positive read of the shard's leaves hits its source leaf, reads of other shards and random reads hit (almost)
    nothing, counts are additive over a partition of the reads, independent of the query path, and idempotent."""
    from hipbuf import DeviceBuffer, synchronize
    from phagefilter_amd import _ffi
    L = _ffi.lib()
    n_g, glen, k, h, nbits, n_reads, shard = 16384, 50000, 21, 10, 71887936, 4 * 1024 * 1024, 5
    ids = [f"G{i:05d}" for i in range(n_g)]
    dg = DeviceBuffer(n_g * glen)
    _ffi.check(L.pfq_synth_genomes_device(dg.ptr, n_g, glen, 0x5EED0000, None))
    synchronize()
    gt = BloomTree.build_balanced_subtree_device(dg.ptr, glen, n_g, ids, k, nbits, h, 0x0123456789ABCDEF,
                                                 0xFEDCBA9876543210, 3, shard, 0.001, 5000000)
    info = gt.info()
    assert (info.n_leaves, info.shard_first_leaf, info.tree_leaves, info.superset_verified) == (2048, shard * 2048, n_g, 1)
    assert info.n_nodes == 2 * n_g - 1 and info.n_filters == 4095 + 3
    assert [t for t, _ in gt.get_leaf_counts()] == ids[shard * 2048:(shard + 1) * 2048]
    dr = DeviceBuffer(n_reads * 150 + 64)
    _ffi.check(L.pfq_synth_reads_device(dr.ptr, 0, n_reads, 150, dg.ptr, glen, n_g, 0x5EED1234, None))
    off = DeviceBuffer.from_numpy(np.arange(n_reads + 1, dtype=np.uint64) * 150)
    synchronize()
    pos, leaf = _read_plan(0x5EED1234, 0, n_reads, n_g)
    mine = pos & (leaf >= shard * 2048) & (leaf < (shard + 1) * 2048)
    expect = np.bincount(leaf[mine] - shard * 2048, minlength=2048)
    assert expect.sum() > n_reads // 20

    def counts(path, lo, hi):
        gt.reset_counts()
        gt.set_path(path)
        gt.query_device(dr.ptr + lo * 150, off.ptr, hi - lo, (hi - lo) * 150, 1.0, 0)
        synchronize()
        st = gt.last_stats()
        assert st.path == path
        return np.array([c for _, c in gt.get_leaf_counts()], dtype=np.int64)

    whole = counts(1, 0, n_reads)
    assert (whole >= expect).all()                                   # every positive read of the shard hits its leaf
    assert 0 <= int(whole.sum() - expect.sum()) <= n_reads // 100000  # foreign and random reads: Bloom false positives only
    half = n_reads // 2
    assert np.array_equal(counts(1, 0, half) + counts(1, half, n_reads), whole)          # additive
    thirds = [0, 1000003, 2500000, n_reads]
    assert np.array_equal(sum(counts(0, a, b) for a, b in zip(thirds, thirds[1:])), whole)  # path/block independent
    assert np.array_equal(counts(1, 0, n_reads), whole)                                  # idempotent
    gt.close()


# ---------------------------------------------------------------------------------------------------------------
# replicas behind one process: pfq_trees_allreduce_counts (what `phage_filter query --devices` ends with)
# ---------------------------------------------------------------------------------------------------------------
def test_replicas_on_one_device_allreduce_to_single_tree_counts(gpu):
    genomes = [rand_dna(700) for _ in range(9)]
    ot, ids = oracle_tree(genomes, 21, 60013, 6)
    reads = make_reads(genomes, 400, 80, 150, 21)
    orc.query_batch(ot, reads, 0.7, want_hits=False)
    want = ot.leaf_counts()
    reps = [gpu_tree(genomes, ids, 21, 60013, 6) for _ in range(3)]
    cuts = [0, 150, 151, len(reads)]                              # replica i classifies its own share of the reads
    for rep, a, b in zip(reps, cuts, cuts[1:]):
        seq, off = pack_reads(reads[a:b])
        rep.query_packed(seq, off, 0.7)
    import os
    os.environ["PFQ_RCCL_ALWAYS"] = "1"                           # one device needs no communicator: ask for a one-rank one
    try:
        ranks = allreduce_counts(reps)
    finally:
        del os.environ["PFQ_RCCL_ALWAYS"]
    assert ranks == 1                                             # librccl loaded, ncclCommInitAll + ncclAllReduce ran
    for rep in reps:
        assert rep.get_leaf_counts() == want                      # every replica holds the job's totals
    for rep, a, b in zip(reps, cuts, cuts[1:]):                   # again without RCCL (replicas of one device are added there)
        rep.reset_counts()
        seq, off = pack_reads(reads[a:b])
        rep.query_packed(seq, off, 0.7)
    assert allreduce_counts(reps) == 0
    for rep in reps:
        assert rep.get_leaf_counts() == want
    other = gpu_tree(genomes[:5], ids[:5], 21, 60013, 6)
    with pytest.raises(PfqError):
        allreduce_counts([reps[0], other])                        # not replicas of one database
    with pytest.raises(PfqError):
        allreduce_counts([reps[0], reps[0]])
    for t in reps + [other]:
        t.close()


def test_replicas_of_a_database_with_stored_counts(gpu, tmp_path):
    """A database saved after a query holds non-zero mapped_reads (pfq_tree_save writes the live counters back); the
    reference accumulates on the loaded value (query.rs:143).  Replicas of it must report stored + new — not
    replicas x stored + new — and reducing twice must change nothing."""
    genomes = [rand_dna(600) for _ in range(7)]
    ot, ids = oracle_tree(genomes, 21, 60013, 6)
    reads = make_reads(genomes, 300, 60, 150, 21)
    first, second = reads[:200], reads[200:]
    gt = gpu_tree(genomes, ids, 21, 60013, 6)
    seq, off = pack_reads(first)
    gt.query_packed(seq, off, 0.8)
    d = str(tmp_path / "db")
    gt.save(d)                                                    # mapped_reads of `first` are in tree.bin now
    gt.close()
    orc.query_batch(ot, first, 0.8, want_hits=False)
    stored = ot.leaf_counts()
    assert sum(c for _, c in stored) > 50
    orc.query_batch(ot, second, 0.8, want_hits=False)            # the oracle goes on counting on the same tree
    want = ot.leaf_counts()
    reps = [BloomTree.load(d) for _ in range(3)]
    for rep in reps:
        assert rep.get_leaf_counts() == stored
    cuts = [0, 40, 41, len(second)]
    for rep, a, b in zip(reps, cuts, cuts[1:]):
        seq, off = pack_reads(second[a:b])
        rep.query_packed(seq, off, 0.8)
    allreduce_counts(reps)
    for rep in reps:
        assert rep.get_leaf_counts() == want                      # stored once + every replica's new counts
    allreduce_counts(reps)                                        # again, no new queries: nothing changes
    for rep in reps:
        assert rep.get_leaf_counts() == want
    # the layout is rebuilt in between (a knob of the layout): what a replica counted since the last reduction survives
    extra = reads[:30]
    seq, off = pack_reads(extra)
    reps[1].query_packed(seq, off, 0.8)
    reps[1].set_option("PFQ_COARSE", "0")
    orc.query_batch(ot, extra, 0.8, want_hits=False)
    allreduce_counts(reps)
    for rep in reps:
        assert rep.get_leaf_counts() == ot.leaf_counts()
    # the multi-process hooks: deltas out, sum in
    from hipbuf import DeviceBuffer, synchronize
    n = len(ids)
    seq, off = pack_reads(reads[:25])
    reps[0].query_packed(seq, off, 0.8)
    reps[2].query_packed(seq, off, 0.8)
    orc.query_batch(ot, reads[:25], 0.8, want_hits=False)
    orc.query_batch(ot, reads[:25], 0.8, want_hits=False)
    bufs = [DeviceBuffer(8 * n) for _ in reps]
    for rep, b in zip(reps, bufs):
        rep.export_counts_delta(b.ptr)
    synchronize()
    deltas = [b.to_numpy(np.uint64) for b in bufs]
    assert int(deltas[1].sum()) == 0 and int(deltas[0].sum()) == int(deltas[2].sum()) > 0
    total = DeviceBuffer.from_numpy(deltas[0] + deltas[1] + deltas[2])
    for rep in reps:
        rep.import_counts_delta(total.ptr)
    synchronize()
    for rep in reps:
        assert rep.get_leaf_counts() == ot.leaf_counts()
    for rep in reps:
        rep.close()


# ---------------------------------------------------------------------------------------------------------------
# related genomes: a read passes many leaves (what a phage database is)
# ---------------------------------------------------------------------------------------------------------------
def test_family_workload_parity_300k_reads(gpu):
    """Families of 8 strains 0.2 % apart, 300 000 reads (>= 2^18: the bucketed path is chosen on its own): a positive read
    is a candidate for ~8 leaves, so the first call outgrows the pair buffer (the overflow is certified inline) and the
    following calls size it from what they saw.  Per-leaf counts and every per-read hit set equal the oracle's."""
    import os
    rng = np.random.default_rng(4242)
    n_fam, fam, glen, k, h, nbits, n_reads = 8, 8, 4000, 21, 7, (1 << 20) + 7, 300000
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genomes_np = np.empty((n_fam * fam, glen), dtype=np.uint8)
    for f in range(n_fam):
        base = rng.choice(acgt, glen)
        for j in range(fam):
            g = base.copy()
            mut = rng.random(glen) < 0.002
            g[mut] = rng.choice(acgt, int(mut.sum()))
            genomes_np[f * fam + j] = g
    genomes = [g.tobytes() for g in genomes_np]
    ot, ids = oracle_tree(genomes, k, nbits, h)
    gt = gpu_tree(genomes, ids, k, nbits, h)
    src = rng.integers(0, n_fam * fam, n_reads)
    off0 = rng.integers(0, glen - 150, n_reads)
    reads_np = genomes_np[src[:, None], off0[:, None] + np.arange(150)[None, :]]
    neg = rng.random(n_reads) < 0.4
    reads_np[neg] = rng.choice(acgt, (int(neg.sum()), 150))
    err = rng.random(reads_np.shape) < 0.002                       # a few sequencing errors
    reads_np[err] = rng.choice(acgt, int(err.sum()))
    seq = np.concatenate([reads_np.reshape(-1), np.zeros(16, dtype=np.uint8)])
    off = np.arange(n_reads + 1, dtype=np.uint64) * 150
    for thr in (1.0, 0.6):
        for v in range(ot.n_nodes):
            ot.mapped_reads[v] = 0
        ohits, _, _ = orc.query_batch_packed(ot, seq, off, thr, threads=min(32, os.cpu_count() or 8))
        want = np.array(oracle_hits(ot, ohits), dtype=np.int64).reshape(-1, 2)
        assert len(want) > 2 * n_reads                               # several leaves per positive read
        for call in range(4):
            gt.reset_counts()
            gt.set_path(-1)
            # call 0: the pair pipeline, forced (a read is a candidate for ~8 leaves: its first such call outgrows the pair
            # buffer and the overflow is certified inline); calls 1..3: the library's own choice — block mode from the very
            # first one (round 3: a call without history screens a sample of its own reads), at 0.6 with k-mer entries
            gt.set_option("PFQ_BLOCK", "0" if call == 0 else None)
            offs, leaves = gt.query_packed(seq, off, thr, want_hits=True)
            st = gt.last_stats()
            assert st.path == 1
            assert st.tile_mode == (1 if call == 0 else 2), (thr, call, st.tile_mode)
            assert gt.get_leaf_counts() == ot.leaf_counts(), (thr, call)
            got = np.stack([np.repeat(np.arange(n_reads), np.diff(offs).astype(np.int64)), leaves.astype(np.int64)], 1)
            assert np.array_equal(got, want), (thr, call)
    gt.close()


@pytest.mark.parametrize("n_genomes,k,nbits,h", [(13, 21, 100003, 5), (64, 21, 65536, 6), (130, 16, 262144, 10), (2300, 21, 65521, 4)])
def test_block_mode_forced(gpu, n_genomes, k, nbits, h):
    """Block mode (pairs = read x block of 8 leaves x candidate mask, one entry tests a probe for all candidates): forced with
    PFQ_BLOCK=1 on trees whose leaf count is no multiple of 8, with twins and near-twins inside and across blocks, through
    one and several column groups; also with too little room for the probe buckets (several passes; chunks larger than the
    buffer, which the fallback certifies against the sliced matrix) and with none at all."""
    genomes = [rand_dna(int(RNG.integers(200, 500))) for _ in range(n_genomes)]
    genomes[1] = genomes[0]                                          # twins in one block
    genomes[9 % n_genomes] = genomes[2]                              # twins in different blocks
    genomes[5] = genomes[4][:180] + genomes[5][180:]                 # near-twins
    genomes[n_genomes - 1] = genomes[3]
    ot, ids = oracle_tree(genomes, k, nbits, h)
    gt = gpu_tree(genomes, ids, k, nbits, h)
    reads = make_reads(genomes, 600, 200, 150, k) + make_reads(genomes, 10, 5, 420, k) + make_reads(genomes, 20, 5, k + 3, k)
    reads += [genomes[0][:150], genomes[2][10:170], genomes[4][:150], genomes[3][:k], b"ACGT", b""]
    gt.set_option("PFQ_BLOCK", "1")
    for env in ({}, {"PFQ_TILE_ENTRIES": "200000"}, {"PFQ_TILE_ENTRIES": "3000"}, {"PFQ_TILE_GB": "0"}):
        for key, val in env.items():
            gt.set_option(key, val)
        try:
            st = check_query(gt, ot, reads, 1.0, path=1)
            assert st.path == 1 and st.tile_mode == 2, (env, st.tile_mode)
        finally:
            for key in env:
                gt.set_option(key, None)
    # thresholds below 1: k-mer entries against the block tables, buckets by (block, mask)
    for env in ({}, {"PFQ_TILE_ENTRIES": "200000"}, {"PFQ_TILE_ENTRIES": "3000"}, {"PFQ_TILE_GB": "0"}):
        for key, val in env.items():
            gt.set_option(key, val)
        try:
            for thr in (0.5, 0.3, 0.9):
                assert check_query(gt, ot, reads, thr, path=1).tile_mode == 2, (env, thr)
        finally:
            for key in env:
                gt.set_option(key, None)
    # PFQ_BLOCK=0 forbids block mode
    gt.set_option("PFQ_BLOCK", "0")
    assert check_query(gt, ot, reads, 0.5, path=1).tile_mode == 1
    assert check_query(gt, ot, reads, 1.0, path=1).tile_mode == 1
    gt.close()


def test_prefix_certificates_and_round_overflow(gpu):
    """Thresholds below 1 through the LDS-tile passes with k-mer entries.  (a) The passes bin only a prefix of a read's
    k-mers (need + slack): reads whose errors sit in the prefix pass only thanks to their tail, or fail for good — both are
    left undecided by the prefix and counted in full by the record kernel.  (b) One leaf receives thousands of pairs of
    long reads: more rounds per chunk than the entries' tags can name, the rest of the chunk takes the fallback."""
    k, nbits, h = 21, 262144, 5
    genomes = [rand_dna(3000) for _ in range(12)]
    ot, ids = oracle_tree(genomes, k, nbits, h)
    gt = gpu_tree(genomes, ids, k, nbits, h)
    gt.set_option("PFQ_BLOCK", "0")   # (the pair pipeline is what this test is about)
    acgt = b"ACGT"

    def mutate(read, positions):
        r = bytearray(read)
        for p_ in positions:
            r[p_] = acgt[(acgt.index(bytes([r[p_]])) + 1) % 4]
        return bytes(r)

    reads = []
    for i in range(400):
        g = genomes[i % 12]
        o = int(RNG.integers(0, 3000 - 150))
        rd = g[o:o + 150]
        reads.append(rd)                                           # clean
        reads.append(mutate(rd, [10, 30, 50, 70]))                 # prefix ruined, tail of 59 clean k-mers: passes at 0.3 only
        reads.append(mutate(rd, [10, 30, 50, 70, 90, 110, 130]))   # ruined throughout: fails
        reads.append(mutate(rd, [100, 120, 140]))                  # tail ruined, prefix clean: decided by the prefix
    for thr in (0.3, 0.45, 0.7):
        st = check_query(gt, ot, reads, thr, path=1)
        assert st.path == 1 and st.tile_mode == 1
    # (b) 3000 reads of 600 bp from one genome: 3000 pairs in one leaf's bucket, 580 k-mers each
    long_reads = []
    for i in range(3000):
        o = int(RNG.integers(0, 3000 - 600))
        rd = genomes[3][o:o + 600]
        long_reads.append(mutate(rd, [int(x) for x in RNG.integers(0, 600, 3)]) if i % 3 == 0 else rd)
    long_reads += [genomes[5][:600], genomes[6][100:700]]
    for thr in (0.5, 0.9):
        st = check_query(gt, ot, long_reads, thr, path=1)
        assert st.path == 1 and st.tile_mode == 1 and st.n_fallback_pairs > 0
    gt.close()


def test_harness_geometry_10010_genomes_on_one_gpu(gpu):
    """The reference's own benchmark configuration names a 10 010-genome database (benchmarking/config.yaml:2) at the
    harness's filter geometry (--false-pos-rate 0.00001 --largest-genome 500000: 11 981 322 bits, 17 hashes, k = 20;
    bench/tools/phage_filter.py:84-85): 20 019 filters = 30 GB node-major + 5 column groups of the sliced matrix.  It must
    open and classify on one GPU.  Size-independent properties (the oracle cannot hold it): every positive read hits its
    source leaf, nothing else is hit beyond Bloom false positives, counts are additive and independent of the path."""
    from hipbuf import DeviceBuffer, synchronize
    from phagefilter_amd import _ffi
    L = _ffi.lib()
    n_g, glen, k, n_reads = 10010, 5000, 20, 1 << 20
    nbits = orc.needed_bits(0.00001, 500000)
    h = orc.optimal_num_hashes(nbits, 500000)
    assert (nbits, h) == (11981322, 17)
    ids = [f"G{i:05d}" for i in range(n_g)]
    dg = DeviceBuffer(n_g * glen)
    _ffi.check(L.pfq_synth_genomes_device(dg.ptr, n_g, glen, 0x5EED0000, None))
    synchronize()
    gt = BloomTree.build_balanced_device(dg.ptr, glen, n_g, ids, k, nbits, h, 0x0123456789ABCDEF, 0xFEDCBA9876543210,
                                         0.00001, 500000)
    info = gt.info()
    assert (info.n_nodes, info.n_leaves, info.superset_verified) == (2 * n_g - 1, n_g, 1)
    dr = DeviceBuffer(n_reads * 150 + 64)
    _ffi.check(L.pfq_synth_reads_device(dr.ptr, 0, n_reads, 150, dg.ptr, glen, n_g, 0x5EED1234, None))
    off = DeviceBuffer.from_numpy(np.arange(n_reads + 1, dtype=np.uint64) * 150)
    synchronize()
    pos, leaf = _read_plan(0x5EED1234, 0, n_reads, n_g)
    expect = np.bincount(leaf[pos], minlength=n_g)

    def counts(path, lo, hi, thr=1.0):
        gt.reset_counts()
        gt.set_path(path)
        gt.query_device(dr.ptr + lo * 150, off.ptr, hi - lo, (hi - lo) * 150, thr, 0)
        synchronize()
        assert gt.last_stats().path == path
        return np.array([c for _, c in gt.get_leaf_counts()], dtype=np.int64)

    whole = counts(1, 0, n_reads)
    assert (whole >= expect).all() and 0 <= int(whole.sum() - expect.sum()) <= 16
    half = n_reads // 2
    assert np.array_equal(counts(1, 0, half) + counts(1, half, n_reads), whole)
    assert np.array_equal(counts(0, 0, 300001) + counts(0, 300001, n_reads), whole)
    loose = counts(1, 0, n_reads, 0.3)                                # the harness's own threshold (config.yaml:4)
    assert (loose >= whole).all() and int(loose.sum() - whole.sum()) <= n_reads // 1000
    assert np.array_equal(counts(0, 0, n_reads, 0.3), loose)
    gt.close()


@pytest.mark.parametrize("seed", [int(__import__("os").environ.get("PFQ_PARITY_SEED0", "0")) + i
                                  for i in range(int(__import__("os").environ.get("PFQ_GUARD_SEEDS", "6")))])
def test_randomized_parity_with_colliding_names(gpu, tmp_path, seed):
    """Random trees in which some internal nodes alias another node's filter or lost bits (what colliding
    Internal_Node_<u16> names do), read-length mixes from k to 20 kb, thresholds, both paths and forced
    bucket-buffer sizes: guard pairs on the bucketed path against the oracle's full traversal."""
    import os
    rng = np.random.default_rng(7000 + seed)
    n_genomes = int(rng.choice([5, 33, 130, 400, 2100]))
    k = int(rng.choice([15, 21, 31]))
    h = int(rng.choice([3, 7, 10]))
    nbits = int(rng.choice([40009, 131072, 1 << 20]))
    glen = int(rng.integers(max(k + 5, 80), 600))

    def dna(n):
        return bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n).astype(np.uint8))

    base = [dna(glen) for _ in range(max(2, n_genomes // int(rng.choice([1, 4]))))]
    genomes = []
    for i in range(n_genomes):
        g = bytearray(base[int(rng.integers(0, len(base)))])
        for _ in range(int(rng.integers(0, 4))):
            g[int(rng.integers(0, len(g)))] = ord("ACGT"[int(rng.integers(0, 4))])
        genomes.append(bytes(g))
    ot, ids = oracle_tree(genomes, k, nbits, h)
    internal = [v for v in range(ot.n_nodes) if not ot.is_leaf(v)]
    n_bad = max(1, len(internal) // int(rng.choice([3, 10, 40])))
    for v in rng.choice(internal, size=min(n_bad, len(internal)), replace=False):
        v = int(v)
        if rng.random() < 0.5 and len(internal) > 1:
            w = int(rng.choice(internal))
            ot.bf_path[v], ot.filter_of[v] = ot.bf_path[w], ot.filter_of[w]
        else:
            ot.bits[ot.filter_of[v]][::int(rng.choice([2, 5, 50]))] = 0
    d = str(tmp_path / "db")
    fmt.write_db(ot, d)
    reads = []
    for _ in range(int(rng.integers(150, 400))):
        src = genomes[int(rng.integers(0, n_genomes))] * int(rng.choice([1, 1, 4, 40]))
        L = int(min(len(src), rng.choice([k, k + 1, 100, 150, 151, 300, 2000, 20000])))
        o = int(rng.integers(0, len(src) - L + 1))
        r = bytearray(src[o:o + L])
        for _ in range(int(rng.choice([0, 0, 1, 4]))):
            r[int(rng.integers(0, L))] = ord("ACGTN"[int(rng.integers(0, 5))])
        reads.append(bytes(r) if rng.random() < 0.5 else orc.revcomp(bytes(r)))
    reads += [dna(int(rng.integers(0, 300))) for _ in range(40)] + [b"", dna(k - 1)]
    entries = int(rng.choice([0, 0, 30_000, 400_000]))
    if entries:
        os.environ["PFQ_TILE_ENTRIES"] = str(entries)
    if seed % 2:
        os.environ["PFQ_TILE_COUNTS"] = "0"   # (record kernel alone; default: tile passes with k-mer entries)
    if seed % 3 == 0:
        os.environ["PFQ_BLOCK"] = "1"         # (block mode with guard columns)
    try:
        gt = BloomTree.load(d)
        for thr in (1.0, float(rng.choice([0.1, 0.5, 0.9])), float(rng.choice([0.0, 0.75, 0.999]))):
            for path in (1, 0):
                check_query(gt, ot, reads, thr, path=path)
        gt.close()
    finally:
        os.environ.pop("PFQ_TILE_ENTRIES", None)
        os.environ.pop("PFQ_TILE_COUNTS", None)
        os.environ.pop("PFQ_BLOCK", None)
