"""GPU parity tests: the HIP path (through the C ABI of libpfq) against the CPU oracle, bit-exact.

Everything here is integer / byte / index work, so the bar is equality: k-mer probe indices, filter words,
per-leaf counts, per-read hit sets and CLASSIFICATION.csv bytes.  Run with `pytest -m gpu` on an MI355X."""
import os

import numpy as np
import pytest

from oracle import pfq_format as fmt
from oracle import pfq_oracle as orc
from phagefilter_amd import BloomTree, PfqError, ResultMap, pack_reads, query_batch

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(20240601)


def rand_dna(n, alphabet=b"ACGT"):
    return RNG.choice(np.frombuffer(alphabet, dtype=np.uint8), int(n)).astype(np.uint8).tobytes()


def make_reads(genomes, n_pos, n_neg, length, k, *, errors=True):
    reads = []
    for i in range(n_pos):
        g = genomes[int(RNG.integers(0, len(genomes)))]
        L = int(min(length, len(g)))
        o = int(RNG.integers(0, len(g) - L + 1))
        r = bytearray(g[o:o + L])
        if errors and i % 4 == 1 and L > 3:
            r[int(RNG.integers(0, L))] = ord("N")
        if errors and i % 7 == 2 and L > 3:
            p = int(RNG.integers(0, L))
            r[p] = ord("ACGT"[(b"ACGT".find(bytes([r[p]])) + 1) % 4]) if bytes([r[p]]) in b"ACGT" else r[p]
        if i % 3 == 0:
            r = bytearray(orc.revcomp(bytes(r)))
        if errors and i % 11 == 5:
            r = bytearray(bytes(r).lower())
        reads.append(bytes(r))
    reads += [rand_dna(length) for _ in range(n_neg)]
    reads += [b"", b"A", rand_dna(max(k - 1, 0)), rand_dna(k), rand_dna(k + 1)]
    order = RNG.permutation(len(reads))
    return [reads[i] for i in order]


def oracle_tree(genomes, k, nbits, h, seeds=(5, 10)):
    ids = [f"G{i:05d}" for i in range(len(genomes))]
    return orc.build_balanced_tree(genomes, ids, k, nbits, h, seeds[0], seeds[1]), ids


def gpu_tree(genomes, ids, k, nbits, h, seeds=(5, 10)):
    return BloomTree.build_balanced(genomes, ids, k, nbits, h, seeds[0], seeds[1])


def hits_of(offs, leaves):
    return sorted((r, int(leaves[j])) for r in range(len(offs) - 1) for j in range(int(offs[r]), int(offs[r + 1])))


def oracle_hits(t, hits):
    col = {v: i for i, v in enumerate(t.leaves_dfs())}
    return sorted((r, col[v]) for r, v in hits)


# ---------------------------------------------------------------------------------------------------------------
# K1: canonical k-mer + FxHash + double hashing + exact mod
# ---------------------------------------------------------------------------------------------------------------
K1_CASES = [(k, nbits, h, s) for k in (1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 20, 21, 31, 32, 33, 47, 48, 49, 63, 64)
            for nbits, h, s in ((14378, 10, (5, 10)),)] + [
    (21, 71887936, 10, (0x0123456789ABCDEF, 0xFEDCBA9876543210)),
    (20, 11981322, 17, (0xFFFFFFFFFFFFFFFF, 1)),
    (21, 1 << 20, 10, (7, 9)),
    (21, (1 << 32) - 1, 12, (3, 4)),
    (21, 1, 3, (3, 4)),
    (21, 3, 200, (0, 0)),
    (31, 4294967291, 5, (11, 13)),
]


@pytest.mark.parametrize("k,nbits,h,seeds", K1_CASES)
def test_kmer_indices_match_oracle(gpu, k, nbits, h, seeds):
    t = gpu_tree([rand_dna(max(k, 4))], ["g"], k, nbits, h, seeds)
    seqs = [rand_dna(300), rand_dna(200, b"ACGTN"), rand_dna(150, b"acgtACGTNnRYKMSWBDHVryx-*"),
            b"ACGT" * 40, b"A" * 130, (b"ACGTACGTAC" + b"GTACGTACGT") * 8, rand_dna(k), rand_dna(k + 63), rand_dna(k + 64),
            bytes(RNG.integers(0, 256, 257, dtype=np.uint8))]
    for s in seqs:
        got = t.kmer_indices(s)
        kmers = orc.get_kmers(s, k)
        assert got.shape == (len(kmers), h)
        want = np.array([orc.probe_indices(seeds[0], seeds[1], h, nbits, km) for km in kmers], dtype=np.uint64).reshape(len(kmers), h)
        assert np.array_equal(got, want), (k, nbits, s[:40])
    assert t.kmer_indices(rand_dna(max(k - 1, 0))).shape[0] == 0
    t.close()


def test_unsupported_parameters_fail_loudly(gpu):
    with pytest.raises(PfqError):
        gpu_tree([rand_dna(100)], ["g"], 65, 1000, 3)
    with pytest.raises(PfqError):
        gpu_tree([rand_dna(100)], ["g"], 0, 1000, 3)
    with pytest.raises(PfqError):
        gpu_tree([rand_dna(100)], ["g"], 21, 1 << 32, 3)


# ---------------------------------------------------------------------------------------------------------------
# database construction: filters of every node equal the oracle's
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_genomes,k,nbits,h", [(1, 21, 50021, 7), (2, 5, 14378, 10), (5, 20, 100003, 10), (8, 31, 65536, 4),
                                                  (13, 21, 200000, 10)])
def test_balanced_build_matches_oracle(gpu, n_genomes, k, nbits, h):
    genomes = [rand_dna(int(RNG.integers(k, 3000)), b"ACGTN" if i % 3 == 0 else b"ACGT") for i in range(n_genomes)]
    ot, ids = oracle_tree(genomes, k, nbits, h)
    gt = gpu_tree(genomes, ids, k, nbits, h)
    info = gt.info()
    assert (info.n_nodes, info.n_leaves, info.superset_verified) == (ot.n_nodes, n_genomes, 1)
    for v in range(ot.n_nodes):
        assert np.array_equal(gt.node_filter(v), ot.bits[ot.filter_of[v]]), v
    assert [t for t, _ in gt.get_leaf_counts()] == [t for t, _ in ot.leaf_counts()]
    gt.close()


# ---------------------------------------------------------------------------------------------------------------
# query parity
# ---------------------------------------------------------------------------------------------------------------
def check_query(gt, ot, reads, thr, path=-1):
    gt.reset_counts()
    for v in range(ot.n_nodes):
        ot.mapped_reads[v] = 0
    gt.set_path(path)
    seq, off = pack_reads(reads)
    offs, leaves = gt.query_packed(seq, off, thr, want_hits=True)
    ohits, _, _ = orc.query_batch(ot, reads, thr)
    assert gt.get_leaf_counts() == ot.leaf_counts(), (thr, path)
    assert hits_of(offs, leaves) == oracle_hits(ot, ohits), (thr, path)
    st = gt.last_stats()
    assert st.n_hits + st.n_allhit_reads * len(ot.leaves_dfs()) == len(ohits)
    return st


TREES = [(1, 21, 30011, 5), (2, 21, 30011, 10), (3, 11, 20000, 3), (8, 21, 100003, 10), (33, 20, 150001, 10), (64, 21, 65536, 6),
         (70, 16, 200003, 10), (130, 21, 262144, 10)]


@pytest.mark.parametrize("n_genomes", [300, 600, 1100])
def test_wide_trees_all_thresholds(gpu, n_genomes):
    """Trees of 16, 32 and 64 row words: the dense screens (AND-frontier at theta = 1, miss counters below) run with
    4, 8 and 16 lanes per read; reads of >= 256 k-mers take the wide-counter launch."""
    k, nbits, h = 21, 120011, 4
    genomes = [rand_dna(int(RNG.integers(200, 500))) for _ in range(n_genomes)]
    genomes[5] = genomes[4]
    genomes[7] = genomes[6][:150] + genomes[7][150:]
    ot, ids = oracle_tree(genomes, k, nbits, h)
    gt = gpu_tree(genomes, ids, k, nbits, h)
    reads = make_reads(genomes, 400, 150, 150, k) + make_reads(genomes, 12, 6, 420, k) + make_reads(genomes, 20, 5, k + 3, k)
    for thr in (1.0, 0.3, 0.7, 0.95, 0.0):
        for path in (0, 1):
            st = check_query(gt, ot, reads, thr, path=path)
            assert st.path == (path if 0.0 < thr <= 1.0 else 0)
    gt.close()


@pytest.mark.parametrize("n_genomes,nbits,h", [(300, 8009, 3), (1100, 16001, 2)])
def test_counting_screen_on_full_filters(gpu, n_genomes, nbits, h):
    """Thresholds below 1 against filters that are 10 % (and more) full: foreign leaves survive the screen's first
    maxmiss + 9 k-mers by luck, so reads at their limit go on while a live leaf is about to die — reads of mixed lengths in
    one wave (the limits differ; a read that goes on after a longer neighbour's segment was staged gets its bytes anew)."""
    k = 21
    genomes = [rand_dna(int(RNG.integers(250, 400))) for _ in range(n_genomes)]
    genomes[5] = genomes[4]
    ot, ids = oracle_tree(genomes, k, nbits, h)
    gt = gpu_tree(genomes, ids, k, nbits, h)
    reads = []
    for L in (60, 100, 150, 151, 220, 250):
        reads += make_reads(genomes, 120, 60, L, k)
    RNG.shuffle(reads)
    reads += make_reads(genomes, 10, 5, 600, k)            # a few long ones (>= 256 k-mers: the wide-counter launch)
    for thr in (0.3, 0.5, 0.8):
        for path in (0, 1):
            st = check_query(gt, ot, reads, thr, path=path)
            assert st.path == path
    gt.close()


@pytest.mark.parametrize("n_genomes,k,nbits,h", TREES)
def test_query_matches_oracle(gpu, n_genomes, k, nbits, h):
    genomes = [rand_dna(int(RNG.integers(300, 1500))) for _ in range(n_genomes)]
    if n_genomes > 3:
        genomes[1] = genomes[0][:200] + genomes[1][200:]          # shared prefix: reads hit several leaves
        genomes[2] = genomes[0]                                    # identical twin
    ot, ids = oracle_tree(genomes, k, nbits, h)
    gt = gpu_tree(genomes, ids, k, nbits, h)
    reads = make_reads(genomes, 150, 60, 150, k) + make_reads(genomes, 10, 5, 700, k) + make_reads(genomes, 20, 5, k + 2, k)
    for thr in (1.0, 0.0, 0.3, 0.5, 0.75, 0.999, 1.5, -1.0, float("nan")):
        st = check_query(gt, ot, reads, thr, path=0)
        assert st.path == 0
    for thr in (0.3, 0.5, 0.75, 0.999):                            # bucketed at thresholds < 1: per-k-mer miss bytes
        st = check_query(gt, ot, reads, thr, path=1)
        assert st.path == 1    # (tile passes or not: chosen from the previous call's share of clean pairs)
    for thr in (0.0, 1.5, float("nan")):                           # nothing to certify: stays on the direct kernel
        assert check_query(gt, ot, reads, thr, path=1).path == 0
    st = check_query(gt, ot, reads, 1.0, path=1)                   # bucketed: screen + records + LDS-tile certificates
    assert st.path == 1 and st.tile_mode == 1
    for env, val, tile in (("PFQ_TILE_GB", "0", 1),                # no room for probe buckets: every pair takes the fallback
                           ("PFQ_TILE", "0", 0),                   # record-driven L2-sliced verify only
                           ("PFQ_RECORD_GB", "0", 0)):             # no probe records: re-hash per slice
        gt.set_option(env, val)    # (the environment is read once per tree; pfq_set_option changes a knob afterwards)
        try:
            st = check_query(gt, ot, reads, 1.0, path=1)
            assert st.path == 1 and st.tile_mode == tile
            if env == "PFQ_TILE_GB":
                assert 0 < st.n_fallback_pairs <= st.n_candidates and st.n_fallback_pairs >= 0.9 * st.n_candidates
        finally:
            gt.set_option(env, None)
    gt.close()


def test_counts_accumulate_and_want_hits_off(gpu):  # query.rs:143, :356-380
    genomes = [rand_dna(800) for _ in range(6)]
    ot, ids = oracle_tree(genomes, 21, 60000, 8)
    gt = gpu_tree(genomes, ids, 21, 60000, 8)
    a, b = make_reads(genomes, 80, 30, 100, 21), make_reads(genomes, 50, 50, 150, 21)
    for reads, thr in ((a, 1.0), (b, 0.4), (a, 0.0)):
        seq, off = pack_reads(reads)
        assert gt.query_packed(seq, off, thr, want_hits=False) is None
        orc.query_batch(ot, reads, thr)
    assert gt.get_leaf_counts() == ot.leaf_counts()
    gt.close()


def test_reference_query_fixtures_on_gpu(gpu):  # query.rs:267-380 through the reference-shaped host interface
    nbits = orc.needed_bits(0.001, 1000)
    h = orc.optimal_num_hashes(nbits, 1000)
    t = BloomTree.build_balanced([b"ATCGCA"], ["genome"], 3, nbits, h, 5, 10)
    for reads, thr, want in (([b"ATCG"], 1.0, 1), ([b"AAAA"], 1.0, 0), ([b"ATCG", b"AAAA"], 0.0, 2)):
        t.reset_counts()
        query_batch(t, reads, thr)
        assert t.get_leaf_counts() == [("genome", want)]
    t.close()
    four = [b"ATCAG", b"TTTAG", b"CTCAG", b"ATTAG"]
    names = ["baseline", "diff", "onediff_first", "onediff_mid"]
    t = BloomTree.build_balanced(four, names, 4, nbits, h, 5, 10)
    rm = ResultMap()
    query_batch(t, [b"TCAG"], 0.1, rm, ["read_tcag"])
    query_batch(t, [b"ATCA"], 0.1, rm, ["read_atca"])
    c = dict(t.get_leaf_counts())
    assert c["baseline"] >= 2 and c["diff"] == 0
    assert rm.read_mapped("read_tcag") and "baseline" in rm.get_ext_id("read_tcag")
    t.close()


# ---------------------------------------------------------------------------------------------------------------
# database files: reader, writer, CLASSIFICATION.csv, irregular trees
# ---------------------------------------------------------------------------------------------------------------
def test_db_load_save_and_csv(gpu, tmp_path):
    genomes = [rand_dna(int(RNG.integers(400, 900))) for _ in range(9)]
    ot, ids = oracle_tree(genomes, 20, 80021, 10, seeds=(0xDEADBEEF12345678, 0x0BADF00D87654321))
    d1 = str(tmp_path / "db1")
    fmt.write_db(ot, d1)
    gt = BloomTree.load(d1)
    i = gt.info()
    assert (i.kmer_size, i.nbits, i.num_hashes, i.seed1, i.seed2, i.n_nodes, i.superset_verified) == (
        20, 80021, 10, 0xDEADBEEF12345678, 0x0BADF00D87654321, ot.n_nodes, 1)
    reads = make_reads(genomes, 120, 40, 120, 20)
    check_query(gt, ot, reads, 1.0)
    check_query(gt, ot, reads, 0.35)
    out = str(tmp_path / "CLASSIFICATION.csv")
    gt.save_leaf_counts(out)
    assert open(out).read() == ot.classification_csv()
    # writer: what the product saves, the format restatement reads back identically
    d2 = str(tmp_path / "db2")
    gt.save(d2)
    u = fmt.read_db(d2)
    assert (u.kmer_size, u.nbits, u.num_hashes, u.seed1, u.seed2) == (20, 80021, 10, ot.seed1, ot.seed2)
    assert u.left == ot.left and u.right == ot.right and u.tax_id == ot.tax_id and u.bf_path == ot.bf_path
    assert u.mapped_reads == ot.mapped_reads
    for v in range(ot.n_nodes):
        assert np.array_equal(u.bits[u.filter_of[v]], ot.bits[ot.filter_of[v]])
    gt.close()
    with pytest.raises(PfqError):
        BloomTree.load(str(tmp_path / "missing"))
    os.remove(os.path.join(d1, ot.bf_path[3]))
    with pytest.raises(PfqError):
        BloomTree.load(d1)


def test_non_superset_and_shared_filters(gpu, tmp_path):
    """Trees the reference can produce: an internal filter that lost bits, and two nodes naming one
    .bf.  The walk is then not pure pruning; results must still equal the full reference traversal."""
    genomes = [rand_dna(600) for _ in range(8)]
    ot, ids = oracle_tree(genomes, 21, 50021, 6)
    internal = [v for v in range(ot.n_nodes) if not ot.is_leaf(v)]
    ot.bits[ot.filter_of[internal[1]]][::2] = 0                 # drop half the words of one internal filter
    ot.bits[ot.filter_of[internal[-1]]][:] = 0                  # and blank another
    a, b = internal[2], internal[3]
    ot.bf_path[b] = ot.bf_path[a]                               # name collision: two nodes alias one filter
    ot.filter_of[b] = ot.filter_of[a]
    d = str(tmp_path / "db")
    fmt.write_db(ot, d)
    gt = BloomTree.load(d)
    assert gt.info().superset_verified == 0
    reads = make_reads(genomes, 200, 40, 150, 21, errors=False)
    for thr in (1.0, 0.6, 0.0):
        check_query(gt, ot, reads, thr)
    gt.close()


@pytest.mark.parametrize("depth", [0, 1, 2, 3, 10])
def test_prune_tree_matches_oracle(gpu, depth):  # bloom_tree.rs:302-330
    genomes = [rand_dna(500) for _ in range(11)]
    ot, ids = oracle_tree(genomes, 21, 40009, 5)
    gt = gpu_tree(genomes, ids, 21, 40009, 5)
    ot.prune(depth)
    gt.prune_tree(depth)
    assert [t for t, _ in gt.get_leaf_counts()] == [t for t, _ in ot.leaf_counts()]
    reads = make_reads(genomes, 100, 30, 150, 21)
    check_query(gt, ot, reads, 1.0)
    check_query(gt, ot, reads, 0.5)
    gt.close()


# ---------------------------------------------------------------------------------------------------------------
# synthetic workload generators (bench data) and a BASELINE config-2-shaped case
# ---------------------------------------------------------------------------------------------------------------
def test_synthetic_generators_match_oracle(gpu):
    from hipbuf import DeviceBuffer, synchronize
    from phagefilter_amd import _ffi
    n_g, glen, rlen, n_r = 5, 1000, 150, 4000
    dg = DeviceBuffer(n_g * glen)
    _ffi.check(_ffi.lib().pfq_synth_genomes_device(dg.ptr, n_g, glen, 0x5EED0000, None))
    want_g = np.stack([np.frombuffer(orc.synth_genome(0x5EED0000 + i, glen), dtype=np.uint8) for i in range(n_g)])
    synchronize()
    assert np.array_equal(dg.to_numpy().reshape(n_g, glen), want_g)
    dr = DeviceBuffer(n_r * rlen)
    _ffi.check(_ffi.lib().pfq_synth_reads_device(dr.ptr, 100, n_r, rlen, dg.ptr, glen, n_g, 0x5EED1234, None))
    synchronize()
    want_r = orc.synth_reads(0x5EED1234, 100, n_r, rlen, want_g, glen)
    assert np.array_equal(dr.to_numpy().reshape(n_r, rlen), want_r)


def test_config2_shape_64_leaves_k21(gpu):
    """BASELINE config 2 shape (64-leaf SBT, k=21, 150 bp, theta=1, 50/50 mix) at a size the oracle finishes in
    seconds: nbits scaled down with the genome length so the fill profile matches (0.69 % per leaf)."""
    n_g, glen, k, h = 64, 5000, 21, 10
    nbits = 7188793
    genomes_np = np.stack([np.frombuffer(orc.synth_genome(0x5EED0000 + i, glen), dtype=np.uint8) for i in range(n_g)])
    genomes = [g.tobytes() for g in genomes_np]
    ot, ids = oracle_tree(genomes, k, nbits, h, seeds=(0x0123456789ABCDEF, 0xFEDCBA9876543210))
    gt = gpu_tree(genomes, ids, k, nbits, h, seeds=(0x0123456789ABCDEF, 0xFEDCBA9876543210))
    n_reads = 300000
    reads_np = orc.synth_reads(0x5EED1234, 0, n_reads, 150, genomes_np, glen)
    seq = np.concatenate([reads_np.reshape(-1), np.zeros(16, dtype=np.uint8)])
    off = (np.arange(n_reads + 1, dtype=np.uint64) * 150)
    ohits, _, _ = orc.query_batch_packed(ot, seq, off, 1.0, threads=8)
    for path in (0, 1):
        gt.reset_counts()
        gt.set_path(path)
        offs, leaves = gt.query_packed(seq, off, 1.0, want_hits=True)
        assert gt.get_leaf_counts() == ot.leaf_counts()
        got = np.stack([np.repeat(np.arange(n_reads), np.diff(offs).astype(np.int64)), leaves.astype(np.int64)], 1)
        want = np.array(oracle_hits(ot, ohits), dtype=np.int64).reshape(-1, 2)
        assert np.array_equal(got, want)
        assert gt.last_stats().path == path
    # every positive read hits (at least) its source leaf: about half of the reads
    assert 0.49 * n_reads < len(ohits) < 0.52 * n_reads
    # thresholds below 1 at the same size (>= 2^18 reads: the bucketed path is chosen on its own), reads with errors
    rng = np.random.default_rng(99)
    noisy = seq.copy()
    where = rng.random(noisy.size - 16) < 0.02
    noisy[:-16][where] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(where.sum()))
    # (the LDS-tile passes leave every k-mer that is not contained in the chunks' miss bitmaps; PFQ_TILE_COUNTS=0: the record
    # kernel counts alone; with a small entry buffer, chunks of passes that were not launched go to the record kernel too)
    for thr in (0.3, 0.7):
        for v in range(ot.n_nodes):
            ot.mapped_reads[v] = 0
        oh, _, _ = orc.query_batch_packed(ot, noisy, off, thr, threads=8)
        want = np.array(oracle_hits(ot, oh), dtype=np.int64).reshape(-1, 2)
        for env in ({}, {"PFQ_TILE_COUNTS": "0"}, {"PFQ_TILE_ENTRIES": "6000000"}, {"PFQ_TILE_ENTRIES": "40000"}):
            for key, val in env.items():
                gt.set_option(key, val)
            try:
                for call in range(2):
                    gt.reset_counts()
                    gt.set_path(-1)
                    offs, leaves = gt.query_packed(noisy, off, thr, want_hits=True)
                    st = gt.last_stats()
                    assert st.path == 1 and st.tile_mode == (0 if "PFQ_TILE_COUNTS" in env else 1)
                    assert gt.get_leaf_counts() == ot.leaf_counts(), (thr, env, call)
                    got = np.stack([np.repeat(np.arange(n_reads), np.diff(offs).astype(np.int64)), leaves.astype(np.int64)], 1)
                    assert np.array_equal(got, want), (thr, env, call)
            finally:
                for key in env:
                    gt.set_option(key, None)
    # clean reads at a threshold below 1
    for v in range(ot.n_nodes):
        ot.mapped_reads[v] = 0
    oh, _, _ = orc.query_batch_packed(ot, seq, off, 0.5, threads=8)
    want = np.array(oracle_hits(ot, oh), dtype=np.int64).reshape(-1, 2)
    modes = []
    for call in range(3):
        gt.reset_counts()
        offs, leaves = gt.query_packed(seq, off, 0.5, want_hits=True)
        modes.append(gt.last_stats().tile_mode)
        assert gt.get_leaf_counts() == ot.leaf_counts(), call
        got = np.stack([np.repeat(np.arange(n_reads), np.diff(offs).astype(np.int64)), leaves.astype(np.int64)], 1)
        assert np.array_equal(got, want), call
    assert modes[-1] == 1, modes
    for v in range(ot.n_nodes):
        ot.mapped_reads[v] = 0
    orc.query_batch_packed(ot, seq, off, 1.0, threads=8)     # (restore the threshold-1 counts used below)
    gt.close()
    # The probe buckets are reused pass after pass when they cannot hold all pairs.  The first call does not know how
    # many passes the plan needs (one is launched, the record kernel certifies the chunks of the others); the next
    # call launches as many as the previous one needed.  Same results every time.
    want_counts = ot.leaf_counts()
    for entries in (6_000_000, 60_000_000, 40_000):      # ~4 passes / 1 pass / chunks larger than the whole buffer
        os.environ["PFQ_TILE_ENTRIES"] = str(entries)
        try:
            g2 = gpu_tree(genomes, ids, k, nbits, h, seeds=(0x0123456789ABCDEF, 0xFEDCBA9876543210))
            g2.set_path(1)
            seen = []
            for call in range(3):
                g2.reset_counts()
                offs, leaves = g2.query_packed(seq, off, 1.0, want_hits=True)
                assert g2.get_leaf_counts() == want_counts, (entries, call)
                assert int(offs[-1]) == len(ohits)
                st = g2.last_stats()
                assert st.tile_mode == 1
                seen.append((st.tile_passes_launched, st.tile_passes_needed, st.n_fallback_pairs))
            if entries == 6_000_000:
                assert seen[0][0] == 1 and seen[0][1] > 1 and seen[1][0] == seen[0][1] and seen[2][0] == seen[1][1], seen
            if entries == 40_000:
                assert all(s[2] > 0 for s in seen), seen   # every chunk takes the fallback
            g2.close()
        finally:
            del os.environ["PFQ_TILE_ENTRIES"]


# ---------------------------------------------------------------------------------------------------------------
# BASELINE.json full-size configurations
# ---------------------------------------------------------------------------------------------------------------
def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def _read_plan(seed, first, count, n_genomes):
    """(is_positive, leaf) of reads first..first+count from the generator's definition (oracle/pfq_oracle.c)."""
    with np.errstate(over="ignore"):
        r = np.arange(first, first + count, dtype=np.uint64)
        w0 = _splitmix64(_splitmix64(np.uint64(seed)) + np.uint64(8) * r)
    return (w0 & np.uint64(1)).astype(bool), ((w0 >> np.uint64(8)) % np.uint64(n_genomes)).astype(np.int64)


def test_config2_full_size_1m_reads_64_leaves(gpu):
    """BASELINE config 2 at its real size: 1 M synthetic 150 bp reads vs a 64-leaf SBT, k=21, nbits=71 887 936,
    10 hashes, 50 kbp genomes; per-leaf counts and every per-read hit set equal the oracle's, on both paths."""
    from hipbuf import DeviceBuffer
    n_g, glen, k, h, nbits, n_reads = 64, 50000, 21, 10, 71887936, 1000000
    seeds = (0x0123456789ABCDEF, 0xFEDCBA9876543210)
    genomes_np = np.stack([np.frombuffer(orc.synth_genome(0x5EED0000 + i, glen), dtype=np.uint8) for i in range(n_g)])
    ids = [f"G{i:05d}" for i in range(n_g)]
    ot = orc.build_balanced_tree([g.tobytes() for g in genomes_np], ids, k, nbits, h, *seeds)
    dg = DeviceBuffer.from_numpy(genomes_np.reshape(-1))
    gt = BloomTree.build_balanced_device(dg.ptr, glen, n_g, ids, k, nbits, h, *seeds)
    for v in (0, 1, 63, 126):
        assert np.array_equal(gt.node_filter(v), ot.bits[v])
    reads_np = orc.synth_reads(0x5EED1234, 0, n_reads, 150, genomes_np, glen)
    seq = np.concatenate([reads_np.reshape(-1), np.zeros(16, dtype=np.uint8)])
    off = np.arange(n_reads + 1, dtype=np.uint64) * 150
    ohits, _, _ = orc.query_batch_packed(ot, seq, off, 1.0, threads=min(32, os.cpu_count() or 8))
    want = np.array(oracle_hits(ot, ohits), dtype=np.int64).reshape(-1, 2)
    for path in (1, 0):
        gt.reset_counts()
        gt.set_path(path)
        offs, leaves = gt.query_packed(seq, off, 1.0, want_hits=True)
        assert gt.get_leaf_counts() == ot.leaf_counts()
        got = np.stack([np.repeat(np.arange(n_reads), np.diff(offs).astype(np.int64)), leaves.astype(np.int64)], 1)
        assert np.array_equal(got, want)
    pos, leaf = _read_plan(0x5EED1234, 0, n_reads, n_g)
    assert np.array_equal(np.unique(want[:, 0]), np.nonzero(pos)[0])          # exactly the positive reads hit
    gt.close()


@pytest.mark.parametrize("thr", [1.0, 0.3])
def test_config3_properties_1024_leaves(gpu, thr):
    """BASELINE config 3/4 shape (1024-leaf SBT, full parameters) through size-independent properties: every
    positive read hits its source leaf; counts are additive over a partition of the reads (what read sharding across
    GPUs relies on) and independent of the query path and of the block size; false-positive leaves are rare."""
    from hipbuf import DeviceBuffer, synchronize
    from phagefilter_amd import _ffi
    L = _ffi.lib()
    n_g, glen, k, h, nbits, n_reads = 1024, 50000, 21, 10, 71887936, 4 * 1024 * 1024
    ids = [f"G{i:05d}" for i in range(n_g)]
    dg = DeviceBuffer(n_g * glen)
    _ffi.check(L.pfq_synth_genomes_device(dg.ptr, n_g, glen, 0x5EED0000, None))
    synchronize()
    gt = BloomTree.build_balanced_device(dg.ptr, glen, n_g, ids, k, nbits, h, 0x0123456789ABCDEF, 0xFEDCBA9876543210)
    info = gt.info()
    assert (info.n_nodes, info.n_leaves, info.superset_verified) == (2047, 1024, 1)
    dr = DeviceBuffer(n_reads * 150 + 64)
    _ffi.check(L.pfq_synth_reads_device(dr.ptr, 0, n_reads, 150, dg.ptr, glen, n_g, 0x5EED1234, None))
    off = DeviceBuffer.from_numpy(np.arange(n_reads + 1, dtype=np.uint64) * 150)
    synchronize()
    pos, leaf = _read_plan(0x5EED1234, 0, n_reads, n_g)
    expect = np.bincount(leaf[pos], minlength=n_g)

    def counts(path, lo, hi):
        gt.reset_counts()
        gt.set_path(path)
        gt.query_device(dr.ptr + lo * 150, off.ptr, hi - lo, (hi - lo) * 150, thr, 0)
        synchronize()
        return np.array([c for _, c in gt.get_leaf_counts()], dtype=np.int64)

    whole = counts(1, 0, n_reads)
    assert (whole >= expect).all()                                   # every positive read hits its source leaf
    assert 0 <= int(whole.sum() - expect.sum()) <= n_reads // 100000  # Bloom false-positive leaves are rare
    half = n_reads // 2
    assert np.array_equal(counts(1, 0, half) + counts(1, half, n_reads), whole)          # additive over shards
    thirds = [0, 1000003, 2500000, n_reads]
    assert np.array_equal(sum(counts(0, a, b) for a, b in zip(thirds, thirds[1:])), whole)  # path/block independent
    assert np.array_equal(counts(1, 0, n_reads), whole)                                  # idempotent
    gt.close()


# ---------------------------------------------------------------------------------------------------------------
# subtree shards (BASELINE config 5: a tree larger than one GPU, one shard per rank, every rank sees all reads)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("depth,corrupt_root", [(0, False), (1, False), (2, False), (3, True), (3, False), (6, False)])
def test_subtree_shards_concatenate_to_whole_tree(gpu, tmp_path, depth, corrupt_root):
    genomes = [rand_dna(int(RNG.integers(400, 900))) for _ in range(13)]
    genomes[5] = genomes[4]                                        # a read hitting leaves of two different shards
    ot, ids = oracle_tree(genomes, 21, 50021, 7)
    if corrupt_root:                                               # ancestors outside the shard must still be honoured
        ot.bits[ot.filter_of[ot.root]][::3] = 0
    d = str(tmp_path / "db")
    fmt.write_db(ot, d)
    reads = make_reads(genomes, 150, 40, 120, 21, errors=False)
    seq, off = pack_reads(reads)
    for thr in (1.0, 0.5):
        for v in range(ot.n_nodes):
            ot.mapped_reads[v] = 0
        ohits, _, _ = orc.query_batch(ot, reads, thr)
        want_counts, want_hits = ot.leaf_counts(), oracle_hits(ot, ohits)
        got_counts, got_hits, next_leaf, i = [], [], 0, 0
        while True:
            try:
                sh = BloomTree.load_subtree(d, depth, i)
            except PfqError as e:
                assert e.code == -1 and i > 0
                break
            info = sh.info()
            assert info.shard_first_leaf == next_leaf and info.tree_leaves == len(want_counts)
            offs, leaves = sh.query_packed(seq, off, thr, want_hits=True)
            got_counts += sh.get_leaf_counts()
            got_hits += [(r, c + next_leaf) for r, c in hits_of(offs, leaves)]
            next_leaf += int(info.n_leaves)
            with pytest.raises(PfqError):
                sh.save(str(tmp_path / "nope"))
            sh.close()
            i += 1
        assert next_leaf == len(want_counts)
        assert got_counts == want_counts, (depth, thr)
        assert sorted(got_hits) == want_hits, (depth, thr)


# ---------------------------------------------------------------------------------------------------------------
# `build` / `add`: the reference's greedy insertion (bloom_tree.rs:128-245) on the device
# ---------------------------------------------------------------------------------------------------------------
def _preorder(t):
    """(is_leaf, tax_id) of an OracleTree's nodes in pre-order + its filters in that order."""
    return [(t.is_leaf(v), t.tax_id[v]) for v in range(t.n_nodes)]


def _gpu_greedy(genomes, ids, k, fpr, largest, seeds=(5, 10)):
    gt = BloomTree.new(k, fpr, largest, seeds[0], seeds[1])
    for g, i in zip(genomes, ids):
        gt.insert(g, i)
    return gt


def test_greedy_host_walk_knob(gpu, tmp_path):
    """PFQ_GREEDY_HOST=1: the descent level by level from the host (no kernel with a grid barrier) builds the same database
    as the one-launch walk on the device, and both equal the oracle's greedy insertion."""
    genomes = [rand_dna(int(RNG.integers(300, 900))) for _ in range(37)]
    genomes[9] = genomes[4]
    ids = [f"h{i}" for i in range(len(genomes))]
    ot = orc.build_greedy_tree(genomes, ids, 21, 0.001, 2000, 5, 10)
    os.environ["PFQ_GREEDY_HOST"] = "1"
    try:
        gh = _gpu_greedy(genomes, ids, 21, 0.001, 2000)
    finally:
        del os.environ["PFQ_GREEDY_HOST"]
    gd = _gpu_greedy(genomes, ids, 21, 0.001, 2000)
    _assert_same_database(gh, ot, tmp_path, "host")
    _assert_same_database(gd, ot, tmp_path, "device")
    gh.close()
    gd.close()


def _assert_same_database(gt, ot, tmp_path, name):
    """Saved database == oracle tree: topology, names, leaf order and every node's filter."""
    from oracle import pfq_format as fmt
    d = tmp_path / name
    d.mkdir()
    gt.save(str(d))
    lt = fmt.read_db(str(d))
    assert _preorder(lt) == _preorder(ot)
    assert lt.bf_path == ot.bf_path and (lt.kmer_size, lt.nbits, lt.num_hashes, lt.seed1, lt.seed2) == \
        (ot.kmer_size, ot.nbits, ot.num_hashes, ot.seed1, ot.seed2)
    assert (lt.left, lt.right) == (ot.left, ot.right)
    for v in range(ot.n_nodes):
        assert np.array_equal(lt.bits[lt.filter_of[v]], ot.bits[ot.filter_of[v]]), v
        assert np.array_equal(gt.node_filter(v), ot.bits[ot.filter_of[v]]), v


@pytest.mark.gpu
@pytest.mark.parametrize("third,shape", [("ATCAG", "left"), ("TTTAG", "right")])
def test_greedy_insert_reference_fixtures(gpu, tmp_path, third, shape):
    """bloom_tree.rs:586-734 (test_nested_tree_insert_left / _right): the third genome equals the first (second), so it
    is placed next to it whatever the hash seeds are."""
    seqs = [("test1", b"ATCAG"), ("test2", b"TTTAG"), ("test3", third.encode())]
    for seeds in ((5, 10), (0x0123456789ABCDEF, 0xFEDCBA9876543210)):
        gt = _gpu_greedy([s for _, s in seqs], [i for i, _ in seqs], 5, 0.001, 1000, seeds)
        info = gt.info()
        assert (info.n_nodes, info.n_leaves, info.superset_verified) == (5, 3, 1)
        leaves = [t for t, _ in gt.get_leaf_counts()]
        assert leaves == (["test1", "test3", "test2"] if shape == "left" else ["test1", "test2", "test3"])
        ot = orc.build_greedy_tree([s for _, s in seqs], [i for i, _ in seqs], 5, 0.001, 1000, *seeds)
        _assert_same_database(gt, ot, tmp_path, f"db_{shape}_{seeds[0] & 0xff}")
        gt.close()


@pytest.mark.gpu
def test_greedy_insert_empty_and_single(gpu, tmp_path):
    """bloom_tree.rs:457-585 (test_empty_tree_insert / test_one_elem_tree_insert)."""
    gt = BloomTree.new(5, 0.001, 1000, 5, 10)
    assert gt.info().n_nodes == 0
    gt.insert(b"ATCAGTTTAG", "only")
    assert (gt.info().n_nodes, gt.info().n_leaves) == (1, 1)
    gt.insert(b"TTTAGGGGGA", "second", internal_name="Internal_Node_77")
    assert (gt.info().n_nodes, gt.info().n_leaves) == (3, 2)
    assert [t for t, _ in gt.get_leaf_counts()] == ["only", "second"]   # old leaf left, new leaf right (bloom_tree.rs:241-242)
    offs, leaves = gt.query_packed(*pack_reads([b"ATCAGTTTAG", b"TTTAGGGGGA", b"ACACACACAC"]), 1.0, want_hits=True)
    assert hits_of(offs, leaves) == [(0, 0), (1, 1)]
    gt.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_genomes,k,fpr,largest", [(12, 9, 0.01, 400), (40, 15, 0.001, 3000), (90, 21, 0.001, 1200)])
def test_greedy_build_matches_oracle_and_queries(gpu, tmp_path, n_genomes, k, fpr, largest):
    """`build` then `add`: same topology, names and filters as the oracle's restatement of the greedy insertion, for
    genome families (mutated copies cluster), unrelated genomes, very short ones and exact ties."""
    base = [rand_dna(int(RNG.integers(200, 1200))) for _ in range(max(3, n_genomes // 4))]
    genomes = []
    for i in range(n_genomes):
        g = bytearray(base[int(RNG.integers(0, len(base)))])
        for _ in range(int(RNG.integers(0, 12))):
            g[int(RNG.integers(0, len(g)))] = ord("ACGT"[int(RNG.integers(0, 4))])
        genomes.append(bytes(g))
    genomes[3] = genomes[2]                    # identical twin: distance 0
    genomes[5] = b"ACGT"[: max(1, k - 6)]      # shorter than k: an empty filter (equidistant: ties go left)
    ids = [f"g{i}" for i in range(n_genomes)]
    first = n_genomes * 2 // 3
    gt = _gpu_greedy(genomes[:first], ids[:first], k, fpr, largest)
    ot = orc.build_greedy_tree(genomes[:first], ids[:first], k, fpr, largest, 5, 10)
    _assert_same_database(gt, ot, tmp_path, "built")
    gt.close()
    # `add`: load the saved database and insert the rest
    gt = BloomTree.load(str(tmp_path / "built"))
    for g, i in zip(genomes[first:], ids[first:]):
        gt.insert(g, i)
        orc.greedy_insert(ot, g, i)
    orc.renumber_preorder(ot)
    _assert_same_database(gt, ot, tmp_path, "added")
    reads = make_reads(genomes, 120, 40, 150, k)
    for thr in (1.0, 0.5):
        check_query(gt, ot, reads, thr)
    gt.close()


@pytest.mark.gpu
def test_page_locked_host_buffers(gpu):
    """pfq_host_alloc / pfq_host_free: the batch handed to pfq_query_batch may live in page-locked memory."""
    import ctypes as C
    from phagefilter_amd import _ffi
    genomes = [rand_dna(600) for _ in range(5)]
    ot, ids = oracle_tree(genomes, 21, 60013, 5)
    gt = gpu_tree(genomes, ids, 21, 60013, 5)
    reads = make_reads(genomes, 60, 20, 150, 21)
    seq, off = pack_reads(reads)
    L = _ffi.lib()
    p_seq, p_off = C.c_void_p(), C.c_void_p()
    _ffi.check(L.pfq_host_alloc(seq.nbytes + 16, C.byref(p_seq)))
    _ffi.check(L.pfq_host_alloc(off.nbytes, C.byref(p_off)))
    C.memmove(p_seq, seq.ctypes.data, seq.nbytes)
    C.memmove(p_off, off.ctypes.data, off.nbytes)
    _ffi.check(L.pfq_query_batch(gt._h, p_seq, p_off, len(reads), 1.0, 0, None))
    orc.query_batch(ot, reads, 1.0)
    assert gt.get_leaf_counts() == ot.leaf_counts()
    _ffi.check(L.pfq_host_free(p_seq))
    _ffi.check(L.pfq_host_free(p_off))
    _ffi.check(L.pfq_host_free(None))
    gt.close()


@pytest.mark.gpu
def test_async_host_blocks_thresholds_below_one(gpu):
    """Block loop of the CLI: pfq_query_batch without PFQ_WANT_HITS returns before the kernels ran, blocks follow each
    other through the two input buffers, and the sizing hints of one call (pairs per read, share of pairs with a k-mer
    missing) steer the next.  Counts after the loop equal the oracle's over all reads, at thresholds below 1 too."""
    import ctypes as C
    from phagefilter_amd import _ffi
    n_g, glen, k, h, nbits = 64, 3000, 21, 10, 4000037
    genomes_np = np.stack([np.frombuffer(orc.synth_genome(0xABC0000 + i, glen), dtype=np.uint8) for i in range(n_g)])
    genomes = [g.tobytes() for g in genomes_np]
    ot, ids = oracle_tree(genomes, k, nbits, h)
    gt = gpu_tree(genomes, ids, k, nbits, h)
    gt.set_path(1)
    L = _ffi.lib()
    n_blocks, per = 4, 30000
    reads_np = orc.synth_reads(0x77771234, 0, n_blocks * per, 150, genomes_np, glen)
    rng = np.random.default_rng(5)
    for thr, err in ((0.5, 0.0), (0.3, 0.02), (1.0, 0.0)):
        data = reads_np.copy().reshape(-1)
        if err:
            where = rng.random(data.size) < err
            data[where] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(where.sum()))
        seq = np.concatenate([data, np.zeros(16, dtype=np.uint8)])
        off = np.arange(n_blocks * per + 1, dtype=np.uint64) * 150
        for v in range(ot.n_nodes):
            ot.mapped_reads[v] = 0
        orc.query_batch_packed(ot, seq, off, thr, threads=8)
        gt.reset_counts()
        for b in range(n_blocks):   # (nothing in the loop waits for the device)
            bseq = np.ascontiguousarray(seq[b * per * 150:(b + 1) * per * 150 + 16])
            boff = np.arange(per + 1, dtype=np.uint64) * 150
            _ffi.check(L.pfq_query_batch(gt._h, bseq.ctypes.data_as(C.c_void_p), boff.ctypes.data_as(C.c_void_p), per,
                                         C.c_float(thr), 0, None))
        assert gt.get_leaf_counts() == ot.leaf_counts(), (thr, err)
        assert gt.last_stats().path == 1
    gt.close()


# ---------------------------------------------------------------------------------------------------------------
# randomized configurations: every combination of tree width, read-length mix, threshold, query path and bucket
# buffer size goes through the same comparison with the oracle (per-leaf counts and every per-read hit set)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("seed", [int(os.environ.get("PFQ_PARITY_SEED0", "0")) + i
                                  for i in range(int(os.environ.get("PFQ_PARITY_SEEDS", "10")))])   # (soak runs: more seeds)
def test_randomized_parity(gpu, seed):
    rng = np.random.default_rng(1000 + seed)
    n_genomes = int(rng.choice([3, 17, 40, 130, 290, 520, 1040]))
    k = int(rng.choice([9, 15, 20, 21, 31, 32, 47]))
    h = int(rng.choice([2, 3, 7, 10, 17]))
    nbits = int(rng.choice([40009, 131072, 300007, 1 << 20, 2500003]))
    glen = int(rng.integers(max(k + 5, 60), 900))

    def dna(n):
        return bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n).astype(np.uint8))

    base = [dna(glen) for _ in range(max(2, n_genomes // int(rng.choice([1, 3, 8]))))]
    genomes = []
    for i in range(n_genomes):
        g = bytearray(base[int(rng.integers(0, len(base)))])
        for _ in range(int(rng.integers(0, 6))):
            g[int(rng.integers(0, len(g)))] = ord("ACGT"[int(rng.integers(0, 4))])
        genomes.append(bytes(g))
    ot, ids = oracle_tree(genomes, k, nbits, h)
    reads = []
    for _ in range(int(rng.integers(150, 500))):
        src = genomes[int(rng.integers(0, n_genomes))] * int(rng.choice([1, 1, 1, 3, 25]))   # some reads span repeats: long reads
        L = int(min(len(src), rng.choice([k, k + 1, 64 + k - 1, 100, 150, 151, 250, 300, 700, 2000, 20000])))
        o = int(rng.integers(0, len(src) - L + 1))
        r = bytearray(src[o:o + L])
        for _ in range(int(rng.choice([0, 0, 1, 3, 10]))):
            r[int(rng.integers(0, L))] = ord("ACGTN"[int(rng.integers(0, 5))])
        reads.append(bytes(r) if rng.random() < 0.5 else orc.revcomp(bytes(r)))
    reads += [dna(int(rng.integers(0, 400))) for _ in range(60)] + [b"", dna(k - 1)]
    entries = int(rng.choice([0, 0, 30_000, 400_000, 5_000_000]))
    if entries:
        os.environ["PFQ_TILE_ENTRIES"] = str(entries)
    if seed % 3 == 1:
        os.environ["PFQ_TILE_COUNTS"] = "0"   # thresholds < 1: the record kernel counts alone (default: LDS-tile passes with k-mer entries)
    if seed % 3 == 2:
        os.environ["PFQ_BLOCK"] = "1"         # threshold 1: block mode from the first call (default: once reads pass several leaves)
    try:
        gt = gpu_tree(genomes, ids, k, nbits, h)
        for thr in (1.0, float(rng.choice([0.05, 0.3, 0.5, 0.9])), float(rng.choice([0.0, 0.2, 0.75, 0.999, 1.5]))):
            for path in (1, 0, 1):
                check_query(gt, ot, reads, thr, path=path)
        gt.close()
    finally:
        os.environ.pop("PFQ_TILE_ENTRIES", None)
        os.environ.pop("PFQ_TILE_COUNTS", None)
        os.environ.pop("PFQ_BLOCK", None)
