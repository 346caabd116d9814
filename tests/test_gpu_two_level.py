"""GPU parity tests of the two-level frontier (round 3): trees of more than 2048 leaves are screened against a coarse
level of internal nodes first and every group of leaf columns only sees the reads with a live ancestor there — the
device's form of the reference descending with the survivors only (query.rs:113-141).  Through the C ABI, against the
CPU oracle's full DFS; bit-exact, and equal to the flat frontier (PFQ_COARSE=0) on the same inputs."""
import os
import sys

import numpy as np
import pytest

from oracle import pfq_format as fmt
from oracle import pfq_oracle as orc
from phagefilter_amd import BloomTree
from test_gpu_parity import RNG, check_query, gpu_tree, make_reads, oracle_tree, rand_dna

pytestmark = pytest.mark.gpu


def _genomes(n, lo, hi, n_long=0, long_len=700):
    g = [rand_dna(int(RNG.integers(lo, hi))) for _ in range(n)]
    for i in range(n_long):                                     # a few genomes long enough for reads of >= 256 k-mers
        g[(i * 37 + 5) % n] = rand_dna(long_len)
    return g


def _reads(genomes, k, n_pos=600, n_neg=200):
    return (make_reads(genomes, n_pos, n_neg, 150, k) + make_reads(genomes, 40, 10, 420, k) +
            [genomes[3][:150], genomes[7][:100], genomes[len(genomes) - 1][:90]])


@pytest.mark.parametrize("n_genomes,nbits,h", [(2600, 65521, 4), (3100, 8191, 4), (5000, 30011, 6)])
def test_two_level_equals_oracle_and_flat(gpu, n_genomes, nbits, h):
    """Balanced trees of 3 to 5 leaf groups; the 8191-bit filters make the coarse level a third full, so its screens need
    several probes per k-mer.  Every threshold class, both query paths, coarse level on / off."""
    k = 21
    genomes = _genomes(n_genomes, 150, 260, n_long=6)
    genomes[2050] = genomes[3]                                  # twins in different leaf groups
    genomes[n_genomes - 1] = genomes[7][:120] + genomes[n_genomes - 1][120:]
    ot, ids = oracle_tree(genomes, k, nbits, h)
    gt = gpu_tree(genomes, ids, k, nbits, h)
    reads = _reads(genomes, k)
    for thr in (1.0, 0.5, 0.97, 0.0, 0.3):
        st = check_query(gt, ot, reads, thr, path=0)
        assert st.path == 0 and st.leaf_groups == (n_genomes + 1023) // 1024
        if thr > 0:
            assert st.coarse_cols >= 1024 and st.coarse_probes >= 1, (thr, st.coarse_cols)
            # pruning happened: far fewer (read, group) screens than reads x groups
            assert st.group_reads < len(reads) * st.leaf_groups * 0.6, (thr, st.group_reads)
    for thr in (1.0, 0.5, 0.97, 0.3):
        st = check_query(gt, ot, reads, thr, path=1)
        assert st.path == 1 and st.coarse_cols >= 1024
    # the flat frontier on the same tree (layout rebuilt without a coarse level, groups of 2048 columns again)
    gt.set_option("PFQ_COARSE", "0")
    try:
        for thr in (1.0, 0.5):
            for path in (0, 1):
                st = check_query(gt, ot, reads, thr, path=path)
                assert st.coarse_cols == 0 and st.leaf_groups == (n_genomes + 2047) // 2048
    finally:
        gt.set_option("PFQ_COARSE", None)
    gt.close()


@pytest.mark.parametrize("opts", [{"PFQ_COARSE_PROBES": "1"}, {"PFQ_COARSE_PROBES": "4"}, {"PFQ_COARSE_PROBES": "6"},
                                  {"PFQ_GROUP_LOG2": "11"}, {"PFQ_COARSE_COLS": "2048"}, {"PFQ_COARSE_COLS": "300"},
                                  {"PFQ_COARSE_COLS": "2048", "PFQ_GROUP_LOG2": "11", "PFQ_COARSE_PROBES": "3"},
                                  {"PFQ_BLOCK": "1"}, {"PFQ_TILE": "0"}, {"PFQ_RECORD_GB": "0"}])
def test_two_level_shapes_and_knobs(gpu, opts):
    """Rows of 16 / 32 / 64 words at the coarse level, groups of 1024 and of 2048 leaf columns, 1 to 6 probes per k-mer,
    block mode and the fallback certificate kernels behind the leaf groups: results never depend on a knob."""
    k, nbits, h = 20, 16381, 5
    genomes = _genomes(2300, 150, 240, n_long=4)
    genomes[2100] = genomes[10]
    ot, ids = oracle_tree(genomes, k, nbits, h)
    gt = gpu_tree(genomes, ids, k, nbits, h)
    reads = _reads(genomes, k, 500, 150)
    for key, val in opts.items():
        gt.set_option(key, val)
    for thr in (1.0, 0.4, 0.9):
        for path in (0, 1):
            st = check_query(gt, ot, reads, thr, path=path)
            assert st.coarse_cols > 0, (opts, thr, path)
            if "PFQ_COARSE_PROBES" in opts:
                assert st.coarse_probes == min(int(opts["PFQ_COARSE_PROBES"]), 4 if thr < 1 else 6, h)
            if opts.get("PFQ_COARSE_COLS") == "300":
                assert st.coarse_cols <= 300
            if "PFQ_GROUP_LOG2" in opts:
                assert st.leaf_groups == 2
    gt.close()


def _random_shape_tree(genomes, k, nbits, h, skew=0.9, chain=300):
    """An unbalanced SBT: random split points (one side gets 2 - 98 % of the leaves), and one caterpillar of `chain`
    leaves (what the reference's greedy insertion makes of similar genomes).  Internal filters = OR of the children."""
    ids = [f"G{i:05d}" for i in range(len(genomes))]
    t = orc.OracleTree(k, nbits, h, 5, 10)
    counter = [0]
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 5000))

    def rec(lo, hi):
        if hi - lo == 1:
            v = t.add_node(ids[lo], f"{ids[lo]}.bf", -1)
            t.filter_of[v] = v
            return v
        name = f"Internal_Node_{counter[0]}"
        counter[0] += 1
        v = t.add_node(name, f"{name}.bf", -1)
        t.filter_of[v] = v
        n = hi - lo
        if lo < chain:
            mid = lo + 1                                        # caterpillar: one leaf, then the rest
        else:
            f = float(RNG.uniform(0.02, 0.98)) if RNG.random() < skew else 0.5
            mid = lo + min(max(int(n * f), 1), n - 1)
        l = rec(lo, mid)
        r = rec(mid, hi)
        t.left[v], t.right[v] = l, r
        return v

    t.root = rec(0, len(genomes))
    t.bits = np.zeros((t.n_nodes, t.n_words), dtype=np.uint64)
    for i, v in enumerate(t.leaves_dfs()):
        orc.insert_sequence(t, v, genomes[i])
    for v in reversed(range(t.n_nodes)):
        if not t.is_leaf(v):
            t.bits[v] = t.bits[t.left[v]] | t.bits[t.right[v]]
    return t, ids


def test_two_level_on_unbalanced_trees(gpu, tmp_path):
    """The antichain of the coarse level on trees that are nothing like balanced: leaves at every depth, a caterpillar
    whose nodes cover hundreds of leaves, coarse columns that are leaves themselves or span several leaf groups."""
    k, nbits, h = 21, 32749, 5
    genomes = _genomes(2500, 150, 230, n_long=4)
    ot, ids = _random_shape_tree(genomes, k, nbits, h)
    d = str(tmp_path / "db")
    fmt.write_db(ot, d)
    gt = BloomTree.load(d)
    assert gt.info().n_leaves == 2500
    reads = _reads(genomes, k)
    for thr in (1.0, 0.5, 0.3):
        for path in (0, 1):
            st = check_query(gt, ot, reads, thr, path=path)
            assert st.coarse_cols > 0 and st.leaf_groups == 3
    # pruned: internal nodes become leaves (bloom_tree.rs:302-330); depth 9 leaves fewer than 2049 of them -> no coarse level
    for depth in (40, 9):
        ot.prune(depth)
        gt.prune_tree(depth)
        assert [t for t, _ in gt.get_leaf_counts()] == [t for t, _ in ot.leaf_counts()]
        nl = len(ot.leaves_dfs())
        for thr in (1.0, 0.5):
            st = check_query(gt, ot, reads, thr, path=1)
            assert (st.coarse_cols > 0) == (nl > 2048), (depth, nl, st.coarse_cols)
    gt.close()


def test_two_level_with_colliding_internal_names(gpu, tmp_path):
    """Reference-built trees alias filters of internal nodes: a coarse column then tests the aliased filter —
    which is what the reference tests at that node — and the leaves' guard columns still certify every unverified
    ancestor.  2600 leaves, 60 collisions; oracle = the reference's DFS over the aliased files."""
    k, nbits, h = 21, 65521, 4
    genomes = _genomes(2600, 200, 320)
    genomes[11] = genomes[10]
    ot, ids = oracle_tree(genomes, k, nbits, h)
    internal = [v for v in range(ot.n_nodes) if not ot.is_leaf(v)]
    picks = RNG.choice(len(internal), size=120, replace=False)
    for a, b in zip(picks[:60], picks[60:]):
        a, b = internal[int(a)], internal[int(b)]
        ot.bf_path[a] = ot.bf_path[b]
        ot.filter_of[a] = ot.filter_of[b]
    d = str(tmp_path / "db")
    fmt.write_db(ot, d)
    gt = BloomTree.load(d)
    assert gt.info().superset_verified == 0
    reads = make_reads(genomes, 700, 150, 150, k)
    for thr in (1.0, 0.7, 0.3):
        for path in (0, 1):
            st = check_query(gt, ot, reads, thr, path=path)
            assert st.coarse_cols > 0 and st.path == path
    # block mode behind the leaf groups of a tree with guard columns (the guards of the candidates that stand are certified
    # against the sliced matrix, whose groups are 1024 columns wide here)
    gt.set_option("PFQ_BLOCK", "1")
    for thr in (1.0, 0.5):
        st = check_query(gt, ot, reads, thr, path=1)
        assert st.coarse_cols > 0 and st.tile_mode == 2
    gt.close()


# ---------------------------------------------------------------------------------------------------------------
# randomized: tree shape, size, filter geometry, fill of the coarse level, reads, thresholds, paths and layout knobs
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", [int(os.environ.get("PFQ_TWO_LEVEL_SEED0", "0")) + i
                                  for i in range(int(os.environ.get("PFQ_TWO_LEVEL_SEEDS", "6")))])   # (soak runs: more seeds)
def test_two_level_randomized(gpu, tmp_path, seed):
    """Every draw goes through the same comparison with the oracle's DFS (per-leaf counts and every per-read hit set): 2100 to
    4200 leaves, balanced or random shapes with a caterpillar, filters from nearly empty to half full at the coarse level,
    related genomes (a read is a candidate in several leaf groups), short and long reads, and a random layout knob."""
    rng = np.random.default_rng(77000 + seed)
    n_genomes = int(rng.choice([2100, 2600, 3300, 4200]))
    k = int(rng.choice([15, 20, 21, 31]))
    h = int(rng.choice([3, 5, 10, 17]))
    nbits = int(rng.choice([4099, 16381, 65521, 262147]))
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def dna(n):
        return bytes(rng.choice(acgt, int(n)).astype(np.uint8))

    base = [dna(rng.integers(k + 40, 420)) for _ in range(max(2, n_genomes // int(rng.choice([1, 1, 4, 16]))))]
    genomes = []
    for i in range(n_genomes):
        g = bytearray(base[int(rng.integers(0, len(base)))])
        for _ in range(int(rng.integers(0, 5))):
            g[int(rng.integers(0, len(g)))] = ord("ACGT"[int(rng.integers(0, 4))])
        genomes.append(bytes(g))
    for i in range(3):                                          # a few genomes long enough for reads of >= 256 k-mers
        genomes[int(rng.integers(0, n_genomes))] = dna(700)
    if rng.random() < 0.5:
        ot, ids = oracle_tree(genomes, k, nbits, h)
        gt = gpu_tree(genomes, ids, k, nbits, h)
    else:
        ot, ids = _random_shape_tree(genomes, k, nbits, h, skew=float(rng.choice([0.5, 0.9])), chain=int(rng.choice([0, 40, 300])))
        d = str(tmp_path / "db")
        fmt.write_db(ot, d)
        gt = BloomTree.load(d)
    reads = []
    for _ in range(int(rng.integers(300, 700))):
        src = genomes[int(rng.integers(0, n_genomes))]
        L = int(min(len(src), rng.choice([k, k + 1, 100, 150, 151, 300, 700])))
        o = int(rng.integers(0, len(src) - L + 1))
        r = bytearray(src[o:o + L])
        for _ in range(int(rng.choice([0, 0, 1, 3]))):
            r[int(rng.integers(0, L))] = ord("ACGTN"[int(rng.integers(0, 5))])
        reads.append(bytes(r) if rng.random() < 0.5 else orc.revcomp(bytes(r)))
    reads += [dna(rng.integers(0, 300)) for _ in range(150)] + [b"", dna(k - 1)]
    knob = [None, ("PFQ_GROUP_LOG2", "11"), ("PFQ_COARSE_COLS", "2048"), ("PFQ_COARSE_COLS", "512"), ("PFQ_COARSE_PROBES", "2"),
            ("PFQ_BLOCK", "1"), ("PFQ_TILE_COUNTS", "0"), ("PFQ_COARSE", "1")][int(rng.integers(0, 8))]
    if knob:
        gt.set_option(*knob)
    for thr in (1.0, float(rng.choice([0.05, 0.3, 0.5, 0.9])), float(rng.choice([0.0, 0.2, 0.75, 0.999, 1.5]))):
        for path in (1, 0):
            st = check_query(gt, ot, reads, thr, path=path)
            assert st.leaf_groups >= 2, (seed, st.leaf_groups)
    gt.close()
