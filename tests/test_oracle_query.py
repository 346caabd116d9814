"""The reference's query-level tests (query.rs:267-380, bloom_tree prune) re-expressed on the oracle.  CPU only.
The reference builds its test trees with the greedy `insert` (out of scope); the same genomes are placed in the
balanced synthetic tree, which the assertions do not depend on."""
import numpy as np
import pytest

from oracle import pfq_oracle as orc


def tree_of(genomes, ids, k, seeds=(5, 10), fpr=0.001, items=1000):
    nbits = orc.needed_bits(fpr, items)
    return orc.build_balanced_tree(genomes, ids, k, nbits, orc.optimal_num_hashes(nbits, items), seeds[0], seeds[1],
                                   fpr, items)


@pytest.mark.parametrize("seeds", [(5, 10), (1, 2), (0xABCDEF, 0x123456789)])
def test_query_passes(seeds):  # query.rs:267-290
    for thr, want_same, want_diff in ((1.0, 1, 0), (0.0, 1, 1)):
        t = tree_of([b"ATCGCA"], ["genome"], 3, seeds)
        orc.query_batch(t, [b"ATCG"], thr)
        assert t.mapped_reads[0] == want_same
        t = tree_of([b"ATCGCA"], ["genome"], 3, seeds)
        orc.query_batch(t, [b"AAAA"], thr)
        assert t.mapped_reads[0] == want_diff


FOUR = ([b"ATCAG", b"TTTAG", b"CTCAG", b"ATTAG"], ["baseline", "diff", "onediff_first", "onediff_mid"])


def counts(t):
    return dict(t.leaf_counts())


def test_query_and_leaf_counts():  # query.rs:292-311
    t = tree_of(*FOUR, 5)
    orc.query_batch(t, [b"ATCAG"], 0.1)
    c = counts(t)
    assert c["baseline"] >= 1 and c["diff"] == 0


def test_query_smaller_kmer():  # query.rs:313-333
    t = tree_of(*FOUR, 4)
    orc.query_batch(t, [b"TCAG"], 0.1)
    c = counts(t)
    assert c["baseline"] >= 1 and c["onediff_first"] >= 1 and c["diff"] == 0


def test_query_multiple_reads():  # query.rs:335-354
    t = tree_of(*FOUR, 4)
    orc.query_batch(t, [b"TCAG", b"ATCA"], 0.51)
    c = counts(t)
    assert c["baseline"] >= 1 and c["diff"] == 0


def test_counts_accumulate_across_calls():  # query.rs:356-380
    t = tree_of(*FOUR, 4)
    orc.query_batch(t, [b"TCAG"], 0.1)
    orc.query_batch(t, [b"ATCA"], 0.1)
    c = counts(t)
    assert c["baseline"] >= 2 and c["diff"] == 0


def test_short_reads_and_zero_threshold_hit_every_leaf():
    t = tree_of(*FOUR, 5)
    hits, _, _ = orc.query_batch(t, [b"ACG", b""], 1.0)
    assert [c for _, c in t.leaf_counts()] == [2, 2, 2, 2]
    assert len(hits) == 8


def test_faithful_equals_fast_and_threads():
    rng = np.random.default_rng(1)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genomes = [rng.choice(acgt, 300).astype(np.uint8).tobytes() for _ in range(7)]
    ids = [f"g{i}" for i in range(7)]
    reads = []
    for i in range(60):
        g = genomes[i % 7]
        o = int(rng.integers(0, 250))
        r = bytearray(g[o:o + 50])
        if i % 3 == 0:
            r[10] = ord("N")
        if i % 5 == 0:
            r = bytearray(orc.revcomp(bytes(r)))
        reads.append(bytes(r))
    reads += [rng.choice(acgt, 50).astype(np.uint8).tobytes() for _ in range(20)]
    for thr in (1.0, 0.5, 0.2):
        res = []
        for faithful, threads in ((True, 1), (False, 1), (False, 3)):
            t = tree_of(genomes, ids, 11, items=400)
            hits, probes, _ = orc.query_batch(t, reads, thr, faithful=faithful, threads=threads)
            res.append((hits, t.leaf_counts(), probes))
        assert res[0][:2] == res[1][:2] == res[2][:2]
        assert res[0][2] == res[1][2]  # same reference-semantics probe count


def test_prune_tree():  # bloom_tree.rs:302-330
    genomes = [bytes([65 + i % 4]) * 30 for i in range(8)]
    t = tree_of(genomes, [f"g{i}" for i in range(8)], 5)
    assert len(t.leaves_dfs()) == 8
    t.prune(2)
    lv = t.leaves_dfs()
    assert len(lv) == 4 and all(t.tax_id[v].startswith("Internal_Node_") for v in lv)
    t.prune(0)
    assert t.leaves_dfs() == [t.root]


def test_csv_format():  # query.rs:173-183
    t = tree_of(*FOUR, 5)
    orc.query_batch(t, [b"ATCAG", b"ATCAG", b"TTTAG"], 1.0)
    lines = t.classification_csv().splitlines()
    assert "baseline,2" in lines and "diff,1" in lines
    assert all(not l.endswith(",0") for l in lines)


# ---------------------------------------------------------------------------------------------------------------
# greedy insertion (`build` / `add`): the reference's own fixtures, bloom_tree.rs:457-734
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seeds", [(5, 10), (0x0123456789ABCDEF, 0xFEDCBA9876543210), (1, 1)])
def test_greedy_insert_reference_fixtures(seeds):
    t = orc.build_greedy_tree([], [], 5, 0.001, 1000, *seeds)
    assert t.root == -1 and t.n_nodes == 0
    t = orc.build_greedy_tree([b"ATCAG"], ["test1"], 5, 0.001, 1000, *seeds)                     # :457-522
    assert (t.n_nodes, t.is_leaf(0), t.tax_id[0], t.bf_path[0]) == (1, True, "test1", "test1.bf")
    t = orc.build_greedy_tree([b"ATCAG", b"TTTAG"], ["test1", "test2"], 5, 0.001, 1000, *seeds)  # :523-585
    assert [t.tax_id[v] for v in (t.left[0], t.right[0])] == ["test1", "test2"] and not t.is_leaf(0)
    assert np.array_equal(t.bits[t.filter_of[0]], t.bits[t.filter_of[1]] | t.bits[t.filter_of[2]])
    # :586-660 — the third genome equals the first: it joins it under the left child
    t = orc.build_greedy_tree([b"ATCAG", b"TTTAG", b"ATCAG"], ["test1", "test2", "test3"], 5, 0.001, 1000, *seeds)
    root = t.root
    assert t.is_leaf(t.right[root]) and t.tax_id[t.right[root]] == "test2"
    left = t.left[root]
    assert sorted(t.tax_id[c] for c in (t.left[left], t.right[left])) == ["test1", "test3"]
    # :661-734 — equals the second: joins it under the right child
    t = orc.build_greedy_tree([b"ATCAG", b"TTTAG", b"TTTAG"], ["test1", "test2", "test3"], 5, 0.001, 1000, *seeds)
    root = t.root
    assert t.is_leaf(t.left[root]) and t.tax_id[t.left[root]] == "test1"
    right = t.right[root]
    assert sorted(t.tax_id[c] for c in (t.left[right], t.right[right])) == ["test2", "test3"]
    # every internal filter is the union of its subtree's leaves
    for v in range(t.n_nodes):
        if not t.is_leaf(v):
            assert np.array_equal(t.bits[t.filter_of[v]], t.bits[t.filter_of[t.left[v]]] | t.bits[t.filter_of[t.right[v]]])
