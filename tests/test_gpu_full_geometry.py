"""GPU parity at the FULL geometry of BASELINE config 3 below threshold 1 (missing #3): the 1024-leaf tree of
50 kbp genomes with nbits = 71 887 936 and 10 hashes is built on the device, every node's filter is copied back
(pfq_debug_node_filter, as bench.py's own check does) and the oracle's DFS runs on that copy.  300 000 reads of 150 bp —
more than 2^18, so the library takes the bucketed pipeline on its own: 549 / 1097 filter tiles, prefix certificates, the
counting screen's "reads at their limit go on" rule — half of them from the genomes with 1 % substitutions, at the
reference's operating thresholds 0.3 (benchmarking/config.yaml:4) and 0.7 (misc/slurm_scripts/run_phagefilter.sh:25-32)
and at 1.0.  Per-leaf counts AND per-read hit sets, bit-exact; the automatic path, the record kernel alone
(PFQ_TILE_COUNTS=0) and block mode (PFQ_BLOCK=1).  A second tree of 128 families of 8 related genomes at the same
geometry: a read is a candidate for up to 8 leaves."""
import ctypes as C

import numpy as np
import pytest

from hipbuf import DeviceBuffer, synchronize
from oracle import pfq_oracle as orc
from phagefilter_amd import BloomTree, _ffi

pytestmark = pytest.mark.gpu

K, NBITS, H = 21, 71887936, 10
SEEDS = (0x0123456789ABCDEF, 0xFEDCBA9876543210)
N_LEAVES, GLEN, RLEN, N_READS = 1024, 50000, 150, 300000
COMP = np.arange(256, dtype=np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    COMP[a] = b


def _oracle_copy(gt, ids):
    """The oracle's tree over the device's own filters (every node copied back from HBM)."""
    ot = orc.balanced_topology(ids, K, NBITS, H, SEEDS[0], SEEDS[1], 0.001, 5000000, alloc_bits=False)
    ot.bits = np.empty((ot.n_nodes, ot.n_words), dtype=np.uint64)
    for v in range(ot.n_nodes):
        ot.bits[v] = gt.node_filter(v)
        ot.filter_of[v] = v
    return ot


def _reads(genomes, rng, n, err):
    """n reads of RLEN: even indices from the genomes (uniform leaf / offset / strand, `err` substitutions per base), odd
    ones uniform random.  Returns (seq, off) packed."""
    n_pos = n // 2
    g = rng.integers(0, genomes.shape[0], n_pos)
    o = rng.integers(0, genomes.shape[1] - RLEN + 1, n_pos)
    pos = genomes[g[:, None], o[:, None] + np.arange(RLEN)[None, :]]
    rc = rng.random(n_pos) < 0.5
    pos[rc] = COMP[pos[rc]][:, ::-1]
    sub = rng.random(pos.shape) < err
    alt = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(sub.sum()))]
    pos[sub] = np.where(alt == pos[sub], COMP[alt], alt)           # always a different base
    neg = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n - n_pos, RLEN))]
    reads = np.empty((n, RLEN), dtype=np.uint8)
    reads[0::2] = pos
    reads[1::2] = neg
    seq = np.concatenate([reads.reshape(-1), np.zeros(16, dtype=np.uint8)])
    off = np.arange(n + 1, dtype=np.uint64) * RLEN
    return seq, off


def _gpu_hits(gt, seq, off, thr):
    gt.reset_counts()
    offs, leaves = gt.query_packed(seq, off, thr, want_hits=True)
    reads = np.repeat(np.arange(len(off) - 1, dtype=np.int64), np.diff(offs).astype(np.int64))
    return gt.get_leaf_counts(), np.stack([reads, leaves.astype(np.int64)], axis=1), gt.last_stats()


def _oracle_hits(ot, seq, off, thr):
    for v in range(ot.n_nodes):
        ot.mapped_reads[v] = 0
    hits, _, _ = orc.query_batch_packed(ot, seq, off, thr, threads=16)
    col = {v: i for i, v in enumerate(ot.leaves_dfs())}
    h = np.array([(r, col[v]) for r, v in hits], dtype=np.int64).reshape(-1, 2)
    return ot.leaf_counts(), h


def _compare(gt, ot, seq, off, thr, variants):
    want_counts, want_hits = _oracle_hits(ot, seq, off, thr)
    # below threshold 1 nearly every read that stems from a genome hits; at 1.0 the error-free ones do (0.99^150 = 22 %)
    assert len(want_hits) > (0.3 if thr < 1 else 0.08) * (len(off) - 1)
    out = {}
    for name, opts, expect_tile_mode in variants:
        for key, val in opts.items():
            gt.set_option(key, val)
        try:
            counts, hits, st = _gpu_hits(gt, seq, off, thr)
        finally:
            for key in opts:
                gt.set_option(key, None)
        assert st.path == 1, (thr, name)                           # >= 2^18 reads: the bucketed pipeline, chosen by the library
        assert st.tile_mode == expect_tile_mode, (thr, name, st.tile_mode)
        assert counts == want_counts, (thr, name)
        assert np.array_equal(hits, want_hits), (thr, name)
        out[name] = st
    return out


def test_config3_geometry_thresholds_below_one_vs_oracle(gpu):
    L = _ffi.lib()
    d_gen = DeviceBuffer(N_LEAVES * GLEN)
    _ffi.check(L.pfq_synth_genomes_device(d_gen.ptr, N_LEAVES, GLEN, 0x5EED0000, None))
    synchronize()
    ids = [f"G{i:05d}" for i in range(N_LEAVES)]
    gt = BloomTree.build_balanced_device(d_gen.ptr, GLEN, N_LEAVES, ids, K, NBITS, H, SEEDS[0], SEEDS[1], 0.001, 5000000)
    genomes = d_gen.to_numpy().reshape(N_LEAVES, GLEN)
    d_gen.free()
    ot = _oracle_copy(gt, ids)
    rng = np.random.default_rng(20261005)
    seq, off = _reads(genomes, rng, N_READS, 0.01)
    for thr in (0.3, 0.7, 1.0):
        variants = [("auto", {}, 1), ("block", {"PFQ_BLOCK": "1"}, 2)]
        if thr < 1:
            variants.append(("records", {"PFQ_TILE_COUNTS": "0"}, 0))
        st = _compare(gt, ot, seq, off, thr, variants)
        assert st["auto"].n_fallback_pairs < 0.02 * N_READS
    gt.close()


def test_config3_geometry_families_of_8_vs_oracle(gpu):
    """128 families of 8 genomes 0.5 % apart at nbits 71 887 936: a read is a candidate for up to 8 leaves (block mode is what
    the library turns to on such a workload, from the first call on)."""
    rng = np.random.default_rng(77)
    base = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (N_LEAVES // 8, GLEN))]
    genomes = np.repeat(base, 8, axis=0)
    mut = rng.random(genomes.shape) < 0.005
    mut[0::8] = False                                              # the first strain of a family is the ancestor itself
    alt = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(mut.sum()))]
    genomes[mut] = np.where(alt == genomes[mut], COMP[alt], alt)
    d_gen = DeviceBuffer.from_numpy(genomes)
    ids = [f"S{i:05d}" for i in range(N_LEAVES)]
    gt = BloomTree.build_balanced_device(d_gen.ptr, GLEN, N_LEAVES, ids, K, NBITS, H, SEEDS[0], SEEDS[1], 0.001, 5000000)
    d_gen.free()
    ot = _oracle_copy(gt, ids)
    seq, off = _reads(genomes, rng, N_READS, 0.01)
    for thr in (0.3, 1.0):
        want_counts, want_hits = _oracle_hits(ot, seq, off, thr)
        # several strains per read below threshold 1; at 1.0 the strains that are identical over an error-free read
        assert len(want_hits) > (1.5 if thr < 1 else 0.4) * (N_READS // 2)
        for name, opts in (("first-call", {}), ("block", {"PFQ_BLOCK": "1"}), ("second-call", {})):
            for key, val in opts.items():
                gt.set_option(key, val)
            try:
                counts, hits, st = _gpu_hits(gt, seq, off, thr)
            finally:
                for key in opts:
                    gt.set_option(key, None)
            assert st.path == 1 and counts == want_counts and np.array_equal(hits, want_hits), (thr, name)
            # block mode every time: the first call on the tree has no history and screens a sample of its own reads
            # (several candidate leaves per read), the forced call, then the choice from what the call before saw
            assert st.tile_mode == 2, (thr, name)
    gt.close()
