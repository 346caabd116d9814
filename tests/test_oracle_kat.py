"""Pins the CPU oracle (oracle/pfq_oracle.c) against (a) upstream rustc-hash known answers and (b) every exact or
relational fixture the reference's own unit tests hold for the query path.  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import pfq_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")
M64 = (1 << 64) - 1


def test_fxhash_upstream_kats():
    kat = json.load(open(os.path.join(GOLD, "fxhash_kat.json")))
    for v, want in kat["write_u8"]:
        assert orc.fx_finish_write_u64(v) == want
    for hexbytes, want in kat["write_bytes"]:
        assert orc.fx_finish_write_bytes(bytes.fromhex(hexbytes)) == want


def test_composite_vectors():
    g = json.load(open(os.path.join(GOLD, "composite_vectors.json")))
    for v in g["vectors"]:
        kmer = v["kmer"].encode()
        canon = orc.get_lex_less(kmer)
        assert canon == v["canon"].encode()
        assert orc.fx_hash_bytes(canon) == v["hb"]
        if "h1" in v:
            assert orc.seeded_hash(v["seed1"], canon) == v["h1"]
            assert orc.seeded_hash(v["seed2"], canon) == v["h2"]
        assert orc.probe_indices(v["seed1"], v["seed2"], v["num_hashes"], v["nbits"], canon) == v["idx"]
    for s in g["sizing"]:
        assert orc.needed_bits(s["fpr"], s["items"]) == s["bits"]
        assert orc.optimal_num_hashes(s["bits"], s["items"]) == s["hashes"]


# ---- hash_iter.rs:66-101 -------------------------------------------------------------------------------------
@pytest.mark.parametrize("item", [b"hello", b"world", b"ACGTACGTACGTACGTACGTA", b"x" * 40])
def test_hash_iter_formula(item):
    s1, s2, nbits = 0x1234567, 0xDEADBEEFCAFE, (1 << 61) - 1
    h1, h2 = orc.seeded_hash(s1, item), orc.seeded_hash(s2, item)
    for count in (0, 1, 2, 5, 17):
        vals = orc.probe_indices(s1, s2, count, nbits, item)
        assert len(vals) == count                                   # test_count_items
        if count >= 2:
            assert vals[0] == h1 % nbits and vals[1] == h2 % nbits  # test_first_is_h1_second_is_h2
        for i in range(2, count):
            assert vals[i] == (((h1 + i) & M64) * h2 & M64) % nbits  # test_formula_for_i_ge_2


def test_hasher_seeds_differ():  # hasher.rs:36-48
    assert orc.seeded_hash(5, b"Hello world!") != orc.seeded_hash(10, b"Hello world!")


# ---- bloom_filter.rs:378-475 -----------------------------------------------------------------------------------
def test_distance_kats():
    a = np.array([0b00101101], dtype=np.uint64)
    b = np.array([0b10100111], dtype=np.uint64)
    assert orc.distance(a, b) == 3
    assert orc.distance(np.array([0], dtype=np.uint64), np.array([0xFF], dtype=np.uint64)) == 8


def _tiny_tree(k=3, fpr=0.001, items=1000, s1=5, s2=10, n_filters=1):
    nbits = orc.needed_bits(fpr, items)
    t = orc.OracleTree(k, nbits, orc.optimal_num_hashes(nbits, items), s1, s2, fpr, items)
    t.bits = np.zeros((n_filters, t.n_words), dtype=np.uint64)
    return t


def test_insert_contains_union_needed_bits():
    t = _tiny_tree(n_filters=2)
    for item in (b"abc", b"ACG", bytes([1, 2, 3])):
        assert not orc.bf_contains(t, 0, item) or True
        orc.lib().orc_bf_insert(t.bits[0].ctypes.data_as(orc.C.POINTER(orc.C.c_uint64)), t.nbits, t.num_hashes, t.seed1,
                                t.seed2, item, len(item))
        assert orc.bf_contains(t, 0, item)                     # insert => contains (:397-409)
    t.bits[1] |= t.bits[0]                                      # union superset (:411-428)
    assert orc.bf_contains(t, 1, b"abc")
    assert orc.needed_bits(0.01, 1000) > 1000                  # :466-475 ("~9585")
    assert abs(orc.needed_bits(0.01, 1000) - 9585) <= 1


# ---- file_parser.rs:380-407 --------------------------------------------------------------------------------------
def test_get_kmers_kats():
    assert orc.get_kmers(b"", 1) == []
    assert orc.get_kmers(bytes([1, 2, 3]), 0) == []
    assert orc.get_kmers(bytes([1, 2, 3]), 1) == [bytes([1]), bytes([2]), bytes([3])]
    assert orc.get_kmers(bytes([1, 2, 3]), 2) == [bytes([1, 2]), bytes([2, 3])]
    assert orc.get_kmers(bytes([1, 2, 3]), 3) == [bytes([1, 2, 3])]
    assert orc.get_kmers(b"ACG", 4) == []


def test_get_lex_less_kats():
    assert orc.get_lex_less(b"ACGT") == b"ACGT"
    assert orc.get_lex_less(b"AATG") == b"AATG"
    assert orc.get_lex_less(b"GTAG") == b"CTAC"


def test_complement_table():
    t = orc.complement_table()
    for a, b in zip(b"AGCTYRWSKMDVHBN", b"TCGARYWSMKHBDVN"):
        assert t[a] == b and t[a + 32] == b + 32
    for c in list(range(0, 65)) + [ord("E"), ord("X"), ord("e"), 200, 255]:
        assert t[c] == c


# ---- query.rs:38-49 threshold arithmetic ---------------------------------------------------------------------------
def test_need_f32():
    f32 = np.float32
    for thr in (0.0, 0.1, 0.3, 0.51, 0.7, 0.999, 1.0, 1.5, -0.5):
        for n in (0, 1, 2, 3, 10, 81, 130, 131, 1000, 99999):
            want = int(max(0.0, float(np.ceil(f32(thr) * f32(n)))))
            assert orc.need(thr, n) == want, (thr, n)
    assert orc.need(0.3, 81) == 25      # 0.3f32 * 81 = 24.300001
    assert orc.need(float("nan"), 10) == 0
    assert orc.need(1.0, 130) == 130
