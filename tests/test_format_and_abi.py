"""On-disk format restatement round trip, host-side mirrors, and the C-ABI surface of libpfq.  CPU only:
no compute call is made (libpfq refuses to compute without a device)."""
import os
import re

import numpy as np
import pytest

from oracle import pfq_format as fmt
from oracle import pfq_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def small_tree():
    genomes = [b"ATCAGGATTACA", b"TTTAGCCGGAAT", b"CTCAGTTTTTTT"]
    nbits = orc.needed_bits(0.001, 1000)
    return orc.build_balanced_tree(genomes, ["a", "b", "c"], 5, nbits, orc.optimal_num_hashes(nbits, 1000), 5, 10,
                                   0.001, 1000)


def test_db_roundtrip(tmp_path):
    t = small_tree()
    fmt.write_db(t, str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == sorted(["tree.bin"] + [p for p in set(t.bf_path)])
    u = fmt.read_db(str(tmp_path))
    assert (u.kmer_size, u.nbits, u.num_hashes, u.seed1, u.seed2) == (t.kmer_size, t.nbits, t.num_hashes, 5, 10)
    assert u.left == t.left and u.right == t.right and u.tax_id == t.tax_id and u.bf_path == t.bf_path
    for v in range(t.n_nodes):
        assert np.array_equal(u.bits[u.filter_of[v]], t.bits[t.filter_of[v]])
    # byte-level facts of the bincode/bitvec layout
    raw = open(os.path.join(tmp_path, t.bf_path[0]), "rb").read()
    assert raw[:8] == (19).to_bytes(8, "little") and raw[8:27] == b"bitvec::order::Lsb0"
    assert raw[27] == 64 and raw[28] == 0 and int.from_bytes(raw[29:37], "little") == t.nbits


def test_shared_filter_paths(tmp_path):
    t = small_tree()
    t.bf_path[2] = t.bf_path[1]          # two nodes naming one .bf
    t.filter_of[2] = t.filter_of[1]
    fmt.write_db(t, str(tmp_path))
    u = fmt.read_db(str(tmp_path))
    assert u.filter_of[2] == u.filter_of[1]


def test_abi_exports_every_declared_symbol():
    from phagefilter_amd import _ffi
    L = _ffi.lib()
    header = open(os.path.join(ROOT, "include", "pfq.h")).read()
    declared = set(re.findall(r"^(?:int|void|uint32_t|const char \*)\s*(pfq_[a-z_0-9]+)\(", header, re.M))
    assert declared == set(_ffi.SYMBOLS), declared ^ set(_ffi.SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s
    assert b"gfx950" in L.pfq_version()


def test_result_map_ext_id():  # result_map.rs:52-123
    from phagefilter_amd import ResultMap
    m = ResultMap()
    m.add_read_map("read1", "genomeA")
    assert m.get_ext_id("read1") == "read1 |genomeA"
    assert m.get_ext_id("unknown") == "unknown |"
    m.add_read_map("read1", "genomeA")
    m.add_read_map("read1", "genomeB")
    assert set(m.get_ext_id("read1").split("|")[1].split(",")) == {"genomeA", "genomeB"}
    assert m.read_mapped("read1") and not m.read_mapped("read2")
    m.empty_read_map()
    assert not m.read_mapped("read1")


def test_no_device_fails_loudly():
    """The product has no CPU fallback: without a device every compute entry point errors out."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0:
            pytest.skip("a device is present")
    except OSError:
        pass
    from phagefilter_amd import BloomTree, PfqError
    with pytest.raises(PfqError) as e:
        BloomTree.build_balanced([b"ACGTACGT"], ["g"], 5, 1000, 3, 1, 2)
    assert e.value.code == -5
