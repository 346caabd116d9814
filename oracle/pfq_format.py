"""Restatement of the reference's on-disk database format — TEST INFRASTRUCTURE (see pfq_oracle.c).

`<db>/tree.bin`  = bincode(BloomTree)   bloom_tree.rs:29-61, written at :339-355, read at :364-386
`<db>/<name>.bf` = bincode(BloomFilter) bloom_filter.rs:84-93, written at :176-205, read at :153-174

bincode 1.3.3 defaults: little-endian, fixed-width integers, u64 length prefixes, Option = 1-byte tag,
usize as u64, struct fields in declaration order, #[serde(skip)] fields absent.  bitvec 1.0.1 serialises a
BitVec<usize, Lsb0> as struct BitSeq{order: str, head: BitIdx{width: u8, index: u8}, bits: u64, data: seq<usize>}.
Neither layout is pinned by a reference fixture (the reference only has save->load round-trip tests,
bloom_filter.rs:444-464, bloom_tree.rs:736-781): "parity unpinned" for the byte layout.

This module is used by tests to write databases the product's C++ reader must load, and to read back
what the product's writer produced.  The product has its own reader/writer (phagefilter_amd/csrc/pfq_db.cpp).
"""
from __future__ import annotations

import os
import struct
from typing import List, Optional, Tuple

import numpy as np

from .pfq_oracle import OracleTree

ORDER_NAME = b"bitvec::order::Lsb0"
TREE_FILENAME = "tree.bin"  # bloom_tree.rs:26


# ------------------------------------------------------------------ writers
def _str(b: bytes) -> bytes:
    return struct.pack("<Q", len(b)) + b


def _opt_str(s: Optional[str]) -> bytes:
    return b"\x00" if s is None else b"\x01" + _str(s.encode())


def encode_node(t: OracleTree, v: int) -> bytes:
    """BloomNode in field order: left_child, right_child, bloom_filter_path, tax_id, mapped_reads."""
    out = []
    for c in (t.left[v], t.right[v]):
        out.append(b"\x00" if c < 0 else b"\x01" + encode_node(t, c))
    out.append(_str(t.bf_path[v].encode()))
    out.append(_opt_str(t.tax_id[v]))
    out.append(struct.pack("<Q", t.mapped_reads[v]))
    return b"".join(out)


def encode_tree(t: OracleTree) -> bytes:
    """BloomTree: root, false_pos_rate f32, largest_expected_genome u32, kmer_size usize, hash_states (seed, seed)."""
    root = b"\x00" if t.root < 0 else b"\x01" + encode_node(t, t.root)
    return root + struct.pack("<fIQQQ", t.false_pos_rate, t.largest_expected_genome, t.kmer_size, t.seed1, t.seed2)


def encode_filter(words: np.ndarray, nbits: int, num_hashes: int, seed1: int, seed2: int,
                  file_path: Optional[str]) -> bytes:
    words = np.ascontiguousarray(words, dtype="<u8")
    assert words.size == (nbits + 63) // 64
    head = _str(ORDER_NAME) + struct.pack("<BBQ", 64, 0, nbits) + struct.pack("<Q", words.size)
    tail = struct.pack("<IQQ", num_hashes, seed1, seed2) + _opt_str(file_path)
    return head + words.tobytes() + tail


def write_db(t: OracleTree, directory: str) -> None:
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, TREE_FILENAME), "wb") as f:
        f.write(encode_tree(t))
    seen = set()
    for v in range(t.n_nodes):
        if t.bf_path[v] in seen:
            continue
        seen.add(t.bf_path[v])
        path = os.path.join(directory, t.bf_path[v])
        with open(path, "wb") as f:
            f.write(encode_filter(t.bits[t.filter_of[v]], t.nbits, t.num_hashes, t.seed1, t.seed2, path))


# ------------------------------------------------------------------ readers
class _Cur:
    def __init__(self, b: bytes):
        self.b, self.p = b, 0

    def take(self, n: int) -> bytes:
        if self.p + n > len(self.b):
            raise ValueError("truncated")
        s = self.b[self.p:self.p + n]
        self.p += n
        return s

    def u8(self) -> int:
        return self.take(1)[0]

    def u32(self) -> int:
        return struct.unpack("<I", self.take(4))[0]

    def u64(self) -> int:
        return struct.unpack("<Q", self.take(8))[0]

    def f32(self) -> float:
        return struct.unpack("<f", self.take(4))[0]

    def string(self) -> str:
        return self.take(self.u64()).decode()

    def opt_string(self) -> Optional[str]:
        tag = self.u8()
        if tag == 0:
            return None
        if tag != 1:
            raise ValueError("bad Option tag")
        return self.string()


def _decode_node(c: _Cur, t: OracleTree) -> int:
    v = t.add_node(None, "", -1)
    kids = []
    for _ in range(2):
        tag = c.u8()
        kids.append(_decode_node(c, t) if tag == 1 else -1)
    t.left[v], t.right[v] = kids
    t.bf_path[v] = c.string()
    t.tax_id[v] = c.opt_string()
    t.mapped_reads[v] = c.u64()
    return v


def decode_filter(b: bytes) -> Tuple[np.ndarray, int, int, int, int, Optional[str]]:
    c = _Cur(b)
    if c.take(c.u64()) != ORDER_NAME:
        raise ValueError("unexpected bit order")
    width, index, nbits = c.u8(), c.u8(), c.u64()
    n_words = c.u64()
    if width != 64 or index != 0 or n_words != (nbits + 63) // 64:
        raise ValueError("unexpected BitSeq head")
    words = np.frombuffer(c.take(8 * n_words), dtype="<u8").copy()
    num_hashes, s1, s2 = c.u32(), c.u64(), c.u64()
    path = c.opt_string()
    if c.p != len(b):
        raise ValueError("trailing bytes")
    return words, nbits, num_hashes, s1, s2, path


def read_db(directory: str) -> OracleTree:
    with open(os.path.join(directory, TREE_FILENAME), "rb") as f:
        c = _Cur(f.read())
    t = OracleTree(0, 0, 0, 0, 0)
    tag = c.u8()
    if tag == 1:
        t.root = _decode_node(c, t)
    t.false_pos_rate, t.largest_expected_genome, t.kmer_size = c.f32(), c.u32(), c.u64()
    t.seed1, t.seed2 = c.u64(), c.u64()
    if c.p != len(c.b):
        raise ValueError("trailing bytes in tree.bin")
    rows: List[np.ndarray] = []
    row_of = {}
    for v in range(t.n_nodes):
        p = t.bf_path[v]
        if p not in row_of:
            with open(os.path.join(directory, p), "rb") as f:
                words, nbits, nh, s1, s2, _ = decode_filter(f.read())
            if rows and (nbits != t.nbits or nh != t.num_hashes):
                raise ValueError("filters differ in size")
            t.nbits, t.num_hashes = nbits, nh
            if (s1, s2) != (t.seed1, t.seed2):
                raise ValueError("filter seeds differ from tree seeds")
            row_of[p] = len(rows)
            rows.append(words)
        t.filter_of[v] = row_of[p]
    t.bits = np.stack(rows) if rows else np.zeros((1, 1), dtype=np.uint64)
    return t
