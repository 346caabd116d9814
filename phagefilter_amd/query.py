"""Host-side mirror of the reference's query interface over libpfq.

Names follow the reference: `BloomTree.load` (bloom_tree.rs:364-386), `prune_tree` (:302-330),
`query_batch` (query.rs:66-82), `get_leaf_counts` / `save_leaf_counts` (query.rs:173-218), `ResultMap`
(result_map.rs:9-46).  All filter work happens in the HIP kernels of libpfq; nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Set, Tuple

import numpy as np

from . import _ffi


class ResultMap:
    """result_map.rs:9-46 — read id -> set of genome ids for the current block."""

    def __init__(self) -> None:
        self.read_map: Dict[str, Set[str]] = {}

    def add_read_map(self, read_id: str, genome_id: str) -> None:
        self.read_map.setdefault(read_id, set()).add(genome_id)

    def get_ext_id(self, read_id: str) -> str:  # "{id} |{g1,g2}" (set order is unspecified in the reference too)
        return f"{read_id} |{','.join(self.read_map.get(read_id, ()))}"

    def read_mapped(self, read_id: str) -> bool:
        return read_id in self.read_map

    def empty_read_map(self) -> None:
        self.read_map.clear()


def pack_reads(reads: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    if reads:
        off[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
    seq = np.frombuffer(b"".join(reads) + b"\0" * 16, dtype=np.uint8).copy()
    return seq, off


class BloomTree:
    """A Sequence Bloom Tree resident in HBM (BloomTree, bloom_tree.rs:29-48)."""

    def __init__(self, handle: C.c_void_p, device: int):
        self._h = handle
        self.device = device

    # ---- construction
    @classmethod
    def load(cls, directory: str, device: int = 0) -> "BloomTree":
        h = C.c_void_p()
        _ffi.check(_ffi.lib().pfq_tree_open(directory.encode(), device, C.byref(h)))
        return cls(h, device)

    @classmethod
    def load_subtree(cls, directory: str, depth: int, index: int, device: int = 0) -> "BloomTree":
        """Shard `index` of the depth-`depth` frontier (pfq_tree_open_subtree): that node, its subtree and its
        ancestor chain.  For trees that do not fit one GPU: one shard per rank, every rank sees all reads."""
        h = C.c_void_p()
        _ffi.check(_ffi.lib().pfq_tree_open_subtree(directory.encode(), device, depth, index, C.byref(h)))
        return cls(h, device)

    @classmethod
    def new(cls, kmer_size: int, false_pos_rate: float, largest_expected_genome: int, seed1: int, seed2: int,
            expected_genomes: int = 0, device: int = 0) -> "BloomTree":
        """BloomTree::new (bloom_tree.rs:100-118) with explicit hash seeds; fill it with insert()."""
        h = C.c_void_p()
        _ffi.check(_ffi.lib().pfq_tree_create(kmer_size, false_pos_rate, largest_expected_genome, seed1, seed2,
                                              expected_genomes, device, C.byref(h)))
        return cls(h, device)

    def insert(self, genome: bytes, tax_id: str, internal_name: Optional[str] = None) -> None:
        """BloomTree::insert (bloom_tree.rs:128-143): greedy placement by Hamming distance, on the device."""
        buf = np.frombuffer(genome, dtype=np.uint8) if len(genome) else np.zeros(1, dtype=np.uint8)
        _ffi.check(_ffi.lib().pfq_tree_insert(self._h, buf.ctypes.data, len(genome), tax_id.encode(),
                                              internal_name.encode() if internal_name is not None else None))

    @classmethod
    def build_balanced(cls, genomes: Sequence[bytes], tax_ids: Sequence[str], kmer_size: int, nbits: int,
                       num_hashes: int, seed1: int, seed2: int, false_pos_rate: float = 0.001,
                       largest_expected_genome: int = 1000000, device: int = 0) -> "BloomTree":
        seq, off = pack_reads(genomes)
        ids = (C.c_char_p * max(len(tax_ids), 1))(*[t.encode() for t in tax_ids])
        h = C.c_void_p()
        _ffi.check(_ffi.lib().pfq_tree_build_balanced(seq.ctypes.data, off.ctypes.data, len(genomes), ids, kmer_size,
                                                      nbits, num_hashes, seed1, seed2, false_pos_rate,
                                                      largest_expected_genome, device, C.byref(h)))
        return cls(h, device)

    @classmethod
    def build_balanced_device(cls, d_genomes: int, genome_len: int, n_genomes: int, tax_ids: Sequence[str],
                              kmer_size: int, nbits: int, num_hashes: int, seed1: int, seed2: int,
                              false_pos_rate: float = 0.001, largest_expected_genome: int = 1000000,
                              device: int = 0) -> "BloomTree":
        ids = (C.c_char_p * max(len(tax_ids), 1))(*[t.encode() for t in tax_ids])
        h = C.c_void_p()
        _ffi.check(_ffi.lib().pfq_tree_build_balanced_device(d_genomes, genome_len, n_genomes, ids, kmer_size, nbits,
                                                             num_hashes, seed1, seed2, false_pos_rate,
                                                             largest_expected_genome, device, C.byref(h)))
        return cls(h, device)

    @classmethod
    def build_balanced_subtree_device(cls, d_genomes: int, genome_len: int, n_genomes: int, tax_ids: Sequence[str],
                                      kmer_size: int, nbits: int, num_hashes: int, seed1: int, seed2: int, depth: int,
                                      index: int, false_pos_rate: float = 0.001, largest_expected_genome: int = 1000000,
                                      device: int = 0) -> "BloomTree":
        """Subtree shard `index` of the depth-`depth` frontier of the balanced tree over all `n_genomes` genomes
        (BASELINE config 5), built without the rest of the tree."""
        ids = (C.c_char_p * max(len(tax_ids), 1))(*[t.encode() for t in tax_ids])
        h = C.c_void_p()
        _ffi.check(_ffi.lib().pfq_tree_build_balanced_subtree_device(d_genomes, genome_len, n_genomes, ids, kmer_size,
                                                                     nbits, num_hashes, seed1, seed2, false_pos_rate,
                                                                     largest_expected_genome, depth, index, device,
                                                                     C.byref(h)))
        return cls(h, device)

    def close(self) -> None:
        if self._h:
            _ffi.lib().pfq_tree_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference surface
    def save(self, directory: str) -> None:
        import os
        os.makedirs(directory, exist_ok=True)
        _ffi.check(_ffi.lib().pfq_tree_save(self._h, directory.encode()))

    def prune_tree(self, search_depth: int) -> None:
        _ffi.check(_ffi.lib().pfq_tree_prune(self._h, search_depth))

    def info(self) -> _ffi.Info:
        i = _ffi.Info()
        _ffi.check(_ffi.lib().pfq_tree_info(self._h, C.byref(i)))
        return i

    @property
    def kmer_size(self) -> int:
        return int(self.info().kmer_size)

    def get_leaf_counts(self) -> List[Tuple[str, int]]:
        ids = C.POINTER(C.c_char_p)()
        cnt = C.POINTER(C.c_uint64)()
        n = C.c_uint64()
        _ffi.check(_ffi.lib().pfq_leaf_counts(self._h, C.byref(ids), C.byref(cnt), C.byref(n)))
        return [(ids[i].decode(), int(cnt[i])) for i in range(n.value)]

    def save_leaf_counts(self, path: str) -> None:
        _ffi.check(_ffi.lib().pfq_save_leaf_counts(self._h, path.encode()))

    def reset_counts(self) -> None:
        _ffi.check(_ffi.lib().pfq_leaf_counts_reset(self._h))

    # ---- measurement / test hooks
    def set_option(self, name: str, value: Optional[str]) -> None:
        """One tuning / test knob of this tree; None = the built-in choice."""
        _ffi.check(_ffi.lib().pfq_set_option(self._h, name.encode(), None if value is None else str(value).encode()))

    def set_path(self, path: int) -> None:
        _ffi.check(_ffi.lib().pfq_set_path(self._h, path))

    def last_stats(self) -> _ffi.Stats:
        s = _ffi.Stats()
        _ffi.check(_ffi.lib().pfq_last_stats(self._h, C.byref(s)))
        return s

    def profile_begin(self, max_calls: int) -> None:
        _ffi.check(_ffi.lib().pfq_profile_begin(self._h, max_calls))

    def profile_end(self) -> _ffi.Profile:
        p = _ffi.Profile()
        _ffi.check(_ffi.lib().pfq_profile_end(self._h, C.byref(p)))
        return p

    def kmer_indices(self, seq: bytes) -> np.ndarray:
        n = C.c_uint64()
        buf = np.frombuffer(seq + b"\0", dtype=np.uint8).copy()
        _ffi.check(_ffi.lib().pfq_debug_kmer_indices(self._h, buf.ctypes.data, len(seq), None, C.byref(n)))
        out = np.zeros((n.value, int(self.info().num_hashes)), dtype=np.uint64)
        if n.value:
            _ffi.check(_ffi.lib().pfq_debug_kmer_indices(self._h, buf.ctypes.data, len(seq), out.ctypes.data, C.byref(n)))
        return out

    def node_filter(self, node: int) -> np.ndarray:
        i = self.info()
        out = np.zeros((int(i.nbits) + 63) // 64, dtype=np.uint64)
        _ffi.check(_ffi.lib().pfq_debug_node_filter(self._h, node, out.ctypes.data, out.size))
        return out

    # ---- query
    def query_packed(self, seq: np.ndarray, off: np.ndarray, threshold: float, want_hits: bool = False):
        """One block of reads from host memory.  Returns None or (offsets, leaves) CSR."""
        n = len(off) - 1
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        hits = _ffi.Hits()
        _ffi.check(_ffi.lib().pfq_query_batch(self._h, seq.ctypes.data, off.ctypes.data, n, threshold,
                                              _ffi.WANT_HITS if want_hits else 0, C.byref(hits)))
        if not want_hits:
            return None
        offs = np.ctypeslib.as_array(hits.offsets, shape=(n + 1,)).copy() if n else np.zeros(1, dtype=np.uint64)
        total = int(offs[-1])
        leaves = np.ctypeslib.as_array(hits.leaves, shape=(total,)).copy() if total else np.zeros(0, dtype=np.uint32)
        return offs, leaves

    def query_device(self, d_seq: int, d_off: int, n_reads: int, total_bytes: int, threshold: float,
                     stream: int = 0) -> None:
        """One block already resident in HBM (raw device pointers), asynchronous on `stream`."""
        _ffi.check(_ffi.lib().pfq_query_batch_device(self._h, d_seq, d_off, n_reads, total_bytes, threshold, 0,
                                                     stream, None))

    def query_device_hits(self, d_seq: int, d_off: int, n_reads: int, total_bytes: int, threshold: float, stream: int = 0):
        """The same block with PFQ_WANT_HITS: synchronous, returns the CSR (offsets, leaves) as views of the library's buffers
        (valid until the next call on this tree)."""
        hits = _ffi.Hits()
        _ffi.check(_ffi.lib().pfq_query_batch_device(self._h, d_seq, d_off, n_reads, total_bytes, threshold, _ffi.WANT_HITS,
                                                     stream, C.byref(hits)))
        offs = np.ctypeslib.as_array(hits.offsets, shape=(n_reads + 1,)) if n_reads else np.zeros(1, dtype=np.uint64)
        total = int(offs[-1])
        leaves = np.ctypeslib.as_array(hits.leaves, shape=(total,)) if total else np.zeros(0, dtype=np.uint32)
        return offs, leaves

    def export_counts(self, d_dst: int, stream: int = 0) -> None:
        _ffi.check(_ffi.lib().pfq_leaf_counts_export(self._h, d_dst, stream))

    def import_counts(self, d_src: int, stream: int = 0) -> None:
        _ffi.check(_ffi.lib().pfq_leaf_counts_import(self._h, d_src, stream))

    def export_counts_delta(self, d_dst: int, stream: int = 0) -> None:
        """What this replica counted since it was opened / last reset, imported or reduced (counters - base)."""
        _ffi.check(_ffi.lib().pfq_leaf_counts_export_delta(self._h, d_dst, stream))

    def import_counts_delta(self, d_src: int, stream: int = 0) -> None:
        """counters = base + d_src (the sum of the ranks' deltas); that becomes the new base."""
        _ffi.check(_ffi.lib().pfq_leaf_counts_import_delta(self._h, d_src, stream))


def query_batch(bloom_tree: BloomTree, read_set: Sequence[bytes], threshold: float,
                result_map: Optional[ResultMap] = None, read_ids: Optional[Sequence[str]] = None) -> BloomTree:
    """query::query_batch (query.rs:66-82).  Leaf counts accumulate in the tree; when `result_map` is given
    (the reference fills it when reads carry their sequence, query.rs:146-154) every (read id, tax id) hit is added."""
    seq, off = pack_reads(read_set)
    res = bloom_tree.query_packed(seq, off, threshold, want_hits=result_map is not None)
    if result_map is not None:
        offs, leaves = res
        names = [t for t, _ in bloom_tree.get_leaf_counts()]
        for r in range(len(read_set)):
            rid = read_ids[r] if read_ids is not None else str(r)
            for j in range(int(offs[r]), int(offs[r + 1])):
                result_map.add_read_map(rid, names[int(leaves[j])])
    return bloom_tree


def get_leaf_counts(bloom_tree: BloomTree) -> List[Tuple[str, int]]:
    return bloom_tree.get_leaf_counts()


def save_leaf_counts(bloom_tree: BloomTree, path: str) -> None:
    bloom_tree.save_leaf_counts(path)


def allreduce_counts(trees: Sequence[BloomTree]) -> int:
    """Sum the per-leaf counters of replicas of one database (one per GPU, or several on one GPU) so that every replica
    holds the totals; returns the number of RCCL ranks used.  The in-process counterpart of dist.all_reduce_counts."""
    hs = (C.c_void_p * len(trees))(*[t._h for t in trees])
    _ffi.check(_ffi.lib().pfq_trees_allreduce_counts(hs, len(trees)))
    return int(_ffi.lib().pfq_last_allreduce_ranks())
