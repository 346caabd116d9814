"""ctypes front-end of the CPU oracle (oracle/pfq_oracle.c).

TEST INFRASTRUCTURE ONLY — see the header of pfq_oracle.c.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module.  The product path (phagefilter_amd) never
does and fails loudly when its HIP library is missing.

Also holds the reference's data model in its simplest form: `OracleTree` = the `BloomTree` /
`BloomNode` topology of bloom_tree.rs:29-61 flattened into arrays, plus node-major filters in the
reference's bit order (`BitVec<usize, Lsb0>`, bloom_filter.rs:84-93).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpfq_oracle.so")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "pfq_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpfq_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u8p, u64p, i64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_int64)
        L.orc_fx_hash_bytes.restype = C.c_uint64
        L.orc_fx_hash_bytes.argtypes = [C.c_char_p, C.c_uint64]
        L.orc_fx_finish_write_bytes.restype = C.c_uint64
        L.orc_fx_finish_write_bytes.argtypes = [C.c_char_p, C.c_uint64]
        L.orc_fx_finish_write_u64.restype = C.c_uint64
        L.orc_fx_finish_write_u64.argtypes = [C.c_uint64]
        L.orc_seeded_hash.restype = C.c_uint64
        L.orc_seeded_hash.argtypes = [C.c_uint64, C.c_char_p, C.c_uint64]
        L.orc_probe_indices.restype = None
        L.orc_probe_indices.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_char_p, C.c_uint64, u64p]
        L.orc_complement_table.argtypes = [u8p]
        L.orc_revcomp.argtypes = [C.c_char_p, C.c_uint64, u8p]
        L.orc_get_lex_less.argtypes = [C.c_char_p, C.c_uint64, u8p]
        L.orc_kmer_count.restype = C.c_uint64
        L.orc_kmer_count.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_get_kmers.restype = C.c_uint64
        L.orc_get_kmers.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, u8p]
        L.orc_needed_bits.restype = C.c_uint64
        L.orc_needed_bits.argtypes = [C.c_float, C.c_uint32]
        L.orc_optimal_num_hashes.restype = C.c_uint32
        L.orc_optimal_num_hashes.argtypes = [C.c_uint64, C.c_uint32]
        L.orc_distance.restype = C.c_uint64
        L.orc_distance.argtypes = [u64p, u64p, C.c_uint64]
        L.orc_union.argtypes = [u64p, u64p, C.c_uint64]
        L.orc_bf_insert.restype = C.c_int
        L.orc_bf_insert.argtypes = [u64p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_char_p, C.c_uint64]
        L.orc_bf_contains.restype = C.c_int
        L.orc_bf_contains.argtypes = [u64p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_char_p, C.c_uint64]
        L.orc_bf_insert_sequence.argtypes = [u64p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_char_p, C.c_uint64, C.c_uint64]
        L.orc_need.restype = C.c_uint64
        L.orc_need.argtypes = [C.c_float, C.c_uint64]
        L.orc_query_batch.restype = C.c_int
        L.orc_query_batch.argtypes = [C.c_void_p, u8p, u64p, C.c_uint64, C.c_float, C.c_int, C.c_int, u64p, u64p,
                                      C.c_uint64, u64p, u64p, C.POINTER(C.c_double)]
        L.orc_synth_genome.argtypes = [C.c_uint64, C.c_uint64, u8p]
        L.orc_synth_reads.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, u8p, C.c_uint64, C.c_uint64, u8p]
        _lib = L
    return _lib


def _u8(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _u64(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


# ------------------------------------------------------------------------------------------------
# hashing / k-mers
# ------------------------------------------------------------------------------------------------
def fx_hash_bytes(b: bytes) -> int:
    return lib().orc_fx_hash_bytes(b, len(b))


def fx_finish_write_bytes(b: bytes) -> int:
    """FxHasher::default(); write(bytes); finish()."""
    return lib().orc_fx_finish_write_bytes(b, len(b))


def fx_finish_write_u64(v: int) -> int:
    """FxHasher::default(); write_u8/u16/u32/u64/usize(v); finish()."""
    return lib().orc_fx_finish_write_u64(v)


def seeded_hash(seed: int, item: bytes) -> int:
    """HashSeed{seed}.hash_one(&Vec<u8>)  (hasher.rs:12-21, hash_iter.rs:37-38)."""
    return lib().orc_seeded_hash(seed, item, len(item))


def probe_indices(seed1: int, seed2: int, num_hashes: int, nbits: int, item: bytes) -> List[int]:
    out = np.zeros(num_hashes, dtype=np.uint64)
    lib().orc_probe_indices(seed1, seed2, num_hashes, nbits, item, len(item), _u64(out))
    return [int(x) for x in out]


def complement_table() -> np.ndarray:
    t = np.zeros(256, dtype=np.uint8)
    lib().orc_complement_table(_u8(t))
    return t


def revcomp(b: bytes) -> bytes:
    out = np.zeros(max(len(b), 1), dtype=np.uint8)
    lib().orc_revcomp(b, len(b), _u8(out))
    return out[: len(b)].tobytes()


def get_lex_less(kmer: bytes) -> bytes:
    out = np.zeros(max(len(kmer), 1), dtype=np.uint8)
    lib().orc_get_lex_less(kmer, len(kmer), _u8(out))
    return out[: len(kmer)].tobytes()


def get_kmers(seq: bytes, k: int) -> List[bytes]:
    n = lib().orc_kmer_count(len(seq), k)
    out = np.zeros(max(n * k, 1), dtype=np.uint8)
    lib().orc_get_kmers(seq, len(seq), k, _u8(out))
    return [out[i * k:(i + 1) * k].tobytes() for i in range(n)]


def needed_bits(fpr: float, items: int) -> int:
    return lib().orc_needed_bits(fpr, items)


def optimal_num_hashes(bits: int, items: int) -> int:
    return lib().orc_optimal_num_hashes(bits, items)


def need(threshold: float, n_kmers: int) -> int:
    return lib().orc_need(threshold, n_kmers)


def distance(a: np.ndarray, b: np.ndarray) -> int:
    return lib().orc_distance(_u64(a), _u64(b), a.size)


# ------------------------------------------------------------------------------------------------
# tree model
# ------------------------------------------------------------------------------------------------
class _CTree(C.Structure):
    _fields_ = [("n_nodes", C.c_int64), ("root", C.c_int64), ("left", C.c_void_p), ("right", C.c_void_p),
                ("filter", C.c_void_p), ("bits", C.c_void_p), ("n_words", C.c_uint64), ("nbits", C.c_uint64),
                ("num_hashes", C.c_uint32), ("seed1", C.c_uint64), ("seed2", C.c_uint64), ("kmer_size", C.c_uint64)]


@dataclass
class OracleTree:
    """BloomTree + BloomNodes (bloom_tree.rs:29-61) as arrays; node 0.. in pre-order (root = 0)."""
    kmer_size: int
    nbits: int
    num_hashes: int
    seed1: int
    seed2: int
    false_pos_rate: float = 0.001
    largest_expected_genome: int = 1000000
    left: List[int] = field(default_factory=list)       # -1 = None
    right: List[int] = field(default_factory=list)
    tax_id: List[Optional[str]] = field(default_factory=list)
    bf_path: List[str] = field(default_factory=list)     # relative .bf filename (cache.rs:62 joins it to the db dir)
    mapped_reads: List[int] = field(default_factory=list)
    filter_of: List[int] = field(default_factory=list)  # node -> row of `bits`
    bits: Optional[np.ndarray] = None                    # [n_filters, n_words] uint64
    root: int = -1

    @property
    def n_words(self) -> int:
        return (self.nbits + 63) // 64

    @property
    def n_nodes(self) -> int:
        return len(self.left)

    def is_leaf(self, v: int) -> bool:  # bloom_tree.rs:416-418
        return self.left[v] < 0 and self.right[v] < 0

    def add_node(self, tax_id: Optional[str], bf_path: str, filter_row: int, left: int = -1, right: int = -1) -> int:
        self.left.append(left)
        self.right.append(right)
        self.tax_id.append(tax_id)
        self.bf_path.append(bf_path)
        self.mapped_reads.append(0)
        self.filter_of.append(filter_row)
        return len(self.left) - 1

    def leaves_dfs(self) -> List[int]:
        """Leaf nodes in get_leaf_counts order (query.rs:197-218): DFS, left before right."""
        out: List[int] = []
        if self.root < 0:
            return out
        stack = [self.root]
        while stack:
            v = stack.pop()
            if self.is_leaf(v):
                out.append(v)
            else:
                if self.right[v] >= 0:
                    stack.append(self.right[v])
                if self.left[v] >= 0:
                    stack.append(self.left[v])
        return out

    def prune(self, search_depth: int) -> None:
        """prune_tree (bloom_tree.rs:302-330): nodes at depth >= search_depth lose their children."""
        if self.root < 0:
            raise RuntimeError("prune_tree on an empty tree (reference unwraps root)")
        stack = [(self.root, 0)]
        while stack:
            v, d = stack.pop()
            if d < search_depth:
                if self.left[v] >= 0:
                    stack.append((self.left[v], d + 1))
                if self.right[v] >= 0:
                    stack.append((self.right[v], d + 1))
            else:
                self.left[v] = -1
                self.right[v] = -1

    def leaf_counts(self) -> List[tuple]:
        """get_leaf_counts (query.rs:197-218), zeros included."""
        out = []
        for v in self.leaves_dfs():
            if self.tax_id[v] is None:
                raise RuntimeError("leaf without tax_id (reference unwraps)")
            out.append((self.tax_id[v], self.mapped_reads[v]))
        return out

    def classification_csv(self) -> str:
        """save_leaf_counts (query.rs:173-183): '{id},{count}\\n' for count > 0, no header."""
        return "".join(f"{i},{c}\n" for i, c in self.leaf_counts() if c > 0)


def insert_sequence(tree: OracleTree, filter_row: int, seq: bytes) -> None:
    """init_leaf_node's k-mer insertion (bloom_tree.rs:154-168)."""
    row = tree.bits[filter_row]
    lib().orc_bf_insert_sequence(_u64(row), tree.nbits, tree.num_hashes, tree.seed1, tree.seed2, seq, len(seq),
                                 tree.kmer_size)


def bf_contains(tree: OracleTree, filter_row: int, item: bytes) -> bool:
    return bool(lib().orc_bf_contains(_u64(tree.bits[filter_row]), tree.nbits, tree.num_hashes, tree.seed1,
                                      tree.seed2, item, len(item)))


def balanced_topology(tax_ids: Sequence[str], kmer_size: int, nbits: int, num_hashes: int, seed1: int, seed2: int,
                      fpr: float = 0.001, largest: int = 1000000, alloc_bits: bool = True) -> OracleTree:
    """Shape and naming of the synthetic SBT of SURVEY §8d (same numbering as libpfq's balanced builder):
    complete-as-possible balanced binary tree over the leaves in order, nodes in pre-order, internal nodes named
    Internal_Node_<pre-order counter>, one filter row per node."""
    g = len(tax_ids)
    t = OracleTree(kmer_size, nbits, num_hashes, seed1, seed2, fpr, largest)
    counter = [0]

    def rec(lo: int, hi: int) -> int:
        if hi - lo == 1:
            v = t.add_node(tax_ids[lo], f"{tax_ids[lo]}.bf", -1)
            t.filter_of[v] = v
            return v
        name = f"Internal_Node_{counter[0]}"
        counter[0] += 1
        v = t.add_node(name, f"{name}.bf", -1)
        t.filter_of[v] = v
        mid = lo + (hi - lo + 1) // 2
        l = rec(lo, mid)
        r = rec(mid, hi)
        t.left[v], t.right[v] = l, r
        return v

    if g:
        t.root = rec(0, g)
    if alloc_bits:
        t.bits = np.zeros((max(t.n_nodes, 1), t.n_words), dtype=np.uint64)
    return t


def build_balanced_tree(genomes: Sequence[bytes], tax_ids: Sequence[str], kmer_size: int, nbits: int,
                        num_hashes: int, seed1: int, seed2: int, fpr: float = 0.001,
                        largest: int = 1000000) -> OracleTree:
    """Synthetic SBT of SURVEY §8d: leaf filters by init_leaf_node's insertion (bloom_tree.rs:154-168), internal
    filter = OR of children (node_union, bloom_tree.rs:238-239).  NOT the reference's greedy `insert` (out of
    scope); produces trees the reference's `query` accepts."""
    t = balanced_topology(tax_ids, kmer_size, nbits, num_hashes, seed1, seed2, fpr, largest)
    leaves = t.leaves_dfs()
    for i, v in enumerate(leaves):
        insert_sequence(t, v, genomes[i])
    for v in reversed(range(t.n_nodes)):  # pre-order numbering: children have larger indices than their parent
        if not t.is_leaf(v):
            t.bits[v] = t.bits[t.left[v]] | t.bits[t.right[v]]
    return t


def subtree_shard(t: OracleTree, depth: int, index: int):
    """Subtree shard of a tree (BASELINE config 5; the rule of pfq_tree_open_subtree in include/pfq.h): the shards are
    the nodes of the depth-`depth` frontier left to right (nodes at that depth, plus leaves above it); shard `index`
    keeps that node, everything below it and the chain of its ancestors, each reduced to the child on the path.
    Returns (shard tree sharing `t.bits`, position of its first leaf in the whole tree's leaf order); the reference's
    DFS (query.rs:99-158) over the shard visits the ancestors first, exactly as it would in the whole tree."""
    import copy
    frontier, parent = [], {t.root: -1}
    stack = [(t.root, 0)]
    while stack:
        v, d = stack.pop()
        if d == depth or t.is_leaf(v):
            frontier.append(v)
            continue
        for c in (t.right[v], t.left[v]):
            if c >= 0:
                parent[c] = v
                stack.append((c, d + 1))
    if not 0 <= index < len(frontier):
        raise IndexError(f"subtree index {index} out of range: the depth-{depth} frontier has {len(frontier)} nodes")

    def n_leaves(root: int) -> int:
        n, st = 0, [root]
        while st:
            v = st.pop()
            n += t.is_leaf(v)
            st += [c for c in (t.left[v], t.right[v]) if c >= 0]
        return n

    first = sum(n_leaves(f) for f in frontier[:index])
    sh = copy.copy(t)                       # arrays below are replaced; `bits` stays shared
    sh.left, sh.right, sh.mapped_reads = list(t.left), list(t.right), [0] * t.n_nodes
    c, v = frontier[index], parent[frontier[index]]
    while v >= 0:
        if sh.left[v] != c:
            sh.left[v] = -1
        if sh.right[v] != c:
            sh.right[v] = -1
        c, v = v, parent[v]
    return sh, first


def renumber_preorder(t: OracleTree) -> None:
    """Nodes in pre-order (root = 0) again; filter rows stay where they are."""
    order: List[int] = []
    if t.root >= 0:
        stack = [t.root]
        while stack:
            v = stack.pop()
            order.append(v)
            if t.right[v] >= 0:
                stack.append(t.right[v])
            if t.left[v] >= 0:
                stack.append(t.left[v])
    new_of = {v: i for i, v in enumerate(order)}
    t.left, t.right, t.tax_id, t.bf_path, t.mapped_reads, t.filter_of = (
        [new_of.get(t.left[v], -1) for v in order], [new_of.get(t.right[v], -1) for v in order],
        [t.tax_id[v] for v in order], [t.bf_path[v] for v in order], [t.mapped_reads[v] for v in order],
        [t.filter_of[v] for v in order])
    t.root = 0 if order else -1


def greedy_insert(t: OracleTree, genome: bytes, tax_id: str, internal_name: Optional[str] = None) -> None:
    """BloomTree::insert (bloom_tree.rs:128-143): init_leaf_node (:154-168), add_to_tree (:187-214) and
    init_internal_node (:226-245).  Filter rows are appended to t.bits; nodes are appended (call renumber_preorder
    when done).  internal_name replaces the reference's random "Internal_Node_<u16>" (:231-233); default
    "Internal_Node_<n>" with n counting the internal nodes made so far, skipping names already in the tree."""
    def new_row() -> int:
        t.bits = np.zeros((1, t.n_words), dtype=np.uint64) if t.bits is None or t.n_nodes == 0 else \
            np.vstack([t.bits, np.zeros((1, t.n_words), dtype=np.uint64)])
        return t.bits.shape[0] - 1

    r = new_row()
    nv = t.add_node(tax_id, f"{tax_id}.bf", r)
    insert_sequence(t, r, genome)
    if t.root < 0:
        t.root = nv
        return
    cur, parent, went_right = t.root, -1, False
    while True:
        if t.left[cur] >= 0 and t.right[cur] >= 0:
            t.bits[t.filter_of[cur]] |= t.bits[r]                                   # node_union(current, node), :195
            dr = distance(t.bits[t.filter_of[t.right[cur]]], t.bits[r])            # :198
            dl = distance(t.bits[t.filter_of[t.left[cur]]], t.bits[r])             # :199
            parent, went_right = cur, dr < dl                                       # :201 ties go left
            cur = t.right[cur] if went_right else t.left[cur]
        elif t.is_leaf(cur):
            if internal_name is None:
                n = getattr(t, "_internal_counter", 0)
                while f"Internal_Node_{n}.bf" in t.bf_path:
                    n += 1
                internal_name = f"Internal_Node_{n}"
                t._internal_counter = n + 1
            ri = new_row()
            t.bits[ri] = t.bits[r] | t.bits[t.filter_of[cur]]                       # :236-237
            ni = t.add_node(internal_name, f"{internal_name}.bf", ri, left=cur, right=nv)   # :241-242
            if parent < 0:
                t.root = ni
            elif went_right:
                t.right[parent] = ni
            else:
                t.left[parent] = ni
            return
        else:
            raise RuntimeError("Node with only one child encountered - should not happen.")  # :209


def build_greedy_tree(genomes: Sequence[bytes], tax_ids: Sequence[str], kmer_size: int, false_pos_rate: float,
                      largest_expected_genome: int, seed1: int, seed2: int) -> OracleTree:
    """`phage_filter build` (main.rs:148-200) with explicit seeds: BloomTree::new + insert per genome."""
    nbits = needed_bits(false_pos_rate, largest_expected_genome)
    t = OracleTree(kmer_size, nbits, optimal_num_hashes(nbits, largest_expected_genome), seed1, seed2, false_pos_rate,
                   largest_expected_genome)
    for g, i in zip(genomes, tax_ids):
        greedy_insert(t, g, i)
    renumber_preorder(t)
    return t


def query_batch(tree: OracleTree, reads: Sequence[bytes], threshold: float, *, faithful: bool = False,
                threads: int = 1, want_hits: bool = True):
    """query::query_batch (query.rs:66-82).  Accumulates tree.mapped_reads; returns
    (hits, probes, seconds): hits = sorted list of (read index, leaf node) pairs."""
    n = len(reads)
    off = np.zeros(n + 1, dtype=np.uint64)
    if n:
        off[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
    seq = np.frombuffer(b"".join(reads) + b"\0", dtype=np.uint8).copy()
    return query_batch_packed(tree, seq, off, threshold, faithful=faithful, threads=threads, want_hits=want_hits)


def query_batch_packed(tree: OracleTree, seq: np.ndarray, off: np.ndarray, threshold: float, *,
                       faithful: bool = False, threads: int = 1, want_hits: bool = True, hit_cap: int = 0):
    n = len(off) - 1
    left = np.asarray(tree.left, dtype=np.int64)
    right = np.asarray(tree.right, dtype=np.int64)
    filt = np.asarray(tree.filter_of, dtype=np.int64)
    bits = np.ascontiguousarray(tree.bits, dtype=np.uint64)
    ct = _CTree(tree.n_nodes, tree.root, left.ctypes.data, right.ctypes.data, filt.ctypes.data, bits.ctypes.data,
                tree.n_words, tree.nbits, tree.num_hashes, tree.seed1, tree.seed2, tree.kmer_size)
    mapped = np.zeros(max(tree.n_nodes, 1), dtype=np.uint64)
    n_leaves = max(len(tree.leaves_dfs()), 1)
    cap = hit_cap or (max(n, 1) * min(n_leaves, 64) if want_hits else 0)
    pairs = np.zeros((max(cap, 1), 2), dtype=np.uint64)
    n_hits, probes, secs = C.c_uint64(0), C.c_uint64(0), C.c_double(0.0)
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    lib().orc_query_batch(C.byref(ct), _u8(seq), _u64(off), n, threshold, int(faithful), threads, _u64(mapped),
                          _u64(pairs) if want_hits else None, cap, C.byref(n_hits), C.byref(probes), C.byref(secs))
    if want_hits and n_hits.value > cap:
        return query_batch_packed(tree, seq, off, threshold, faithful=faithful, threads=threads, want_hits=True,
                                  hit_cap=int(n_hits.value))
    for v in range(tree.n_nodes):
        tree.mapped_reads[v] += int(mapped[v])
    hits = sorted((int(a), int(b)) for a, b in pairs[: n_hits.value]) if want_hits else None
    return hits, int(probes.value), float(secs.value)


# ------------------------------------------------------------------------------------------------
# synthetic workload
# ------------------------------------------------------------------------------------------------
def synth_genome(seed: int, length: int) -> bytes:
    out = np.zeros(max(length, 1), dtype=np.uint8)
    lib().orc_synth_genome(seed, length, _u8(out))
    return out[:length].tobytes()


def synth_reads(seed: int, first: int, count: int, read_len: int, genomes: np.ndarray, genome_len: int) -> np.ndarray:
    """genomes: [n_genomes, genome_len] uint8.  Returns [count, read_len] uint8."""
    genomes = np.ascontiguousarray(genomes, dtype=np.uint8)
    out = np.zeros((max(count, 1), read_len), dtype=np.uint8)
    lib().orc_synth_reads(seed, first, count, read_len, _u8(genomes), genome_len, genomes.shape[0], _u8(out))
    return out[:count]
