"""ctypes binding of libpfq (include/pfq.h).  No fallback: importing the product without the built HIP
library raises, and every compute call needs a gfx950 device."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PFQ_LIBPFQ: another build of the library, for A/B measurements of two builds on one box)
LIB_PATH = os.environ.get("PFQ_LIBPFQ") or os.path.join(_HERE, "libpfq.so")

# every symbol include/pfq.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "pfq_tree_open", "pfq_tree_open_subtree", "pfq_tree_create", "pfq_tree_insert", "pfq_tree_build_balanced", "pfq_tree_build_balanced_device",
    "pfq_tree_build_balanced_subtree_device", "pfq_trees_allreduce_counts", "pfq_last_allreduce_ranks", "pfq_device_count", "pfq_set_option", "pfq_tree_save", "pfq_tree_info",
    "pfq_tree_prune", "pfq_tree_close", "pfq_query_batch", "pfq_query_batch_device", "pfq_leaf_counts",
    "pfq_save_leaf_counts", "pfq_leaf_counts_export", "pfq_leaf_counts_import", "pfq_leaf_counts_reset",
    "pfq_leaf_counts_export_delta", "pfq_leaf_counts_import_delta",
    "pfq_last_stats", "pfq_set_path", "pfq_profile_begin", "pfq_profile_end", "pfq_debug_kmer_indices", "pfq_debug_node_filter", "pfq_synth_genomes_device",
    "pfq_synth_reads_device", "pfq_host_alloc", "pfq_host_free", "pfq_last_error", "pfq_version",
]


class PfqError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libpfq error {code}: {msg}")
        self.code = code


class Info(C.Structure):
    _fields_ = [("kmer_size", C.c_uint64), ("nbits", C.c_uint64), ("num_hashes", C.c_uint32),
                ("largest_expected_genome", C.c_uint32), ("false_pos_rate", C.c_float),
                ("superset_verified", C.c_uint32), ("seed1", C.c_uint64), ("seed2", C.c_uint64),
                ("n_nodes", C.c_uint64), ("n_leaves", C.c_uint64), ("n_filters", C.c_uint64),
                ("device_bytes", C.c_uint64), ("shard_first_leaf", C.c_uint64), ("tree_leaves", C.c_uint64)]


class Hits(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("offsets", C.POINTER(C.c_uint64)), ("leaves", C.POINTER(C.c_uint32))]


class Stats(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_candidates", C.c_uint64), ("n_hits", C.c_uint64),
                ("n_allhit_reads", C.c_uint64), ("algorithmic_bytes", C.c_uint64), ("path", C.c_uint32),
                ("n_slices", C.c_uint32), ("tile_mode", C.c_uint32), ("n_fallback_pairs", C.c_uint32),
                ("n_chunks", C.c_uint64), ("tile_entries", C.c_uint64), ("tile_passes_launched", C.c_uint32),
                ("tile_passes_needed", C.c_uint32), ("leaf_groups", C.c_uint32), ("coarse_cols", C.c_uint32),
                ("coarse_probes", C.c_uint32), ("pad_", C.c_uint32), ("group_reads", C.c_uint64)]


class Profile(C.Structure):
    _fields_ = [("calls", C.c_uint64), ("classify_ms", C.c_double), ("bucket_ms", C.c_double), ("bin_ms", C.c_double),
                ("test_ms", C.c_double), ("verify_ms", C.c_double), ("finalize_ms", C.c_double)]


WANT_HITS = 1
_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C phagefilter_amd/csrc` "
                          "(or __graft_entry__.build()); phagefilter_amd has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, u8p, u64p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
    L.pfq_last_error.restype = C.c_char_p
    L.pfq_version.restype = C.c_char_p
    L.pfq_tree_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.pfq_tree_open_subtree.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(vp)]
    L.pfq_tree_build_balanced.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_char_p), C.c_uint64, C.c_uint64,
                                          C.c_uint32, C.c_uint64, C.c_uint64, C.c_float, C.c_uint32, C.c_int,
                                          C.POINTER(vp)]
    L.pfq_tree_build_balanced_device.argtypes = [vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_char_p), C.c_uint64,
                                                 C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_float, C.c_uint32,
                                                 C.c_int, C.POINTER(vp)]
    L.pfq_tree_build_balanced_subtree_device.argtypes = [vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_char_p), C.c_uint64,
                                                         C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_float, C.c_uint32,
                                                         C.c_uint64, C.c_uint64, C.c_int, C.POINTER(vp)]
    L.pfq_trees_allreduce_counts.argtypes = [C.POINTER(vp), C.c_uint32]
    L.pfq_last_allreduce_ranks.restype = C.c_uint32
    L.pfq_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.pfq_tree_create.argtypes = [C.c_uint64, C.c_float, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(vp)]
    L.pfq_tree_insert.argtypes = [vp, vp, C.c_uint64, C.c_char_p, C.c_char_p]
    L.pfq_host_alloc.argtypes = [C.c_uint64, C.POINTER(vp)]
    L.pfq_host_free.argtypes = [vp]
    L.pfq_tree_save.argtypes = [vp, C.c_char_p]
    L.pfq_tree_info.argtypes = [vp, C.POINTER(Info)]
    L.pfq_tree_prune.argtypes = [vp, C.c_uint64]
    L.pfq_tree_close.argtypes = [vp]
    L.pfq_tree_close.restype = None
    L.pfq_query_batch.argtypes = [vp, vp, vp, C.c_uint64, C.c_float, C.c_uint32, C.POINTER(Hits)]
    L.pfq_query_batch_device.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64, C.c_float, C.c_uint32, vp, C.POINTER(Hits)]
    L.pfq_leaf_counts.argtypes = [vp, C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(u64p), u64p]
    L.pfq_save_leaf_counts.argtypes = [vp, C.c_char_p]
    L.pfq_leaf_counts_export.argtypes = [vp, vp, vp]
    L.pfq_leaf_counts_import.argtypes = [vp, vp, vp]
    L.pfq_leaf_counts_export_delta.argtypes = [vp, vp, vp]
    L.pfq_leaf_counts_import_delta.argtypes = [vp, vp, vp]
    L.pfq_leaf_counts_reset.argtypes = [vp]
    L.pfq_last_stats.argtypes = [vp, C.POINTER(Stats)]
    L.pfq_set_path.argtypes = [vp, C.c_int]
    L.pfq_profile_begin.argtypes = [vp, C.c_uint32]
    L.pfq_profile_end.argtypes = [vp, C.POINTER(Profile)]
    L.pfq_debug_kmer_indices.argtypes = [vp, vp, C.c_uint64, vp, u64p]
    L.pfq_debug_node_filter.argtypes = [vp, C.c_uint64, vp, C.c_uint64]
    L.pfq_synth_genomes_device.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, vp]
    L.pfq_synth_reads_device.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, vp, C.c_uint64, C.c_uint64,
                                         C.c_uint64, vp]
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != 0:
        raise PfqError(rc, lib().pfq_last_error().decode(errors="replace"))


def source_stamp() -> str:
    """sha256 over the sources of libpfq (csrc/*.hip, *.cpp, *.h + include/pfq.h): stamps measurements that are taken in a
    separate pass (profiles/pmc_traffic.json) so that bench.py can tell whether they still describe the code it runs."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.cpp")) +
                   glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(os.path.dirname(_HERE), "include", "pfq.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
