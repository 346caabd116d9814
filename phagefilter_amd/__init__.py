"""phagefilter_amd — MI355X-native read classification for PhageFilter Sequence Bloom Trees.

Only the `phage_filter query` path; hand-written HIP kernels behind the C ABI of include/pfq.h.
"""
from .query import BloomTree, ResultMap, get_leaf_counts, pack_reads, query_batch, save_leaf_counts  # noqa: F401
from ._ffi import PfqError, lib  # noqa: F401
