/*
 * pfq.h — C ABI of libpfq: MI355X-native read classification against a PhageFilter Sequence Bloom Tree.
 *
 * This is the drop-in boundary for ONE path of Dreycey/PhageFilter: `phage_filter query`
 * (src/main.rs:249-376 -> src/query.rs:66-158).  The reference has no FFI of its own; each entry point
 * below replaces the in-process call a Rust `main.rs` makes at that seam and is what its FFI (`extern "C"`
 * block) would bind.  Paths are relative to the reference repository root.
 *
 * Conventions: every function returns PFQ_OK (0) or a negative pfq_status; nothing unwinds across the
 * boundary; `pfq_last_error()` gives the message of the last failure on the calling thread.  The caller owns
 * every input buffer; outputs marked "library-owned" stay valid until the next call on the same tree.
 * One calling thread per pfq_tree (the reference's block loop is serial, main.rs:334-368).
 * There is NO CPU fallback: every entry point that computes needs a gfx950 device and fails with
 * PFQ_ERR_DEVICE otherwise.
 */
#ifndef PFQ_H
#define PFQ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pfq_status {
    PFQ_OK = 0,
    PFQ_ERR_ARG = -1,         /* bad argument */
    PFQ_ERR_IO = -2,          /* file missing / unreadable (reference: panic in bloom_tree.rs:375-379, bloom_filter.rs:155-168) */
    PFQ_ERR_FORMAT = -3,      /* tree.bin / .bf does not parse or is inconsistent */
    PFQ_ERR_UNSUPPORTED = -4, /* valid database outside the device path's limits */
    PFQ_ERR_DEVICE = -5,      /* HIP error / no gfx950 device */
    PFQ_ERR_STATE = -6        /* call order (e.g. query on an empty tree) */
} pfq_status;

typedef struct pfq_tree pfq_tree; /* BloomTree (bloom_tree.rs:29-48) + its filters, resident in HBM */

/* Tree-wide parameters: BloomTree fields (bloom_tree.rs:39-47) + the per-filter constants every node shares
 * (bloom_filter.rs:86-89; identical for all nodes because bloom_tree.rs:279-290 builds every filter alike). */
typedef struct pfq_info {
    uint64_t kmer_size;
    uint64_t nbits;
    uint32_t num_hashes;
    uint32_t largest_expected_genome;
    float false_pos_rate;
    uint32_t superset_verified; /* 1: parent ⊇ child holds on every edge (checked on the device at load) */
    uint64_t seed1, seed2;
    uint64_t n_nodes, n_leaves, n_filters;
    uint64_t device_bytes; /* HBM held by this tree */
    uint64_t shard_first_leaf; /* subtree shards: position of this shard's first leaf in the whole tree's leaf order */
    uint64_t tree_leaves;      /* leaves of the whole tree (== n_leaves unless this is a subtree shard) */
} pfq_info;

/* Per-read results of one pfq_query_batch call: CSR read -> leaf indices (indices into pfq_leaf_counts'
 * left-to-right leaf order).  Replaces ResultMap (result_map.rs:9-46) at the seam of query.rs:146-154. */
typedef struct pfq_hits {
    uint64_t n_reads;
    const uint64_t *offsets; /* [n_reads + 1], library-owned */
    const uint32_t *leaves;  /* [offsets[n_reads]], ascending within a read, library-owned */
} pfq_hits;

#define PFQ_WANT_HITS 1u /* fill pfq_hits (needed for POS/NEG filtering, main.rs:345-361) */

/* ---- database ---- */

/* BloomTree::load (bloom_tree.rs:364-386) + every BloomFilter::load_from_file the LRU cache would do lazily
 * (cache.rs:56-77, bloom_filter.rs:153-174): parses <db_dir>/tree.bin and each node's .bf (filters are keyed
 * by their relative path exactly like the cache), uploads them, verifies parent ⊇ child per edge and builds
 * the device layout.  `device` = HIP device ordinal. */
int pfq_tree_open(const char *db_dir, int device, pfq_tree **out);

/* Subtree shard of a database, for trees larger than one GPU's HBM (BASELINE config 5): the shards
 * are the nodes of the depth-`depth` frontier in left-to-right order (nodes at that depth, plus leaves above it).
 * Shard `index` keeps that node, everything below it and the chain of its ancestors, each reduced to the child on
 * the path; only those .bf files are read.  Every rank classifies ALL reads against its shard; the shards' leaf
 * ranges are disjoint and contiguous in the whole tree's leaf order (pfq_info.shard_first_leaf), so the whole
 * job's counts are the concatenation of the shards' counts.  Ancestors that are not verified supersets become
 * guard columns, so results equal the reference's whole-tree traversal. */
int pfq_tree_open_subtree(const char *db_dir, int device, uint64_t depth, uint64_t index, pfq_tree **out);

/* The reference's `build` / `add` (main.rs:148-247) on the device.
 * pfq_tree_create = BloomTree::new (bloom_tree.rs:100-118): an empty tree; filter geometry from
 *   (false_pos_rate, largest_expected_genome) exactly like with_rate (bloom_filter.rs:229-240,:342-357, f32 arithmetic);
 *   the two hash seeds are explicit (the reference draws them at random, hasher.rs:24-30).  expected_genomes sizes the
 *   filter storage up front (2n-1 rows); 0 = grow on demand.
 * pfq_tree_insert = BloomTree::insert (bloom_tree.rs:128-143): a leaf filter holding every canonical k-mer of `seq`
 *   (:154-168), then the greedy descent (:187-214): every two-child node on the way absorbs the new filter and the walk
 *   continues into the child at smaller Hamming distance (right only if strictly smaller, :201); the leaf reached is
 *   replaced by a new internal node (left = old leaf, right = new leaf, filter = union, :226-245).  internal_name
 *   names that node (tax_id, "<name>.bf"); NULL = "Internal_Node_<n>" with a running n unique in the tree (the
 *   reference draws a random u16, :231-233).  Works on trees from pfq_tree_open as well (`add`). */
int pfq_tree_create(uint64_t kmer_size, float false_pos_rate, uint32_t largest_expected_genome, uint64_t seed1,
                    uint64_t seed2, uint64_t expected_genomes, int device, pfq_tree **out);
int pfq_tree_insert(pfq_tree *tree, const uint8_t *seq, uint64_t len, const char *tax_id, const char *internal_name);

/* Synthetic balanced SBT built on the device: leaf i = all canonical k-mers of genome i
 * (what bloom_tree.rs:154-168 inserts), internal = OR of children (bloom_tree.rs:238-239), complete-as-possible
 * balanced shape, leaf tax_id = tax_ids[i], internal tax_id = "Internal_Node_<n>".  genomes/offsets are HOST
 * buffers: genome i = genomes[offsets[i] .. offsets[i+1]).  NOT the reference's greedy `build`. */
int pfq_tree_build_balanced(const uint8_t *genomes, const uint64_t *offsets, uint64_t n_genomes,
                            const char *const *tax_ids, uint64_t kmer_size, uint64_t nbits, uint32_t num_hashes,
                            uint64_t seed1, uint64_t seed2, float false_pos_rate, uint32_t largest_expected_genome,
                            int device, pfq_tree **out);
/* Same, with genomes already in device memory (n_genomes x genome_len bytes, contiguous). */
int pfq_tree_build_balanced_device(const uint8_t *d_genomes, uint64_t genome_len, uint64_t n_genomes,
                                   const char *const *tax_ids, uint64_t kmer_size, uint64_t nbits,
                                   uint32_t num_hashes, uint64_t seed1, uint64_t seed2, float false_pos_rate,
                                   uint32_t largest_expected_genome, int device, pfq_tree **out);

/* One subtree shard of that synthetic tree without ever holding the whole tree (BASELINE config 5: 16 384 leaves =
 * 294 GB of filters, one 2048-leaf shard per GPU): the same topology and names as pfq_tree_build_balanced_device over all
 * n_genomes, reduced like pfq_tree_open_subtree(depth, index); the shard's subtree is built from its own genomes and every
 * ancestor on the chain holds the union of ALL genomes below it in the whole tree.  d_genomes: all n_genomes genomes. */
int pfq_tree_build_balanced_subtree_device(const uint8_t *d_genomes, uint64_t genome_len, uint64_t n_genomes,
                                           const char *const *tax_ids, uint64_t kmer_size, uint64_t nbits,
                                           uint32_t num_hashes, uint64_t seed1, uint64_t seed2, float false_pos_rate,
                                           uint32_t largest_expected_genome, uint64_t depth, uint64_t index, int device,
                                           pfq_tree **out);

/* BloomTree::save (bloom_tree.rs:339-355) + the .bf files BloomFilter::save_to_file writes
 * (bloom_filter.rs:176-205), so the reference binary can open a tree built here. */
int pfq_tree_save(const pfq_tree *tree, const char *db_dir);

int pfq_tree_info(const pfq_tree *tree, pfq_info *out);

/* BloomTree::prune_tree (bloom_tree.rs:302-330): nodes at depth >= search_depth become leaves. */
int pfq_tree_prune(pfq_tree *tree, uint64_t search_depth);

void pfq_tree_close(pfq_tree *tree);

/* ---- query ---- */

/* query::query_batch (query.rs:66-82) for one block of reads given as raw bytes: read i =
 * seq[offsets[i] .. offsets[i+1]) (HOST buffers).  k-mer extraction (file_parser.rs:135-148) happens on the
 * device.  Leaf counts accumulate across calls like BloomNode::mapped_reads (query.rs:143).  `hits` may be NULL
 * unless PFQ_WANT_HITS is set.  The input buffers may be reused as soon as the call returns.  Without
 * PFQ_WANT_HITS the call returns when the block has been copied and its kernels are queued (the copy of the next
 * block overlaps them); the calls that read results (pfq_leaf_counts, pfq_save_leaf_counts, pfq_last_stats,
 * pfq_tree_close) wait for the device. */
int pfq_query_batch(pfq_tree *tree, const uint8_t *seq, const uint64_t *offsets, uint64_t n_reads, float threshold,
                    uint32_t flags, pfq_hits *hits);

/* Same with the block already resident in HBM (device pointers) on HIP stream `stream` (hipStream_t, may be
 * NULL for the default stream).  total_bytes = offsets[n_reads], the size of the sequence buffer (0 if unknown:
 * the library then skips optimisations that need it).  Asynchronous unless PFQ_WANT_HITS is set; counts are final
 * after the stream is synchronised.  The tree's scratch buffers are reused call after call: calls on one stream are
 * ordered by it; a call on another stream than the previous one (pfq_query_batch uses the default stream) first waits
 * for that one.  This is the entry the benchmark times. */
int pfq_query_batch_device(pfq_tree *tree, const uint8_t *d_seq, const uint64_t *d_offsets, uint64_t n_reads,
                           uint64_t total_bytes, float threshold, uint32_t flags, void *stream, pfq_hits *hits);

/* get_leaf_counts (query.rs:197-218): leaves left-to-right, zeros included.  Library-owned arrays. */
int pfq_leaf_counts(pfq_tree *tree, const char *const **tax_ids, const uint64_t **counts, uint64_t *n_leaves);
/* save_leaf_counts (query.rs:173-183): "<tax_id>,<count>\n" for count > 0, no header. */
int pfq_save_leaf_counts(pfq_tree *tree, const char *csv_path);

/* Multi-GPU reduction hooks (one process per GPU; the host framework all-reduces with RCCL):
 * copy the u64[n_leaves] device counters out to / in from a device buffer on `stream`. */
int pfq_leaf_counts_export(pfq_tree *tree, uint64_t *d_dst, void *stream);
int pfq_leaf_counts_import(pfq_tree *tree, const uint64_t *d_src, void *stream);
int pfq_leaf_counts_reset(pfq_tree *tree);
/* The same hooks for what THIS replica counted: export_delta writes counters - base, where the base is what the counters held
 * when the tree was opened (BloomNode::mapped_reads stored in tree.bin — non-zero in a database that was saved after a query),
 * last reset, imported or reduced; import_delta(sum of the ranks' deltas) sets counters = base + sum and makes that the new
 * base.  Reducing deltas keeps stored counts from being added once per rank: every rank ends with stored + new, like one
 * device and like the reference (query.rs:143 accumulates on the loaded value).  pfq_leaf_counts_import also sets the base. */
int pfq_leaf_counts_export_delta(pfq_tree *tree, uint64_t *d_dst, void *stream);
int pfq_leaf_counts_import_delta(pfq_tree *tree, const uint64_t *d_src, void *stream);

/* Number of HIP devices this process can use (`--devices all` of the CLI). */
int pfq_device_count(int *n);

/* Several GPUs behind one process (the block loop of main.rs:334-368 dealt over devices): `trees` are replicas of one
 * database (pfq_tree_open of the same directory, same pruning) on any devices, each fed its own share of the reads by its
 * own host thread.  This sums their per-leaf counters so that afterwards EVERY replica holds the job's totals
 * (mapped_reads of query.rs:143 as if one tree had seen all reads): what each replica counted since it was opened (or last
 * reduced) is added — replicas that share a device on that device, then ONE ncclAllReduce(sum, uint64, n_leaves) over RCCL /
 * xGMI across the distinct devices (8 KiB at 1024 leaves) — onto the counts the database was opened with, which therefore
 * count once; calling it again without new queries changes nothing.  The communicator of a device set is created on first
 * use and kept until the last tree of the process is closed.  Waits for the replicas' queued work.  librccl is loaded when this first meets replicas on more than one device
 * (PFQ_RCCL_ALWAYS=1: a one-rank communicator even then, for exercising the path on a one-GPU box). */
int pfq_trees_allreduce_counts(pfq_tree *const *trees, uint32_t n_trees);
/* Number of RCCL ranks the last pfq_trees_allreduce_counts on this thread used (0: no communicator was needed). */
uint32_t pfq_last_allreduce_ranks(void);

/* ---- measurement / test hooks ---- */

/* Tuning / test knobs, e.g. ("PFQ_TILE", "0").  The PFQ_* environment variables of the same names are
 * read once, when a tree is created or opened; this changes one knob of one tree afterwards.  value NULL or "": back
 * to the built-in choice.  Results never depend on a knob. */
int pfq_set_option(pfq_tree *tree, const char *name, const char *value);


/* Per-call statistics of the last pfq_query_batch[_device] (valid after the stream is synchronised). */
typedef struct pfq_stats {
    uint64_t n_reads, n_candidates, n_hits, n_allhit_reads;
    uint64_t algorithmic_bytes; /* sum_r L(r) + |hits(r)| * need(r) * num_hashes * 32 */
    uint32_t path;              /* 0 = direct kernel, 1 = bucketed (screen, pairs sorted by leaf, certificates out of LDS tiles / L2 slices) */
    uint32_t n_slices;
    uint32_t tile_mode;         /* 1: certificates tested out of LDS tiles (k_tile_*), k_verify_rec only as fallback
                                 * (thresholds < 1: entries name k-mers, the passes leave per-chunk miss bytes);
                                 * 2: block mode — pairs are (read, block of 8 leaves, candidate mask), one entry tests a probe for
                                 * all candidates of the block (chosen when reads pass several related leaves; any threshold in (0, 1]) */
    uint32_t n_fallback_pairs;  /* pairs the LDS-tile pass could not bin (certified by the fallback kernel) */
    uint64_t n_chunks, tile_entries;
    uint32_t tile_passes_launched, tile_passes_needed;  /* LDS-tile stage: passes over the reused probe buckets */
    /* two-level frontier (trees of more than 2048 leaves): the reads are screened against a coarse level of internal
     * nodes first and every group of leaf columns only sees the reads with a live ancestor there (query.rs:119-141) */
    uint32_t leaf_groups;       /* groups of leaf columns of the sliced matrix (1 for trees of up to 2048 columns) */
    uint32_t coarse_cols;       /* columns (internal nodes) of the coarse level this call used; 0: flat frontier */
    uint32_t coarse_probes;     /* probes per k-mer its screens looked at */
    uint32_t pad_;
    uint64_t group_reads;       /* (read, leaf group) combinations the coarse level let through to the leaf level */
} pfq_stats;
int pfq_last_stats(pfq_tree *tree, pfq_stats *out);
/* Force a query path: -1 auto, 0 direct, 1 bucketed. */
int pfq_set_path(pfq_tree *tree, int path);

/* Per-kernel device time of the query path, measured with HIP events recorded on the stream the kernels are
 * launched on.  begin: record around the kernels of the next (up to max_calls) query calls; end: synchronise and sum. */
typedef struct pfq_profile {
    uint64_t calls;
    double classify_ms; /* k_classify (pre-screen, frontier, probe records or inline certificates) */
    double bucket_ms;   /* bucket scan + scatter */
    double bin_ms;      /* k_tile_plan + k_tile_bin (probes binned by leaf chunk and filter tile) */
    double test_ms;     /* k_tile_test (tiles tested out of LDS) */
    double verify_ms;   /* k_verify_rec / k_verify (L2-sliced certificates; only the fallback pairs in tile mode) */
    double finalize_ms; /* k_finalize */
} pfq_profile;
int pfq_profile_begin(pfq_tree *tree, uint32_t max_calls);
int pfq_profile_end(pfq_tree *tree, pfq_profile *out);

/* K1 parity hook: the num_hashes bit indices of every canonical k-mer of `seq` (HOST buffers), exactly what
 * BloomFilter::contains probes (bloom_filter.rs:312-332 via hash_iter.rs:13-45): out_idx[(kmer * num_hashes) + i]. */
int pfq_debug_kmer_indices(pfq_tree *tree, const uint8_t *seq, uint64_t len, uint64_t *out_idx, uint64_t *n_kmers);
/* Copy one node's filter words (Lsb0 u64, bloom_filter.rs:86) to the host; node = pre-order index. */
int pfq_debug_node_filter(pfq_tree *tree, uint64_t node, uint64_t *out_words, uint64_t n_words);

/* Synthetic workload generators of SURVEY §8d on the device (counter-based splitmix64); bench/test data only. */
int pfq_synth_genomes_device(uint8_t *d_out, uint64_t n_genomes, uint64_t genome_len, uint64_t seed_base,
                             void *stream);
int pfq_synth_reads_device(uint8_t *d_out, uint64_t first_read, uint64_t n_reads, uint64_t read_len,
                           const uint8_t *d_genomes, uint64_t genome_len, uint64_t n_genomes, uint64_t seed,
                           void *stream);

/* Page-locked host memory for the buffers handed to pfq_query_batch: the host-to-device copy then runs at PCIe rate
 * instead of going through the runtime's pageable staging path.  (The reference keeps reads in ordinary Vec<u8>s,
 * file_parser.rs:150-172; this is the transfer-side counterpart of that buffer.)  A tree must be open on the device. */
int pfq_host_alloc(uint64_t bytes, void **out);
int pfq_host_free(void *p);

const char *pfq_last_error(void);
const char *pfq_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PFQ_H */
