// pfq_kernels.h — argument blocks and launch wrappers shared by the kernels (pfq_kernels.hip) and the host
// side of libpfq (pfq_host.cpp).  Device layout, see DESIGN.md §3:
//
//   bits   u64[n_filters][n_words]   node-major filters in the reference's own bit order (BitVec<usize,Lsb0>,
//                                    bloom_filter.rs:86): bit idx = word idx>>6, mask 1<<(idx&63).
//   S      u32[groups][n_words*64 + 1][rw]   "sliced" matrix (last row of a group: all ones, the target of predicated-off
//                                    gathers): row = bit index, column = leaf (left-to-right DFS order) followed by guard
//                                    columns (ancestors whose ⊇ check failed).  One 128-B line answers the same probe for
//                                    1024 leaves.  Trees of more than 2048 columns are cut into column groups of 2048
//                                    (rw = 64 each); the frontier kernels run once per group.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pfq_device.h"

namespace pfq {

// Per-wave reservations in the deferred-pair buffer (slots) and in the miss-word buffer (u64 words): every wave of every
// classify launch may leave one of each partly used, which the host adds to the buffers' sizes.
// Experiment switches of k_tile_bin that give wrong results are compiled in with -DPFQ_EXPERIMENTS only: no environment
// variable or option can change the results of the library as shipped.
#ifdef PFQ_EXPERIMENTS
#define PFQ_DEBUG_BITS(a) ((a).debug)
#else
#define PFQ_DEBUG_BITS(a) 0u
#endif
constexpr uint32_t PAIR_RESERVE = 32, MISS_RESERVE = 256;
enum StatSlot { ST_CANDIDATES = 0, ST_HITS = 1, ST_ALLHIT = 2, ST_ALG_BYTES = 3, ST_DEFERRED = 4, ST_LISTED = 5 /* (read, leaf group) entries of a two-level frontier */, ST_N = 8 };

struct QueryArgs {
    HashParams hp;
    const uint8_t *seq;
    const uint64_t *off;
    uint64_t n_reads;
    float threshold;
    // sliced matrix
    const uint32_t *S;           // the column group this launch screens
    const uint32_t *S_all;       // group 0 (certificates of arbitrary columns: guards may live in another group)
    uint64_t group_stride;       // dwords between the matrices of consecutive groups
    uint32_t col0;               // first column of this launch's group (columns are global everywhere else)
    uint32_t group_log2;         // log2 of the columns per group of the sliced matrix (11; 10 for trees with a coarse level)
    // two-level frontier (trees of several column groups): the launch of a leaf group only sees the reads the coarse launch
    // (k_coarse) listed for it — reads with a live ancestor of one of the group's leaves
    const uint32_t *read_list;   // reads of this launch (entries 0xffffffff: unused slots of a reservation), nullptr: all reads
    const unsigned int *n_list;  // slots of read_list in use
    // ONE launch serves every leaf group of a two-level frontier: blockIdx.y = the group; S, col0, n_leaves, read_list, n_list
    // (and long_list / n_long) of the group follow from these (the host passes group 0's pointers)
    uint32_t grid_groups;        // leaf groups of the launch (0: the launch is for the one group the fields above describe)
    uint32_t total_leaves;       // leaves of the tree
    uint32_t list_cap;           // slots per group in read_list and long_list
    uint32_t first_group;        // 1: this launch accounts for the per-read statistics (read bytes, all-hit reads)
    uint32_t ones_row;           // index of the all-ones row stored behind the last bit row of S
    uint32_t rw, rw_log2;        // row words (power of two <= 64)
    uint32_t n_leaves, n_cols;   // leaf columns of this group; leaf+guard columns of the tree
    const uint32_t *guard_off;   // [total leaves + 1] CSR into guard_col (may be all zeros), indexed by global leaf column
    const uint32_t *guard_col;   // global columns
    // results
    unsigned long long *counts;  // [n_leaves]  mapped_reads, accumulating (query.rs:143)
    uint2 *hit_pairs;            // (read, leaf) or nullptr
    uint64_t hit_cap;
    unsigned long long *hit_cursor;
    uint8_t *allhit_flag;        // per read: passes everything (need == 0), or nullptr
    unsigned long long *stats;   // [ST_N]
    // deferral to the bucketed verify pass (DEFER kernels only)
    uint2 *pairs;
    uint64_t pair_cap;
    unsigned long long *pair_cursor;
    uint32_t *bucket_cnt;        // [n_leaves << sub_log2]: bucket = (leaf << sub_log2) | (read & (subs-1))
    uint32_t sub_log2;           // sub-buckets per leaf (so that no counter is a hot spot when leaves are few)
    uint4 *recs;                 // probe records, indexed by (read byte offset + k-mer position), or nullptr
    uint64_t rec_cap;            // entries in recs (reads whose records would not fit are certified inline)
    uint32_t *long_list;         // thresholds < 1: reads of >= 256 k-mers, classified by a second launch (wider counters)
    unsigned int *n_long;
    uint32_t batch_tails;        // theta == 1 with records: last windows of <= batch_tails k-mers (0: none, 16 or 32) are left to k_tail_records
    uint32_t block_pairs;        // 1 (DEFER, theta == 1, no guard columns): defer (read, block of 8 leaves | candidate mask << 24)
    uint32_t screen_recs;        // thresholds < 1: the dense counting screen writes the probe records of the k-mers it hashes
    uint32_t screen_only;        // (launches without deferral) count the frontier's candidate leaves, certify nothing, count no read
    // thresholds < 1: every deferred pair owns ceil(n/64) u64 words of k-mer miss bits.  k_classify only accounts for
    // them (per bucket, and against the buffer's capacity through per-wave reservations); k_bucket_scatter places them.
    uint32_t *bucket_words;            // [n_leaves << sub_log2] miss words per bucket, or nullptr (threshold 1)
    unsigned long long *miss_cursor;   // words reserved so far
    uint64_t miss_cap;                 // words in the buffer
};

// Two-level frontier (k_coarse).  A tree of more than one column group gets a COARSE sliced matrix over an antichain of
// internal nodes that covers every leaf (each node as close to the leaves as <= 1024 / 2048 columns allow).  A read that
// does not pass a node reaches no leaf below it (query.rs:119-141 descends with the survivors only), whatever the
// threshold and whether or not the node's filter is a superset of its children's; the screens of the leaf groups then run
// on the reads listed for them instead of on every read.  Filters of internal nodes are fuller than those of leaves, so
// the coarse screens look at n_probes probes per k-mer instead of one (two at threshold 1).
constexpr uint32_t MAX_LEAF_GROUPS = 64;   // groups a coarse launch can list reads for (wider trees: flat frontier)
constexpr uint32_t COARSE_MAX_PROBES = 6;  // probes per k-mer of the coarse AND-screen (threshold 1); the counting screen takes <= 4
struct CoarseArgs {
    const uint32_t *cgrp;        // [coarse columns] first leaf group | last leaf group << 16 of the leaves below the column's node
    uint32_t n_groups;           // leaf groups (<= MAX_LEAF_GROUPS)
    uint32_t *lists;             // [n_groups][list_cap] reads per leaf group, appended through per-wave reservations of 32
    unsigned int *cursors;       // [n_groups] slots handed out
    uint32_t list_cap;           // >= n_reads + 32 per wave of the launch: no list can overflow
    uint32_t n_probes;           // probes per k-mer the coarse screen looks at
    uint32_t scr_extra;          // counting screen: k-mers looked at beyond maxmiss + 1
    uint32_t total_leaves;       // reads that pass every node count at every leaf of the tree
};
void launch_coarse(const QueryArgs &a, const CoarseArgs &ca, bool counts_mode, int blocks, hipStream_t st);
// out[i] = set bits of filter row rows[i]
void launch_row_popcount(const uint64_t *bits, uint64_t n_words, const uint32_t *d_rows, uint32_t n_rows, unsigned long long *d_out, hipStream_t st);

// Guard pairs (k_expand_guards): second region of the pair buffer, slots slot0 .. slot0 + cap of the whole buffer.
struct GuardArgs {
    uint2 *pairs;                // = pair buffer + slot0
    uint64_t cap;                // slots in the region (whole reservations of 32)
    unsigned long long *cursor;  // slots reserved so far
    uint32_t slot0;              // index of the region's first slot in the whole pair buffer
    uint32_t *owner;             // [whole pair buffer] slot of the leaf pair a slot belongs to (a leaf pair owns itself)
    uint32_t *gfail;             // [whole pair buffer] indexed by the leaf pair's slot: a guard of it did not pass
};
void launch_expand_guards(const QueryArgs &a, const GuardArgs &ga, int blocks, hipStream_t st);

struct ChunkDesc;
struct VerifyArgs {
    HashParams hp;
    const uint8_t *seq;
    const uint64_t *off;
    const uint64_t *bits;        // node-major filters
    const uint32_t *col_row;     // column -> filter row
    uint64_t n_words;
    const uint2 *sorted;         // (read, leaf) sorted by leaf
    const uint4 *meta;           // per sorted pair: (read byte offset lo, hi, read length, filter row)
    const uint32_t *n_pairs_ptr; // &bucket_off[n_leaves]
    uint32_t *fail;              // [pair_cap]
    uint32_t n_slices, slice_bits;
    unsigned int *queue;         // work cursors: [n_slices] (re-hash kernel) / [8 * n_sub] (record kernel)
    uint32_t n_sub;              // sub-queues per XCD (record kernel)
    uint32_t only_flagged;       // fallback after the LDS-tile pass, pairs whose fail word has bit 1 set: 1 = from the compact
                                 // list when they are few, 2 = by walking all sorted pairs when they are many
    uint32_t flag_cap;           // entries of flag_list
    // the tile passes are launched without waiting for the plan: chunks of passes >= launched_passes (their number
    // was guessed from the previous call) are certified here as well (mode 2)
    const uint32_t *pair_chunk;
    const struct ChunkDesc *chunks;
    const unsigned long long *entry_cursor;
    uint64_t entry_cap;
    uint32_t launched_passes;
    const unsigned int *n_flagged;  // number of such pairs (kernel returns at once when 0)
    const uint32_t *flag_list;      // their sorted-pair indices
    const uint4 *recs;           // probe records written by k_classify<DEFER> (nullptr: re-hash per slice)
    unsigned long long *miss_words;  // thresholds < 1: bit q of the pair's words = a probed bit of its k-mer q is 0
    const uint32_t *miss_pos;        // [sorted pair] first word of the pair (nullptr at threshold 1: any miss fails the pair)
    uint32_t chunk;
};

struct FinalizeArgs {
    HashParams hp;
    const uint64_t *off;
    const uint2 *sorted;
    const uint32_t *bucket_off;  // [(n_leaves << sub_log2) + 1]
    uint32_t sub_log2;
    const uint32_t *fail;
    const unsigned long long *miss_words;  // thresholds < 1 (see VerifyArgs); nullptr at threshold 1
    const uint32_t *miss_pos;
    float threshold;
    uint32_t c0, c1;             // buckets (columns) of this launch: the guard columns first, then the leaves
    uint32_t guards;             // 1: guard columns — a pair that does not pass marks its leaf pair in gfail; no counts, no hits
    const uint32_t *owner_sorted;  // trees with guard columns: per sorted pair, the slot of its leaf pair (else nullptr)
    uint32_t *gfail;             // per pair slot: a guard of this leaf pair did not pass
    unsigned long long *counts;
    uint2 *hit_pairs;
    uint64_t hit_cap;
    unsigned long long *hit_cursor;
    unsigned long long *stats;
    unsigned long long *n_dirty;  // thresholds < 1: += pairs with at least one k-mer missing (sizes the next call's certificate stage)
    const uint8_t *failb;         // block mode (else nullptr): sorted pairs are (read, block | mask << 24), [pair][8] failure bytes
    // thresholds < 1 after LDS-tile passes with k-mer entries: a pair that was binned in a launched pass and never flagged
    // has its miss bits in its chunk's bitmap (kmiss); every other pair in its own miss words (written by k_verify_rec)
    const uint8_t *kmiss;         // nullptr: miss words only
    const uint8_t *kall;          // block mode with k-mer entries (see TileArgs)
    const uint32_t *pair_kpos;
    const uint32_t *pair_chunk;
    const ChunkDesc *chunks;
    uint32_t launched_passes;
};

// ---- LDS-tile certificates (bucketed path, records available) ------------------------------------------------------
// Every probe of every sorted pair is binned by (chunk of <= 4096 pairs of one leaf, 2^20-bit tile of the filter);
// a block then loads one tile of one leaf into LDS and tests all its probes there.
constexpr uint32_t TILE_LOG2 = 20;                 // bits per tile = 128 KiB of filter (one 1024-thread test block per CU;
                                                   // 64 KiB tiles, two blocks per CU: bin 11.4 / test 5.9 ms vs 10.8 / 5.9)
constexpr uint32_t CHUNK_PAIRS_LOG2 = 10;          // pairs per chunk: local pair id and tile offset share one u32 entry; a chunk
                                                   // is the work item of k_tile_bin (one block bins it alone)
constexpr uint32_t MAX_TILES = 1024;               // tiles per filter the passes take (k_tile_bin keeps a bin per tile in LDS)
// Thresholds < 1: the passes must tell WHICH k-mers are not contained, so an entry names a k-mer instead of a pair:
// [round tag : 2][k-mer of the round : 11][offset in tile : 19] — k_tile_bin bins a chunk in rounds of <= 2^11 flattened
// k-mers, the four entries of a 16-byte vector (runs are 16-byte aligned) carry the round's number in their tag bits, and
// round_k0[chunk][round] turns (round, k-mer of the round) into the k-mer's position in the chunk's miss bitmap.
constexpr uint32_t TILE_LOG2_COUNTS = 19;          // 64 KiB tiles (twice the tiles, 128-entry deeper bins than needed)
constexpr uint32_t ROUND_KMERS_LOG2 = 11;          // flattened k-mers per round of k_tile_bin (= 2 windows x 16 waves)
constexpr uint32_t MAX_ROUNDS = 256;               // rounds per chunk the tags can name; later pairs take the fallback
// Reads that pass several related leaves (a phage database is full of strains): BLOCK MODE.  A pair is (read, block of 8
// consecutive leaf columns, mask of the candidate leaves in it); the block's "filter" is a byte per Bloom bit index — bit j =
// that bit of leaf 8b + j — so ONE entry tests a probe for all candidate leaves of the block: a read that passes 8 strains
// costs the probes of one pair instead of eight.  The pairs are bucketed by (block, candidate mask): all pairs of a chunk
// share their mask, which travels in the chunk's descriptor, and the entries stay [pair of the chunk:10][byte offset in a
// 128 KiB tile:17] (threshold 1) or [round tag:2][k-mer:11][byte offset:17] (thresholds below 1: the miss array then holds
// 8 bytes per k-mer, one per leaf of the block).  The columns of the passes are the blocks, whatever the buckets.
constexpr uint32_t TILE_LOG2_BLOCK = 17;           // 2^17 bit indices x 8 leaves = 128 KiB
constexpr uint32_t BLOCK_LEAVES_LOG2 = 3;
struct ChunkDesc {
    uint32_t row;      // filter row of the leaf (block mode: the block)
    uint32_t first;    // first sorted pair
    uint32_t n;        // pairs
    uint32_t cap;      // entries per (chunk, tile) bucket; 0: no room, the chunk's pairs take the fallback
    uint64_t base;     // first entry of tile 0's bucket
    uint32_t leaf;
    uint32_t pass;     // the probe buckets are reused: chunks are binned and tested pass after pass
    uint32_t kbase;    // thresholds < 1: where the chunk's k-mer miss array starts, in 16-byte units (a byte per k-mer of the chunk
                       // in the order of its pairs; block mode: 8 bytes per k-mer, one per leaf of the block)
    uint32_t kwords;   //                 its size in 16-byte units
    uint32_t mask;     // block mode with k-mer entries: the candidate mask all pairs of the chunk share
    uint32_t pad_;
};
struct TileArgs {
    uint32_t pass;               // k_tile_bin / k_tile_test: only the chunks of this pass
    HashParams hp;
    const uint64_t *bits;
    uint64_t n_words;
    const uint4 *recs;
    const uint4 *meta;           // per sorted pair (read offset lo, hi, length, row)
    const uint32_t *col_row;     // column -> filter row
    const uint32_t *bucket_off;  // [(n_leaves << sub_log2) + 1]
    uint32_t sub_log2, n_leaves, n_tiles;   // n_leaves: buckets = leaf + guard columns
    uint32_t bin_shape;          // 0 auto, 1 force the 8 x 128 build of k_tile_bin, 2 force 16 x 256
    uint32_t debug;              // builds with -DPFQ_EXPERIMENTS only (timing experiments, results wrong): 1 no bucket stores, 2 no LDS binning,
                                 // 8 / 16 block mode without the flags of full bins / full buckets.  The default build ignores the field.
    ChunkDesc *chunks;           // [max_chunks]
    uint32_t max_chunks;
    uint32_t *leaf_chunk0;       // [n_leaves + 1] first chunk of each leaf (chunks of a leaf are contiguous)
    uint32_t *pair_chunk;        // [pair_cap] chunk of each sorted pair
    unsigned int *n_chunks;      // counter
    unsigned long long *entry_cursor;  // virtual: pass = cursor / entry_cap, position in the pass = cursor % entry_cap
    uint64_t entry_cap;
    uint32_t *entries;
    unsigned int *gfill;         // [max_chunks * n_tiles] entries (and padding) in every bucket, written by k_tile_bin
    unsigned int *bin_queue;     // k_tile_bin of this pass: next chunk to take
    uint32_t *fail;              // bit 0: a probed bit was 0; bit 1: pair must be verified by the fallback kernel
    unsigned int *n_flagged;     // pairs with bit 1
    uint32_t *flag_list;         // [flag_cap] their sorted-pair indices
    uint32_t flag_cap;
    const uint32_t *n_pairs_ptr;
    // block mode (see TILE_LOG2_BLOCK)
    uint32_t blocks;             // 1: pairs are (read, block | mask << 24), meta.w likewise; bits = block tables, n_words = bytes / 8 of one
                                 // (with `counts`: buckets = (block << 8) | mask, n_leaves = blocks: a column of the passes is a block)
    uint32_t chunk_log2;         // pairs per chunk (CHUNK_PAIRS_LOG2)
    uint8_t *failb;              // block mode: [sorted pair][8] a probed bit of that candidate leaf was 0
    // thresholds < 1 (entries name k-mers, see TILE_LOG2_COUNTS)
    float threshold;             // (which prefix of a read's k-mers is binned depends on it)
    uint32_t counts;             // 1: k-mer entries, 64 KiB tiles, miss bits; 0: pair entries, 128 KiB tiles, fail words
    uint8_t *kmiss;              // k-mer miss bytes of all chunks (a byte, not a bit: set with plain stores from any XCD)
    uint64_t kmiss_cap;          // its bytes, < 2^32 (chunks that find no room take the fallback)
    unsigned long long *kmiss_used;  // bytes handed out by k_tile_assign (cleared before the passes)
    uint8_t *kall;               // block mode: [k-mer] the k-mer is in no candidate leaf of its pair's block (one byte for all eight)
    uint32_t *round_k0;          // [max_chunks][MAX_ROUNDS] position (in the chunk) of the first k-mer of every round
    uint32_t *n_rounds;          // [max_chunks]
    uint32_t *pair_kpos;         // [sorted pair] position of the pair's first k-mer in its chunk
};
void launch_tile_plan(const TileArgs &a, hipStream_t st);
void launch_tile_bin(const TileArgs &a, int blocks, hipStream_t st);
void launch_tile_test(const TileArgs &a, int blocks, hipStream_t st);

// launches (all asynchronous on `st`)
void launch_classify(const QueryArgs &a, bool defer, bool counts_mode, int blocks, hipStream_t st);
void launch_tail_records(const QueryArgs &a, int blocks, hipStream_t st);  // after launch_classify when a.batch_tails
void launch_bucket_scan(const uint32_t *bucket_cnt, uint32_t *bucket_off, uint32_t *bucket_cur, uint32_t n, hipStream_t st);
// words_off / words_cur / miss_pos: thresholds < 1 (miss words of a bucket start at words_off[bucket]); else nullptr
// key_mode 0: the bucket of a pair is its column; 1 (block mode): the block = low 24 bits of its second word, which goes to
// meta.w whole; 2 (block mode with k-mer entries): (block << 8) | candidate mask
void launch_bucket_scatter(const uint2 *pairs, const unsigned long long *n_pairs_ptr, uint64_t pair_cap,
                           const uint32_t *bucket_off, uint32_t *bucket_cur, uint32_t sub_log2, uint2 *sorted, uint4 *meta,
                           const uint64_t *read_off, const uint32_t *col_row, const uint32_t *words_off, uint32_t *words_cur,
                           uint32_t *miss_pos, uint32_t kmer_size, const uint32_t *owner, uint32_t *owner_sorted, uint32_t key_mode,
                           int blocks, hipStream_t st);
// block tables: T[b][i] = byte whose bit j is bit i of the filter of leaf column 8b + j (zero for columns past the last leaf)
void launch_block_tables(const uint64_t *bits, uint64_t n_words, const uint32_t *d_col_row, uint32_t n_leaves, uint8_t *T, hipStream_t st);
// block mode: flagged pairs (fail bit 1) certified leaf by leaf against the sliced matrix; then the counts / hits of all pairs
// (chunks == nullptr: no tile passes ran, every pair is certified here)
// (n_flagged / flag_list: the compact list of the flagged pairs, when the tile passes ran)
void launch_block_fallback(const QueryArgs &a, const uint2 *sorted, const uint32_t *n_pairs_ptr, const uint32_t *fail, uint8_t *failb,
                           const uint32_t *pair_chunk, const ChunkDesc *chunks, uint32_t launched_passes, const unsigned int *n_flagged,
                           const uint32_t *flag_list, uint32_t flag_cap, hipStream_t st);
// block mode with k-mer entries: turns the miss bytes of the binned pairs into failure bytes (FinalizeArgs: fail, kmiss, kall,
// pair_kpos, pair_chunk, chunks, launched_passes, sorted, off, hp, threshold)
void launch_block_count(const FinalizeArgs &a, const uint32_t *n_pairs_ptr, uint8_t *failb, hipStream_t st);
// block mode on trees with guard columns: a candidate leaf that has not failed passes only if its guards pass (certified
// against the sliced matrix)
void launch_block_guards(const QueryArgs &a, const uint2 *sorted, const uint32_t *n_pairs_ptr, uint8_t *failb, hipStream_t st);
void launch_verify(const VerifyArgs &a, int blocks, int threads, hipStream_t st);
// list = the sorted pairs with a non-zero fail word, in order within runs; *n_out += their number (thresholds < 1 after tile passes)
void launch_collect_open(const uint32_t *fail, const uint32_t *n_pairs_ptr, uint32_t *list, uint32_t cap, unsigned int *n_out, hipStream_t st);
void launch_finalize(const FinalizeArgs &a, hipStream_t st);
// thresholds < 1 after tile passes with k-mer entries: flags (fail bit 1) the binned pairs whose prefix of k-mers does not decide them
void launch_prefix_open(const FinalizeArgs &a, const uint4 *meta, const uint32_t *n_pairs_ptr, uint32_t *fail, hipStream_t st);

void launch_insert(const HashParams &hp, const uint8_t *d_genomes, const uint64_t *d_goff, uint32_t n_genomes,
                   const uint32_t *d_leaf_row, uint64_t *bits, uint64_t n_words, hipStream_t st);
// one genome (device memory, len bytes) into filter row `row`
void launch_insert_one(const HashParams &hp, const uint8_t *d_genome, uint64_t len, uint32_t row, uint64_t *bits, uint64_t n_words, hipStream_t st);
// The greedy placement of one new leaf (BloomTree::insert, bloom_tree.rs:187-245) walked on the device in one launch; the tree's
// shape is mirrored in device memory: a TopoNode per node, state[0] = the root's index (-1: empty; state[2 + (seq & 1)] is
// where launch number seq reads it, state[2 + (~seq & 1)] where it leaves it), state[1] = error word
// (1: a node with one child was met; 2: the grid's barrier timed out), sync = GREEDY_SYNC_LINES lines of GREEDY_SYNC_STRIDE
// bytes, zeroed once (the barrier's counters, generation words and distance accumulators, a line each: see grid_turn).
// `blocks` must not exceed the number of CUs (every block has to be resident).
struct TopoNode {
    int32_t left, right;   // node indices, -1: none
    uint32_t row;          // filter row
    uint32_t pad_;
};
constexpr int GREEDY_MAX_BLOCKS = 256;
constexpr uint32_t GREEDY_SYNC_STRIDE = 4096, GREEDY_SYNC_LINES = 49;
void launch_greedy_insert(uint64_t *bits, uint64_t n_words, TopoNode *topo, int *state, unsigned long long *sync, int leaf_node, int internal_node,
                          uint32_t new_row, uint32_t int_row, uint32_t seq, int blocks, hipStream_t st);
// dst[i] = a[i] | b[i] over rows given as triples (dst,a,b); b == 0xffffffff: copy a.
void launch_union(uint64_t *bits, uint64_t n_words, const uint32_t *d_triples, uint32_t n_triples, hipStream_t st);
// bits[cur] |= bits[new]; per-block partial Hamming distances: sum_b d_out[2b] = hamming(bits[left], bits[new]),
// sum_b d_out[2b+1] = hamming(bits[right], bits[new]), b < INSERT_STEP_BLOCKS
constexpr uint32_t INSERT_STEP_BLOCKS = 1024;
void launch_insert_step(uint64_t *bits, uint64_t n_words, uint32_t cur_row, uint32_t new_row, uint32_t left_row,
                        uint32_t right_row, unsigned long long *d_out, hipStream_t st);
// fail[e] != 0 iff child has a bit the parent lacks
void launch_superset(const uint64_t *bits, uint64_t n_words, const uint32_t *d_edges /*(parent,child)*/, uint32_t n_edges,
                     uint32_t *d_fail, hipStream_t st);
void launch_transpose(const uint64_t *bits, uint64_t n_words, const uint32_t *d_col_row, uint32_t n_cols, uint32_t *S,
                      uint32_t rw, uint64_t group_stride, uint32_t group_log2, hipStream_t st);
// dst[i] = a[i] + b[i], or a[i] - b[i] (u64 counters; the count reductions over replicas / ranks work on what a replica
// counted since it was opened: counters - base)
void launch_counts_op(unsigned long long *dst, const unsigned long long *a, const unsigned long long *b, uint32_t n, bool subtract, hipStream_t st);
// PFQ_WANT_HITS: the CSR read -> leaves from the unordered (read, leaf) hit pairs, on the device.  launch_hits_csr: counts per
// read (d_cnt, zeroed by the caller; reads flagged in d_allhit list every leaf) and their exclusive scan d_off[n_reads + 1];
// launch_hits_fill: the leaves, ascending within a read.  d_sums: ceil(n_reads / 4096) + 1 words of scratch.
void launch_hits_csr(const uint2 *d_pairs, uint64_t n_pairs, const uint8_t *d_allhit, uint64_t n_reads, uint32_t n_leaves, bool any_allhit,
                     uint32_t *d_cnt, unsigned long long *d_sums, unsigned long long *d_off, hipStream_t st);
void launch_hits_fill(const uint2 *d_pairs, uint64_t n_pairs, const uint8_t *d_allhit, uint64_t n_reads, const unsigned long long *d_off,
                      uint32_t *d_cnt, uint32_t *d_leaves, hipStream_t st);
void launch_debug_indices(const HashParams &hp, const uint8_t *d_seq, uint64_t len, uint64_t *d_out, hipStream_t st);
void launch_synth_genomes(uint8_t *d_out, uint64_t n_genomes, uint64_t genome_len, uint64_t seed_base, hipStream_t st);
void launch_synth_reads(uint8_t *d_out, uint64_t first, uint64_t n_reads, uint64_t read_len, const uint8_t *d_genomes,
                        uint64_t genome_len, uint64_t n_genomes, uint64_t seed, hipStream_t st);

}  // namespace pfq
