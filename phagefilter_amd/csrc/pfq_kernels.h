// pfq_kernels.h — argument blocks and launch wrappers shared by the kernels (pfq_kernels.hip) and the host
// side of libpfq (pfq_host.cpp).  Device layout, see DESIGN.md §3:
//
//   bits   u64[n_filters][n_words]   node-major filters in the reference's own bit order (BitVec<usize,Lsb0>,
//                                    bloom_filter.rs:86): bit idx = word idx>>6, mask 1<<(idx&63).
//   S      u32[n_words*64][rw]       "sliced" matrix: row = bit index, column = leaf (left-to-right DFS order)
//                                    followed by guard columns (ancestors whose ⊇ check failed).  One 128-B line
//                                    answers the same probe for 1024 leaves.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pfq_device.h"

namespace pfq {

enum StatSlot { ST_CANDIDATES = 0, ST_HITS = 1, ST_ALLHIT = 2, ST_ALG_BYTES = 3, ST_DEFERRED = 4, ST_N = 8 };

struct QueryArgs {
    HashParams hp;
    const uint8_t *seq;
    const uint64_t *off;
    uint64_t n_reads;
    float threshold;
    // sliced matrix
    const uint32_t *S;
    uint32_t rw, rw_log2;        // row words (power of two <= 64)
    uint32_t n_leaves, n_cols;   // leaf columns, leaf+guard columns
    const uint32_t *guard_off;   // [n_leaves+1] CSR into guard_col (may be all zeros)
    const uint32_t *guard_col;
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
};

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
    const uint4 *recs;           // probe records written by k_classify<DEFER> (nullptr: re-hash per slice)
    uint32_t chunk;
};

struct FinalizeArgs {
    HashParams hp;
    const uint64_t *off;
    const uint2 *sorted;
    const uint32_t *bucket_off;  // [(n_leaves << sub_log2) + 1]
    uint32_t sub_log2;
    const uint32_t *fail;
    uint32_t n_leaves;
    unsigned long long *counts;
    uint2 *hit_pairs;
    uint64_t hit_cap;
    unsigned long long *hit_cursor;
    unsigned long long *stats;
};

// launches (all asynchronous on `st`)
void launch_classify(const QueryArgs &a, bool defer, bool counts_mode, int blocks, hipStream_t st);
void launch_bucket_scan(const uint32_t *bucket_cnt, uint32_t *bucket_off, uint32_t *bucket_cur, uint32_t n, hipStream_t st);
void launch_bucket_scatter(const uint2 *pairs, const unsigned long long *n_pairs_ptr, uint64_t pair_cap,
                           const uint32_t *bucket_off, uint32_t *bucket_cur, uint32_t sub_log2, uint2 *sorted, uint4 *meta,
                           const uint64_t *read_off, const uint32_t *col_row, int blocks, hipStream_t st);
void launch_verify(const VerifyArgs &a, int blocks, int threads, hipStream_t st);
void launch_finalize(const FinalizeArgs &a, hipStream_t st);

void launch_insert(const HashParams &hp, const uint8_t *d_genomes, const uint64_t *d_goff, uint32_t n_genomes,
                   const uint32_t *d_leaf_row, uint64_t *bits, uint64_t n_words, hipStream_t st);
// dst[i] = a[i] | b[i] over rows given as triples (dst,a,b); b == 0xffffffff: copy a.
void launch_union(uint64_t *bits, uint64_t n_words, const uint32_t *d_triples, uint32_t n_triples, hipStream_t st);
// fail[e] != 0 iff child has a bit the parent lacks
void launch_superset(const uint64_t *bits, uint64_t n_words, const uint32_t *d_edges /*(parent,child)*/, uint32_t n_edges,
                     uint32_t *d_fail, hipStream_t st);
void launch_transpose(const uint64_t *bits, uint64_t n_words, const uint32_t *d_col_row, uint32_t n_cols, uint32_t *S,
                      uint32_t rw, hipStream_t st);
void launch_debug_indices(const HashParams &hp, const uint8_t *d_seq, uint64_t len, uint64_t *d_out, hipStream_t st);
void launch_synth_genomes(uint8_t *d_out, uint64_t n_genomes, uint64_t genome_len, uint64_t seed_base, hipStream_t st);
void launch_synth_reads(uint8_t *d_out, uint64_t first, uint64_t n_reads, uint64_t read_len, const uint8_t *d_genomes,
                        uint64_t genome_len, uint64_t n_genomes, uint64_t seed, hipStream_t st);

}  // namespace pfq
