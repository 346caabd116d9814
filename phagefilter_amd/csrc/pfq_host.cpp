// pfq_host.cpp — host side of libpfq: the tree model (BloomTree/BloomNode, bloom_tree.rs:29-61), the
// tree.bin / .bf reader and writer (bincode 1.3.3 + bitvec 1.0.1 layouts, bloom_tree.rs:339-386,
// bloom_filter.rs:153-205), the HBM layout builder and the query orchestration behind the C ABI of include/pfq.h.
// Compiled with hipcc together with pfq_kernels.hip.  No CPU compute path exists here: all filter work is
// done by the kernels and every entry point fails with PFQ_ERR_DEVICE when no gfx950 device is usable.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl is loaded when pfq_trees_allreduce_counts first needs it
#include <dlfcn.h>

#include <algorithm>
#include <thread>
#include <mutex>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/pfq.h"
#include "pfq_kernels.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(PFQ_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + \
                                            std::to_string(__LINE__) + ")");                                 \
    } while (0)
#define PFQ_TRY(expr)              \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != PFQ_OK) return rc_; \
    } while (0)

struct Node {
    int32_t left = -1, right = -1, parent = -1;
    std::string bf_path;  // relative file name, joined to the db dir like cache.rs:62
    bool has_tax = false;
    std::string tax_id;
    uint64_t mapped_reads = 0;
    uint64_t base_reads = 0;  // mapped_reads as stored / last reset, imported or reduced (see pfq_tree::d_counts_base)
    uint32_t filter = 0;  // row of d_bits
    uint32_t depth = 0;
    bool is_leaf() const { return left < 0 && right < 0; }  // bloom_tree.rs:416-418
};

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    hipError_t ensure(size_t want) {
        if (want <= n) return hipSuccess;
        release();
        hipError_t e = hipMalloc(&p, want * sizeof(T));
        if (e == hipSuccess) n = want;
        return e;
    }
    size_t bytes() const { return n * sizeof(T); }
};

// Tuning / test knobs.  Never needed for correct results.  The environment is read ONCE, when a tree is
// created or opened; afterwards pfq_set_option changes a knob of that tree.  -1 / unset = the built-in choice.
struct Knobs {
    long long record_gb = -1, tile_gb = -1, tile_entries = -1, slice_kb = -1;
    long long verify_blocks = -1, verify_chunk = -1, verify_sub = -1, verify_threads = -1, bin_blocks = -1, test_blocks = -1;
    long long tile = -1, tile_counts = -1, no_tail_batch = -1, bin_narrow = -1, bin_wide = -1, bin_debug = -1, block = -1;
    long long coarse = -1, coarse_cols = -1, coarse_probes = -1, group_log2 = -1, screen_recs = -1, coarse_min_leaves = -1, greedy_host = -1;
};
struct KnobName {
    const char *name;
    long long Knobs::*field;
};
const KnobName KNOBS[] = {
    {"PFQ_RECORD_GB", &Knobs::record_gb},       {"PFQ_TILE_GB", &Knobs::tile_gb},
    {"PFQ_TILE_ENTRIES", &Knobs::tile_entries}, {"PFQ_SLICE_KB", &Knobs::slice_kb},
    {"PFQ_VERIFY_BLOCKS", &Knobs::verify_blocks}, {"PFQ_VERIFY_CHUNK", &Knobs::verify_chunk},
    {"PFQ_VERIFY_SUB", &Knobs::verify_sub},     {"PFQ_VERIFY_THREADS", &Knobs::verify_threads},
    {"PFQ_BIN_BLOCKS", &Knobs::bin_blocks},     {"PFQ_TEST_BLOCKS", &Knobs::test_blocks},
    {"PFQ_TILE", &Knobs::tile},                 {"PFQ_TILE_COUNTS", &Knobs::tile_counts},
    {"PFQ_NO_TAIL_BATCH", &Knobs::no_tail_batch}, {"PFQ_BIN_NARROW", &Knobs::bin_narrow},
    {"PFQ_BIN_WIDE", &Knobs::bin_wide},
#ifdef PFQ_EXPERIMENTS  // (timing experiments with wrong results: not in the library as shipped)
    {"PFQ_BIN_DEBUG", &Knobs::bin_debug},
#endif
    {"PFQ_BLOCK", &Knobs::block},
    {"PFQ_COARSE", &Knobs::coarse},             {"PFQ_COARSE_COLS", &Knobs::coarse_cols},
    {"PFQ_COARSE_PROBES", &Knobs::coarse_probes}, {"PFQ_GROUP_LOG2", &Knobs::group_log2},
    {"PFQ_SCREEN_RECS", &Knobs::screen_recs},   {"PFQ_COARSE_MIN_LEAVES", &Knobs::coarse_min_leaves},
    {"PFQ_GREEDY_HOST", &Knobs::greedy_host},
};
bool set_knob(Knobs &k, const char *name, const char *value) {
    for (const KnobName &kn : KNOBS)
        if (!strcmp(kn.name, name)) {
            k.*(kn.field) = (value && *value) ? strtoll(value, nullptr, 10) : -1;
            return true;
        }
    return false;
}
Knobs knobs_from_env() {
    Knobs k;
    for (const KnobName &kn : KNOBS)
        if (const char *e = getenv(kn.name)) set_knob(k, kn.name, e);
    return k;
}

}  // namespace

namespace {
std::atomic<long> g_open_trees{0};
void release_communicators();  // (the kept RCCL communicators go when the last tree of the process is closed)
}  // namespace

struct pfq_tree {
    pfq_tree() { ++g_open_trees; }
    pfq_tree(const pfq_tree &) = delete;
    ~pfq_tree() {
        if (--g_open_trees == 0) release_communicators();
    }
    int device = 0;
    Knobs knobs = knobs_from_env();
    // ---- model
    std::vector<Node> nodes;  // pre-order, root = 0
    int32_t root = -1;
    float false_pos_rate = 0.001f;
    uint32_t largest_expected_genome = 0;
    uint64_t kmer_size = 0, nbits = 0, seed1 = 0, seed2 = 0, n_words = 0;
    uint32_t num_hashes = 0;
    std::vector<std::string> filter_paths;
    std::unordered_set<std::string> path_set;  // filter_paths as a set, kept by pfq_tree_insert (a name's .bf must be new: a scan per insertion was 40 us at 2000 nodes)
    std::vector<uint8_t> edge_ok;  // per node: parent(v) ⊇ v verified
    bool superset_all = true;
    uint64_t shard_first_leaf = 0, tree_leaves = 0;  // subtree shards (pfq_tree_open_subtree)
    bool is_shard = false;
    // deferred pairs per read seen by recent calls (related genomes: a read passes several leaves): sizes the pair
    // buffer and the probe buckets of the next call.  The cursor of a call lands in pinned memory asynchronously.
    unsigned long long *h_pair_cursor = nullptr;
    hipEvent_t hint_ev = nullptr;
    uint64_t hint_reads = 0, hint_entry_cap = 0, passes_hint = 1;
    double pairs_per_read = 1.0, hits_per_read = 1.0;
    double dirty_frac = 1.0;           // thresholds < 1: share of the last call's deferred pairs with a k-mer missing (1: unknown)
    bool hint_counts = false;
    uint32_t last_sub_log2 = 0;
    bool topology_dirty = false;       // nodes appended by pfq_tree_insert: renumber + verify before the next use
    uint64_t internal_counter = 0;     // names of internal nodes created by pfq_tree_insert
    size_t n_rows = 0, row_capacity = 0;  // filter rows in use / allocated in d_bits
    DevBuf<uint32_t> d_build;          // insert scratch: leaf row, union triple
    DevBuf<unsigned long long> d_dist; // insert scratch: the device walk's barrier lines (GREEDY_SYNC_*); the host walk's per-block partials
    // pfq_tree_insert walks the tree on the device (k_greedy_insert): its shape is mirrored there and the host's left / right /
    // root are brought up to date when they are next needed (finish_topology)
    DevBuf<pfq::TopoNode> d_topo;
    DevBuf<int> d_walk;                // [0] root (for the host), [1] error word, [2..3] the root as the walk's launches hand it on
    bool topo_on_device = false;       // d_topo / d_walk mirror the host's nodes
    bool topo_pending = false;         // insertions ran since the host last read the shape back
    DevBuf<uint8_t> d_gseq[4];         // genomes of the insertions in flight (ring)
    hipEvent_t gseq_free[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t gseq_copied[4] = {nullptr, nullptr, nullptr, nullptr};
    uint8_t *h_gseq[4] = {nullptr, nullptr, nullptr, nullptr};  // page-locked staging of the same ring (the caller's buffer is free when pfq_tree_insert returns)
    size_t h_gseq_n[4] = {0, 0, 0, 0};
    uint32_t gseq_next = 0;
    int greedy_blocks = 0;
    uint32_t walk_seq = 0;             // launches of the device walk since d_walk was written (parity: where the root is read / left)

    pfq::HashParams hp{};
    // ---- device: node-major filters
    DevBuf<uint64_t> d_bits;
    // ---- device: layout of the current leaf set
    bool layout_valid = false;
    std::vector<int32_t> leaves;     // node ids, left-to-right
    std::vector<uint32_t> col_row;   // column -> filter row
    std::vector<uint32_t> guard_off, guard_col;
    uint32_t rw = 1, rw_log2 = 0, n_cols = 0;
    uint64_t leaf_cap = 0, guard_cap = 0;  // regions of the deferred-pair buffer
    uint32_t n_groups = 1;           // column groups of the sliced matrix (2^group_log2 columns each when there are several)
    uint32_t group_log2 = 11;
    uint64_t group_stride = 0;       // dwords per group: (n_words * 64 + 1) * rw
    // two-level frontier (trees of several leaf groups): coarse sliced matrix over an antichain of internal nodes
    bool coarse_valid = false;
    uint32_t coarse_cols = 0, coarse_rw = 0, coarse_rw_log2 = 0;
    double coarse_fill = 0.0;        // mean share of set bits of the coarse columns' filters
    DevBuf<uint32_t> d_Sc, d_cgrp, d_glists, d_glong;
    DevBuf<unsigned int> d_gcur;
    uint32_t last_coarse_cols = 0, last_coarse_probes = 0, last_leaf_groups = 0;
    DevBuf<uint32_t> d_S, d_col_row, d_guard_off, d_guard_col;
    DevBuf<uint32_t> d_owner, d_owner_sorted, d_gfail;  // trees with guard columns, bucketed path: leaf pair of every pair slot
    DevBuf<unsigned long long> d_counts;
    // what the counters held when the tree was opened (BloomNode::mapped_reads stored in tree.bin), last reset, imported or
    // reduced: the reductions over replicas / ranks add up counters - base, so that stored counts are not added once per replica
    DevBuf<unsigned long long> d_counts_base, d_counts_delta;
    // ---- query scratch
    DevBuf<unsigned long long> d_stats, d_cursors;  // cursors: [0] hit, [1] pair, [2] tile entries, [3] lo: chunks, hi: flagged pairs, [4] long reads, [5] miss words, [6] dirty pairs, [7] lo: open pairs after the tile passes (thresholds < 1), [8] guard pairs, [9] k-mer miss bytes handed out
    DevBuf<uint32_t> d_entries, d_pair_chunk, d_leaf_chunk0, d_flag_list;  // LDS-tile certificates
    DevBuf<pfq::ChunkDesc> d_chunks;
    DevBuf<unsigned int> d_gfill, d_binq;
    DevBuf<uint8_t> d_kmiss;
    DevBuf<uint8_t> d_kall;            // block mode with k-mer entries: per k-mer, "in no candidate leaf of the block"
    DevBuf<uint8_t> d_T, d_failb;      // block mode: byte-per-index tables of the blocks of 8 leaves; failure bytes per (pair, leaf)
    bool tables_valid = false;         // d_T matches the current leaf set
    double cand_per_read = 1.0;        // candidate leaves per read seen by recent calls (related genomes: several): chooses block mode
    bool have_cand_hint = false;       // cand_per_read describes this tree's workload (else a sample of the block is screened first)
    uint32_t last_block_mode = 0;
    DevBuf<uint32_t> d_round_k0, d_n_rounds, d_pair_kpos;  // thresholds < 1: LDS-tile passes with k-mer entries
    uint32_t last_tile_mode = 0, last_passes = 1;
    DevBuf<uint2> d_hit_pairs, d_pairs, d_sorted;
    DevBuf<uint32_t> d_bucket, d_fail;  // bucket: cnt[n], off[n+1], cur[n]
    DevBuf<unsigned int> d_queue;
    DevBuf<uint8_t> d_allhit, d_seq, d_seq2;  // d_seq / d_seq2, d_off / d_off2: input buffers of pfq_query_batch, alternating
    DevBuf<uint64_t> d_off2;
    hipStream_t copy_stream = nullptr;
    hipEvent_t in_free[2] = {nullptr, nullptr};
    bool in_used[2] = {false, false};
    int in_slot = 0;
    DevBuf<unsigned long long> d_miss_words;  // thresholds < 1: k-mer miss bits of every deferred pair
    DevBuf<uint32_t> d_miss_pos, d_bucket_w;  // first word per sorted pair; per-bucket word counts / offsets / cursors
    DevBuf<uint32_t> d_long;
    DevBuf<uint4> d_recs;  // probe records of the bucketed path (16 B per read byte)
    DevBuf<uint4> d_meta;  // resolved per-pair metadata for the record-driven verify
    DevBuf<uint64_t> d_off;
    DevBuf<unsigned long long> d_counts_snapshot;
    hipStream_t last_stream = nullptr;
    bool have_last_stream = false;
    int force_path = -1;
    // ---- profiling (HIP events on the launch stream)
    std::vector<hipEvent_t> prof_ev;  // PROF_EV events per recorded call
    size_t prof_cap = 0, prof_used = 0;
    std::vector<uint8_t> prof_bucketed;
    uint32_t last_path = 0, last_slices = 1;
    uint64_t last_n_reads = 0;
    // ---- outputs (library-owned)
    std::vector<std::string> out_tax;
    std::vector<const char *> out_tax_ptr;
    std::vector<uint64_t> out_counts;
    // per-read hit lists of the last PFQ_WANT_HITS call: built on the device, copied into page-locked host memory
    DevBuf<uint32_t> d_hit_cnt, d_hit_leaves;
    DevBuf<unsigned long long> d_hit_sums, d_hit_off;
    uint64_t *h_hit_off = nullptr;
    uint32_t *h_hit_leaves = nullptr;
    size_t h_hit_off_cap = 0, h_hit_leaves_cap = 0;
};

namespace {

// ------------------------------------------------------------------------------------------------------------
// bincode cursor
// ------------------------------------------------------------------------------------------------------------
struct Cur {
    const uint8_t *b;
    size_t n, p = 0;
    bool ok = true;
    const uint8_t *take(size_t k) {
        if (!ok || k > n - p) {
            ok = false;
            return nullptr;
        }
        const uint8_t *r = b + p;
        p += k;
        return r;
    }
    uint8_t u8() { auto q = take(1); return q ? *q : 0; }
    uint32_t u32() { auto q = take(4); uint32_t v = 0; if (q) memcpy(&v, q, 4); return v; }
    uint64_t u64() { auto q = take(8); uint64_t v = 0; if (q) memcpy(&v, q, 8); return v; }
    float f32() { auto q = take(4); float v = 0; if (q) memcpy(&v, q, 4); return v; }
    std::string str() {
        uint64_t len = u64();
        if (!ok || len > n - p) { ok = false; return {}; }
        auto q = take((size_t)len);
        return q ? std::string((const char *)q, (size_t)len) : std::string();
    }
};

bool read_file(const std::string &path, std::vector<uint8_t> &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(sz > 0 ? (size_t)sz : 0);
    size_t got = out.empty() ? 0 : fread(out.data(), 1, out.size(), f);
    fclose(f);
    return got == out.size();
}

// BloomNode, pre-order (bloom_tree.rs:50-61): left, right, bloom_filter_path, tax_id, mapped_reads
int parse_node(Cur &c, pfq_tree &t, int32_t parent, uint32_t depth, int32_t &out_idx) {
    if (depth > 100000) return fail(PFQ_ERR_FORMAT, "tree.bin: nesting too deep");
    int32_t v = (int32_t)t.nodes.size();
    t.nodes.emplace_back();
    t.nodes[v].parent = parent;
    t.nodes[v].depth = depth;
    for (int side = 0; side < 2; ++side) {
        uint8_t tag = c.u8();
        if (!c.ok || tag > 1) return fail(PFQ_ERR_FORMAT, "tree.bin: bad Option tag in BloomNode");
        int32_t child = -1;
        if (tag == 1) PFQ_TRY(parse_node(c, t, v, depth + 1, child));
        (side == 0 ? t.nodes[v].left : t.nodes[v].right) = child;
    }
    t.nodes[v].bf_path = c.str();
    uint8_t tag = c.u8();
    if (!c.ok || tag > 1) return fail(PFQ_ERR_FORMAT, "tree.bin: bad Option tag for tax_id");
    if (tag == 1) {
        t.nodes[v].has_tax = true;
        t.nodes[v].tax_id = c.str();
    }
    t.nodes[v].mapped_reads = c.u64();
    t.nodes[v].base_reads = t.nodes[v].mapped_reads;
    if (!c.ok) return fail(PFQ_ERR_FORMAT, "tree.bin: truncated BloomNode");
    out_idx = v;
    return PFQ_OK;
}

uint64_t pow2_64_mod(uint64_t d) {  // 2^64 mod d
    uint64_t r = (~0ull) % d;       // (2^64 - 1) mod d
    return (r + 1 == d) ? 0 : r + 1;
}

int setup_hash_params(pfq_tree &t) {
    if (t.kmer_size < 1 || t.kmer_size > pfq::KMAX)
        return fail(PFQ_ERR_UNSUPPORTED, "kmer_size " + std::to_string(t.kmer_size) + " outside the device path's 1.." +
                                              std::to_string(pfq::KMAX));
    if (t.nbits < 1 || t.nbits >= (1ull << 32))
        return fail(PFQ_ERR_UNSUPPORTED, "filter size " + std::to_string(t.nbits) + " bits outside 1..2^32-1");
    if (t.num_hashes < 1) return fail(PFQ_ERR_FORMAT, "num_hashes == 0");
    t.n_words = (t.nbits + 63) / 64;
    t.hp.k = (uint32_t)t.kmer_size;
    t.hp.num_hashes = t.num_hashes;
    t.hp.nbits = t.nbits;
    t.hp.bar_m = (~0ull) / t.nbits;
    t.hp.w64 = pow2_64_mod(t.nbits);
    // FxHasher after write_usize(seed) (hasher.rs:16-18) and the length prefix of <[u8] as Hash>::hash
    // (times K once more: the kernels finish with rotl(a + hash_bytes * K), see pfq_device.h)
    t.hp.a1 = (t.seed1 * pfq::FX_K + t.kmer_size) * pfq::FX_K * pfq::FX_K;
    t.hp.a2 = (t.seed2 * pfq::FX_K + t.kmer_size) * pfq::FX_K * pfq::FX_K;
    return PFQ_OK;
}

int use_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(PFQ_ERR_DEVICE, "no HIP device available (libpfq has no CPU fallback)");
    if (device < 0 || device >= n) return fail(PFQ_ERR_ARG, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    return PFQ_OK;
}

void relink(pfq_tree &t) {  // parent/depth after topology edits
    if (t.root < 0) return;
    std::vector<int32_t> st{t.root};
    t.nodes[t.root].parent = -1;
    t.nodes[t.root].depth = 0;
    while (!st.empty()) {
        int32_t v = st.back();
        st.pop_back();
        for (int32_t c : {t.nodes[v].left, t.nodes[v].right})
            if (c >= 0) {
                t.nodes[c].parent = v;
                t.nodes[c].depth = t.nodes[v].depth + 1;
                st.push_back(c);
            }
    }
}

std::vector<int32_t> leaves_dfs(const pfq_tree &t) {  // get_leaf_counts order, query.rs:197-218
    std::vector<int32_t> out;
    if (t.root < 0) return out;
    std::vector<int32_t> st{t.root};
    while (!st.empty()) {
        int32_t v = st.back();
        st.pop_back();
        const Node &nd = t.nodes[v];
        if (nd.is_leaf()) out.push_back(v);
        else {
            if (nd.right >= 0) st.push_back(nd.right);
            if (nd.left >= 0) st.push_back(nd.left);
        }
    }
    return out;
}

// Fold the device counters back into BloomNode::mapped_reads (before the leaf set changes / on read-out).
int sync_counts_to_nodes(pfq_tree &t) {
    if (!t.layout_valid || t.leaves.empty()) return PFQ_OK;
    HIP_TRY(hipDeviceSynchronize());
    std::vector<unsigned long long> h(t.leaves.size());
    HIP_TRY(hipMemcpy(h.data(), t.d_counts.p, h.size() * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < h.size(); ++i) t.nodes[t.leaves[i]].mapped_reads = h[i];
    HIP_TRY(hipMemcpy(h.data(), t.d_counts_base.p, h.size() * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < h.size(); ++i) t.nodes[t.leaves[i]].base_reads = h[i];
    return PFQ_OK;
}

// parent ⊇ child on every edge of the full tree, on the device.
int verify_supersets(pfq_tree &t) {
    t.edge_ok.assign(t.nodes.size(), 1);
    t.superset_all = true;
    std::vector<uint32_t> edges;
    std::vector<int32_t> edge_node;
    std::vector<uint8_t> reach(t.nodes.size(), 0);
    if (t.root >= 0) {
        std::vector<int32_t> st{t.root};
        while (!st.empty()) {
            int32_t v = st.back();
            st.pop_back();
            reach[v] = 1;
            if (t.nodes[v].left >= 0) st.push_back(t.nodes[v].left);
            if (t.nodes[v].right >= 0) st.push_back(t.nodes[v].right);
        }
    }
    for (size_t v = 0; v < t.nodes.size(); ++v)
        if (reach[v] && t.nodes[v].parent >= 0 && t.nodes[t.nodes[v].parent].filter != t.nodes[v].filter) {
            edges.push_back(t.nodes[t.nodes[v].parent].filter);
            edges.push_back(t.nodes[v].filter);
            edge_node.push_back((int32_t)v);
        }
    if (edge_node.empty()) return PFQ_OK;
    DevBuf<uint32_t> d_edges, d_fail;
    HIP_TRY(d_edges.ensure(edges.size()));
    HIP_TRY(d_fail.ensure(edge_node.size()));
    HIP_TRY(hipMemcpy(d_edges.p, edges.data(), edges.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d_fail.p, 0, edge_node.size() * 4));
    // blockIdx.y is limited to 65535
    for (size_t e0 = 0; e0 < edge_node.size(); e0 += 32768) {
        uint32_t ne = (uint32_t)std::min<size_t>(32768, edge_node.size() - e0);
        pfq::launch_superset(t.d_bits.p, t.n_words, d_edges.p + 2 * e0, ne, d_fail.p + e0, nullptr);
    }
    HIP_TRY(hipGetLastError());
    std::vector<uint32_t> h(edge_node.size());
    HIP_TRY(hipMemcpy(h.data(), d_fail.p, h.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < h.size(); ++i)
        if (h[i]) {
            t.edge_ok[edge_node[i]] = 0;
            t.superset_all = false;
        }
    return PFQ_OK;
}

// Build the sliced matrix for the current leaf set.
uint64_t needed_bits_f32(float rate, uint32_t items) {  // bloom_filter.rs:354-357, f32 arithmetic
    const float ln2 = 0.693147180559945309417232121458176568f;
    const float ln22 = ln2 * ln2;
    const float v = roundf((float)items * (logf(1.0f / rate) / ln22));
    return v <= 0 ? 0 : (uint64_t)v;
}
uint32_t optimal_num_hashes_f32(uint64_t bits, uint32_t items) {  // bloom_filter.rs:342-350
    const float ln2 = 0.693147180559945309417232121458176568f;
    const float v = roundf((float)bits / (float)items * ln2);
    const uint32_t h = v <= 0 ? 0 : (uint32_t)v;
    return std::min<uint32_t>(std::max<uint32_t>(h, 2), 200);
}

// Nodes in pre-order again (root = 0) after pfq_tree_insert appended some; parents, depths and the ⊇ flags follow.
// The shape the insertions on the device left (k_greedy_insert): children of every node, the root.
int sync_topology(pfq_tree &t) {
    if (!t.topo_pending) return PFQ_OK;
    HIP_TRY(hipDeviceSynchronize());
    std::vector<pfq::TopoNode> h(t.nodes.size());
    int st[2] = {-1, 0};
    HIP_TRY(hipMemcpy(h.data(), t.d_topo.p, h.size() * sizeof(pfq::TopoNode), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(st, t.d_walk.p, sizeof st, hipMemcpyDeviceToHost));
    t.topo_pending = false;
    if (st[1] == 1) return fail(PFQ_ERR_FORMAT, "Node with only one child encountered - should not happen. (bloom_tree.rs:209)");
    if (st[1] != 0) return fail(PFQ_ERR_DEVICE, "the insertion kernel's grid barrier timed out");
    for (size_t v = 0; v < h.size(); ++v) {
        t.nodes[v].left = h[v].left;
        t.nodes[v].right = h[v].right;
    }
    t.root = st[0];
    return PFQ_OK;
}

int finish_topology(pfq_tree &t) {
    PFQ_TRY(sync_topology(t));
    if (!t.topology_dirty) return PFQ_OK;
    std::vector<int32_t> order, new_of(t.nodes.size(), -1);
    if (t.root >= 0) {
        std::vector<int32_t> st{t.root};
        while (!st.empty()) {
            const int32_t v = st.back();
            st.pop_back();
            new_of[v] = (int32_t)order.size();
            order.push_back(v);
            if (t.nodes[v].right >= 0) st.push_back(t.nodes[v].right);
            if (t.nodes[v].left >= 0) st.push_back(t.nodes[v].left);
        }
    }
    std::vector<Node> nn;
    nn.reserve(order.size());
    for (int32_t v : order) {
        Node nd = std::move(t.nodes[v]);
        if (nd.left >= 0) nd.left = new_of[nd.left];
        if (nd.right >= 0) nd.right = new_of[nd.right];
        nn.push_back(std::move(nd));
    }
    t.nodes.swap(nn);
    t.root = t.nodes.empty() ? -1 : 0;
    relink(t);
    t.topology_dirty = false;
    t.topo_on_device = false;  // (the nodes were renumbered)
    PFQ_TRY(verify_supersets(t));
    return PFQ_OK;
}

// Room for `rows` filter rows in d_bits, keeping the rows in use.
int reserve_rows(pfq_tree &t, size_t rows) {
    if (rows <= t.row_capacity) return PFQ_OK;
    size_t cap = std::max<size_t>(rows, t.row_capacity + t.row_capacity / 2 + 2);
    HIP_TRY(hipDeviceSynchronize());  // (insertions in flight on the tree's own streams still write the old rows)
    uint64_t *p = nullptr;
    HIP_TRY(hipMalloc(&p, cap * t.n_words * 8));
    if (t.n_rows) HIP_TRY(hipMemcpy(p, t.d_bits.p, t.n_rows * t.n_words * 8, hipMemcpyDeviceToDevice));
    t.d_bits.release();
    t.d_bits.p = p;
    t.d_bits.n = cap * t.n_words;
    t.row_capacity = cap;
    return PFQ_OK;
}

// An allocation that may fail without failing the call: the caller then takes the next exact path.
template <typename T>
bool soft_ensure(DevBuf<T> &b, size_t want) {
    if (b.ensure(want) == hipSuccess) return true;
    (void)hipGetLastError();
    return false;
}

// The coarse level of the two-level frontier (pfq::CoarseArgs): an antichain of nodes that covers every leaf, as close to
// the leaves as `max_cols` columns allow (the node with the most leaves below it is split until the budget is spent), in
// left-to-right order; column c = the filter of node anti[c] — the filter the reference tests at that node (cache.rs:56-62
// keys filters by path, so nodes that share a file share the row).  Leaves: t.leaves must be current.
void pick_antichain(const pfq_tree &t, uint32_t max_cols, std::vector<int32_t> &anti, std::vector<uint32_t> &first, std::vector<uint32_t> &count) {
    const size_t nn = t.nodes.size();
    first.assign(nn, 0xffffffffu);
    count.assign(nn, 0);
    for (size_t i = 0; i < t.leaves.size(); ++i) {
        first[t.leaves[i]] = (uint32_t)i;
        count[t.leaves[i]] = 1;
    }
    {   // leaves below every node, first leaf (post-order without recursion)
        std::vector<std::pair<int32_t, int>> st{{t.root, 0}};
        while (!st.empty()) {
            auto &top = st.back();
            const Node &nd = t.nodes[top.first];
            if (top.second == 0) {
                top.second = 1;
                if (nd.left >= 0) st.push_back({nd.left, 0});
                continue;
            }
            if (top.second == 1) {
                top.second = 2;
                if (nd.right >= 0) st.push_back({nd.right, 0});
                continue;
            }
            const int32_t v = top.first;
            st.pop_back();
            for (int32_t c : {t.nodes[v].left, t.nodes[v].right})
                if (c >= 0) {
                    count[v] += count[c];
                    first[v] = std::min(first[v], first[c]);
                }
        }
    }
    auto less = [&](int32_t x, int32_t y) { return count[x] < count[y] || (count[x] == count[y] && x > y); };
    std::vector<int32_t> heap{t.root};
    while (true) {
        const int32_t v = heap.front();
        const Node &nd = t.nodes[v];
        if (nd.is_leaf()) break;  // the largest node is a leaf: every node of the antichain is
        const int kids = (nd.left >= 0) + (nd.right >= 0);
        if (heap.size() + kids - 1 > max_cols) break;
        std::pop_heap(heap.begin(), heap.end(), less);
        heap.pop_back();
        for (int32_t c : {nd.left, nd.right})
            if (c >= 0) {
                heap.push_back(c);
                std::push_heap(heap.begin(), heap.end(), less);
            }
    }
    std::sort(heap.begin(), heap.end(), [&](int32_t x, int32_t y) { return first[x] < first[y]; });
    anti.swap(heap);
}

// Plans the coarse level of the current leaf set: the antichain (1024 columns — one 128-byte line per row — when its filters
// are empty enough, else 2048) and how full its filters are.  n_cols == 0: no coarse level pays (the screens cannot drop a
// column whose filter has nearly every bit set); the frontier then runs flat, every leaf group on every read.
struct CoarsePlan {
    uint32_t n_cols = 0;
    std::vector<uint32_t> rows, cgrp;
    double fill = 0;
};
int plan_coarse(pfq_tree &t, CoarsePlan &plan) {
    plan.n_cols = 0;
    std::vector<uint32_t> budgets;
    if (t.knobs.coarse_cols > 0) budgets.push_back((uint32_t)std::min<long long>(2048, std::max<long long>(2, t.knobs.coarse_cols)));
    else budgets = {1024u, 2048u};
    for (size_t bi = 0; bi < budgets.size(); ++bi) {
        std::vector<int32_t> anti;
        std::vector<uint32_t> first, count;
        pick_antichain(t, budgets[bi], anti, first, count);
        const uint32_t C = (uint32_t)anti.size();
        if (C < 2) return PFQ_OK;
        std::vector<uint32_t> rows(C), cgrp(C);
        for (uint32_t c = 0; c < C; ++c) {
            rows[c] = t.nodes[anti[c]].filter;
            const uint32_t lo = first[anti[c]] >> t.group_log2, hi = (first[anti[c]] + count[anti[c]] - 1) >> t.group_log2;
            cgrp[c] = lo | (hi << 16);
        }
        DevBuf<uint32_t> d_rows;
        DevBuf<unsigned long long> d_pop;
        HIP_TRY(d_rows.ensure(C));
        HIP_TRY(d_pop.ensure(C));
        HIP_TRY(hipMemcpy(d_rows.p, rows.data(), C * 4, hipMemcpyHostToDevice));
        pfq::launch_row_popcount(t.d_bits.p, t.n_words, d_rows.p, C, d_pop.p, nullptr);
        HIP_TRY(hipGetLastError());
        std::vector<unsigned long long> pop(C);
        HIP_TRY(hipMemcpy(pop.data(), d_pop.p, C * 8, hipMemcpyDeviceToHost));
        double fill = 0;
        for (unsigned long long v : pop) fill += (double)v / (double)t.nbits;
        fill /= C;
        if (fill > 0.62 && bi + 1 < budgets.size()) continue;   // try the finer antichain
        if (fill > 0.80 && t.knobs.coarse <= 0) return PFQ_OK;  // too full to prune anything (PFQ_COARSE=1: build it anyway)
        plan.n_cols = C;
        plan.rows.swap(rows);
        plan.cgrp.swap(cgrp);
        plan.fill = fill;
        return PFQ_OK;
    }
    return PFQ_OK;
}
int build_coarse(pfq_tree &t, const CoarsePlan &plan) {
    const uint32_t C = plan.n_cols;
    uint32_t rwc = 16, rwc_log2 = 4;  // (the dense counting screen takes rows of 16, 32 or 64 words)
    while (rwc * 32 < C) {
        rwc <<= 1;
        ++rwc_log2;
    }
    const uint64_t sc_words = ((uint64_t)t.n_words * 64 + 1) * rwc;
    DevBuf<uint32_t> d_rows;
    if (!(soft_ensure(t.d_Sc, sc_words) && soft_ensure(t.d_cgrp, C) && soft_ensure(t.d_gcur, 2 * pfq::MAX_LEAF_GROUPS) && soft_ensure(d_rows, C)))
        return PFQ_OK;  // (no room: flat frontier)
    HIP_TRY(hipMemcpy(d_rows.p, plan.rows.data(), C * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t.d_cgrp.p, plan.cgrp.data(), C * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(t.d_Sc.p, 0, sc_words * 4, nullptr));
    HIP_TRY(hipMemsetAsync(t.d_Sc.p + sc_words - rwc, 0xff, rwc * 4, nullptr));
    pfq::launch_transpose(t.d_bits.p, t.n_words, d_rows.p, C, t.d_Sc.p, rwc, sc_words, 11, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    t.coarse_cols = C;
    t.coarse_rw = rwc;
    t.coarse_rw_log2 = rwc_log2;
    t.coarse_fill = plan.fill;
    t.coarse_valid = true;
    return PFQ_OK;
}

int build_layout(pfq_tree &t) {
    if (t.layout_valid) return PFQ_OK;
    PFQ_TRY(finish_topology(t));
    t.leaves = leaves_dfs(t);
    const size_t nl = t.leaves.size();
    for (int32_t v : t.leaves)
        if (!t.nodes[v].has_tax)
            return fail(PFQ_ERR_FORMAT, "leaf node without tax_id (reference: unwrap panic, query.rs:146)");
    t.col_row.clear();
    t.guard_off.assign(nl + 1, 0);
    t.guard_col.clear();
    std::map<uint32_t, uint32_t> guard_column_of_row;  // filter row -> guard column
    std::vector<uint32_t> guard_rows;
    for (size_t i = 0; i < nl; ++i) {
        t.col_row.push_back(t.nodes[t.leaves[i]].filter);
        bool ok = true;
        for (int32_t v = t.leaves[i]; t.nodes[v].parent >= 0; v = t.nodes[v].parent) {
            ok = ok && t.edge_ok[v];
            if (!ok) {
                uint32_t row = t.nodes[t.nodes[v].parent].filter;
                if (row == t.nodes[t.leaves[i]].filter) continue;  // same filter as the leaf itself
                auto it = guard_column_of_row.find(row);
                uint32_t col;
                if (it == guard_column_of_row.end()) {
                    col = (uint32_t)(nl + guard_rows.size());
                    guard_column_of_row[row] = col;
                    guard_rows.push_back(row);
                } else col = it->second;
                if (std::find(t.guard_col.begin() + t.guard_off[i], t.guard_col.end(), col) == t.guard_col.end())
                    t.guard_col.push_back(col);
            }
        }
        t.guard_off[i + 1] = (uint32_t)t.guard_col.size();
    }
    for (uint32_t r : guard_rows) t.col_row.push_back(r);
    t.n_cols = (uint32_t)t.col_row.size();
    uint32_t need_words = std::max<uint32_t>(1, (t.n_cols + 31) / 32);
    // a wave holds one row of up to 64 dwords (2048 columns) across its lanes; wider trees are cut into column groups,
    // each with a sliced matrix of its own, and the frontier kernels run once per group — on the reads the coarse level
    // lists for the group when the tree has one (two-level frontier: groups of 1024 columns, one 128-byte line per row)
    t.coarse_valid = false;
    t.coarse_cols = 0;
    const size_t coarse_min = t.knobs.coarse_min_leaves >= 1024 ? (size_t)t.knobs.coarse_min_leaves : 2048;
    bool want_coarse = nl > coarse_min && t.knobs.coarse != 0;
    t.group_log2 = 11;
    CoarsePlan plan;
    if (want_coarse) {
        t.group_log2 = (t.knobs.group_log2 == 10 || t.knobs.group_log2 == 11) ? (uint32_t)t.knobs.group_log2 : 10u;
        if (((nl - 1) >> t.group_log2) + 1 > pfq::MAX_LEAF_GROUPS) t.group_log2 = 11;
        if (((nl - 1) >> t.group_log2) + 1 > pfq::MAX_LEAF_GROUPS) want_coarse = false;  // (flat frontier: every group sees every read)
        if (want_coarse) PFQ_TRY(plan_coarse(t, plan));
        if (plan.n_cols == 0) {
            want_coarse = false;
            t.group_log2 = 11;
        }
    }
    if (!want_coarse) t.d_Sc.release();
    const uint32_t group_cols = 1u << t.group_log2;
    t.rw = 4;  // at least 16-byte rows: the dense pre-screen gathers rows with dwordx4 loads
    t.rw_log2 = 2;
    while (t.rw < need_words && t.rw < group_cols / 32) {
        t.rw <<= 1;
        ++t.rw_log2;
    }
    t.n_groups = std::max<uint32_t>(1, (t.n_cols + group_cols - 1) / group_cols);
    t.group_stride = ((uint64_t)t.n_words * 64 + 1) * t.rw;  // + the all-ones row of the group
    if (nl == 0) {
        t.layout_valid = true;
        t.tables_valid = false;
        return PFQ_OK;
    }
    const size_t s_words = (size_t)t.group_stride * t.n_groups;
    if (t.d_S.ensure(s_words) != hipSuccess) {
        (void)hipGetLastError();
        return fail(PFQ_ERR_DEVICE, "not enough device memory for the sliced matrix of " + std::to_string(t.n_cols) +
                                        " leaf+guard columns (" + std::to_string(s_words * 4 >> 20) + " MiB)");
    }
    HIP_TRY(t.d_col_row.ensure(t.col_row.size()));
    HIP_TRY(t.d_guard_off.ensure(t.guard_off.size() + t.n_cols));  // guard lists are indexed by column; pad for guard columns
    HIP_TRY(t.d_guard_col.ensure(std::max<size_t>(1, t.guard_col.size())));
    HIP_TRY(t.d_counts.ensure(nl));
    HIP_TRY(hipMemcpy(t.d_col_row.p, t.col_row.data(), t.col_row.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t.d_guard_off.p, t.guard_off.data(), t.guard_off.size() * 4, hipMemcpyHostToDevice));
    if (!t.guard_col.empty())
        HIP_TRY(hipMemcpy(t.d_guard_col.p, t.guard_col.data(), t.guard_col.size() * 4, hipMemcpyHostToDevice));
    std::vector<unsigned long long> h(nl);
    for (size_t i = 0; i < nl; ++i) h[i] = t.nodes[t.leaves[i]].mapped_reads;
    HIP_TRY(hipMemcpy(t.d_counts.p, h.data(), nl * 8, hipMemcpyHostToDevice));
    HIP_TRY(t.d_counts_base.ensure(nl));
    HIP_TRY(t.d_counts_delta.ensure(nl));
    for (size_t i = 0; i < nl; ++i) h[i] = t.nodes[t.leaves[i]].base_reads;
    HIP_TRY(hipMemcpy(t.d_counts_base.p, h.data(), nl * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(t.d_S.p, 0, s_words * 4, nullptr));
    for (uint32_t g = 0; g < t.n_groups; ++g)
        HIP_TRY(hipMemsetAsync(t.d_S.p + (g + 1) * t.group_stride - t.rw, 0xff, t.rw * 4, nullptr));
    pfq::launch_transpose(t.d_bits.p, t.n_words, t.d_col_row.p, t.n_cols, t.d_S.p, t.rw, t.group_stride, t.group_log2, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (want_coarse) PFQ_TRY(build_coarse(t, plan));
    t.have_cand_hint = false;
    t.tables_valid = false;
    t.layout_valid = true;
    return PFQ_OK;
}

int ensure_scratch(pfq_tree &t, uint64_t n_reads, bool want_hits) {
    HIP_TRY(t.d_stats.ensure(pfq::ST_N));
    HIP_TRY(t.d_cursors.ensure(16));
    // two hits per read, or 1.3 x what recent blocks reported (a block that overflows is run again, see query_device)
    const uint64_t cap = (uint64_t)(std::max(2.0, 1.3 * t.hits_per_read) * (double)n_reads) + 1024;
    if (want_hits) {
        HIP_TRY(t.d_hit_pairs.ensure(cap));
        HIP_TRY(t.d_allhit.ensure(n_reads + 1));
        HIP_TRY(t.d_counts_snapshot.ensure(t.leaves.size() + 1));
    }
    return PFQ_OK;
}
constexpr uint64_t CLASSIFY_MAX_BLOCKS = 4096;  // blocks of 4 waves; every wave may leave one reservation partly used

// Scratch of the bucketed path.  false: not enough device memory, the caller stays on the direct kernel.
bool ensure_bucket_scratch(pfq_tree &t, uint64_t n_reads, bool with_guards, uint64_t launch_waves) {
    if (!t.h_pair_cursor) {
        if (hipHostMalloc((void **)&t.h_pair_cursor, 64, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            t.h_pair_cursor = nullptr;
            return false;
        }
        for (int i = 0; i < 8; ++i) t.h_pair_cursor[i] = 0;
        if (hipEventCreateWithFlags(&t.hint_ev, hipEventDisableTiming) != hipSuccess) return false;
    } else if (t.hint_reads && hipEventQuery(t.hint_ev) == hipSuccess) {
        t.pairs_per_read = std::max(t.pairs_per_read, (double)t.h_pair_cursor[0] / (double)t.hint_reads);
        if (t.hint_entry_cap) t.passes_hint = std::max<uint64_t>(1, (t.h_pair_cursor[1] + t.hint_entry_cap - 1) / t.hint_entry_cap);
        if (t.hint_counts) t.dirty_frac = (double)t.h_pair_cursor[2] / (double)std::max<unsigned long long>(1, t.h_pair_cursor[3] & 0xffffffffull);
        t.cand_per_read = (double)t.h_pair_cursor[4] / (double)t.hint_reads;
        t.have_cand_hint = true;
        t.hint_reads = 0;
    }
    (void)hipGetLastError();  // hipEventQuery reports "not ready" through the error state
    // room for two candidates per read, or for 1.3 x what recent calls deferred (at most 24 per read: 40 B per slot);
    // + one partially used reservation per wave.  Pairs that do not fit are certified inline (exact, slow).
    const double per_read = std::min(24.0, std::max(2.0, 1.3 * t.pairs_per_read));
    uint64_t cap = ((uint64_t)(per_read * (double)n_reads) + pfq::PAIR_RESERVE * launch_waves + 1024 + 31) & ~31ull;
    // pairs deferred by k_classify: slots [0, leaf_cap); guard pairs (k_expand_guards): [leaf_cap, leaf_cap + guard_cap),
    // sized by the tree's guards per leaf (what does not fit is certified inline there)
    if (t.d_pairs.n && t.leaf_cap >= cap) cap = t.leaf_cap;  // (the buffers only grow)
    t.leaf_cap = cap;
    t.guard_cap = 0;
    if (with_guards) {
        const double per_leaf = (double)t.guard_col.size() / (double)std::max<size_t>(1, t.leaves.size());
        t.guard_cap = ((uint64_t)((double)cap * std::min(4.0, std::max(0.25, per_leaf))) + 32 * 4 * 2048 + 31) & ~31ull;
        cap += t.guard_cap;
    }
    bool ok = soft_ensure(t.d_pairs, cap) && soft_ensure(t.d_sorted, cap) && soft_ensure(t.d_fail, cap) &&
              soft_ensure(t.d_bucket, 3 * ((size_t)t.n_cols << 6) + 2) &&  // up to 64 sub-buckets per column
              soft_ensure(t.d_queue, 128);
    if (ok && with_guards)
        ok = soft_ensure(t.d_owner, t.d_pairs.n) && soft_ensure(t.d_owner_sorted, t.d_pairs.n) && soft_ensure(t.d_gfail, t.d_pairs.n);
    return ok;
}

constexpr int PROF_EV = 7;  // start, classify, bucket, plan+bin, test, verify, finalize
constexpr uint64_t BUCKET_MIN_READS = 1ull << 18;  // below this the bucketed pass cannot amortise warming the L2 slices
constexpr uint64_t SLICE_TARGET_BYTES = 2560ull << 10;

// One pfq_query_batch[_device] call.  plan() chooses the path and the modes from the tree, the threshold, the knobs, the room
// the device has left and — block mode, on a tree without history — a screened sample of the block's own reads; attempt() is
// one run of the kernels (a second one only when the hit buffer proved too small), stage by stage: frontier(), then on the
// bucketed path setup_pairs() -> frontier(true) -> expand guards / tail records -> bucket_sort() -> setup_verify() ->
// tile_stage() -> finish_blocks() or finish_pairs(); read_hits() brings the per-read hit lists back.  Everything is queued on
// `st`; results never depend on which modes were chosen.
struct QueryRun {
    pfq_tree &t;
    const uint8_t *d_seq;
    const uint64_t *d_off;
    uint64_t n_reads, total_bytes;
    float threshold;
    uint32_t flags;
    hipStream_t st;
    pfq_hits *hits;
    const Knobs &kn;
    // ---- the plan
    bool want_hits = false, with_guards = false, thr_one = false, thr_frac = false, counts_mode = false;
    bool recs_possible = false, bucketed = false, block_mode = false, want_two_level = false;
    size_t nl = 0, nc = 0, guarded = 0, nb = 0, mem_free = 0, mem_total = 0;
    uint32_t group_cols = 0, leaf_groups = 1, n_tiles_block = 0, sub_log2 = 0;
    int blocks = 0, blocks_group = 0;
    uint64_t launch_waves = 0, n_blocks = 0, miss_cap = 0, hit_cap = 0;
    // ---- one attempt
    pfq::QueryArgs a{};
    hipEvent_t *ev = nullptr;
    uint32_t *cnt = nullptr, *off = nullptr, *cur = nullptr, *cntw = nullptr, *offw = nullptr, *curw = nullptr;  // bucket histograms / offsets / cursors
    uint4 *recs = nullptr;
    uint32_t n_slices = 1;
    pfq::GuardArgs ga{};
    pfq::VerifyArgs v{};
    int vblocks = 512, vthreads = 512;

    QueryRun(pfq_tree &t_, const uint8_t *seq_, const uint64_t *off_, uint64_t n_, uint64_t bytes_, float thr_, uint32_t flags_, hipStream_t st_,
             pfq_hits *hits_)
        : t(t_), d_seq(seq_), d_off(off_), n_reads(n_), total_bytes(bytes_), threshold(thr_), flags(flags_), st(st_), hits(hits_), kn(t_.knobs) {}

    int plan() {
        if (t.root < 0) return fail(PFQ_ERR_STATE, "query on an empty tree");
        PFQ_TRY(build_layout(t));
        want_hits = (flags & PFQ_WANT_HITS) != 0;
        if (want_hits && !hits) return fail(PFQ_ERR_ARG, "PFQ_WANT_HITS set but hits == NULL");
        if (n_reads >= (1ull << 31) - 1024) return fail(PFQ_ERR_ARG, "more than 2^31 reads in one block");
        // the scratch buffers are reused call after call: calls on one stream are ordered by it, a change of stream waits
        if (t.have_last_stream && t.last_stream != st) HIP_TRY(hipStreamSynchronize(t.last_stream));
        t.last_stream = st;
        t.have_last_stream = true;
        t.last_n_reads = n_reads;
        PFQ_TRY(ensure_scratch(t, n_reads, want_hits));
        nl = t.leaves.size();
        nc = t.n_cols;  // leaf + guard columns = buckets of the bucketed path (block mode: blocks of 8 leaf columns)
        with_guards = !t.guard_col.empty();
        // bucketed path: threshold 1 (any certificate kernel), or 0 < threshold < 1 with probe records (per-pair k-mer miss bits)
        thr_one = threshold == 1.0f;
        thr_frac = threshold > 0.0f && threshold < 1.0f;
        // budgets of the two large scratch buffers: what the device can still give (plus what the tree already holds of it),
        // at most 64 GB each, unless a knob says otherwise; a failed allocation degrades to the next exact path below
        if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) { (void)hipGetLastError(); mem_free = 0; }
        uint64_t rec_budget = std::min<uint64_t>(64ull << 30, (uint64_t)((double)(mem_free + t.d_recs.bytes()) * 0.45));
        if (kn.record_gb >= 0) rec_budget = (uint64_t)kn.record_gb << 30;
        recs_possible = total_bytes && t.nbits < (1ull << 30) && t.num_hashes <= 35 && total_bytes * 16 <= rec_budget;
        // (or as many bases as 2^18 reads of 150 bp: long reads bring the same certificate work with fewer reads)
        bucketed = (t.force_path == 1) || (t.force_path < 0 && (n_reads >= BUCKET_MIN_READS || total_bytes >= BUCKET_MIN_READS * 150));
        if (!(thr_one || thr_frac) || nl == 0 || n_reads == 0) bucketed = false;
        // probe records: hash survivors once instead of once per slice (needs d < 2^30, <= 35 hashes, room)
        if (bucketed && recs_possible && !soft_ensure(t.d_recs, total_bytes + 64)) recs_possible = false;
        if (bucketed && thr_frac && !recs_possible) bucketed = false;  // the miss bits of thresholds < 1 come from the records
        // the classify launches of this call: one per group of leaf columns (two for thresholds < 1: reads of >= 256 k-mers), on
        // all reads, or — two-level frontier — on the reads the coarse launch lists for the group
        group_cols = 1u << t.group_log2;
        leaf_groups = (uint32_t)std::max<size_t>(1, (nl + group_cols - 1) / group_cols);
        blocks = (int)std::min<uint64_t>((n_reads + 3) / 4, CLASSIFY_MAX_BLOCKS);  // (2048: 10.1 ms, 4096: 9.9 ms per step)
        want_two_level = t.coarse_valid && leaf_groups > 1 && (kn.coarse > 0 || threshold >= 1.0f || t.coarse_fill <= 0.70);
        blocks_group = want_two_level ? std::min(blocks, 1024) : blocks;  // (a list holds a fraction of the reads)
        launch_waves = 4ull * (uint64_t)blocks_group * leaf_groups * (threshold >= 1.0f ? 1 : 2);
        if (bucketed && !ensure_bucket_scratch(t, n_reads, with_guards, launch_waves)) bucketed = false;
        if (bucketed && recs_possible && !soft_ensure(t.d_meta, t.d_pairs.n)) recs_possible = false;
        if (bucketed && thr_frac && !recs_possible) bucketed = false;
        counts_mode = !(threshold >= 1.0f);  // theta >= 1: need >= n for every read with k-mers
        // Block mode (pfq::TILE_LOG2_BLOCK): reads that pass several leaves of a block of 8 (strains of one phage) are certified
        // once per block.  Threshold 1, no guard columns, records and room for the tables; chosen when recent calls saw more
        // than 1.5 candidate leaves per read (PFQ_BLOCK=1 / 0 forces / forbids it).  Results do not depend on the choice.
        n_blocks = (nl + 7) / 8;
        n_tiles_block = (uint32_t)((t.n_words * 64 + (1ull << pfq::TILE_LOG2_BLOCK) - 1) >> pfq::TILE_LOG2_BLOCK);
        // (Guard columns — reference-built trees whose internal names collide: the guards of a hit are certified against the
        // sliced matrix afterwards, 1300 line gathers each, which pays while few leaves have guards: <= 5 % of them.)
        guarded = 0;
        for (size_t c = 0; with_guards && c < nl; ++c) guarded += t.guard_off[c + 1] > t.guard_off[c];
        // Candidate leaves per read decide between the pair pipeline and block mode.  Later calls take the figure of the call
        // before; a call without that history (the first on a tree, the first after its layout changed) screens a sample of its
        // OWN reads first — the frontier only, nothing is certified or counted — so that a workload of related genomes does not
        // run its first block through the pair pipeline.  One small launch per column group and one read-back.
        const bool block_eligible = bucketed && (thr_one || (thr_frac && kn.tile_counts != 0)) && recs_possible && n_tiles_block <= 560 &&
                                    n_blocks < (1u << 16) && (kn.tile < 0 || kn.tile != 0);
        if (block_eligible && kn.block < 0 && !t.have_cand_hint && guarded * 20 <= nl) PFQ_TRY(sample_candidates());
        // At thresholds below 1 block mode keeps the k-mer entries: buckets by (block, candidate mask), 8 miss bytes per k-mer.
        block_mode = block_eligible && (kn.block >= 0 ? kn.block != 0 : (t.cand_per_read > 1.5 && guarded * 20 <= nl));
        if (block_mode && !soft_ensure(t.d_T, n_blocks * t.n_words * 64)) block_mode = false;
        if (block_mode && !soft_ensure(t.d_failb, t.d_pairs.n * 8)) block_mode = false;
        if (block_mode) nc = n_blocks * 256;  // buckets by (block, candidate mask)
        t.last_block_mode = 0;
        miss_cap = 0;
        nb = 0;
        sub_log2 = 0;
        if (bucketed) {
            // sub-buckets (keyed by the read index) keep every histogram counter cold when columns are few
            while (sub_log2 < 6 && (nc << sub_log2) < 1024) ++sub_log2;
            nb = nc << sub_log2;
            if (block_mode && !soft_ensure(t.d_bucket, 3 * nb + 2)) bucketed = false;  // (buckets by (block, mask) outnumber the columns of small trees)
            if (counts_mode && !block_mode) {  // thresholds < 1: every deferred pair owns ceil(n/64) words of k-mer miss bits that the slices OR into
                const uint64_t avg_len = n_reads ? total_bytes / n_reads : 0;
                miss_cap = std::min<uint64_t>((t.leaf_cap + t.guard_cap) * ((avg_len >> 6) + 2) + (launch_waves + 4 * 2048) * (uint64_t)pfq::MISS_RESERVE, 0xfffffff0ull);
                if (!(soft_ensure(t.d_miss_words, miss_cap) && soft_ensure(t.d_miss_pos, t.d_pairs.n) && soft_ensure(t.d_bucket_w, 3 * nb + 2)))
                    bucketed = false;
            }
        }
        t.last_path = bucketed ? 1 : 0;
        hit_cap = t.d_hit_pairs.n;
        return PFQ_OK;
    }

    // Candidates per read of the first reads of this block: the frontier only, nothing is certified or counted.
    int sample_candidates() {
        const uint64_t n_s = std::min<uint64_t>(n_reads, 16384);
        HIP_TRY(hipMemsetAsync(t.d_stats.p, 0, pfq::ST_N * 8, st));
        HIP_TRY(hipMemsetAsync(t.d_cursors.p, 0, 128, st));
        pfq::QueryArgs sa{};
        sa.hp = t.hp;
        sa.seq = d_seq;
        sa.off = d_off;
        sa.n_reads = n_s;
        sa.threshold = threshold;
        sa.S_all = t.d_S.p;
        sa.group_stride = t.group_stride;
        sa.group_log2 = t.group_log2;
        sa.ones_row = (uint32_t)(t.n_words * 64);
        sa.rw = t.rw;
        sa.rw_log2 = t.rw_log2;
        sa.n_cols = t.n_cols;
        sa.guard_off = t.d_guard_off.p;
        sa.guard_col = t.d_guard_col.p;
        sa.counts = t.d_counts.p;
        sa.hit_cursor = t.d_cursors.p;
        sa.stats = t.d_stats.p;
        sa.screen_only = 1;
        if (!(threshold >= 1.0f)) {
            HIP_TRY(t.d_long.ensure(n_reads + 1));
            sa.long_list = t.d_long.p;
            sa.n_long = reinterpret_cast<unsigned int *>(t.d_cursors.p + 4);
        }
        for (uint32_t g = 0; g < leaf_groups; ++g) {
            sa.S = t.d_S.p + (uint64_t)g * t.group_stride;
            sa.col0 = g * group_cols;
            sa.n_leaves = (uint32_t)std::min<size_t>(group_cols, nl - (size_t)g * group_cols);
            sa.first_group = g == 0;
            if (g && !(threshold >= 1.0f)) HIP_TRY(hipMemsetAsync(t.d_cursors.p + 4, 0, 8, st));
            pfq::launch_classify(sa, false, !(threshold >= 1.0f), (int)std::min<uint64_t>((n_s + 3) / 4, CLASSIFY_MAX_BLOCKS), st);
        }
        HIP_TRY(hipGetLastError());
        unsigned long long cand = 0;
        HIP_TRY(hipMemcpyAsync(&cand, t.d_stats.p + pfq::ST_CANDIDATES, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        t.cand_per_read = (double)cand / (double)n_s;
        t.have_cand_hint = true;
        return PFQ_OK;
    }

    // the frontier kernels, once per column group that holds leaves (a tree of up to 2048 columns has one group)
    int frontier(bool defer) {
        // Two-level frontier: the coarse launch screens every read against an antichain of internal nodes and lists
        // it for the leaf groups below its live columns; a group's launch then sees only its list.
        // (thresholds < 1: only while the coarse filters are empty enough for <= 4 probes per k-mer to tell a miss)
        bool two_level = want_two_level;
        uint32_t list_cap = 0;
        if (two_level) {
            // every read at most once per list + one partly used reservation of 32 per wave of the (two) coarse launches
            list_cap = (uint32_t)((n_reads + 2 * 32ull * 4 * (uint64_t)blocks + 63) & ~31ull);
            if (!soft_ensure(t.d_glists, (size_t)list_cap * leaf_groups)) two_level = false;
            if (two_level && counts_mode && !soft_ensure(t.d_glong, (size_t)list_cap * leaf_groups)) two_level = false;
        }
        t.last_leaf_groups = leaf_groups;
        t.last_coarse_cols = t.last_coarse_probes = 0;
        if (two_level) {
            pfq::QueryArgs ac = a;
            ac.S = ac.S_all = t.d_Sc.p;
            ac.group_stride = 0;
            ac.group_log2 = 11;
            ac.col0 = 0;
            ac.n_leaves = ac.n_cols = t.coarse_cols;
            ac.rw = t.coarse_rw;
            ac.rw_log2 = t.coarse_rw_log2;
            ac.first_group = 1;
            pfq::CoarseArgs ca{};
            ca.cgrp = t.d_cgrp.p;
            ca.n_groups = leaf_groups;
            ca.lists = t.d_glists.p;
            ca.cursors = t.d_gcur.p;
            ca.list_cap = list_cap;
            ca.total_leaves = (uint32_t)nl;
            // probes per k-mer: enough for a foreign read to lose every coarse column.  Threshold 1 (AND over 4 k-mers):
            // columns x fill^(4 p) <= 0.02; below 1: a k-mer must be a miss with probability >= 0.85, 1 - fill^p
            const double f = std::min(0.999, std::max(1e-6, t.coarse_fill));
            uint32_t np;
            if (!counts_mode) {
                np = (uint32_t)std::ceil(std::log(0.02 / t.coarse_cols) / (4.0 * std::log(f)));
                np = std::min<uint32_t>(std::max<uint32_t>(np, 2), pfq::COARSE_MAX_PROBES);
            } else {
                np = 1;
                while (np < 4 && 1.0 - std::pow(f, (double)np) < 0.85) ++np;
                // k-mers looked at beyond maxmiss + 1 + 8, per 256 of maxmiss + 1: what the k-mers that are no misses cost
                const double pm = 1.0 - std::pow(f, (double)np);
                ca.scr_extra = (uint32_t)std::min(256.0, std::ceil(256.0 * (1.0 / pm - 1.0)));
            }
            if (kn.coarse_probes > 0) np = (uint32_t)std::min<long long>(kn.coarse_probes, counts_mode ? 4 : pfq::COARSE_MAX_PROBES);
            np = std::max<uint32_t>(1, std::min<uint32_t>(np, t.num_hashes));
            ca.n_probes = np;
            t.last_coarse_cols = t.coarse_cols;
            t.last_coarse_probes = np;
            HIP_TRY(hipMemsetAsync(t.d_gcur.p, 0, 2 * pfq::MAX_LEAF_GROUPS * sizeof(unsigned int), st));  // lists' cursors, long reads' cursors
            // (1024 blocks: every wave of the coarse launch may leave a reservation of 32 slots partly used in every group's list —
            // with 4096 blocks and ten groups the lists were half unused slots, and the leaf groups' dense screens half idle)
            int coarse_blocks = std::min(blocks, 1024);
            pfq::launch_coarse(ac, ca, counts_mode, coarse_blocks, st);
        }
        if (two_level) {
            // ONE launch for all leaf groups (blockIdx.y = the group: its matrix, columns and list follow from it); the
            // queues of reads of >= 256 k-mers (thresholds < 1) are per group as well, behind the groups' cursors
            a.S = t.d_S.p;
            a.col0 = 0;
            a.n_leaves = (uint32_t)std::min<size_t>(group_cols, nl);
            a.first_group = 0;
            a.read_list = t.d_glists.p;
            a.n_list = t.d_gcur.p;
            a.grid_groups = leaf_groups;
            a.total_leaves = (uint32_t)nl;
            a.list_cap = list_cap;
            if (counts_mode) {
                a.long_list = t.d_glong.p;
                a.n_long = t.d_gcur.p + pfq::MAX_LEAF_GROUPS;
            }
            pfq::launch_classify(a, defer, counts_mode, blocks_group, st);
            a.read_list = nullptr;
            a.n_list = nullptr;
            a.grid_groups = 0;
            if (counts_mode) {
                a.long_list = t.d_long.p;
                a.n_long = reinterpret_cast<unsigned int *>(t.d_cursors.p + 4);
            }
            return PFQ_OK;
        }
        for (uint32_t g = 0; g < leaf_groups; ++g) {
            a.S = t.d_S.p + (uint64_t)g * t.group_stride;
            a.col0 = g * group_cols;
            a.n_leaves = (uint32_t)std::min<size_t>(group_cols, nl - (size_t)g * group_cols);
            a.first_group = g == 0;
            if (g && counts_mode) HIP_TRY(hipMemsetAsync(t.d_cursors.p + 4, 0, 8, st));  // the queue of long reads is per launch
            pfq::launch_classify(a, defer, counts_mode, blocks, st);
        }
        return PFQ_OK;
    }

    int attempt(int attempt_no) {
        HIP_TRY(hipMemsetAsync(t.d_stats.p, 0, pfq::ST_N * 8, st));
        HIP_TRY(hipMemsetAsync(t.d_cursors.p, 0, 128, st));
        if (want_hits) {
            HIP_TRY(hipMemsetAsync(t.d_allhit.p, 0, n_reads + 1, st));
            if (attempt_no == 0 && nl)
                HIP_TRY(hipMemcpyAsync(t.d_counts_snapshot.p, t.d_counts.p, nl * 8, hipMemcpyDeviceToDevice, st));
        }
        if (n_reads && nl) {
            a = pfq::QueryArgs{};
            a.hp = t.hp;
            a.seq = d_seq;
            a.off = d_off;
            a.n_reads = n_reads;
            a.threshold = threshold;
            a.S_all = t.d_S.p;
            a.group_stride = t.group_stride;
            a.group_log2 = t.group_log2;
            a.ones_row = (uint32_t)(t.n_words * 64);
            a.rw = t.rw;
            a.rw_log2 = t.rw_log2;
            a.n_cols = t.n_cols;
            a.guard_off = t.d_guard_off.p;
            a.guard_col = t.d_guard_col.p;
            a.counts = t.d_counts.p;
            a.hit_pairs = want_hits ? t.d_hit_pairs.p : nullptr;
            a.hit_cap = hit_cap;
            a.hit_cursor = t.d_cursors.p;
            a.allhit_flag = want_hits ? t.d_allhit.p : nullptr;
            a.stats = t.d_stats.p;
            if (counts_mode) {
                HIP_TRY(t.d_long.ensure(n_reads + 1));
                a.long_list = t.d_long.p;
                a.n_long = reinterpret_cast<unsigned int *>(t.d_cursors.p + 4);
            }
            ev = nullptr;
            if (t.prof_used < t.prof_cap) {
                ev = &t.prof_ev[PROF_EV * t.prof_used];
                t.prof_bucketed[t.prof_used] = bucketed;
                ++t.prof_used;
            }
            if (ev) HIP_TRY(hipEventRecord(ev[0], st));
            if (bucketed) {
                PFQ_TRY(setup_pairs());
                PFQ_TRY(frontier(true));
                PFQ_TRY(guards_and_tails());
                PFQ_TRY(bucket_sort());
                PFQ_TRY(setup_verify());
                PFQ_TRY(tile_stage());
                if (block_mode) PFQ_TRY(finish_blocks());
                else PFQ_TRY(finish_pairs());
            } else {
                PFQ_TRY(frontier(false));
                if (ev) HIP_TRY(hipEventRecord(ev[1], st));
            }
            HIP_TRY(hipGetLastError());
        }
        if (bucketed) {  // how many pair slots this call used, for the next call's sizing
            HIP_TRY(hipMemcpyAsync(t.h_pair_cursor, t.d_cursors.p + 1, 16, hipMemcpyDeviceToHost, st));  // pair cursor, bucket cursor
            HIP_TRY(hipMemcpyAsync(t.h_pair_cursor + 2, t.d_cursors.p + 6, 8, hipMemcpyDeviceToHost, st));  // pairs with a k-mer missing
            HIP_TRY(hipMemcpyAsync(t.h_pair_cursor + 4, t.d_stats.p + pfq::ST_CANDIDATES, 8, hipMemcpyDeviceToHost, st));  // candidate leaves
            if (n_reads && nl)  // ... of how many sorted pairs (the pair cursor also counts partly used reservations)
                HIP_TRY(hipMemcpyAsync(t.h_pair_cursor + 3, t.d_bucket.p + 2 * (nc << t.last_sub_log2), 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipEventRecord(t.hint_ev, st));
            t.hint_reads = n_reads;
        }
        return PFQ_OK;
    }

    // deferred-pair buffer, bucket histograms, probe records, miss words; block tables on first use
    int setup_pairs() {
        t.last_sub_log2 = sub_log2;
        cnt = t.d_bucket.p;
        off = cnt + nb;
        cur = off + nb + 1;
        a.pairs = t.d_pairs.p;
        a.pair_cap = t.leaf_cap;  // whole reservations only (PAIR_CHUNK = 32)
        a.pair_cursor = t.d_cursors.p + 1;
        a.bucket_cnt = cnt;
        a.sub_log2 = sub_log2;
        recs = recs_possible ? t.d_recs.p : nullptr;
        a.recs = recs;
        a.rec_cap = recs ? t.d_recs.n : 0;
        cntw = offw = curw = nullptr;
        if (counts_mode && !block_mode) {
            cntw = t.d_bucket_w.p;
            offw = cntw + nb;
            curw = offw + nb + 1;
            HIP_TRY(hipMemsetAsync(t.d_miss_words.p, 0, miss_cap * 8, st));
            HIP_TRY(hipMemsetAsync(cntw, 0, nb * 4, st));
            a.bucket_words = cntw;
            a.miss_cursor = t.d_cursors.p + 5;
            a.miss_cap = miss_cap;
        }
        n_slices = 1;
        uint64_t slice_target = SLICE_TARGET_BYTES;
        if (kn.slice_kb > 0) slice_target = (uint64_t)kn.slice_kb << 10;
        while (n_slices < 8 && (t.n_words * 8 + n_slices - 1) / n_slices > slice_target) n_slices <<= 1;
        t.last_slices = n_slices;
        HIP_TRY(hipMemsetAsync(cnt, 0, nb * 4, st));
        HIP_TRY(hipMemsetAsync(t.d_fail.p, 0, t.d_fail.n * 4, st));
        if (with_guards) HIP_TRY(hipMemsetAsync(t.d_gfail.p, 0, t.d_gfail.n * 4, st));
        HIP_TRY(hipMemsetAsync(t.d_queue.p, 0, 128 * 4, st));
        // last windows of few k-mers are hashed several reads per pass afterwards: of up to 16 k-mers, or of up to 32 when the
        // reads' (average) length makes such tails — 100 bp reads at k = 20 have 64 + 17 k-mers
        a.batch_tails = 0;
        if (recs && !counts_mode && kn.no_tail_batch <= 0) {
            const uint64_t avg_len = n_reads ? total_bytes / n_reads : 0;
            const uint64_t tl = avg_len >= t.kmer_size ? ((avg_len - t.kmer_size + 1) & 63u) : 0;
            a.batch_tails = (tl > 16 && tl <= 32) ? 32u : 16u;
        }
        a.block_pairs = block_mode ? 1u : 0u;
        a.screen_recs = kn.screen_recs >= 0 ? (uint32_t)(kn.screen_recs != 0) : 1u;
        if (block_mode) {
            if (!t.tables_valid) {  // (the leaf set changed, or first use)
                pfq::launch_block_tables(t.d_bits.p, t.n_words, t.d_col_row.p, (uint32_t)nl, t.d_T.p, st);
                t.tables_valid = true;
            }
            HIP_TRY(hipMemsetAsync(t.d_failb.p, 0, t.d_failb.n, st));
        }
        return PFQ_OK;
    }
    // guard columns of the deferred pairs (pairs of their own), the records of the last windows
    int guards_and_tails() {
        ga = pfq::GuardArgs{};
        if (with_guards && !block_mode) {  // every guard of a deferred pair's leaf becomes a pair of its own (second region of the buffer)
            ga.pairs = t.d_pairs.p + t.leaf_cap;
            ga.cap = t.guard_cap;
            ga.cursor = t.d_cursors.p + 8;
            ga.slot0 = (uint32_t)t.leaf_cap;
            ga.owner = t.d_owner.p;
            ga.gfail = t.d_gfail.p;
            pfq::launch_expand_guards(a, ga, 2048, st);
        }
        if (a.batch_tails) pfq::launch_tail_records(a, 2048, st);
        if (ev) HIP_TRY(hipEventRecord(ev[1], st));
        return PFQ_OK;
    }
    // counting sort of the pairs by column (block mode: by (block, candidate mask))
    int bucket_sort() {
        pfq::launch_bucket_scan(cnt, off, cur, (uint32_t)nb, st);
        if (counts_mode && !block_mode) pfq::launch_bucket_scan(cntw, offw, curw, (uint32_t)nb, st);
        pfq::launch_bucket_scatter(t.d_pairs.p, t.d_cursors.p + 1, a.pair_cap, off, cur, sub_log2, t.d_sorted.p,
                                   recs ? t.d_meta.p : nullptr, d_off, block_mode ? nullptr : t.d_col_row.p, offw, curw,
                                   (counts_mode && !block_mode) ? t.d_miss_pos.p : nullptr, (uint32_t)t.kmer_size,
                                   (with_guards && !block_mode) ? t.d_owner.p : nullptr,
                                   (with_guards && !block_mode) ? t.d_owner_sorted.p : nullptr,
                                   block_mode ? 2u : 0u, 1024, st);
        if (with_guards && !block_mode)
            pfq::launch_bucket_scatter(ga.pairs, ga.cursor, ga.cap, off, cur, sub_log2, t.d_sorted.p,
                                       recs ? t.d_meta.p : nullptr, d_off, t.d_col_row.p, offw, curw,
                                       counts_mode ? t.d_miss_pos.p : nullptr, (uint32_t)t.kmer_size,
                                       t.d_owner.p + t.leaf_cap, t.d_owner_sorted.p, 0u, 256, st);
        if (ev) HIP_TRY(hipEventRecord(ev[2], st));
        return PFQ_OK;
    }
    int setup_verify() {
        v = pfq::VerifyArgs{};
        v.hp = t.hp;
        v.seq = d_seq;
        v.off = d_off;
        v.bits = t.d_bits.p;
        v.col_row = t.d_col_row.p;
        v.n_words = t.n_words;
        v.sorted = t.d_sorted.p;
        v.n_pairs_ptr = off + nb;
        v.fail = t.d_fail.p;
        v.recs = recs;
        v.miss_words = (counts_mode && !block_mode) ? t.d_miss_words.p : nullptr;
        v.miss_pos = (counts_mode && !block_mode) ? t.d_miss_pos.p : nullptr;
        v.meta = t.d_meta.p;
        v.n_slices = n_slices;
        uint64_t sb = (t.n_words * 64 + n_slices - 1) / n_slices;
        v.slice_bits = (uint32_t)((sb + 63) & ~63ull);
        v.queue = t.d_queue.p;
        // window of pairs in flight per slice = (blocks/8)*(8/n_slices)*4*chunk: about one leaf bucket
        vblocks = 512;
        v.chunk = 1;
        if (kn.verify_blocks >= 0) vblocks = std::max(8, (int)kn.verify_blocks & ~7);
        if (kn.verify_chunk >= 0) v.chunk = (uint32_t)std::max(1, (int)kn.verify_chunk);
        v.n_sub = 8;
        if (kn.verify_sub >= 0) v.n_sub = (uint32_t)std::min(16, std::max(1, (int)kn.verify_sub));
        vthreads = 512;
        if (kn.verify_threads >= 0) vthreads = std::min(1024, std::max(64, (int)kn.verify_threads & ~63));
        if (!recs) { vthreads = 256; vblocks = 1024; v.chunk = 4; }  // re-hash fallback kernel: 4-wave blocks
        return PFQ_OK;
    }
    int tile_stage() {
        // LDS-tile certificates: every probe binned by (leaf chunk, 128 KiB filter tile), tiles tested out of LDS;
        // k_verify_rec then only sees the pairs that could not be binned
        // (thresholds < 1: entries name k-mers, tiles are half the size — pfq::TILE_LOG2_COUNTS)
        const uint32_t tile_log2 = block_mode ? pfq::TILE_LOG2_BLOCK : (counts_mode ? pfq::TILE_LOG2_COUNTS : pfq::TILE_LOG2);
        const uint32_t n_tiles = (uint32_t)((t.n_words * 64 + (1ull << tile_log2) - 1) >> tile_log2);
        const uint32_t chunk_log2 = pfq::CHUNK_PAIRS_LOG2;
        // Thresholds < 1: the tile passes leave the k-mers that are not contained in per-chunk miss bitmaps; k_verify_rec
        // only sees what could not be binned.  (PFQ_TILE_COUNTS=0: record kernel only.)  Results do not depend on the choice.
        bool tile_counts = true;
        if (kn.tile_counts >= 0) tile_counts = kn.tile_counts != 0;
        bool tile_mode = recs && (!counts_mode || tile_counts) && (block_mode || n_tiles < 256);
        if (kn.tile >= 0) tile_mode = tile_mode && kn.tile != 0;
        if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) { (void)hipGetLastError(); mem_free = 0; }
        uint64_t tile_budget = std::min<uint64_t>(64ull << 30, (uint64_t)((double)(mem_free + t.d_entries.bytes()) * 0.8));
        if (kn.tile_gb >= 0) tile_budget = (uint64_t)kn.tile_gb << 30;
        t.last_tile_mode = 0;
        if (tile_mode) {
            // every read may survive with one candidate: (bases - (k-1) per read) * hashes * 1.125 + slack per bucket
            const uint64_t max_chunks = nc + ((t.leaf_cap + t.guard_cap) >> chunk_log2) + 2;
            uint64_t want = (uint64_t)((double)total_bytes * t.num_hashes * 1.13 * std::max(1.0, t.pairs_per_read)) +
                            max_chunks * n_tiles * 544ull;
            if (want * 4 > tile_budget) want = tile_budget / 4;
            if (kn.tile_entries >= 0) want = std::max<uint64_t>(1, (uint64_t)kn.tile_entries);  // tests: force passes
            // thresholds < 1: one miss byte per k-mer of every pair the recent calls make expect (chunks that find no room
            // take the fallback), the rounds' positions, the pairs' positions
            // (block mode: 8 bytes per k-mer, one per leaf of the block; chunk offsets are in 16-byte units)
            const uint64_t kmiss_cap = counts_mode ? std::min<uint64_t>(((uint64_t)((double)total_bytes * std::max(1.0, 1.3 * t.pairs_per_read)) + 16 * max_chunks + 64) * (block_mode ? 8u : 1u),
                                                                            block_mode ? (48ull << 30) : 0xfffffff0ull) & ~15ull : 0;
            bool ok = soft_ensure(t.d_entries, want) && soft_ensure(t.d_pair_chunk, t.d_pairs.n) && soft_ensure(t.d_flag_list, t.d_pairs.n) &&
                      soft_ensure(t.d_leaf_chunk0, nc + 1) && soft_ensure(t.d_chunks, max_chunks) && soft_ensure(t.d_gfill, max_chunks * n_tiles) && soft_ensure(t.d_binq, 256);
            if (ok && counts_mode)
                ok = soft_ensure(t.d_kmiss, kmiss_cap) && soft_ensure(t.d_round_k0, max_chunks * pfq::MAX_ROUNDS) &&
                     soft_ensure(t.d_n_rounds, max_chunks) && soft_ensure(t.d_pair_kpos, t.d_pairs.n);
            if (ok && counts_mode && block_mode) ok = soft_ensure(t.d_kall, (kmiss_cap >> 3) + 64);
            if (!ok) {
                tile_mode = false;  // not enough HBM for the probe buckets: stay with the record kernel
            } else {
                HIP_TRY(hipMemsetAsync(t.d_gfill.p, 0, max_chunks * n_tiles * 4, st));
                HIP_TRY(hipMemsetAsync(t.d_binq.p, 0, 256 * 4, st));
                pfq::TileArgs ta{};
                ta.hp = t.hp;
                ta.bits = t.d_bits.p;
                ta.n_words = t.n_words;
                ta.chunk_log2 = chunk_log2;
                if (block_mode) {  // the "filter" of a bucket is its block's table: n_words * 64 bytes
                    ta.blocks = 1;
                    if (counts_mode) ta.kall = t.d_kall.p;  // (buckets by (block, mask); the passes' columns are the blocks)
                    ta.bits = reinterpret_cast<const uint64_t *>(t.d_T.p);
                    ta.n_words = t.n_words * 8;
                    ta.failb = t.d_failb.p;
                }
                ta.recs = recs;
                ta.meta = t.d_meta.p;
                ta.col_row = t.d_col_row.p;
                ta.bucket_off = off;
                ta.sub_log2 = sub_log2;
                ta.n_leaves = (uint32_t)(block_mode ? n_blocks : nc);  // (columns of the passes: blocks, whatever the buckets)
                ta.n_tiles = n_tiles;
                ta.chunks = t.d_chunks.p;
                ta.max_chunks = (uint32_t)max_chunks;
                ta.leaf_chunk0 = t.d_leaf_chunk0.p;
                ta.pair_chunk = t.d_pair_chunk.p;
                ta.n_chunks = reinterpret_cast<unsigned int *>(t.d_cursors.p + 3);
                ta.n_flagged = reinterpret_cast<unsigned int *>(t.d_cursors.p + 3) + 1;
                ta.flag_list = t.d_flag_list.p;
                ta.flag_cap = (uint32_t)std::min<uint64_t>(t.d_flag_list.n, 0xffffffffu);
                ta.entry_cursor = t.d_cursors.p + 2;
                // (a multiple of 32 entries: buckets then start on 128-byte boundaries, k_tile_test reads them 16 bytes at a time)
                ta.entry_cap = std::min<uint64_t>(want, t.d_entries.n) & ~31ull;  // (the buffer only grows; the budget of this call is `want`)
                ta.entries = t.d_entries.p;
                ta.gfill = t.d_gfill.p;
                ta.fail = t.d_fail.p;
                ta.n_pairs_ptr = off + nb;
                if (counts_mode) {
                    ta.counts = 1;
                    ta.threshold = threshold;
                    ta.kmiss = t.d_kmiss.p;
                    ta.kmiss_cap = kmiss_cap;
                    ta.kmiss_used = t.d_cursors.p + 9;
                    ta.round_k0 = t.d_round_k0.p;
                    ta.n_rounds = t.d_n_rounds.p;
                    ta.pair_kpos = t.d_pair_kpos.p;
                    HIP_TRY(hipMemsetAsync(t.d_n_rounds.p, 0, max_chunks * 4, st));
                }
                int bin_blocks = 512, test_blocks = 512;
                if (kn.bin_blocks >= 0) bin_blocks = std::max(1, (int)kn.bin_blocks);
                if (kn.test_blocks >= 0) test_blocks = std::max(1, (int)kn.test_blocks);
                ta.bin_shape = kn.bin_narrow > 0 ? (uint32_t)kn.bin_narrow : (kn.bin_wide > 0 ? 2u : 0u);
                ta.debug = kn.bin_debug > 0 ? (uint32_t)kn.bin_debug : 0u;
                pfq::launch_tile_plan(ta, st);
                // The probe buckets of all pairs may exceed the buffer (reads that pass many leaves): the plan
                // spreads the chunks over passes that reuse it.  Their number is known on the device only; as
                // many passes as the previous call needed are launched without waiting, and chunks of later
                // passes (if any) are certified by the record kernel below — exact either way.
                uint64_t n_passes = std::min<uint64_t>(std::max<uint64_t>(1, t.passes_hint), 256);  // (more: the record kernel takes the rest)
                if (block_mode) {  // the fallback of block mode is slow: wait for the plan and launch every pass it needs
                    HIP_TRY(hipMemcpyAsync(t.h_pair_cursor + 5, t.d_cursors.p + 2, 8, hipMemcpyDeviceToHost, st));
                    HIP_TRY(hipStreamSynchronize(st));
                    n_passes = ta.entry_cap ? std::min<uint64_t>(std::max<uint64_t>(1, (t.h_pair_cursor[5] + ta.entry_cap - 1) / ta.entry_cap), 256) : 1;
                }
                for (uint64_t p = 0; p < n_passes; ++p) {
                    ta.pass = (uint32_t)p;
                    ta.bin_queue = t.d_binq.p + p;
                    pfq::launch_tile_bin(ta, bin_blocks, st);
                    if (p == 0 && ev) HIP_TRY(hipEventRecord(ev[3], st));
                    pfq::launch_tile_test(ta, test_blocks, st);
                }
                v.pair_chunk = t.d_pair_chunk.p;
                v.chunks = t.d_chunks.p;
                v.entry_cursor = t.d_cursors.p + 2;
                v.entry_cap = ta.entry_cap;
                v.launched_passes = (uint32_t)n_passes;
                if (counts_mode && !block_mode) {  // binned pairs whose prefix of k-mers leaves them undecided go to the record kernel
                    pfq::FinalizeArgs pf{};
                    pf.hp = t.hp;
                    pf.threshold = threshold;
                    pf.fail = t.d_fail.p;
                    pf.kmiss = t.d_kmiss.p;
                    pf.pair_kpos = t.d_pair_kpos.p;
                    pf.pair_chunk = t.d_pair_chunk.p;
                    pf.chunks = t.d_chunks.p;
                    pf.launched_passes = (uint32_t)n_passes;
                    pfq::launch_prefix_open(pf, t.d_meta.p, off + nb, t.d_fail.p, st);
                }
                t.hint_entry_cap = ta.entry_cap;
                t.last_passes = (uint32_t)std::max<uint64_t>(n_passes, 1);
                if (ev) HIP_TRY(hipEventRecord(ev[4], st));
                v.only_flagged = 1;
                v.n_flagged = ta.n_flagged;
                v.flag_list = t.d_flag_list.p;
                v.flag_cap = ta.flag_cap;
                t.last_tile_mode = 1;
            }
        }
        if (ev && !t.last_tile_mode) {
            HIP_TRY(hipEventRecord(ev[3], st));
            HIP_TRY(hipEventRecord(ev[4], st));
        }
        return PFQ_OK;
    }
    int finish_blocks() {
        // what the passes did not bin (overflows, no room, or no passes at all) is certified leaf by leaf against S
        a.S = t.d_S.p;
        a.col0 = 0;
        a.n_leaves = (uint32_t)std::min<size_t>(2048, nl);
        if (counts_mode && t.last_tile_mode) {  // the miss bytes of the binned pairs decide their candidates
            pfq::FinalizeArgs cf{};
            cf.hp = t.hp;
            cf.off = d_off;
            cf.sorted = t.d_sorted.p;
            cf.threshold = threshold;
            cf.fail = t.d_fail.p;
            cf.kmiss = t.d_kmiss.p;
            cf.kall = t.d_kall.p;
            cf.pair_kpos = t.d_pair_kpos.p;
            cf.pair_chunk = t.d_pair_chunk.p;
            cf.chunks = t.d_chunks.p;
            cf.launched_passes = v.launched_passes;
            pfq::launch_block_count(cf, off + nb, t.d_failb.p, st);
        }
        pfq::launch_block_fallback(a, t.d_sorted.p, off + nb, t.d_fail.p, t.d_failb.p, t.d_pair_chunk.p,
                                   t.last_tile_mode ? t.d_chunks.p : nullptr, v.launched_passes,
                                   t.last_tile_mode ? v.n_flagged : nullptr, t.last_tile_mode ? v.flag_list : nullptr, v.flag_cap, st);
        // ancestors that are not provably supersets must pass too (query.rs:119-141): the guards of every candidate
        // that is still standing
        if (with_guards) pfq::launch_block_guards(a, t.d_sorted.p, off + nb, t.d_failb.p, st);
        if (ev) HIP_TRY(hipEventRecord(ev[5], st));
        pfq::FinalizeArgs f{};
        f.hp = t.hp;
        f.off = d_off;
        f.sorted = t.d_sorted.p;
        f.bucket_off = off;
        f.sub_log2 = sub_log2;
        f.threshold = threshold;
        f.counts = t.d_counts.p;
        f.hit_pairs = a.hit_pairs;
        f.hit_cap = hit_cap;
        f.hit_cursor = t.d_cursors.p;
        f.stats = t.d_stats.p;
        f.failb = t.d_failb.p;
        f.c0 = 0;
        f.c1 = (uint32_t)nc;
        pfq::launch_finalize(f, st);
        if (ev) HIP_TRY(hipEventRecord(ev[6], st));
        t.last_block_mode = 1;
        if (t.last_tile_mode) t.last_tile_mode = 2;
        t.hint_counts = false;
        return PFQ_OK;
    }
    int finish_pairs() {
        if (v.only_flagged == 1 && counts_mode) {
            // thresholds < 1: the list becomes every pair the tile passes left open (a k-mer missing, or not binned)
            unsigned int *n_open = reinterpret_cast<unsigned int *>(t.d_cursors.p + 7);
            pfq::launch_collect_open(t.d_fail.p, off + nb, t.d_flag_list.p, v.flag_cap, n_open, st);
            v.n_flagged = n_open;
        }
        pfq::launch_verify(v, vblocks, vthreads, st);
        if (v.only_flagged == 1) {  // many flagged pairs (no room for their probe buckets): walk all pairs in leaf order instead
            v.only_flagged = 2;
            if (counts_mode) v.chunk = 8;  // (the walk pulls an item per 64 pairs, not per 8)
            pfq::launch_verify(v, vblocks, vthreads, st);
        }
        if (ev) HIP_TRY(hipEventRecord(ev[5], st));
        pfq::FinalizeArgs f{};
        f.hp = t.hp;
        f.off = d_off;
        f.sorted = t.d_sorted.p;
        f.bucket_off = off;
        f.sub_log2 = sub_log2;
        f.fail = t.d_fail.p;
        f.miss_words = v.miss_words;
        f.miss_pos = v.miss_pos;
        f.threshold = threshold;
        f.counts = t.d_counts.p;
        f.hit_pairs = a.hit_pairs;
        f.hit_cap = hit_cap;
        f.hit_cursor = t.d_cursors.p;
        f.stats = t.d_stats.p;
        f.n_dirty = t.d_cursors.p + 6;
        if (counts_mode && t.last_tile_mode) {  // miss bits of the binned pairs: in their chunks' bitmaps
            f.kmiss = t.d_kmiss.p;
            f.pair_kpos = t.d_pair_kpos.p;
            f.pair_chunk = t.d_pair_chunk.p;
            f.chunks = t.d_chunks.p;
            f.launched_passes = v.launched_passes;
        }
        f.owner_sorted = with_guards ? t.d_owner_sorted.p : nullptr;
        f.gfail = with_guards ? t.d_gfail.p : nullptr;
        t.hint_counts = counts_mode;
        if (with_guards) {  // the guard columns first: a guard that does not pass marks its leaf pair
            f.c0 = (uint32_t)nl;
            f.c1 = (uint32_t)nc;
            f.guards = 1;
            pfq::launch_finalize(f, st);
        }
        f.c0 = 0;
        f.c1 = (uint32_t)nl;
        f.guards = 0;
        pfq::launch_finalize(f, st);
        if (ev) HIP_TRY(hipEventRecord(ev[6], st));
        return PFQ_OK;
    }

    // PFQ_WANT_HITS: the per-read hit lists of this attempt; done = false: the hit buffer was too small, run again
    int read_hits(bool &done) {
        done = true;
        HIP_TRY(hipStreamSynchronize(st));
        unsigned long long cursors[2] = {0, 0};
        HIP_TRY(hipMemcpy(cursors, t.d_cursors.p, 16, hipMemcpyDeviceToHost));
        if (n_reads) t.hits_per_read = std::max(t.hits_per_read, (double)cursors[0] / (double)n_reads);
        if (cursors[0] <= hit_cap) {
            // CSR read -> leaves (ascending; reads that pass every node list every leaf), built on the device from the
            // unordered hit pairs and copied into page-locked host buffers
            auto host_room = [&](void **p, size_t &cap, size_t want) -> int {
                if (want <= cap) return PFQ_OK;
                if (*p) (void)hipHostFree(*p);
                *p = nullptr;
                cap = 0;
                const size_t grown = want + want / 4 + 4096;
                HIP_TRY(hipHostMalloc(p, grown, hipHostMallocDefault));
                cap = grown;
                return PFQ_OK;
            };
            PFQ_TRY(host_room((void **)&t.h_hit_off, t.h_hit_off_cap, ((size_t)n_reads + 1) * 8));
            t.h_hit_off[0] = 0;
            uint64_t total = 0;
            if (n_reads) {
                unsigned long long n_allhit = 0;
                HIP_TRY(hipMemcpy(&n_allhit, t.d_stats.p + pfq::ST_ALLHIT, 8, hipMemcpyDeviceToHost));
                HIP_TRY(t.d_hit_cnt.ensure(n_reads + 1));
                HIP_TRY(t.d_hit_off.ensure(n_reads + 2));
                HIP_TRY(t.d_hit_sums.ensure((n_reads + 4095) / 4096 + 2));
                HIP_TRY(hipMemsetAsync(t.d_hit_cnt.p, 0, (n_reads + 1) * 4, st));
                pfq::launch_hits_csr(t.d_hit_pairs.p, cursors[0], t.d_allhit.p, n_reads, (uint32_t)nl, n_allhit != 0, t.d_hit_cnt.p, t.d_hit_sums.p,
                                     t.d_hit_off.p, st);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(t.h_hit_off, t.d_hit_off.p, (n_reads + 1) * 8, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                total = t.h_hit_off[n_reads];
                PFQ_TRY(host_room((void **)&t.h_hit_leaves, t.h_hit_leaves_cap, (size_t)total * 4 + 4));
                if (total) {
                    HIP_TRY(t.d_hit_leaves.ensure(total));
                    pfq::launch_hits_fill(t.d_hit_pairs.p, cursors[0], t.d_allhit.p, n_reads, t.d_hit_off.p, t.d_hit_cnt.p, t.d_hit_leaves.p, st);
                    HIP_TRY(hipGetLastError());
                    HIP_TRY(hipMemcpyAsync(t.h_hit_leaves, t.d_hit_leaves.p, total * 4, hipMemcpyDeviceToHost, st));
                    HIP_TRY(hipStreamSynchronize(st));
                }
            } else PFQ_TRY(host_room((void **)&t.h_hit_leaves, t.h_hit_leaves_cap, 4));
            hits->n_reads = n_reads;
            hits->offsets = t.h_hit_off;
            hits->leaves = t.h_hit_leaves;
            return PFQ_OK;
        }
        // hit buffer too small: restore the counters and run the block again with room for every hit
        if (nl) HIP_TRY(hipMemcpy(t.d_counts.p, t.d_counts_snapshot.p, nl * 8, hipMemcpyDeviceToDevice));
        HIP_TRY(t.d_hit_pairs.ensure(cursors[0] + 1024));
        hit_cap = t.d_hit_pairs.n;
        done = false;
        return PFQ_OK;
    }

    int run() {
        for (int attempt_no = 0; attempt_no < 2; ++attempt_no) {
            PFQ_TRY(attempt(attempt_no));
            if (!want_hits) return PFQ_OK;
            bool done = false;
            PFQ_TRY(read_hits(done));
            if (done) return PFQ_OK;
        }
        return fail(PFQ_ERR_DEVICE, "hit buffer overflow persisted");
    }
};

int query_device(pfq_tree &t, const uint8_t *d_seq, const uint64_t *d_off, uint64_t n_reads, uint64_t total_bytes,
                 float threshold, uint32_t flags, hipStream_t st, pfq_hits *hits) {
    QueryRun q(t, d_seq, d_off, n_reads, total_bytes, threshold, flags, st, hits);
    PFQ_TRY(q.plan());
    return q.run();
}

// Reduce a whole tree to subtree shard `index` of the depth-`depth` frontier (pfq_tree_open_subtree): the shard's node,
// everything below it and the chain of its ancestors, each reduced to the child on the path.  `reachable` marks the nodes
// that stay; `chain` lists the ancestors, root first.  Needs parent / depth of every node (relink).
int apply_shard(pfq_tree &t, uint64_t depth, uint64_t index, std::vector<uint8_t> &reachable, std::vector<int32_t> &chain) {
    if (t.root < 0) return fail(PFQ_ERR_STATE, "subtree shard of an empty tree");
    // frontier at depth `depth`, left to right: nodes at that depth and leaves above it
    std::vector<int32_t> frontier, st{t.root};
    while (!st.empty()) {
        int32_t v = st.back();
        st.pop_back();
        const Node &nd = t.nodes[v];
        if (nd.depth == depth || nd.is_leaf()) { frontier.push_back(v); continue; }
        if (nd.right >= 0) st.push_back(nd.right);
        if (nd.left >= 0) st.push_back(nd.left);
    }
    if (index >= frontier.size())
        return fail(PFQ_ERR_ARG, "subtree index " + std::to_string(index) + " out of range: the depth-" +
                                     std::to_string(depth) + " frontier has " + std::to_string(frontier.size()) + " nodes");
    const int32_t target = frontier[index];
    // leaves before the shard in the whole tree's order
    auto count_leaves = [&](int32_t root) {
        uint64_t n = 0;
        std::vector<int32_t> s2{root};
        while (!s2.empty()) {
            int32_t v = s2.back();
            s2.pop_back();
            const Node &nd = t.nodes[v];
            if (nd.is_leaf()) ++n;
            if (nd.left >= 0) s2.push_back(nd.left);
            if (nd.right >= 0) s2.push_back(nd.right);
        }
        return n;
    };
    t.shard_first_leaf = 0;
    for (uint64_t i = 0; i < index; ++i) t.shard_first_leaf += count_leaves(frontier[i]);
    // reduce every ancestor to the child on the path
    chain.clear();
    for (int32_t c = target, v = t.nodes[target].parent; v >= 0; c = v, v = t.nodes[v].parent) {
        if (t.nodes[v].left != c) t.nodes[v].left = -1;
        if (t.nodes[v].right != c) t.nodes[v].right = -1;
        chain.insert(chain.begin(), v);
    }
    reachable.assign(t.nodes.size(), 0);
    std::vector<int32_t> s3{t.root};
    while (!s3.empty()) {
        int32_t v = s3.back();
        s3.pop_back();
        reachable[v] = 1;
        if (t.nodes[v].left >= 0) s3.push_back(t.nodes[v].left);
        if (t.nodes[v].right >= 0) s3.push_back(t.nodes[v].right);
    }
    t.is_shard = true;
    return PFQ_OK;
}

// Balanced synthetic tree topology; same numbering as oracle/pfq_oracle.py:build_balanced_tree.
int32_t build_balanced_rec(pfq_tree &t, const char *const *tax_ids, uint64_t lo, uint64_t hi, int32_t parent,
                           uint32_t depth, uint64_t &internal_counter, std::vector<std::pair<uint64_t, uint64_t>> &range) {
    int32_t v = (int32_t)t.nodes.size();
    t.nodes.emplace_back();
    range.emplace_back(lo, hi);  // genomes below node v
    t.nodes[v].parent = parent;
    t.nodes[v].depth = depth;
    t.nodes[v].filter = (uint32_t)v;
    t.nodes[v].has_tax = true;
    if (hi - lo == 1) {
        t.nodes[v].tax_id = tax_ids[lo];
        t.nodes[v].bf_path = t.nodes[v].tax_id + ".bf";
        return v;
    }
    t.nodes[v].tax_id = "Internal_Node_" + std::to_string(internal_counter++);
    t.nodes[v].bf_path = t.nodes[v].tax_id + ".bf";
    uint64_t mid = lo + (hi - lo + 1) / 2;
    int32_t l = build_balanced_rec(t, tax_ids, lo, mid, v, depth + 1, internal_counter, range);
    int32_t r = build_balanced_rec(t, tax_ids, mid, hi, v, depth + 1, internal_counter, range);
    t.nodes[v].left = l;
    t.nodes[v].right = r;
    return v;
}

// shard: only subtree shard `shard_index` of the depth-`shard_depth` frontier is materialised (its subtree bottom-up from
// its own genomes; every ancestor on the chain = the union of ALL genomes below it in the whole tree, inserted directly).
int build_balanced_common(const uint8_t *d_genomes, const uint64_t *d_goff, uint64_t n_genomes, const char *const *tax_ids,
                          uint64_t kmer_size, uint64_t nbits, uint32_t num_hashes, uint64_t seed1, uint64_t seed2,
                          float fpr, uint32_t largest, int device, bool shard, uint64_t shard_depth, uint64_t shard_index,
                          pfq_tree **out) {
    std::unique_ptr<pfq_tree> t(new pfq_tree());
    t->device = device;
    t->kmer_size = kmer_size;
    t->nbits = nbits;
    t->num_hashes = num_hashes;
    t->seed1 = seed1;
    t->seed2 = seed2;
    t->false_pos_rate = fpr;
    t->largest_expected_genome = largest;
    PFQ_TRY(setup_hash_params(*t));
    if (n_genomes > (1u << 24)) return fail(PFQ_ERR_UNSUPPORTED, "more than 2^24 genomes in one balanced build");
    std::vector<std::pair<uint64_t, uint64_t>> range;
    if (n_genomes) {
        uint64_t counter = 0;
        t->root = build_balanced_rec(*t, tax_ids, 0, n_genomes, -1, 0, counter, range);
    }
    t->tree_leaves = n_genomes;
    std::vector<uint8_t> reachable(t->nodes.size(), 1);
    std::vector<int32_t> chain;
    if (shard) PFQ_TRY(apply_shard(*t, shard_depth, shard_index, reachable, chain));
    std::vector<uint8_t> on_chain(t->nodes.size(), 0);
    for (int32_t v : chain) on_chain[v] = 1;
    // filter rows: one per node that stays (a whole tree: row == node index)
    for (size_t v = 0; v < t->nodes.size(); ++v) {
        if (!reachable[v]) continue;
        t->nodes[v].filter = (uint32_t)t->filter_paths.size();
        t->filter_paths.push_back(t->nodes[v].bf_path);
    }
    const size_t nn = t->nodes.size(), nrows = t->filter_paths.size();
    if (nrows) {
        if (t->d_bits.ensure(nrows * t->n_words) != hipSuccess) {
            (void)hipGetLastError();
            return fail(PFQ_ERR_DEVICE, "not enough device memory for " + std::to_string(nrows) + " filters of " +
                                            std::to_string(t->n_words * 8) + " bytes");
        }
        t->n_rows = t->row_capacity = nrows;
        HIP_TRY(hipMemset(t->d_bits.p, 0, nrows * t->n_words * 8));
        // genome -> filter rows it is inserted into: its leaf (when the leaf stays) and every ancestor on the shard's chain
        // (k_insert takes one row per genome and launch; the chain is a handful of nodes)
        DevBuf<uint32_t> d_leaf_row, d_triples;
        HIP_TRY(d_leaf_row.ensure(n_genomes + 1));
        std::vector<uint32_t> leaf_row(n_genomes, 0xffffffffu);
        for (size_t v = 0; v < nn; ++v)
            if (reachable[v] && range[v].second - range[v].first == 1 && t->nodes[v].is_leaf() && !on_chain[v])
                leaf_row[range[v].first] = t->nodes[v].filter;
        auto insert_range = [&](uint64_t lo, uint64_t hi) -> int {  // genomes [lo, hi) with rows leaf_row[lo..hi)
            // runs of genomes that have a row; blockIdx.y is limited to 65535 genomes per launch
            for (uint64_t g = lo; g < hi;) {
                if (leaf_row[g] == 0xffffffffu) { ++g; continue; }
                uint64_t e = g;
                while (e < hi && e - g < 32768 && leaf_row[e] != 0xffffffffu) ++e;
                HIP_TRY(hipMemcpy(d_leaf_row.p, leaf_row.data() + g, (e - g) * 4, hipMemcpyHostToDevice));
                pfq::launch_insert(t->hp, d_genomes, d_goff + g, (uint32_t)(e - g), d_leaf_row.p, t->d_bits.p, t->n_words, nullptr);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipDeviceSynchronize());
                g = e;
            }
            return PFQ_OK;
        };
        PFQ_TRY(insert_range(0, n_genomes));
        for (int32_t v : chain) {  // ancestors of the shard: every genome below them, straight into their row
            std::fill(leaf_row.begin(), leaf_row.end(), 0xffffffffu);
            for (uint64_t g = range[v].first; g < range[v].second; ++g) leaf_row[g] = t->nodes[v].filter;
            PFQ_TRY(insert_range(range[v].first, range[v].second));
        }
        // internal nodes of the (sub)tree bottom-up, one launch per depth
        uint32_t max_depth = 0;
        for (auto &nd : t->nodes) max_depth = std::max(max_depth, nd.depth);
        for (int d = (int)max_depth; d >= 0; --d) {
            std::vector<uint32_t> triples;
            for (size_t v = 0; v < nn; ++v) {
                const Node &nd = t->nodes[v];
                if ((int)nd.depth != d || nd.is_leaf() || !reachable[v] || on_chain[v]) continue;
                triples.push_back(nd.filter);
                triples.push_back(t->nodes[nd.left].filter);
                triples.push_back(t->nodes[nd.right].filter);
            }
            if (triples.empty()) continue;
            HIP_TRY(d_triples.ensure(triples.size()));
            HIP_TRY(hipMemcpy(d_triples.p, triples.data(), triples.size() * 4, hipMemcpyHostToDevice));
            for (size_t t0 = 0; t0 < triples.size() / 3; t0 += 32768) {
                uint32_t nt = (uint32_t)std::min<size_t>(32768, triples.size() / 3 - t0);
                pfq::launch_union(t->d_bits.p, t->n_words, d_triples.p + 3 * t0, nt, nullptr);
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipDeviceSynchronize());
        }
        HIP_TRY(hipDeviceSynchronize());
        PFQ_TRY(verify_supersets(*t));
    }
    *out = t.release();
    return PFQ_OK;
}

void put_u64(std::vector<uint8_t> &o, uint64_t v) { o.insert(o.end(), (uint8_t *)&v, (uint8_t *)&v + 8); }
void put_u32(std::vector<uint8_t> &o, uint32_t v) { o.insert(o.end(), (uint8_t *)&v, (uint8_t *)&v + 4); }
void put_str(std::vector<uint8_t> &o, const std::string &s) {
    put_u64(o, s.size());
    o.insert(o.end(), s.begin(), s.end());
}
void encode_node(const pfq_tree &t, int32_t v, std::vector<uint8_t> &o) {
    const Node &nd = t.nodes[v];
    for (int32_t c : {nd.left, nd.right}) {
        o.push_back(c >= 0 ? 1 : 0);
        if (c >= 0) encode_node(t, c, o);
    }
    put_str(o, nd.bf_path);
    o.push_back(nd.has_tax ? 1 : 0);
    if (nd.has_tax) put_str(o, nd.tax_id);
    put_u64(o, nd.mapped_reads);
}

const char ORDER_NAME[] = "bitvec::order::Lsb0";

}  // namespace

// =================================================================================================================
// C ABI
// =================================================================================================================
extern "C" {

int pfq_host_alloc(uint64_t bytes, void **out) {
    if (!out) return fail(PFQ_ERR_ARG, "null argument");
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return PFQ_OK;
}
int pfq_host_free(void *p) {
    if (p) HIP_TRY(hipHostFree(p));
    return PFQ_OK;
}

const char *pfq_last_error(void) { return g_err.c_str(); }
const char *pfq_version(void) { return "libpfq 0.1 (gfx950)"; }

static int open_impl(const char *db_dir, int device, bool shard, uint64_t shard_depth, uint64_t shard_index, pfq_tree **out) {
    if (!db_dir || !out) return fail(PFQ_ERR_ARG, "null argument");
    *out = nullptr;
    PFQ_TRY(use_device(device));
    std::unique_ptr<pfq_tree> t(new pfq_tree());
    t->device = device;
    std::string dir(db_dir);
    std::vector<uint8_t> buf;
    if (!read_file(dir + "/tree.bin", buf))
        return fail(PFQ_ERR_IO, "cannot read " + dir + "/tree.bin (reference: panic at bloom_tree.rs:375)");
    Cur c{buf.data(), buf.size()};
    uint8_t tag = c.u8();
    if (!c.ok || tag > 1) return fail(PFQ_ERR_FORMAT, "tree.bin: bad Option tag for root");
    if (tag == 1) PFQ_TRY(parse_node(c, *t, -1, 0, t->root));
    t->false_pos_rate = c.f32();
    t->largest_expected_genome = c.u32();
    t->kmer_size = c.u64();
    t->seed1 = c.u64();
    t->seed2 = c.u64();
    if (!c.ok || c.p != c.n) return fail(PFQ_ERR_FORMAT, "tree.bin: truncated or trailing bytes");
    t->tree_leaves = leaves_dfs(*t).size();
    std::vector<uint8_t> reachable(t->nodes.size(), 1);
    std::vector<int32_t> chain;
    if (shard) PFQ_TRY(apply_shard(*t, shard_depth, shard_index, reachable, chain));
    // filters keyed by relative path, exactly like the LRU cache key (cache.rs:56-62)
    std::map<std::string, uint32_t> row_of;
    for (size_t vi = 0; vi < t->nodes.size(); ++vi) {
        auto &nd = t->nodes[vi];
        if (!reachable[vi]) continue;  // outside this shard: its .bf is never read
        auto it = row_of.find(nd.bf_path);
        if (it == row_of.end()) {
            nd.filter = (uint32_t)t->filter_paths.size();
            row_of[nd.bf_path] = nd.filter;
            t->filter_paths.push_back(nd.bf_path);
        } else nd.filter = it->second;
    }
    // One .bf per filter: read, check, upload.  The first one fixes the geometry; the others are loaded by a few
    // threads side by side (a 1024-leaf database is 2047 files of 9 MB: reading them one after the other takes longer
    // than classifying a hundred million reads).  Returns a status and leaves the message in `msg`.
    pfq_tree *tp = t.get();
    auto load_one = [&](size_t f, std::vector<uint8_t> &fb, bool first, std::string &msg) -> int {
        const std::string path = dir + "/" + tp->filter_paths[f];
        if (!read_file(path, fb)) {
            msg = "cannot read Bloom filter file " + path + " (reference: panic at bloom_filter.rs:155)";
            return PFQ_ERR_IO;
        }
        Cur b{fb.data(), fb.size()};
        std::string order = b.str();
        uint8_t width = b.u8(), index = b.u8();
        uint64_t nbits = b.u64(), nwords = b.u64();
        if (!b.ok || order != ORDER_NAME || width != 64 || index != 0 || nwords != (nbits + 63) / 64) {
            msg = path + ": not a BitVec<usize, Lsb0> BloomFilter (order/head/length mismatch)";
            return PFQ_ERR_FORMAT;
        }
        const uint8_t *words = b.take((size_t)nwords * 8);
        uint32_t nh = b.u32();
        uint64_t s1 = b.u64(), s2 = b.u64();
        uint8_t ptag = b.u8();
        if (b.ok && ptag == 1) (void)b.str();
        if (!b.ok || ptag > 1 || b.p != b.n) {
            msg = path + ": truncated or trailing bytes";
            return PFQ_ERR_FORMAT;
        }
        if (first) {
            tp->nbits = nbits;
            tp->num_hashes = nh;
            int rc = setup_hash_params(*tp);
            if (rc != PFQ_OK) {
                msg = g_err;
                return rc;
            }
            if (tp->d_bits.ensure(tp->filter_paths.size() * tp->n_words) != hipSuccess) {
                (void)hipGetLastError();
                msg = "not enough device memory for " + std::to_string(tp->filter_paths.size()) + " filters";
                return PFQ_ERR_DEVICE;
            }
            tp->n_rows = tp->row_capacity = tp->filter_paths.size();
        }
        if (nbits != tp->nbits || nh != tp->num_hashes) {
            msg = path + ": filter size / num_hashes differ from the other nodes";
            return PFQ_ERR_UNSUPPORTED;
        }
        if (s1 != tp->seed1 || s2 != tp->seed2) {
            msg = path + ": hash seeds differ from tree.bin's (node-independent indices need one seed pair)";
            return PFQ_ERR_UNSUPPORTED;
        }
        if (hipMemcpy(tp->d_bits.p + f * tp->n_words, words, (size_t)nwords * 8, hipMemcpyHostToDevice) != hipSuccess) {
            msg = std::string("hipMemcpy of ") + path + ": " + hipGetErrorString(hipGetLastError());
            return PFQ_ERR_DEVICE;
        }
        return PFQ_OK;
    };
    const size_t n_files = t->filter_paths.size();
    if (n_files) {
        std::vector<uint8_t> fb;
        std::string msg;
        int rc = load_one(0, fb, true, msg);
        if (rc != PFQ_OK) return fail(rc, msg);
    }
    if (n_files > 1) {
        unsigned n_threads = (unsigned)std::min<size_t>(8, n_files - 1);
        if (const char *e = getenv("PFQ_LOAD_THREADS")) n_threads = (unsigned)std::min<size_t>(std::max(1, atoi(e)), n_files - 1);
        std::atomic<size_t> next{1};
        std::mutex err_mu;
        int err_rc = PFQ_OK;
        size_t err_file = ~(size_t)0;  // the lowest-numbered failing file is reported, like a sequential reader would
        std::string err_msg;
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < n_threads; ++w)
            pool.emplace_back([&] {
                (void)hipSetDevice(device);
                std::vector<uint8_t> fb;
                std::string msg;
                for (size_t f; (f = next.fetch_add(1)) < n_files;) {
                    int rc = load_one(f, fb, false, msg);
                    if (rc != PFQ_OK) {
                        std::lock_guard<std::mutex> lk(err_mu);
                        if (f < err_file) {
                            err_file = f;
                            err_rc = rc;
                            err_msg = msg;
                        }
                    }
                }
            });
        for (auto &th : pool) th.join();
        if (err_rc != PFQ_OK) return fail(err_rc, err_msg);
    }
    if (t->root >= 0) PFQ_TRY(verify_supersets(*t));
    *out = t.release();
    return PFQ_OK;
}

int pfq_tree_open(const char *db_dir, int device, pfq_tree **out) { return open_impl(db_dir, device, false, 0, 0, out); }

int pfq_tree_open_subtree(const char *db_dir, int device, uint64_t depth, uint64_t index, pfq_tree **out) {
    return open_impl(db_dir, device, true, depth, index, out);
}

// BloomTree::new (bloom_tree.rs:100-118) with explicit hash seeds: an empty tree whose filters are sized like
// create_bloom_filter / with_rate (bloom_filter.rs:55-70,:229-240).
int pfq_tree_create(uint64_t kmer_size, float false_pos_rate, uint32_t largest_expected_genome, uint64_t seed1, uint64_t seed2,
                    uint64_t expected_genomes, int device, pfq_tree **out) {
    if (!out) return fail(PFQ_ERR_ARG, "null argument");
    *out = nullptr;
    PFQ_TRY(use_device(device));
    std::unique_ptr<pfq_tree> t(new pfq_tree());
    t->device = device;
    t->kmer_size = kmer_size;
    t->false_pos_rate = false_pos_rate;
    t->largest_expected_genome = largest_expected_genome;
    t->seed1 = seed1;
    t->seed2 = seed2;
    if (!(false_pos_rate > 0.0f) || largest_expected_genome == 0)
        return fail(PFQ_ERR_ARG, "false_pos_rate must be > 0 and largest_expected_genome > 0");
    t->nbits = needed_bits_f32(false_pos_rate, largest_expected_genome);
    t->num_hashes = optimal_num_hashes_f32(t->nbits, largest_expected_genome);
    PFQ_TRY(setup_hash_params(*t));
    if (expected_genomes) PFQ_TRY(reserve_rows(*t, 2 * expected_genomes - 1));
    *out = t.release();
    return PFQ_OK;
}

// BloomTree::insert (bloom_tree.rs:128-143): init_leaf_node (:154-168) + add_to_tree (:187-214) + init_internal_node
// (:226-245).  Internal nodes are named `internal_name` or "Internal_Node_<n>" with a running n that is unique in the
// tree (the reference draws a random u16, :231-233, which can collide and then shares a .bf file between two nodes).
int pfq_tree_insert(pfq_tree *tree, const uint8_t *seq, uint64_t len, const char *tax_id, const char *internal_name) {
    if (!tree || !tax_id || (len && !seq)) return fail(PFQ_ERR_ARG, "null argument");
    PFQ_TRY(use_device(tree->device));
    pfq_tree &t = *tree;
    if (t.is_shard) return fail(PFQ_ERR_STATE, "a subtree shard cannot be extended");
    if (t.n_words == 0) return fail(PFQ_ERR_STATE, "tree has no filter geometry");
    // One .bf per node here.  (The reference keys filters by file name: a second node called <tax_id> gets a fresh empty
    // filter under the first one's key, bloom_tree.rs:294 / cache.rs:83-87, and the two then share one file on disk —
    // SURVEY H4.  Such databases can be OPENED; building one is refused.)
    if (t.path_set.size() != t.filter_paths.size()) {
        t.path_set.clear();
        t.path_set.insert(t.filter_paths.begin(), t.filter_paths.end());
    }
    auto path_taken = [&](const std::string &pth) { return t.path_set.count(pth) != 0; };
    if (path_taken(std::string(tax_id) + ".bf"))
        return fail(PFQ_ERR_ARG, std::string("a node named ") + tax_id + " exists already: two nodes would share " + tax_id + ".bf");
    if (internal_name && (path_taken(std::string(internal_name) + ".bf") || !strcmp(internal_name, tax_id)))
        return fail(PFQ_ERR_ARG, std::string("a node named ") + internal_name + " exists already: two nodes would share " + internal_name + ".bf");
    PFQ_TRY(sync_counts_to_nodes(t));
    t.layout_valid = false;
    PFQ_TRY(reserve_rows(t, t.n_rows + 2));
    if (!t.greedy_blocks) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, t.device));
        // every block must be resident (one per CU at most); blocks of 1024 threads, half as many as CUs: they stream the filters
        // as fast as all would (measured: 128 blocks 5300, 192 5270, 256 4790 genomes/s)
        int want = prop.multiProcessorCount / 2;
        if (const char *e = getenv("PFQ_GREEDY_BLOCKS")) want = atoi(e);
        t.greedy_blocks = std::max(1, std::min(std::min(pfq::GREEDY_MAX_BLOCKS, prop.multiProcessorCount), want));
    }
    // the device's copy of the shape (all of it once; afterwards the kernel keeps it current)
    const size_t n_after = t.nodes.size() + 2;
    if (t.knobs.greedy_host > 0) HIP_TRY(t.d_dist.ensure(2 * (size_t)pfq::INSERT_STEP_BLOCKS));
    else if (!t.topo_on_device || t.d_topo.n < n_after) {
        PFQ_TRY(sync_topology(t));
        HIP_TRY(hipDeviceSynchronize());
        const size_t cap = std::max<size_t>(n_after, 2 * t.d_topo.n + 1024);
        HIP_TRY(t.d_topo.ensure(cap));
        HIP_TRY(t.d_walk.ensure(4));
        const size_t sync_words = (size_t)pfq::GREEDY_SYNC_LINES * pfq::GREEDY_SYNC_STRIDE / 8;
        HIP_TRY(t.d_dist.ensure(std::max<size_t>(sync_words, 2 * (size_t)pfq::INSERT_STEP_BLOCKS)));
        HIP_TRY(hipMemset(t.d_dist.p, 0, sync_words * 8));  // the walk's counters, generation words and accumulators start clear
        std::vector<pfq::TopoNode> h(t.nodes.size());
        for (size_t v = 0; v < h.size(); ++v) h[v] = pfq::TopoNode{t.nodes[v].left, t.nodes[v].right, t.nodes[v].filter, 0u};
        if (!h.empty()) HIP_TRY(hipMemcpy(t.d_topo.p, h.data(), h.size() * sizeof(pfq::TopoNode), hipMemcpyHostToDevice));
        const int st[4] = {t.root, 0, t.root, t.root};
        t.walk_seq = 0;
        HIP_TRY(hipMemcpy(t.d_walk.p, st, sizeof st, hipMemcpyHostToDevice));
        t.topo_on_device = true;
    }
    // the new leaf's filter: the genome goes to one of four staging buffers (the copy of genome i + 1 does not wait for the
    // kernels of genome i), its k-mers are inserted into a cleared row
    const uint32_t new_row = (uint32_t)t.n_rows++, int_row = (uint32_t)t.n_rows++;  // (the internal node's row stays unused by the first leaf of a tree)
    const uint32_t slot = t.gseq_next++ & 3u;
    if (!t.copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&t.copy_stream, hipStreamNonBlocking));
        for (auto &e : t.in_free) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (!t.gseq_free[slot]) {
        HIP_TRY(hipEventCreateWithFlags(&t.gseq_free[slot], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&t.gseq_copied[slot], hipEventDisableTiming));
    } else HIP_TRY(hipEventSynchronize(t.gseq_free[slot]));   // the insertion that used these buffers four calls ago has read them
    if (t.d_gseq[slot].n < len + 16) {
        HIP_TRY(t.d_gseq[slot].ensure(std::max<size_t>(len + 16, 2 * t.d_gseq[slot].n)));
    }
    if (t.h_gseq_n[slot] < len + 16) {
        if (t.h_gseq[slot]) HIP_TRY(hipHostFree(t.h_gseq[slot]));
        t.h_gseq[slot] = nullptr;
        t.h_gseq_n[slot] = 0;
        const size_t want = std::max<size_t>(len + 16, 2 * t.h_gseq_n[slot]);
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&t.h_gseq[slot]), want, hipHostMallocDefault));
        t.h_gseq_n[slot] = want;
    }
    // (through page-locked staging on a stream of its own: the copy neither waits for the kernels of the insertions before
    // nor holds the caller — the kernels wait for it by event)
    if (len) {
        memcpy(t.h_gseq[slot], seq, len);
        HIP_TRY(hipMemcpyAsync(t.d_gseq[slot].p, t.h_gseq[slot], len, hipMemcpyHostToDevice, t.copy_stream));
    }
    HIP_TRY(hipEventRecord(t.gseq_copied[slot], t.copy_stream));
    HIP_TRY(hipStreamWaitEvent(nullptr, t.gseq_copied[slot], 0));
    HIP_TRY(hipMemsetAsync(t.d_bits.p + (uint64_t)new_row * t.n_words, 0, t.n_words * 8, nullptr));
    pfq::launch_insert_one(t.hp, t.d_gseq[slot].p, len, new_row, t.d_bits.p, t.n_words, nullptr);
    HIP_TRY(hipEventRecord(t.gseq_free[slot], nullptr));
    Node leaf;
    leaf.has_tax = true;
    leaf.tax_id = tax_id;
    leaf.bf_path = std::string(tax_id) + ".bf";
    leaf.filter = new_row;
    t.filter_paths.push_back(leaf.bf_path);
    t.path_set.insert(leaf.bf_path);
    const int32_t nv = (int32_t)t.nodes.size();
    t.nodes.push_back(leaf);
    t.topology_dirty = true;
    // PFQ_GREEDY_HOST=1: the descent level by level from the host (one launch and one read-back per level; no kernel with a
    // grid barrier) — for devices that are shared with other work, where not every block of such a kernel stays resident
    if (t.knobs.greedy_host > 0) {
        PFQ_TRY(sync_topology(t));
        t.topo_on_device = false;
        if (t.root < 0 || nv == 0) {
            t.root = nv;
            --t.n_rows;  // (no internal node: its row is not used)
            HIP_TRY(hipDeviceSynchronize());
            return PFQ_OK;
        }
        std::vector<unsigned long long> part(2 * pfq::INSERT_STEP_BLOCKS);
        int32_t cur = t.root, parent = -1;
        bool went_right = false;
        while (true) {
            const Node &c = t.nodes[cur];
            if (c.left >= 0 && c.right >= 0) {
                pfq::launch_insert_step(t.d_bits.p, t.n_words, c.filter, new_row, t.nodes[c.left].filter, t.nodes[c.right].filter, t.d_dist.p, nullptr);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpy(part.data(), t.d_dist.p, part.size() * 8, hipMemcpyDeviceToHost));
                unsigned long long d[2] = {0, 0};
                for (uint32_t b = 0; b < pfq::INSERT_STEP_BLOCKS; ++b) {
                    d[0] += part[2 * b];
                    d[1] += part[2 * b + 1];
                }
                parent = cur;
                went_right = d[1] < d[0];  // `if right_distance < left_distance` (bloom_tree.rs:201): ties go left
                cur = went_right ? c.right : c.left;
            } else if (c.is_leaf()) {
                std::string name;
                if (internal_name) name = internal_name;
                else {
                    do name = "Internal_Node_" + std::to_string(t.internal_counter++);
                    while (path_taken(name + ".bf"));
                }
                HIP_TRY(t.d_build.ensure(8));
                const uint32_t triple[3] = {int_row, c.filter, new_row};
                HIP_TRY(hipMemcpy(t.d_build.p + 4, triple, 12, hipMemcpyHostToDevice));
                pfq::launch_union(t.d_bits.p, t.n_words, t.d_build.p + 4, 1, nullptr);
                HIP_TRY(hipGetLastError());
                Node in;
                in.has_tax = true;
                in.tax_id = name;
                in.bf_path = name + ".bf";
                in.filter = int_row;
                in.left = cur;   // the node already in the tree (bloom_tree.rs:241)
                in.right = nv;   // the new leaf (:242)
                t.filter_paths.push_back(in.bf_path);
                t.path_set.insert(in.bf_path);
                const int32_t ni_h = (int32_t)t.nodes.size();
                t.nodes.push_back(in);
                if (parent < 0) t.root = ni_h;
                else (went_right ? t.nodes[parent].right : t.nodes[parent].left) = ni_h;
                break;
            } else {
                return fail(PFQ_ERR_FORMAT, "Node with only one child encountered - should not happen. (bloom_tree.rs:209)");
            }
        }
        HIP_TRY(hipDeviceSynchronize());
        return PFQ_OK;
    }
    // BloomTree::insert (bloom_tree.rs:128-143): the first leaf is the root; every later one is placed by the greedy descent,
    // which ends in a new internal node (left = the leaf it reached, right = the new leaf, filter = their union)
    int32_t ni = -1;
    if (nv > 0 || t.root >= 0) {
        std::string name;
        if (internal_name) name = internal_name;
        else {
            do name = "Internal_Node_" + std::to_string(t.internal_counter++);
            while (path_taken(name + ".bf"));
        }
        Node in;
        in.has_tax = true;
        in.tax_id = name;
        in.bf_path = name + ".bf";
        in.filter = int_row;
        in.right = nv;   // the new leaf (:242); `left` = the leaf the walk reaches, known on the device (sync_topology)
        t.filter_paths.push_back(in.bf_path);
        t.path_set.insert(in.bf_path);
        ni = (int32_t)t.nodes.size();
        t.nodes.push_back(in);
    } else {
        --t.n_rows;  // (no internal node: its row is not used)
    }
    if (ni < 0) t.root = nv;  // (the host's root is only a hint while insertions are pending; empty vs. not is what counts)
    pfq::launch_greedy_insert(t.d_bits.p, t.n_words, t.d_topo.p, t.d_walk.p, t.d_dist.p, nv, ni < 0 ? nv : ni, new_row, int_row, t.walk_seq++, t.greedy_blocks, nullptr);
    HIP_TRY(hipGetLastError());
    t.topo_pending = true;
    return PFQ_OK;
}

int pfq_tree_build_balanced(const uint8_t *genomes, const uint64_t *offsets, uint64_t n_genomes, const char *const *tax_ids,
                            uint64_t kmer_size, uint64_t nbits, uint32_t num_hashes, uint64_t seed1, uint64_t seed2,
                            float false_pos_rate, uint32_t largest_expected_genome, int device, pfq_tree **out) {
    if (!out || (n_genomes && (!genomes || !offsets || !tax_ids))) return fail(PFQ_ERR_ARG, "null argument");
    *out = nullptr;
    PFQ_TRY(use_device(device));
    DevBuf<uint8_t> d_g;
    DevBuf<uint64_t> d_o;
    uint64_t total = n_genomes ? offsets[n_genomes] : 0;
    HIP_TRY(d_g.ensure(total + 1));
    HIP_TRY(d_o.ensure(n_genomes + 1));
    if (total) HIP_TRY(hipMemcpy(d_g.p, genomes, total, hipMemcpyHostToDevice));
    if (n_genomes) HIP_TRY(hipMemcpy(d_o.p, offsets, (n_genomes + 1) * 8, hipMemcpyHostToDevice));
    return build_balanced_common(d_g.p, d_o.p, n_genomes, tax_ids, kmer_size, nbits, num_hashes, seed1, seed2,
                                 false_pos_rate, largest_expected_genome, device, false, 0, 0, out);
}

int pfq_tree_build_balanced_device(const uint8_t *d_genomes, uint64_t genome_len, uint64_t n_genomes,
                                   const char *const *tax_ids, uint64_t kmer_size, uint64_t nbits, uint32_t num_hashes,
                                   uint64_t seed1, uint64_t seed2, float false_pos_rate, uint32_t largest_expected_genome,
                                   int device, pfq_tree **out) {
    if (!out || (n_genomes && (!d_genomes || !tax_ids))) return fail(PFQ_ERR_ARG, "null argument");
    *out = nullptr;
    PFQ_TRY(use_device(device));
    std::vector<uint64_t> off(n_genomes + 1);
    for (uint64_t i = 0; i <= n_genomes; ++i) off[i] = i * genome_len;
    DevBuf<uint64_t> d_o;
    HIP_TRY(d_o.ensure(n_genomes + 1));
    HIP_TRY(hipMemcpy(d_o.p, off.data(), off.size() * 8, hipMemcpyHostToDevice));
    return build_balanced_common(d_genomes, d_o.p, n_genomes, tax_ids, kmer_size, nbits, num_hashes, seed1, seed2,
                                 false_pos_rate, largest_expected_genome, device, false, 0, 0, out);
}

int pfq_tree_build_balanced_subtree_device(const uint8_t *d_genomes, uint64_t genome_len, uint64_t n_genomes,
                                           const char *const *tax_ids, uint64_t kmer_size, uint64_t nbits, uint32_t num_hashes,
                                           uint64_t seed1, uint64_t seed2, float false_pos_rate, uint32_t largest_expected_genome,
                                           uint64_t depth, uint64_t index, int device, pfq_tree **out) {
    if (!out || (n_genomes && (!d_genomes || !tax_ids))) return fail(PFQ_ERR_ARG, "null argument");
    *out = nullptr;
    PFQ_TRY(use_device(device));
    std::vector<uint64_t> off(n_genomes + 1);
    for (uint64_t i = 0; i <= n_genomes; ++i) off[i] = i * genome_len;
    DevBuf<uint64_t> d_o;
    HIP_TRY(d_o.ensure(n_genomes + 1));
    HIP_TRY(hipMemcpy(d_o.p, off.data(), off.size() * 8, hipMemcpyHostToDevice));
    return build_balanced_common(d_genomes, d_o.p, n_genomes, tax_ids, kmer_size, nbits, num_hashes, seed1, seed2,
                                 false_pos_rate, largest_expected_genome, device, true, depth, index, out);
}

int pfq_tree_save(const pfq_tree *tree, const char *db_dir) {
    if (!tree || !db_dir) return fail(PFQ_ERR_ARG, "null argument");
    PFQ_TRY(use_device(tree->device));
    PFQ_TRY(finish_topology(*const_cast<pfq_tree *>(tree)));
    // BloomTree::save writes the live mapped_reads (bloom_tree.rs:339-355): fold the device counters back first
    PFQ_TRY(sync_counts_to_nodes(*const_cast<pfq_tree *>(tree)));
    const pfq_tree &t = *tree;
    if (t.is_shard) return fail(PFQ_ERR_STATE, "a subtree shard is not a whole database and cannot be saved");
    std::string dir(db_dir);
    std::vector<uint8_t> o;
    o.push_back(t.root >= 0 ? 1 : 0);
    if (t.root >= 0) encode_node(t, t.root, o);
    o.insert(o.end(), (const uint8_t *)&t.false_pos_rate, (const uint8_t *)&t.false_pos_rate + 4);
    put_u32(o, t.largest_expected_genome);
    put_u64(o, t.kmer_size);
    put_u64(o, t.seed1);
    put_u64(o, t.seed2);
    {
        FILE *f = fopen((dir + "/tree.bin").c_str(), "wb");
        if (!f) return fail(PFQ_ERR_IO, "cannot create " + dir + "/tree.bin: " + strerror(errno));
        bool ok = fwrite(o.data(), 1, o.size(), f) == o.size();
        ok = (fclose(f) == 0) && ok;
        if (!ok) return fail(PFQ_ERR_IO, "short write to tree.bin");
    }
    std::vector<uint64_t> words((size_t)t.n_words);
    for (size_t fi = 0; fi < t.filter_paths.size(); ++fi) {
        HIP_TRY(hipMemcpy(words.data(), t.d_bits.p + fi * t.n_words, (size_t)t.n_words * 8, hipMemcpyDeviceToHost));
        std::vector<uint8_t> h;
        put_str(h, ORDER_NAME);
        h.push_back(64);
        h.push_back(0);
        put_u64(h, t.nbits);
        put_u64(h, t.n_words);
        std::vector<uint8_t> tail;
        put_u32(tail, t.num_hashes);
        put_u64(tail, t.seed1);
        put_u64(tail, t.seed2);
        const std::string path = dir + "/" + t.filter_paths[fi];
        tail.push_back(1);
        put_str(tail, path);
        FILE *f = fopen(path.c_str(), "wb");
        if (!f) return fail(PFQ_ERR_IO, "cannot create " + path + ": " + strerror(errno));
        bool ok = fwrite(h.data(), 1, h.size(), f) == h.size();
        ok = ok && fwrite(words.data(), 8, words.size(), f) == words.size();
        ok = ok && fwrite(tail.data(), 1, tail.size(), f) == tail.size();
        ok = (fclose(f) == 0) && ok;
        if (!ok) return fail(PFQ_ERR_IO, "short write to " + path);
    }
    return PFQ_OK;
}

int pfq_tree_info(const pfq_tree *tree, pfq_info *out) {
    if (!tree || !out) return fail(PFQ_ERR_ARG, "null argument");
    if (tree->topology_dirty) {
        PFQ_TRY(use_device(tree->device));
        PFQ_TRY(finish_topology(*const_cast<pfq_tree *>(tree)));
    }
    const pfq_tree &t = *tree;
    out->kmer_size = t.kmer_size;
    out->nbits = t.nbits;
    out->num_hashes = t.num_hashes;
    out->largest_expected_genome = t.largest_expected_genome;
    out->false_pos_rate = t.false_pos_rate;
    out->superset_verified = t.superset_all ? 1 : 0;
    out->seed1 = t.seed1;
    out->seed2 = t.seed2;
    out->n_nodes = t.nodes.size();
    out->n_leaves = leaves_dfs(t).size();
    out->n_filters = t.filter_paths.size();
    out->shard_first_leaf = t.shard_first_leaf;
    out->tree_leaves = t.is_shard ? t.tree_leaves : out->n_leaves;
    out->device_bytes = t.d_bits.bytes() + t.d_S.bytes() + t.d_pairs.bytes() + t.d_sorted.bytes() + t.d_fail.bytes() +
                        t.d_hit_pairs.bytes() + t.d_seq.bytes() + t.d_off.bytes() + t.d_recs.bytes() + t.d_entries.bytes();
    return PFQ_OK;
}

int pfq_tree_prune(pfq_tree *tree, uint64_t search_depth) {
    if (!tree) return fail(PFQ_ERR_ARG, "null argument");
    PFQ_TRY(use_device(tree->device));
    pfq_tree &t = *tree;
    PFQ_TRY(finish_topology(t));
    if (t.root < 0) return fail(PFQ_ERR_STATE, "prune_tree on an empty tree (reference: unwrap panic, bloom_tree.rs:310)");
    PFQ_TRY(sync_counts_to_nodes(t));
    relink(t);
    for (auto &nd : t.nodes)
        if (nd.depth >= search_depth) nd.left = nd.right = -1;  // bloom_tree.rs:322-325
    // nodes below the cut are unreachable now; they keep their slots (and filters) but never appear as leaves
    t.layout_valid = false;
    return PFQ_OK;
}

void pfq_tree_close(pfq_tree *tree) {
    if (!tree) return;
    (void)hipSetDevice(tree->device);
    (void)hipDeviceSynchronize();
    if (tree->copy_stream) {
        (void)hipStreamDestroy(tree->copy_stream);
        for (auto e : tree->in_free) (void)hipEventDestroy(e);
    }
    for (auto e : tree->gseq_free)
        if (e) (void)hipEventDestroy(e);
    for (auto e : tree->gseq_copied)
        if (e) (void)hipEventDestroy(e);
    for (auto h : tree->h_gseq)
        if (h) (void)hipHostFree(h);
    if (tree->h_hit_off) (void)hipHostFree(tree->h_hit_off);
    if (tree->h_hit_leaves) (void)hipHostFree(tree->h_hit_leaves);

    if (tree->h_pair_cursor) (void)hipHostFree(tree->h_pair_cursor);
    if (tree->hint_ev) (void)hipEventDestroy(tree->hint_ev);
    delete tree;
}

int pfq_query_batch_device(pfq_tree *tree, const uint8_t *d_seq, const uint64_t *d_offsets, uint64_t n_reads,
                           uint64_t total_bytes, float threshold, uint32_t flags, void *stream, pfq_hits *hits) {
    if (!tree || (n_reads && (!d_seq || !d_offsets))) return fail(PFQ_ERR_ARG, "null argument");
    PFQ_TRY(use_device(tree->device));
    return query_device(*tree, d_seq, d_offsets, n_reads, total_bytes, threshold, flags, (hipStream_t)stream, hits);
}

int pfq_query_batch(pfq_tree *tree, const uint8_t *seq, const uint64_t *offsets, uint64_t n_reads, float threshold,
                    uint32_t flags, pfq_hits *hits) {
    if (!tree || (n_reads && (!seq || !offsets))) return fail(PFQ_ERR_ARG, "null argument");
    PFQ_TRY(use_device(tree->device));
    pfq_tree &t = *tree;
    uint64_t total = n_reads ? offsets[n_reads] : 0;
    // Two input buffers and a copy stream of its own: the copy of this block runs while the kernels of the previous
    // block (which read the other buffer) are still at work.  A call that wants no hits returns once its kernels are
    // queued; counts are read by calls that synchronise (pfq_leaf_counts, pfq_last_stats, pfq_tree_close).
    if (!t.copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&t.copy_stream, hipStreamNonBlocking));
        for (auto &e : t.in_free) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const int slot = (t.in_slot ^= 1);
    if (t.in_used[slot]) HIP_TRY(hipEventSynchronize(t.in_free[slot]));  // the kernels that read this buffer are done
    DevBuf<uint8_t> &ds = slot ? t.d_seq2 : t.d_seq;
    DevBuf<uint64_t> &dof = slot ? t.d_off2 : t.d_off;
    HIP_TRY(ds.ensure(total + 16));
    HIP_TRY(dof.ensure(n_reads + 1));
    if (total) HIP_TRY(hipMemcpyAsync(ds.p, seq, total, hipMemcpyHostToDevice, t.copy_stream));
    if (n_reads) HIP_TRY(hipMemcpyAsync(dof.p, offsets, (n_reads + 1) * 8, hipMemcpyHostToDevice, t.copy_stream));
    HIP_TRY(hipStreamSynchronize(t.copy_stream));
    PFQ_TRY(query_device(t, ds.p, dof.p, n_reads, total, threshold, flags, nullptr, hits));
    HIP_TRY(hipEventRecord(t.in_free[slot], nullptr));
    t.in_used[slot] = true;
    if (flags & PFQ_WANT_HITS) HIP_TRY(hipStreamSynchronize(nullptr));
    return PFQ_OK;
}

int pfq_leaf_counts(pfq_tree *tree, const char *const **tax_ids, const uint64_t **counts, uint64_t *n_leaves) {
    if (!tree || !n_leaves) return fail(PFQ_ERR_ARG, "null argument");
    PFQ_TRY(use_device(tree->device));
    pfq_tree &t = *tree;
    PFQ_TRY(build_layout(t));
    PFQ_TRY(sync_counts_to_nodes(t));
    t.out_tax.clear();
    t.out_counts.clear();
    for (int32_t v : t.leaves) {
        t.out_tax.push_back(t.nodes[v].tax_id);
        t.out_counts.push_back(t.nodes[v].mapped_reads);
    }
    t.out_tax_ptr.clear();
    for (auto &s : t.out_tax) t.out_tax_ptr.push_back(s.c_str());
    if (tax_ids) *tax_ids = t.out_tax_ptr.data();
    if (counts) *counts = t.out_counts.data();
    *n_leaves = t.leaves.size();
    return PFQ_OK;
}

int pfq_save_leaf_counts(pfq_tree *tree, const char *csv_path) {
    if (!tree || !csv_path) return fail(PFQ_ERR_ARG, "null argument");
    const char *const *ids = nullptr;
    const uint64_t *cnt = nullptr;
    uint64_t n = 0;
    PFQ_TRY(pfq_leaf_counts(tree, &ids, &cnt, &n));
    FILE *f = fopen(csv_path, "wb");
    if (!f) return fail(PFQ_ERR_IO, std::string("cannot create ") + csv_path + ": " + strerror(errno));
    for (uint64_t i = 0; i < n; ++i)
        if (cnt[i] > 0) fprintf(f, "%s,%llu\n", ids[i], (unsigned long long)cnt[i]);  // query.rs:177-182
    if (fclose(f) != 0) return fail(PFQ_ERR_IO, "short write to CLASSIFICATION.csv");
    return PFQ_OK;
}

int pfq_leaf_counts_export(pfq_tree *tree, uint64_t *d_dst, void *stream) {
    if (!tree || !d_dst) return fail(PFQ_ERR_ARG, "null argument");
    PFQ_TRY(use_device(tree->device));
    PFQ_TRY(build_layout(*tree));
    if (!tree->leaves.empty())
        HIP_TRY(hipMemcpyAsync(d_dst, tree->d_counts.p, tree->leaves.size() * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PFQ_OK;
}
int pfq_leaf_counts_import(pfq_tree *tree, const uint64_t *d_src, void *stream) {
    if (!tree || !d_src) return fail(PFQ_ERR_ARG, "null argument");
    PFQ_TRY(use_device(tree->device));
    PFQ_TRY(build_layout(*tree));
    if (!tree->leaves.empty()) {
        HIP_TRY(hipMemcpyAsync(tree->d_counts.p, d_src, tree->leaves.size() * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        HIP_TRY(hipMemcpyAsync(tree->d_counts_base.p, d_src, tree->leaves.size() * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return PFQ_OK;
}
int pfq_leaf_counts_export_delta(pfq_tree *tree, uint64_t *d_dst, void *stream) {
    if (!tree || !d_dst) return fail(PFQ_ERR_ARG, "null argument");
    PFQ_TRY(use_device(tree->device));
    PFQ_TRY(build_layout(*tree));
    pfq::launch_counts_op(reinterpret_cast<unsigned long long *>(d_dst), tree->d_counts.p, tree->d_counts_base.p, (uint32_t)tree->leaves.size(), true,
                          (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return PFQ_OK;
}
int pfq_leaf_counts_import_delta(pfq_tree *tree, const uint64_t *d_src, void *stream) {
    if (!tree || !d_src) return fail(PFQ_ERR_ARG, "null argument");
    PFQ_TRY(use_device(tree->device));
    PFQ_TRY(build_layout(*tree));
    const uint32_t nl = (uint32_t)tree->leaves.size();
    pfq::launch_counts_op(tree->d_counts.p, tree->d_counts_base.p, reinterpret_cast<const unsigned long long *>(d_src), nl, false, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    if (nl) HIP_TRY(hipMemcpyAsync(tree->d_counts_base.p, tree->d_counts.p, (size_t)nl * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PFQ_OK;
}
// ---- several replicas behind one process: one RCCL all-reduce of the per-leaf counters -------------------------------
extern "C++" {
namespace {
struct Rccl {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};
Rccl &rccl() {  // loaded once per process; RCCL is a run-time dependency of multi-GPU runs only
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // (RTLD_LOCAL: an embedding framework may bring an RCCL of its own — PyTorch does — and the two must not see each
        // other's symbols)
        r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) r.handle = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) {
            r.error = std::string("cannot load librccl: ") + dlerror();
            return;
        }
        auto sym = [&](const char *name) {
            void *p = dlsym(r.handle, name);
            if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + name;
            return p;
        };
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return r;
}
thread_local uint32_t g_last_ranks = 0;
// The communicator of a device set is made on first use and kept while the process has a tree open (a caller may reduce
// after every block; ncclCommInitAll costs seconds).  Destroyed with the last tree — not at process exit, where the HIP
// runtime and other users of it (an embedding framework) are already tearing down.
std::mutex g_comm_mutex;
std::map<std::vector<int>, std::vector<ncclComm_t>> g_comm_cache;
}  // namespace
}  // extern "C++"
namespace {
void release_communicators() {
    std::lock_guard<std::mutex> lock(g_comm_mutex);
    if (g_comm_cache.empty()) return;
    Rccl &r = rccl();
    for (auto &kv : g_comm_cache)
        for (auto c : kv.second) (void)r.CommDestroy(c);
    g_comm_cache.clear();
}
}  // namespace

uint32_t pfq_last_allreduce_ranks(void) { return g_last_ranks; }

int pfq_device_count(int *n) {
    if (!n) return fail(PFQ_ERR_ARG, "null argument");
    *n = 0;
    if (hipGetDeviceCount(n) != hipSuccess || *n <= 0) {
        (void)hipGetLastError();
        *n = 0;
        return fail(PFQ_ERR_DEVICE, "no HIP device available (libpfq has no CPU fallback)");
    }
    return PFQ_OK;
}

int pfq_trees_allreduce_counts(pfq_tree *const *trees, uint32_t n_trees) {
    g_last_ranks = 0;
    if (!trees || n_trees == 0) return fail(PFQ_ERR_ARG, "null argument");
    for (uint32_t i = 0; i < n_trees; ++i)
        if (!trees[i]) return fail(PFQ_ERR_ARG, "null tree");
    // every replica: same leaf set, queued work finished
    std::map<int, std::vector<pfq_tree *>> by_dev;  // device -> replicas on it, the first one leads
    for (uint32_t i = 0; i < n_trees; ++i) {
        pfq_tree &t = *trees[i];
        for (uint32_t j = 0; j < i; ++j)
            if (trees[j] == trees[i]) return fail(PFQ_ERR_ARG, "the same tree listed twice");
        PFQ_TRY(use_device(t.device));
        PFQ_TRY(build_layout(t));
        HIP_TRY(hipDeviceSynchronize());
        if (t.leaves.size() != trees[0]->leaves.size())
            return fail(PFQ_ERR_ARG, "the trees are not replicas of one database: " + std::to_string(t.leaves.size()) + " vs " +
                                         std::to_string(trees[0]->leaves.size()) + " leaves");
        for (size_t l = 0; l < t.leaves.size(); ++l)
            if (t.nodes[t.leaves[l]].tax_id != trees[0]->nodes[trees[0]->leaves[l]].tax_id)
                return fail(PFQ_ERR_ARG, "the trees are not replicas of one database: leaf " + std::to_string(l) + " differs");
        by_dev[t.device].push_back(&t);
    }
    const uint32_t nl = (uint32_t)trees[0]->leaves.size();
    if (n_trees == 1 || nl == 0) return PFQ_OK;
    // What a replica adds to the job is what it counted since it was opened (or last reduced): counters - base.  The stored
    // mapped_reads of a database that was saved after a query are in every replica's base and must count once, as on one
    // device and in the reference (query.rs:143 accumulates on the loaded value).
    // (1) every replica's delta; replicas that share a device are added into the device's first replica
    for (auto &kv : by_dev) {
        HIP_TRY(hipSetDevice(kv.first));
        for (size_t r = 0; r < kv.second.size(); ++r) {
            pfq_tree &t = *kv.second[r];
            pfq::launch_counts_op(t.d_counts_delta.p, t.d_counts.p, t.d_counts_base.p, nl, true, nullptr);
            if (r) pfq::launch_counts_op(kv.second[0]->d_counts_delta.p, kv.second[0]->d_counts_delta.p, t.d_counts_delta.p, nl, false, nullptr);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
    }
    // (2) one all-reduce (sum, u64[n_leaves]) over RCCL across the devices, in place in each device's first replica.
    // Replicas that all share one device need no communicator (loading librccl and ncclCommInitAll cost seconds);
    // PFQ_RCCL_ALWAYS=1 makes a one-rank communicator anyway, so that the RCCL path can be exercised on a one-GPU box.
    // The communicator of a device set is made once per process and kept (a caller may reduce after every block).
    const char *always = getenv("PFQ_RCCL_ALWAYS");
    if (by_dev.size() > 1 || (always && atoi(always) != 0)) {
        Rccl &r = rccl();
        if (!r.error.empty()) return fail(PFQ_ERR_DEVICE, r.error);
        std::vector<int> devs;
        for (auto &kv : by_dev) devs.push_back(kv.first);
        auto &comm_cache = g_comm_cache;
        std::lock_guard<std::mutex> lock(g_comm_mutex);  // (one collective of this process at a time)
        auto it = comm_cache.find(devs);
        if (it == comm_cache.end()) {
            std::vector<ncclComm_t> fresh(devs.size());
            const ncclResult_t rc0 = r.CommInitAll(fresh.data(), (int)devs.size(), devs.data());
            if (rc0 != ncclSuccess) return fail(PFQ_ERR_DEVICE, std::string("ncclCommInitAll: ") + r.GetErrorString(rc0));
            it = comm_cache.emplace(devs, std::move(fresh)).first;
        }
        std::vector<ncclComm_t> &comms = it->second;
        ncclResult_t rc = r.GroupStart();
        for (size_t i = 0; i < devs.size() && rc == ncclSuccess; ++i) {
            (void)hipSetDevice(devs[i]);
            unsigned long long *buf = by_dev[devs[i]][0]->d_counts_delta.p;
            rc = r.AllReduce(buf, buf, nl, ncclUint64, ncclSum, comms[i], nullptr);
        }
        const ncclResult_t rc_end = r.GroupEnd();
        if (rc == ncclSuccess) rc = rc_end;
        hipError_t he = hipSuccess;
        for (size_t i = 0; i < devs.size(); ++i) {
            (void)hipSetDevice(devs[i]);
            const hipError_t e = hipDeviceSynchronize();
            if (he == hipSuccess) he = e;
        }
        if (rc != ncclSuccess || he != hipSuccess) {  // a communicator that failed is not reused
            for (auto c : comms) (void)r.CommDestroy(c);
            comm_cache.erase(it);
        }
        if (rc != ncclSuccess) return fail(PFQ_ERR_DEVICE, std::string("ncclAllReduce: ") + r.GetErrorString(rc));
        if (he != hipSuccess) return fail(PFQ_ERR_DEVICE, std::string("all-reduce of the leaf counters: ") + hipGetErrorString(he));
        g_last_ranks = (uint32_t)devs.size();
    }
    // (3) every replica: counters = its base + the job's delta, and that is its new base (a second call changes nothing)
    for (auto &kv : by_dev) {
        HIP_TRY(hipSetDevice(kv.first));
        const unsigned long long *total = kv.second[0]->d_counts_delta.p;
        for (size_t q = 0; q < kv.second.size(); ++q) {
            pfq_tree &t = *kv.second[q];
            pfq::launch_counts_op(t.d_counts.p, t.d_counts_base.p, total, nl, false, nullptr);
            HIP_TRY(hipMemcpyAsync(t.d_counts_base.p, t.d_counts.p, (size_t)nl * 8, hipMemcpyDeviceToDevice, nullptr));
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
    }
    return PFQ_OK;
}

int pfq_leaf_counts_reset(pfq_tree *tree) {
    if (!tree) return fail(PFQ_ERR_ARG, "null argument");
    PFQ_TRY(use_device(tree->device));
    PFQ_TRY(build_layout(*tree));
    HIP_TRY(hipDeviceSynchronize());
    if (!tree->leaves.empty()) {
        HIP_TRY(hipMemset(tree->d_counts.p, 0, tree->leaves.size() * 8));
        HIP_TRY(hipMemset(tree->d_counts_base.p, 0, tree->leaves.size() * 8));
    }
    for (auto &nd : tree->nodes) nd.mapped_reads = nd.base_reads = 0;
    return PFQ_OK;
}

int pfq_last_stats(pfq_tree *tree, pfq_stats *out) {
    if (!tree || !out) return fail(PFQ_ERR_ARG, "null argument");
    PFQ_TRY(use_device(tree->device));
    pfq_tree &t = *tree;
    memset(out, 0, sizeof *out);
    if (!t.d_stats.p) return PFQ_OK;
    HIP_TRY(hipStreamSynchronize(t.last_stream));
    unsigned long long h[pfq::ST_N];
    HIP_TRY(hipMemcpy(h, t.d_stats.p, sizeof h, hipMemcpyDeviceToHost));
    out->n_reads = t.last_n_reads;
    out->n_candidates = h[pfq::ST_CANDIDATES];
    out->n_hits = h[pfq::ST_HITS];
    out->n_allhit_reads = h[pfq::ST_ALLHIT];
    out->algorithmic_bytes = h[pfq::ST_ALG_BYTES];
    out->path = t.last_path;
    out->n_slices = t.last_slices;
    out->tile_mode = t.last_tile_mode;
    if (t.d_cursors.p) {
        unsigned long long c[4];
        HIP_TRY(hipMemcpy(c, t.d_cursors.p, sizeof c, hipMemcpyDeviceToHost));
        out->n_chunks = (uint32_t)c[3];
        out->n_fallback_pairs = (uint32_t)(c[3] >> 32);
        out->tile_entries = c[2];
        if (t.last_tile_mode && t.hint_entry_cap) {
            out->tile_passes_launched = t.last_passes;
            out->tile_passes_needed = (uint32_t)std::max<uint64_t>(1, (c[2] + t.hint_entry_cap - 1) / t.hint_entry_cap);
        }
    }
    out->leaf_groups = t.last_leaf_groups;
    out->coarse_cols = t.last_coarse_cols;
    out->coarse_probes = t.last_coarse_probes;
    out->group_reads = h[pfq::ST_LISTED];
    return PFQ_OK;
}
int pfq_profile_begin(pfq_tree *tree, uint32_t max_calls) {
    if (!tree) return fail(PFQ_ERR_ARG, "null argument");
    PFQ_TRY(use_device(tree->device));
    pfq_tree &t = *tree;
    while (t.prof_ev.size() < PROF_EV * (size_t)max_calls) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        t.prof_ev.push_back(e);
    }
    t.prof_cap = max_calls;
    t.prof_used = 0;
    t.prof_bucketed.assign(max_calls, 0);
    return PFQ_OK;
}
int pfq_profile_end(pfq_tree *tree, pfq_profile *out) {
    if (!tree || !out) return fail(PFQ_ERR_ARG, "null argument");
    PFQ_TRY(use_device(tree->device));
    pfq_tree &t = *tree;
    memset(out, 0, sizeof *out);
    HIP_TRY(hipStreamSynchronize(t.last_stream));
    for (size_t c = 0; c < t.prof_used; ++c) {
        hipEvent_t *ev = &t.prof_ev[PROF_EV * c];
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[1]));
        out->classify_ms += ms;
        if (t.prof_bucketed[c]) {
            HIP_TRY(hipEventElapsedTime(&ms, ev[1], ev[2]));
            out->bucket_ms += ms;
            HIP_TRY(hipEventElapsedTime(&ms, ev[2], ev[3]));
            out->bin_ms += ms;
            HIP_TRY(hipEventElapsedTime(&ms, ev[3], ev[4]));
            out->test_ms += ms;
            HIP_TRY(hipEventElapsedTime(&ms, ev[4], ev[5]));
            out->verify_ms += ms;
            HIP_TRY(hipEventElapsedTime(&ms, ev[5], ev[6]));
            out->finalize_ms += ms;
        }
    }
    out->calls = t.prof_used;
    t.prof_cap = t.prof_used = 0;
    return PFQ_OK;
}
int pfq_set_option(pfq_tree *tree, const char *name, const char *value) {
    if (!tree || !name) return fail(PFQ_ERR_ARG, "null argument");
    if (!set_knob(tree->knobs, name, value)) return fail(PFQ_ERR_ARG, std::string("unknown option ") + name);
    // knobs of the device layout (column groups, coarse level): the layout is rebuilt before the next use
    if (!strcmp(name, "PFQ_COARSE") || !strcmp(name, "PFQ_COARSE_COLS") || !strcmp(name, "PFQ_GROUP_LOG2") || !strcmp(name, "PFQ_COARSE_MIN_LEAVES")) {
        if (tree->layout_valid) {
            PFQ_TRY(use_device(tree->device));
            PFQ_TRY(sync_counts_to_nodes(*tree));
            tree->layout_valid = false;
        }
    }
    return PFQ_OK;
}
int pfq_set_path(pfq_tree *tree, int path) {
    if (!tree || path < -1 || path > 1) return fail(PFQ_ERR_ARG, "bad argument");
    tree->force_path = path;
    return PFQ_OK;
}

int pfq_debug_kmer_indices(pfq_tree *tree, const uint8_t *seq, uint64_t len, uint64_t *out_idx, uint64_t *n_kmers) {
    if (!tree || !n_kmers || (len && !seq)) return fail(PFQ_ERR_ARG, "null argument");
    PFQ_TRY(use_device(tree->device));
    pfq_tree &t = *tree;
    uint64_t n = (t.kmer_size >= 1 && len >= t.kmer_size) ? len - t.kmer_size + 1 : 0;
    *n_kmers = n;
    if (!n || !out_idx) return PFQ_OK;
    DevBuf<uint8_t> d_s;
    DevBuf<uint64_t> d_o;
    HIP_TRY(d_s.ensure(len + 16));
    HIP_TRY(d_o.ensure(n * t.num_hashes));
    HIP_TRY(hipMemcpy(d_s.p, seq, len, hipMemcpyHostToDevice));
    pfq::launch_debug_indices(t.hp, d_s.p, len, d_o.p, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out_idx, d_o.p, n * t.num_hashes * 8, hipMemcpyDeviceToHost));
    return PFQ_OK;
}

int pfq_debug_node_filter(pfq_tree *tree, uint64_t node, uint64_t *out_words, uint64_t n_words) {
    if (!tree || !out_words) return fail(PFQ_ERR_ARG, "null argument");
    PFQ_TRY(use_device(tree->device));
    PFQ_TRY(finish_topology(*tree));
    if (node >= tree->nodes.size() || n_words != tree->n_words) return fail(PFQ_ERR_ARG, "node / n_words out of range");
    HIP_TRY(hipMemcpy(out_words, tree->d_bits.p + (uint64_t)tree->nodes[node].filter * tree->n_words, n_words * 8,
                      hipMemcpyDeviceToHost));
    return PFQ_OK;
}

int pfq_synth_genomes_device(uint8_t *d_out, uint64_t n_genomes, uint64_t genome_len, uint64_t seed_base, void *stream) {
    if (!d_out) return fail(PFQ_ERR_ARG, "null argument");
    pfq::launch_synth_genomes(d_out, n_genomes, genome_len, seed_base, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return PFQ_OK;
}
int pfq_synth_reads_device(uint8_t *d_out, uint64_t first_read, uint64_t n_reads, uint64_t read_len, const uint8_t *d_genomes,
                           uint64_t genome_len, uint64_t n_genomes, uint64_t seed, void *stream) {
    if (!d_out) return fail(PFQ_ERR_ARG, "null argument");
    pfq::launch_synth_reads(d_out, first_read, n_reads, read_len, d_genomes, genome_len, n_genomes, seed, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return PFQ_OK;
}

}  // extern "C"
