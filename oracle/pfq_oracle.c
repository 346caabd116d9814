/*
 * pfq_oracle.c — CPU ORACLE for the PhageFilter read-classification path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is a plain-C restatement of the reference's
 * (Dreycey/PhageFilter, Rust) `phage_filter query` hot path.  It is the checker that the HIP
 * path is compared against; it is never the thing shipped or measured as the product.  Only
 * `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may load it.
 *
 * PARITY PINNING.  The reference path is Rust and cannot be compiled in this image (no
 * cargo/rustc, no vendored crates), so there is no `oracle/_ref`.  The oracle is pinned by
 *   (1) every exact known-answer / relational fixture the reference's own unit tests hold for the
 *       path (hash_iter formula hash_iter.rs:66-101; get_kmers / get_lex_less KATs
 *       file_parser.rs:380-407; threshold semantics query.rs:267-290; accumulation
 *       query.rs:356-380; Hamming KATs bloom_filter.rs:378-391) — see tests/test_oracle_*.py;
 *   (2) upstream `rustc-hash` 2.1 unit-test values (the hash lives in an un-vendored crate,
 *       Cargo.toml:20) — tests/golden/fxhash_kat.json.
 * The on-disk layout produced by `bitvec 1.0.1` + `bincode 1.3.3` is restated from their
 * published formats and is NOT pinned by any reference fixture ("parity unpinned" for the file
 * layout only).
 *
 * Every function cites the reference file:line it follows (paths relative to the reference
 * repository root).
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* rustc-hash 2.1.x FxHasher (64-bit) — external crate used at src/bloom_filter/hasher.rs:2,13 */
/* ------------------------------------------------------------------------------------------ */
#define FX_K 0xf1357aea2e62a9c5ull
#define FX_SEED1 0x243f6a8885a308d3ull
#define FX_SEED2 0x13198a2e03707344ull
#define FX_PTZC 0xa4093822299f31d0ull /* PREVENT_TRIVIAL_ZERO_COLLAPSE */

static inline uint64_t le64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint64_t le32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

/* multiply_mix: full 64x64->128 product, lo XOR hi. */
static inline uint64_t fx_mm(uint64_t x, uint64_t y) {
    __uint128_t p = (__uint128_t)x * (__uint128_t)y;
    return (uint64_t)p ^ (uint64_t)(p >> 64);
}

/* rustc-hash `hash_bytes`. */
uint64_t orc_fx_hash_bytes(const uint8_t *b, uint64_t n) {
    uint64_t s0 = FX_SEED1, s1 = FX_SEED2;
    if (n <= 16) {
        if (n >= 8) {
            s0 ^= le64(b);
            s1 ^= le64(b + n - 8);
        } else if (n >= 4) {
            s0 ^= le32(b);
            s1 ^= le32(b + n - 4);
        } else if (n > 0) {
            uint64_t lo = b[0], mid = b[n / 2], hi = b[n - 1];
            s0 ^= lo;
            s1 ^= (hi << 8) | mid;
        }
    } else {
        uint64_t off = 0;
        while (off < n - 16) {
            uint64_t x = le64(b + off), y = le64(b + off + 8);
            uint64_t t = fx_mm(s0 ^ x, FX_PTZC ^ y);
            s0 = s1;
            s1 = t;
            off += 16;
        }
        s0 ^= le64(b + n - 16);
        s1 ^= le64(b + n - 8);
    }
    return fx_mm(s0, s1) ^ n;
}

/* FxHasher::add_to_hash / finish. */
static inline uint64_t fx_add(uint64_t st, uint64_t v) { return (st + v) * FX_K; }
static inline uint64_t fx_finish(uint64_t st) { return rotl64(st, 26); }

/* KAT helpers: `FxHasher::default()`, one write, `finish()`. */
uint64_t orc_fx_finish_write_bytes(const uint8_t *b, uint64_t n) {
    return fx_finish(fx_add(0, orc_fx_hash_bytes(b, n)));
}
uint64_t orc_fx_finish_write_u64(uint64_t v) { return fx_finish(fx_add(0, v)); }

/*
 * HashSeed::build_hasher + hash_one(&Vec<u8>)            (hasher.rs:12-21, hash_iter.rs:37-38)
 *   FxHasher::default(); write_usize(seed);               hasher.rs:16-18
 *   <[u8] as Hash>::hash: write_length_prefix(len) (= write_usize), write(bytes); finish().
 */
uint64_t orc_seeded_hash(uint64_t seed, const uint8_t *item, uint64_t len) {
    uint64_t st = 0;
    st = fx_add(st, seed);
    st = fx_add(st, len);
    st = fx_add(st, orc_fx_hash_bytes(item, len));
    return fx_finish(st);
}

/* HashIter::next (hash_iter.rs:13-27): probe i = h1 / h2 / (h1+i)*h2 wrapping. */
static inline uint64_t probe_value(uint64_t h1, uint64_t h2, uint32_t i) {
    if (i == 0) return h1;
    if (i == 1) return h2;
    return (h1 + (uint64_t)i) * h2;
}

/* The bit indices `contains`/`insert` touch (bloom_filter.rs:291-332): probe % nbits. */
void orc_probe_indices(uint64_t seed1, uint64_t seed2, uint32_t num_hashes, uint64_t nbits,
                       const uint8_t *item, uint64_t len, uint64_t *out_idx) {
    uint64_t h1 = orc_seeded_hash(seed1, item, len);
    uint64_t h2 = orc_seeded_hash(seed2, item, len);
    for (uint32_t i = 0; i < num_hashes; ++i) out_idx[i] = probe_value(h1, h2, i) % nbits;
}

/* ------------------------------------------------------------------------------------------ */
/* Canonical k-mers — file_parser.rs:114-148; bio::alphabets::dna::revcomp (bio 2.2.0)          */
/* ------------------------------------------------------------------------------------------ */
static uint8_t COMP[256];
static pthread_once_t comp_once = PTHREAD_ONCE_INIT;
static void comp_init(void) {
    /* bio::alphabets::dna: identity, then the IUPAC pairs and their lowercase forms. */
    static const char a[] = "AGCTYRWSKMDVHBN", b[] = "TCGARYWSMKHBDVN";
    for (int i = 0; i < 256; ++i) COMP[i] = (uint8_t)i;
    for (int i = 0; a[i]; ++i) {
        COMP[(uint8_t)a[i]] = (uint8_t)b[i];
        COMP[(uint8_t)a[i] + 32] = (uint8_t)(b[i] + 32);
    }
}
void orc_complement_table(uint8_t *out256) {
    pthread_once(&comp_once, comp_init);
    memcpy(out256, COMP, 256);
}
void orc_revcomp(const uint8_t *in, uint64_t n, uint8_t *out) {
    pthread_once(&comp_once, comp_init);
    for (uint64_t i = 0; i < n; ++i) out[i] = COMP[in[n - 1 - i]];
}
/* get_lex_less (file_parser.rs:114-121): min(kmer, revcomp) bytewise; tie -> forward. */
void orc_get_lex_less(const uint8_t *kmer, uint64_t k, uint8_t *out) {
    uint8_t rc_small[64];
    uint8_t *rc = k <= sizeof rc_small ? rc_small : (uint8_t *)malloc(k);
    orc_revcomp(kmer, k, rc);
    memcpy(out, memcmp(kmer, rc, k) <= 0 ? kmer : rc, k);
    if (rc != rc_small) free(rc);
}
/* get_kmers (file_parser.rs:135-148): [] if k > len or k == 0, else len-k+1 canonical k-mers. */
uint64_t orc_kmer_count(uint64_t len, uint64_t k) { return (k == 0 || k > len) ? 0 : len - k + 1; }
uint64_t orc_get_kmers(const uint8_t *seq, uint64_t len, uint64_t k, uint8_t *out /* n*k */) {
    uint64_t n = orc_kmer_count(len, k);
    for (uint64_t i = 0; i < n; ++i) orc_get_lex_less(seq + i, k, out + i * k);
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* Bloom filter — bloom_filter.rs:84-93 (BitVec<usize, Lsb0>), :142-149, :275-357               */
/* ------------------------------------------------------------------------------------------ */
static inline int bit_get(const uint64_t *w, uint64_t idx) { return (int)((w[idx >> 6] >> (idx & 63)) & 1); }
static inline void bit_set(uint64_t *w, uint64_t idx) { w[idx >> 6] |= 1ull << (idx & 63); }

/* needed_bits / optimal_num_hashes (bloom_filter.rs:342-357), f32 arithmetic as written. */
uint64_t orc_needed_bits(float false_pos_rate, uint32_t num_items) {
    const float ln2 = 0.693147180559945309417232121458176568f; /* core::f32::consts::LN_2 */
    float ln22 = ln2 * ln2;
    float v = roundf((float)num_items * (logf(1.0f / false_pos_rate) / ln22));
    return v <= 0.0f ? 0 : (uint64_t)v;
}
uint32_t orc_optimal_num_hashes(uint64_t num_bits, uint32_t num_items) {
    const float ln2 = 0.693147180559945309417232121458176568f;
    float v = roundf((float)num_bits / (float)num_items * ln2);
    uint32_t h = v <= 0.0f ? 0 : (v >= 4294967296.0f ? 0xffffffffu : (uint32_t)v);
    if (h < 2) h = 2;
    if (h > 200) h = 200;
    return h;
}
/* DistanceChecker::distance (bloom_filter.rs:142-149). */
uint64_t orc_distance(const uint64_t *a, const uint64_t *b, uint64_t n_words) {
    uint64_t d = 0;
    for (uint64_t i = 0; i < n_words; ++i) d += (uint64_t)__builtin_popcountll(a[i] ^ b[i]);
    return d;
}
/* union (bloom_filter.rs:275-278). */
void orc_union(uint64_t *dst, const uint64_t *src, uint64_t n_words) {
    for (uint64_t i = 0; i < n_words; ++i) dst[i] |= src[i];
}
/* ASMS::insert (bloom_filter.rs:291-307); returns !contained-before like the reference. */
int orc_bf_insert(uint64_t *bits, uint64_t nbits, uint32_t num_hashes, uint64_t seed1, uint64_t seed2,
                  const uint8_t *item, uint64_t len) {
    uint64_t h1 = orc_seeded_hash(seed1, item, len), h2 = orc_seeded_hash(seed2, item, len);
    int contained = 1;
    for (uint32_t i = 0; i < num_hashes; ++i) {
        uint64_t idx = probe_value(h1, h2, i) % nbits;
        contained = bit_get(bits, idx);
        bit_set(bits, idx);
    }
    return !contained;
}
/* ASMS::contains (bloom_filter.rs:312-332): early exit at the first 0 bit. `probes` counts the
 * bit reads actually executed (reference-semantics work, SURVEY §8d P_ref). */
static inline int bf_contains(const uint64_t *bits, uint64_t nbits, uint32_t num_hashes, uint64_t seed1,
                              uint64_t seed2, const uint8_t *item, uint64_t len, uint64_t *probes) {
    uint64_t h1 = orc_seeded_hash(seed1, item, len), h2 = orc_seeded_hash(seed2, item, len);
    for (uint32_t i = 0; i < num_hashes; ++i) {
        uint64_t idx = probe_value(h1, h2, i) % nbits;
        ++*probes;
        if (!bit_get(bits, idx)) return 0;
    }
    return 1;
}
int orc_bf_contains(const uint64_t *bits, uint64_t nbits, uint32_t num_hashes, uint64_t seed1, uint64_t seed2,
                    const uint8_t *item, uint64_t len) {
    uint64_t p = 0;
    return bf_contains(bits, nbits, num_hashes, seed1, seed2, item, len, &p);
}
/* init_leaf_node's serial insert of all k-mers of a genome (bloom_tree.rs:154-168). */
void orc_bf_insert_sequence(uint64_t *bits, uint64_t nbits, uint32_t num_hashes, uint64_t seed1, uint64_t seed2,
                            const uint8_t *seq, uint64_t len, uint64_t k) {
    uint64_t n = orc_kmer_count(len, k);
    uint8_t *canon = (uint8_t *)malloc(k ? k : 1);
    for (uint64_t i = 0; i < n; ++i) {
        orc_get_lex_less(seq + i, k, canon);
        orc_bf_insert(bits, nbits, num_hashes, seed1, seed2, canon, k);
    }
    free(canon);
}

/* ------------------------------------------------------------------------------------------ */
/* query — query.rs:38-158                                                                     */
/* ------------------------------------------------------------------------------------------ */
/* `(threshold * n as f32).ceil() as usize` (query.rs:48): IEEE f32 multiply, ceil, saturating cast. */
uint64_t orc_need(float threshold, uint64_t n_kmers) {
    volatile float prod = threshold * (float)n_kmers; /* volatile: no contraction / excess precision */
    float c = ceilf(prod);
    if (!(c > 0.0f)) return 0; /* NaN and <= 0 saturate to 0 */
    if (c >= 18446744073709551616.0f) return UINT64_MAX;
    return (uint64_t)c;
}

typedef struct {
    /* topology (BloomNode, bloom_tree.rs:50-61) flattened; -1 = None */
    int64_t n_nodes, root;
    const int64_t *left, *right;
    const int64_t *filter; /* node -> filter row (nodes sharing a .bf path share a row) */
    /* filters (BloomFilter, bloom_filter.rs:84-93): row-major [n_filters][n_words] u64, Lsb0 */
    const uint64_t *bits;
    uint64_t n_words, nbits;
    uint32_t num_hashes;
    uint64_t seed1, seed2, kmer_size;
} orc_tree;

typedef struct {
    const orc_tree *t;
    const uint8_t *seq;
    const uint64_t *off;
    float threshold;
    int faithful; /* 1: re-hash every k-mer at every node like bloom_filter.rs:313-318 */
    /* per-thread slice */
    uint64_t r0, r1;
    uint64_t *mapped;      /* [n_nodes] private accumulator (mapped_reads, query.rs:143) */
    uint64_t probes;       /* P_ref */
    uint64_t *hit_pairs;   /* optional (read, node) pairs */
    uint64_t hit_cap, n_hits;
    /* k-mers of the slice, materialised before the query as file_parser.rs:191-224 does */
    uint8_t *kmers;        /* canonical bytes */
    uint64_t *kmer_off;    /* [r1-r0+1] in k-mers */
    uint64_t *idx;         /* fast mode: [total_kmers * num_hashes] precomputed indices */
} orc_job;

/* query_passes (query.rs:38-49): count(contains) >= need; no early exit across k-mers. */
static int query_passes(orc_job *j, int64_t node, uint64_t local_read) {
    const orc_tree *t = j->t;
    const uint64_t *bits = t->bits + (uint64_t)t->filter[node] * t->n_words;
    uint64_t k0 = j->kmer_off[local_read], k1 = j->kmer_off[local_read + 1], n = k1 - k0, matches = 0;
    if (j->faithful) {
        for (uint64_t q = k0; q < k1; ++q)
            matches += (uint64_t)bf_contains(bits, t->nbits, t->num_hashes, t->seed1, t->seed2,
                                             j->kmers + q * t->kmer_size, t->kmer_size, &j->probes);
    } else {
        for (uint64_t q = k0; q < k1; ++q) {
            const uint64_t *ix = j->idx + q * t->num_hashes;
            int ok = 1;
            for (uint32_t i = 0; i < t->num_hashes; ++i) {
                ++j->probes;
                if (!bit_get(bits, ix[i])) { ok = 0; break; }
            }
            matches += (uint64_t)ok;
        }
    }
    return matches >= orc_need(j->threshold, n);
}

/* _query_batch (query.rs:99-158): filter the read set at this node; recurse left then right with the
 * survivors if internal and survivors != {}; at a leaf mapped_reads += |pass| and record the pairs. */
static void query_batch_rec(orc_job *j, int64_t node, const uint64_t *reads, uint64_t n_reads) {
    const orc_tree *t = j->t;
    uint64_t *pass = (uint64_t *)malloc((n_reads ? n_reads : 1) * sizeof(uint64_t)), np = 0;
    for (uint64_t i = 0; i < n_reads; ++i)
        if (query_passes(j, node, reads[i])) pass[np++] = reads[i];
    int is_leaf = t->left[node] < 0 && t->right[node] < 0; /* is_leafnode, bloom_tree.rs:416-418 */
    if (!is_leaf) {
        if (np) {
            if (t->left[node] >= 0) query_batch_rec(j, t->left[node], pass, np);
            if (t->right[node] >= 0) query_batch_rec(j, t->right[node], pass, np);
        }
    } else {
        j->mapped[node] += np;
        for (uint64_t i = 0; i < np; ++i) {
            if (j->hit_pairs && j->n_hits < j->hit_cap) {
                j->hit_pairs[2 * j->n_hits] = j->r0 + pass[i];
                j->hit_pairs[2 * j->n_hits + 1] = (uint64_t)node;
            }
            ++j->n_hits;
        }
    }
    free(pass);
}

static void *job_main(void *arg) {
    orc_job *j = (orc_job *)arg;
    const orc_tree *t = j->t;
    uint64_t nr = j->r1 - j->r0, k = t->kmer_size;
    /* materialise k-mers (file_parser.rs:135-148) — outside the reference's query_batch */
    j->kmer_off = (uint64_t *)malloc((nr + 1) * sizeof(uint64_t));
    uint64_t total = 0;
    for (uint64_t r = 0; r < nr; ++r) {
        j->kmer_off[r] = total;
        total += orc_kmer_count(j->off[j->r0 + r + 1] - j->off[j->r0 + r], k);
    }
    j->kmer_off[nr] = total;
    j->kmers = (uint8_t *)malloc(total * k + 1);
    for (uint64_t r = 0; r < nr; ++r)
        orc_get_kmers(j->seq + j->off[j->r0 + r], j->off[j->r0 + r + 1] - j->off[j->r0 + r], k,
                      j->kmers + j->kmer_off[r] * k);
    j->idx = NULL;
    if (!j->faithful) {
        j->idx = (uint64_t *)malloc((total * t->num_hashes + 1) * sizeof(uint64_t));
        for (uint64_t q = 0; q < total; ++q)
            orc_probe_indices(t->seed1, t->seed2, t->num_hashes, t->nbits, j->kmers + q * k, k,
                              j->idx + q * t->num_hashes);
    }
    return NULL;
}
static void *job_query(void *arg) {
    orc_job *j = (orc_job *)arg;
    uint64_t nr = j->r1 - j->r0;
    uint64_t *all = (uint64_t *)malloc((nr ? nr : 1) * sizeof(uint64_t));
    for (uint64_t r = 0; r < nr; ++r) all[r] = r;
    /* query_batch (query.rs:66-82): root.map(|root| _query_batch(...)) over the whole block */
    if (j->t->root >= 0) query_batch_rec(j, j->t->root, all, nr);
    free(all);
    return NULL;
}

/*
 * Classify reads [0, n_reads) against the tree.
 *   mapped_reads[n_nodes]  += per-node count (leaves only), accumulating like query.rs:143.
 *   hit_pairs (optional)    (read index, leaf node index) pairs, capacity hit_cap pairs;
 *                           *n_hits_out gets the total number found (may exceed hit_cap).
 *   probes_out              reference-semantics probe count P_ref.
 *   query_seconds_out       wall time of the query phase only (k-mers already materialised).
 * Threads split the block into contiguous slices; per-read results do not depend on the split.
 */
#include <time.h>
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

int orc_query_batch(const orc_tree *t, const uint8_t *seq, const uint64_t *off, uint64_t n_reads, float threshold,
                    int faithful, int n_threads, uint64_t *mapped_reads, uint64_t *hit_pairs, uint64_t hit_cap,
                    uint64_t *n_hits_out, uint64_t *probes_out, double *query_seconds_out) {
    if (n_threads < 1) n_threads = 1;
    if ((uint64_t)n_threads > n_reads) n_threads = n_reads ? (int)n_reads : 1;
    orc_job *jobs = (orc_job *)calloc((size_t)n_threads, sizeof(orc_job));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int i = 0; i < n_threads; ++i) {
        orc_job *j = &jobs[i];
        j->t = t; j->seq = seq; j->off = off; j->threshold = threshold; j->faithful = faithful;
        j->r0 = n_reads * (uint64_t)i / (uint64_t)n_threads;
        j->r1 = n_reads * (uint64_t)(i + 1) / (uint64_t)n_threads;
        j->mapped = (uint64_t *)calloc((size_t)t->n_nodes, sizeof(uint64_t));
        if (hit_pairs) { j->hit_cap = hit_cap; j->hit_pairs = (uint64_t *)malloc((hit_cap ? hit_cap : 1) * 16); }
    }
    for (int i = 0; i < n_threads; ++i) pthread_create(&th[i], NULL, job_main, &jobs[i]);
    for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
    double t0 = now_s();
    for (int i = 0; i < n_threads; ++i) pthread_create(&th[i], NULL, job_query, &jobs[i]);
    for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
    double t1 = now_s();
    uint64_t nh = 0, probes = 0;
    for (int i = 0; i < n_threads; ++i) {
        orc_job *j = &jobs[i];
        for (int64_t v = 0; v < t->n_nodes; ++v) mapped_reads[v] += j->mapped[v];
        if (hit_pairs) {
            uint64_t stored = j->n_hits < j->hit_cap ? j->n_hits : j->hit_cap;
            for (uint64_t h = 0; h < stored && nh + h < hit_cap; ++h) {
                hit_pairs[2 * (nh + h)] = j->hit_pairs[2 * h];
                hit_pairs[2 * (nh + h) + 1] = j->hit_pairs[2 * h + 1];
            }
        }
        nh += j->n_hits;
        probes += j->probes;
        free(j->mapped); free(j->hit_pairs); free(j->kmers); free(j->kmer_off); free(j->idx);
    }
    if (n_hits_out) *n_hits_out = nh;
    if (probes_out) *probes_out = probes;
    if (query_seconds_out) *query_seconds_out = t1 - t0;
    free(jobs); free(th);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Synthetic workload generator: counter-based splitmix64, so any shard can         */
/* regenerate any genome / read.  Not reference code — the bench's data definition, restated    */
/* on the device in phagefilter_amd/csrc/pfq_kernels.hip (parity-tested against this).          */
/* ------------------------------------------------------------------------------------------ */
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static inline uint64_t rnd(uint64_t seed, uint64_t i) { return splitmix64(splitmix64(seed) + i); }
static const char ACGT[4] = {'A', 'C', 'G', 'T'};

void orc_synth_genome(uint64_t seed, uint64_t len, uint8_t *out) {
    for (uint64_t j = 0; j < len; ++j) out[j] = (uint8_t)ACGT[(rnd(seed, j >> 5) >> (2 * (j & 31))) & 3];
}
/* Read r: word0 bit0 = positive?, bit1 = reverse strand, bits 8.. = leaf; word1 = offset.
 * Negative reads: i.i.d. uniform ACGT from words 2.. */
void orc_synth_reads(uint64_t seed, uint64_t first, uint64_t count, uint64_t read_len, const uint8_t *genomes,
                     uint64_t genome_len, uint64_t n_genomes, uint8_t *out) {
    pthread_once(&comp_once, comp_init);
    for (uint64_t c = 0; c < count; ++c) {
        uint64_t r = first + c, w0 = rnd(seed, 8 * r), w1 = rnd(seed, 8 * r + 1);
        uint8_t *dst = out + c * read_len;
        if ((w0 & 1) && n_genomes && genome_len >= read_len) {
            uint64_t g = (w0 >> 8) % n_genomes, o = w1 % (genome_len - read_len + 1);
            const uint8_t *src = genomes + g * genome_len + o;
            if (w0 & 2) for (uint64_t j = 0; j < read_len; ++j) dst[j] = COMP[src[read_len - 1 - j]];
            else memcpy(dst, src, read_len);
        } else {
            for (uint64_t j = 0; j < read_len; ++j)
                dst[j] = (uint8_t)ACGT[(rnd(seed ^ 0xA5A5A5A5A5A5A5A5ull, r * 64 + (j >> 5)) >> (2 * (j & 31))) & 3];
        }
    }
}
