// pfq_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) of the classification path.
//
//   k_classify   K1+K2+K3 in one launch: one wave per read.  The frontier over the leaf level of the SBT is a
//                bitmask spread over the lanes (one dword = 32 leaves per lane); each probe of the sliced matrix S
//                ANDs one 128-B line (the same Bloom bit for 1024 leaves) into it, `__ballot` tests it for
//                emptiness (subtree/leaf pruning), and surviving leaves get the full certificate of
//                query_passes (query.rs:38-49): lanes = k-mers, `__ballot` of "all probed bits set",
//                `__popcll` accumulate, compare with need.  DEFER variant hands survivors to the bucketed pass.
//   k_verify_rec the certificate for survivors bucketed by leaf: every XCD keeps one slice of the current
//                leaf's node-major filter hot in its own L2 and checks only the probes that fall in its slice;
//                indices come from the probe records k_classify<DEFER> wrote (k_verify: same, re-hashing).
//   k_insert / k_union / k_superset / k_transpose   database construction on the device.
//
// Reference semantics: query.rs:38-158, bloom_filter.rs:312-332, hash_iter.rs:13-45, file_parser.rs:114-148.
#include "pfq_kernels.h"

#include <atomic>

namespace pfq {

__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((31u << 11) | 20u) & 0xFu; }

// Block barrier for data exchanged through LDS only.  __syncthreads() also waits for every outstanding global access of
// the wave (s_waitcnt vmcnt(0)): in a software-pipelined loop that is a memory round trip per iteration.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ unsigned long long bcast_u64(unsigned long long v, int src) {
    return ((unsigned long long)bcast_u32((uint32_t)(v >> 32), src) << 32) | bcast_u32((uint32_t)v, src);
}

constexpr uint32_t SCREEN_ROUNDS = 2;  // row loads in flight per lane and probe in the theta=1 screen
constexpr uint32_t SCREEN_KMERS = 4;   // k-mers the theta=1 screen looks at (rows of narrow trees would allow more per load)
constexpr uint32_t PAIR_CHUNK = 32;    // slots a wave reserves at a time in the deferred-pair buffer
static_assert(PAIR_CHUNK == PAIR_RESERVE && 256u == MISS_RESERVE, "the host sizes the buffers' slack by these");
constexpr uint32_t SCREEN_EXTRA = 8;   // k-mers the theta<1 screens look at beyond the maxmiss + 1 that can empty a frontier
constexpr uint32_t NPLANES = 16;       // vertical-counter planes of the theta<1 screen (k-mers per read < 65536)

struct ReadCtx {
    const uint8_t *read;
    uint64_t n;         // k-mers
    uint64_t need;      // ceil_f32(threshold * n)
    uint64_t maxmiss;   // n - need
};

// ---- K2: certificate of one column of S --------------------------------------------------------------------------
// Returns whether #k-mers with all num_hashes bits set in column `col` >= need (query_passes, query.rs:38-49).
__device__ __forceinline__ bool verify_column(BlockLds &lds, uint32_t wave, const QueryArgs &a, const ReadCtx &rc,
                                              uint32_t col) {
    const uint32_t lane = lane_id();
    // `col` is a global column: group col >> group_log2 of the sliced matrix (a single group holds every column of a narrow tree)
    const uint32_t *Sg = a.S_all + (uint64_t)(col >> a.group_log2) * a.group_stride;
    const uint32_t cw = (col & ((1u << a.group_log2) - 1u)) >> 5, cb = col & 31u;
    uint64_t hits = 0, seen = 0;
    for (uint64_t base = 0; base < rc.n; base += WIN_KMERS) {
        uint32_t cnt = (uint32_t)((rc.n - base) < WIN_KMERS ? (rc.n - base) : WIN_KMERS);
        stage_window(lds, wave, rc.read, base, cnt, a.hp.k);
        bool valid = lane < cnt;
        uint64_t h1, h2;
        kmer_hashes(lds, wave, lane, cnt, valid, a.hp, h1, h2);
        ProbeIter it;
        it.init(h1, h2, a.hp);
        uint32_t ok = 1;
        for_each_probe(it, a.hp, [&](uint32_t idx) {
            uint32_t v = valid ? Sg[(uint64_t)idx * a.rw + cw] : 0u;
            ok &= (v >> cb);
        });
        uint64_t b = ballot64(valid && (ok & 1u));
        hits += (uint64_t)__popcll(b);
        seen += cnt;
        if (hits >= rc.need) return true;
        if (seen - hits > rc.maxmiss) return false;
    }
    return hits >= rc.need;
}

// ---- K3 (theta == 1): AND-frontier over the leaf level ---------------------------------------------------------------
// Probes 0 and 1 of the first SCREEN_ROUNDS*slots k-mers: every row is one line of S.  A leaf survives only if all
// probed bits are set (necessary for passing at need == n).
__device__ __forceinline__ uint32_t screen_all(BlockLds &lds, uint32_t wave, const QueryArgs &a, const ReadCtx &rc,
                                               uint32_t colmask, uint64_t &h1, uint64_t &h2) {
    const uint32_t lane = lane_id(), rw = a.rw, slots = 64u >> a.rw_log2;
    const uint32_t word = lane & (rw - 1u), slot = lane >> a.rw_log2;
    uint32_t cnt = (uint32_t)(rc.n < WIN_KMERS ? rc.n : WIN_KMERS);
    stage_window(lds, wave, rc.read, 0, cnt, a.hp.k);
    kmer_hashes(lds, wave, lane, cnt, lane < cnt, a.hp, h1, h2);
    uint32_t i0 = mod_d(h1, a.hp), i1 = mod_d(h2, a.hp);
    uint32_t v0[SCREEN_ROUNDS], v1[SCREEN_ROUNDS];
#pragma unroll
    for (uint32_t j = 0; j < SCREEN_ROUNDS; ++j) {
        uint32_t kk = j * slots + slot;
        bool valid = kk < cnt && kk < SCREEN_KMERS;  // four k-mers x two probes empty the frontier of a foreign read
        uint32_t src = valid ? kk : 0u;
        uint32_t r0 = (uint32_t)__shfl((int)i0, (int)src), r1 = (uint32_t)__shfl((int)i1, (int)src);
        v0[j] = valid ? a.S[(uint64_t)r0 * rw + word] : ~0u;
        v1[j] = (valid && a.hp.num_hashes > 1) ? a.S[(uint64_t)r1 * rw + word] : ~0u;
    }
    uint32_t live = colmask;
#pragma unroll
    for (uint32_t j = 0; j < SCREEN_ROUNDS; ++j) live &= v0[j] & v1[j];
    for (uint32_t s = rw; s < 64u; s <<= 1) live &= (uint32_t)__shfl_xor((int)live, (int)s);
    return live;
}

// ---- K3 (theta < 1): miss-counting frontier ---------------------------------------------------------------------------
// A k-mer is a definite miss for a leaf if its first probed bit is 0.  Per-leaf miss counts are kept bit-sliced
// ("vertical counters": plane p holds bit p of the count of each of the lane's 32 leaves).  Rows are consumed eight
// at a time through a carry-save adder tree (7 CSAs = 14 three-input bit ops for eight rows, then one ripple add of
// the "eights" into the upper planes) instead of one ripple-carry add per row; a leaf leaves the frontier once
// misses > n - need.
__device__ __forceinline__ void csa(uint32_t &hi, uint32_t &lo, uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t u = a ^ b;
    hi = (u & c) | (~u & a);  // majority: where a and b differ c decides, else a — one v_bfi_b32
    lo = u ^ c;
}
// Carry-save adder in three full-rate instructions: the majority is one bit-field insert (where a and b differ c decides,
// else a), which the compiler does not form on its own (it builds the adder from five and / or / xor operations).
// (v_bitop3_b32 would do each output in one instruction, but it is slow on gfx950: the dense counting screen took 31.6 ms
// with two v_bitop3 per adder against 25.6 with the five plain operations.)
__device__ __forceinline__ uint32_t bfi32(uint32_t mask, uint32_t x, uint32_t y) {  // (mask & x) | (~mask & y)
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ void csa3(uint32_t &hi, uint32_t &lo, uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t u = a ^ b;
    hi = bfi32(u, c, a);
    lo = u ^ c;
}
template <uint32_t P, uint32_t BATCH>  // counter planes (counts up to 2^P - 1); rows per lane in flight
__device__ __forceinline__ uint32_t screen_counts_p(BlockLds &lds, uint32_t wave, const QueryArgs &a, const ReadCtx &rc,
                                                    uint32_t colmask) {
    const uint32_t lane = lane_id(), rw = a.rw, slots = 64u >> a.rw_log2;
    const uint32_t word = lane & (rw - 1u), slot = lane >> a.rw_log2;
    uint32_t c[P];  // c[0] = ones, c[1] = twos, c[2] = fours, c[3..] = eights and up
#pragma unroll
    for (uint32_t p = 0; p < P; ++p) c[p] = 0;
    uint32_t live = colmask;
    const uint64_t n_scr = rc.n < rc.maxmiss + 1u + SCREEN_EXTRA ? rc.n : rc.maxmiss + 1u + SCREEN_EXTRA;  // (see dense_counts)
    for (uint64_t base = 0; base < n_scr; base += WIN_KMERS) {
        uint32_t cnt = (uint32_t)((n_scr - base) < WIN_KMERS ? (n_scr - base) : WIN_KMERS);
        stage_window(lds, wave, rc.read, base, cnt, a.hp.k);
        uint64_t h1, h2;
        kmer_hashes(lds, wave, lane, cnt, lane < cnt, a.hp, h1, h2);
        uint32_t i0 = mod_d(h1, a.hp);
        for (uint32_t t = 0; t < cnt; t += BATCH * slots) {
            uint32_t m[BATCH];
#pragma unroll
            for (uint32_t j = 0; j < BATCH; ++j) {
                const uint32_t kk = t + j * slots + slot;
                const bool valid = kk < cnt;
                const uint32_t r0 = (uint32_t)__shfl((int)i0, (int)(valid ? kk : 0u));
                const uint32_t v = valid ? a.S[(uint64_t)r0 * rw + word] : ~0u;
                m[j] = ~v & live;
            }
#pragma unroll
            for (uint32_t g = 0; g < BATCH; g += 8) {
                uint32_t t2a, t2b, t4a, t4b, t8;
                csa(t2a, c[0], c[0], m[g + 0], m[g + 1]);
                csa(t2b, c[0], c[0], m[g + 2], m[g + 3]);
                csa(t4a, c[1], c[1], t2a, t2b);
                csa(t2a, c[0], c[0], m[g + 4], m[g + 5]);
                csa(t2b, c[0], c[0], m[g + 6], m[g + 7]);
                csa(t4b, c[1], c[1], t2a, t2b);
                csa(t8, c[2], c[2], t4a, t4b);
#pragma unroll
                for (uint32_t p = 3; p < P; ++p) {  // ripple the eights upwards
                    const uint32_t carry = c[p] & t8;
                    c[p] ^= t8;
                    t8 = carry;
                }
            }
            // total over the slots (ripple-carry add of vertical counters), then misses > maxmiss ?
            uint32_t tot[P];
#pragma unroll
            for (uint32_t p = 0; p < P; ++p) tot[p] = c[p];
            for (uint32_t s = rw; s < 64u; s <<= 1) {
                uint32_t carry = 0;
#pragma unroll
                for (uint32_t p = 0; p < P; ++p) {
                    uint32_t o = (uint32_t)__shfl_xor((int)tot[p], (int)s);
                    uint32_t x = tot[p] ^ o;
                    uint32_t sum = x ^ carry;
                    carry = (tot[p] & o) | (carry & x);
                    tot[p] = sum;
                }
            }
            uint32_t gt = 0, eq = ~0u;
#pragma unroll
            for (int p = (int)P - 1; p >= 0; --p) {
                uint32_t mb = ((rc.maxmiss >> p) & 1ull) ? ~0u : 0u;
                gt |= eq & tot[p] & ~mb;
                eq &= ~(tot[p] ^ mb);
            }
            live &= ~gt;
            if (ballot64(live != 0) == 0) return 0;
        }
    }
    return live;
}
// Reads of fewer than 256 k-mers take eight planes; longer ones are queued for a second launch of the kernel built
// with NPLANES planes, so that the common case keeps its registers (and its occupancy).
constexpr uint64_t SHORT_KMERS = 256;

// ---- dense pre-screen (theta == 1) -------------------------------------------------------------------------------------
// Most reads of a metagenome hit nothing, and the AND-frontier needs only their first four k-mers: a wave screens
// DENSE_READS reads per pass with lane = (read, k-mer), so one hashing pass serves 16 reads instead of one.
// Returns the mask of regular reads whose frontier is not empty — their frontier words are left in
// live_out[j*rw + w], so the per-read path does not gather them again — and in `irregular` the reads it did not
// screen (no k-mers, need != n).  Reads in neither mask are finished: no leaf can pass them.
// The group is reads r0 .. r0 + n_in_group - 1, or list[r0 ..] when a list is given (entries 0xffffffff are no reads);
// `rid` returns the lane's read.  MULTI (the coarse level of a two-level frontier, whose filters are fuller): n_probes
// probes per k-mer (2 .. COARSE_MAX_PROBES) instead of two.
template <bool MULTI>
__device__ __forceinline__ uint32_t dense_screen(uint32_t *fw, uint32_t *rw_, const uint8_t *comp, uint32_t *live_out,
                                                 const QueryArgs &a, const uint32_t *list, uint64_t r0, uint32_t n_in_group,
                                                 uint32_t colmask, uint32_t n_probes, uint32_t &irregular, uint64_t &lane_len,
                                                 uint32_t &rid) {
    const uint32_t lane = lane_id(), j = lane >> 2, t = lane & 3u, k = a.hp.k;
    const uint32_t rw = a.rw;
    uint64_t o0 = 0, L = 0;
    rid = (uint32_t)(r0 + j);
    bool present = j < n_in_group;
    if (list) {
        rid = present ? list[r0 + j] : 0xffffffffu;
        present = rid != 0xffffffffu;  // (reads are indexed with 31 bits)
    }
    if (present) {
        o0 = a.off[list ? (uint64_t)rid : r0 + j];
        L = a.off[(list ? (uint64_t)rid : r0 + j) + 1] - o0;
    }
    lane_len = (t == 0) ? L : 0;  // read length, on the first lane of each read
    const uint64_t n = (L >= k) ? (L - k + 1) : 0;
    const bool regular = present && n >= 1 && need_kmers(a.threshold, n) == n;
    const uint32_t nk = regular ? (uint32_t)(n < DENSE_KMERS ? n : DENSE_KMERS) : 0u;
    const uint32_t W = nk ? nk + k - 1 : 0u;  // bytes of this read that are staged
    uint8_t *fwd = reinterpret_cast<uint8_t *>(fw), *rcb = reinterpret_cast<uint8_t *>(rw_);
    const uint32_t mb = j * MINI_BYTES + WIN_PAD;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i0 = 0; i0 < k + DENSE_KMERS - 1; i0 += 24) {  // six bytes per lane and batch, loads before uses
        uint8_t b[6];
#pragma unroll
        for (uint32_t u = 0; u < 6; ++u) {
            const uint32_t idx = i0 + 4u * u + t;
            b[u] = idx < W ? a.seq[o0 + idx] : (uint8_t)0;
        }
#pragma unroll
        for (uint32_t u = 0; u < 6; ++u) {
            const uint32_t idx = i0 + 4u * u + t;
            if (idx < W) {
                fwd[mb + idx] = b[u];
                rcb[mb + (W - 1 - idx)] = comp[b[u]];
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    const bool valid = t < nk;
    uint64_t h1, h2;
    kmer_hashes_at(fw, rw_, mb + t, mb + (W - t - k), valid, a.hp, h1, h2);
    uint32_t i0v, i1v;
    uint32_t ixs[MULTI ? COARSE_MAX_PROBES : 1];  // MULTI: probe p of the lane's k-mer
    if (MULTI) {
        ProbeIter pit;
        pit.init(h1, h2, a.hp);
        ixs[0] = i0v = pit.i0;
        ixs[1] = i1v = pit.g;
        ixs[2] = pit.x;
#pragma unroll
        for (uint32_t p = 3; p < COARSE_MAX_PROBES; ++p) ixs[p] = p < n_probes ? pit.step(a.hp) : 0u;  // (n_probes <= num_hashes, wave-uniform)
    } else {
        i0v = mod_d(h1, a.hp);
        i1v = mod_d(h2, a.hp);
    }
    const bool two = a.hp.num_hashes > 1;
    const uint32_t ppk = MULTI ? n_probes : 2u;                  // rows per k-mer
    const uint32_t ppk_inv = (65536u + ppk - 1u) / ppk;          // p / ppk == (p * ppk_inv) >> 16 for p < 64
    const uint64_t present_b = ballot64(present && t == 0);
    uint32_t survive = 0;
    irregular = 0;
#ifdef PFQ_DENSE_V1
    if constexpr (MULTI) {
#else
    {
#endif
        // The screening rows (4 k-mers x 2 probes per read; n_probes at the coarse level) are ANDed in REGISTERS: the row indices go through LDS (the
        // staging buffers are free now; forward and reverse buffer are adjacent), rw / 4 lanes per read own 16 bytes of every
        // row of their read, 64 / (rw / 4) reads per sub-pass, eight 16-byte gathers in flight per lane.  (Lanes as (row, part)
        // with the indices picked by ds_bpermute and the AND finished by shuffles cost ~490 LDS-pipe instructions per pass of
        // 16 reads against ~20 here: k_coarse at theta = 1 was bound by them.)
        const uint32_t lpr_log2 = a.rw_log2 - 2u, lpr = 1u << lpr_log2, part = lane & (lpr - 1u), rows = DENSE_KMERS * ppk;
        uint32_t cm[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t w = part * 4u + u;
            cm[u] = (w * 32u < a.n_leaves) ? ((a.n_leaves - w * 32u >= 32u) ? ~0u : ((1u << (a.n_leaves - w * 32u)) - 1u)) : 0u;
        }
        uint32_t *idx = fw;  // [read][k-mer][probe]: 64 x n_probes <= 384 of the 672 dwords
        __builtin_amdgcn_wave_barrier();
        if constexpr (MULTI) {
#pragma unroll
            for (uint32_t p = 0; p < COARSE_MAX_PROBES; ++p)
                if (p < ppk) idx[lane * ppk + p] = valid ? ixs[p] : a.ones_row;  // (k-mers that do not exist: the all-ones row)
        } else {
            idx[lane * 2u] = valid ? i0v : a.ones_row;
            idx[lane * 2u + 1u] = (valid && two) ? i1v : a.ones_row;
        }
        __builtin_amdgcn_wave_barrier();
        const uint64_t irr_b = ballot64(present && t == 0 && nk == 0);  // (no screening possible: the per-read path)
        const uint32_t rps = 64u >> lpr_log2;  // reads per sub-pass
        for (uint32_t j0 = 0; j0 < DENSE_READS; j0 += rps) {
            const uint32_t jj = j0 + (lane >> lpr_log2);
            const bool rd = jj < DENSE_READS && ((present_b >> (jj * 4u)) & 1ull) && !((irr_b >> (jj * 4u)) & 1ull);
            const uint32_t *my = idx + (rd ? jj : 0u) * rows;
            uint4 acc = make_uint4(~0u, ~0u, ~0u, ~0u);
            for (uint32_t b = 0; b < rows; b += 8u) {
                const uint4 xa = *reinterpret_cast<const uint4 *>(my + b);
                const uint4 xb = *reinterpret_cast<const uint4 *>(my + (b + 4u < rows ? b + 4u : b));
                const uint32_t xs[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
                uint4 m[8];
#pragma unroll
                for (uint32_t u = 0; u < 8; ++u) {
                    const uint32_t x = (rd && b + u < rows) ? xs[u] : a.ones_row;
                    m[u] = *reinterpret_cast<const uint4 *>(a.S + (uint64_t)x * rw + part * 4u);
                }
#pragma unroll
                for (uint32_t u = 0; u < 8; ++u) {
                    acc.x &= m[u].x; acc.y &= m[u].y; acc.z &= m[u].z; acc.w &= m[u].w;
                }
            }
            acc.x &= cm[0]; acc.y &= cm[1]; acc.z &= cm[2]; acc.w &= cm[3];
            const uint64_t bal = ballot64(rd && (acc.x | acc.y | acc.z | acc.w) != 0u);
            const uint64_t mine_b = (bal >> ((lane >> lpr_log2) << lpr_log2)) & (lpr >= 64u ? ~0ull : ((1ull << lpr) - 1ull));
            if (rd && mine_b) *reinterpret_cast<uint4 *>(live_out + jj * rw + part * 4u) = acc;
            for (uint32_t r = 0; r < rps && j0 + r < DENSE_READS; ++r)
                if ((bal >> (r << lpr_log2)) & (lpr >= 64u ? ~0ull : ((1ull << lpr) - 1ull))) survive |= 1u << (j0 + r);
        }
        for (uint32_t jj = 0; jj < DENSE_READS; ++jj)
            if ((irr_b >> (jj * 4u)) & 1ull) irregular |= 1u << jj;
        __builtin_amdgcn_wave_barrier();
        return survive;
    }
    // Row gathers, 16 bytes per lane: a read's 2*nk rows of rw dwords are covered by lanes (row = lane / (rw/4),
    // part = lane % (rw/4)); with rw = 32 one load instruction fetches all 8 rows of a read.  The loads of
    // DENSE_BATCH reads are issued before any is consumed.  (rw >= 4 here; smaller trees skip the pre-screen.)
    const uint32_t lpr = rw >> 2, lpr_log2 = a.rw_log2 - 2u;       // lanes per row
    const uint32_t rpi = 64u >> lpr_log2;                           // rows per load instruction
    const uint32_t part = lane & (lpr - 1u), rsel = lane >> lpr_log2;
    // leaf-column mask of my four dwords
    uint32_t cm[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t w = part * 4u + u;
        cm[u] = (w * 32u < a.n_leaves) ? ((a.n_leaves - w * 32u >= 32u) ? ~0u : ((1u << (a.n_leaves - w * 32u)) - 1u)) : 0u;
    }
    constexpr uint32_t DENSE_BATCH = 8;
    for (uint32_t jb = 0; jb < n_in_group; jb += DENSE_BATCH) {
        uint4 acc[DENSE_BATCH];
#pragma unroll
        for (uint32_t u = 0; u < DENSE_BATCH; ++u) {
            const uint32_t jj = jb + u;
            acc[u] = make_uint4(~0u, ~0u, ~0u, ~0u);
            if (jj >= n_in_group) continue;
            const uint32_t nk_j = bcast_u32(nk, (int)((jj < DENSE_READS ? jj : 0u) * 4u));
            const uint32_t rows = ppk * nk_j;
            for (uint32_t p0 = 0; p0 < rows; p0 += rpi) {
                const uint32_t p = p0 + rsel;
                uint32_t idx;
                bool pv;
                if (MULTI) {
                    pv = p < rows;
                    const uint32_t km = ((pv ? p : 0u) * ppk_inv) >> 16, pr = (pv ? p : 0u) - km * ppk;
                    const int src = (int)(jj * 4u + km);
                    idx = (uint32_t)__shfl((int)ixs[0], src);
#pragma unroll
                    for (uint32_t q = 1; q < COARSE_MAX_PROBES; ++q) {
                        if (q >= ppk) break;  // (wave-uniform)
                        const uint32_t xq = (uint32_t)__shfl((int)ixs[q], src);
                        idx = pr == q ? xq : idx;
                    }
                } else {
                    pv = p < rows && ((p & 1u) == 0 || two);
                    const int src = (int)(jj * 4u + ((pv ? p : 0u) >> 1));
                    const uint32_t x0 = (uint32_t)__shfl((int)i0v, src), x1 = (uint32_t)__shfl((int)i1v, src);
                    idx = (p & 1u) ? x1 : x0;
                }
                if (pv) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(a.S + (uint64_t)idx * rw + part * 4u);
                    acc[u].x &= v.x; acc[u].y &= v.y; acc[u].z &= v.z; acc[u].w &= v.w;
                }
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < DENSE_BATCH; ++u) {
            const uint32_t jj = jb + u;
            if (jj >= n_in_group) continue;
            const uint32_t nk_j = bcast_u32(nk, (int)(jj * 4u));
            if (nk_j == 0) {  // irregular read: leave it to the per-read path (unused slots of a list are no reads)
                if ((present_b >> (jj * 4u)) & 1ull) irregular |= 1u << jj;
                continue;
            }
            uint4 l4 = acc[u];
            for (uint32_t sft = lpr; sft < 64u; sft <<= 1) {  // AND over the rows held by other lanes
                l4.x &= (uint32_t)__shfl_xor((int)l4.x, (int)sft);
                l4.y &= (uint32_t)__shfl_xor((int)l4.y, (int)sft);
                l4.z &= (uint32_t)__shfl_xor((int)l4.z, (int)sft);
                l4.w &= (uint32_t)__shfl_xor((int)l4.w, (int)sft);
            }
            l4.x &= cm[0]; l4.y &= cm[1]; l4.z &= cm[2]; l4.w &= cm[3];
            if (ballot64((l4.x | l4.y | l4.z | l4.w) != 0)) {
                survive |= 1u << jj;
                if (rsel == 0) *reinterpret_cast<uint4 *>(live_out + jj * rw + part * 4u) = l4;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    return survive;
}

// ---- dense miss-counting screen (theta < 1) -------------------------------------------------------------------------
// The per-read frontier above is a chain of dependent round trips (stage, hash, gather, count) per read and wave.
// Here a wave screens 64/lpr consecutive reads at once, lpr = rw/4 lanes per read: lane (j, q) hashes k-mer
// pos+q of read j and owns dwords 4q..4q+3 (128 leaf columns) of every row of that read, so the vertical counters
// never cross lanes; each pass issues lpr row gathers of 16 bytes per lane before any is consumed.  Needs
// 16 <= rw <= 64 and reads that may miss fewer than 2^P k-mers (n - need < 2^P; P counter planes: 8 in the main launch,
// 16 in the launch for long reads); other reads are returned in `irregular` for the per-read path.  The group is reads r0 .. r0+n-1, or
// list[r0 .. r0+n-1] when a list is given; `rid` returns the lane's read.  On return live_out[j*rw + w] holds the
// frontier words of read j and `survive` the reads with a non-empty frontier.
// MULTI (the coarse level of a two-level frontier): a k-mer is a definite miss for a column if ANY of its first n_probes
// probed bits is 0 (1 .. 4; the filters of internal nodes are too full for one bit to tell), and the screen looks at
// scr_extra k-mers beyond maxmiss + 1 instead of SCREEN_EXTRA.
// RECS (the launches that defer survivors to the bucketed certificate stage): the screen hashes most k-mers of a read that
// survives anyway, so it leaves their probe records behind — recs[read offset + k-mer] — and `n_rec` returns how many k-mers
// from the read's first on have theirs; the per-read path then only hashes what is left (the survivors were hashed twice
// before: ~10 of 25 ms at threshold 0.3).  Reads that die cost a few 16-byte stores each.
template <uint32_t P, uint32_t LPR_LOG2, bool MULTI = false, bool RECS = false>  // counter planes; log2 of the lanes per read (rw = 4 << LPR_LOG2: 16, 32 or 64)
__device__ __forceinline__ void dense_counts(uint32_t *fw, uint32_t *rw_, const uint8_t *comp, uint32_t *live_out,
                                             const QueryArgs &a, const uint32_t *list, uint64_t r0, uint32_t n_in_group,
                                             uint32_t &survive, uint32_t &irregular, uint64_t &lane_len, uint64_t &rid,
                                             uint32_t &n_rec, uint32_t n_probes = 1, uint32_t scr_extra = SCREEN_EXTRA) {
    const uint32_t lane = lane_id(), k = a.hp.k;
    static_assert(LPR_LOG2 >= 2 && LPR_LOG2 <= 4, "rows of 16, 32 or 64 words");
    constexpr uint32_t lpr_log2 = LPR_LOG2, lpr = 1u << lpr_log2, rpw = 64u >> lpr_log2, rw = 4u << lpr_log2;
    const uint32_t j = lane >> lpr_log2, q = lane & (lpr - 1u);
    uint64_t o0 = 0, L = 0;
    rid = 0;
    bool in_group = j < n_in_group;
    if (in_group) {
        rid = list ? (uint64_t)list[r0 + j] : r0 + j;
        if (list && (uint32_t)rid == 0xffffffffu) {  // unused slot of a reservation in the list
            in_group = false;
            rid = 0;
        }
    }
    if (in_group) {
        o0 = a.off[rid];
        L = a.off[rid + 1] - o0;
    }
    lane_len = (q == 0) ? L : 0;
    const uint64_t n64 = (L >= k) ? (L - k + 1) : 0;
    const uint64_t need = need_kmers(a.threshold, n64);
    // (the counters only have to reach maxmiss + 1: reads of any length whose n - need fits P bits are regular, e.g.
    // 300 bp at theta 0.5 with 8 planes)
    const bool regular = in_group && n64 >= 1 && n64 < (1ull << 31) && need >= 1 && need <= n64 && n64 - need < (1ull << P);
    const uint32_t n = regular ? (uint32_t)n64 : 0u, maxmiss = regular ? (uint32_t)(n64 - need) : 0u;
    // The screen may stop after ANY prefix of the k-mers — a leaf is dropped only for misses it has seen, what is left
    // is certified exactly — and the first maxmiss + 1 (+ a few for foreign leaves' lucky first bits) already empty the
    // frontier of a read that hits nothing; a read that does hit would otherwise walk all its k-mers for nothing.
    // (filters that are fuller than a per cent leave lucky leaves alive after that many — 2 % of the leaves survive 65 k-mers
    // at a fill of 7 %, none 81: the limit moves on, a pass at a time, while a leaf of the read is alive but within 2^b misses
    // of dying, 2^b about half of maxmiss; leaves that miss few k-mers — the read's real candidates — do not prolong it)
    uint32_t n_scr = (maxmiss + 1u + (MULTI ? scr_extra : SCREEN_EXTRA) + lpr - 1u) & ~(lpr - 1u);  // (whole passes: the lanes are there anyway)
    if (n_scr > n) n_scr = n;
    irregular = 0;
    {
        uint64_t b = ballot64(in_group && !regular && q == 0);
        while (b) {
            const int l = __ffsll((unsigned long long)b) - 1;
            b &= b - 1;
            irregular |= 1u << ((uint32_t)l >> lpr_log2);
        }
    }
    // Leaves (of my four dwords) die when their miss count exceeds maxmiss.  The counters start at 2^P - 1 - maxmiss, so
    // "misses > maxmiss" is the carry out of the top plane: no comparison is needed, and rows need no masking either
    // (counters of dead leaves and of padding columns may wrap, they stay dead).  Everything is kept COMPLEMENTED — counter
    // planes, carries, and `live` = ~dead: the misses to add are the complements of the rows, and majority and parity of
    // complemented operands are the complements of majority and parity, so the adders take the rows as they come.
    // (columns that are no leaves start out dead: no separate column mask has to stay in registers)
    uint32_t live[4];
    const uint32_t bias = regular ? ((1u << P) - 1u - maxmiss) : 0u;
    uint32_t c[P][4];  // complements of the counter planes
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t w = q * 4u + u;
        uint32_t cmu = (w * 32u < a.n_leaves) ? ((a.n_leaves - w * 32u >= 32u) ? ~0u : ((1u << (a.n_leaves - w * 32u)) - 1u)) : 0u;
        if (!regular) cmu = 0u;
        live[u] = cmu;
#pragma unroll
        for (uint32_t p = 0; p < P; ++p) c[p][u] = ((bias >> p) & 1u) ? 0u : ~0u;
    }
    uint8_t *fwd = reinterpret_cast<uint8_t *>(fw), *rcb = reinterpret_cast<uint8_t *>(rw_);
    const uint32_t stride = (DENSE_READS / rpw) * MINI_BYTES;  // bytes of LDS per read and direction (rpw <= 16)
    const uint32_t base = j * stride + WIN_PAD;
    bool alive = regular;  // my read still has a live leaf
    n_rec = 0;
    const bool rec_ok = RECS && a.screen_recs && a.recs != nullptr && regular && o0 + n64 <= a.rec_cap;  // (room for all the read's records, as the per-read path asks)
    // The bytes of a read are staged a segment at a time — as many k-mers as the read's share of the LDS holds, 128 of a
    // 150 bp read at k = 21 — so that a pass needs no global load of its own (a round trip per pass of eight k-mers before).
    const uint32_t seg_cap = stride - 2u * WIN_PAD;
    const uint32_t seg_kmers = ((seg_cap - k + 1u) / lpr) * lpr;  // (>= lpr: stride >= 84 bytes, k <= 64, lpr <= stride / 21)
    uint32_t seg_pos = 0, W = 0;  // first k-mer of the staged segment; bytes staged for my read
    auto one_pass = [&](const uint32_t pos, const bool restage) {  // k-mers pos .. pos + lpr - 1 of the reads that are still active
        const bool active = alive && pos < n_scr;
        if (pos - seg_pos >= seg_kmers || pos == 0 || restage) {  // (wave-uniform)
            seg_pos = pos;
            const uint32_t kn = active ? (n - pos < seg_kmers ? n - pos : seg_kmers) : 0u;  // (to the read's end: the limit may move)
            W = kn ? kn + k - 1u : 0u;
            __builtin_amdgcn_wave_barrier();
            // six bytes per lane and batch; the loads are unconditional (clamped address) so that none waits for another
            const uint8_t *src = a.seq + (active ? o0 + pos : 0ull);
            for (uint32_t i0 = 0; i0 < k + seg_kmers - 1u; i0 += 6u * lpr) {
                if (ballot64(i0 < W) == 0) break;
                uint8_t b[6];
#pragma unroll
                for (uint32_t u = 0; u < 6; ++u) {
                    const uint32_t idx = i0 + lpr * u + q;
                    b[u] = src[idx < W ? idx : 0u];
                }
#pragma unroll
                for (uint32_t u = 0; u < 6; ++u) {
                    const uint32_t idx = i0 + lpr * u + q;
                    if (idx < W) {
                        fwd[base + idx] = b[u];
                        rcb[base + (W - 1u - idx)] = comp[b[u]];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        const uint32_t x = pos - seg_pos + q;  // my k-mer's place in the segment
        const uint32_t nk = active ? (n_scr - pos < lpr ? n_scr - pos : lpr) : 0u;
        const bool valid = q < nk;
        uint64_t h1, h2;
        kmer_hashes_at(fw, rw_, base + x, base + (W - x - k), valid, a.hp, h1, h2);
        if (RECS) {
            if (rec_ok && valid) a.recs[o0 + pos + q] = make_probe_record(h1, h2, a.hp);
            if (rec_ok && active && pos == n_rec) n_rec = pos + nk;  // (contiguous from the first k-mer on: passes after a pause do not count)
        }
        // row indices of the wave go through LDS (live_out is free until the end): a lane fetches the eight of its
        // read with two 16-byte reads; k-mers that do not exist point at the all-ones row behind S (no miss)
        if (MULTI) {  // probe p of the wave's k-mers at live_out[64 p ..]
            ProbeIter pit;
            pit.init(h1, h2, a.hp);
            live_out[lane] = valid ? pit.i0 : a.ones_row;
            if (n_probes > 1) live_out[64u + lane] = valid ? pit.g : a.ones_row;
            if (n_probes > 2) live_out[128u + lane] = valid ? pit.x : a.ones_row;
            if (n_probes > 3) {
                const uint32_t x3 = pit.step(a.hp);
                live_out[192u + lane] = valid ? x3 : a.ones_row;
            }
        } else live_out[lane] = valid ? mod_d(h1, a.hp) : a.ones_row;
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1  // (rows of 64 words: two batches of eight gathers; unrolled, the sixteen cost the third wave per SIMD: 33.8 -> 41.4 ms)
        for (uint32_t t0 = 0; t0 < lpr; t0 += 8u) {
            const uint4 xa = *reinterpret_cast<const uint4 *>(live_out + (j << lpr_log2) + t0);
            const uint4 xb = *reinterpret_cast<const uint4 *>(live_out + (j << lpr_log2) + t0 + 4u);
            const uint32_t xs[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
            uint4 m[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                const uint32_t x = (t0 + u < lpr) ? xs[u] : a.ones_row;  // lpr == 4: the upper four belong to the next read
                m[u] = *reinterpret_cast<const uint4 *>(a.S + ((uint64_t)x << (lpr_log2 + 2u)) + q * 4u);
            }
            if (MULTI) {  // a k-mer is contained only where all its probed bits are set: AND of the probes' rows
                for (uint32_t pr = 1; pr < n_probes; ++pr) {
                    const uint4 ya = *reinterpret_cast<const uint4 *>(live_out + 64u * pr + (j << lpr_log2) + t0);
                    const uint4 yb = *reinterpret_cast<const uint4 *>(live_out + 64u * pr + (j << lpr_log2) + t0 + 4u);
                    const uint32_t ys[8] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
                    uint4 mm[8];
#pragma unroll
                    for (uint32_t u = 0; u < 8; ++u) {
                        const uint32_t x = (t0 + u < lpr) ? ys[u] : a.ones_row;
                        mm[u] = *reinterpret_cast<const uint4 *>(a.S + ((uint64_t)x << (lpr_log2 + 2u)) + q * 4u);
                    }
#pragma unroll
                    for (uint32_t u = 0; u < 8; ++u) {
                        m[u].x &= mm[u].x; m[u].y &= mm[u].y; m[u].z &= mm[u].z; m[u].w &= mm[u].w;
                    }
                }
            }
#define PFQ_CSA_WORD(F, W_)                                                   \
    {                                                                         \
        uint32_t t2a, t2b, t4a, t4b, t8;  /* complements of the carries */    \
        csa3(t2a, c[0][W_], c[0][W_], m[0].F, m[1].F);                        \
        csa3(t2b, c[0][W_], c[0][W_], m[2].F, m[3].F);                        \
        csa3(t4a, c[1][W_], c[1][W_], t2a, t2b);                              \
        csa3(t2a, c[0][W_], c[0][W_], m[4].F, m[5].F);                        \
        csa3(t2b, c[0][W_], c[0][W_], m[6].F, m[7].F);                        \
        csa3(t4b, c[1][W_], c[1][W_], t2a, t2b);                              \
        csa3(t8, c[2][W_], c[2][W_], t4a, t4b);                               \
        _Pragma("unroll") for (uint32_t p = 3; p < P; ++p) {                  \
            uint32_t carry = c[p][W_] | t8;         /* ~(plane & carry) */    \
            c[p][W_] = ~(c[p][W_] ^ t8);            /* ~(plane ^ carry) */    \
            asm("" : "+v"(carry));  /* (else the compiler folds two steps into one v_bitop3_b32, which is slow) */ \
            t8 = carry;                                                       \
        }                                                                     \
        live[W_] &= t8;                                                       \
    }
            PFQ_CSA_WORD(x, 0)
            PFQ_CSA_WORD(y, 1)
            PFQ_CSA_WORD(z, 2)
            PFQ_CSA_WORD(w, 3)
#undef PFQ_CSA_WORD
        }
        uint32_t any = live[0] | live[1] | live[2] | live[3];
        for (uint32_t sft = 1; sft < lpr; sft <<= 1) any |= (uint32_t)__shfl_xor((int)any, (int)sft);
        alive = alive && any != 0;
    };
    uint32_t pos = 0;
    for (;; pos += lpr) {
        if (ballot64(alive && pos < n_scr) == 0) break;
        one_pass(pos, false);
    }
    // Reads that are still alive at their limit: while one of their live leaves is about to die, one more pass (rare where
    // the filters are a per cent full; kept out of the loop above, where it cost 3 - 7 %).  `pos` is where the slowest read
    // of the wave stopped — a multiple of lpr at or past every read's limit; a read that goes on takes k-mers pos .. pos +
    // lpr - 1: skipping k-mers is as valid as stopping.  Its bytes are staged anew (a segment staged while it was idle
    // does not hold them).
    for (;; pos += lpr) {
        const bool cand = alive && pos < n;
        if (ballot64(cand) == 0) break;
        // counter = 2^P - 1 - maxmiss + misses: the planes b .. P-1 all set <=> misses > maxmiss - 2^b (complemented planes: all clear)
        const uint32_t b = 31u - (uint32_t)__clz((int)((maxmiss >> 1) | 1u));
        uint32_t nearw = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4; ++w) {
            uint32_t o = 0;
#pragma unroll
            for (uint32_t p = 0; p < P; ++p) o |= (p >= b) ? c[p][w] : 0u;
            nearw |= live[w] & ~o;
        }
        uint32_t near = nearw != 0u ? 1u : 0u;
        for (uint32_t sft = 1; sft < lpr; sft <<= 1) near |= (uint32_t)__shfl_xor((int)near, (int)sft);
        const bool go = cand && near != 0u;
        if (ballot64(go) == 0) break;
        n_scr = go ? (pos + lpr < n ? pos + lpr : n) : 0u;
        one_pass(pos, true);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < 4; ++u) live_out[j * rw + q * 4u + u] = live[u];
    __builtin_amdgcn_wave_barrier();
    survive = 0;
    {
        uint64_t b = ballot64(alive && q == 0);
        while (b) {
            const int l = __ffsll((unsigned long long)b) - 1;
            b &= b - 1;
            survive |= 1u << ((uint32_t)l >> lpr_log2);
        }
    }
}

// A read's last window of records is made by k_tail_records when it holds at most tail_max k-mers (16: four reads per pass;
// 32: two — the host asks for 32 when the reads' length makes such tails, e.g. 100 bp reads at k = 20: 81 = 64 + 17 k-mers).
__device__ __forceinline__ bool has_batched_tail(uint64_t n, uint32_t tail_max) {
    const uint32_t tl = (uint32_t)(n & (WIN_KMERS - 1u));
    return n > WIN_KMERS && n < (1ull << 32) && tl != 0 && tl <= tail_max;
}

// Reads of >= 256 k-mers are queued for the launch with wider counters.  The irregular reads of a dense group that belong
// there are queued together — ONE atomic on the queue's cursor per group instead of one per read (a single address sustains
// ~88 atomics/us: a block of 1000 bp reads spent 12 of its 30 ms there).  lane_len: the read's length on the first lane of
// its lanes (0 elsewhere), rid_lo its index there; returns `irregular` without the queued reads.
__device__ __forceinline__ uint32_t queue_long_reads(const QueryArgs &a, uint32_t irregular, uint64_t lane_len, uint32_t rid_lo, uint32_t lanes_log2) {
    const uint32_t lane = lane_id(), jj = lane >> lanes_log2;
    const bool first = (lane & ((1u << lanes_log2) - 1u)) == 0u;
    const uint64_t n = lane_len >= a.hp.k ? lane_len - a.hp.k + 1 : 0, need = need_kmers(a.threshold, n);
    const bool is_long = first && ((irregular >> jj) & 1u) && n >= SHORT_KMERS && need != 0 && need <= n;
    const uint64_t m = ballot64(is_long);
    if (!m) return irregular;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(a.n_long, (unsigned int)__popcll(m));
    base = bcast_u32(base, 0);
    if (is_long) a.long_list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = rid_lo;
    uint64_t mm = m;
    while (mm) {
        const uint32_t l = (uint32_t)__ffsll((unsigned long long)mm) - 1u;
        mm &= mm - 1ull;
        irregular &= ~(1u << (l >> lanes_log2));
    }
    return irregular;
}

// ---- the classification kernel ---------------------------------------------------------------------------------------
// LPR_LOG2 (thresholds < 1): log2 of the lanes per read of the dense counting screen, rw = 4 << LPR_LOG2 (16, 32 or 64 row
// words: a build per row width keeps the screen free of run-time shapes); 0: rows narrower than 16 words, per-read screen.
// (the counting builds for rows of 16 and 32 words are held to three waves per SIMD — 168 VGPRs; the allocator stops a few
// registers above on its own.  The 64-word build is left alone: bounded, it spills and the harness geometry loses 6 %.)
// BLOCKS (DEFER, theta == 1, no guard columns): survivors are deferred per block of 8 leaf columns — (read, block | mask of
// the candidate leaves << 24) — see TILE_LOG2_BLOCK.
// LIST: the launch of a leaf group of a two-level frontier — its reads are the ones k_coarse listed for the group (read_list).
template <bool DEFER, bool COUNTS, bool LONG = false, uint32_t LPR_LOG2 = 0, bool BLOCKS = false, bool LIST = false>
__global__ void __launch_bounds__(256, (COUNTS && !LONG && LPR_LOG2 != 0 && LPR_LOG2 != 4) ? 3 : 1) k_classify(QueryArgs a) {
    __shared__ BlockLds lds;
    __shared__ DenseLds<(DEFER && !LONG) || COUNTS> dlds;
    fill_complement(lds.comp);
    __syncthreads();
    if ((LIST || LONG) && a.grid_groups) {  // (block-uniform) one launch for all leaf groups of a two-level frontier: mine is blockIdx.y
        const uint32_t g = blockIdx.y;
        a.S = a.S_all + (uint64_t)g * a.group_stride;
        a.col0 = g << a.group_log2;
        a.n_leaves = a.total_leaves - a.col0 < (1u << a.group_log2) ? a.total_leaves - a.col0 : (1u << a.group_log2);
        a.read_list += (uint64_t)g * a.list_cap;
        a.n_list += g;
        if (COUNTS) {
            a.long_list += (uint64_t)g * a.list_cap;
            a.n_long += g;
        }
    }
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint64_t gw = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + wave, nw = (uint64_t)gridDim.x * WAVES_PER_BLOCK;
    const uint32_t rw = a.rw;
    const uint32_t word = lane & (rw - 1u), slot = lane >> a.rw_log2;
    uint32_t colmask = 0;  // leaf columns of my dword
    if (word * 32u < a.n_leaves) colmask = (a.n_leaves - word * 32u >= 32u) ? ~0u : ((1u << (a.n_leaves - word * 32u)) - 1u);
    unsigned long long st_cand = 0, st_hits = 0, st_all = 0, st_bytes = 0, st_def = 0;
    unsigned long long dense_bytes = 0;  // per lane
    // deferred pairs are appended through per-wave reservations of PAIR_CHUNK slots: one atomic on the shared
    // cursor per chunk (a single hot address saturates at ~88 atomics/us); unused slots are voided at the end
    unsigned long long pair_base = 0;
    uint32_t pair_used = PAIR_CHUNK;
    uint32_t miss_left = 0;  // thresholds < 1: miss words this wave may still hand out (reserved MISS_RESERVE at a time)

    // pre: frontier words from the dense screen; n_done: k-mers (from the first on) whose probe records the screen wrote
    auto process_read = [&](uint64_t r, const uint32_t *pre, uint32_t n_done = 0) {
        const uint64_t o0 = a.off[r], L = a.off[r + 1] - o0;
        ReadCtx rc;
        rc.read = a.seq + o0;
        rc.n = (L >= a.hp.k) ? (L - a.hp.k + 1) : 0;  // get_kmers, file_parser.rs:135-138
        rc.need = need_kmers(a.threshold, rc.n);       // query.rs:48
        if (COUNTS && !LONG && !pre && rc.n >= SHORT_KMERS && rc.need != 0 && rc.need <= rc.n) {
            if (lane == 0) a.long_list[atomicAdd(a.n_long, 1u)] = (uint32_t)r;  // second launch (LONG)
            return;
        }
        if (a.first_group) st_bytes += L;
        if (rc.need == 0) {  // 0 >= 0 at every node: the read reaches and counts at every leaf (of every column group)
            ++st_all;
            if (a.allhit_flag && lane == 0) a.allhit_flag[r] = 1;
            return;
        }
        if (rc.need > rc.n) return;  // cannot pass any node
        rc.maxmiss = rc.n - rc.need;

        // The AND-frontier is only valid when no miss is tolerated; (n as f32) rounds for n >= 2^24, so even at
        // theta == 1 a read can have maxmiss > 0: such reads are certified against every leaf instead.
        bool prepared = false, have_w0 = false;
        uint64_t w0_h1 = 0, w0_h2 = 0;  // hashes of the first window when the AND-frontier computed them
        uint32_t live;
        if (COUNTS) {
            if (pre) live = pre[word] & colmask;
            else if (!LONG) live = screen_counts_p<8, 16>(lds, wave, a, rc, colmask);
            else if (rc.n < (1ull << NPLANES)) live = screen_counts_p<NPLANES, 8>(lds, wave, a, rc, colmask);
            else live = colmask;  // counters too narrow: no screening, certify every leaf
        }
        else if (rc.maxmiss == 0 && pre) live = pre[word] & colmask;  // frontier from the dense pre-screen
        else if (rc.maxmiss == 0) { live = screen_all(lds, wave, a, rc, colmask, w0_h1, w0_h2); have_w0 = true; }
        else live = colmask;

        while (true) {
            uint64_t b = ballot64(live != 0 && slot == 0);
            if (!b) break;
            int src = __ffsll((unsigned long long)b) - 1;
            uint32_t wv = bcast_u32(live, src);
            uint32_t bit = (uint32_t)__ffs((int)wv) - 1u;
            uint32_t mask8 = 1u;  // BLOCKS: the candidates among the 8 leaves of the block, taken together
            if (BLOCKS) {
                bit &= ~7u;
                mask8 = (wv >> bit) & 0xffu;
            }
            const uint32_t col = a.col0 + (uint32_t)src * 32u + bit;  // global column (BLOCKS: the block's first)
            if ((int)lane == src) live &= ~((BLOCKS ? 0xffu : 1u) << bit);
            st_cand += BLOCKS ? (uint32_t)__popc(mask8) : 1u;
            // (guard columns — ancestors that are not provably supersets — of a deferred pair: k_expand_guards)
            const uint32_t miss_need = (COUNTS && a.bucket_words) ? (rc.n < (1ull << 37) ? (uint32_t)((rc.n + 63) >> 6) : 0xffffffffu) : 0u;
            if (DEFER && COUNTS && a.bucket_words && miss_left < miss_need && miss_need != 0xffffffffu) {  // wave-uniform
                const uint32_t want = miss_need > MISS_RESERVE ? miss_need : MISS_RESERVE;
                unsigned long long got = 0;
                if (lane == 0) got = atomicAdd(a.miss_cursor, (unsigned long long)want);
                got = ((unsigned long long)bcast_u32((uint32_t)(got >> 32), 0) << 32) | bcast_u32((uint32_t)got, 0);
                if (got + want <= a.miss_cap) miss_left = want;  // else: no room, this pair is certified inline
            }
            if (DEFER && (!a.recs || o0 + rc.n <= a.rec_cap) && miss_left >= miss_need) {
                if (pair_used == PAIR_CHUNK) {  // wave-uniform
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(a.pair_cursor, (unsigned long long)PAIR_CHUNK);
                    base = ((unsigned long long)bcast_u32((uint32_t)(base >> 32), 0) << 32) | bcast_u32((uint32_t)base, 0);
                    if (base + PAIR_CHUNK <= a.pair_cap) {
                        pair_base = base;
                        pair_used = 0;
                    }
                }
                if (pair_used < PAIR_CHUNK) {
                    if (lane == 0) {
                        // (block mode: buckets by (block, mask) — the pairs of a chunk share their mask)
                        const uint32_t blk = col >> BLOCK_LEAVES_LOG2;
                        const uint32_t key = BLOCKS ? ((blk << 8) | mask8) : col;
                        const uint32_t bkt = (key << a.sub_log2) | ((uint32_t)r & ((1u << a.sub_log2) - 1u));
                        a.pairs[pair_base + pair_used] = make_uint2((uint32_t)r, BLOCKS ? (blk | (mask8 << 24)) : col);
                        atomicAdd(&a.bucket_cnt[bkt], 1u);
                        if (COUNTS && a.bucket_words) atomicAdd(&a.bucket_words[bkt], miss_need);
                    }
                    miss_left -= miss_need;
                    ++pair_used;
                    ++st_def;
                    if (a.recs && !prepared) {  // hash the read once; every slice of the verify reuses the records
                        prepared = true;
                        // 150 bp reads at k = 20..23 have 128 + (1..3) k-mers: a third hashing pass for two or three
                        // k-mers.  Such last windows are left to k_tail_records (four reads per pass).
                        const bool split_tail = a.batch_tails && has_batched_tail(rc.n, a.batch_tails);
                        for (uint64_t base = n_done; base < rc.n; base += WIN_KMERS) {
                            uint32_t cnt = (uint32_t)((rc.n - base) < WIN_KMERS ? (rc.n - base) : WIN_KMERS);
                            if (split_tail && base + WIN_KMERS > rc.n) break;
                            uint64_t h1 = w0_h1, h2 = w0_h2;
                            if (base != 0 || !have_w0) {
                                stage_window(lds, wave, rc.read, base, cnt, a.hp.k);
                                kmer_hashes(lds, wave, lane, cnt, lane < cnt, a.hp, h1, h2);
                            }
                            uint4 rec = make_probe_record(h1, h2, a.hp);
                            if (lane < cnt) a.recs[o0 + base + lane] = rec;
                        }
                    }
                    continue;
                }
                // no room left in the pair buffer: certify inline below (results stay exact)
            }
            if (!DEFER && a.screen_only) continue;  // (a sample of the block is screened to see how many candidates a read has)
            for (uint32_t jb = 0; jb < (BLOCKS ? 8u : 1u); ++jb) {  // (BLOCKS: the candidates of the block one by one)
                if (!((mask8 >> jb) & 1u)) continue;
                const uint32_t cj = col + jb;
                bool pass = verify_column(lds, wave, a, rc, cj);
                // ancestors that are not provably supersets must pass too (query.rs:119-141 visits children only with
                // reads that passed the parent)
                for (uint32_t g = a.guard_off[cj]; pass && g < a.guard_off[cj + 1]; ++g)
                    pass = verify_column(lds, wave, a, rc, a.guard_col[g]);
                if (pass) {
                    ++st_hits;
                    st_bytes += rc.need * a.hp.num_hashes * 32ull;
                    if (lane == 0) {
                        atomicAdd(&a.counts[cj], 1ull);
                        if (a.hit_pairs) {
                            unsigned long long pos = atomicAdd(a.hit_cursor, 1ull);
                            if (pos < a.hit_cap) a.hit_pairs[pos] = make_uint2((uint32_t)r, cj);
                        }
                    }
                }
            }
        }
    };

    // the reads of this launch: all of them, or (leaf group of a two-level frontier) the ones k_coarse listed for the group
    uint64_t n_work = a.n_reads;
    if (LIST) n_work = *a.n_list;
    const uint32_t *const read_list = LIST ? a.read_list : nullptr;
    if (DEFER && !COUNTS && a.rw >= 4u) {
        // groups of DENSE_READS consecutive reads: dense pre-screen, then the per-read path for the survivors
        const uint64_t n_groups = (n_work + DENSE_READS - 1) / DENSE_READS;
        for (uint64_t g = gw; g < n_groups; g += nw) {
            const uint64_t r0 = g * DENSE_READS;
            const uint32_t cnt = (uint32_t)(n_work - r0 < DENSE_READS ? n_work - r0 : DENSE_READS);
            uint64_t lane_len;
            uint32_t irregular, rid;
            uint32_t survive = dense_screen<false>(dlds.mini[wave][0], dlds.mini[wave][1], lds.comp, dlds.live[wave], a, read_list, r0, cnt,
                                                   colmask, 2u, irregular, lane_len, rid);
            if (!(((survive | irregular) >> (lane >> 2)) & 1u)) dense_bytes += lane_len;  // reads finished here still count their bytes
            while (survive) {
                const uint32_t jj = (uint32_t)__ffs((int)survive) - 1u;
                survive &= survive - 1u;
                process_read(LIST ? (uint64_t)bcast_u32(rid, (int)(jj * 4u)) : r0 + jj, dlds.live[wave] + jj * rw);
            }
            while (irregular) {
                const uint32_t jj = (uint32_t)__ffs((int)irregular) - 1u;
                irregular &= irregular - 1u;
                process_read(LIST ? (uint64_t)bcast_u32(rid, (int)(jj * 4u)) : r0 + jj, nullptr);
            }
        }
    } else if constexpr (COUNTS && !LONG && LPR_LOG2 >= 2u) {
        // thresholds < 1: groups of 64/(rw/4) consecutive reads through the dense counting screen
        const uint32_t rpw = 256u >> a.rw_log2, lpr_log2 = a.rw_log2 - 2u;
        const uint64_t n_groups = (n_work + rpw - 1) / rpw;
        for (uint64_t g = gw; g < n_groups; g += nw) {
            const uint64_t r0 = g * rpw;
            const uint32_t cnt = (uint32_t)(n_work - r0 < rpw ? n_work - r0 : rpw);
            uint64_t lane_len, rid;
            uint32_t survive, irregular, n_rec;
            dense_counts<8, LPR_LOG2, false, DEFER>(dlds.mini[wave][0], dlds.mini[wave][1], lds.comp, dlds.live[wave], a, read_list, r0, cnt, survive,
                                                    irregular, lane_len, rid, n_rec);
            if (!(((survive | irregular) >> (lane >> lpr_log2)) & 1u)) dense_bytes += lane_len;
            const uint32_t rid_lo = (uint32_t)rid;  // reads are indexed with 31 bits (query_device checks)
            while (survive) {
                const uint32_t jj = (uint32_t)__ffs((int)survive) - 1u;
                survive &= survive - 1u;
                process_read(LIST ? (uint64_t)bcast_u32(rid_lo, (int)(jj << lpr_log2)) : r0 + jj, dlds.live[wave] + jj * rw,
                             DEFER ? bcast_u32(n_rec, (int)(jj << lpr_log2)) : 0u);
            }
            if (irregular) irregular = queue_long_reads(a, irregular, lane_len, rid_lo, lpr_log2);
            while (irregular) {
                const uint32_t jj = (uint32_t)__ffs((int)irregular) - 1u;
                irregular &= irregular - 1u;
                process_read(LIST ? (uint64_t)bcast_u32(rid_lo, (int)(jj << lpr_log2)) : r0 + jj, nullptr);
            }
        }
    } else if constexpr (COUNTS && LONG && LPR_LOG2 >= 2u) {
        // the reads of >= 256 k-mers queued by the first launch: the same dense screen with 16 counter planes
        const uint32_t rpw = 256u >> a.rw_log2, lpr_log2 = a.rw_log2 - 2u;
        const uint64_t n_long = *a.n_long, n_groups = (n_long + rpw - 1) / rpw;
        for (uint64_t g = gw; g < n_groups; g += nw) {
            const uint64_t r0 = g * rpw;
            const uint32_t cnt = (uint32_t)(n_long - r0 < rpw ? n_long - r0 : rpw);
            uint64_t lane_len, rid;
            uint32_t survive, irregular, n_rec;
            dense_counts<NPLANES, LPR_LOG2, false, DEFER>(dlds.mini[wave][0], dlds.mini[wave][1], lds.comp, dlds.live[wave], a, a.long_list, r0, cnt,
                                                          survive, irregular, lane_len, rid, n_rec);
            if (!(((survive | irregular) >> (lane >> lpr_log2)) & 1u)) dense_bytes += lane_len;
            const uint32_t rid_lo = (uint32_t)rid;  // reads are indexed with 31 bits (query_device checks)
            while (survive) {
                const uint32_t jj = (uint32_t)__ffs((int)survive) - 1u;
                survive &= survive - 1u;
                process_read(bcast_u32(rid_lo, (int)(jj << lpr_log2)), dlds.live[wave] + jj * rw, DEFER ? bcast_u32(n_rec, (int)(jj << lpr_log2)) : 0u);
            }
            while (irregular) {
                const uint32_t jj = (uint32_t)__ffs((int)irregular) - 1u;
                irregular &= irregular - 1u;
                process_read(bcast_u32(rid_lo, (int)(jj << lpr_log2)), nullptr);
            }
        }
    } else if (LONG) {
        const uint64_t n_long = *a.n_long;
        for (uint64_t i = gw; i < n_long; i += nw) process_read(a.long_list[i], nullptr);
    } else if (LIST) {
        for (uint64_t i = gw; i < n_work; i += nw) {
            const uint32_t r = read_list[i];
            if (r != 0xffffffffu) process_read(r, nullptr);
        }
    } else {
        for (uint64_t r = gw; r < a.n_reads; r += nw) process_read(r, nullptr);
    }
    if (DEFER && pair_used < PAIR_CHUNK)
        for (uint32_t i = pair_used + lane; i < PAIR_CHUNK; i += 64) a.pairs[pair_base + i] = make_uint2(0xffffffffu, 0xffffffffu);
    // reads that pass every node (need == 0) count at every leaf (query.rs:143 reached through every path)
    if (st_all && !(!DEFER && a.screen_only))
        for (uint32_t c = lane; c < a.n_leaves; c += 64) atomicAdd(&a.counts[a.col0 + c], st_all);
    if (!a.first_group) st_all = 0;  // (statistics count a read once)
    for (int dd = 32; dd > 0; dd >>= 1) dense_bytes += __shfl_down(dense_bytes, dd);
    if (a.first_group) st_bytes += bcast_u32((uint32_t)dense_bytes, 0) | ((unsigned long long)bcast_u32((uint32_t)(dense_bytes >> 32), 0) << 32);
    if (lane == 0) {
        if (st_cand) atomicAdd(&a.stats[ST_CANDIDATES], st_cand);
        if (st_hits) atomicAdd(&a.stats[ST_HITS], st_hits);
        if (st_all) atomicAdd(&a.stats[ST_ALLHIT], st_all);
        if (st_bytes) atomicAdd(&a.stats[ST_ALG_BYTES], st_bytes);
        if (st_def) atomicAdd(&a.stats[ST_DEFERRED], st_def);
    }
}

template <bool DEFER, uint32_t LPR_LOG2, bool LIST>
static void launch_classify_counts(const QueryArgs &a, dim3 g, dim3 b, hipStream_t st) {
    if (DEFER && a.block_pairs) {
        hipLaunchKernelGGL((k_classify<DEFER, true, false, LPR_LOG2, DEFER, LIST>), g, b, 0, st, a);
        hipLaunchKernelGGL((k_classify<DEFER, true, true, LPR_LOG2, DEFER>), g, b, 0, st, a);
        return;
    }
    hipLaunchKernelGGL((k_classify<DEFER, true, false, LPR_LOG2, false, LIST>), g, b, 0, st, a);
    hipLaunchKernelGGL((k_classify<DEFER, true, true, LPR_LOG2>), g, b, 0, st, a);  // the reads of >= 256 k-mers it queued
}
template <bool LIST>
static void launch_classify_l(const QueryArgs &a, bool defer, bool counts_mode, dim3 g, dim3 b, hipStream_t st) {
    if (!counts_mode) {
        if (defer && a.block_pairs) hipLaunchKernelGGL((k_classify<true, false, false, 0, true, LIST>), g, b, 0, st, a);
        else if (defer) hipLaunchKernelGGL((k_classify<true, false, false, 0, false, LIST>), g, b, 0, st, a);
        else hipLaunchKernelGGL((k_classify<false, false, false, 0, false, LIST>), g, b, 0, st, a);
        return;
    }
    const uint32_t lp = (a.rw_log2 >= 4u && a.rw_log2 <= 6u) ? a.rw_log2 - 2u : 0u;
    if (defer) {
        if (lp == 2) launch_classify_counts<true, 2, LIST>(a, g, b, st);
        else if (lp == 3) launch_classify_counts<true, 3, LIST>(a, g, b, st);
        else if (lp == 4) launch_classify_counts<true, 4, LIST>(a, g, b, st);
        else launch_classify_counts<true, 0, LIST>(a, g, b, st);
    } else {
        if (lp == 2) launch_classify_counts<false, 2, LIST>(a, g, b, st);
        else if (lp == 3) launch_classify_counts<false, 3, LIST>(a, g, b, st);
        else if (lp == 4) launch_classify_counts<false, 4, LIST>(a, g, b, st);
        else launch_classify_counts<false, 0, LIST>(a, g, b, st);
    }
}
void launch_classify(const QueryArgs &a, bool defer, bool counts_mode, int blocks, hipStream_t st) {
    dim3 g(blocks, a.grid_groups ? a.grid_groups : 1u), b(256);
    if (a.read_list) launch_classify_l<true>(a, defer, counts_mode, g, b, st);
    else launch_classify_l<false>(a, defer, counts_mode, g, b, st);
}

// ---- the coarse level of a two-level frontier ----------------------------------------------------------------------------
// Trees of more than one column group: `a` describes the COARSE sliced matrix (columns = an antichain of internal nodes
// that covers every leaf, see CoarseArgs).  Every read goes through the dense screens against it — the AND-frontier at
// threshold 1, the miss-counting frontier below — with n_probes probes per k-mer; a read then joins the list of every
// leaf group that holds a leaf below one of its live columns, and the launches of the leaf groups (k_classify with
// read_list) screen, certify or defer exactly those reads.  Exact for any tree: a read that does not pass a node — and the
// screens only drop a column for bits that are 0 — reaches no leaf below it (query.rs:119-141), superset or not.
// This launch also does what is per read and not per leaf group: the reads' bytes (statistics), reads that pass every
// node (need == 0: they count at every leaf of the tree) and reads that pass none (need > n).  Reads the dense screens do
// not take (no screening possible) are listed for every group.
template <bool COUNTS, bool LONG, uint32_t LPR_LOG2>
__global__ void __launch_bounds__(256, 2) k_coarse(QueryArgs a, CoarseArgs ca) {
    __shared__ uint8_t s_comp[256];
    __shared__ DenseLds<true> dlds;
    __shared__ uint32_t s_lbase[WAVES_PER_BLOCK][MAX_LEAF_GROUPS], s_lused[WAVES_PER_BLOCK][MAX_LEAF_GROUPS];
    fill_complement(s_comp);
    for (uint32_t i = threadIdx.x; i < WAVES_PER_BLOCK * MAX_LEAF_GROUPS; i += blockDim.x) (&s_lused[0][0])[i] = PAIR_CHUNK;
    __syncthreads();
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint64_t gw = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + wave, nw = (uint64_t)gridDim.x * WAVES_PER_BLOCK;
    const uint32_t rw = a.rw;
    unsigned long long st_all = 0, st_bytes = 0, dense_bytes = 0, st_listed = 0;
    const uint64_t all_groups = ca.n_groups >= 64u ? ~0ull : ((1ull << ca.n_groups) - 1ull);

    // read r joins the lists of the leaf groups below its live coarse columns (pre: the frontier words of the read, in LDS;
    // nullptr: every group)
    auto emit = [&](uint32_t r, const uint32_t *pre) {
        uint64_t gset = all_groups;
        if (pre) {
            gset = 0;
            uint32_t w = lane < rw ? pre[lane] : 0u;
            while (w) {
                const uint32_t b = (uint32_t)__ffs((int)w) - 1u;
                w &= w - 1u;
                const uint32_t rng = ca.cgrp[lane * 32u + b], lo = rng & 0xffffu, hi = rng >> 16;
                gset |= (hi >= 63u ? ~0ull : ((1ull << (hi + 1u)) - 1ull)) & ~((1ull << lo) - 1ull);
            }
            for (int d = 32; d > 0; d >>= 1) gset |= __shfl_xor(gset, d);
        }
        gset = bcast_u64(gset, 0);
        while (gset) {
            const uint32_t g = (uint32_t)__ffsll((unsigned long long)gset) - 1u;
            gset &= gset - 1ull;
            uint32_t used = s_lused[wave][g];
            if (used == PAIR_CHUNK) {  // (wave-uniform) a new reservation of 32 slots in the group's list
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&ca.cursors[g], PAIR_CHUNK);
                base = bcast_u32(base, 0);
                if (lane == 0) s_lbase[wave][g] = base;
                used = 0;
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                ca.lists[(uint64_t)g * ca.list_cap + s_lbase[wave][g] + used] = r;
                s_lused[wave][g] = used + 1u;
            }
            ++st_listed;
            __builtin_amdgcn_wave_barrier();
        }
    };
    // a read the dense screen did not take
    auto slow = [&](uint32_t r, bool queued_before) {
        const uint64_t o0 = a.off[r], L = a.off[(uint64_t)r + 1] - o0;
        const uint64_t n = (L >= a.hp.k) ? (L - a.hp.k + 1) : 0, need = need_kmers(a.threshold, n);
        if (!queued_before) {
            if (need == 0) {  // 0 >= 0 at every node: the read reaches and counts at every leaf
                ++st_all;
                if (a.allhit_flag && lane == 0) a.allhit_flag[r] = 1;
                return;
            }
            if (need > n) return;  // cannot pass any node
            if (COUNTS && !LONG && n >= SHORT_KMERS) {  // the second launch screens it with wider counters
                if (lane == 0) a.long_list[atomicAdd(a.n_long, 1u)] = r;
                return;
            }
        }
        emit(r, nullptr);
    };
    // The survivors of a dense group (up to 64 / lanes-per-read reads), appended together: lane jj < n_rd owns read jj, walks
    // its frontier words (rotated by the lane: no LDS bank is hit twice) for the leaf groups below its live columns; then per
    // group ONE step appends all its reads — slots from the wave's reservation, a new one when it runs over.  (One read at a
    // time, each with its own LDS round trips, the appends cost more than the screens: 8.5 survivors per pass at theta = 1.)
    auto emit_batch = [&](uint32_t survive, uint32_t rid_lo, uint32_t lanes_log2) {
        const uint32_t n_rd = 64u >> lanes_log2;
        const bool mine = lane < n_rd && ((survive >> lane) & 1u);
        const uint32_t my_rid = (uint32_t)__shfl((int)rid_lo, (int)((lane < n_rd ? lane : 0u) << lanes_log2));
        uint64_t gset = 0;
        if (mine) {
            const uint32_t *pre = dlds.live[wave] + lane * rw;
            for (uint32_t i = 0; i < rw; ++i) {
                const uint32_t wi = (i + lane) & (rw - 1u);
                uint32_t w = pre[wi];
                while (w) {
                    const uint32_t b = (uint32_t)__ffs((int)w) - 1u;
                    w &= w - 1u;
                    const uint32_t rng = ca.cgrp[wi * 32u + b], lo = rng & 0xffffu, hi = rng >> 16;
                    gset |= (hi >= 63u ? ~0ull : ((1ull << (hi + 1u)) - 1ull)) & ~((1ull << lo) - 1ull);
                }
            }
        }
        uint64_t present = gset;
        for (int d = 32; d > 0; d >>= 1) present |= __shfl_xor(present, d);
        present = bcast_u64(present, 0);
        while (present) {
            const uint32_t g = (uint32_t)__ffsll((unsigned long long)present) - 1u;
            present &= present - 1ull;
            const bool in = mine && ((gset >> g) & 1ull);
            const uint64_t m = ballot64(in);
            const uint32_t c = (uint32_t)__popcll(m), rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            const uint32_t used = s_lused[wave][g], base_old = s_lbase[wave][g];
            uint32_t base_new = 0;
            if (used + c > PAIR_CHUNK) {  // (wave-uniform) the reservation runs over: a new one takes the rest
                if (lane == 0) base_new = atomicAdd(&ca.cursors[g], PAIR_CHUNK);
                base_new = bcast_u32(base_new, 0);
            }
            if (in) {
                const uint32_t p = used + rank;
                ca.lists[(uint64_t)g * ca.list_cap + (p < PAIR_CHUNK ? base_old + p : base_new + (p - PAIR_CHUNK))] = my_rid;
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                if (used + c > PAIR_CHUNK) {
                    s_lbase[wave][g] = base_new;
                    s_lused[wave][g] = used + c - PAIR_CHUNK;
                } else s_lused[wave][g] = used + c;
            }
            st_listed += c;
            __builtin_amdgcn_wave_barrier();
        }
    };
    auto finish_group = [&](uint32_t survive, uint32_t irregular, uint32_t rid_lo, uint32_t lanes_log2, uint64_t lane_len) {
        // (reads of >= 256 k-mers: queued for the second launch together; their bytes were counted with everybody's)
        if (COUNTS && !LONG && irregular) irregular = queue_long_reads(a, irregular, lane_len, rid_lo, lanes_log2);
        if (!COUNTS) {
            if (survive) emit_batch(survive, rid_lo, lanes_log2);
            survive = 0;
        }
        while (survive) {
            const uint32_t jj = (uint32_t)__ffs((int)survive) - 1u;
            survive &= survive - 1u;
            emit(bcast_u32(rid_lo, (int)(jj << lanes_log2)), dlds.live[wave] + jj * rw);
        }
        while (irregular) {
            const uint32_t jj = (uint32_t)__ffs((int)irregular) - 1u;
            irregular &= irregular - 1u;
            slow(bcast_u32(rid_lo, (int)(jj << lanes_log2)), LONG);
        }
    };
    if constexpr (!COUNTS) {
        uint32_t colmask = 0;  // (unused by the dense screen's survivors; kept for its signature)
        const uint64_t n_groups = (a.n_reads + DENSE_READS - 1) / DENSE_READS;
        for (uint64_t g = gw; g < n_groups; g += nw) {
            const uint64_t r0 = g * DENSE_READS;
            const uint32_t cnt = (uint32_t)(a.n_reads - r0 < DENSE_READS ? a.n_reads - r0 : DENSE_READS);
            uint64_t lane_len;
            uint32_t irregular, rid;
            const uint32_t survive = dense_screen<true>(dlds.mini[wave][0], dlds.mini[wave][1], s_comp, dlds.live[wave], a, nullptr, r0, cnt,
                                                        colmask, ca.n_probes, irregular, lane_len, rid);
            dense_bytes += lane_len;
            finish_group(survive, irregular, rid, 2u, lane_len);
        }
    } else {
        constexpr uint32_t lpr_log2 = LPR_LOG2, rpw = 64u >> lpr_log2;
        const uint64_t n_work = LONG ? (uint64_t)*a.n_long : a.n_reads;
        const uint64_t n_groups = (n_work + rpw - 1) / rpw;
        for (uint64_t g = gw; g < n_groups; g += nw) {
            const uint64_t r0 = g * rpw;
            const uint32_t cnt = (uint32_t)(n_work - r0 < rpw ? n_work - r0 : rpw);
            uint64_t lane_len, rid;
            uint32_t survive, irregular, n_rec;
            dense_counts<LONG ? NPLANES : 8, LPR_LOG2, true>(dlds.mini[wave][0], dlds.mini[wave][1], s_comp, dlds.live[wave], a,
                                                            LONG ? a.long_list : nullptr, r0, cnt, survive, irregular, lane_len, rid,
                                                            n_rec, ca.n_probes, ca.scr_extra);
            if (!LONG) dense_bytes += lane_len;  // (the first launch saw every read)
            finish_group(survive, irregular, (uint32_t)rid, lpr_log2, lane_len);
        }
    }
    // unused slots of the last reservations
    for (uint32_t g = 0; g < ca.n_groups; ++g) {
        const uint32_t used = s_lused[wave][g];
        if (used < PAIR_CHUNK && used + lane < PAIR_CHUNK) ca.lists[(uint64_t)g * ca.list_cap + s_lbase[wave][g] + used + lane] = 0xffffffffu;
    }
    if (st_all)
        for (uint32_t c = lane; c < ca.total_leaves; c += 64) atomicAdd(&a.counts[c], st_all);
    for (int dd = 32; dd > 0; dd >>= 1) dense_bytes += __shfl_down(dense_bytes, dd);
    st_bytes += bcast_u64(dense_bytes, 0);
    if (lane == 0) {
        if (st_all) atomicAdd(&a.stats[ST_ALLHIT], st_all);
        if (st_bytes) atomicAdd(&a.stats[ST_ALG_BYTES], st_bytes);
        if (st_listed) atomicAdd(&a.stats[ST_LISTED], st_listed);
    }
}
template <uint32_t LPR_LOG2>
static void launch_coarse_counts(const QueryArgs &a, const CoarseArgs &ca, dim3 g, dim3 b, hipStream_t st) {
    hipLaunchKernelGGL((k_coarse<true, false, LPR_LOG2>), g, b, 0, st, a, ca);
    hipLaunchKernelGGL((k_coarse<true, true, LPR_LOG2>), g, b, 0, st, a, ca);  // the reads of >= 256 k-mers it queued
}
void launch_coarse(const QueryArgs &a, const CoarseArgs &ca, bool counts_mode, int blocks, hipStream_t st) {
    dim3 g(blocks), b(256);
    if (!counts_mode) {
        hipLaunchKernelGGL((k_coarse<false, false, 0>), g, b, 0, st, a, ca);
        return;
    }
    // (the host builds coarse matrices with rows of 16, 32 or 64 words)
    if (a.rw_log2 == 4u) launch_coarse_counts<2>(a, ca, g, b, st);
    else if (a.rw_log2 == 5u) launch_coarse_counts<3>(a, ca, g, b, st);
    else launch_coarse_counts<4>(a, ca, g, b, st);
}

// Guard columns on the bucketed path.  k_classify<DEFER> defers (read, leaf) pairs only; for trees with guard columns
// (ancestors whose parent ⊇ child check failed: reference-built trees with colliding node names) this kernel
// walks the deferred pairs and gives every guard of the pair's leaf a pair of its own, (read, guard column), in a second
// region of the pair buffer.  owner[slot] names the leaf pair a slot belongs to (a leaf pair owns itself); k_finalize
// lets a leaf pair pass only if none of its guard pairs failed (query.rs:119-141: children are visited only with the
// reads that passed the parent).  A guard pair that finds no room is certified here, inline, and a failure is written
// to gfail directly — results never depend on the capacities.  One wave per pair with guards; cold: trees whose edges
// are all verified never launch it.
__global__ void __launch_bounds__(256) k_expand_guards(QueryArgs a, GuardArgs ga) {
    __shared__ BlockLds lds;
    fill_complement(lds.comp);
    __syncthreads();
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    unsigned long long n_slots = *a.pair_cursor;
    if (n_slots > a.pair_cap) n_slots = a.pair_cap;
    const uint64_t gw = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + wave, nw = (uint64_t)gridDim.x * WAVES_PER_BLOCK;
    unsigned long long g_base = 0;
    uint32_t g_used = PAIR_CHUNK, miss_left = 0;
    for (uint64_t s0 = gw * 64u; s0 < n_slots; s0 += nw * 64u) {
        const uint64_t slot = s0 + lane;
        uint2 p = make_uint2(0xffffffffu, 0xffffffffu);
        if (slot < n_slots) p = a.pairs[slot];
        const bool valid = p.y != 0xffffffffu;
        if (valid) ga.owner[slot] = (uint32_t)slot;  // a leaf pair owns itself
        const uint32_t g0 = valid ? a.guard_off[p.y] : 0u, ng = valid ? a.guard_off[p.y + 1] - g0 : 0u;
        uint64_t todo = ballot64(ng != 0);
        while (todo) {
            const int src = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const uint32_t r = bcast_u32(p.x, src), g_lo = bcast_u32(g0, src), n_guard = bcast_u32(ng, src);
            const uint32_t own = (uint32_t)(s0 + (uint32_t)src);
            const uint64_t o0 = a.off[r], L = a.off[r + 1] - o0;
            ReadCtx rc;
            rc.read = a.seq + o0;
            rc.n = L - a.hp.k + 1;  // deferred reads have k-mers and 1 <= need <= n
            rc.need = need_kmers(a.threshold, rc.n);
            rc.maxmiss = rc.n - rc.need;
            const uint32_t miss_one = a.bucket_words ? (uint32_t)((rc.n + 63) >> 6) : 0u;  // (deferred reads have < 2^30 k-mers)
            for (uint32_t gg = 0; gg < n_guard; gg += PAIR_CHUNK) {
                const uint32_t cnt = n_guard - gg < PAIR_CHUNK ? n_guard - gg : PAIR_CHUNK;
                const uint32_t miss_need = miss_one * cnt;
                if (a.bucket_words && miss_left < miss_need) {
                    const uint32_t want = miss_need > MISS_RESERVE ? miss_need : MISS_RESERVE;
                    unsigned long long got = 0;
                    if (lane == 0) got = atomicAdd(a.miss_cursor, (unsigned long long)want);
                    got = ((unsigned long long)bcast_u32((uint32_t)(got >> 32), 0) << 32) | bcast_u32((uint32_t)got, 0);
                    if (got + want <= a.miss_cap) miss_left = want;
                }
                bool placed = false;
                if (miss_left >= miss_need) {
                    if (g_used + cnt > PAIR_CHUNK) {  // what is left of the reservation is voided, a new one taken
                        for (uint32_t i = g_used + lane; i < PAIR_CHUNK; i += 64) ga.pairs[g_base + i] = make_uint2(0xffffffffu, 0xffffffffu);
                        g_used = PAIR_CHUNK;
                        unsigned long long base = 0;
                        if (lane == 0) base = atomicAdd(ga.cursor, (unsigned long long)PAIR_CHUNK);
                        base = ((unsigned long long)bcast_u32((uint32_t)(base >> 32), 0) << 32) | bcast_u32((uint32_t)base, 0);
                        if (base + PAIR_CHUNK <= ga.cap) {
                            g_base = base;
                            g_used = 0;
                        }
                    }
                    if (g_used + cnt <= PAIR_CHUNK) {
                        if (lane < cnt) {
                            const uint32_t c = a.guard_col[g_lo + gg + lane];
                            const uint32_t bkt = (c << a.sub_log2) | (r & ((1u << a.sub_log2) - 1u));
                            ga.pairs[g_base + g_used + lane] = make_uint2(r, c);
                            ga.owner[ga.slot0 + g_base + g_used + lane] = own;
                            atomicAdd(&a.bucket_cnt[bkt], 1u);
                            if (a.bucket_words) atomicAdd(&a.bucket_words[bkt], miss_one);
                        }
                        g_used += cnt;
                        miss_left -= miss_need;
                        placed = true;
                    }
                }
                if (!placed) {  // no room: certify these guards here
                    bool pass = true;
                    for (uint32_t u = 0; pass && u < cnt; ++u) pass = verify_column(lds, wave, a, rc, a.guard_col[g_lo + gg + u]);
                    if (!pass && lane == 0) ga.gfail[own] = 1u;
                }
            }
        }
    }
    if (g_used < PAIR_CHUNK)
        for (uint32_t i = g_used + lane; i < PAIR_CHUNK; i += 64) ga.pairs[g_base + i] = make_uint2(0xffffffffu, 0xffffffffu);
}
void launch_expand_guards(const QueryArgs &a, const GuardArgs &ga, int blocks, hipStream_t st) {
    hipLaunchKernelGGL(k_expand_guards, dim3(blocks), dim3(256), 0, st, a, ga);
}

// Records of the last windows k_classify<DEFER> left out (has_batched_tail): one pass serves 64 / TAIL deferred pairs,
// lane = (pair j, k-mer t of up to TAIL); the TAIL = 32 build takes the tails of 17 .. 32 k-mers, the TAIL = 16 build the
// shorter ones.  Walks the deferred-pair buffer; a read deferred for two leaves gets its tail records written twice (same
// values).
template <uint32_t TAIL>
__global__ void __launch_bounds__(256) k_tail_records(QueryArgs a) {
    constexpr uint32_t PAIRS = 64u / TAIL, STRIDE = 384u / PAIRS, LOADS = (TAIL + KMAX - 1u + TAIL - 1u) / TAIL;  // bytes per pair in LDS; byte loads per lane
    static_assert(WIN_PAD + TAIL + KMAX - 1u + WIN_PAD <= STRIDE + WIN_PAD, "a tail's bytes fit its share of the LDS");
    __shared__ uint32_t s_fw[WAVES_PER_BLOCK][4 * 96 / 4 + 4], s_rc[WAVES_PER_BLOCK][4 * 96 / 4 + 4];
    __shared__ uint8_t s_comp[256];
    fill_complement(s_comp);
    __syncthreads();
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6, k = a.hp.k;
    const uint32_t j = lane / TAIL, t = lane % TAIL;
    unsigned long long n_slots = *a.pair_cursor;
    if (n_slots > a.pair_cap) n_slots = a.pair_cap;
    uint32_t *tfw = s_fw[wave], *trc = s_rc[wave];
    uint8_t *tfwd = reinterpret_cast<uint8_t *>(tfw), *trcb = reinterpret_cast<uint8_t *>(trc);
    const uint64_t gw = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + wave, nw = (uint64_t)gridDim.x * WAVES_PER_BLOCK;
    for (uint64_t s0 = gw * PAIRS; s0 < n_slots; s0 += nw * PAIRS) {
        const uint64_t slot = s0 + j;
        uint32_t r = 0xffffffffu;
        if (slot < n_slots) r = a.pairs[slot].x;
        uint64_t o0 = 0, n = 0;
        if (r != 0xffffffffu) {
            o0 = a.off[r];
            const uint64_t L = a.off[r + 1] - o0;
            n = (L >= k) ? (L - k + 1) : 0;
        }
        const uint32_t tl = (uint32_t)(n & (WIN_KMERS - 1u));
        // (tails of up to 16 k-mers belong to the TAIL = 16 build, longer ones to the TAIL = 32 build)
        const bool have = r != 0xffffffffu && has_batched_tail(n, a.batch_tails) && (TAIL == 16 ? tl <= 16u : tl > 16u) && o0 + n <= a.rec_cap;
        if (ballot64(have) == 0) continue;
        const uint64_t base = n - tl;
        const uint32_t W = have ? tl + k - 1u : 0u;  // <= TAIL + KMAX - 1 bytes
        const uint32_t mb = j * STRIDE + WIN_PAD;
        const uint8_t *src = a.seq + (have ? o0 + base : 0ull);
        __builtin_amdgcn_wave_barrier();
        uint8_t b[LOADS];
#pragma unroll
        for (uint32_t u = 0; u < LOADS; ++u) {
            const uint32_t idx = TAIL * u + t;
            b[u] = src[idx < W ? idx : 0u];
        }
#pragma unroll
        for (uint32_t u = 0; u < LOADS; ++u) {
            const uint32_t idx = TAIL * u + t;
            if (idx < W) {
                tfwd[mb + idx] = b[u];
                trcb[mb + (W - 1u - idx)] = s_comp[b[u]];
            }
        }
        __builtin_amdgcn_wave_barrier();
        const bool valid = have && t < tl;
        uint64_t h1, h2;
        kmer_hashes_at(tfw, trc, mb + t, mb + (W - t - k), valid, a.hp, h1, h2);
        const uint4 rec = make_probe_record(h1, h2, a.hp);
        if (valid) a.recs[o0 + base + t] = rec;
    }
}
void launch_tail_records(const QueryArgs &a, int blocks, hipStream_t st) {
    hipLaunchKernelGGL(k_tail_records<16>, dim3(blocks), dim3(256), 0, st, a);
    if (a.batch_tails > 16u) hipLaunchKernelGGL(k_tail_records<32>, dim3(blocks), dim3(256), 0, st, a);
}

// ---- bucketing of deferred (read, leaf) pairs by leaf --------------------------------------------------------------
__global__ void __launch_bounds__(64) k_bucket_scan(const uint32_t *cnt, uint32_t *off, uint32_t *cur, uint32_t n) {
    // one wave; n <= 2048 leaves: lane l owns entries [l*per, (l+1)*per)
    uint32_t lane = lane_id(), per = (n + 63u) / 64u;
    uint32_t lo = lane * per, hi = lo + per < n ? lo + per : n;
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += cnt[i];
    uint32_t incl = s;
    for (uint32_t d = 1; d < 64; d <<= 1) {
        uint32_t t = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += t;
    }
    uint32_t run = incl - s;
    for (uint32_t i = lo; i < hi; ++i) {
        off[i] = run;
        cur[i] = 0;
        run += cnt[i];
    }
    if (lane == 63) off[n] = incl;
}
void launch_bucket_scan(const uint32_t *bucket_cnt, uint32_t *bucket_off, uint32_t *bucket_cur, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(64), 0, st, bucket_cnt, bucket_off, bucket_cur, n);
}

// Scatter into leaf order; `meta` gets everything the record-driven verify needs about a pair in one 16-byte
// entry (read byte offset, read length, filter row) so that kernel has no dependent metadata loads.
__global__ void __launch_bounds__(256) k_bucket_scatter(const uint2 *pairs, const unsigned long long *n_pairs_ptr,
                                                        uint64_t pair_cap, const uint32_t *off, uint32_t *cur, uint32_t sub_log2,
                                                        uint2 *sorted, uint4 *meta, const uint64_t *read_off,
                                                        const uint32_t *col_row, const uint32_t *words_off,
                                                        uint32_t *words_cur, uint32_t *miss_pos, uint32_t kmer_size,
                                                        const uint32_t *owner, uint32_t *owner_sorted, uint32_t key_mode) {
    uint64_t n = *n_pairs_ptr;
    if (n > pair_cap) n = pair_cap;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint2 p = pairs[i];
        if (p.y == 0xffffffffu) continue;  // voided slot of a partially used reservation
        // (block mode: p.y = block | mask << 24; keyed by the block, or — k-mer entries — by (block, mask))
        const uint32_t key = key_mode == 0 ? p.y : (key_mode == 2 ? (((p.y & 0xffffffu) << 8) | (p.y >> 24)) : (p.y & 0xffffffu));
        const uint32_t bkt = (key << sub_log2) | (p.x & ((1u << sub_log2) - 1u));
        uint32_t pos = off[bkt] + atomicAdd(&cur[bkt], 1u);
        sorted[pos] = p;
        if (owner) owner_sorted[pos] = owner[i];
        if (meta) {
            uint64_t o0 = read_off[p.x], L = read_off[p.x + 1] - o0;
            meta[pos] = make_uint4((uint32_t)o0, (uint32_t)(o0 >> 32), (uint32_t)L, key_mode == 0 ? col_row[p.y] : p.y);
            if (miss_pos) {  // thresholds < 1: the pair's miss words, ceil(n/64) of them, inside its bucket's range
                const uint32_t words = (uint32_t)((L - kmer_size + 1 + 63) >> 6);
                miss_pos[pos] = words_off[bkt] + atomicAdd(&words_cur[bkt], words);
            }
        }
    }
}
void launch_bucket_scatter(const uint2 *pairs, const unsigned long long *n_pairs_ptr, uint64_t pair_cap,
                           const uint32_t *bucket_off, uint32_t *bucket_cur, uint32_t sub_log2, uint2 *sorted, uint4 *meta,
                           const uint64_t *read_off, const uint32_t *col_row, const uint32_t *words_off, uint32_t *words_cur,
                           uint32_t *miss_pos, uint32_t kmer_size, const uint32_t *owner, uint32_t *owner_sorted, uint32_t key_mode,
                           int blocks, hipStream_t st) {
    hipLaunchKernelGGL(k_bucket_scatter, dim3(blocks), dim3(256), 0, st, pairs, n_pairs_ptr, pair_cap, bucket_off, bucket_cur,
                       sub_log2, sorted, meta, read_off, col_row, words_off, words_cur, miss_pos, kmer_size, owner, owner_sorted, key_mode);
}

// ---- K2 for bucketed survivors: L2-resident filter slices ------------------------------------------------------------
// The pairs are sorted by leaf.  Every block serves one slice of the filters' bit range at a time and pulls
// items (4*chunk consecutive sorted pairs) from that slice's queue, so all blocks of a slice stay within a window
// of (blocks per slice)*item pairs = about one leaf bucket (a static split drifts apart and thrashes the L2).
// Blocks are dealt to XCDs round-robin, so with home slice = XCC_ID % n_slices all CUs of an XCD gather from
// the same <= 2.5 MB slice of the current leaf, which therefore stays in that XCD's 4 MiB L2.  Placement only
// affects speed: every (item, slice) is taken exactly once whatever the placement, and a block whose home queue
// is drained helps with the other slices.  theta == 1 only: every probed bit must be set, so slices are
// independent and any miss fails the pair.
__global__ void __launch_bounds__(256) k_verify(VerifyArgs a) {
    __shared__ BlockLds lds;
    __shared__ uint32_t s_item;
    fill_complement(lds.comp);
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint64_t n_pairs = *a.n_pairs_ptr;  // = bucket_off[n_leaves]: deferred pairs, voided slots excluded
    const uint32_t per_wave = a.chunk, item_pairs = per_wave * WAVES_PER_BLOCK;
    const uint64_t n_items = (n_pairs + item_pairs - 1) / item_pairs;
    const uint32_t home = xcc_id() % a.n_slices;
    for (uint32_t att = 0; att < a.n_slices; ++att) {
        const uint32_t s = (home + att) % a.n_slices;
        const uint32_t lo = s * a.slice_bits;
        while (true) {
            __syncthreads();  // everyone is done with the previous item (and, first time, the complement table is filled)
            if (threadIdx.x == 0) s_item = atomicAdd(&a.queue[s], 1u);
            __syncthreads();
            const uint64_t it = s_item;
            if (it >= n_items) break;
            uint64_t e0 = it * item_pairs + (uint64_t)wave * per_wave, e1 = e0 + per_wave < n_pairs ? e0 + per_wave : n_pairs;
            for (uint64_t e = e0; e < e1; ++e) {
                const uint2 p = a.sorted[e];
                const uint64_t o0 = a.off[p.x], L = a.off[p.x + 1] - o0;
                const uint64_t n = L - a.hp.k + 1;  // candidates always have n >= 1
                const uint8_t *read = a.seq + o0;
                const uint32_t *bm = reinterpret_cast<const uint32_t *>(a.bits + (uint64_t)a.col_row[p.y] * a.n_words);
                bool miss = false;
                for (uint64_t base = 0; base < n; base += WIN_KMERS) {
                    uint32_t cnt = (uint32_t)((n - base) < WIN_KMERS ? (n - base) : WIN_KMERS);
                    stage_window(lds, wave, read, base, cnt, a.hp.k);
                    bool valid = lane < cnt;
                    uint64_t h1, h2;
                    kmer_hashes(lds, wave, lane, cnt, valid, a.hp, h1, h2);
                    ProbeIter pit;
                    pit.init(h1, h2, a.hp);
                    uint32_t ok = 1;
                    for_each_probe(pit, a.hp, [&](uint32_t idx) {
                        bool in = valid && (idx - lo) < a.slice_bits;
                        uint32_t v = in ? bm[idx >> 5] : ~0u;
                        ok &= v >> (idx & 31u);
                    });
                    miss = !(ok & 1u);
                    if (ballot64(miss)) { miss = true; break; }
                }
                if (miss && lane == 0) a.fail[e] = 1u;
            }
        }
    }
}
// Record-driven variant: the probe records written by k_classify<DEFER> replace all hashing.  Built for
// memory-level parallelism: per pair, the records of three windows and their 3*num_hashes probes are independent
// loads issued back to back; the item loop is software-pipelined (the queue pull for item t+2 and the metadata
// load for item t+1 are in flight while item t is probed; one barrier per item on a double-buffered LDS slot).
// One queue per XCD (8): XCD q serves slice q % n_slices and, of that slice's items, those congruent to
// q / n_slices modulo the XCDs per slice; XCDs never share a queue (each has its own L2, so they need not move
// in step).  A block whose own queue is drained helps the others.
__global__ void __launch_bounds__(1024) k_verify_rec(VerifyArgs a) {
    __shared__ uint32_t s_item[2];
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    // Fallback after the LDS-tile passes, two launches of which at most one works: few flagged pairs (bin or bucket
    // overflows) are taken from the compact list (mode 1); many, or chunks whose pass was not launched, are found by
    // walking all sorted pairs in leaf order and skipping the certified ones, which keeps the slices L2-resident (mode 2).
    bool leftover = false;
    if (a.only_flagged) {
        const unsigned int nf = *a.n_flagged;
        leftover = a.entry_cursor && *a.entry_cursor > (unsigned long long)a.launched_passes * a.entry_cap;
        // (thresholds < 1: the list is k_collect_open's — every pair with a non-zero fail word, in sorted order — and is
        // used whatever its length: a walk over all pairs leaves most waves of an item idle when few pairs are open)
        const bool many = a.miss_words ? nf > a.flag_cap : ((uint64_t)nf * 16u > (uint64_t)*a.n_pairs_ptr || nf > a.flag_cap);
        const bool scan = many || leftover;
        if (a.only_flagged == 1 ? (nf == 0 || scan) : !scan) return;
    }
    const bool from_list = a.only_flagged == 1, skip_certified = a.only_flagged == 2;
    const uint64_t n_pairs = from_list ? *a.n_flagged : *a.n_pairs_ptr;
    // an item = (waves per block) x chunk consecutive pairs, pulled with ONE queue atomic (a same-address returning
    // atomic costs ~90 ns at the memory side, so items must be large) while the pairs in flight per XCD
    // (blocks per XCD x item) stay at about one leaf's share
    const uint32_t per_wave = a.chunk < 64u ? a.chunk : 64u, item_pairs = per_wave * (blockDim.x >> 6);
    const uint64_t n_items = (n_pairs + item_pairs - 1) / item_pairs;
    const uint32_t home = xcc_id() & 7u, groups = 8u / a.n_slices;
    const uint32_t d = (uint32_t)a.hp.nbits, dw = d - (uint32_t)a.hp.w64, k = a.hp.k, nh = a.hp.num_hashes;

    auto pair_index = [&](uint64_t i) -> uint32_t { return from_list ? a.flag_list[i] : (uint32_t)i; };
    auto load_meta = [&](uint64_t it, uint32_t &e_idx) -> uint4 {
        const uint64_t i = it * item_pairs + (uint64_t)wave * per_wave + lane;
        bool ok = it < n_items && lane < per_wave && i < n_pairs;
        e_idx = ok ? pair_index(i) : 0u;
        // certified by a tile pass ... if its pass was launched.  Thresholds < 1: a tile pass only certifies pairs with no
        // k-mer missing (fail == 0, miss words stay zero); a pair with bit 0 set still needs its missing k-mers counted here
        if (ok && skip_certified && !(a.fail[e_idx] & (a.miss_words ? 3u : 2u))) {
            bool pending = false;
            if (leftover) {
                const uint32_t c = a.pair_chunk[e_idx];
                pending = c != 0xffffffffu && a.chunks[c].cap != 0 && a.chunks[c].pass >= a.launched_passes;
            }
            ok = pending;
        }
        return ok ? a.meta[e_idx] : make_uint4(0, 0, 0, 0);  // length 0 marks a pair to skip
    };

    // Every XCD's item sequence is dealt round-robin to n_sub sub-queues (own counter each) so that no counter
    // sees more than a few tens of thousands of pulls per launch; the sub-queues of an XCD advance at the same
    // average rate, so the XCD still works on one leaf's share at a time.
    const uint32_t n_sub = a.n_sub, home_sub = (blockIdx.x >> 3) % n_sub;
    for (uint32_t att = 0; att < 8u * n_sub; ++att) {
        const uint32_t sub = (home_sub + att) % n_sub, q = (home + att / n_sub) & 7u;
        const uint32_t s = q % a.n_slices, part = q / a.n_slices;
        const uint32_t lo = s * a.slice_bits;
        unsigned int *qctr = &a.queue[q * n_sub + sub];
        const uint64_t stride = (uint64_t)groups * n_sub, first = (uint64_t)sub * groups + part;
        __syncthreads();
        if (threadIdx.x == 0) {
            s_item[0] = atomicAdd(qctr, 1u);
            s_item[1] = atomicAdd(qctr, 1u);
        }
        __syncthreads();
        uint64_t it_cur = (uint64_t)s_item[0] * stride + first, it_nxt = (uint64_t)s_item[1] * stride + first;
        __syncthreads();
        uint32_t idx_cur, idx_nxt;
        uint4 meta_cur = load_meta(it_cur, idx_cur);
        uint32_t buf = 0;
        while (it_cur < n_items) {
            uint32_t pend = 0;
            if (threadIdx.x == 0) pend = atomicAdd(qctr, 1u);  // item t+2
            const uint4 meta_nxt = load_meta(it_nxt, idx_nxt);          // item t+1
            const uint64_t e0 = it_cur * item_pairs + (uint64_t)wave * per_wave;
            const uint32_t W = e0 >= n_pairs ? 0u : (uint32_t)(e0 + per_wave < n_pairs ? per_wave : n_pairs - e0);
            for (uint32_t j = 0; j < W; ++j) {
                const uint32_t len_j = bcast_u32(meta_cur.z, j);
                if (len_j == 0) continue;  // already certified (mode 2)
                const uint64_t o0 = ((uint64_t)bcast_u32(meta_cur.y, j) << 32) | bcast_u32(meta_cur.x, j);
                const uint64_t n = (uint64_t)len_j - k + 1;  // candidates always have n >= 1
                const uint4 *rp = a.recs + o0;
                const uint32_t row = bcast_u32(meta_cur.w, j);
                const uint32_t *bm = reinterpret_cast<const uint32_t *>(a.bits + (uint64_t)row * a.n_words);
                bool miss = false;
                for (uint64_t g0 = 0; g0 < n; g0 += 3 * WIN_KMERS) {
                    uint4 rec[3];
                    bool valid[3];
#pragma unroll
                    for (int w = 0; w < 3; ++w) {
                        const uint64_t qq = g0 + 64u * w + lane;
                        valid[w] = qq < n;
                        rec[w] = valid[w] ? rp[qq] : make_uint4(0, 0, 0, 0);
                    }
                    // probes 0..2 of the three windows: nine independent gathers, then the walk two probes at a
                    // time per window (six gathers in flight); uses come after the loads of a batch
                    RecordIter rit[3];
                    uint32_t okw[3] = {1u, 1u, 1u};  // per window: AND of this lane's probed bits in the slice
                    {
                        uint32_t ix[9], vv[9];
#pragma unroll
                        for (int w = 0; w < 3; ++w) {
                            rit[w].init(rec[w]);
                            ix[3 * w] = rit[w].i0;
                            ix[3 * w + 1] = rit[w].g;
                            ix[3 * w + 2] = rit[w].x;
                        }
#pragma unroll
                        for (int u = 0; u < 9; ++u) {
                            const bool in = valid[u / 3] && (u % 3) < (int)nh && (ix[u] - lo) < a.slice_bits;
                            vv[u] = in ? bm[ix[u] >> 5] : ~0u;
                        }
#pragma unroll
                        for (int u = 0; u < 9; ++u) okw[u / 3] &= vv[u] >> (ix[u] & 31u);
                    }
                    for (uint32_t i = 3; i < nh; i += 2) {
                        const bool second = i + 1 < nh;  // wave-uniform
                        uint32_t ix[6], vv[6];
#pragma unroll
                        for (int w = 0; w < 3; ++w) {
                            ix[2 * w] = rit[w].step(d, dw);
                            ix[2 * w + 1] = second ? rit[w].step(d, dw) : 0u;
                        }
#pragma unroll
                        for (int u = 0; u < 6; ++u) {
                            const bool in = valid[u / 2] && ((u & 1) == 0 || second) && (ix[u] - lo) < a.slice_bits;
                            vv[u] = in ? bm[ix[u] >> 5] : ~0u;
                        }
#pragma unroll
                        for (int u = 0; u < 6; ++u) okw[u / 2] &= vv[u] >> (ix[u] & 31u);
                    }
                    if (a.miss_words) {  // thresholds < 1: record which k-mers are not contained (OR: a k-mer can miss in several slices)
                        const uint32_t mp = a.miss_pos[bcast_u32(idx_cur, j)];
#pragma unroll
                        for (int w = 0; w < 3; ++w) {
                            const uint64_t mm = ballot64(valid[w] && !(okw[w] & 1u));
                            if (mm && lane == 0) atomicOr(&a.miss_words[mp + (uint32_t)(g0 >> 6) + (uint32_t)w], (unsigned long long)mm);
                        }
                    } else {
                        miss = miss || !(okw[0] & okw[1] & okw[2] & 1u);
                    }
                }
                if (ballot64(miss) && lane == 0) atomicOr(&a.fail[bcast_u32(idx_cur, j)], 1u);
            }
            if (threadIdx.x == 0) s_item[buf] = pend;
            lds_barrier();  // (not __syncthreads: the next item's metadata load stays in flight)
            const uint64_t it_new = (uint64_t)s_item[buf] * stride + first;
            buf ^= 1u;
            it_cur = it_nxt;
            meta_cur = meta_nxt;
            idx_cur = idx_nxt;
            it_nxt = it_new;
        }
    }
}
// ---- LDS-tile certificates ------------------------------------------------------------------------------------------------
// (1) plan: one block per leaf cuts the leaf's sorted pairs into chunks of <= 1024, sizes the chunk's per-tile
//     buckets from its k-mer count and reserves them; (2) bin: a block takes whole chunks, regenerates every probe
//     index of their pairs from the records and appends (local pair, offset in tile) to LDS bins, flushed per round to
//     the (chunk, tile) buckets in full runs; (3) test: a block loads one tile of one leaf into LDS (128 KiB) and tests
//     all probes binned for it.  A probe found 0 sets bit 0 of the pair's fail word.  Whatever cannot be binned
//     (bucket or entry buffer full) sets bit 1: those pairs are certified by k_verify_rec afterwards.
// Thresholds < 1: a pair passes as soon as `need` of its k-mers are known to be contained, so the passes bin only a PREFIX
// of every read's k-mers — need + slack, the slack covering the k-mers of two sequencing errors (or need / 2) — and a
// pair whose prefix holds fewer than `need` contained k-mers is left to the record kernel, which counts all of them
// (k_prefix_open flags it).  At threshold 0.3 that is 81 of the 130 k-mers of a 150 bp read.  Same result as counting
// everything: the count only grows with more k-mers (query_passes, query.rs:38-49).
__device__ __forceinline__ uint32_t prefix_kmers(float threshold, uint32_t n, uint32_t k) {
    const uint64_t need = need_kmers(threshold, n);
    const uint64_t slack = need / 2 > 2ull * k ? need / 2 : 2ull * k;
    return need + slack < n ? (uint32_t)(need + slack) : n;
}
// Where a pair's miss information is after the tile passes with k-mer entries: in its chunk's miss bytes (true; kbase = the
// chunk's first byte) when a launched pass binned it and nothing flagged it, else in its own miss words (k_verify_rec).
__device__ __forceinline__ bool pair_in_chunk(const uint32_t *fail, const uint32_t *pair_chunk, const ChunkDesc *chunks,
                                              uint32_t launched_passes, uint32_t e, uint32_t &kbase) {
    if (fail[e] & 2u) return false;
    const uint32_t c = pair_chunk[e];
    if (c == 0xffffffffu) return false;
    const ChunkDesc dsc = chunks[c];
    kbase = dsc.kbase;
    return dsc.cap != 0 && dsc.pass < launched_passes;
}
// bytes [s0, s1) of a chunk's miss array that are set (0 or 1 each; the array starts 16-byte aligned)
__device__ __forceinline__ uint32_t count_miss_bytes(const uint8_t *kmiss, uint64_t s0, uint64_t s1) {
    const uint4 *kw = reinterpret_cast<const uint4 *>(kmiss);
    uint32_t cntb = 0;
    for (uint64_t q = s0 >> 4; q <= (s1 - 1) >> 4; ++q) {
        const uint4 v4 = kw[q];
        const uint32_t vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            const uint64_t lo = (q << 4) + 4u * i;
            uint32_t m = ~0u;
            if (lo < s0) m = (s0 - lo >= 4) ? 0u : m << (8u * (uint32_t)(s0 - lo));
            if (lo + 4 > s1) m = (lo >= s1) ? 0u : m & (~0u >> (8u * (uint32_t)(lo + 4 - s1)));
            cntb += (uint32_t)__popc(vv[i] & m);
        }
    }
    return cntb;
}
__global__ void __launch_bounds__(256) k_prefix_open(FinalizeArgs a, const uint4 *meta, const uint32_t *n_pairs_ptr, uint32_t *fail) {
    const uint32_t n_pairs = *n_pairs_ptr;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n_pairs; e += gridDim.x * blockDim.x) {
        uint32_t kbase;
        if (!pair_in_chunk(a.fail, a.pair_chunk, a.chunks, a.launched_passes, e, kbase)) continue;
        const uint32_t n = meta[e].z - a.hp.k + 1, pn = prefix_kmers(a.threshold, n, a.hp.k);
        if (pn == n) continue;  // everything was binned
        const uint64_t s0 = a.pair_kpos[e];
        const uint32_t missing = count_miss_bytes(a.kmiss + (uint64_t)kbase * 16u, s0, s0 + pn);
        if (pn - missing < need_kmers(a.threshold, n)) atomicOr(&fail[e], 2u);  // undecided: the record kernel counts all k-mers
    }
}
void launch_prefix_open(const FinalizeArgs &a, const uint4 *meta, const uint32_t *n_pairs_ptr, uint32_t *fail, hipStream_t st) {
    hipLaunchKernelGGL(k_prefix_open, dim3(2048), dim3(256), 0, st, a, meta, n_pairs_ptr, fail);
}
// One block per column (block mode with k-mer entries: per block of 8 leaves — its pairs lie in up to 256 buckets, one per
// candidate mask; the chunks of a column are numbered contiguously and never mix two masks).
__global__ void __launch_bounds__(256) k_tile_plan(TileArgs a) {
    __shared__ unsigned long long s_sum;
    __shared__ uint32_t s_chunk0, s_w[4];
    const uint32_t c = blockIdx.x, cl = a.chunk_log2;
    const uint32_t n_sub = a.blocks ? 256u : 1u;  // buckets of this column
    // chunks of the column = sum over its buckets
    uint32_t mine = 0;
    if (threadIdx.x < n_sub) {
        const uint32_t bk = c * n_sub + threadIdx.x;
        mine = (a.bucket_off[(bk + 1) << a.sub_log2] - a.bucket_off[bk << a.sub_log2] + (1u << cl) - 1u) >> cl;
    }
    uint32_t tot = mine;
    for (int d = 32; d > 0; d >>= 1) tot += __shfl_down(tot, d);
    if (lane_id() == 0) s_w[threadIdx.x >> 6] = tot;
    __syncthreads();
    const uint32_t n_col = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    if (threadIdx.x == 0) {
        s_chunk0 = n_col ? atomicAdd(a.n_chunks, n_col) : 0u;
        a.leaf_chunk0[c] = n_col ? s_chunk0 : 0xffffffffu;
    }
    __syncthreads();
    uint32_t chunk = s_chunk0;
    for (uint32_t sb = 0; sb < n_sub; ++sb) {
        const uint32_t bk = c * n_sub + sb;
        const uint32_t lo = a.bucket_off[bk << a.sub_log2], hi = a.bucket_off[(bk + 1) << a.sub_log2];
        const uint32_t n_ch = (hi - lo + (1u << cl) - 1u) >> cl;
        for (uint32_t ci = 0; ci < n_ch; ++ci, ++chunk) {
            const uint32_t first = lo + (ci << cl), n = (hi - first) < (1u << cl) ? hi - first : (1u << cl);
            if (threadIdx.x == 0) s_sum = 0;
            __syncthreads();
            unsigned long long kmers = 0;
            for (uint32_t e = first + threadIdx.x; e < first + n; e += blockDim.x) {
                const uint32_t n_e = a.meta[e].z - a.hp.k + 1;
                kmers += (a.counts && !a.blocks) ? prefix_kmers(a.threshold, n_e, a.hp.k) : n_e;  // (block mode bins every k-mer)
                a.pair_chunk[e] = chunk < a.max_chunks ? chunk : 0xffffffffu;
            }
            for (int d = 32; d > 0; d >>= 1) kmers += __shfl_down(kmers, d);
            if (lane_id() == 0 && kmers) atomicAdd(&s_sum, kmers);
            __syncthreads();
            if (threadIdx.x == 0 && chunk < a.max_chunks) {
                // mean probes per tile + 8 standard deviations of a Poisson count + slack (anything beyond falls back) + the
                // padding of k_tile_bin's runs (up to 3 entries per round: a round brings >= 96 probes per tile, or 32 pairs)
                unsigned long long mean = (s_sum * a.hp.num_hashes + a.n_tiles - 1) / a.n_tiles;
                // (block mode, small buckets: correlated reads, see k_tile_bin — three times the deviation)
                const unsigned long long dev = (unsigned long long)(((a.blocks && n < 512u) ? 24.0f : 8.0f) * sqrtf((float)mean)) + 64;
                uint32_t cap = (uint32_t)((mean + dev + (mean >> 5) + 3 * ((n + 31) / 32 + 2) + 31) & ~31ull);
                unsigned long long need = (unsigned long long)cap * a.n_tiles;
                // (where the chunk's buckets go — pass and position in the reused buffer — is decided by k_tile_assign)
                const bool fits = need <= a.entry_cap;
                ChunkDesc dsc;
                dsc.row = a.blocks ? (a.meta[first].w & 0xffffffu) : a.meta[first].w;
                dsc.first = first;
                dsc.n = n;
                dsc.cap = fits ? cap : 0u;   // a single chunk larger than the whole buffer: its pairs take the fallback
                dsc.base = 0;
                dsc.leaf = c;
                dsc.pass = 0;
                dsc.kbase = 0;   // (placed by k_tile_assign)
                // one miss byte per k-mer of the chunk (block mode: eight, one per leaf of the block), in 16-byte units
                dsc.kwords = a.counts ? (uint32_t)(((a.blocks ? s_sum * 8 : s_sum) + 15) >> 4) : 0u;
                dsc.mask = a.blocks ? a.meta[first].w >> 24 : 0u;  // (the mask all pairs of the chunk share)
                dsc.pad_ = 0;
                if (a.counts && s_sum >= (1ull << 28)) dsc.cap = 0;
                a.chunks[chunk] = dsc;
            }
            __syncthreads();
        }
    }
}
// The bucket space is virtual: pass = position / entry_cap, place in the (reused) buffer = position % entry_cap.
// One wave packs the chunks in order: when everything fits one pass (the normal case) a chunk's place is the prefix sum of
// the needs before it (wave scans, 64 chunks per step); else chunk after chunk, a chunk that does not fit the rest of
// the current pass starting the next one, so that no reservation straddles two passes.  (Reserving with atomics from the
// plan blocks was tried twice: add-and-retry inflates the pass count without bound when a chunk nearly fills the buffer;
// compare-and-swap on one word from 1024 blocks cost 5 ms.)  entry_cursor receives the end of the packing, for the
// host's pass count.
__global__ void __launch_bounds__(64) k_tile_assign(TileArgs a) {
    const uint32_t lane = lane_id();
    const uint32_t n_chunks = *a.n_chunks < a.max_chunks ? *a.n_chunks : a.max_chunks;
    if (a.counts) {  // thresholds < 1: the chunks' k-mer miss bytes, packed in order; a chunk that finds no room takes the fallback
        unsigned long long run = 0;  // wave-uniform
        for (uint32_t c0 = 0; c0 < n_chunks; c0 += 64) {
            const uint32_t c = c0 + lane;
            const bool have = c < n_chunks && a.chunks[c].cap != 0;
            const unsigned long long need = have ? (unsigned long long)a.chunks[c].kwords : 0u;  // (16-byte units)
            unsigned long long incl = need;
            for (int dd = 1; dd < 64; dd <<= 1) {
                const unsigned long long o = __shfl_up(incl, dd);
                if ((int)lane >= dd) incl += o;
            }
            if (have) {
                if ((run + incl) * 16u <= a.kmiss_cap) a.chunks[c].kbase = (uint32_t)(run + incl - need);
                else a.chunks[c].cap = 0;
            }
            run += __shfl(incl, 63);
        }
        if (lane == 0) *a.kmiss_used = run * 16u < a.kmiss_cap ? run * 16u : a.kmiss_cap;  // (k_zero16 clears that many bytes)
    }
    unsigned long long total = 0;
    for (uint32_t c = lane; c < n_chunks; c += 64) total += (unsigned long long)a.chunks[c].cap * a.n_tiles;
    for (int dd = 32; dd > 0; dd >>= 1) total += __shfl_xor(total, dd);
    if (total <= a.entry_cap) {  // one pass
        unsigned long long run = 0;  // wave-uniform
        for (uint32_t c0 = 0; c0 < n_chunks; c0 += 64) {
            const uint32_t c = c0 + lane;
            const uint32_t cap = c < n_chunks ? a.chunks[c].cap : 0u;
            const unsigned long long need = (unsigned long long)cap * a.n_tiles;
            unsigned long long incl = need;
            for (int dd = 1; dd < 64; dd <<= 1) {
                const unsigned long long o = __shfl_up(incl, dd);
                if ((int)lane >= dd) incl += o;
            }
            if (cap) {
                a.chunks[c].base = run + incl - need;
                a.chunks[c].pass = 0;
            }
            run += __shfl(incl, 63);
        }
        if (lane == 0) *a.entry_cursor = run;
        return;
    }
    unsigned long long off = 0;  // wave-uniform
    uint32_t pass = 0;
    for (uint32_t c0 = 0; c0 < n_chunks; c0 += 64) {
        const uint32_t c = c0 + lane;
        const uint32_t cap = c < n_chunks ? a.chunks[c].cap : 0u;
        const unsigned long long need = (unsigned long long)cap * a.n_tiles;
        unsigned long long my_off = 0;
        uint32_t my_pass = 0;
        for (int i = 0; i < 64; ++i) {
            const unsigned long long ni = ((unsigned long long)bcast_u32((uint32_t)(need >> 32), i) << 32) | bcast_u32((uint32_t)need, i);
            if (ni == 0) continue;
            if (off + ni > a.entry_cap) {
                ++pass;
                off = 0;
            }
            if ((int)lane == i) {
                my_off = off;
                my_pass = pass;
            }
            off += ni;
        }
        if (cap) {
            a.chunks[c].base = my_off;
            a.chunks[c].pass = my_pass;
        }
    }
    if (lane == 0) *a.entry_cursor = (unsigned long long)pass * a.entry_cap + off;
}
__global__ void __launch_bounds__(256) k_zero16(uint4 *p, const unsigned long long *n_bytes) {
    const uint64_t n = (*n_bytes + 15) >> 4;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = make_uint4(0, 0, 0, 0);
}
void launch_tile_plan(const TileArgs &a, hipStream_t st) {
    if (!a.n_leaves) return;
    hipLaunchKernelGGL(k_tile_plan, dim3(a.n_leaves), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_tile_assign, dim3(1), dim3(64), 0, st, a);
    if (a.counts) hipLaunchKernelGGL(k_zero16, dim3(2048), dim3(256), 0, st, reinterpret_cast<uint4 *>(a.kmiss), a.kmiss_used);
    // (block mode: the per-k-mer "no leaf of the block has it" bytes, one per 8 miss bytes — cleared whole, they are small)
    if (a.counts && a.blocks) (void)hipMemsetAsync(a.kall, 0, (a.kmiss_cap >> 3) + 16, st);
}

// k_tile_bin: a block takes whole chunks from a queue and bins every probe of the chunk's pairs by filter tile.
//   * Exclusive chunks: the block alone fills the chunk's (chunk, tile) buckets, so the fill marks live in LDS and no
//     global atomic is on the path (one per round and tile before: ~1.5 us of latency per round).
//   * Rounds by k-mer budget: a round takes as many consecutive pairs (<= 32) as bring at most KB k-mers — 0.75 BIN_CAP
//     probes per tile on average, and at most WPI windows of 64 k-mers per wave.  The k-mers of a round's pairs form ONE
//     flattened index space that the waves walk in windows (a 150 bp read has 130 k-mers: one pair per wave would leave a
//     third window with two busy lanes).  A read with more k-mers than the budget is binned over several rounds.
//   * Branch-free appends: a lane without a probe counts in a counter of its own; the windows of an iteration advance
//     together, so their LDS atomics are in flight together.
//   * Two barriers per round; every wave flushes its own tiles (bin -> bucket as one 16-byte-aligned run, padded with
//     ENTRY_PAD to a multiple of four entries) and clears their counters.
// Whatever cannot be binned — an LDS bin or a bucket overflowing (skewed probe distributions), no room for the buckets —
// flags the pair (bit 1 of its fail word, compact list) and k_verify_rec certifies exactly those pairs afterwards.
constexpr uint32_t ENTRY_PAD = 0xffffffffu;  // no real entry: local pair ids have CHUNK_PAIRS_LOG2 <= 12 bits
constexpr uint32_t ROUND_PAIRS = 32;         // lanes 0..31 of every wave hold the round's pairs
__device__ __forceinline__ void flag_fallback(const TileArgs &a, uint32_t e) {
    if (!(atomicOr(&a.fail[e], 2u) & 2u)) {
        const uint32_t pos = atomicAdd(a.n_flagged, 1u);
        if (pos < a.flag_cap) a.flag_list[pos] = e;
    }
}
// MODE 0: (pair, offset) entries, 128 KiB tiles.  MODE 1 (thresholds < 1): k-mer entries.  MODE 2 (block mode): entries
// [pair:10][byte offset:17] against the byte-per-index table of a block of 8 leaves (the candidate mask comes with the chunk).
// MODE 3 (block mode at thresholds < 1): k-mer entries against that table.
template <uint32_t BIN_WAVES, uint32_t BIN_CAP, uint32_t MODE>
__global__ void __launch_bounds__(BIN_WAVES * 64) k_tile_bin(TileArgs a) {
    constexpr bool COUNTS = MODE == 1 || MODE == 3, BLK = MODE == 2;  // (MODE 3: k-mer entries against block tables, no prefix)
    // COUNTS (thresholds < 1): an entry is [round tag:2][flattened k-mer of the round:11][offset in a 64 KiB tile:19]; the
    // tags are ORed in at flush time (position in the 16-byte vector -> two bits of the round's number), runs are padded
    // with copies of their last entry (testing a probe twice changes nothing), and per round the chunk position of its
    // first k-mer goes to round_k0, per pair the position of its first k-mer to pair_kpos.
    constexpr uint32_t TL = MODE == 1 ? TILE_LOG2_COUNTS : (MODE >= 2 ? TILE_LOG2_BLOCK : TILE_LOG2);
    constexpr uint32_t BIN_STRIDE = BIN_CAP + 4;  // rows stay 16-byte aligned; room for the padding of a full bin
    constexpr uint32_t WPI = 2;                    // windows per wave and iteration
    const uint32_t NT = (a.n_tiles + 63u) & ~63u;  // tiles, rounded up
    const uint32_t DUMMY = NT;                     // counters NT .. NT + 63: where lanes without a probe count
    extern __shared__ uint32_t s_dyn[];            // cnt[NT + 64], fill[NT], bins[n_tiles][BIN_STRIDE]
    __shared__ uint32_t s_chunk;
    uint32_t *cnt = s_dyn, *fillp = s_dyn + NT + 64, *bins = fillp + NT;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t d = (uint32_t)a.hp.nbits, dw = d - (uint32_t)a.hp.w64, k = a.hp.k, nh = a.hp.num_hashes;
    const uint32_t n_chunks = *a.n_chunks < a.max_chunks ? *a.n_chunks : a.max_chunks;
    // k-mers per round: the bins fill to 3/4 on average (shallow bins of many tiles — block mode — to 1/2: an overflow costs
    // the round's pairs the fallback, and 30 +- 5.5 entries must stay below 60)
    uint32_t KB = (uint32_t)(((uint64_t)(BIN_CAP < 128 ? BIN_CAP / 2 : BIN_CAP - BIN_CAP / 4) * a.n_tiles) / nh);
    if (KB > WPI * BIN_WAVES * WIN_KMERS) KB = WPI * BIN_WAVES * WIN_KMERS;
    if (COUNTS && KB > (1u << ROUND_KMERS_LOG2)) KB = 1u << ROUND_KMERS_LOG2;
    if (KB == 0) KB = 1;
    for (uint32_t t = threadIdx.x; t < NT + 64; t += blockDim.x) cnt[t] = 0;
    while (true) {
        __syncthreads();  // everybody is done with the previous chunk
        if (threadIdx.x == 0) s_chunk = atomicAdd(a.bin_queue, 1u);
        for (uint32_t t = threadIdx.x; t < NT; t += blockDim.x) fillp[t] = 0;
        __syncthreads();
        const uint32_t c = s_chunk;
        if (c >= n_chunks) break;
        const ChunkDesc dsc = a.chunks[c];
        if (dsc.cap == 0) {  // no bucket space for this chunk: its pairs take the fallback
            if (a.pass == 0)
                for (uint32_t i = threadIdx.x; i < dsc.n; i += blockDim.x) flag_fallback(a, dsc.first + i);
            continue;
        }
        if (dsc.pass != a.pass) continue;  // binned in another pass
        uint32_t *bucket0 = a.entries + dsc.base;
        // (block mode: the pairs of a small bucket — a rare candidate mask — are reads that start next to the same few
        // mutation sites and share most of their k-mers, so their probes come in multiples: a quarter of the k-mer budget keeps
        // the 60-entry bins from overflowing; families of 8: 6000 flagged pairs per step without it)
        const uint32_t KBc = (BLK && dsc.n < 512u && KB > 3u) ? KB / 4u : KB;
        // A round = pairs [p, p + P) of the chunk, K flattened k-mers (what is left of pair p after `koff`, then whole pairs).
        // Lanes 0..31 of every wave hold the candidates p + lane (every wave computes the same); `incl` = inclusive prefix
        // sums of their k-mer counts, qb = record index of flattened k-mer 0 of the lane's pair.
        struct Round {
            uint32_t p, P, K, incl, start;  // start: flattened index of the first k-mer of the lane's pair
            unsigned long long koff, qb;
            bool partial;
        };
        auto compose = [&](uint32_t p, unsigned long long koff, const uint4 &m) {
            Round r;
            r.p = p;
            r.koff = koff;
            const bool cand = lane < ROUND_PAIRS && p + lane < dsc.n;
            // (k-mer entries: only the prefix of the read's k-mers that decides nearly every pair, see prefix_kmers)
            const uint32_t n_all = cand ? m.z - k + 1u : 0u, n_bin = (MODE == 1 && cand) ? prefix_kmers(a.threshold, n_all, k) : n_all;
            const unsigned long long n64 = cand ? (unsigned long long)n_bin - (lane == 0 ? koff : 0ull) : 0ull;
            const uint32_t n_l = (uint32_t)(n64 > KBc ? KBc + 1u : n64);  // (more than the budget is all the same)
            uint32_t incl = n_l;
            for (uint32_t sft = 1; sft < ROUND_PAIRS; sft <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)incl, (int)sft);
                if (lane >= sft) incl += o;
            }
            r.incl = incl;
            r.start = incl - n_l;
            r.P = (uint32_t)__popcll(ballot64(cand && incl <= KBc));  // incl is monotone: a prefix of the candidates
            r.partial = r.P == 0 && p < dsc.n;  // the first pair alone exceeds the budget: the next KBc k-mers of it
            if (r.partial) {
                r.P = 1;
                r.K = KBc;
            } else r.K = r.P ? bcast_u32(incl, (int)r.P - 1) : 0u;
            r.qb = (((unsigned long long)m.y << 32) | m.x) + (lane == 0 ? koff : 0ull) - (unsigned long long)(incl - n_l);
            return r;
        };
        auto load_meta = [&](uint32_t p) {
            return (lane < ROUND_PAIRS && p + lane < dsc.n) ? a.meta[dsc.first + p + lane] : make_uint4(0, 0, 0, 0);
        };
        // the records of a round's windows of this wave: WPI windows of 64 flattened k-mers, window u = wave + u * BIN_WAVES
        auto load_recs = [&](const Round &r, uint4 (&rec)[WPI], uint32_t (&local)[WPI], bool (&valid)[WPI]) {
#pragma unroll
            for (uint32_t u = 0; u < WPI; ++u) {
                const uint32_t f0 = (wave + u * BIN_WAVES) * WIN_KMERS, f = f0 + lane;
                valid[u] = f < r.K;
                // the pair of flattened k-mer f: j0 = the pair of the window's first k-mer, then the pair ends inside the window
                uint32_t j0 = (uint32_t)__popcll(ballot64(lane < r.P && r.incl <= f0));
                if (j0 >= r.P) j0 = r.P ? r.P - 1 : 0;  // (window past the end: no valid lane)
                uint32_t j = j0;
                unsigned long long qb = bcast_u64(r.qb, (int)j0);
                for (uint32_t t = j0; t + 1 < r.P; ++t) {
                    const uint32_t e = bcast_u32(r.incl, (int)t);
                    if (e > f0 + 63u) break;
                    const unsigned long long qn = bcast_u64(r.qb, (int)t + 1);
                    if (f >= e) {
                        j = t + 1;
                        qb = qn;
                    }
                }
                rec[u] = valid[u] ? a.recs[qb + f] : make_uint4(0, 0, 0, 0);  // (no k-mer: every index 0, see `put`)
                local[u] = (COUNTS ? f : r.p + j) << TL;
            }
        };
        // Software pipeline over the rounds: while round r is binned, the records of round r + 1 and the pair metadata of
        // round r + 2 are in flight (all waves of the block move in step, so a load issued where it is needed would expose
        // its whole latency every round: 3 of 10 ms).
        uint4 m_nxt = load_meta(0);
        Round cur = compose(0, 0, m_nxt);
        uint4 rec[WPI];
        uint32_t local[WPI];
        bool valid[WPI];
        load_recs(cur, rec, local, valid);
        {
            const uint32_t p1 = cur.partial ? cur.p : cur.p + cur.P;
            m_nxt = load_meta(p1);
        }
        uint32_t rho = 0, k0 = 0;  // COUNTS: number of this round; chunk position of its first k-mer
        while (cur.p < dsc.n) {
            if (COUNTS) {
                if (rho >= MAX_ROUNDS) {  // the tags cannot name more rounds: what is left of the chunk takes the fallback
                    for (uint32_t i = cur.p + threadIdx.x; i < dsc.n; i += blockDim.x) flag_fallback(a, dsc.first + i);
                    break;
                }
                if (threadIdx.x == 0) a.round_k0[(uint64_t)c * MAX_ROUNDS + rho] = k0;
                // (a pair binned over several rounds keeps the position of its first piece; the pieces are contiguous)
                if (wave == 0 && lane < cur.P && (lane > 0 || cur.koff == 0)) a.pair_kpos[dsc.first + cur.p + lane] = k0 + (cur.partial ? 0u : cur.start);
            }
            const uint32_t p1 = cur.partial ? cur.p : cur.p + cur.P;
            const Round nxt = compose(p1, cur.partial ? cur.koff + KBc : 0ull, m_nxt);
            uint4 rec_n[WPI];
            uint32_t local_n[WPI];
            bool valid_n[WPI];
            load_recs(nxt, rec_n, local_n, valid_n);
            m_nxt = load_meta(nxt.partial ? nxt.p : nxt.p + nxt.P);
            if (!(PFQ_DEBUG_BITS(a) & 2u)) {
                RecordIter1 rit[WPI];
                uint32_t cbase[WPI];  // counter index of tile 0 for this lane
#pragma unroll
                for (uint32_t u = 0; u < WPI; ++u) {
                    rit[u].init(rec[u], d, dw);
                    cbase[u] = valid[u] ? 0u : DUMMY + lane;
                }
                // Branch-free appends: a lane without a k-mer walks index 0 (a zero record), counts in a counter of its own
                // and writes nothing; the windows advance together, so that their LDS atomics are in flight together.  An
                // overflowing bin keeps overwriting its last slot and is detected at flush time (cnt > BIN_CAP).
                auto put = [&](const uint32_t (&ix)[WPI]) {
                    uint32_t tile[WPI], slot[WPI];
#pragma unroll
                    for (uint32_t u = 0; u < WPI; ++u) tile[u] = ix[u] >> TL;
#pragma unroll
                    for (uint32_t u = 0; u < WPI; ++u) slot[u] = atomicAdd(&cnt[cbase[u] + tile[u]], 1u);
#pragma unroll
                    for (uint32_t u = 0; u < WPI; ++u)
                        if (valid[u]) bins[tile[u] * BIN_STRIDE + min(slot[u], BIN_CAP - 1u)] = local[u] | (ix[u] & ((1u << TL) - 1u));
                    if (BLK) {  // (block mode: exactly the pairs whose probe found its bin full take the fallback — each costs 8 x 1300 line gathers there)
#pragma unroll
                        for (uint32_t u = 0; u < WPI; ++u)
                            if (valid[u] && slot[u] >= BIN_CAP && !(PFQ_DEBUG_BITS(a) & 8u)) flag_fallback(a, dsc.first + (local[u] >> TL));
                    }
                };
                uint32_t ix[WPI];
#pragma unroll
                for (uint32_t u = 0; u < WPI; ++u) ix[u] = rit[u].i0;
                put(ix);
                if (nh > 1) {
#pragma unroll
                    for (uint32_t u = 0; u < WPI; ++u) ix[u] = rit[u].g;
                    put(ix);
                }
                if (nh > 2) {
#pragma unroll
                    for (uint32_t u = 0; u < WPI; ++u) ix[u] = rit[u].x;
                    put(ix);
                }
                for (uint32_t i = 3; i < nh; ++i) {
#pragma unroll
                    for (uint32_t u = 0; u < WPI; ++u) ix[u] = rit[u].step(d);
                    put(ix);
                }
            }
            lds_barrier();  // every probe of the round is in its bin
            // (the next round's records are taken over BEFORE the bucket stores are issued: waiting for them afterwards would
            // also wait for every store, which are younger in the same counter)
            const uint32_t flush_p = cur.p, flush_P = cur.P, flush_K = cur.K;
            cur = nxt;
#pragma unroll
            for (uint32_t u = 0; u < WPI; ++u) {
                rec[u] = rec_n[u];
                local[u] = local_n[u];
                valid[u] = valid_n[u];
            }
            // flush: every bin goes to its bucket as one run at the fill mark (kept in LDS: the block owns the chunk).  Sixteen
            // lanes per tile, four tiles per wave and pass, the 16-byte reads of a lane issued together: one LDS round trip
            // per batch instead of one per copy instruction (the block's waves flush in step, nothing else hides the latency).
            // (bins of <= 256 entries — many tiles — hold ~150 per round: eight lanes per tile, eight tiles per wave and pass)
            constexpr uint32_t LPT = BIN_CAP <= 256 ? 8 : 16, TPW = 64 / LPT, STEP = LPT * 4;
            for (uint32_t t0 = wave * TPW; t0 < a.n_tiles; t0 += BIN_WAVES * TPW) {
                const uint32_t t = t0 + lane / LPT, sl = lane % LPT;
                const bool have = t < a.n_tiles;
                const uint32_t cn = have ? cnt[t] : 0u, pos = have ? fillp[t] : 0u;
                const uint32_t cc = cn < BIN_CAP ? cn : BIN_CAP, c4 = (cc + 3u) & ~3u;
                uint32_t *row = bins + t * BIN_STRIDE;
                if (sl < c4 - cc) row[cc + sl] = COUNTS ? row[cc - 1u] : ENTRY_PAD;  // (c4 > cc only when cc >= 1)
                __builtin_amdgcn_wave_barrier();
                // what fits is written (k_tile_test reads min(fill, cap) entries: every slot below cap must hold an entry
                // or padding); the pairs whose probes are dropped — bucket full — take the fallback
                const uint32_t room = pos < dsc.cap ? dsc.cap - pos : 0u, wr = c4 < room ? c4 : room;
                uint32_t *dst = bucket0 + (uint64_t)(have ? t : 0u) * dsc.cap + pos;
                constexpr uint32_t FB = BIN_CAP < 128 ? 2 : 5;  // 16-byte reads in flight per lane: two batches cover a full bin (516 entries / 64)
                const uint32_t wr_eff = (PFQ_DEBUG_BITS(a) & 1u) ? 0u : wr;
                for (uint32_t i0 = sl * 4u; i0 < wr; i0 += STEP * FB) {
                    uint4 v[FB];
#pragma unroll
                    for (uint32_t u = 0; u < FB; ++u)  // (unconditional, clamped into the row: nothing waits for a branch)
                        v[u] = *reinterpret_cast<const uint4 *>(row + min(i0 + u * STEP, BIN_STRIDE - 4u));
#pragma unroll
                    for (uint32_t u = 0; u < FB; ++u)  // (keeps the compiler from sinking every read next to its store)
                        asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));
                    if (COUNTS) {  // the round's number, two bits per entry of a vector
#pragma unroll
                        for (uint32_t u = 0; u < FB; ++u) {
                            v[u].x |= (rho & 3u) << 30;
                            v[u].y |= ((rho >> 2) & 3u) << 30;
                            v[u].z |= ((rho >> 4) & 3u) << 30;
                            v[u].w |= ((rho >> 6) & 3u) << 30;
                        }
                    }
#pragma unroll
                    for (uint32_t u = 0; u < FB; ++u)
                        if (i0 + u * STEP < wr_eff) *reinterpret_cast<uint4 *>(dst + i0 + u * STEP) = v[u];
                }
                if (!COUNTS && !(PFQ_DEBUG_BITS(a) & 16u))
                    for (uint32_t i = wr + sl; i < cc; i += LPT) flag_fallback(a, dsc.first + (row[i] >> TL));
                if (!BLK && (cn > BIN_CAP || (COUNTS && wr < cc))) {  // the LDS bin (or, with k-mer entries, the bucket) overflowed: whose probes were lost is unknown
                    for (uint32_t i = sl; i < flush_P; i += LPT) flag_fallback(a, dsc.first + flush_p + i);
                }
                __builtin_amdgcn_wave_barrier();
                if (have && sl == 0 && cn) {
                    fillp[t] = pos + c4;
                    cnt[t] = 0;
                }
            }
            lds_barrier();  // bins and counters are free again
            k0 += flush_K;
            ++rho;
        }
        if (COUNTS && threadIdx.x == 0) a.n_rounds[c] = rho;
        for (uint32_t t = threadIdx.x; t < a.n_tiles; t += blockDim.x) a.gfill[(uint64_t)c * a.n_tiles + t] = fillp[t];
    }
}
// Function attributes are set once per device (the CLI runs replicas on several GPUs from one process).
static bool first_on_device(std::atomic<uint64_t> &done) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    return !(done.fetch_or(bit) & bit);
}
template <uint32_t W, uint32_t CAP>
static void launch_tile_bin_shape(const TileArgs &a, int blocks, size_t lds, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};  // (per instantiation; a bit per device: replicas on several GPUs in one process)
    if (first_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_tile_bin<W, CAP, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_tile_bin<W, CAP, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_tile_bin<W, CAP, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_tile_bin<W, CAP, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    }
    if (a.counts && a.blocks) hipLaunchKernelGGL((k_tile_bin<W, CAP, 3>), dim3(blocks), dim3(W * 64), lds, st, a);
    else if (a.counts) hipLaunchKernelGGL((k_tile_bin<W, CAP, 1>), dim3(blocks), dim3(W * 64), lds, st, a);
    else if (a.blocks) hipLaunchKernelGGL((k_tile_bin<W, CAP, 2>), dim3(blocks), dim3(W * 64), lds, st, a);
    else hipLaunchKernelGGL((k_tile_bin<W, CAP, 0>), dim3(blocks), dim3(W * 64), lds, st, a);
}
void launch_tile_bin(const TileArgs &a, int blocks, hipStream_t st) {
    const size_t nt = (a.n_tiles + 63u) & ~63u;
    auto lds_of = [&](size_t cap) { return (2 * nt + 64 + (size_t)a.n_tiles * (cap + 4)) * 4; };
    if (a.bin_shape == 3 && lds_of(256) <= 74 * 1024) {  // experiment: two 8-wave blocks per CU
        launch_tile_bin_shape<8, 256>(a, blocks, lds_of(256), st);
    } else if (lds_of(2048) <= 148 * 1024 && a.bin_shape == 0) {  // few tiles (small filters): deeper bins, longer rounds
        launch_tile_bin_shape<16, 2048>(a, (blocks + 1) / 2, lds_of(2048), st);
    } else if (lds_of(1024) <= 148 * 1024 && a.bin_shape == 0) {
        launch_tile_bin_shape<16, 1024>(a, (blocks + 1) / 2, lds_of(1024), st);
    } else if (lds_of(512) <= 148 * 1024 && a.bin_shape == 0) {
        launch_tile_bin_shape<16, 512>(a, (blocks + 1) / 2, lds_of(512), st);
    } else if (lds_of(256) <= 148 * 1024 && a.bin_shape != 1) {
        launch_tile_bin_shape<16, 256>(a, (blocks + 1) / 2, lds_of(256), st);
    } else if (lds_of(128) <= 148 * 1024) {
        launch_tile_bin_shape<8, 128>(a, blocks, lds_of(128), st);
    } else {  // several hundred tiles (block mode of large filters): shallow bins
        launch_tile_bin_shape<16, 60>(a, (blocks + 1) / 2, lds_of(60), st);
    }
}

constexpr uint32_t TEST_LOADS = 4;   // 16-byte entry loads in flight per thread (8 measured slower: 6.1 vs 5.9 ms)
constexpr uint32_t TEST_GROUP = 32;  // chunks of a leaf whose buckets are streamed as one sequence
// A task = (column, tile): the block loads that tile of the column's filter into LDS (128 KiB; 64 KiB with k-mer entries)
// and tests every probe binned for it.  The tasks are software-pipelined: while task i's entries stream, the tile of task
// i + 1 is already on its way into registers and its bucket descriptors into the other half of the descriptor arrays (a tile
// takes >= 5 us at the per-CU load rate; trees with few pairs per leaf — subtree shards, thousands of leaves — spend most of
// a task there).
// COUNTS (thresholds < 1): an entry names a k-mer of its round (see TILE_LOG2_COUNTS); a probed bit that is 0 sets the
// k-mer's byte in its chunk's miss array — position round_k0[chunk][round] + k-mer of the round, the table rows of the
// group's chunks staged in LDS (and, like the tile, fetched while the previous task streams).
// MODE 2 (block mode): the tile is 2^17 BYTES of the block's table, bit j of a byte = that Bloom bit of leaf 8b + j; an entry
// carries the mask of the pair's candidate leaves; a candidate whose bit is 0 gets its failure byte set (plain stores: one
// byte per (pair, leaf)), reported once per block and task through the LDS bitmap.
template <uint32_t MODE>
__global__ void __launch_bounds__(1024) k_tile_test(TileArgs a) {
    // MODE 3 (block mode at thresholds < 1): k-mer entries against the block's byte table; the candidate mask comes with the
    // chunk (buckets are keyed by (block, mask)); a candidate whose bit is 0 gets the k-mer's miss byte of that leaf set.
    constexpr bool COUNTS = MODE == 1 || MODE == 3, BLK = MODE == 2, BLKC = MODE == 3;
    constexpr uint32_t TL = MODE == 1 ? TILE_LOG2_COUNTS : (MODE >= 2 ? TILE_LOG2_BLOCK : TILE_LOG2);
    constexpr uint32_t TILE_BYTES = MODE >= 2 ? (1u << TL) : (1u << (TL - 3));
    constexpr uint32_t TV = TILE_BYTES / 8192u;        // 8-byte loads per thread and tile
    constexpr uint32_t CL = CHUNK_PAIRS_LOG2;  // pairs per chunk (LDS bitmap of reported failures)
    // chunks per unit (block mode: chunks of 128 pairs, a unit should still bring >= 16 k entries; MODE 3: the round tables
    // of 16 chunks are what fits beside the 128 KiB tile)
    constexpr uint32_t TG = (BLK || BLKC) ? 16 : TEST_GROUP;
    constexpr uint32_t RK = COUNTS ? MAX_ROUNDS : 1u;
    extern __shared__ uint32_t s_tile[];  // 2^TL bits
    __shared__ uint32_t s_pref[2][TG + 1], s_first[2][TG], s_misc[2][2], s_kbase[2][TG], s_mask[2][TG];
    const uint32_t n_cols = a.n_leaves;
    auto col_of = [&](uint32_t li) { return li; };
    // failures this block already reported: a bit per pair of the group (MODE 0), per (pair, leaf of the block) (MODE 2)
    __shared__ uint32_t s_failed[COUNTS ? 1 : (BLK ? (TG << (CL - 2)) : (TG << (CL - 5)))];
    constexpr uint32_t N_FAILED = COUNTS ? 0 : (BLK ? (TG << (CL - 2)) : (TG << (CL - 5)));
    __shared__ uint32_t s_rk0[COUNTS ? TG : 1][RK];                               // round_k0 rows of the group's chunks
    __shared__ unsigned long long s_base[2][TG];
    const uint32_t tile_words = TILE_BYTES / 4u;
    const uint64_t n_words32 = a.n_words * 2;  // (block mode: n_words = bytes / 8 of one block's table)
    const uint64_t n_tasks = (uint64_t)n_cols * a.n_tiles;
    const uint32_t n_chunks = *a.n_chunks < a.max_chunks ? *a.n_chunks : a.max_chunks;
    // next task of this block at or after `task` whose column has pairs at all (block-uniform)
    auto valid_task = [&](uint64_t task) {
        while (task < n_tasks && a.leaf_chunk0[col_of((uint32_t)(task / a.n_tiles))] == 0xffffffffu) task += gridDim.x;
        return task;
    };
    // The buckets (this tile's) of up to 32 chunks of the column starting at chunk g0, described in LDS: they are streamed
    // as ONE sequence of entries (a chunk's bucket alone, ~20 k entries, would leave the block's 16 k-entry steps half idle).
    auto describe = [&](uint32_t buf, uint32_t leaf, uint32_t t, uint32_t g0) {  // threads 0..63
        const uint32_t i = threadIdx.x, c = g0 + i;
        ChunkDesc dsc{};
        const bool mine = i < TG && c < n_chunks && (dsc = a.chunks[c], dsc.leaf == leaf);
        const bool usable = mine && dsc.cap != 0 && dsc.pass == a.pass;
        uint32_t fill = usable ? a.gfill[(uint64_t)c * a.n_tiles + t] : 0u;
        if (fill > dsc.cap) fill = dsc.cap;
        uint32_t incl = fill;
        for (uint32_t sft = 1; sft < TG; sft <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, (int)sft);
            if (i >= sft) incl += o;
        }
        if (i < TG) {
            s_pref[buf][i + 1] = incl;
            s_first[buf][i] = dsc.first;
            s_kbase[buf][i] = dsc.kbase;
            s_mask[buf][i] = dsc.mask;
            s_base[buf][i] = dsc.base + (uint64_t)t * dsc.cap;
        }
        const uint64_t mm = ballot64(mine);
        if (i == 0) {
            s_pref[buf][0] = 0;
            s_misc[buf][0] = (uint32_t)__popcll(mm);  // chunks of the column in this group
        }
    };
    // the column's tile: words [t * tile_words, ...) of its filter row (zero beyond the filter's end); filter rows are only
    // 8-byte aligned: 8-byte loads, all of a thread in flight
    auto tile_loads = [&](uint32_t leaf, uint32_t t, uint2 (&v)[TV]) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.bits + (uint64_t)((BLK || BLKC) ? leaf : a.col_row[leaf]) * a.n_words);
        const uint64_t w0 = (uint64_t)t * tile_words;
#pragma unroll
        for (uint32_t u = 0; u < TV; ++u) {
            const uint32_t i = threadIdx.x * 2 + u * 2048u;
            v[u] = (w0 + i + 1 < n_words32) ? *reinterpret_cast<const uint2 *>(src + w0 + i) : make_uint2(0, 0);
        }
    };
    auto tile_store = [&](const uint2 (&v)[TV]) {
#pragma unroll
        for (uint32_t u = 0; u < TV; ++u) *reinterpret_cast<uint2 *>(s_tile + threadIdx.x * 2 + u * 2048u) = v[u];
        if (!COUNTS)
            for (uint32_t i = threadIdx.x; i < N_FAILED; i += blockDim.x) s_failed[i] = 0;
    };
    // COUNTS: the round_k0 rows of the 32 chunks from g0 on: 32 threads per chunk, 8 rounds per thread (rounds the chunk
    // did not have — and chunks of other columns — are never looked up)
    constexpr uint32_t RK_TPC = 1024u / TG, RK_V = TG / 16u;  // threads per chunk; 16-byte loads per thread (4 rounds each)
    auto rk_loads = [&](uint32_t g0, uint4 (&rk)[2]) {
        const uint32_t c = g0 + threadIdx.x / RK_TPC, r0 = (threadIdx.x % RK_TPC) * 4u * RK_V;
        rk[0] = rk[1] = make_uint4(0, 0, 0, 0);
        if (COUNTS && c < n_chunks && r0 < a.n_rounds[c]) {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.round_k0 + (uint64_t)c * MAX_ROUNDS + r0);
            rk[0] = src[0];
            if (RK_V > 1) rk[1] = src[1];
        }
    };
    auto rk_store = [&](const uint4 (&rk)[2]) {
        if (COUNTS) {
            static_assert(!COUNTS || TG == 32 || TG == 16, "1024 threads fetch 256 rounds of TG chunks in one or two 16-byte loads each");
            uint4 *dst = reinterpret_cast<uint4 *>(&s_rk0[threadIdx.x / RK_TPC][(threadIdx.x % RK_TPC) * 4u * RK_V]);
            dst[0] = rk[0];
            if (RK_V > 1) dst[1] = rk[1];
        }
    };
    // 16 entries in flight per thread as four 16-byte loads (buckets start on 128-byte boundaries, their fill marks are
    // multiples of four entries: a load never straddles two buckets; k_tile_bin pads its runs)
    auto stream = [&](uint32_t buf) {
        const uint32_t total = s_pref[buf][TG];
        uint32_t ci = 0;  // bucket of the thread's current position (positions only grow)
        for (uint32_t v0 = threadIdx.x * 4u; v0 < total; v0 += blockDim.x * 4u * TEST_LOADS) {
            uint4 en[TEST_LOADS];
            uint32_t first[TEST_LOADS], cidx[TEST_LOADS];
            bool have[TEST_LOADS];
#pragma unroll
            for (uint32_t u = 0; u < TEST_LOADS; ++u) {
                const uint32_t v = v0 + u * blockDim.x * 4u;
                en[u] = make_uint4(ENTRY_PAD, ENTRY_PAD, ENTRY_PAD, ENTRY_PAD);
                first[u] = 0;
                cidx[u] = 0;
                have[u] = v < total;
                if (have[u]) {
                    while (v >= s_pref[buf][ci + 1]) ++ci;
                    en[u] = *reinterpret_cast<const uint4 *>(a.entries + s_base[buf][ci] + (v - s_pref[buf][ci]));
                    first[u] = s_first[buf][ci];
                    cidx[u] = COUNTS ? ci : ci << CL;
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < TEST_LOADS; ++u) {
                const uint32_t ev[4] = {en[u].x, en[u].y, en[u].z, en[u].w};
                if (BLKC) {
                    if (!have[u]) continue;
                    const uint32_t mk = s_mask[buf][cidx[u]];
                    uint32_t bad[4], any = 0;
#pragma unroll
                    for (uint32_t c = 0; c < 4; ++c) {
                        bad[c] = mk & ~(uint32_t)reinterpret_cast<const uint8_t *>(s_tile)[ev[c] & ((1u << TL) - 1u)];
                        any |= bad[c];
                    }
                    if (any) {
                        const uint32_t rho = (ev[0] >> 30) | ((ev[1] >> 30) << 2) | ((ev[2] >> 30) << 4) | ((ev[3] >> 30) << 6);
                        const uint64_t k0 = (uint64_t)s_kbase[buf][cidx[u]] * 16u + (uint64_t)s_rk0[cidx[u]][rho] * 8u;
#pragma unroll
                        for (uint32_t c = 0; c < 4; ++c) {  // plain idempotent byte stores
                            if (!bad[c]) continue;
                            const uint64_t kq = (k0 >> 3) + ((ev[c] >> TL) & ((1u << ROUND_KMERS_LOG2) - 1u));  // the k-mer's index
                            // a k-mer over a sequencing error is in none of the block's leaves: ONE byte says so for all candidates
                            if (bad[c] == mk) a.kall[kq] = 1;
                            else
                                for (uint32_t m = bad[c]; m; m &= m - 1u) a.kmiss[kq * 8u + (uint32_t)__ffs((int)m) - 1u] = 1;  // a byte per (k-mer, leaf)
                        }
                    }
                } else if (COUNTS) {
                    if (!have[u]) continue;
                    uint32_t bad = 0;
#pragma unroll
                    for (uint32_t c = 0; c < 4; ++c) {
                        const uint32_t off = ev[c] & ((1u << TL) - 1u);
                        bad |= (((s_tile[off >> 5] >> (off & 31u)) & 1u) ^ 1u) << c;
                    }
                    if (bad) {
                        const uint32_t rho = (ev[0] >> 30) | ((ev[1] >> 30) << 2) | ((ev[2] >> 30) << 4) | ((ev[3] >> 30) << 6);
                        const uint32_t kb = s_rk0[cidx[u]][rho], wb = s_kbase[buf][cidx[u]];
#pragma unroll
                        for (uint32_t c = 0; c < 4; ++c) {
                            if (!((bad >> c) & 1u)) continue;
                            // a plain byte store: idempotent, no read-modify-write (device-scope atomics are performed at
                            // the memory side on this part — 2.7 G/s when a k-mer over a sequencing error fails in ten tiles)
                            a.kmiss[(uint64_t)wb * 16u + kb + ((ev[c] >> TL) & ((1u << ROUND_KMERS_LOG2) - 1u))] = 1;
                        }
                    }
                } else if (BLK) {
                    if (!have[u]) continue;
#pragma unroll
                    for (uint32_t c = 0; c < 4; ++c) {
                        const uint32_t off = ev[c] & ((1u << TL) - 1u);
                        const uint32_t bad = ev[c] != ENTRY_PAD ? s_mask[buf][cidx[u] >> CL] & ~(uint32_t)reinterpret_cast<const uint8_t *>(s_tile)[off] : 0u;
                        if (bad) {  // candidates whose bit is 0; only what this block has not reported yet goes to memory
                            // (a candidate over a mutation fails hundreds of probes: a plain read sorts out the repeats)
                            const uint32_t lp = (ev[c] >> TL) & ((1u << CL) - 1u), fb = cidx[u] + lp, sh = 8u * (fb & 3u);
                            if (bad & ~(s_failed[fb >> 2] >> sh)) {
                                const uint32_t fresh = bad & ~(atomicOr(&s_failed[fb >> 2], bad << sh) >> sh);
                                for (uint32_t m = fresh; m; m &= m - 1u) a.failb[((uint64_t)(first[u] + lp) << 3) + (uint32_t)__ffs((int)m) - 1u] = 1;
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (uint32_t c = 0; c < 4; ++c) {
                        const uint32_t off = ev[c] & ((1u << TL) - 1u);
                        if (ev[c] != ENTRY_PAD && !((s_tile[off >> 5] >> (off & 31u)) & 1u)) {
                            // a probed bit is 0: the pair fails.  A failing pair usually has hundreds of such probes (every
                            // k-mer over a sequencing error): only the first one this block sees goes to memory.
                            const uint32_t lp = ev[c] >> TL, fb = cidx[u] + lp;
                            if (!(atomicOr(&s_failed[fb >> 5], 1u << (fb & 31u)) & (1u << (fb & 31u)))) atomicOr(&a.fail[first[u] + lp], 1u);
                        }
                    }
                }
            }
        }
    };

    // The unit of the pipeline is (task, group of TG chunks of the task's column): while a unit's entries stream, the
    // next unit's bucket descriptors (and round tables) are fetched into the other half of the descriptor arrays, and — when
    // the next unit belongs to another task — its tile into registers.  (A column of a family workload has hundreds of
    // chunks: fetching every group's descriptors where they are needed cost 9 of 16 ms.)
    uint64_t task = valid_task(blockIdx.x);
    if (task >= n_tasks) return;
    uint32_t buf = 0, g0 = a.leaf_chunk0[col_of((uint32_t)(task / a.n_tiles))];
    {
        const uint32_t leaf = col_of((uint32_t)(task / a.n_tiles)), t = (uint32_t)(task % a.n_tiles);
        uint2 v[TV];
        uint4 rk[2];
        tile_loads(leaf, t, v);
        rk_loads(g0, rk);
        if (threadIdx.x < 64) describe(0, leaf, t, g0);
        tile_store(v);
        rk_store(rk);
        __syncthreads();
    }
    while (true) {
        // the next unit: the column's next group (it may turn out empty), else the first group of the block's next task
        const bool more = s_misc[buf][0] == TG;  // (block-uniform)
        const uint64_t ntask = more ? task : valid_task(task + gridDim.x);
        const bool have_n = ntask < n_tasks;
        const uint32_t nleaf = have_n ? col_of((uint32_t)(ntask / a.n_tiles)) : 0u, nt = have_n ? (uint32_t)(ntask % a.n_tiles) : 0u;
        const uint32_t ng0 = more ? g0 + TG : (have_n ? a.leaf_chunk0[nleaf] : 0u);
        uint2 vn[TV];
        uint4 rkn[2];
        if (have_n && !more) tile_loads(nleaf, nt, vn);  // in flight while this unit's entries stream
        if (have_n) rk_loads(ng0, rkn);
        if (have_n && threadIdx.x < 64) describe(buf ^ 1u, nleaf, nt, ng0);
        stream(buf);
        __syncthreads();  // everybody is done with this unit's tile, descriptors and tables; the next unit's descriptors are written
        if (!have_n) break;
        if (!more) tile_store(vn);  // (clears the bitmap of reported failures)
        else if (!COUNTS)
            for (uint32_t i = threadIdx.x; i < N_FAILED; i += blockDim.x) s_failed[i] = 0;
        rk_store(rkn);
        __syncthreads();
        task = ntask;
        g0 = ng0;
        buf ^= 1u;
    }
}
void launch_tile_test(const TileArgs &a, int blocks, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};
    if (first_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_tile_test<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        // (static LDS of the k-mer-entry build: the 32 KiB of round_k0 rows; static + dynamic must stay within the CU's 160 KiB)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_tile_test<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 1 << (TILE_LOG2_COUNTS - 3));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_tile_test<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 1 << TILE_LOG2_BLOCK);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_tile_test<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 1 << TILE_LOG2_BLOCK);
    }
    const dim3 g((blocks + 1) / 2), b(1024);
    if (a.counts && a.blocks) hipLaunchKernelGGL(k_tile_test<3>, g, b, (size_t)(1u << TILE_LOG2_BLOCK), st, a);
    else if (a.counts) hipLaunchKernelGGL(k_tile_test<1>, g, b, (size_t)(1u << (TILE_LOG2_COUNTS - 3)), st, a);
    else if (a.blocks) hipLaunchKernelGGL(k_tile_test<2>, g, b, (size_t)(1u << TILE_LOG2_BLOCK), st, a);
    else hipLaunchKernelGGL(k_tile_test<0>, g, b, (size_t)(1u << (TILE_LOG2 - 3)), st, a);
}

// Thresholds < 1 after the tile passes: the compact list of the pairs whose fail word is non-zero (a probe found 0, or the
// pair could not be binned).  Every block takes one contiguous range of the sorted pairs and appends its open pairs in
// order, so the list stays leaf-ordered in runs of thousands of pairs (the slices stay L2-resident for k_verify_rec).
__global__ void __launch_bounds__(256) k_collect_open(const uint32_t *fail, const uint32_t *n_pairs_ptr, uint32_t *list,
                                                      uint32_t cap, unsigned int *n_out) {
    __shared__ uint32_t s_w[4], s_base;
    const uint32_t n = *n_pairs_ptr, lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t per = ((n + gridDim.x - 1) / gridDim.x + 255u) & ~255u;
    const uint32_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    if (lo >= n) return;  // (block-uniform)
    uint32_t c = 0;
    for (uint32_t e = lo + threadIdx.x; e < hi; e += 256) c += fail[e] != 0u;
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
    if (lane == 0) s_w[wave] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        s_base = tot ? atomicAdd(n_out, tot) : 0u;
    }
    __syncthreads();
    uint32_t run = s_base;
    if (s_w[0] + s_w[1] + s_w[2] + s_w[3] == 0) return;  // (block-uniform)
    __syncthreads();
    for (uint32_t e0 = lo; e0 < hi; e0 += 256) {
        const uint32_t e = e0 + threadIdx.x;
        const bool open = e < hi && fail[e] != 0u;
        const uint64_t b = ballot64(open);
        if (lane == 0) s_w[wave] = (uint32_t)__popcll(b);
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (uint32_t w = 0; w < 4; ++w) {
            before += w < wave ? s_w[w] : 0u;
            all += s_w[w];
        }
        const uint32_t pos = run + before + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
        if (open && pos < cap) list[pos] = e;
        run += all;
        __syncthreads();
    }
}
void launch_collect_open(const uint32_t *fail, const uint32_t *n_pairs_ptr, uint32_t *list, uint32_t cap, unsigned int *n_out, hipStream_t st) {
    hipLaunchKernelGGL(k_collect_open, dim3(512), dim3(256), 0, st, fail, n_pairs_ptr, list, cap, n_out);
}
void launch_verify(const VerifyArgs &a, int blocks, int threads, hipStream_t st) {
    if (a.recs) hipLaunchKernelGGL(k_verify_rec, dim3(blocks), dim3(threads), 0, st, a);
    else hipLaunchKernelGGL(k_verify, dim3(blocks), dim3(256), 0, st, a);
}

// Block tables (block mode): T[b][i] = byte whose bit j is bit i of the filter of leaf column 8b + j.  A wave takes one
// 64-bit word of the 8 filters at a time: every lane reads the 8 words (one transaction each), lane i assembles byte i.
__global__ void __launch_bounds__(256) k_block_tables(const uint64_t *bits, uint64_t n_words, const uint32_t *col_row, uint32_t n_leaves, uint8_t *T) {
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6, b = blockIdx.y;
    uint8_t *dst = T + (uint64_t)b * n_words * 64;
    const uint64_t *src[8];
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) src[j] = (8u * b + j < n_leaves) ? bits + (uint64_t)col_row[8u * b + j] * n_words : nullptr;
    for (uint64_t w = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + wave; w < n_words; w += (uint64_t)gridDim.x * WAVES_PER_BLOCK) {
        uint32_t byte = 0;
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            const uint64_t v = src[j] ? src[j][w] : 0ull;
            byte |= (uint32_t)((v >> lane) & 1ull) << j;
        }
        dst[w * 64 + lane] = (uint8_t)byte;
    }
}
void launch_block_tables(const uint64_t *bits, uint64_t n_words, const uint32_t *d_col_row, uint32_t n_leaves, uint8_t *T, hipStream_t st) {
    if (!n_leaves) return;
    uint64_t bx = (n_words + 3) / 4;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(k_block_tables, dim3((uint32_t)bx, (n_leaves + 7) / 8), dim3(256), 0, st, bits, n_words, d_col_row, n_leaves, T);
}

// Block mode: the pairs k_tile_bin could not bin (fail bit 1: a bin or a bucket overflowing, no room for the buckets) are
// certified here, candidate leaf by candidate leaf, against the sliced matrix (exact, slow: one line gather per probe).
__global__ void __launch_bounds__(256) k_block_fallback(QueryArgs a, const uint2 *sorted, const uint32_t *n_pairs_ptr, const uint32_t *fail, uint8_t *failb,
                                                        const uint32_t *pair_chunk, const ChunkDesc *chunks, uint32_t launched_passes,
                                                        const unsigned int *n_flagged, const uint32_t *flag_list, uint32_t flag_cap) {
    __shared__ BlockLds lds;
    fill_complement(lds.comp);
    __syncthreads();
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6, n_pairs = *n_pairs_ptr;
    const uint64_t gw = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + wave, nw = (uint64_t)gridDim.x * WAVES_PER_BLOCK;
    auto certify = [&](uint64_t e, uint32_t r, uint32_t y, uint32_t j) {  // candidate leaf j of pair e = (r, y), by the whole wave
        const uint64_t o0 = a.off[r], L = a.off[r + 1] - o0;
        ReadCtx rc;
        rc.read = a.seq + o0;
        rc.n = L - a.hp.k + 1;  // deferred reads have k-mers and 1 <= need <= n
        rc.need = need_kmers(a.threshold, rc.n);
        rc.maxmiss = rc.n - rc.need;
        const bool pass = verify_column(lds, wave, a, rc, ((y & 0xffffffu) << BLOCK_LEAVES_LOG2) + j);
        if (!pass && lane == 0) failb[(e << 3) + j] = 1;
    };
    // The flagged pairs come in clusters (the pairs of one round of k_tile_bin are neighbours): from the compact list, a wave
    // per (pair, candidate leaf), so that a cluster does not wait for one wave.
    const uint32_t nf = n_flagged ? *n_flagged : 0u;
    const bool listed = flag_list && nf <= flag_cap;
    if (listed) {
        for (uint64_t i = gw; i < (uint64_t)nf * 8u; i += nw) {
            const uint32_t e = flag_list[i >> 3], j = (uint32_t)(i & 7u);
            const uint2 p = sorted[e];
            if ((p.y >> (24 + j)) & 1u) certify(e, p.x, p.y, j);
        }
    }
    // Pairs that were not binned at all: no tile passes (chunks == nullptr), or their chunk's pass was not launched (more than
    // 256 passes); and the flagged ones when their list overflowed.
    for (uint64_t e0 = gw * 64u; e0 < n_pairs; e0 += nw * 64u) {
        const uint64_t e = e0 + lane;
        bool todo_e = e < n_pairs && (fail[e] & 2u) && !listed;
        if (e < n_pairs && !(fail[e] & 2u)) {
            if (!chunks) todo_e = true;
            else {
                const uint32_t c = pair_chunk[e];
                todo_e = c == 0xffffffffu || chunks[c].pass >= launched_passes;
            }
        }
        uint2 p = make_uint2(0, 0);
        if (todo_e) p = sorted[e];
        uint64_t todo = ballot64(todo_e);
        while (todo) {
            const int src = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const uint32_t r = bcast_u32(p.x, src), y = bcast_u32(p.y, src);
            for (uint32_t j = 0; j < 8; ++j)
                if ((y >> (24 + j)) & 1u) certify(e0 + (uint32_t)src, r, y, j);
        }
    }
}
void launch_block_fallback(const QueryArgs &a, const uint2 *sorted, const uint32_t *n_pairs_ptr, const uint32_t *fail, uint8_t *failb,
                           const uint32_t *pair_chunk, const ChunkDesc *chunks, uint32_t launched_passes, const unsigned int *n_flagged,
                           const uint32_t *flag_list, uint32_t flag_cap, hipStream_t st) {
    hipLaunchKernelGGL(k_block_fallback, dim3(1024), dim3(256), 0, st, a, sorted, n_pairs_ptr, fail, failb, pair_chunk, chunks, launched_passes,
                       n_flagged, flag_list, flag_cap);
}

// Block mode with k-mer entries: the miss bytes of the binned pairs decide their candidates.  A wave per pair, lanes =
// k-mers: the 8 miss bytes of 64 k-mers are one coalesced read, a ballot per leaf counts them; a candidate with fewer than
// `need` contained k-mers gets its failure byte (what k_finalize reads).  Pairs that were not binned are left to
// k_block_fallback.
__global__ void __launch_bounds__(256) k_block_count(FinalizeArgs a, const uint32_t *n_pairs_ptr, uint8_t *failb) {
    const uint32_t lane = lane_id(), n_pairs = *n_pairs_ptr;
    const uint64_t gw = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6), nw = (uint64_t)gridDim.x * WAVES_PER_BLOCK;
    for (uint64_t e = gw; e < n_pairs; e += nw) {
        uint32_t kb16 = 0;
        if (!pair_in_chunk(a.fail, a.pair_chunk, a.chunks, a.launched_passes, (uint32_t)e, kb16)) continue;  // (wave-uniform)
        const uint2 p = a.sorted[e];
        const uint64_t o0 = a.off[p.x], L = a.off[p.x + 1] - o0, n = L - a.hp.k + 1, need = need_kmers(a.threshold, n);
        const uint64_t q0 = (uint64_t)kb16 * 2u + a.pair_kpos[e];  // first k-mer of the pair
        const uint2 *km = reinterpret_cast<const uint2 *>(a.kmiss) + q0;
        uint32_t miss[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint64_t k0 = 0; k0 < n; k0 += 64) {
            const uint64_t kk = k0 + lane;
            uint2 w = make_uint2(0, 0);
            if (kk < n) {
                w = km[kk];
                if (a.kall[q0 + kk]) w = make_uint2(0x01010101u, 0x01010101u);  // the k-mer is in no candidate leaf of the block
            }
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                miss[j] += (uint32_t)__popcll(ballot64((w.x >> (8u * j)) & 0xffu));
                miss[4 + j] += (uint32_t)__popcll(ballot64((w.y >> (8u * j)) & 0xffu));
            }
        }
        if (lane < 8 && ((p.y >> (24 + lane)) & 1u)) {
            uint32_t mj = 0;
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) mj = lane == j ? miss[j] : mj;
            if (n - mj < need) failb[(e << 3) + lane] = 1;
        }
    }
}
void launch_block_count(const FinalizeArgs &a, const uint32_t *n_pairs_ptr, uint8_t *failb, hipStream_t st) {
    hipLaunchKernelGGL(k_block_count, dim3(2048), dim3(256), 0, st, a, n_pairs_ptr, failb);
}

__global__ void __launch_bounds__(256) k_block_guards(QueryArgs a, const uint2 *sorted, const uint32_t *n_pairs_ptr, uint8_t *failb) {
    __shared__ BlockLds lds;
    fill_complement(lds.comp);
    __syncthreads();
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6, n_pairs = *n_pairs_ptr;
    const uint64_t gw = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + wave, nw = (uint64_t)gridDim.x * WAVES_PER_BLOCK;
    for (uint64_t e0 = gw * 64u; e0 < n_pairs; e0 += nw * 64u) {
        const uint64_t e = e0 + lane;
        uint2 p = make_uint2(0, 0);
        uint32_t todo_mask = 0;  // candidates of my pair that are still standing and have guards
        if (e < n_pairs) {
            p = sorted[e];
            const uint2 fb = *reinterpret_cast<const uint2 *>(failb + (e << 3));
            const uint32_t c0 = (p.y & 0xffffffu) << BLOCK_LEAVES_LOG2;
            for (uint32_t j = 0; j < 8; ++j) {
                const uint32_t f = ((j < 4 ? fb.x : fb.y) >> (8u * (j & 3u))) & 0xffu;
                if (((p.y >> (24 + j)) & 1u) && !f && a.guard_off[c0 + j + 1] > a.guard_off[c0 + j]) todo_mask |= 1u << j;
            }
        }
        uint64_t todo = ballot64(todo_mask != 0);
        while (todo) {
            const int src = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const uint32_t r = bcast_u32(p.x, src), y = bcast_u32(p.y, src), tm = bcast_u32(todo_mask, src);
            const uint64_t o0 = a.off[r], L = a.off[r + 1] - o0;
            ReadCtx rc;
            rc.read = a.seq + o0;
            rc.n = L - a.hp.k + 1;
            rc.need = need_kmers(a.threshold, rc.n);
            rc.maxmiss = rc.n - rc.need;
            for (uint32_t j = 0; j < 8; ++j) {
                if (!((tm >> j) & 1u)) continue;
                const uint32_t col = ((y & 0xffffffu) << BLOCK_LEAVES_LOG2) + j;
                bool pass = true;
                for (uint32_t g = a.guard_off[col]; pass && g < a.guard_off[col + 1]; ++g) pass = verify_column(lds, wave, a, rc, a.guard_col[g]);
                if (!pass && lane == 0) failb[((e0 + (uint32_t)src) << 3) + j] = 1;
            }
        }
    }
}
void launch_block_guards(const QueryArgs &a, const uint2 *sorted, const uint32_t *n_pairs_ptr, uint8_t *failb, hipStream_t st) {
    hipLaunchKernelGGL(k_block_guards, dim3(1024), dim3(256), 0, st, a, sorted, n_pairs_ptr, failb);
}

// One block per leaf bucket: pairs that no slice failed are hits (mapped_reads += |pass|, query.rs:143).
__global__ void __launch_bounds__(256) k_finalize(FinalizeArgs a) {
    __shared__ unsigned long long s_cnt, s_bytes;
    if (a.failb) {  // block mode: a candidate leaf of a pair hits unless one of its failure bytes is set
        // The pairs are sorted by (block, mask) and nearly all of a block's pairs sit in one bucket: the blocks of this kernel
        // take slices of 2048 consecutive pairs instead of buckets; a slice spans few blocks of leaves, counted in LDS.
        __shared__ unsigned int s_leaf[64][8];
        const uint32_t n_pairs = a.bucket_off[a.c1 << a.sub_log2];
        for (uint32_t s0 = blockIdx.x * 2048u; s0 < n_pairs; s0 += gridDim.x * 2048u) {
            for (uint32_t i = threadIdx.x; i < 64 * 8; i += blockDim.x) (&s_leaf[0][0])[i] = 0;
            if (threadIdx.x == 0) s_bytes = 0;
            __syncthreads();
            const uint32_t blk0 = a.sorted[s0].y & 0xffffffu, s1 = s0 + 2048u < n_pairs ? s0 + 2048u : n_pairs;
            unsigned long long bytes = 0, hits = 0;
            for (uint32_t e = s0 + threadIdx.x; e < s1; e += blockDim.x) {
                const uint2 p = a.sorted[e];
                const uint32_t blk = p.y & 0xffffffu;
                const uint2 fb = *reinterpret_cast<const uint2 *>(a.failb + ((uint64_t)e << 3));
                uint32_t failed = 0;
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    failed |= ((fb.x >> (8u * j)) & 0xffu) ? 1u << j : 0u;
                    failed |= ((fb.y >> (8u * j)) & 0xffu) ? 16u << j : 0u;
                }
                const uint64_t o0 = a.off[p.x], L = a.off[p.x + 1] - o0, n = L - a.hp.k + 1, need = need_kmers(a.threshold, n);
                for (uint32_t m = (p.y >> 24) & ~failed; m; m &= m - 1u) {
                    const uint32_t j = (uint32_t)__ffs((int)m) - 1u;
                    if (blk - blk0 < 64u) atomicAdd(&s_leaf[blk - blk0][j], 1u);
                    else atomicAdd(&a.counts[(blk << BLOCK_LEAVES_LOG2) + j], 1ull);
                    ++hits;
                    bytes += need * a.hp.num_hashes * 32ull;
                    if (a.hit_pairs) {
                        unsigned long long pos = atomicAdd(a.hit_cursor, 1ull);
                        if (pos < a.hit_cap) a.hit_pairs[pos] = make_uint2(p.x, (blk << BLOCK_LEAVES_LOG2) + j);
                    }
                }
            }
            for (int d = 32; d > 0; d >>= 1) {
                bytes += __shfl_down(bytes, d);
                hits += __shfl_down(hits, d);
            }
            if (lane_id() == 0 && hits) {
                atomicAdd(&s_bytes, bytes);
                atomicAdd(&a.stats[ST_HITS], hits);
            }
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < 64 * 8; i += blockDim.x) {
                const unsigned int v = (&s_leaf[0][0])[i];
                if (v) atomicAdd(&a.counts[((blk0 + (i >> 3)) << BLOCK_LEAVES_LOG2) + (i & 7u)], (unsigned long long)v);
            }
            if (threadIdx.x == 0 && s_bytes) atomicAdd(&a.stats[ST_ALG_BYTES], s_bytes);
            __syncthreads();
        }
        return;
    }
    for (uint32_t c = a.c0 + blockIdx.x; c < a.c1; c += gridDim.x) {
        if (threadIdx.x == 0) { s_cnt = 0; s_bytes = 0; }
        __syncthreads();
        unsigned long long cnt = 0, bytes = 0;
        uint32_t dirty = 0;
        for (uint32_t e = a.bucket_off[c << a.sub_log2] + threadIdx.x; e < a.bucket_off[(c + 1) << a.sub_log2]; e += blockDim.x) {
            uint2 p = a.sorted[e];
            const uint64_t o0 = a.off[p.x], L = a.off[p.x + 1] - o0, n = L - a.hp.k + 1;
            uint64_t need = n;
            bool pass;
            if (a.miss_words) {  // thresholds < 1: contained k-mers = n - missing ones; query_passes (query.rs:38-49)
                uint64_t missing = 0;
                // where the pair's miss information is: in its chunk's miss bytes when a launched LDS-tile pass binned it and
                // nothing flagged it (bit 1 of the fail word), else in its own words, written by k_verify_rec
                uint32_t kbase = 0;
                const bool in_chunk = a.kmiss && pair_in_chunk(a.fail, a.pair_chunk, a.chunks, a.launched_passes, e, kbase);
                uint64_t n_seen = n;  // k-mers the count is about
                if (in_chunk) {  // the binned prefix of the read's k-mers (k_prefix_open saw to it that it decides the pair)
                    n_seen = prefix_kmers(a.threshold, (uint32_t)n, a.hp.k);
                    const uint64_t s0 = a.pair_kpos[e];
                    missing = count_miss_bytes(a.kmiss + (uint64_t)kbase * 16u, s0, s0 + n_seen);
                } else {
                    const unsigned long long *mw = a.miss_words + a.miss_pos[e];
                    const uint32_t nw = (uint32_t)((n + 63) >> 6);
#pragma unroll 4
                    for (uint32_t w = 0; w < nw; ++w) missing += (uint64_t)__popcll(mw[w]);
                }
                need = need_kmers(a.threshold, n);
                pass = n_seen - missing >= need;
                dirty += missing != 0;
            } else pass = !(a.fail[e] & 1u);
            if (a.guards) {  // a guard that does not pass takes its leaf pair with it (query.rs:119-141: children are only
                             // visited with the reads that passed the parent)
                if (!pass) a.gfail[a.owner_sorted[e]] = 1u;
                continue;
            }
            if (a.owner_sorted && a.gfail[a.owner_sorted[e]]) pass = false;
            if (pass) {
                ++cnt;
                bytes += need * a.hp.num_hashes * 32ull;
                if (a.hit_pairs) {
                    unsigned long long pos = atomicAdd(a.hit_cursor, 1ull);
                    if (pos < a.hit_cap) a.hit_pairs[pos] = p;
                }
            }
        }
        for (int d = 32; d > 0; d >>= 1) {
            cnt += __shfl_down(cnt, d);
            bytes += __shfl_down(bytes, d);
            dirty += __shfl_down(dirty, d);
        }
        if (lane_id() == 0 && cnt) { atomicAdd(&s_cnt, cnt); atomicAdd(&s_bytes, bytes); }
        if (lane_id() == 0 && dirty && a.n_dirty) atomicAdd(a.n_dirty, (unsigned long long)dirty);
        __syncthreads();
        if (threadIdx.x == 0 && s_cnt) {
            atomicAdd(&a.counts[c], s_cnt);
            atomicAdd(&a.stats[ST_HITS], s_cnt);
            atomicAdd(&a.stats[ST_ALG_BYTES], s_bytes);
        }
        __syncthreads();
    }
}
void launch_finalize(const FinalizeArgs &a, hipStream_t st) {
    uint32_t blocks = a.c1 - a.c0 < 2048u ? a.c1 - a.c0 : 2048u;
    if (a.failb) blocks = 2048u;  // (block mode: slices of the sorted pairs, not buckets)
    if (a.c1 <= a.c0) return;
    hipLaunchKernelGGL(k_finalize, dim3(blocks), dim3(256), 0, st, a);
}

// ---- database construction --------------------------------------------------------------------------------------------
// Leaf filters: insert every canonical k-mer of genome g (what init_leaf_node does serially, bloom_tree.rs:154-168;
// bits as ASMS::insert sets them, bloom_filter.rs:291-307).
__global__ void __launch_bounds__(256) k_insert(HashParams hp, const uint8_t *genomes, const uint64_t *goff,
                                                const uint32_t *leaf_row, uint64_t *bits, uint64_t n_words, uint64_t len1, uint32_t row1) {
    __shared__ BlockLds lds;
    fill_complement(lds.comp);
    __syncthreads();
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6, g = blockIdx.y;
    // (goff == nullptr: ONE genome of len1 bytes into filter row row1 — no argument arrays to upload)
    const uint64_t o0 = goff ? goff[g] : 0, L = goff ? goff[g + 1] - o0 : len1;
    const uint64_t n = (L >= hp.k) ? (L - hp.k + 1) : 0;
    unsigned long long *row = reinterpret_cast<unsigned long long *>(bits + (uint64_t)(goff ? leaf_row[g] : row1) * n_words);
    const uint64_t wid = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + wave, stride = (uint64_t)gridDim.x * WAVES_PER_BLOCK;
    for (uint64_t base = wid * WIN_KMERS; base < n; base += stride * WIN_KMERS) {
        uint32_t cnt = (uint32_t)((n - base) < WIN_KMERS ? (n - base) : WIN_KMERS);
        stage_window(lds, wave, genomes + o0, base, cnt, hp.k);
        bool valid = lane < cnt;
        uint64_t h1, h2;
        kmer_hashes(lds, wave, lane, cnt, valid, hp, h1, h2);
        ProbeIter it;
        it.init(h1, h2, hp);
        for_each_probe(it, hp, [&](uint32_t idx) {
            if (valid) atomicOr(&row[idx >> 6], 1ull << (idx & 63u));
        });
    }
}
void launch_insert(const HashParams &hp, const uint8_t *d_genomes, const uint64_t *d_goff, uint32_t n_genomes,
                   const uint32_t *d_leaf_row, uint64_t *bits, uint64_t n_words, hipStream_t st) {
    if (!n_genomes) return;
    hipLaunchKernelGGL(k_insert, dim3(64, n_genomes), dim3(256), 0, st, hp, d_genomes, d_goff, d_leaf_row, bits, n_words, 0ull, 0u);
}
void launch_insert_one(const HashParams &hp, const uint8_t *d_genome, uint64_t len, uint32_t row, uint64_t *bits, uint64_t n_words, hipStream_t st) {
    // (a window of 64 k-mers per wave where the genome is short enough: the atomics of a window wait for nothing but each other)
    const uint64_t windows = (len + WIN_KMERS - 1) / WIN_KMERS;
    const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(1024, (windows + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK));
    hipLaunchKernelGGL(k_insert, dim3(blocks, 1), dim3(256), 0, st, hp, d_genome, (const uint64_t *)nullptr, (const uint32_t *)nullptr, bits, n_words, len, row);
}

// ---- BloomTree::insert's greedy descent (add_to_tree, bloom_tree.rs:187-245) in ONE launch -------------------------------
// The whole walk happens on the device: at every two-child node the node absorbs the new leaf's filter (node_union) and the
// walk continues into the child at smaller Hamming distance to the new leaf (right only if strictly closer, :201); the leaf it
// reaches is replaced by a new internal node (left = the old leaf, right = the new leaf, filter = their union, :226-245).  The
// topology lives in device memory (TopoNode per node, the root's index), so consecutive insertions need no host round trip:
// the host only learns the shape when it next needs it.  The blocks of the grid meet at a barrier once per level (all of them
// are resident: one per CU); every block then adds up the per-block partial distances itself, so all take the same turn.
// A barrier that is not passed within a bounded number of polls sets the error word and the block leaves; the others then
// time out as well: the grid drains.
//
// The barrier carries the level's decision.  Everything the blocks exchange goes through read-modify-writes, which are
// performed at the memory side (the XCDs' L2s are not coherent with each other: a load was seen to stay on a stale line), and
// a line takes only ~90 of them per microsecond — so the words are spread over lines of their own, GREEDY_SYNC_STRIDE apart:
//   group g = block & 15: an arrival counter, a pair of distance accumulators (left, right) and a copy of the generation word;
//   one top counter for the groups.
// A block adds its distances to its group's accumulators and arrives at its group's counter; the last of a group arrives at
// the top counter; the last of all collects (and clears) the 16 pairs of accumulators with one exchange per lane, takes the
// turn and publishes [generation:31][turn:1] in the 16 generation words, where each group polls its own copy.  (Before:
// one counter and one generation word for 128 pollers, the partial sums of every block read back by every block: ~20 us per
// level, of which the filters' streaming is 9.)
__device__ __forceinline__ uint32_t *sync_word(unsigned long long *sync, uint32_t line) {
    return reinterpret_cast<uint32_t *>(sync + (uint64_t)line * (GREEDY_SYNC_STRIDE / 8u));
}
// Returns 0 / 1 = the turn (right iff strictly closer), -1 = timed out.  Called by every thread of the block; dl / dr are the
// block's sums, valid in thread 0.
__device__ __forceinline__ int grid_turn(unsigned long long *sync, unsigned int &my_gen, unsigned long long dl, unsigned long long dr, int *err) {
    __shared__ int s_turn;
    const uint32_t G = gridDim.x, g = blockIdx.x & 15u, n_groups = G < 16u ? G : 16u, in_group = (G - g + 15u) / 16u;
    uint32_t *top = sync_word(sync, 0), *gcount = sync_word(sync, 1u + g), *gen = sync_word(sync, 17u + g);
    __syncthreads();  // the block's stores of this level are issued
    if (threadIdx.x < 64) {  // wave 0
        const uint32_t lane = threadIdx.x;
        int last = 0;
        if (lane == 0) {
            unsigned long long *acc = sync + (uint64_t)(33u + g) * (GREEDY_SYNC_STRIDE / 8u);
            atomicAdd(&acc[0], dl);
            atomicAdd(&acc[1], dr);
            __threadfence();  // the block's node_union stores and its sums are performed before it arrives
            if (atomicAdd(gcount, 1u) == in_group - 1u) {
                // (the counter is clear BEFORE the group is reported: with small filters a block of the group is back here
                // within a microsecond of the generation's change, and a clearing still on its way would wipe its arrival)
                atomicExch(gcount, 0u);
                __threadfence();
                if (atomicAdd(top, 1u) == n_groups - 1u) {
                    atomicExch(top, 0u);
                    last = 1;
                }
            }
        }
        last = __builtin_amdgcn_readfirstlane(last);
        int turn = -1;
        if (last) {
            // every block has arrived: lanes 0..31 fetch and clear one accumulator each (left of group lane/2 for even lanes)
            unsigned long long v = 0;
            if (lane < 2u * n_groups) v = atomicExch(sync + (uint64_t)(33u + (lane >> 1)) * (GREEDY_SYNC_STRIDE / 8u) + (lane & 1u), 0ull);
            unsigned long long l = (lane & 1u) ? 0ull : v, r = (lane & 1u) ? v : 0ull;
            for (int d = 16; d > 0; d >>= 1) {
                l += __shfl_down(l, d);
                r += __shfl_down(r, d);
            }
            l = bcast_u64(l, 0);
            r = bcast_u64(r, 0);
            turn = r < l ? 1 : 0;  // `if right_distance < left_distance` (bloom_tree.rs:201): ties go left
            __threadfence();        // (the top counter's clearing, too, is performed before anybody can arrive again)
            if (lane < n_groups) atomicExch(sync_word(sync, 17u + lane), (((my_gen + 1u) & 0x7fffffffu) << 1) | (uint32_t)turn);
        } else if (lane == 0) {
            unsigned int polls = 0;
            while (true) {
                const uint32_t v = atomicAdd(gen, 0u);
                if ((v >> 1) != my_gen) {
                    turn = (int)(v & 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
                if (++polls > (1u << 21)) {  // (seconds: some block of the grid never arrived)
                    atomicExch(err, 2);
                    break;
                }
            }
            __threadfence();
        }
        if (lane == 0) s_turn = turn;
    }
    my_gen = (my_gen + 1u) & 0x7fffffffu;
    __syncthreads();
    return s_turn;
}
// The root is read from state[2 + (seq & 1)] and handed on in state[2 + (~seq & 1)], seq = the launch's number: a walk that ends
// at the root itself passes no barrier, and block 0 would otherwise install the new root while blocks that start late still
// read the old one (they would take the new internal node for the walk's first level and wait at a barrier nobody comes to).
__global__ void __launch_bounds__(1024) k_greedy_insert(uint64_t *bits, uint64_t n_words, TopoNode *topo, int *state /* root, err, root in/out */,
                                                        unsigned long long *sync, int leaf_node, int internal_node, uint32_t new_row, uint32_t int_row,
                                                        uint32_t seq) {
    __shared__ unsigned long long s_l[16], s_r[16];
    __shared__ unsigned int s_gen;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6, G = gridDim.x;
    // (the generation the block's copy holds when this launch starts; nobody changes it before all have read it: the first
    //  change needs every block's arrival)
    if (threadIdx.x == 0) s_gen = atomicAdd(sync_word(sync, 17u + (blockIdx.x & 15u)), 0u) >> 1;
    __syncthreads();
    unsigned int my_gen = s_gen;
    const uint64_t *nw = bits + (uint64_t)new_row * n_words;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (uint64_t)G * blockDim.x;
    int *root_out = &state[2u + (~seq & 1u)];
    int cur = state[2u + (seq & 1u)], parent = -1, side = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) topo[leaf_node] = TopoNode{-1, -1, new_row, 0u};
    if (cur < 0) {  // empty tree: the new leaf is the root
        if (threadIdx.x == 0 && blockIdx.x == 0) state[0] = *root_out = leaf_node;
        return;
    }
    const int root = cur;
    while (true) {
        const TopoNode c = topo[cur];
        if (c.left >= 0 && c.right >= 0) {
            uint64_t *crow = bits + (uint64_t)c.row * n_words;
            const uint64_t *l = bits + (uint64_t)topo[c.left].row * n_words, *r = bits + (uint64_t)topo[c.right].row * n_words;
            unsigned long long dl = 0, dr = 0;
            uint64_t i = tid;
            // (four words per thread and filter in flight: a thread sees ~9 of each filter's words)
            for (; i + 3u * nthreads < n_words; i += 4u * nthreads) {
                uint64_t v[4], cv[4], lv[4], rv[4];
#pragma unroll
                for (uint32_t u = 0; u < 4; ++u) {
                    v[u] = nw[i + u * nthreads];
                    cv[u] = crow[i + u * nthreads];
                    lv[u] = l[i + u * nthreads];
                    rv[u] = r[i + u * nthreads];
                }
#pragma unroll
                for (uint32_t u = 0; u < 4; ++u) {
                    crow[i + u * nthreads] = cv[u] | v[u];                       // node_union (bloom_tree.rs:194)
                    dl += (unsigned long long)__popcll(lv[u] ^ v[u]);          // distance (bloom_filter.rs:142-149)
                    dr += (unsigned long long)__popcll(rv[u] ^ v[u]);
                }
            }
            for (; i < n_words; i += nthreads) {
                const uint64_t v = nw[i];
                crow[i] |= v;
                dl += (unsigned long long)__popcll(l[i] ^ v);
                dr += (unsigned long long)__popcll(r[i] ^ v);
            }
            for (int d = 32; d > 0; d >>= 1) {
                dl += __shfl_down(dl, d);
                dr += __shfl_down(dr, d);
            }
            if (lane == 0) {
                s_l[wave] = dl;
                s_r[wave] = dr;
            }
            __syncthreads();
            dl = dr = 0;
            if (threadIdx.x == 0)
                for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) {
                    dl += s_l[w];
                    dr += s_r[w];
                }
            const int turn = grid_turn(sync, my_gen, dl, dr, &state[1]);
            if (turn < 0) return;
            parent = cur;
            side = turn;
            cur = side ? c.right : c.left;
        } else if (c.left < 0 && c.right < 0) {
            // the leaf is replaced by a new internal node over it and the new leaf (bloom_tree.rs:226-245)
            uint64_t *irow = bits + (uint64_t)int_row * n_words;
            const uint64_t *lrow = bits + (uint64_t)c.row * n_words;
            for (uint64_t i = tid; i < n_words; i += nthreads) irow[i] = lrow[i] | nw[i];
            if (threadIdx.x == 0 && blockIdx.x == 0) {
                topo[internal_node] = TopoNode{cur, leaf_node, int_row, 0u};
                state[0] = *root_out = parent < 0 ? internal_node : root;   // (state[0]: for the host)
                if (parent >= 0) {
                    if (side) topo[parent].right = internal_node;
                    else topo[parent].left = internal_node;
                }
            }
            return;
        } else {
            if (threadIdx.x == 0 && blockIdx.x == 0) atomicExch(&state[1], 1);  // "Node with only one child encountered" (bloom_tree.rs:209)
            return;
        }
    }
}
void launch_greedy_insert(uint64_t *bits, uint64_t n_words, TopoNode *topo, int *state, unsigned long long *sync, int leaf_node, int internal_node,
                          uint32_t new_row, uint32_t int_row, uint32_t seq, int blocks, hipStream_t st) {
    hipLaunchKernelGGL(k_greedy_insert, dim3(blocks), dim3(1024), 0, st, bits, n_words, topo, state, sync, leaf_node, internal_node, new_row, int_row, seq);
}

// Internal filter = OR of its children (node_union, bloom_tree.rs:238-239 / bloom_filter.rs:275-278).
__global__ void __launch_bounds__(256) k_union(uint64_t *bits, uint64_t n_words, const uint32_t *triples) {
    const uint32_t d = triples[3 * blockIdx.y], x = triples[3 * blockIdx.y + 1], y = triples[3 * blockIdx.y + 2];
    uint64_t *dst = bits + (uint64_t)d * n_words;
    const uint64_t *a = bits + (uint64_t)x * n_words, *b = bits + (uint64_t)(y == 0xffffffffu ? x : y) * n_words;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x)
        dst[i] = a[i] | b[i];
}
void launch_union(uint64_t *bits, uint64_t n_words, const uint32_t *d_triples, uint32_t n_triples, hipStream_t st) {
    if (!n_triples) return;
    uint32_t bx = (uint32_t)((n_words + 255) / 256);
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(k_union, dim3(bx, n_triples), dim3(256), 0, st, bits, n_words, d_triples);
}

// One level of the reference's greedy placement (add_to_tree, bloom_tree.rs:187-214) in a single pass over four
// filters: the internal node absorbs the new leaf (node_union), and the Hamming distances of the new leaf to both
// children (bloom_filter.rs:142-149: popcount of the XOR over the raw words) are written as per-block partial sums
// out[2b] (left), out[2b+1] (right).  A pure HBM-streaming kernel: 3 rows read + 1 row read-modify-written.
__global__ void __launch_bounds__(256) k_insert_step(uint64_t *bits, uint64_t n_words, uint32_t cur_row, uint32_t new_row,
                                                     uint32_t left_row, uint32_t right_row, unsigned long long *out) {
    uint64_t *cur = bits + (uint64_t)cur_row * n_words;
    const uint64_t *nw = bits + (uint64_t)new_row * n_words, *l = bits + (uint64_t)left_row * n_words,
                   *r = bits + (uint64_t)right_row * n_words;
    unsigned long long dl = 0, dr = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t v = nw[i];
        cur[i] |= v;
        dl += (unsigned long long)__popcll(l[i] ^ v);
        dr += (unsigned long long)__popcll(r[i] ^ v);
    }
    // per-block partial sums, no atomics: 16 K same-address atomics cost 190 us here (one counter sustains ~88/us),
    // twenty times the streaming itself; the host adds the INSERT_STEP_BLOCKS pairs
    __shared__ unsigned long long s_l[4], s_r[4];
    for (int d = 32; d > 0; d >>= 1) {
        dl += __shfl_down(dl, d);
        dr += __shfl_down(dr, d);
    }
    if (lane_id() == 0) {
        s_l[threadIdx.x >> 6] = dl;
        s_r[threadIdx.x >> 6] = dr;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = s_l[0] + s_l[1] + s_l[2] + s_l[3];
        out[2 * blockIdx.x + 1] = s_r[0] + s_r[1] + s_r[2] + s_r[3];
    }
}
void launch_insert_step(uint64_t *bits, uint64_t n_words, uint32_t cur_row, uint32_t new_row, uint32_t left_row,
                        uint32_t right_row, unsigned long long *d_out, hipStream_t st) {
    hipLaunchKernelGGL(k_insert_step, dim3(INSERT_STEP_BLOCKS), dim3(256), 0, st, bits, n_words, cur_row, new_row, left_row,
                       right_row, d_out);
}

// parent ⊇ child per edge (the invariant that makes the tree walk pure pruning).
__global__ void __launch_bounds__(256) k_superset(const uint64_t *bits, uint64_t n_words, const uint32_t *edges, uint32_t *fail) {
    const uint64_t *p = bits + (uint64_t)edges[2 * blockIdx.y] * n_words, *c = bits + (uint64_t)edges[2 * blockIdx.y + 1] * n_words;
    uint64_t bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x)
        bad |= c[i] & ~p[i];
    if (ballot64(bad != 0) && lane_id() == 0) atomicOr(&fail[blockIdx.y], 1u);
}
void launch_superset(const uint64_t *bits, uint64_t n_words, const uint32_t *d_edges, uint32_t n_edges, uint32_t *d_fail, hipStream_t st) {
    if (!n_edges) return;
    uint32_t bx = (uint32_t)((n_words + 255) / 256);
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(k_superset, dim3(bx, n_edges), dim3(256), 0, st, bits, n_words, d_edges, d_fail);
}

// node-major -> sliced: wave = 64 columns x 4 consecutive u64 words; `__ballot` transposes 64 columns x 1 bit.
__global__ void __launch_bounds__(256) k_transpose(const uint64_t *bits, uint64_t n_words, const uint32_t *col_row,
                                                   uint32_t n_cols, uint32_t *S, uint32_t rw, uint64_t group_stride, uint32_t group_log2) {
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6, cg = blockIdx.y;
    S += (uint64_t)(cg >> (group_log2 - 6u)) * group_stride;  // 2^(group_log2 - 6) x 64 columns per group of the sliced matrix
    const uint32_t cgl = cg & ((1u << (group_log2 - 6u)) - 1u);
    const uint32_t col = cg * 64u + lane;
    const bool has = col < n_cols;
    const uint64_t *src = bits + (uint64_t)(has ? col_row[col] : 0u) * n_words;
    const uint64_t n_groups = (n_words + 3) / 4;
    for (uint64_t grp = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + wave; grp < n_groups; grp += (uint64_t)gridDim.x * WAVES_PER_BLOCK) {
        uint64_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint64_t wi = grp * 4 + j;
            v[j] = (has && wi < n_words) ? src[wi] : 0ull;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint64_t wi = grp * 4 + j;
            if (wi >= n_words) break;
            uint64_t keep = 0;
            for (uint32_t b = 0; b < 64; ++b) {
                uint64_t m = ballot64((v[j] >> b) & 1ull);
                if (lane == b) keep = m;
            }
            uint32_t *dst = S + (wi * 64 + lane) * (uint64_t)rw + cgl * 2u;
            if (cgl * 2u < rw) dst[0] = (uint32_t)keep;
            if (cgl * 2u + 1u < rw) dst[1] = (uint32_t)(keep >> 32);
        }
    }
}
__global__ void __launch_bounds__(256) k_counts_op(unsigned long long *dst, const unsigned long long *a, const unsigned long long *b, uint32_t n, int sub) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = sub ? a[i] - b[i] : a[i] + b[i];
}
void launch_counts_op(unsigned long long *dst, const unsigned long long *a, const unsigned long long *b, uint32_t n, bool subtract, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_counts_op, dim3((n + 255) / 256), dim3(256), 0, st, dst, a, b, n, subtract ? 1 : 0);
}
void launch_transpose(const uint64_t *bits, uint64_t n_words, const uint32_t *d_col_row, uint32_t n_cols, uint32_t *S,
                      uint32_t rw, uint64_t group_stride, uint32_t group_log2, hipStream_t st) {
    if (!n_cols) return;
    uint32_t groups = (n_cols + 63) / 64;
    uint64_t bx = (n_words + 15) / 16;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(k_transpose, dim3((uint32_t)bx, groups), dim3(256), 0, st, bits, n_words, d_col_row, n_cols, S, rw, group_stride, group_log2);
}

// Set bits per filter row (how full the coarse level's filters are decides how many probes its screens look at).
__global__ void __launch_bounds__(256) k_row_popcount(const uint64_t *bits, uint64_t n_words, const uint32_t *rows, unsigned long long *out) {
    __shared__ unsigned long long s_w[4];
    const uint64_t *src = bits + (uint64_t)rows[blockIdx.x] * n_words;
    unsigned long long c = 0;
    for (uint64_t i = threadIdx.x; i < n_words; i += blockDim.x) c += (unsigned long long)__popcll(src[i]);
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
    if (lane_id() == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
void launch_row_popcount(const uint64_t *bits, uint64_t n_words, const uint32_t *d_rows, uint32_t n_rows, unsigned long long *d_out, hipStream_t st) {
    if (n_rows) hipLaunchKernelGGL(k_row_popcount, dim3(n_rows), dim3(256), 0, st, bits, n_words, d_rows, d_out);
}

// ---- per-read hit lists (PFQ_WANT_HITS): CSR read -> leaves built on the device ----------------------------------------
// The kernels append (read, leaf) hit pairs in whatever order their waves finish; the seam of query.rs:146-154 wants, per
// read, its leaves.  Count per read, exclusive scan, scatter, sort each read's few leaves (reads that pass every node list
// every leaf): all of it at device bandwidth instead of host loops over millions of reads.
__global__ void __launch_bounds__(256) k_hit_count(const uint2 *pairs, uint64_t n_pairs, uint32_t *cnt) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pairs; i += (uint64_t)gridDim.x * blockDim.x) atomicAdd(&cnt[pairs[i].x], 1u);
}
__global__ void __launch_bounds__(256) k_hit_allhit(const uint8_t *allhit, uint64_t n_reads, uint32_t *cnt, uint32_t n_leaves) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (uint64_t)gridDim.x * blockDim.x)
        if (allhit[r]) cnt[r] = n_leaves;
}
constexpr uint32_t SCAN_ITEMS = 4096;  // elements per block of 1024 threads
__global__ void __launch_bounds__(1024) k_scan_sums(const uint32_t *cnt, uint64_t n, unsigned long long *sums) {
    __shared__ unsigned long long s_w[16];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_ITEMS;
    unsigned long long v = 0;
    for (uint32_t j = 0; j < 4; ++j) {
        const uint64_t i = base + threadIdx.x * 4u + j;
        if (i < n) v += cnt[i];
    }
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    if (lane_id() == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < 16; ++w) t += s_w[w];
        sums[blockIdx.x] = t;
    }
}
__global__ void __launch_bounds__(1024) k_scan_top(unsigned long long *sums, uint32_t n_blocks) {  // one block: exclusive scan in place
    __shared__ unsigned long long s_w[16];
    __shared__ unsigned long long s_run;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 1024) {
        const uint32_t i = b0 + threadIdx.x;
        const unsigned long long x = i < n_blocks ? sums[i] : 0ull;
        unsigned long long incl = x;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long o = __shfl_up(incl, d);
            if ((int)lane_id() >= d) incl += o;
        }
        if (lane_id() == 63) s_w[threadIdx.x >> 6] = incl;
        __syncthreads();
        unsigned long long before = s_run;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += s_w[w];
        if (i < n_blocks) sums[i] = before + incl - x;
        __syncthreads();
        if (threadIdx.x == 1023) s_run = before + incl;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(1024) k_scan_apply(const uint32_t *cnt, uint64_t n, const unsigned long long *sums, unsigned long long *off) {
    __shared__ unsigned long long s_w[16];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_ITEMS + threadIdx.x * 4u;
    uint32_t c[4];
    unsigned long long mine = 0;
    for (uint32_t j = 0; j < 4; ++j) {
        c[j] = base + j < n ? cnt[base + j] : 0u;
        mine += c[j];
    }
    unsigned long long incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(incl, d);
        if ((int)lane_id() >= d) incl += o;
    }
    if (lane_id() == 63) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned long long run = sums[blockIdx.x] + incl - mine;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) run += s_w[w];
    for (uint32_t j = 0; j < 4; ++j) {
        if (base + j < n) off[base + j] = run;
        run += c[j];
        if (base + j + 1 == n) off[n] = run;  // the total
    }
}
__global__ void __launch_bounds__(256) k_hit_scatter(const uint2 *pairs, uint64_t n_pairs, const unsigned long long *off, uint32_t *cnt, uint32_t *leaves) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pairs; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint2 p = pairs[i];
        leaves[off[p.x] + (atomicSub(&cnt[p.x], 1u) - 1u)] = p.y;  // (filled from the end: the order is fixed by the sort below)
    }
}
__global__ void __launch_bounds__(256) k_hit_sort(const unsigned long long *off, uint32_t *leaves, const uint8_t *allhit, uint64_t n_reads) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long o0 = off[r], o1 = off[r + 1];
        if (allhit[r]) {  // passes every node: every leaf, in order
            for (unsigned long long i = o0; i < o1; ++i) leaves[i] = (uint32_t)(i - o0);
            continue;
        }
        for (unsigned long long i = o0 + 1; i < o1; ++i) {  // ascending within a read (a handful of leaves: insertion sort)
            const uint32_t v = leaves[i];
            unsigned long long j = i;
            while (j > o0 && leaves[j - 1] > v) {
                leaves[j] = leaves[j - 1];
                --j;
            }
            leaves[j] = v;
        }
    }
}
void launch_hits_csr(const uint2 *d_pairs, uint64_t n_pairs, const uint8_t *d_allhit, uint64_t n_reads, uint32_t n_leaves, bool any_allhit,
                     uint32_t *d_cnt, unsigned long long *d_sums, unsigned long long *d_off, hipStream_t st) {
    // d_cnt: [n_reads] zeroed by the caller; d_sums: [ceil(n_reads / SCAN_ITEMS) + 1]; d_off: [n_reads + 1]
    const uint32_t n_blocks = (uint32_t)((n_reads + SCAN_ITEMS - 1) / SCAN_ITEMS);
    if (n_pairs) hipLaunchKernelGGL(k_hit_count, dim3(2048), dim3(256), 0, st, d_pairs, n_pairs, d_cnt);
    if (any_allhit) hipLaunchKernelGGL(k_hit_allhit, dim3(2048), dim3(256), 0, st, d_allhit, n_reads, d_cnt, n_leaves);
    hipLaunchKernelGGL(k_scan_sums, dim3(n_blocks), dim3(1024), 0, st, d_cnt, n_reads, d_sums);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, st, d_sums, n_blocks);
    hipLaunchKernelGGL(k_scan_apply, dim3(n_blocks), dim3(1024), 0, st, d_cnt, n_reads, d_sums, d_off);
}
void launch_hits_fill(const uint2 *d_pairs, uint64_t n_pairs, const uint8_t *d_allhit, uint64_t n_reads, const unsigned long long *d_off,
                      uint32_t *d_cnt, uint32_t *d_leaves, hipStream_t st) {
    if (n_pairs) hipLaunchKernelGGL(k_hit_scatter, dim3(2048), dim3(256), 0, st, d_pairs, n_pairs, d_off, d_cnt, d_leaves);
    hipLaunchKernelGGL(k_hit_sort, dim3(4096), dim3(256), 0, st, d_off, d_leaves, d_allhit, n_reads);
}

// ---- test / bench helpers ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_debug_indices(HashParams hp, const uint8_t *seq, uint64_t len, uint64_t *out) {
    __shared__ BlockLds lds;
    fill_complement(lds.comp);
    __syncthreads();
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint64_t n = (len >= hp.k) ? (len - hp.k + 1) : 0;
    const uint64_t wid = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + wave, stride = (uint64_t)gridDim.x * WAVES_PER_BLOCK;
    for (uint64_t base = wid * WIN_KMERS; base < n; base += stride * WIN_KMERS) {
        uint32_t cnt = (uint32_t)((n - base) < WIN_KMERS ? (n - base) : WIN_KMERS);
        stage_window(lds, wave, seq, base, cnt, hp.k);
        bool valid = lane < cnt;
        uint64_t h1, h2;
        kmer_hashes(lds, wave, lane, cnt, valid, hp, h1, h2);
        ProbeIter it;
        it.init(h1, h2, hp);
        uint32_t i = 0;
        for_each_probe(it, hp, [&](uint32_t idx) {
            if (valid) out[(base + lane) * hp.num_hashes + i] = idx;
            ++i;
        });
    }
}
void launch_debug_indices(const HashParams &hp, const uint8_t *d_seq, uint64_t len, uint64_t *d_out, hipStream_t st) {
    hipLaunchKernelGGL(k_debug_indices, dim3(64), dim3(256), 0, st, hp, d_seq, len, d_out);
}

__device__ __forceinline__ uint8_t acgt(uint32_t v) { return (uint8_t)((0x54474341u >> (8u * (v & 3u))) & 0xffu); }  // "ACGT"

__global__ void __launch_bounds__(256) k_synth_genomes(uint8_t *out, uint64_t n_genomes, uint64_t len, uint64_t seed_base) {
    const uint64_t words = (len + 31) / 32, total = n_genomes * words;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t g = t / words, w = t % words, v = rnd(seed_base + g, w);
        for (uint32_t j = 0; j < 32 && w * 32 + j < len; ++j) out[g * len + w * 32 + j] = acgt((uint32_t)(v >> (2 * j)));
    }
}
void launch_synth_genomes(uint8_t *d_out, uint64_t n_genomes, uint64_t genome_len, uint64_t seed_base, hipStream_t st) {
    hipLaunchKernelGGL(k_synth_genomes, dim3(2048), dim3(256), 0, st, d_out, n_genomes, genome_len, seed_base);
}

__device__ __forceinline__ uint8_t comp_acgt(uint8_t b) { return b == 'A' ? 'T' : b == 'C' ? 'G' : b == 'G' ? 'C' : b == 'T' ? 'A' : b; }

__global__ void __launch_bounds__(256) k_synth_reads(uint8_t *out, uint64_t first, uint64_t n_reads, uint64_t read_len,
                                                     const uint8_t *genomes, uint64_t genome_len, uint64_t n_genomes, uint64_t seed) {
    // one wave per read so the 150-byte rows are written coalesced
    const uint32_t lane = lane_id();
    const uint64_t wid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t c = wid; c < n_reads; c += nw) {
        const uint64_t r = first + c, w0 = rnd(seed, 8 * r), w1 = rnd(seed, 8 * r + 1);
        uint8_t *dst = out + c * read_len;
        if ((w0 & 1ull) && n_genomes && genome_len >= read_len) {
            const uint64_t g = (w0 >> 8) % n_genomes, o = w1 % (genome_len - read_len + 1);
            const uint8_t *src = genomes + g * genome_len + o;
            for (uint64_t j = lane; j < read_len; j += 64)
                dst[j] = (w0 & 2ull) ? comp_acgt(src[read_len - 1 - j]) : src[j];
        } else {
            for (uint64_t j = lane; j < read_len; j += 64)
                dst[j] = acgt((uint32_t)(rnd(seed ^ 0xA5A5A5A5A5A5A5A5ull, r * 64 + (j >> 5)) >> (2 * (j & 31))));
        }
    }
}
void launch_synth_reads(uint8_t *d_out, uint64_t first, uint64_t n_reads, uint64_t read_len, const uint8_t *d_genomes,
                        uint64_t genome_len, uint64_t n_genomes, uint64_t seed, hipStream_t st) {
    hipLaunchKernelGGL(k_synth_reads, dim3(4096), dim3(256), 0, st, d_out, first, n_reads, read_len, d_genomes, genome_len, n_genomes, seed);
}

}  // namespace pfq
