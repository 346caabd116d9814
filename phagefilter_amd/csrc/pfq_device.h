// pfq_device.h — device-side building blocks of the classification path (gfx950 only).
//
//   K1  canonical k-mer + seeded FxHash + double hashing + exact `% nbits`
//       replaces file_parser.rs:114-148 (get_lex_less/get_kmers), hasher.rs:12-21, hash_iter.rs:13-45 and the
//       `h % self.bits.len()` of bloom_filter.rs:319.  k-mers are never materialised: a wave stages a 64-k-mer
//       window of the read (forward bytes + reverse-complement bytes) in LDS and every lane hashes one k-mer
//       straight out of LDS; the num_hashes indices live in registers only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pfq {

// ---- tree-wide constants handed to every kernel (by value) -------------------------------------------------
struct HashParams {
    uint32_t k;           // kmer_size (1..PFQ_KMAX)
    uint32_t num_hashes;  // 2..200 (bloom_filter.rs:342-350)
    uint64_t nbits;       // d  (< 2^32)
    uint64_t bar_m;       // floor((2^64-1)/d)  Barrett multiplier
    uint64_t w64;         // 2^64 mod d
    uint64_t a1, a2;      // ((seed_s*K + k)*K)*K : FxHasher state after write_usize(seed), write_usize(len), times K once more
                          // (the last round is (state + hb)*K = state*K + hb*K: hb*K is shared by the two seeded hashes)
};

constexpr uint32_t KMAX = 64;                        // supported k-mer length on the device path
constexpr uint32_t WIN_KMERS = 64;                   // k-mers per window = lanes per wave
constexpr uint32_t WIN_PAD = 8;                      // slack so aligned dword over-reads stay inside the buffer
constexpr uint32_t WIN_BYTES = 160;                  // >= WIN_PAD + (WIN_KMERS + KMAX - 1) + WIN_PAD, dword multiple
constexpr uint32_t WIN_DWORDS = WIN_BYTES / 4;
constexpr uint32_t WAVES_PER_BLOCK = 4;

constexpr uint64_t FX_K = 0xf1357aea2e62a9c5ull;      // rustc-hash 2.1 (64-bit)
constexpr uint64_t FX_SEED1 = 0x243f6a8885a308d3ull;
constexpr uint64_t FX_SEED2 = 0x13198a2e03707344ull;
constexpr uint64_t FX_PTZC = 0xa4093822299f31d0ull;

// LDS of one block: per-wave forward / reverse-complement windows + the shared complement table.
// (Staging a whole 192- or 448-k-mer segment of the read at once was tried: it removes global round trips for
// 150 bp reads but costs ~30 VGPRs in k_classify (occupancy 6 -> 4 waves/SIMD) and measured 6 % slower.)
struct BlockLds {
    uint32_t win[WAVES_PER_BLOCK][2][WIN_DWORDS];
    uint8_t comp[256];
};

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint64_t ballot64(bool p) { return __ballot(p); }
__device__ __forceinline__ uint32_t bcast_u32(uint32_t v, int src) { return __builtin_amdgcn_readlane(v, src); }

// bio::alphabets::dna complement (bio 2.2.0; used by file_parser.rs:115): identity except the IUPAC pairs
// A<->T C<->G Y<->R W<->W S<->S K<->M D<->H V<->B N<->N and their lowercase forms.  Built at compile time,
// copied to LDS once per block.
struct CompTable {
    uint8_t t[256];
    constexpr CompTable() : t{} {
        for (int i = 0; i < 256; ++i) t[i] = (uint8_t)i;
        const char a[] = "AGCTYRWSKMDVHBN", b[] = "TCGARYWSMKHBDVN";
        for (int i = 0; i < 15; ++i) {
            t[(uint8_t)a[i]] = (uint8_t)b[i];
            t[(uint8_t)a[i] + 32] = (uint8_t)(b[i] + 32);
        }
    }
};
__device__ __constant__ const CompTable COMP_TABLE{};
__device__ __forceinline__ void fill_complement(uint8_t *comp) {
    for (uint32_t i = threadIdx.x; i < 64; i += blockDim.x)
        reinterpret_cast<uint32_t *>(comp)[i] = reinterpret_cast<const uint32_t *>(COMP_TABLE.t)[i];
}

// ---- LDS byte-granular reads ----------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lds_u32(const uint32_t *w, uint32_t a) {  // unaligned little-endian u32
    uint32_t i = a >> 2;
    return __builtin_amdgcn_alignbyte(w[i + 1], w[i], a & 3u);
}
__device__ __forceinline__ uint64_t lds_u64(const uint32_t *w, uint32_t a) {  // unaligned little-endian u64
    uint32_t i = a >> 2, s = a & 3u;
    uint32_t d0 = w[i], d1 = w[i + 1], d2 = w[i + 2];
    uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, s), hi = __builtin_amdgcn_alignbyte(d2, d1, s);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t lds_u8(const uint32_t *w, uint32_t a) { return (w[a >> 2] >> (8u * (a & 3u))) & 0xffu; }

// ---- window staging ---------------------------------------------------------------------------------------------
// Stage k-mers [base, base+cnt) of a read (cnt <= 64): forward bytes read[base .. base+cnt+k-1) at fwd[WIN_PAD..]
// and their reverse complement at rc[WIN_PAD..].  The k-mer at window position q is fwd[WIN_PAD+q, +k) and its
// reverse complement is rc[WIN_PAD + W-q-k, +k), W = cnt+k-1.  Whole wave must call this convergently.
__device__ __forceinline__ void stage_window(BlockLds &lds, uint32_t wave, const uint8_t *__restrict__ read,
                                             uint64_t base, uint32_t cnt, uint32_t k) {
    uint32_t lane = lane_id();
    uint32_t W = cnt + k - 1;  // <= 127
    uint8_t *fwd = reinterpret_cast<uint8_t *>(lds.win[wave][0]);
    uint8_t *rc = reinterpret_cast<uint8_t *>(lds.win[wave][1]);
    __builtin_amdgcn_wave_barrier();
    // both byte loads are issued before either is consumed (W <= 127: two bytes per lane)
    const bool v0 = lane < W, v1 = lane + 64u < W;
    uint8_t b0 = 0, b1 = 0;
    if (v0) b0 = read[base + lane];
    if (v1) b1 = read[base + lane + 64u];
    if (v0) {
        fwd[WIN_PAD + lane] = b0;
        rc[WIN_PAD + (W - 1 - lane)] = lds.comp[b0];
    }
    if (v1) {
        fwd[WIN_PAD + lane + 64u] = b1;
        rc[WIN_PAD + (W - 1 - lane - 64u)] = lds.comp[b1];
    }
    __builtin_amdgcn_wave_barrier();
}

// ---- rustc-hash 2.1 ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t fx_mm(uint64_t x, uint64_t y) { return (x * y) ^ __umul64hi(x, y); }

// hash_bytes over n bytes at byte address `a` of LDS word array `w` (n is wave-uniform); `first8` = the first eight of
// them (the caller has read them for the canonical choice; ignored when n < 8).
__device__ __forceinline__ uint64_t fx_hash_bytes_lds(const uint32_t *w, uint32_t a, uint32_t n, uint64_t first8) {
    uint64_t s0 = FX_SEED1, s1 = FX_SEED2;
    if (n <= 16) {
        if (n >= 8) {
            s0 ^= first8;
            s1 ^= lds_u64(w, a + n - 8);
        } else if (n >= 4) {
            s0 ^= lds_u32(w, a);
            s1 ^= lds_u32(w, a + n - 4);
        } else if (n > 0) {
            uint64_t lo = lds_u8(w, a), mid = lds_u8(w, a + n / 2), hi = lds_u8(w, a + n - 1);
            s0 ^= lo;
            s1 ^= (hi << 8) | mid;
        }
    } else {
        uint64_t x = first8;
        uint32_t off = 0;
        do {
            const uint64_t y = lds_u64(w, a + off + 8);
            uint64_t t = fx_mm(s0 ^ x, FX_PTZC ^ y);
            s0 = s1;
            s1 = t;
            off += 16;
            if (off >= n - 16) break;
            x = lds_u64(w, a + off);
        } while (true);
        s0 ^= lds_u64(w, a + n - 16);
        s1 ^= lds_u64(w, a + n - 8);
    }
    return fx_mm(s0, s1) ^ (uint64_t)n;
}

__device__ __forceinline__ uint64_t rotl64(uint64_t x, uint32_t r) { return (x << r) | (x >> (64 - r)); }

// The two seeded hashes of the canonical k-mer at window position q (per lane).
// Canonical choice (file_parser.rs:116-120): forward unless revcomp is bytewise smaller.
// fw / rw: LDS word arrays holding forward and reverse-complement bytes; fa / ra: per-lane byte addresses of the
// forward k-mer and of its reverse complement inside them.
__device__ __forceinline__ void kmer_hashes_at(const uint32_t *fw, const uint32_t *rw, uint32_t fa, uint32_t ra, bool valid,
                                               const HashParams &hp, uint64_t &h1, uint64_t &h2) {
    const uint32_t k = hp.k;
    if (!valid) { fa = WIN_PAD; ra = WIN_PAD; }
    // bytewise lexicographic compare == compare of the byte-swapped words, first difference decides.  The first eight bytes
    // decide nearly always (and the chosen strand's eight are the first word of the hash); the loop is for what is left.
    bool use_rc = false, decided = !valid;
    uint64_t f8 = 0, r8 = 0;
    uint32_t j0 = 0;
    if (k >= 8) {  // (wave-uniform)
        f8 = lds_u64(fw, fa);
        r8 = lds_u64(rw, ra);
        use_rc = valid && __builtin_bswap64(r8) < __builtin_bswap64(f8);
        decided = decided || f8 != r8;
        j0 = 8;
    }
    if (ballot64(!decided) != 0) {
        for (uint32_t j = j0; j < k; j += 4) {
            uint32_t f = lds_u32(fw, fa + j), r = lds_u32(rw, ra + j);
            uint32_t rem = k - j;
            if (rem < 4) { uint32_t m = (1u << (8u * rem)) - 1u; f &= m; r &= m; }
            if (!decided && f != r) {
                use_rc = __builtin_bswap32(r) < __builtin_bswap32(f);
                decided = true;
            }
            if (ballot64(!decided) == 0) break;
        }
    }
    const uint32_t *cw = use_rc ? rw : fw;
    uint32_t ca = use_rc ? ra : fa;
    const uint64_t hbk = fx_hash_bytes_lds(cw, ca, k, use_rc ? r8 : f8) * FX_K;
    h1 = rotl64(hp.a1 + hbk, 26);
    h2 = rotl64(hp.a2 + hbk, 26);
}
// k-mer at position q of the window staged by stage_window(…, cnt, …)
__device__ __forceinline__ void kmer_hashes(const BlockLds &lds, uint32_t wave, uint32_t q, uint32_t cnt, bool valid,
                                            const HashParams &hp, uint64_t &h1, uint64_t &h2) {
    const uint32_t W = cnt + hp.k - 1;
    kmer_hashes_at(lds.win[wave][0], lds.win[wave][1], WIN_PAD + q, WIN_PAD + (W - q - hp.k), valid, hp, h1, h2);
}

// ---- dense pre-screen staging: the first DENSE_KMERS k-mers of DENSE_READS reads per wave ------------------------------
constexpr uint32_t DENSE_READS = 16, DENSE_KMERS = 4;               // 16 x 4 = 64 lanes
constexpr uint32_t MINI_BYTES = 84;                                  // >= WIN_PAD + (KMAX + DENSE_KMERS - 1) + WIN_PAD, dword multiple
template <bool ENABLED>
struct alignas(16) DenseLds {
    uint32_t mini[WAVES_PER_BLOCK][2][DENSE_READS * MINI_BYTES / 4];
    uint32_t live[WAVES_PER_BLOCK][DENSE_READS * 64];  // frontier words of the reads of a group, [read][row word]
};
template <>
struct DenseLds<false> {
    uint32_t mini[1][2][1];
    uint32_t live[1][1];
};

// ---- exact `% nbits` ----------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mod_nbits(uint64_t r, const HashParams &hp) {
    uint64_t q = __umul64hi(r, hp.bar_m);
    uint64_t rem = r - q * hp.nbits;  // in [0, 2d]
    rem -= (rem >= hp.nbits) ? hp.nbits : 0;
    rem -= (rem >= hp.nbits) ? hp.nbits : 0;
    return (uint32_t)rem;
}

// The same for d < 2^30: r - q*d < 3d < 2^32, so only the low words of r and q*d are needed and the two conditional
// subtractions are 32-bit min() pairs.  (bar_m = floor((2^64-1)/d): floor(r/d) - 2 <= q <= floor(r/d).)
__device__ __forceinline__ uint32_t mod_nbits30(uint64_t r, const HashParams &hp) {
    const uint32_t q = (uint32_t)__umul64hi(r, hp.bar_m), d = (uint32_t)hp.nbits;
    uint32_t rem = (uint32_t)r - q * d;
    rem = min(rem, rem - d);
    rem = min(rem, rem - d);
    return rem;
}

__device__ __forceinline__ uint32_t mod_d(uint64_t r, const HashParams &hp) {  // (wave-uniform choice)
    return hp.nbits < (1ull << 30) ? mod_nbits30(r, hp) : mod_nbits(r, hp);
}

// Iterator over the probe indices of one k-mer: i=0 -> h1 % d, i=1 -> h2 % d, i>=2 -> ((h1+i)*h2 mod 2^64) % d
// (hash_iter.rs:13-27, bloom_filter.rs:319).  For i >= 3 the value advances by +h2 (mod 2^64), so the residue
// advances by (h2 % d) and, when the 64-bit add wraps, by -(2^64 % d): no further wide multiplies.
struct ProbeIter {
    uint64_t r, h2;
    uint32_t x, g, i0;
    // after init: i0 = index 0, g = index 1, x = index 2
    __device__ __forceinline__ void init(uint64_t h1, uint64_t h2_, const HashParams &hp) {
        h2 = h2_;
        i0 = mod_d(h1, hp);
        g = mod_d(h2_, hp);
        r = (h1 + 2) * h2_;
        x = mod_d(r, hp);
    }
    // index 3, 4, ... on successive calls
    __device__ __forceinline__ uint32_t step(const HashParams &hp) {
        uint64_t rn = r + h2;
        bool carry = rn < r;
        r = rn;
        if (hp.nbits < (1ull << 30)) {  // wave-uniform: 32-bit residue arithmetic, x + g + (d - w) < 3d < 2^32
            const uint32_t d = (uint32_t)hp.nbits;
            uint32_t t = x + g + (carry ? d - (uint32_t)hp.w64 : 0u);
            t = min(t, t - d);  // t >= d ? t - d : t   (t - d wraps to a huge value when t < d)
            t = min(t, t - d);
            x = t;
        } else {
            uint64_t t = (uint64_t)x + g + (carry ? (hp.nbits - hp.w64) : 0ull);  // < 3d
            t -= (t >= hp.nbits) ? hp.nbits : 0;
            t -= (t >= hp.nbits) ? hp.nbits : 0;
            x = (uint32_t)t;
        }
        return x;
    }
};

// Calls f(index) for the num_hashes probe indices of a k-mer in order (hash_iter.rs:13-27).
template <typename F>
__device__ __forceinline__ void for_each_probe(ProbeIter &it, const HashParams &hp, F &&f) {
    f(it.i0);
    if (hp.num_hashes > 1) f(it.g);
    if (hp.num_hashes > 2) f(it.x);
    for (uint32_t i = 3; i < hp.num_hashes; ++i) f(it.step(hp));
}

// Probe record of one k-mer: everything the bucketed verify needs to regenerate the num_hashes indices with
// 32-bit arithmetic only: x = h1 % d, y = h2 % d, z = ((h1+2)*h2 mod 2^64) % d, w = wrap-carry bits of the
// 64-bit walk (bit i-3 set iff (h1+i)*h2 wrapped relative to (h1+i-1)*h2, i = 3..num_hashes-1).
// Valid for d < 2^30 and num_hashes <= 35.
__device__ __forceinline__ uint4 make_probe_record(uint64_t h1, uint64_t h2, const HashParams &hp) {
    uint4 rec;
    rec.x = mod_nbits30(h1, hp);
    rec.y = mod_nbits30(h2, hp);
    uint64_t r = (h1 + 2) * h2;
    rec.z = mod_nbits30(r, hp);
    // the walk r += h2 with the carry-out of every step shifted into cm (add, add-with-carry, cm = 2 cm + carry: three
    // instructions a step; compare-and-select code was seven); the first step ends up in the highest of the n bits
    uint32_t rl = (uint32_t)r, rh = (uint32_t)(r >> 32), cm = 0;
    const uint32_t hl = (uint32_t)h2, hh = (uint32_t)(h2 >> 32);
    const uint32_t n = hp.num_hashes > 3 ? hp.num_hashes - 3 : 0;
    for (uint32_t i = 0; i < n; ++i)
        asm("v_add_co_u32 %0, vcc, %0, %3\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\tv_addc_co_u32 %2, vcc, %2, %2, vcc"
            : "+v"(rl), "+v"(rh), "+v"(cm)
            : "v"(hl), "v"(hh)
            : "vcc");
    rec.w = n ? __brev(cm) >> (32u - n) : 0u;
    return rec;
}
struct RecordIter {
    uint32_t x, g, cm, i0;
    // after init: i0 = index 0, g = index 1, x = index 2
    __device__ __forceinline__ void init(const uint4 &rec) { i0 = rec.x; g = rec.y; x = rec.z; cm = rec.w; }
    // index 3, 4, ... on successive calls; d < 2^30, dw = d - (2^64 mod d)
    __device__ __forceinline__ uint32_t step(uint32_t d, uint32_t dw) {
        uint32_t t = x + g + ((cm & 1u) ? dw : 0u);
        cm >>= 1;
        t = min(t, t - d);
        t = min(t, t - d);
        x = t;
        return x;
    }
};

// The same walk with one conditional subtraction per step: the step is g, or g2 = (g + d - 2^64 mod d) mod d when the 64-bit
// add wrapped, both < d, so x + step < 2d.
struct RecordIter1 {
    uint32_t x, g, dg, cm, i0;  // dg = g2 - g (mod 2^32)
    __device__ __forceinline__ void init(const uint4 &rec, uint32_t d, uint32_t dw) {
        i0 = rec.x;
        g = rec.y;
        x = rec.z;
        cm = rec.w;
        uint32_t g2 = g + dw;
        g2 = min(g2, g2 - d);
        dg = g2 - g;
    }
    __device__ __forceinline__ uint32_t step(uint32_t d) {
        uint32_t t = x + g + (dg & (0u - (cm & 1u)));
        cm >>= 1;
        t = min(t, t - d);
        x = t;
        return x;
    }
};

// `(threshold * n as f32).ceil() as usize` (query.rs:48): IEEE f32 multiply (no contraction possible: single op),
// ceil, Rust's saturating float->int cast (NaN -> 0).
__device__ __forceinline__ uint64_t need_kmers(float threshold, uint64_t n) {
    float c = ceilf(__fmul_rn(threshold, (float)n));
    if (!(c > 0.0f)) return 0;
    if (c >= 18446744073709551616.0f) return ~0ull;
    return (uint64_t)c;
}

// ---- synthetic workload PRNG (mirrors oracle/pfq_oracle.c) --------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ uint64_t rnd(uint64_t seed, uint64_t i) { return splitmix64(splitmix64(seed) + i); }

}  // namespace pfq
