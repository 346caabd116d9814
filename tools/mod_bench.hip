// `% nbits` two ways on gfx950: the Barrett reduction the kernels use (mod_nbits30: five quarter-rate integer multiplies) against
// a quotient estimated in f64 (one integer multiply).  Checks both against the exact `%` on random and edge operands, then times
// 256 reductions per thread.   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o /tmp/mod_bench tools/mod_bench.hip && /tmp/mod_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ uint32_t mod_barrett(uint64_t r, uint32_t d, uint64_t bar_m) {
    const uint32_t q = (uint32_t)__umul64hi(r, bar_m);
    uint32_t rem = (uint32_t)r - q * d;
    rem = min(rem, rem - d);
    rem = min(rem, rem - d);
    return rem;
}
// d >= 2^13: the f64 quotient is within 1 of the true one (r (1 + 2^-53) / d (1 + 2^-53) (1 + 2^-53): off by < 2^12.6 / d)
__device__ __forceinline__ uint32_t mod_f64(uint64_t r, uint32_t d, double inv_d) {
    const double x = fma((double)(uint32_t)(r >> 32), 4294967296.0, (double)(uint32_t)r);
    const double q = x * inv_d;
    const double qh = floor(q * (1.0 / 4294967296.0));
    const double ql = fma(-qh, 4294967296.0, q);  // exact: floor(q) mod 2^32 plus q's fraction
    const uint32_t q32 = (uint32_t)ql;
    uint32_t rem = (uint32_t)r - q32 * d + d;      // in [0, 3d)
    rem = min(rem, rem - d);
    rem = min(rem, rem - d);
    return rem;
}
__device__ __forceinline__ uint64_t xs(uint64_t r) {
    r ^= r << 13;
    r ^= r >> 7;
    r ^= r << 17;
    return r;
}
template <int MODE>
__global__ void __launch_bounds__(256) k_time(uint32_t d, uint64_t bar_m, double inv_d, uint32_t *out) {
    uint64_t r = 0x9E3779B97F4A7C15ull * (blockIdx.x * 256 + threadIdx.x + 1);
    uint32_t acc = 0;
    for (int i = 0; i < 256; ++i) {
        r = xs(r);
        if (MODE == 0) acc += (uint32_t)r;
        if (MODE == 1) acc += mod_barrett(r, d, bar_m);
        if (MODE == 2) acc += mod_f64(r, d, inv_d);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_check(uint32_t d, uint64_t bar_m, double inv_d, uint64_t seed, unsigned long long *bad) {
    uint64_t r = seed * (blockIdx.x * 256 + threadIdx.x + 1);
    for (int i = 0; i < 64; ++i) {
        r = xs(r);
        uint64_t v = r;
        if ((i & 7) == 1) v = (r / d) * d + (i >> 3) - 4;  // around the multiples of d
        if ((i & 7) == 2) v = ~0ull - (r & 0xffff);        // the top of the range
        if ((i & 7) == 3) v = r & 0xffffffffull;           // small operands
        const uint32_t want = (uint32_t)(v % d);
        if (mod_barrett(v, d, bar_m) != want) atomicAdd(&bad[0], 1ull);
        if (mod_f64(v, d, inv_d) != want) atomicAdd(&bad[1], 1ull);
    }
}
int main() {
    const uint32_t ds[] = {8192, 8193, 60013, 11981322, 71887936, 536870909, 1073741823};
    unsigned long long *bad;
    uint32_t *out;
    hipMalloc(&bad, 16);
    hipMalloc(&out, 4096 * 256 * 4);
    for (uint32_t d : ds) {
        const uint64_t bar_m = ~0ull / d;
        const double inv_d = 1.0 / (double)d;
        hipMemset(bad, 0, 16);
        for (uint64_t s = 1; s <= 8; ++s) hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, d, bar_m, inv_d, 0x9E3779B97F4A7C15ull * s + 12345, bad);
        unsigned long long h[2];
        hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
        printf("d = %10u: %llu operands, mismatches barrett %llu, f64 %llu\n", d, 8ull * 4096 * 256 * 64, h[0], h[1]);
    }
    const uint32_t d = 71887936;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, 0);
            for (int j = 0; j < 10; ++j) {
                if (mode == 0) hipLaunchKernelGGL(k_time<0>, dim3(4096), dim3(256), 0, 0, d, ~0ull / d, 1.0 / d, out);
                if (mode == 1) hipLaunchKernelGGL(k_time<1>, dim3(4096), dim3(256), 0, 0, d, ~0ull / d, 1.0 / d, out);
                if (mode == 2) hipLaunchKernelGGL(k_time<2>, dim3(4096), dim3(256), 0, 0, d, ~0ull / d, 1.0 / d, out);
            }
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("mode %d (%s): %.3f ms for 10 x 2^28 reductions\n", mode, mode == 0 ? "generator only" : mode == 1 ? "barrett" : "f64", best);
    }
    return 0;
}
