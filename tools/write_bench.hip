// Microbenchmark 4: HBM write bandwidth for the store patterns of k_tile_bin.
//   seq   : every block streams its own contiguous region with 16-byte stores (1 KiB per wave instruction)
//   runs  : every wave writes runs of `run` bytes (16-byte stores) round-robin into `streams` regions that each advance
//           sequentially — the shape of the (chunk, tile) bucket appends (69 tiles, ~1.2 KB runs)
//   mixed : runs + a streaming read of half the volume in the same kernel
// build: hipcc -O3 --offload-arch=gfx950 tools/write_bench.hip -o tools/write_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"HIP error %s at %s:%d\n",hipGetErrorString(e),__FILE__,__LINE__); exit(1);} }while(0)

__global__ void __launch_bounds__(1024) k_seq(uint4 *dst, uint64_t per_block16) {
  uint4 *p = dst + (uint64_t)blockIdx.x * per_block16;
  const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
  for (uint64_t i = threadIdx.x; i < per_block16; i += blockDim.x) p[i] = v;
}
// block b owns `streams` regions of region16 uint4 each; wave w appends runs to streams w, w+16, ...
__global__ void __launch_bounds__(1024) k_runs(uint4 *dst, uint32_t streams, uint64_t region16, uint32_t run16, uint32_t rounds,
                                               const uint4 *src, uint32_t read_per_round16) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint4 *base = dst + (uint64_t)blockIdx.x * streams * region16;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (uint32_t r = 0; r < rounds; ++r) {
    if (src) {
      const uint4 *s = src + ((uint64_t)blockIdx.x * rounds + r) * read_per_round16;
      for (uint32_t i = threadIdx.x; i < read_per_round16; i += blockDim.x) { uint4 x = s[i]; acc.x ^= x.x; acc.y += x.y; }
    }
    for (uint32_t t = wave; t < streams; t += 16) {
      uint4 *p = base + (uint64_t)t * region16 + (uint64_t)r * run16;
      for (uint32_t i = lane; i < run16; i += 64) p[i] = make_uint4(r, t, acc.x, acc.y);
    }
  }
}
int main() {
  const uint64_t total = 24ull << 30;  // bytes written per launch
  uint4 *dst, *src;
  CK(hipMalloc(&dst, total + (1 << 20)));
  CK(hipMalloc(&src, total / 2));
  CK(hipMemset(src, 1, total / 2));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto time = [&](const char *name, auto launch, double bytes) {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-44s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9); fflush(stdout);
  };
  for (int blocks : {256, 512, 2048}) {
    char nm[128]; snprintf(nm, sizeof nm, "seq 16B stores, %d blocks", blocks);
    time(nm, [&] { k_seq<<<blocks, 1024>>>(dst, total / 16 / blocks); }, (double)total);
  }
  for (uint32_t run : {256u, 1216u, 4096u, 16384u})
    for (uint32_t streams : {16u, 69u, 138u}) {
      const int blocks = 256;
      const uint32_t run16 = run / 16;
      const uint64_t per_block = total / blocks;
      const uint32_t rounds = (uint32_t)(per_block / ((uint64_t)streams * run16 * 16));
      const uint64_t region16 = (uint64_t)rounds * run16;
      char nm[128]; snprintf(nm, sizeof nm, "runs of %u B into %u streams/block", run16 * 16, streams);
      time(nm, [&] { k_runs<<<blocks, 1024>>>(dst, streams, region16, run16, rounds, nullptr, 0); }, (double)rounds * streams * run16 * 16 * blocks);
      if (run == 1216u && streams == 69u) {
        const uint32_t rd16 = streams * run16 / 2;
        time("  + streaming read of half the volume", [&] { k_runs<<<blocks, 1024>>>(dst, streams, region16, run16, rounds, src, rd16); },
             (double)rounds * streams * run16 * 16 * blocks * 1.5);
      }
    }
  return 0;
}
