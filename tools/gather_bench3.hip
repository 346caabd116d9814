// Microbenchmark 3: L2-resident random dword gathers. Each block reads its XCC_ID and gathers only from
// that XCD's private slice (slice_bytes each), so every XCD's working set stays in its own 4 MiB L2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"HIP error %s at %s:%d\n",hipGetErrorString(e),__FILE__,__LINE__); exit(1);} }while(0)
__device__ __forceinline__ uint64_t splitmix(uint64_t x){
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x>>30))*0xBF58476D1CE4E5B9ull; x = (x ^ (x>>27))*0x94D049BB133111EBull; return x ^ (x>>31);
}
__device__ __forceinline__ uint32_t xcc_id(){ return __builtin_amdgcn_s_getreg((31u<<11)|20u) & 0xFu; }

__global__ void k_xcc(uint32_t* out){ if(threadIdx.x==0) out[blockIdx.x] = xcc_id(); }

template<int U>
__global__ void __launch_bounds__(256) k_l2(const uint32_t* __restrict__ tab, uint32_t slice_dwords_mask, uint64_t slice_stride_dwords,
                                           uint32_t iters, uint32_t* out, int use_xcc){
  uint64_t gid = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x;
  uint32_t x = use_xcc ? xcc_id() : (blockIdx.x & 7);
  const uint32_t* base = tab + (uint64_t)(x&7)*slice_stride_dwords;
  uint32_t acc=0;
  for(uint32_t it=0; it<iters; ++it){
    uint32_t v[U];
#pragma unroll
    for(int u=0;u<U;++u){
      uint64_t r = splitmix((gid*iters+it)*U+u);
      v[u] = base[(uint32_t)r & slice_dwords_mask];
    }
#pragma unroll
    for(int u=0;u<U;++u) acc ^= v[u];
  }
  if (acc == 0x12345678u) out[gid & 1023] = acc;
}
int main(){
  uint32_t* out; CK(hipMalloc(&out,1<<20));
  // XCC id map
  k_xcc<<<64,64>>>(out); CK(hipDeviceSynchronize());
  uint32_t h[64]; CK(hipMemcpy(h,out,sizeof h,hipMemcpyDeviceToHost));
  printf("xcc of blocks 0..63:"); for(int i=0;i<64;++i) printf(" %u",h[i]); printf("\n");
  const uint64_t stride = (16u<<20)/4; // 16 MiB apart
  uint32_t* tab; CK(hipMalloc(&tab, stride*4*8)); CK(hipMemset(tab,0x5a,stride*4*8)); CK(hipDeviceSynchronize());
  for(int use_xcc=1; use_xcc>=0; --use_xcc)
  for(uint32_t kb : {256u, 512u, 1024u, 2048u, 4096u, 8192u}){
    uint32_t mask = kb*1024/4 - 1;
    for(int blocks : {2048, 4096, 8192}){
      const uint32_t IT=64;
      hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      k_l2<8><<<blocks,256>>>(tab,mask,stride,IT,out,use_xcc); CK(hipDeviceSynchronize());
      CK(hipEventRecord(a));
      k_l2<8><<<blocks,256>>>(tab,mask,stride,IT,out,use_xcc);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms,a,b));
      double n = (double)blocks*256*IT*8;
      printf("use_xcc %d slice %5u KiB/XCD blocks %5d: %.3f ms  %.1f G gathers/s\n", use_xcc, kb, blocks, ms, n/ms/1e6);
      fflush(stdout);
    }
  }
  return 0;
}
