// Microbenchmark 2: random gathers with pow2 tables (mask, no modulo), cache-policy variants and
// access shapes; run under rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum to get bytes/access.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"HIP error %s at %s:%d\n",hipGetErrorString(e),__FILE__,__LINE__); exit(1);} }while(0)
__device__ __forceinline__ uint64_t splitmix(uint64_t x){
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x>>30))*0xBF58476D1CE4E5B9ull; x = (x ^ (x>>27))*0x94D049BB133111EBull; return x ^ (x>>31);
}
template<int FL> __device__ __forceinline__ void ld(uint32_t& v, const uint32_t* p){
  if constexpr (FL==0) asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  if constexpr (FL==1) asm volatile("global_load_dword %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
  if constexpr (FL==2) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  if constexpr (FL==3) asm volatile("global_load_dword %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  if constexpr (FL==4) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  if constexpr (FL==5) asm volatile("global_load_dword %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
  if constexpr (FL==6) asm volatile("global_load_dword %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
  if constexpr (FL==7) asm volatile("global_load_dword %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
}
// SHAPE 0: each lane its own random 32B sector (dword).  SHAPE 1: 2 adjacent lanes share a random 64B block
// (lane parity picks the sector).  SHAPE 2: 4 lanes share a random 128B line (one sector each).
// SHAPE 3: 32 lanes read one random 128B row contiguously (coalesced).  SHAPE 4: 8 lanes cover one 32B sector contiguously.
template<int FL,int SHAPE,int U>
__global__ void __launch_bounds__(256) k_gather(const uint32_t* __restrict__ tab, uint64_t mask_units, uint32_t iters, uint32_t* out){
  uint64_t gid = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for(uint32_t it=0; it<iters; ++it){
    uint32_t v[U];
#pragma unroll
    for(int u=0;u<U;++u){
      uint64_t grp = SHAPE==0? gid : SHAPE==1? (gid>>1) : SHAPE==2? (gid>>2) : SHAPE==3? (gid>>5) : (gid>>3);
      uint64_t key = (grp*iters + it)*U + u;
      uint64_t r = splitmix(key);
      const uint32_t* p;
      if (SHAPE==0) p = tab + (r & mask_units)*8 + ((r>>58)&7);                   // unit=32B
      else if (SHAPE==1) p = tab + ((r & mask_units)>>1)*16 + (gid&1)*8 + ((r>>58)&7);   // 64B blocks
      else if (SHAPE==2) p = tab + ((r & mask_units)>>2)*32 + (gid&3)*8 + ((r>>58)&7);   // 128B lines
      else if (SHAPE==3) p = tab + ((r & mask_units)>>2)*32 + (gid&31);
      else p = tab + (r & mask_units)*8 + (gid&7);
      ld<FL>(v[u], p);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for(int u=0;u<U;++u){ asm volatile("" : "+v"(v[u])); acc ^= v[u]; }
  }
  if (acc == 0x12345678u) out[gid & 1023] = acc;
}
template<int FL,int SHAPE,int U>
void run(const char* name, const uint32_t* tab, uint64_t bytes, int blocks, uint32_t iters, uint32_t* out){
  uint64_t mask = bytes/32 - 1;
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  k_gather<FL,SHAPE,U><<<blocks,256>>>(tab,mask,iters/8+1,out); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  k_gather<FL,SHAPE,U><<<blocks,256>>>(tab,mask,iters,out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms,a,b));
  double lanes = (double)blocks*256*iters*U;
  double div = SHAPE==0?1: SHAPE==1?2: SHAPE==2?4: SHAPE==3?32:8;
  double sect = SHAPE==3? lanes/8 : SHAPE==4? lanes/8 : lanes;   // distinct 32B sectors touched
  printf("  %-22s FL%d SH%d: %.2f ms  lane-loads %.1f G/s  groups %.2f G/s  sectors %.1f G/s (%.2f TB/s @32B/sector)\n",
     name, FL, SHAPE, ms, lanes/ms/1e6, lanes/div/ms/1e6, sect/ms/1e6, sect*32/ms/1e9);
  fflush(stdout);
}
int main(int argc,char**argv){
  uint32_t* out; CK(hipMalloc(&out,4096));
  uint64_t sizes[] = {1ull<<27, 1ull<<33, 1ull<<34};
  int nsz = argc>1 ? atoi(argv[1]) : 3;
  for(int si=0; si<nsz; ++si){
    uint64_t bytes = sizes[si];
    uint32_t* tab; CK(hipMalloc(&tab,bytes)); CK(hipMemset(tab,0x5a,bytes)); CK(hipDeviceSynchronize());
    printf("table %.3f GiB\n", bytes/1073741824.0);
    const int B=4096; const uint32_t IT=32;
    run<0,0,8>("dword plain",tab,bytes,B,IT,out);
    run<1,0,8>("dword sc0",tab,bytes,B,IT,out);
    run<2,0,8>("dword sc1",tab,bytes,B,IT,out);
    run<3,0,8>("dword nt",tab,bytes,B,IT,out);
    run<4,0,8>("dword sc0sc1",tab,bytes,B,IT,out);
    run<5,0,8>("dword sc0nt",tab,bytes,B,IT,out);
    run<6,0,8>("dword sc1nt",tab,bytes,B,IT,out);
    run<7,0,8>("dword sc0sc1nt",tab,bytes,B,IT,out);
    run<0,1,8>("pair64 plain",tab,bytes,B,IT,out);
    run<3,1,8>("pair64 nt",tab,bytes,B,IT,out);
    run<0,2,8>("quad128 plain",tab,bytes,B,IT,out);
    run<3,2,8>("quad128 nt",tab,bytes,B,IT,out);
    run<0,3,8>("row128 plain",tab,bytes,B,IT,out);
    run<3,3,8>("row128 nt",tab,bytes,B,IT,out);
    run<0,4,8>("sector32 plain",tab,bytes,B,IT,out);
    run<3,4,8>("sector32 nt",tab,bytes,B,IT,out);
    CK(hipFree(tab));
  }
  return 0;
}
