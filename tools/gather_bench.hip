// Microbenchmark: random sector-granular gathers from a large HBM-resident table (gfx950).
// Informs the Bloom-probe layout: what random-access rate can the probe kernel count on?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"HIP error %s at %s:%d\n",hipGetErrorString(e),__FILE__,__LINE__); exit(1);} }while(0)

__device__ __forceinline__ uint64_t splitmix(uint64_t x){
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x>>30))*0xBF58476D1CE4E5B9ull; x = (x ^ (x>>27))*0x94D049BB133111EBull; return x ^ (x>>31);
}

// MODE 0: every lane loads one random dword (each in its own random 32B sector); U independent loads in flight per lane.
// MODE 1: same with nontemporal loads.
// MODE 2: wave-cooperative rows: 32 lanes read one random 128B row (2 rows per wave instr).
// MODE 3: every lane loads a random 16B (dwordx4).
template<int MODE, int U>
__global__ void __launch_bounds__(256) gather(const uint32_t* __restrict__ tab, uint64_t n_units, uint32_t iters, uint32_t* out){
  uint64_t gid = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  uint32_t lane = threadIdx.x & 63;
  for(uint32_t it=0; it<iters; ++it){
    uint32_t v[U];
#pragma unroll
    for(int u=0;u<U;++u){
      if (MODE==2){
        uint64_t key = ((gid>>5)*iters + it)*U + u;      // one random row per 32 lanes
        uint64_t r = splitmix(key) % n_units;            // unit = 128B row
        v[u] = tab[r*32 + (lane&31)];
      } else if (MODE==3){
        uint64_t key = (gid*iters + it)*U + u;
        uint64_t r = splitmix(key) % n_units;            // unit = 32B sector
        const uint4* p = reinterpret_cast<const uint4*>(tab + r*8);
        uint4 q = *p; v[u] = q.x ^ q.y ^ q.z ^ q.w;
      } else {
        uint64_t key = (gid*iters + it)*U + u;
        uint64_t r = splitmix(key) % n_units;            // unit = 32B sector
        const uint32_t* p = tab + r*8 + (key & 7);
        if (MODE==1) v[u] = __builtin_nontemporal_load(p); else v[u] = *p;
      }
    }
#pragma unroll
    for(int u=0;u<U;++u) acc ^= v[u];
  }
  if (acc == 0x12345678u) out[gid & 1023] = acc;
}

template<int MODE,int U>
double run(const uint32_t* tab, uint64_t bytes, int blocks, uint32_t iters, uint32_t* out){
  uint64_t n_units = (MODE==2) ? bytes/128 : bytes/32;
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  gather<MODE,U><<<blocks,256>>>(tab,n_units,iters/4+1,out); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  gather<MODE,U><<<blocks,256>>>(tab,n_units,iters,out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms,a,b));
  double accesses = (double)blocks*256*iters*U; if (MODE==2) accesses/=32;  // rows for mode 2
  return accesses/(ms*1e-3);
}

int main(int argc,char**argv){
  double gbs[] = {0.125, 1.0, 9.2, 18.4};
  uint32_t* out; CK(hipMalloc(&out,4096));
  for(double gb: gbs){
    uint64_t bytes = (uint64_t)(gb*1e9); bytes &= ~uint64_t(4095);
    uint32_t* tab; CK(hipMalloc(&tab,bytes)); CK(hipMemset(tab,0x5a,bytes)); CK(hipDeviceSynchronize());
    for(int blocks: {1024, 2048, 4096}){
      uint32_t iters = 64;
      double r0 = run<0,8>(tab,bytes,blocks,iters,out);
      double r0b= run<0,16>(tab,bytes,blocks,iters/2,out);
      double r1 = run<1,8>(tab,bytes,blocks,iters,out);
      double r3 = run<3,8>(tab,bytes,blocks,iters,out);
      double r2 = run<2,8>(tab,bytes,blocks,iters,out);
      printf("table %.3f GB blocks %d | dword U8 %.2f G/s (%.2f TB/s@32B) | U16 %.2f G/s | nt U8 %.2f G/s | x4 U8 %.2f G/s | row128 %.2f Grows/s (%.2f TB/s)\n",
        gb, blocks, r0/1e9, r0*32/1e12, r0b/1e9, r1/1e9, r3/1e9, r2/1e9, r2*128/1e12);
      fflush(stdout);
    }
    CK(hipFree(tab));
  }
  return 0;
}
