// Micro-benchmark: issue rate of the integer VALU instructions the field arithmetic is built from.
// Each kernel runs 32 independent inline-asm instructions per loop iteration; reports cycles per
// wave-instruction per SIMD (at the nominal 2.4 GHz) for a grid that fills every SIMD with 8 waves.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32; typedef uint64_t u64;
#define ITERS 2048
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define R32(X) R8(X) R8(X) R8(X) R8(X)

template<int OP> __global__ void k(u32* out, u32 seed, unsigned long long* clk){
  u32 t=threadIdx.x+blockIdx.x*blockDim.x;
  unsigned long long c0=clock64(), w0=wall_clock64();
  u32 a[8]; u64 c[8]; double d[8];
  for(int i=0;i<8;i++){ a[i]=t*2654435761u+seed*(i+1); c[i]=((u64)a[i]<<32)|(a[i]^0x55aa55aau); d[i]=1.0+a[i]*1e-9; }
  for(int it=0;it<ITERS;it++){
    if(OP==0){
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[i]) : "v"(a[i]), "v"(a[(i+1)&7]) : "vcc");
      R32(X)
#undef X
    } else if(OP==1){
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c[i]) : "v"(c[(i+1)&7]));
      R32(X)
#undef X
    } else if(OP==2){
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i+1)&7]));
      R32(X)
#undef X
    } else if(OP==3){
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i+1)&7]));
      R32(X)
#undef X
    } else if(OP==4){
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i+1)&7]));
      R32(X)
#undef X
    } else if(OP==5){
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i+1)&7]));
      R32(X)
#undef X
    } else if(OP==6){
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i+1)&7]));
      R32(X)
#undef X
    } else if(OP==7){
#define X(i) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(c[i]));
      R32(X)
#undef X
    } else if(OP==8){
#define X(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(a[i]), "+v"(a[(i+4)&7]) : "v"(a[(i+1)&7]), "v"(a[(i+2)&7]) : "vcc");
      R8(X) R8(X)
#undef X
    } else if(OP==9){
#define X(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(a[(i+1)&7]));
      R32(X)
#undef X
    } else if(OP==10){
#define X(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i+1)&7]));
      R32(X)
#undef X
    } else if(OP==11){
#define X(i) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(a[(i+1)&7]));
      R32(X)
#undef X
    } else if(OP==12){  // mixed: 1 mad + 3 adds, to see whether they overlap
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %2, %2, %1\n\tv_add_u32 %1, %1, %2" : "+v"(c[i]), "+v"(a[i]), "+v"(a[(i+1)&7]) :: "vcc");
      R8(X)
#undef X
    }
  }
  u64 cs=0; u32 as=0; double ds=0; for(int i=0;i<8;i++){cs^=c[i]; as+=a[i]; ds+=d[i];}
  out[t]=as+(u32)cs+(u32)(cs>>32)+(u32)ds;
  if(t==0){ clk[0]=clock64()-c0; clk[1]=wall_clock64()-w0; }
}
static unsigned long long* g_clk;
template<int OP> int run(const char* name, int ops_per_iter, u32* d, int wavesPerSimd){
  hipEvent_t e0,e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  int threads=256, blocks=256*wavesPerSimd;
  k<OP><<<blocks,threads>>>(d,1,g_clk); CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0)); for(int r=0;r<4;r++) k<OP><<<blocks,threads>>>(d,2+r,g_clk); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms,e0,e1)); ms/=4;
  double waveinstr=(double)blocks*threads/64*ITERS*ops_per_iter;
  unsigned long long hc[2]; CHK(hipMemcpy(hc,g_clk,16,hipMemcpyDeviceToHost));
  // the first workgroup's own counters: shader-clock ticks (s_memtime) against the constant 100 MHz clock (s_memrealtime)
  double mhz = hc[1] ? (double)hc[0] / ((double)hc[1] / 100.0) : 0.0;
  printf("%-28s waves/SIMD=%d %8.3f ms  %.3e wave-instr/s  %.2f cyc/wave-instr/SIMD @2.4GHz  (s_memtime/s_memrealtime: %.0f MHz)\n",name,wavesPerSimd,ms,waveinstr/(ms*1e-3), 2.4e9*1024/(waveinstr/(ms*1e-3)), mhz);
  return 0;
}
int main(int argc, char** argv){
  u32* d; CHK(hipMalloc(&d, 256*8*256*4)); CHK(hipMalloc(&g_clk, 16));
  if(argc>1){
    // sustained load: the multiply-add mix for ~0.3 s, clock of the last launches (the chip settles below its short-burst clock)
    const int blocks=256*8, threads=256;
    for(int rep=0;rep<6;rep++){
      for(int r=0;r<80;r++) k<12><<<blocks,threads>>>(d,2+r,g_clk);
      CHK(hipDeviceSynchronize());
      unsigned long long hc[2]; CHK(hipMemcpy(hc,g_clk,16,hipMemcpyDeviceToHost));
      printf("sustained mad64+3add32, 8 waves/SIMD, after %d launches: %.0f MHz\n",80*(rep+1),(double)hc[0]/((double)hc[1]/100.0));
    }
    return 0;
  }
  for(int w: {1,2,4,8}){
  run<0>("v_mad_u64_u32",32,d,w); run<1>("v_lshl_add_u64",32,d,w); run<2>("v_add_u32",32,d,w); run<3>("v_mul_lo_u32",32,d,w);
  run<4>("v_mul_hi_u32",32,d,w); run<5>("v_mad_u32_u24",32,d,w); run<6>("v_fma_f64",32,d,w); run<7>("v_lshrrev_b64",32,d,w);
  run<8>("add_co+addc_co (pair)",32,d,w); run<9>("v_mov_b32",32,d,w); run<10>("v_mul_hi_u32_u24",32,d,w); run<11>("v_alignbit_b32",32,d,w);
  run<12>("mad64+3add32 (4 instr)",32,d,w);
  }
  return 0;
}
