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

template<int OP> __global__ void k(u32* out, u32 seed){
  u32 t=threadIdx.x+blockIdx.x*blockDim.x;
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
}
template<int OP> int run(const char* name, int ops_per_iter, u32* d, int wavesPerSimd){
  hipEvent_t e0,e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  int threads=256, blocks=256*wavesPerSimd;
  k<OP><<<blocks,threads>>>(d,1); CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0)); for(int r=0;r<4;r++) k<OP><<<blocks,threads>>>(d,2+r); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms,e0,e1)); ms/=4;
  double waveinstr=(double)blocks*threads/64*ITERS*ops_per_iter;
  printf("%-28s waves/SIMD=%d %8.3f ms  %.3e wave-instr/s  %.2f cyc/wave-instr/SIMD @2.4GHz\n",name,wavesPerSimd,ms,waveinstr/(ms*1e-3), 2.4e9*1024/(waveinstr/(ms*1e-3)));
  return 0;
}
int main(){
  u32* d; CHK(hipMalloc(&d, 256*8*256*4));
  for(int w: {1,2,4,8}){
  run<0>("v_mad_u64_u32",32,d,w); run<1>("v_lshl_add_u64",32,d,w); run<2>("v_add_u32",32,d,w); run<3>("v_mul_lo_u32",32,d,w);
  run<4>("v_mul_hi_u32",32,d,w); run<5>("v_mad_u32_u24",32,d,w); run<6>("v_fma_f64",32,d,w); run<7>("v_lshrrev_b64",32,d,w);
  run<8>("add_co+addc_co (pair)",32,d,w); run<9>("v_mov_b32",32,d,w); run<10>("v_mul_hi_u32_u24",32,d,w); run<11>("v_alignbit_b32",32,d,w);
  run<12>("mad64+3add32 (4 instr)",32,d,w);
  }
  return 0;
}
