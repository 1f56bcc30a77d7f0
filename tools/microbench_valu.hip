// Microbenchmark: integer-VALU issue rates on gfx950 (MI355X).
// Measures the throughput of v_mad_u64_u32 / v_mul_lo_u32 / v_mul_hi_u32 / v_addc / v_lshl_add_u64 /
// v_fma_f64 with 8 independent dependency chains per lane at full occupancy.  The v_mad_u64_u32 figure
// is the "peak_MAC32/s" roofline denominator used by bench.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32; typedef uint64_t u64;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template<int OP> __global__ void __launch_bounds__(256) k_rate(u32* out, const u32* in, int iters){
  int tid=blockIdx.x*blockDim.x+threadIdx.x;
  u32 x=in[tid&1023], y=in[(tid+7)&1023]|1;
  u64 a0=x,a1=x+1,a2=x+2,a3=x+3,a4=x+4,a5=x+5,a6=x+6,a7=x+7;
  u32 t0=1,t1=2,t2=3,t3=4,t4=5,t5=6,t6=7,t7=8;
  double d0=x,d1=x+1,d2=x+2,d3=x+3,d4=x+4,d5=x+5,d6=x+6,d7=x+7; double dy=(double)y*1e-9;
  for(int it=0;it<iters;it++){
    #pragma unroll
    for(int u=0;u<8;u++){
      if(OP==0){
#define M(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a##i) : "v"(x), "v"(y) : "vcc");
        REP8(M)
#undef M
      } else if(OP==1){
#define M(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(t##i) : "v"(y));
        REP8(M)
#undef M
      } else if(OP==2){
#define M(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(t##i) : "v"(y));
        REP8(M)
#undef M
      } else if(OP==3){
#define M(i) asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(t##i) : : "vcc");
        REP8(M)
#undef M
      } else if(OP==4){
#define M(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a##i) : "v"(a0));
        REP8(M)
#undef M
      } else if(OP==5){
#define M(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d##i) : "v"(dy));
        REP8(M)
#undef M
      } else if(OP==6){ // mad + addc pair (the MAC primitive of fp_mul)
#define M(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(a##i), "+v"(t##i) : "v"(x), "v"(y) : "vcc");
        REP8(M)
#undef M
      } else if(OP==7){
#define M(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(t##i) : "v"(y));
        REP8(M)
#undef M
      } else if(OP==8){
#define M(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(t##i) : "v"(y));
        REP8(M)
#undef M
      } else if(OP==9){ // mad with SGPR carry-out to a non-VCC pair
#define M(i) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(a##i) : "v"(x), "v"(y) : "s20","s21");
        REP8(M)
#undef M
      }
    }
  }
  u64 s=a0+a1+a2+a3+a4+a5+a6+a7+t0+t1+t2+t3+t4+t5+t6+t7; double ds=d0+d1+d2+d3+d4+d5+d6+d7;
  out[tid]=(u32)s^(u32)(s>>32)^(u32)ds;
}

template<int OP> int run(const char* name, int waves_per_simd, int ops_per_inst){
  u32 *in,*out; int nblk=256*waves_per_simd; // 256 CUs x (4 SIMD x wps waves) / 4 waves per block
  CK(hipMalloc(&in,4096)); CK(hipMalloc(&out,(size_t)nblk*256*4)); CK(hipMemset(in,0x5a,4096));
  int iters=2000; hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_rate<OP>,dim3(nblk),dim3(256),0,0,out,in,10); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_rate<OP>,dim3(nblk),dim3(256),0,0,out,in,iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  double insts=(double)nblk*256*iters*64.0*ops_per_inst; // per-lane instruction executions
  double rate=insts/(ms*1e-3);
  // cycles per wave-instruction per SIMD at 2.4 GHz
  double wave_insts_per_simd=(double)waves_per_simd*iters*64.0*ops_per_inst;
  double cyc=(ms*1e-3*2.4e9)/wave_insts_per_simd;
  printf("{\"op\":\"%s\",\"waves_per_simd\":%d,\"ms\":%.4f,\"lane_ops_per_s\":%.4e,\"cyc_per_wave_inst_at_2.4GHz\":%.3f}\n",name,waves_per_simd,ms,rate,cyc);
  CK(hipFree(in)); CK(hipFree(out)); return 0;
}
int main(){
  for(int w=1;w<=8;w*=2){
    run<0>("v_mad_u64_u32",w,1); run<9>("v_mad_u64_u32_sgprcarry",w,1); run<1>("v_mul_lo_u32",w,1); run<2>("v_mul_hi_u32",w,1);
    run<3>("v_addc_co_u32",w,1); run<4>("v_lshl_add_u64",w,1); run<5>("v_fma_f64",w,1); run<6>("mad+addc",w,2); run<7>("v_add_u32",w,1); run<8>("v_mul_u32_u24",w,1);
  }
  return 0;
}
