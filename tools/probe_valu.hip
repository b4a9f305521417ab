// GPU probe (not product code): per-instruction VALU rates on gfx950 that decide the scan kernel design.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
template<int MODE> __global__ void __launch_bounds__(256) k(float* out, int iters, float s) {
  float a[16]; float2 p[8];
  for (int i=0;i<16;i++) a[i] = threadIdx.x*1e-3f + i;
  for (int i=0;i<8;i++) p[i] = make_float2(a[2*i], a[2*i+1]);
  for (int it=0; it<iters; ++it) {
    #pragma unroll
    for (int i=0;i<16;i++) {
      if (MODE==0) a[i] = fmaf(a[i], s, 1.0f);                    // plain fma
      if (MODE==1) a[i] = __builtin_amdgcn_exp2f(a[i]*s);         // mul + v_exp_f32
      if (MODE==3) a[i] = __builtin_amdgcn_exp2f(a[i]);           // v_exp_f32 only
      if (MODE==4) a[i] = __builtin_amdgcn_rcpf(a[i]);            // v_rcp_f32
      if (MODE==5) a[i] = __builtin_amdgcn_logf(a[i]);            // v_log_f32
    }
    if (MODE==2) {
      #pragma unroll
      for (int i=0;i<8;i++) { p[i].x = fmaf(p[i].x, s, 1.0f); p[i].y = fmaf(p[i].y, s, 1.0f);}  // SLP -> v_pk_fma_f32?
    }
  }
  float r=0; for (int i=0;i<16;i++) r+=a[i]; for (int i=0;i<8;i++) r+=p[i].x+p[i].y;
  out[blockIdx.x*blockDim.x+threadIdx.x]=r;
}
template<int MODE> int bench(const char* name, double ops_per_iter_per_thread) {
  float* d; int blocks=256*8, iters=20000; CHK(hipMalloc(&d, blocks*256*4));
  hipEvent_t e0,e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  k<MODE><<<blocks,256>>>(d, 100, 0.999f); CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0)); k<MODE><<<blocks,256>>>(d, iters, 0.999f); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms,e0,e1));
  double lane_ops = (double)blocks*256*iters*ops_per_iter_per_thread;
  printf("%-28s %8.3f ms  %8.2f T lane-instr/s\n", name, ms, lane_ops/ms*1e-9);
  CHK(hipFree(d)); return 0;
}
int main(){
  hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p,0)); printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  bench<0>("v_fma_f32 x16", 16); bench<2>("fma pairs (pk?) 16 lanes-ops", 16);
  bench<1>("mul+exp2 x16 (pairs)", 16); bench<3>("exp2 x16", 16); bench<4>("rcp x16", 16); bench<5>("log2 x16", 16);
  return 0;
}
