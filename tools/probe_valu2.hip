// GPU probe (not product code): VALU issue rates on gfx950 that bound the selective-scan kernels, with the effective
// shader clock measured inside the kernel (s_memtime ticks against the 100 MHz s_memrealtime counter).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_valu2.hip -o /tmp/probe_valu2 && /tmp/probe_valu2
// Every op is an asm volatile so that the compiler neither packs, fuses nor reorders them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

#define FMA(a, s)   asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(s))
#define MUL(a, s)   asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(s))
#define EXP(a)      asm volatile("v_exp_f32 %0, %0" : "+v"(a))
#define PKFMA(p, q) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p) : "v"(q))
#define PKMUL(p, q) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(q))

template <int MODE> __global__ void __launch_bounds__(256) k(float* out, unsigned long long* clk, int iters, float s) {
    float a[16];
    f2 p[8], q2 = {s, s};
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 1e-3f + i * 0.01f;
    for (int i = 0; i < 8; i++) p[i] = f2{a[2 * i], a[2 * i + 1]};
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) FMA(a[i], s);
        }
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; i++) PKFMA(p[i], q2);
        }
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; i++) PKMUL(p[i], q2);
        }
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; i++) EXP(a[i]);
        }
        if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 16; i++) MUL(a[i], s);
        }
        if (MODE == 5) {          // one scan step of the apply kernel, packed: 8 pk_mul, 16 exp, 8 pk_mul, 8 pk_fma, 8 pk_fma
#pragma unroll
            for (int i = 0; i < 8; i++) {
                PKMUL(p[i], q2);
                EXP(a[2 * i]); EXP(a[2 * i + 1]);
                PKMUL(p[i], q2);
                PKFMA(p[i], q2);
                PKFMA(p[i], q2);
            }
        }
        if (MODE == 6) {          // the same step with plain ops: 16 mul, 16 exp, 16 mul, 16 fma, 16 fma
#pragma unroll
            for (int i = 0; i < 16; i++) {
                MUL(a[i], s);
                EXP(a[i]);
                MUL(a[i], s);
                FMA(a[i], s);
                FMA(a[i], s);
            }
        }
        if (MODE == 7) {          // exp and fma alternating on independent registers (does anything overlap inside one wave?)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                EXP(a[i]);
                FMA(a[8 + i], s);
            }
        }
        if (MODE == 8) {          // 1 exp : 3 plain, on independent registers
#pragma unroll
            for (int i = 0; i < 4; i++) {
                EXP(a[i]);
                FMA(a[4 + 3 * i], s); FMA(a[5 + 3 * i], s); FMA(a[6 + 3 * i], s);
            }
        }
    }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    float r = 0;
    for (int i = 0; i < 16; i++) r += a[i];
    for (int i = 0; i < 8; i++) r += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int MODE> int bench(const char* name, int instr_per_iter, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 4000;
    float* d; unsigned long long* c;
    CHK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    CHK(hipMalloc(&c, (size_t)blocks * 16));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k<MODE><<<blocks, 256>>>(d, c, 100, 0.999f); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0)); k<MODE><<<blocks, 256>>>(d, c, iters, 0.999f); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * blocks);
    CHK(hipMemcpy(h.data(), c, (size_t)blocks * 16, hipMemcpyDeviceToHost));
    double sc = 0, sw = 0;
    for (int i = 0; i < blocks; i++) { sc += (double)h[2 * i]; sw += (double)h[2 * i + 1]; }
    const double ghz = sc / sw * 0.1;                       // s_memrealtime ticks at 100 MHz
    const double wave_cyc = sc / blocks;                    // shader cycles one wave was inside the loop
    const double per_instr_wave = wave_cyc / ((double)iters * instr_per_iter);          // latency-ish view, one wave
    const double per_instr_simd = per_instr_wave / waves_per_simd;                      // throughput view per SIMD
    printf("%-34s w/SIMD=%d  %7.3f ms  clock %.2f GHz  cyc/instr: per-wave %6.2f  per-SIMD %5.2f\n", name, waves_per_simd, ms, ghz,
           per_instr_wave, per_instr_simd);
    CHK(hipFree(d)); CHK(hipFree(c)); return 0;
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    const int ws[] = {1, 2, 3, 4, 8};
    for (int w : ws) bench<0>("v_fma_f32 x16", 16, w);
    for (int w : ws) bench<4>("v_mul_f32 x16", 16, w);
    for (int w : ws) bench<1>("v_pk_fma_f32 x8", 8, w);
    for (int w : ws) bench<2>("v_pk_mul_f32 x8", 8, w);
    for (int w : ws) bench<3>("v_exp_f32 x16", 16, w);
    for (int w : ws) bench<7>("exp,fma alternating x8+8", 16, w);
    for (int w : ws) bench<8>("1 exp : 3 fma  x4+12", 16, w);
    for (int w : ws) bench<5>("scan step packed (16 exp + 32 pk)", 48, w);
    for (int w : ws) bench<6>("scan step plain (16 exp + 64 valu)", 80, w);
    return 0;
}
