// GPU probe (not product code): does the cost of a packed-fp32 / fp32 multiply-add depend on how many DISTINCT source registers it reads
// and on their VGPR banks?  tools/probe_valu2.hip measured `v_pk_fma_f32 d, d, q, d` (two distinct sources) at ~4.4 cycles per
// wave-instruction; the scan kernels issue `v_pk_fma_f32 d, a, b, d` with three distinct register pairs and take ~1.4x the time the
// rates of that probe predict (profiles/r03_scan_occupancy.log).  Explicit registers, 16 independent chains, 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_valu3.hip -o /tmp/probe_valu3 && /tmp/probe_valu3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// one instruction per chain c (destination pair v[64+2c : 65+2c]); A / B are the pairs named in the template string
#define CH(OP, c) OP(c)
#define X16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)
#define STR2(x) #x
#define STR(x) STR2(x)
#define D0(c) "v[" STR(64 + 2 * c) ":" STR(65 + 2 * c) "]"
#define D1(c) "v" STR(64 + 2 * c)

#define PKFMA_3(c)   "v_pk_fma_f32 v[%c0:%c1], v[2:3], v[4:5], v[%c0:%c1]\n"
// (the assembler wants literal register numbers: the bodies below are spelled out by the generator macro instead)

template <int MODE> __global__ void __launch_bounds__(256) k(float* out, unsigned long long* clk, int iters) {
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    // seed the registers the asm uses
    asm volatile(
        "v_mov_b32 v2, 1.0\n v_mov_b32 v3, 1.0\n v_mov_b32 v4, 0.5\n v_mov_b32 v5, 0.5\n v_mov_b32 v6, 0.5\n v_mov_b32 v7, 0.5\n"
        "v_mov_b32 v8, 0.25\n v_mov_b32 v9, 0.25\n v_mov_b32 v10, 0.25\n v_mov_b32 v11, 0.25\n"
        ::: "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11");
    for (int it = 0; it < iters; ++it) {
#define BODY(INS) asm volatile(INS ::: "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", \
    "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95")
#define REP16(F) F(64) F(66) F(68) F(70) F(72) F(74) F(76) F(78) F(80) F(82) F(84) F(86) F(88) F(90) F(92) F(94)
#define S(x) #x
#define P(d) "v[" S(d) ":" "%=" "]"
        if (MODE == 0) {   // pk_fma d, d, B, d      (two distinct source pairs: the round-2 probe)
#define F(d) "v_pk_fma_f32 v[" #d ":" #d "+1], v[" #d ":" #d "+1], v[4:5], v[" #d ":" #d "+1]\n"
            BODY(REP16(F));
#undef F
        }
        if (MODE == 1) {   // pk_fma d, A, B, d      (three distinct pairs; A on banks 2,3, B on banks 0,1)
#define F(d) "v_pk_fma_f32 v[" #d ":" #d "+1], v[2:3], v[4:5], v[" #d ":" #d "+1]\n"
            BODY(REP16(F));
#undef F
        }
        if (MODE == 2) {   // pk_fma d, A, B, d      (A and B on the same banks: v[2:3], v[6:7])
#define F(d) "v_pk_fma_f32 v[" #d ":" #d "+1], v[2:3], v[6:7], v[" #d ":" #d "+1]\n"
            BODY(REP16(F));
#undef F
        }
        if (MODE == 3) {   // pk_mul d, A, B         (no accumulator read)
#define F(d) "v_pk_mul_f32 v[" #d ":" #d "+1], v[2:3], v[4:5]\n"
            BODY(REP16(F));
#undef F
        }
        if (MODE == 4) {   // pk_mul d, d, B
#define F(d) "v_pk_mul_f32 v[" #d ":" #d "+1], v[" #d ":" #d "+1], v[4:5]\n"
            BODY(REP16(F));
#undef F
        }
        if (MODE == 5) {   // fma d, A, B, d  plain, three distinct registers
#define F(d) "v_fma_f32 v" #d ", v2, v4, v" #d "\n"
            BODY(REP16(F));
#undef F
        }
        if (MODE == 6) {   // fma d, d, B, d  plain
#define F(d) "v_fma_f32 v" #d ", v" #d ", v4, v" #d "\n"
            BODY(REP16(F));
#undef F
        }
        if (MODE == 7) {   // pk_fma d, A, B, d with op_sel_hi broadcast of A's low half (the scan's `x * scalar`)
#define F(d) "v_pk_fma_f32 v[" #d ":" #d "+1], v[2:3], v[4:5], v[" #d ":" #d "+1] op_sel_hi:[0,1,1]\n"
            BODY(REP16(F));
#undef F
        }
        if (MODE == 8) {   // pk_fma d, A, B, C    (four distinct pairs: y = c * h + y style with a separate addend)
#define F(d) "v_pk_fma_f32 v[" #d ":" #d "+1], v[2:3], v[4:5], v[8:9]\n"
            BODY(REP16(F));
#undef F
        }
        if (MODE == 9) {   // v_exp_f32 d, d
#define F(d) "v_exp_f32 v" #d ", v" #d "\n"
            BODY(REP16(F));
#undef F
        }
        if (MODE == 10) {  // the apply step of one state pair as the compiler emits it: pk_mul, 2 exp, pk_mul, pk_fma x2 (distinct sources)
#define F(d) "v_pk_mul_f32 v[" #d ":" #d "+1], v[2:3], v[4:5]\n v_exp_f32 v" #d ", v" #d "\n v_exp_f32 v[" #d "+1], v[" #d "+1]\n" \
             "v_pk_mul_f32 v[" #d ":" #d "+1], v[" #d ":" #d "+1], v[6:7]\n v_pk_fma_f32 v[" #d ":" #d "+1], v[8:9], v[10:11], v[" #d ":" #d "+1]\n" \
             "v_pk_fma_f32 v[" #d ":" #d "+1], v[2:3], v[" #d ":" #d "+1], v[10:11]\n"
            BODY(REP16(F));
#undef F
        }
    }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    float r;
    asm volatile("v_add_f32 %0, v64, v95" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int MODE> int bench(const char* name, int instr_per_iter, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 4000;
    float* d; unsigned long long* c;
    CHK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    CHK(hipMalloc(&c, (size_t)blocks * 16));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k<MODE><<<blocks, 256>>>(d, c, 100); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0)); k<MODE><<<blocks, 256>>>(d, c, iters); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * blocks);
    CHK(hipMemcpy(h.data(), c, (size_t)blocks * 16, hipMemcpyDeviceToHost));
    double sc = 0, sw = 0;
    for (int i = 0; i < blocks; i++) { sc += (double)h[2 * i]; sw += (double)h[2 * i + 1]; }
    const double ghz = sc / sw * 0.1;                       // s_memrealtime ticks at 100 MHz
    // wall-clock view: cycles of the SIMD per wave-instruction = launch time x clock / (instructions one SIMD issued)
    const double per_instr = ms * 1e-3 * ghz * 1e9 / ((double)waves_per_simd * iters * instr_per_iter);
    printf("%-58s w/SIMD=%d  %7.3f ms  clock %.2f GHz  SIMD cycles per wave-instruction %5.2f\n", name, waves_per_simd, ms, ghz, per_instr);
    CHK(hipFree(d)); CHK(hipFree(c)); return 0;
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    const int ws[] = {4, 8};
    for (int w : ws) {
        bench<0>("v_pk_fma d, d, B, d   (2 distinct pairs)", 16, w);
        bench<1>("v_pk_fma d, A, B, d   (3 distinct pairs, A/B other banks)", 16, w);
        bench<2>("v_pk_fma d, A, B, d   (3 distinct pairs, A/B same banks)", 16, w);
        bench<7>("v_pk_fma d, A, B, d   op_sel_hi:[0,1,1]", 16, w);
        bench<8>("v_pk_fma d, A, B, C   (4 distinct pairs)", 16, w);
        bench<3>("v_pk_mul d, A, B", 16, w);
        bench<4>("v_pk_mul d, d, B", 16, w);
        bench<5>("v_fma d, A, B, d      (plain, 3 distinct)", 16, w);
        bench<6>("v_fma d, d, B, d      (plain, 2 distinct)", 16, w);
        bench<9>("v_exp d, d", 16, w);
        bench<10>("apply step of a state pair (2 pk_mul, 2 exp, 2 pk_fma) x16", 96, w);
    }
    return 0;
}
