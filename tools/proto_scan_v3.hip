// GPU probe (not product code): third prototype of the wave-uniform scan pass (see proto_scan_u.hip / proto_scan_mem.hip).
// Rolling prefetch: each step consumes the row elements loaded PD steps ago and re-issues the loads for step t + PD into the
// same registers, so loads, stores and arithmetic of one wave overlap without a second register buffer; B_t / C_t scalar
// loads run one step ahead.  Prints time per launch for PD in {2, 4, 8} and the aggregate-only pass.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast tools/proto_scan_v3.hip -o build/proto_scan_v3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CAS __attribute__((address_space(4)))
#define PINV(x) asm volatile("" : "+v"(x))
#define PINS(x) asm volatile("" : "+s"(x))
#define PINS16(x) asm volatile("" : "+s"(x))
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float bf(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ uint16_t tobf(float f) { return (uint16_t)(__float_as_uint(f) >> 16); }

struct Args {
    uint16_t* u; uint16_t* dl; uint16_t* z; uint16_t* oz; uint16_t* out;
    const float* BC;                          // (rows, 32) fp32: B_t (16) then C_t (16)
    const float* A; const float* carry; float* agg;
    int64_t st, rs;
    int T, dim;
};

template <int PD, bool AGG, int MINB, int ABL = 0, bool FUSE = false> __global__ void __launch_bounds__(256, MINB) scan_v3(Args P) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * 4 + wave;
    const int d = lane;
    const int64_t row0 = (int64_t)item * P.T;
    f2 A2[8], h[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        A2[n] = f2{P.A[d * 16 + 2 * n], P.A[d * 16 + 2 * n + 1]};
        h[n] = AGG ? f2{0.f, 0.f}
                   : f2{P.carry[((int64_t)item * 16 + 2 * n) * P.dim + d], P.carry[((int64_t)item * 16 + 2 * n + 1) * P.dim + d]};
    }
    const uint16_t* up = P.u + row0 * P.st + d;
    const uint16_t* dp = P.dl + row0 * P.st + d;
    const uint16_t* zp = P.z + row0 * P.st + d;
    uint16_t* op = P.oz + row0 * P.st + d;
    uint16_t* yp = P.out + row0 * P.st + d;
    const CAS f32x16* Bp = (const CAS f32x16*)(uintptr_t)(P.BC + row0 * 32);
    const int st = (int)P.st;
    float hx[3] = {0.f, 0.f, 0.f};
    const float cw4[4] = {P.A[d * 16 + 0] * 0.1f, P.A[d * 16 + 1] * 0.1f, P.A[d * 16 + 2] * 0.1f, P.A[d * 16 + 3] * 0.1f};
    const float cbias = P.A[d * 16 + 4] * 0.01f;
    uint32_t ub[PD], db[PD], zb[PD];
#pragma unroll
    for (int j = 0; j < PD; ++j) { ub[j] = up[j * st]; db[j] = dp[j * st]; if (!AGG) zb[j] = zp[j * st]; }
    // scalar B_t / C_t rows, ping-pong between two register sets; the loads are inline asm so that the delivery wait sits
    // where we put it (scalar loads return out of order: every wait is lgkmcnt(0), so take delivery of this step's rows
    // first and only then put the next step's loads in flight)
    f32x16 bs[2], cs[2];
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(bs[0]) : "s"(Bp));
    if (!AGG) asm volatile("s_load_dwordx16 %0, %1, 0x40" : "=s"(cs[0]) : "s"(Bp));
    float sumd = 0.f;
    const int T = P.T;
#pragma unroll 1
    for (int t0 = 0; t0 < T; t0 += PD) {
#pragma unroll
        for (int j = 0; j < PD; ++j) {
            const int t = t0 + j;
            if (AGG) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(bs[j & 1]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(bs[j & 1]), "+s"(cs[j & 1]));
            {
                const int tn = t + 1 < T ? t + 1 : t;
                const CAS f32x16* nx = Bp + tn * 2;
                if (!(ABL & 2)) {
                    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(bs[(j + 1) & 1]) : "s"(nx));
                    if (!AGG) asm volatile("s_load_dwordx16 %0, %1, 0x40" : "=s"(cs[(j + 1) & 1]) : "s"(nx));
                } else { bs[(j + 1) & 1] = bs[j & 1]; cs[(j + 1) & 1] = cs[j & 1]; }
            }
            const f32x16 bw = bs[j & 1];
            f32x16 cw = bs[j & 1];
            if (!AGG) cw = cs[j & 1];
            float uu = bf(ub[j]);
            if (FUSE) {                                    // the "conv1d fused into the scan launch" variant: 4 taps + bias + SiLU per step
                const float xc = uu;
                float o = cw4[3] * xc + cw4[2] * hx[0] + cw4[1] * hx[1] + cw4[0] * hx[2] + cbias;
                hx[2] = hx[1]; hx[1] = hx[0]; hx[0] = xc;
                uu = o * __builtin_amdgcn_rcpf(1.f + fexp2(-o * 1.4426950408889634f));
            }
            float dl = bf(db[j]);
            float zz = 0.f;
            if (!AGG) zz = bf(zb[j]);
            {   // re-issue this slot's loads for step t + PD (clamped inside the chunk)
                const int tp = t + PD < T ? t + PD : t;
                if (!(ABL & 1)) {
                    ub[j] = up[tp * st]; db[j] = dp[tp * st];
                    if (!AGG) zb[j] = zp[tp * st];
                } else { ub[j] += 0x10000; db[j] ^= 0x20000; }
            }
            {
                const float e = fexp2(dl * 1.4426950408889634f);
                const float sp = __builtin_amdgcn_logf(1.0f + e) * 0.6931471805599453f;
                dl = dl > 20.f ? dl : sp;
            }
            const float dlu = dl * uu;
            sumd += dl;
            f2 ya = {0.f, 0.f}, yb = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f2 b0 = {bw[4 * q], bw[4 * q + 1]}, b1 = {bw[4 * q + 2], bw[4 * q + 3]};
                const f2 da0 = A2[2 * q] * dl, da1 = A2[2 * q + 1] * dl;
                const f2 a0 = {fexp2(da0.x), fexp2(da0.y)}, a1 = {fexp2(da1.x), fexp2(da1.y)};
                h[2 * q] = a0 * h[2 * q] + b0 * dlu;
                h[2 * q + 1] = a1 * h[2 * q + 1] + b1 * dlu;
                if (!AGG) {
                    const f2 c0 = {cw[4 * q], cw[4 * q + 1]}, c1 = {cw[4 * q + 2], cw[4 * q + 3]};
                    ya = c0 * h[2 * q] + ya;
                    yb = c1 * h[2 * q + 1] + yb;
                }
                if (q == 1) {                             // keeps the two halves of the state update apart (fewer live temporaries)
                    PINV(h[0]); PINV(h[1]); PINV(h[2]); PINV(h[3]);
                }
            }
            if (!AGG) {
                const float y = (ya.x + yb.x) + (ya.y + yb.y);
                const float sg = __builtin_amdgcn_rcpf(1.f + fexp2(-zz * 1.4426950408889634f));
                if (!(ABL & 1)) { yp[t * st] = tobf(y); op[t * st] = tobf(y * zz * sg); }
                else sumd += y * zz * sg;
            }
#pragma unroll
            for (int n = 0; n < 8; ++n) PINV(h[n]);
        }
    }
    if (!AGG && (ABL & 1) && sumd == 123.f) P.out[0] = 1;
    if (AGG) {
        P.agg[((int64_t)item * 17 + 16) * P.dim + d] = sumd;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            P.agg[((int64_t)item * 17 + 2 * n) * P.dim + d] = h[n].x;
            P.agg[((int64_t)item * 17 + 2 * n + 1) * P.dim + d] = h[n].y;
        }
    }
}

template <int PD, bool AGG, int MINB, int ABL = 0, bool FUSE = false> static float run(const Args& P, int64_t rows, int reps) {
    const int nblk = (int)(rows / P.T / 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((scan_v3<PD, AGG, MINB, ABL, FUSE>), dim3(nblk), dim3(256), 0, 0, P);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((scan_v3<PD, AGG, MINB, ABL, FUSE>), dim3(nblk), dim3(256), 0, 0, P);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
    const int64_t rows = 2 * 262144;
    const int dim = 64;
    const int64_t st = 192, rs = 48;
    std::vector<uint16_t> hs((size_t)rows * st); std::vector<float> hb((size_t)rows * 32);
    auto tb = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return (uint16_t)(u >> 16); };
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f; };
    for (auto& v : hs) v = tb(rnd() - 0.5f);
    for (auto& v : hb) v = rnd() - 0.5f;
    uint16_t *u, *dl, *z, *oz, *out; float* B;
    CHK(hipMalloc(&u, hs.size() * 2)); CHK(hipMalloc(&dl, hs.size() * 2)); CHK(hipMalloc(&z, hs.size() * 2));
    CHK(hipMalloc(&oz, hs.size() * 2)); CHK(hipMalloc(&out, hs.size() * 2)); CHK(hipMalloc(&B, hb.size() * 4));
    CHK(hipMemcpy(u, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dl, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    CHK(hipMemcpy(z, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    CHK(hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hA(dim * 16);
    for (int d = 0; d < dim; ++d) for (int n = 0; n < 16; ++n) hA[d * 16 + n] = -(n + 1) * 1.4426950408889634f;
    float *A, *carry, *agg;
    CHK(hipMalloc(&A, hA.size() * 4)); CHK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    const size_t wsn = (size_t)(rows / 32) * 17 * dim;
    CHK(hipMalloc(&carry, wsn * 4)); CHK(hipMemset(carry, 0, wsn * 4));
    CHK(hipMalloc(&agg, wsn * 4));
    const int which = argc > 1 ? atoi(argv[1]) : -1;
    printf("rows %lld x %d channels (%.1f M elements), bf16; us per launch; variant %d\n", (long long)rows, dim, rows * dim / 1e6, which);
    fflush(stdout);
    for (int T : {64, 128}) {
        Args P{u, dl, z, oz, out, B, A, carry, agg, st, rs, T, dim};
        float r = 0.f; const char* nm = "";
        switch (which) {
            case 0: r = run<2, true, 2>(P, rows, 20); nm = "aggregate PD2"; break;
            case 1: r = run<4, true, 2>(P, rows, 20); nm = "aggregate PD4"; break;
            case 2: r = run<8, true, 2>(P, rows, 20); nm = "aggregate PD8"; break;
            case 3: r = run<2, false, 2>(P, rows, 20); nm = "apply PD2 free regs"; break;
            case 4: r = run<4, false, 2>(P, rows, 20); nm = "apply PD4 free regs"; break;
            case 5: r = run<8, false, 2>(P, rows, 20); nm = "apply PD8 free regs"; break;
            case 6: r = run<2, false, 6>(P, rows, 20); nm = "apply PD2 <= 80 regs"; break;
            case 7: r = run<4, false, 6>(P, rows, 20); nm = "apply PD4 <= 80 regs"; break;
            case 8: r = run<4, true, 2, 1>(P, rows, 20); nm = "aggregate PD4 no vmem"; break;
            case 9: r = run<4, true, 2, 2>(P, rows, 20); nm = "aggregate PD4 no smem"; break;
            case 10: r = run<4, true, 2, 3>(P, rows, 20); nm = "aggregate PD4 valu only"; break;
            case 11: r = run<4, false, 2, 1>(P, rows, 20); nm = "apply PD4 no vmem"; break;
            case 12: r = run<4, false, 2, 2>(P, rows, 20); nm = "apply PD4 no smem"; break;
            case 13: r = run<4, false, 2, 3>(P, rows, 20); nm = "apply PD4 valu only"; break;
            case 20: r = run<4, true, 2, 0, true>(P, rows, 20); nm = "aggregate PD4 + conv1d/SiLU"; break;
            case 21: r = run<4, false, 2, 0, true>(P, rows, 20); nm = "apply PD4 + conv1d/SiLU"; break;
            default: break;
        }
        CHK(hipDeviceSynchronize());
        printf("T=%3d  %-24s %6.1f us\n", T, nm, r);
        fflush(stdout);
    }
    return 0;
}
