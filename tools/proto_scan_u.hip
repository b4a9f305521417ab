// GPU probe (not product code): a selective-scan "apply" / "aggregate" pass with wave-uniform time, i.e. every lane of a
// wave is a channel of the SAME (batch, chunk) work item, so B_t / C_t are scalar loads into SGPRs and enter the
// packed-fp32 state update as scalar operands (no LDS, no per-lane staging).  Prints time per launch for a stage-0 sized
// problem restricted to 64 channels, with ablations (no transcendental / no state math) to see what bounds it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast tools/proto_scan_u.hip -o build/proto_scan_u
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <vector>
#include <cstring>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
#define CAS __attribute__((address_space(4)))
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

struct Args {
    const uint16_t* u; const uint16_t* dl; const uint16_t* z; uint16_t* oz; uint16_t* out;
    const uint16_t* B; const uint16_t* C;     // (rows, 16) bf16, row stride rs elements
    const float* A; const float* carry; float* agg;
    int64_t st;       // row stride of u/dl/z/oz in elements
    int64_t rs;       // row stride of B/C in elements
    int T;            // steps per chunk
    int dim;
};

// MODE 0 apply, 1 apply without v_exp (a = x), 2 apply with exps only (no h/y math), 3 aggregate
template <int MODE> __global__ void __launch_bounds__(256) scan_u(Args P) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * 4 + wave;
    const int d = lane;
    const int64_t row0 = (int64_t)item * P.T;
    f2 A2[8], h[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        A2[n] = f2{P.A[d * 16 + 2 * n], P.A[d * 16 + 2 * n + 1]};
        h[n] = MODE == 3 ? f2{0.f, 0.f}
                         : f2{P.carry[((int64_t)item * 16 + 2 * n) * P.dim + d], P.carry[((int64_t)item * 16 + 2 * n + 1) * P.dim + d]};
    }
    const uint16_t* up = P.u + row0 * P.st;
    const uint16_t* dp = P.dl + row0 * P.st;
    const uint16_t* zp = P.z + row0 * P.st;
    uint16_t* op = P.oz + row0 * P.st;
    uint16_t* yp = P.out + row0 * P.st;
    const CAS u32x8* Bp = (const CAS u32x8*)(uintptr_t)(P.B + row0 * P.rs);
    const CAS u32x8* Cp = (const CAS u32x8*)(uintptr_t)(P.C + row0 * P.rs);
    const int64_t rs8 = P.rs * 2 / 32;
    u32x8 bn = Bp[0], cn = Cp[0];
    float sumd = 0.f;
    for (int t = 0; t < P.T; ++t) {
        const u32x8 bw = bn, cw = cn;
        const int tn = t + 1 < P.T ? t + 1 : t;
        bn = Bp[(int64_t)tn * rs8];
        if (MODE != 3) cn = Cp[(int64_t)tn * rs8];
        const float uu = __uint_as_float((uint32_t)up[(int64_t)t * P.st + d] << 16);
        float dl = __uint_as_float((uint32_t)dp[(int64_t)t * P.st + d] << 16);
        {   // softplus
            const float e = fexp2(dl * 1.4426950408889634f);
            const float sp = __builtin_amdgcn_logf(1.0f + e) * 0.6931471805599453f;
            dl = dl > 20.f ? dl : sp;
        }
        const float dlu = dl * uu;
        sumd += dl;
        f2 ya = {0.f, 0.f}, yb = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f2 b0 = {bf_lo(bw[2 * q]), bf_hi(bw[2 * q])}, b1 = {bf_lo(bw[2 * q + 1]), bf_hi(bw[2 * q + 1])};
            const f2 c0 = {bf_lo(cw[2 * q]), bf_hi(cw[2 * q])}, c1 = {bf_lo(cw[2 * q + 1]), bf_hi(cw[2 * q + 1])};
            const f2 da0 = A2[2 * q] * dl, da1 = A2[2 * q + 1] * dl;
            f2 a0, a1;
            if (MODE == 1) { a0 = da0; a1 = da1; }
            else { a0 = f2{fexp2(da0.x), fexp2(da0.y)}; a1 = f2{fexp2(da1.x), fexp2(da1.y)}; }
            if (MODE == 2) { h[2 * q] += a0; h[2 * q + 1] += a1; }
            else {
                h[2 * q] = a0 * h[2 * q] + b0 * dlu;
                h[2 * q + 1] = a1 * h[2 * q + 1] + b1 * dlu;
            }
            if (MODE != 3 && MODE != 2) {
                ya = c0 * h[2 * q] + ya;
                yb = c1 * h[2 * q + 1] + yb;
            }
        }
        if (MODE != 3) {
            const float zz = __uint_as_float((uint32_t)zp[(int64_t)t * P.st + d] << 16);
            float y = (ya.x + yb.x) + (ya.y + yb.y);
            if (MODE == 2) y = h[0].x + h[7].y;
            const float sg = __builtin_amdgcn_rcpf(1.f + fexp2(-zz * 1.4426950408889634f));
            yp[(int64_t)t * P.st + d] = (uint16_t)(__float_as_uint(y) >> 16);
            op[(int64_t)t * P.st + d] = (uint16_t)(__float_as_uint(y * zz * sg) >> 16);
        }
    }
    if (MODE == 3) {
        P.agg[((int64_t)item * 17 + 16) * P.dim + d] = sumd;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            P.agg[((int64_t)item * 17 + 2 * n) * P.dim + d] = h[n].x;
            P.agg[((int64_t)item * 17 + 2 * n + 1) * P.dim + d] = h[n].y;
        }
    }
}

template <int MODE> static float run(const Args& P, int64_t rows, int reps) {
    const int nblk = (int)(rows / P.T / 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(scan_u<MODE>, dim3(nblk), dim3(256), 0, 0, P);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(scan_u<MODE>, dim3(nblk), dim3(256), 0, 0, P);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}

int main() {
    const int64_t rows = 2 * 262144;           // batch 2 x L = 64^3, flattened
    const int dim = 64;
    const int64_t st = 192, rs = 48;           // xz-like row (2 x 96 channels), padded x_dbl row (48 columns: 96 bytes)
    std::vector<uint16_t> hs((size_t)rows * st), hb((size_t)rows * rs);
    auto bf = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return (uint16_t)(u >> 16); };
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f; };
    for (auto& v : hs) v = bf(rnd() - 0.5f);
    for (auto& v : hb) v = bf(rnd() - 0.5f);
    uint16_t *u, *dl, *z, *oz, *out, *B;
    CHK(hipMalloc(&u, hs.size() * 2)); CHK(hipMalloc(&dl, hs.size() * 2)); CHK(hipMalloc(&z, hs.size() * 2));
    CHK(hipMalloc(&oz, hs.size() * 2)); CHK(hipMalloc(&out, hs.size() * 2)); CHK(hipMalloc(&B, hb.size() * 2));
    CHK(hipMemcpy(u, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dl, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    CHK(hipMemcpy(z, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    CHK(hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> hA(dim * 16);
    for (int d = 0; d < dim; ++d) for (int n = 0; n < 16; ++n) hA[d * 16 + n] = -(n + 1) * 1.4426950408889634f;
    float *A, *carry, *agg;
    CHK(hipMalloc(&A, hA.size() * 4)); CHK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    const size_t wsn = (size_t)(rows / 32) * 17 * dim;
    CHK(hipMalloc(&carry, wsn * 4)); CHK(hipMemset(carry, 0, wsn * 4));
    CHK(hipMalloc(&agg, wsn * 4));
    printf("rows %lld x %d channels (%.1f M elements), bf16 I/O; cycles = per wave-step per SIMD at 2.4 GHz\n", (long long)rows, dim,
           rows * dim / 1e6);
    for (int T : {32, 64, 128, 256, 512}) {
        Args P{u, dl, z, oz, out, B + 4, B + 20, A, carry, agg, st, rs, T, dim};
        const double wavesteps_per_simd = (double)rows / 1024.0;
        float t0 = run<0>(P, rows, 20), t1 = run<1>(P, rows, 20), t2 = run<2>(P, rows, 20), t3 = run<3>(P, rows, 20);
        printf("T=%3d waves=%6lld  apply %7.1f us (%5.0f cyc)  apply-no-exp %7.1f us (%5.0f)  exps-only %7.1f us (%5.0f)  agg %7.1f us (%5.0f)\n",
               T, (long long)(rows / T), t0, t0 * 2400.0 / wavesteps_per_simd, t1, t1 * 2400.0 / wavesteps_per_simd, t2,
               t2 * 2400.0 / wavesteps_per_simd, t3, t3 * 2400.0 / wavesteps_per_simd);
    }
    return 0;
}
