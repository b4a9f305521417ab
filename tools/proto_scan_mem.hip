// GPU probe (not product code): how the per-lane row streams of the scan kernels (u, delta, z in; out, out_z out; one
// 16-bit element per lane and step, rows 2*D elements apart) should be moved.  Same pass as tools/proto_scan_u.hip with four
// access patterns, each with and without the recurrence arithmetic:
//   0  one global_load_ushort per lane, tensor and step, consumed at once (no prefetch)
//   1  8-step sub-tiles, the next sub-tile's 24 loads in flight while the current one is computed
//   2  as 1 with raw buffer loads / stores (32-bit lane offset + scalar row offset)
//   3  cooperative: one 16-byte load per lane fetches an 8 step x 64 channel tile (8 rows of 128 B), transposed through a
//      wave-private LDS tile (ds_write_b128, ds_read_u16); outputs the same way back
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast tools/proto_scan_mem.hip -o build/proto_scan_mem
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CAS __attribute__((address_space(4)))
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float bf(uint16_t w) { return __uint_as_float((uint32_t)w << 16); }
__device__ __forceinline__ uint16_t tobf(float f) { return (uint16_t)(__float_as_uint(f) >> 16); }

struct Args {
    uint16_t* u; uint16_t* dl; uint16_t* z; uint16_t* oz; uint16_t* out;
    const uint16_t* B; const uint16_t* C;
    const float* A; const float* carry;
    int64_t st, rs;
    int T, dim;
};

struct State { f2 A2[8], h[8]; };

template <int MATH>
__device__ __forceinline__ void step(State& S, const u32x8& bw, const u32x8& cw, float uu, float dl, float zz, float& y, float& yz) {
    if (MATH == 0) { y = uu + dl; yz = zz + uu; return; }
    {
        const float e = fexp2(dl * 1.4426950408889634f);
        const float sp = __builtin_amdgcn_logf(1.0f + e) * 0.6931471805599453f;
        dl = dl > 20.f ? dl : sp;
    }
    const float dlu = dl * uu;
    f2 ya = {0.f, 0.f}, yb = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f2 b0 = {bf_lo(bw[2 * q]), bf_hi(bw[2 * q])}, b1 = {bf_lo(bw[2 * q + 1]), bf_hi(bw[2 * q + 1])};
        const f2 c0 = {bf_lo(cw[2 * q]), bf_hi(cw[2 * q])}, c1 = {bf_lo(cw[2 * q + 1]), bf_hi(cw[2 * q + 1])};
        const f2 da0 = S.A2[2 * q] * dl, da1 = S.A2[2 * q + 1] * dl;
        const f2 a0 = {fexp2(da0.x), fexp2(da0.y)}, a1 = {fexp2(da1.x), fexp2(da1.y)};
        S.h[2 * q] = a0 * S.h[2 * q] + b0 * dlu;
        S.h[2 * q + 1] = a1 * S.h[2 * q + 1] + b1 * dlu;
        ya = c0 * S.h[2 * q] + ya;
        yb = c1 * S.h[2 * q + 1] + yb;
    }
    y = (ya.x + yb.x) + (ya.y + yb.y);
    const float sg = __builtin_amdgcn_rcpf(1.f + fexp2(-zz * 1.4426950408889634f));
    yz = y * zz * sg;
}

template <int MODE, int MATH> __global__ void __launch_bounds__(256) scan_m(Args P) {
    __shared__ __attribute__((aligned(16))) uint16_t s_t[4][3][8 * 64];     // MODE 3: per wave, three 8 x 64 tiles
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * 4 + wave;
    const int d = lane;
    const int64_t row0 = (int64_t)item * P.T;
    State S;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        S.A2[n] = f2{P.A[d * 16 + 2 * n], P.A[d * 16 + 2 * n + 1]};
        S.h[n] = f2{P.carry[((int64_t)item * 16 + 2 * n) * P.dim + d], P.carry[((int64_t)item * 16 + 2 * n + 1) * P.dim + d]};
    }
    uint16_t* up = P.u + row0 * P.st;
    uint16_t* dp = P.dl + row0 * P.st;
    uint16_t* zp = P.z + row0 * P.st;
    uint16_t* op = P.oz + row0 * P.st;
    uint16_t* yp = P.out + row0 * P.st;
    const CAS u32x8* Bp = (const CAS u32x8*)(uintptr_t)(P.B + row0 * P.rs);
    const CAS u32x8* Cp = (const CAS u32x8*)(uintptr_t)(P.C + row0 * P.rs);
    const int64_t rs8 = P.rs * 2 / 32;
    const int st = (int)P.st;

    if (MODE == 0) {
        for (int t = 0; t < P.T; ++t) {
            u32x8 bw = {}, cw = {};
            if (MATH) { bw = Bp[(int64_t)t * rs8]; cw = Cp[(int64_t)t * rs8]; }
            float y, yz;
            step<MATH>(S, bw, cw, bf(up[t * st + d]), bf(dp[t * st + d]), bf(zp[t * st + d]), y, yz);
            yp[t * st + d] = tobf(y);
            op[t * st + d] = tobf(yz);
        }
    } else if (MODE == 1 || MODE == 2) {
        __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(up, 0, 0x7fffffff, 0x00020000);
        __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dp, 0, 0x7fffffff, 0x00020000);
        __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(zp, 0, 0x7fffffff, 0x00020000);
        __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(op, 0, 0x7fffffff, 0x00020000);
        __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(yp, 0, 0x7fffffff, 0x00020000);
        uint16_t nu[8], nd[8], nz[8];
        auto fetch = [&](int t0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (MODE == 1) {
                    nu[j] = up[(t0 + j) * st + d]; nd[j] = dp[(t0 + j) * st + d]; nz[j] = zp[(t0 + j) * st + d];
                } else {
                    const int so = (t0 + j) * st * 2;
                    nu[j] = __builtin_amdgcn_raw_buffer_load_b16(ru, d * 2, so, 0);
                    nd[j] = __builtin_amdgcn_raw_buffer_load_b16(rd, d * 2, so, 0);
                    nz[j] = __builtin_amdgcn_raw_buffer_load_b16(rz, d * 2, so, 0);
                }
            }
        };
        fetch(0);
        for (int t0 = 0; t0 < P.T; t0 += 8) {
            uint16_t cu[8], cd[8], cz[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { cu[j] = nu[j]; cd[j] = nd[j]; cz[j] = nz[j]; }
            fetch(t0 + 8 < P.T ? t0 + 8 : t0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                u32x8 bw = {}, cw = {};
                if (MATH) { bw = Bp[(int64_t)(t0 + j) * rs8]; cw = Cp[(int64_t)(t0 + j) * rs8]; }
                float y, yz;
                step<MATH>(S, bw, cw, bf(cu[j]), bf(cd[j]), bf(cz[j]), y, yz);
                if (MODE == 1) { yp[(t0 + j) * st + d] = tobf(y); op[(t0 + j) * st + d] = tobf(yz); }
                else {
                    const int so = (t0 + j) * st * 2;
                    __builtin_amdgcn_raw_buffer_store_b16(tobf(y), ry, d * 2, so, 0);
                    __builtin_amdgcn_raw_buffer_store_b16(tobf(yz), ro, d * 2, so, 0);
                }
            }
        }
    } else {
        // cooperative tiles: lane l moves 16 bytes = channels 8 (l % 8) .. + 7 of row l / 8
        const int tr = lane >> 3, tc = (lane & 7) * 8;
        uint16_t* lt = &s_t[wave][0][0];
        u32x4 nu, nd, nz;
        auto fetch = [&](int t0) {
            nu = *reinterpret_cast<const u32x4*>(up + (int64_t)(t0 + tr) * st + tc);
            nd = *reinterpret_cast<const u32x4*>(dp + (int64_t)(t0 + tr) * st + tc);
            nz = *reinterpret_cast<const u32x4*>(zp + (int64_t)(t0 + tr) * st + tc);
        };
        fetch(0);
        for (int t0 = 0; t0 < P.T; t0 += 8) {
            *reinterpret_cast<u32x4*>(lt + 0 * 512 + tr * 64 + tc) = nu;
            *reinterpret_cast<u32x4*>(lt + 1 * 512 + tr * 64 + tc) = nd;
            *reinterpret_cast<u32x4*>(lt + 2 * 512 + tr * 64 + tc) = nz;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            uint16_t cu[8], cd[8], cz[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { cu[j] = lt[0 * 512 + j * 64 + d]; cd[j] = lt[1 * 512 + j * 64 + d]; cz[j] = lt[2 * 512 + j * 64 + d]; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            fetch(t0 + 8 < P.T ? t0 + 8 : t0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                u32x8 bw = {}, cw = {};
                if (MATH) { bw = Bp[(int64_t)(t0 + j) * rs8]; cw = Cp[(int64_t)(t0 + j) * rs8]; }
                float y, yz;
                step<MATH>(S, bw, cw, bf(cu[j]), bf(cd[j]), bf(cz[j]), y, yz);
                lt[0 * 512 + j * 64 + d] = tobf(y);
                lt[1 * 512 + j * 64 + d] = tobf(yz);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const u32x4 oy = *reinterpret_cast<const u32x4*>(lt + 0 * 512 + tr * 64 + tc);
            const u32x4 oo = *reinterpret_cast<const u32x4*>(lt + 1 * 512 + tr * 64 + tc);
            *reinterpret_cast<u32x4*>(yp + (int64_t)(t0 + tr) * st + tc) = oy;
            *reinterpret_cast<u32x4*>(op + (int64_t)(t0 + tr) * st + tc) = oo;
        }
    }
    if (MATH == 0 && S.h[0].x == 12345.f) P.out[0] = 1;
}

template <int MODE, int MATH> static float run(const Args& P, int64_t rows, int reps) {
    const int nblk = (int)(rows / P.T / 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((scan_m<MODE, MATH>), dim3(nblk), dim3(256), 0, 0, P);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((scan_m<MODE, MATH>), dim3(nblk), dim3(256), 0, 0, P);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}

int main() {
    const int64_t rows = 2 * 262144;
    const int dim = 64;
    const int64_t st = 192, rs = 48;
    std::vector<uint16_t> hs((size_t)rows * st), hb((size_t)rows * rs);
    auto tb = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return (uint16_t)(u >> 16); };
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f; };
    for (auto& v : hs) v = tb(rnd() - 0.5f);
    for (auto& v : hb) v = tb(rnd() - 0.5f);
    uint16_t *u, *dl, *z, *oz, *out, *B;
    CHK(hipMalloc(&u, hs.size() * 2)); CHK(hipMalloc(&dl, hs.size() * 2)); CHK(hipMalloc(&z, hs.size() * 2));
    CHK(hipMalloc(&oz, hs.size() * 2)); CHK(hipMalloc(&out, hs.size() * 2)); CHK(hipMalloc(&B, hb.size() * 2));
    CHK(hipMemcpy(u, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dl, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    CHK(hipMemcpy(z, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    CHK(hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> hA(dim * 16);
    for (int d = 0; d < dim; ++d) for (int n = 0; n < 16; ++n) hA[d * 16 + n] = -(n + 1) * 1.4426950408889634f;
    float *A, *carry;
    CHK(hipMalloc(&A, hA.size() * 4)); CHK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    const size_t wsn = (size_t)(rows / 32) * 17 * dim;
    CHK(hipMalloc(&carry, wsn * 4)); CHK(hipMemset(carry, 0, wsn * 4));
    printf("rows %lld x %d channels (%.1f M elements, 10 B each = %.0f MB), bf16; us per launch (GB/s of those bytes)\n", (long long)rows,
           dim, rows * dim / 1e6, rows * dim * 10 / 1e6);
    const double mb = rows * dim * 10 / 1e6;
    for (int T : {64, 128}) {
        Args P{u, dl, z, oz, out, B + 4, B + 20, A, carry, st, rs, T, dim};
        float a0 = run<0, 0>(P, rows, 20), a1 = run<1, 0>(P, rows, 20), a2 = run<2, 0>(P, rows, 20), a3 = run<3, 0>(P, rows, 20);
        float m0 = run<0, 1>(P, rows, 20), m1 = run<1, 1>(P, rows, 20), m2 = run<2, 1>(P, rows, 20), m3 = run<3, 1>(P, rows, 20);
        printf("T=%3d  copy-only: step %6.1f (%5.0f)  sub-tile %6.1f (%5.0f)  buffer %6.1f (%5.0f)  coop+LDS %6.1f (%5.0f)\n", T, a0,
               mb / a0 * 1e3, a1, mb / a1 * 1e3, a2, mb / a2 * 1e3, a3, mb / a3 * 1e3);
        printf("T=%3d  with math: step %6.1f (%5.0f)  sub-tile %6.1f (%5.0f)  buffer %6.1f (%5.0f)  coop+LDS %6.1f (%5.0f)\n", T, m0,
               mb / m0 * 1e3, m1, mb / m1 * 1e3, m2, mb / m2 * 1e3, m3, mb / m3 * 1e3);
    }
    return 0;
}
