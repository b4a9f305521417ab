// GPU probe (not product code): fourth prototype of the wave-uniform scan pass.  What the earlier probes showed
// (profiles/r02_scan_proto_notes.md): the arithmetic alone needs ~140 (aggregate) / ~195 (apply) cycles per wave-step and
// SIMD, a plain copy of the row streams runs at 5.2 TB/s, but one 16-bit global load / store per lane, tensor and step costs
// ~20 cycles of the CU's address path each - that, not VALU or HBM, bounds the shipped kernels.  Here the row streams move
// as 8 step x 64 channel tiles: one 16-byte global access per lane and tensor per 8 steps, transposed through a wave-private
// LDS tile (lane = channel reads / writes its column with 16-bit LDS accesses).  B_t / C_t: fp32 rows by scalar loads.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast tools/proto_scan_v4.hip -o build/proto_scan_v4
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CAS __attribute__((address_space(4)))
#define PINV(x) asm volatile("" : "+v"(x))
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float bf(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ uint16_t tobf(float f) { return (uint16_t)(__float_as_uint(f) >> 16); }

struct Args {
    uint16_t* u; uint16_t* dl; uint16_t* z; uint16_t* oz; uint16_t* out;
    const float* BC;                          // (rows, 32) fp32: B_t (16) then C_t (16)
    const float* A; const float* carry; float* agg;
    int64_t st;
    int T, dim;
};

constexpr int kTS = 8;                        // steps per tile

template <bool AGG, int MINB, int WPB> __global__ void __launch_bounds__(64 * WPB, MINB) scan_v4(Args P) {
    // per wave: input tiles u, delta, z and output tiles y, y*silu(z): [8 steps][64 channels] 16-bit
    __shared__ __attribute__((aligned(16))) uint16_t s_t[WPB][5][kTS * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * WPB + wave;
    const int d = lane;
    const int64_t row0 = (int64_t)item * P.T;
    f2 A2[8], h[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        A2[n] = f2{P.A[d * 16 + 2 * n], P.A[d * 16 + 2 * n + 1]};
        h[n] = AGG ? f2{0.f, 0.f}
                   : f2{P.carry[((int64_t)item * 16 + 2 * n) * P.dim + d], P.carry[((int64_t)item * 16 + 2 * n + 1) * P.dim + d]};
    }
    const int st = (int)P.st;
    // cooperative tile access: lane l moves the 16 bytes of channels 8 (l % 8) .. + 7 of tile row l / 8
    const int tr = lane >> 3, tc = (lane & 7) * 8;
    const uint32_t goff = (uint32_t)(tr * st + tc) * 2u;                 // constant per lane (bytes)
    const char* ub = (const char*)(P.u + row0 * P.st);                   // wave-uniform bases, advanced per tile
    const char* db = (const char*)(P.dl + row0 * P.st);
    const char* zb = (const char*)(P.z + row0 * P.st);
    char* ob = (char*)(P.oz + row0 * P.st);
    char* yb = (char*)(P.out + row0 * P.st);
    const int64_t tile_bytes = (int64_t)kTS * st * 2;
    uint16_t* lt = &s_t[wave][0][0];
    uint16_t* lcol = lt + d;                                             // this lane's column
    u32x4* lrow = reinterpret_cast<u32x4*>(lt + tr * 64 + tc);           // this lane's 16-byte piece of a tile row
    const CAS f32x16* Bp = (const CAS f32x16*)(uintptr_t)(P.BC + row0 * 32);

    u32x4 nu, nd, nz;
    nu = *reinterpret_cast<const u32x4*>(ub + goff);
    nd = *reinterpret_cast<const u32x4*>(db + goff);
    if (!AGG) nz = *reinterpret_cast<const u32x4*>(zb + goff);
    f32x16 bs[2], cs[2];
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(bs[0]) : "s"(Bp));
    if (!AGG) asm volatile("s_load_dwordx16 %0, %1, 0x40" : "=s"(cs[0]) : "s"(Bp));
    float sumd = 0.f;
    const int T = P.T;
#pragma unroll 1
    for (int t0 = 0; t0 < T; t0 += kTS) {
        // park the tile that was in flight, then put the next one in flight
        lrow[0 * 128] = nu;                                              // tile k starts at lt + k * 512 elements = 128 u32x4
        lrow[1 * 128] = nd;
        if (!AGG) lrow[2 * 128] = nz;
        {
            const bool more = t0 + kTS < T;
            ub += more ? tile_bytes : 0; db += more ? tile_bytes : 0; zb += more ? tile_bytes : 0;
            nu = *reinterpret_cast<const u32x4*>(ub + goff);
            nd = *reinterpret_cast<const u32x4*>(db + goff);
            if (!AGG) nz = *reinterpret_cast<const u32x4*>(zb + goff);
        }
        // this lane's column of the first step
        uint32_t eu = lcol[0 * 512], ed = lcol[1 * 512], ez = AGG ? 0u : lcol[2 * 512];
#pragma unroll
        for (int j = 0; j < kTS; ++j) {
            const int t = t0 + j;
            // delivery of this step's B / C rows and column elements (LDS and scalar loads share the counter), then the next
            // step's go in flight
            if (AGG) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(bs[j & 1]), "+v"(eu), "+v"(ed));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(bs[j & 1]), "+s"(cs[j & 1]), "+v"(eu), "+v"(ed), "+v"(ez));
            const float uu = bf(eu);
            float dl = bf(ed);
            const float zz = bf(ez);
            {
                const int tn = t + 1 < T ? t + 1 : t;
                const CAS f32x16* nx = Bp + tn * 2;
                asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(bs[(j + 1) & 1]) : "s"(nx));
                if (!AGG) asm volatile("s_load_dwordx16 %0, %1, 0x40" : "=s"(cs[(j + 1) & 1]) : "s"(nx));
                if (j + 1 < kTS) {
                    eu = lcol[0 * 512 + (j + 1) * 64]; ed = lcol[1 * 512 + (j + 1) * 64];
                    if (!AGG) ez = lcol[2 * 512 + (j + 1) * 64];
                }
            }
            const f32x16 bw = bs[j & 1];
            f32x16 cw = bs[j & 1];
            if (!AGG) cw = cs[j & 1];
            {
                const float e = fexp2(dl * 1.4426950408889634f);
                const float sp = __builtin_amdgcn_logf(1.0f + e) * 0.6931471805599453f;
                dl = dl > 20.f ? dl : sp;
            }
            const float dlu = dl * uu;
            sumd += dl;
            f2 ya = {0.f, 0.f}, yb2 = {0.f, 0.f};
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
                    yb2 = c1 * h[2 * q + 1] + yb2;
                }
            }
            if (!AGG) {
                const float y = (ya.x + yb2.x) + (ya.y + yb2.y);
                const float sg = __builtin_amdgcn_rcpf(1.f + fexp2(-zz * 1.4426950408889634f));
                lcol[3 * 512 + j * 64] = tobf(y);
                lcol[4 * 512 + j * 64] = tobf(y * zz * sg);
            }
#pragma unroll
            for (int n = 0; n < 8; ++n) PINV(h[n]);
        }
        if (!AGG) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the column writes of all lanes have landed
            const u32x4 oy = lrow[3 * 128], oo = lrow[4 * 128];
            *reinterpret_cast<u32x4*>(yb + goff) = oy;
            *reinterpret_cast<u32x4*>(ob + goff) = oo;
            yb += tile_bytes; ob += tile_bytes;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(bs[0]), "+s"(bs[1]));    // drain the last (clamped) prefetch before the registers are reused
    if (!AGG) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(cs[0]), "+s"(cs[1]));
    if (AGG) {
        P.agg[((int64_t)item * 17 + 16) * P.dim + d] = sumd;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            P.agg[((int64_t)item * 17 + 2 * n) * P.dim + d] = h[n].x;
            P.agg[((int64_t)item * 17 + 2 * n + 1) * P.dim + d] = h[n].y;
        }
    }
}

template <bool AGG, int MINB, int WPB> static float run(const Args& P, int64_t rows, int reps) {
    const int nblk = (int)(rows / P.T / WPB);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((scan_v4<AGG, MINB, WPB>), dim3(nblk), dim3(64 * WPB), 0, 0, P);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((scan_v4<AGG, MINB, WPB>), dim3(nblk), dim3(64 * WPB), 0, 0, P);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
    const int64_t rows = 2 * 262144;
    const int dim = 64;
    const int64_t st = 192;
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
    const int which = argc > 1 ? atoi(argv[1]) : 0;
    for (int T : {64, 128, 256}) {
        Args P{u, dl, z, oz, out, B, A, carry, agg, st, T, dim};
        float r = 0.f; const char* nm = "";
        switch (which) {
            case 0: r = run<true, 2, 4>(P, rows, 20); nm = "aggregate, free regs"; break;
            case 1: r = run<false, 2, 4>(P, rows, 20); nm = "apply, free regs"; break;
            case 2: r = run<false, 6, 4>(P, rows, 20); nm = "apply, 6 waves / SIMD"; break;
            case 3: r = run<false, 8, 4>(P, rows, 20); nm = "apply, 8 waves / SIMD"; break;
            case 4: r = run<true, 8, 4>(P, rows, 20); nm = "aggregate, 8 waves / SIMD"; break;
            default: break;
        }
        CHK(hipDeviceSynchronize());
        printf("T=%3d  %-28s %6.1f us\n", T, nm, r);
        fflush(stdout);
    }
    return 0;
}
