// GPU probe (not product code): fifth prototype of the scan pass (no wave-uniformity needed any more).  What the earlier probes showed
// (profiles/r02_scan_proto_notes.md): the arithmetic alone needs ~140 (aggregate) / ~195 (apply) cycles per wave-step and
// SIMD, a plain copy of the row streams runs at 5.2 TB/s, but one 16-bit global load / store per lane, tensor and step costs
// ~20 cycles of the CU's address path each - that, not VALU or HBM, bounds the shipped kernels.  Here the row streams move
// as 8 step x 64 channel tiles: one 16-byte global access per lane and tensor per 8 steps, transposed through a wave-private
// LDS tile (lane = channel reads / writes its column with 16-bit LDS accesses).  B_t / C_t (streaming them through the scalar
// cache cost 20 - 30 us in proto_scan_v5): an 8 step x 32 fp32 tile in LDS; per step ONE ds_read_b128 per matrix hands lane l
// the four values 4 (l % 4) .. + 3, so every quad holds all 16 and state n = 4 k + i is the DPP operand quad_perm:[k,k,k,k]
// of register i (folded into v_fmac_f32_dpp) - no broadcast reads of 64 bytes per lane, no SGPRs.  Outputs overwrite the
// input tiles in place.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast tools/proto_scan_v5.hip -o build/proto_scan_v5
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
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CAS __attribute__((address_space(4)))
#define PINV(x) asm volatile("" : "+v"(x))
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float bf(uint32_t w) { return __uint_as_float(w << 16); }
// acc += x[lane 4 (l / 4) + k of this lane's quad] * y : one VOP2 with a DPP operand (inline asm: the compiler's DPP combiner
// gives up once it has packed the two multiply-adds of a state pair)
template <int K> __device__ __forceinline__ void fmac_quad(float& acc, float x, float y) {
    if (K == 0) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(y));
    if (K == 1) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(y));
    if (K == 2) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(y));
    if (K == 3) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(y));
}
__device__ __forceinline__ uint16_t tobf(float f) { return (uint16_t)(__float_as_uint(f) >> 16); }

struct Args {
    uint16_t* u; uint16_t* dl; uint16_t* z; uint16_t* oz; uint16_t* out;
    const float* BC;                          // (rows, 32) fp32: B_t (16) then C_t (16)
    const float* A; const float* carry; float* agg;
    int64_t st;
    int T, dim;
};

constexpr int kTS = 8;                        // steps per tile

template <bool AGG, int MINB, int WPB> __global__ void __launch_bounds__(64 * WPB, MINB) scan_v5(Args P) {
    // per wave: input tiles u, delta, z and output tiles y, y*silu(z): [8 steps][64 channels] 16-bit
    __shared__ __attribute__((aligned(16))) uint16_t s_t[WPB][3][kTS * 64];
    __shared__ __attribute__((aligned(16))) float s_bc[WPB][kTS * 32];
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
    const char* bcb = (const char*)(P.BC + row0 * 32);                   // 8 rows x 128 B = one 16-byte piece per lane
    float* lbc = &s_bc[wave][0];
    const float* lq = lbc + 4 * (lane & 3);                              // this lane's quarter of a B row (C: + 16)
    f32x4 nbc = *reinterpret_cast<const f32x4*>(bcb + lane * 16);

    u32x4 nu, nd, nz;
    nu = *reinterpret_cast<const u32x4*>(ub + goff);
    nd = *reinterpret_cast<const u32x4*>(db + goff);
    if (!AGG) nz = *reinterpret_cast<const u32x4*>(zb + goff);
    float sumd = 0.f;
    const int T = P.T;
#pragma unroll 1
    for (int t0 = 0; t0 < T; t0 += kTS) {
        // park the tile that was in flight, then put the next one in flight
        lrow[0 * 128] = nu;                                              // tile k starts at lt + k * 512 elements = 128 u32x4
        lrow[1 * 128] = nd;
        if (!AGG) lrow[2 * 128] = nz;
        reinterpret_cast<f32x4*>(lbc)[lane] = nbc;
        {
            const bool more = t0 + kTS < T;
            ub += more ? tile_bytes : 0; db += more ? tile_bytes : 0; zb += more ? tile_bytes : 0;
            nu = *reinterpret_cast<const u32x4*>(ub + goff);
            nd = *reinterpret_cast<const u32x4*>(db + goff);
            if (!AGG) nz = *reinterpret_cast<const u32x4*>(zb + goff);
            bcb += more ? kTS * 128 : 0;
            nbc = *reinterpret_cast<const f32x4*>(bcb + lane * 16);
        }
        // this lane's column of the first step
        uint32_t eu = lcol[0 * 512], ed = lcol[1 * 512], ez = AGG ? 0u : lcol[2 * 512];
#pragma unroll
        for (int j = 0; j < kTS; ++j) {
            const int t = t0 + j;
            const float uu = bf(eu);
            float dl = bf(ed);
            const float zz = bf(ez);
            const f32x4 xb = *reinterpret_cast<const f32x4*>(lq + j * 32);
            f32x4 xc = xb;
            if (!AGG) xc = *reinterpret_cast<const f32x4*>(lq + j * 32 + 16);
            if (j + 1 < kTS) {
                eu = lcol[0 * 512 + (j + 1) * 64]; ed = lcol[1 * 512 + (j + 1) * 64];
                if (!AGG) ez = lcol[2 * 512 + (j + 1) * 64];
            }
            {
                const float e = fexp2(dl * 1.4426950408889634f);
                const float sp = __builtin_amdgcn_logf(1.0f + e) * 0.6931471805599453f;
                dl = dl > 20.f ? dl : sp;
            }
            const float dlu = dl * uu;
            sumd += dl;
            float ya = 0.f, yb2 = 0.f;
#define STATE_PAIR(p)                                                                                                     \
            {   /* states 2 p, 2 p + 1 = registers (2 p) % 4, + 1 of quad lane (2 p) / 4 */                               \
                const f2 da = A2[p] * dl;                                                                                 \
                const f2 a = {fexp2(da.x), fexp2(da.y)};                                                                  \
                const f2 ah = a * h[p];                                                                                   \
                float hx = ah.x, hy = ah.y;                                                                               \
                constexpr int k = (2 * p) / 4, i = (2 * p) % 4;                                                           \
                fmac_quad<k>(hx, xb[i], dlu);                                                                             \
                fmac_quad<k>(hy, xb[i + 1], dlu);                                                                         \
                h[p] = f2{hx, hy};                                                                                        \
                if (!AGG) { fmac_quad<k>(ya, xc[i], hx); fmac_quad<k>(yb2, xc[i + 1], hy); }                              \
            }
            STATE_PAIR(0) STATE_PAIR(1) STATE_PAIR(2) STATE_PAIR(3) STATE_PAIR(4) STATE_PAIR(5) STATE_PAIR(6) STATE_PAIR(7)
#undef STATE_PAIR
            if (!AGG) {
                const float y = ya + yb2;
                const float sg = __builtin_amdgcn_rcpf(1.f + fexp2(-zz * 1.4426950408889634f));
                lcol[0 * 512 + j * 64] = tobf(y);
                lcol[2 * 512 + j * 64] = tobf(y * zz * sg);
            }
#pragma unroll
            for (int n = 0; n < 8; ++n) PINV(h[n]);
        }
        if (!AGG) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the column writes of all lanes have landed
            const u32x4 oy = lrow[0 * 128], oo = lrow[2 * 128];
            *reinterpret_cast<u32x4*>(yb + goff) = oy;
            *reinterpret_cast<u32x4*>(ob + goff) = oo;
            yb += tile_bytes; ob += tile_bytes;
        }
    }
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
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((scan_v5<AGG, MINB, WPB>), dim3(nblk), dim3(64 * WPB), 0, 0, P);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((scan_v5<AGG, MINB, WPB>), dim3(nblk), dim3(64 * WPB), 0, 0, P);
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
