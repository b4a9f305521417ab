// out (m, n) fp32 = a^T b for a WIDE a (k, m) and a SKINNY b (k, n <= 32), k = tokens (C ABI: segm_skinny_tn).
//
// The weight gradient of Mamba's dt_proj (reference selective_scan_interface.py:272: `einsum("dB,Br->dr", ddelta, x_dbl[:, :R])`)
// at SegMamba's sizes: ddelta (524 288 x 96), dt (524 288 x 3).  As a split-K batched GEMM in the vendor library it takes 72 us
// for 100 MB of operands (profiles/r03_copy_shapes.log: bmm [128, 96, 4096] x [128, 4096, 3], 0.43 ms per step for six of them;
// the same again at stage 1 with n = 6): a GEMM tile with three useful output columns.  It is a streaming reduction: one lane per
// channel m keeps n running sums, reads its column of `a` row by row (a wave reads 128 contiguous bytes per row) and the row of `b`
// through wave-uniform loads; a workgroup's four waves take interleaved rows of one k slab, add their sums through LDS and leave one
// partial per slab, summed in a fixed order by reduce_partials (deterministic).  HBM-bound: bytes = k (m + n) e.
#include <stdlib.h>
#include <string.h>

#include "segm_device.h"

namespace segm {

void launch_reduce_partials(const float* part, int64_t nrows, int K, int dim, float* out0, int K0, float* out1,
                            float* out2, hipStream_t stream);

constexpr int kSkMaxN = 32;
constexpr int kSkRows = 4096;          // rows of k per workgroup (slab)

struct SkinnyDev {
    const char* a; int64_t a_sr;       // (k, m), unit column stride, row stride in elements
    const char* b; int64_t b_sr;       // (k, n)
    float* part;                       // [slab][n][m]
    int64_t k;
    int32_t m, n, nslab;
};

template <typename T, int N>
__global__ void __launch_bounds__(kBlock) skinny_tn_kernel(SkinnyDev P) {
    __shared__ float s_acc[kWavesPerBlock][N][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mm = blockIdx.y * 64 + lane;
    const bool valid = mm < P.m;
    const int64_t k0 = (int64_t)blockIdx.x * kSkRows;
    const int64_t k1 = k0 + kSkRows < P.k ? k0 + kSkRows : P.k;
    const T* a = reinterpret_cast<const T*>(P.a) + (valid ? mm : 0);
    const T* b = reinterpret_cast<const T*>(P.b);
    float acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = 0.f;
    constexpr int U = 8;                                    // rows in flight per wave
    int64_t r = k0 + wave;
    for (; r + (U - 1) * kWavesPerBlock < k1; r += U * kWavesPerBlock) {
        float av[U];
#pragma unroll
        for (int u = 0; u < U; ++u) av[u] = to_f32(a[(r + u * kWavesPerBlock) * P.a_sr]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const T* br = b + (r + u * kWavesPerBlock) * P.b_sr;       // wave-uniform address: broadcast loads
#pragma unroll
            for (int j = 0; j < N; ++j)
                if (j < P.n) acc[j] = fmaf(av[u], to_f32(br[j]), acc[j]);
        }
    }
    for (; r < k1; r += kWavesPerBlock) {
        const float av = to_f32(a[r * P.a_sr]);
        const T* br = b + r * P.b_sr;
#pragma unroll
        for (int j = 0; j < N; ++j)
            if (j < P.n) acc[j] = fmaf(av, to_f32(br[j]), acc[j]);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) s_acc[wave][j][lane] = acc[j];
    __syncthreads();
    if (wave == 0 && valid) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (j >= P.n) break;
            const float t = (s_acc[0][j][lane] + s_acc[1][j][lane]) + (s_acc[2][j][lane] + s_acc[3][j][lane]);
            P.part[((int64_t)blockIdx.x * P.n + j) * P.m + mm] = t;
        }
    }
}

}  // namespace segm

using namespace segm;

extern "C" size_t segm_skinny_tn_workspace_bytes(int32_t m, int32_t n, int64_t k) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    const int64_t nslab = (k + kSkRows - 1) / kSkRows;
    return (size_t)nslab * n * m * sizeof(float);
}

extern "C" int segm_skinny_tn(const segm_skinny_tn_args* a) {
    if (!a) return SEGM_E_NULL;
    if (!a->wide || !a->skinny || !a->out || !a->workspace) return SEGM_E_NULL;
    if (a->m <= 0 || a->n <= 0 || a->n > kSkMaxN || a->k <= 0 || a->wide_stride_row < a->m || a->skinny_stride_row < a->n) return SEGM_E_SHAPE;
    if (a->dtype != SEGM_BF16 && a->dtype != SEGM_F16) return SEGM_E_DTYPE;
    if (a->workspace_bytes < segm_skinny_tn_workspace_bytes(a->m, a->n, a->k)) return SEGM_E_WORKSPACE;
    SkinnyDev P;
    memset(&P, 0, sizeof(P));
    P.a = (const char*)a->wide; P.a_sr = a->wide_stride_row;
    P.b = (const char*)a->skinny; P.b_sr = a->skinny_stride_row;
    P.part = (float*)a->workspace;
    P.k = a->k; P.m = a->m; P.n = a->n;
    P.nslab = (int32_t)((a->k + kSkRows - 1) / kSkRows);
    hipStream_t st = (hipStream_t)a->stream;
    const dim3 grid(P.nslab, (a->m + 63) / 64), block(kBlock);
    const bool f16 = a->dtype == SEGM_F16;
#define SEGM_SK(NN)                                                                                      \
    do {                                                                                                 \
        if (f16) hipLaunchKernelGGL((skinny_tn_kernel<f16_t, NN>), grid, block, 0, st, P);               \
        else hipLaunchKernelGGL((skinny_tn_kernel<bf16_t, NN>), grid, block, 0, st, P);                  \
    } while (0)
    if (a->n <= 4) SEGM_SK(4);
    else if (a->n <= 8) SEGM_SK(8);
    else if (a->n <= 16) SEGM_SK(16);
    else SEGM_SK(32);
#undef SEGM_SK
    launch_reduce_partials(P.part, P.nslab, a->n, a->m, a->out, a->n, nullptr, nullptr, st);
    return (int)hipGetLastError();
}
