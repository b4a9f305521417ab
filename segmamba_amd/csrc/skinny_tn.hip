// out (m, n) fp32 = a^T b for a WIDE a (k, m) and a SKINNY b (k, n <= 32), k = tokens (C ABI: segm_skinny_tn).
//
// The weight gradient of Mamba's dt_proj (reference selective_scan_interface.py:272: `einsum("dB,Br->dr", ddelta, x_dbl[:, :R])`)
// at SegMamba's sizes: ddelta (524 288 x 96), dt (524 288 x 3).  As a split-K batched GEMM in the vendor library it takes 72 us
// for 100 MB of operands (profiles/r03_copy_shapes.log: bmm [128, 96, 4096] x [128, 4096, 3], 0.43 ms per step for six of them;
// the same again at stage 1 with n = 6): a GEMM tile with three useful output columns.  It is a streaming reduction and HBM-bound
// (bytes = k (m + n) e), so the kernel is shaped by the loads, not by the arithmetic:
//   * a thread owns EIGHT consecutive channels (one 16-byte load per row) and a tile of NT columns: 8 NT running sums;
//   * m / 8 threads cover a row, the workgroup's 256 threads cover 256 / (m / 8) consecutive rows per pass and walk their slab
//     with four or eight passes of loads in flight (64 - 128 B per thread);
//   * the rows a workgroup's threads summed separately are folded through LDS in a fixed order, one partial (n x m) per slab is
//     written, and reduce_partials adds the slabs in a fixed order: deterministic, no atomics.
// (The first version - one lane per channel, 2-byte loads, wave-uniform reads of b - ran at 0.1 TB/s: profiles/r03_step_kernels_v4.txt.)
// Round 5 (BVEC): the NT values of b's row are ONE 8- / 16-byte load where the row segment is that aligned (x_dbl in the padded
// layout is), not NT two-byte loads - five load instructions per 16 useful bytes had the address unit, not HBM, as the limit.
#include <stdlib.h>
#include <string.h>

#include "segm_device.h"

namespace segm {

void launch_reduce_partials(const float* part, int64_t nrows, int K, int dim, float* out0, int K0, float* out1,
                            float* out2, hipStream_t stream);

constexpr int kSkMaxN = 32;
constexpr int kSkCh = 8;               // channels per thread = one 16-byte load
constexpr int kSkSlabs = 1024;         // slabs the rows are cut into at most (four workgroups per CU)
constexpr int kSkMaxLds = 12288;       // floats of the fold buffer (m x column tile)

struct SkinnyDev {
    const char* a; int64_t a_sr;       // (k, m), unit column stride, row stride in elements
    const char* b; int64_t b_sr;       // (k, n)
    float* part;                       // [slab][n][m]
    int64_t k, slab_rows;
    int32_t m, n, nslab, threads_per_row, rows_per_pass;
};

// rows of one pass / of one slab for a problem: shared by the launcher and segm_skinny_tn_workspace_bytes
static inline int skinny_tile(int32_t n) { return n <= 4 ? 4 : 8; }          // columns per workgroup (16 x 8 running sums would spill)
static inline int skinny_passes(int nt) { return nt == 4 ? 8 : 4; }          // passes of loads in flight (16 bytes per thread each)
static inline void skinny_geometry(int32_t m, int32_t n, int64_t k, int32_t* threads_per_row, int32_t* rows_per_pass, int64_t* slab_rows,
                                   int32_t* nslab) {
    *threads_per_row = m / kSkCh;
    *rows_per_pass = kBlock / *threads_per_row;
    const int64_t macro = (int64_t)*rows_per_pass * skinny_passes(skinny_tile(n));
    const int64_t nmacro = (k + macro - 1) / macro;
    *slab_rows = ((nmacro + kSkSlabs - 1) / kSkSlabs) * macro;
    *nslab = (int32_t)((k + *slab_rows - 1) / *slab_rows);
}

template <typename T, int NT, int kSkU, bool BVEC = false>
__global__ void __launch_bounds__(kBlock) skinny_tn_kernel(SkinnyDev P) {
    typedef uint32_t bvec_t __attribute__((ext_vector_type(NT / 2)));
    __shared__ float s_fold[kSkMaxLds];
    const int tid = threadIdx.x;
    const int row = tid / P.threads_per_row, cg = tid - row * P.threads_per_row;
    const bool active = row < P.rows_per_pass;
    const int j0 = blockIdx.y * NT;                          // first column of this workgroup's tile
    const int nj = P.n - j0 < NT ? P.n - j0 : NT;
    const int64_t k0 = (int64_t)blockIdx.x * P.slab_rows;
    const int64_t k1 = k0 + P.slab_rows < P.k ? k0 + P.slab_rows : P.k;
    const T* a = reinterpret_cast<const T*>(P.a) + cg * kSkCh;
    const T* b = reinterpret_cast<const T*>(P.b) + j0;
    float acc[kSkCh][NT];
#pragma unroll
    for (int i = 0; i < kSkCh; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = 0.f;
    if (active) {
        for (int64_t r = k0 + row; r < k1; r += (int64_t)kSkU * P.rows_per_pass) {
            uint4 raw[kSkU];
            T bv[kSkU][NT];
#pragma unroll
            for (int u = 0; u < kSkU; ++u) {
                const int64_t ru = r + (int64_t)u * P.rows_per_pass;
                const bool ok = ru < k1;
                raw[u] = ok ? *reinterpret_cast<const uint4*>(a + ru * P.a_sr) : uint4{0u, 0u, 0u, 0u};
                if constexpr (BVEC) {                     // the launcher: segment aligned, NT columns readable in every row
                    bvec_t rb;
#pragma unroll
                    for (int w = 0; w < NT / 2; ++w) rb[w] = 0u;
                    // the LAST row of the matrix with a partial column tile: the vector would read up to NT - 1 elements beyond the
                    // row, i.e. beyond the allocation when `skinny` is a column slice at the end of its buffer -> masked
                    // element loads for that one row
                    const bool edge = nj < NT && ru == P.k - 1;
                    if (ok && !edge) rb = *reinterpret_cast<const bvec_t*>(b + ru * P.b_sr);
                    T tb[NT];
                    memcpy(tb, &rb, NT * 2);
                    if (ok && edge) {
#pragma unroll
                        for (int j = 0; j < NT; ++j) tb[j] = j < nj ? b[ru * P.b_sr + j] : (T)0.f;
                    }
#pragma unroll
                    for (int j = 0; j < NT; ++j) bv[u][j] = j < nj ? tb[j] : (T)0.f;
                } else {
#pragma unroll
                    for (int j = 0; j < NT; ++j) bv[u][j] = (ok && j < nj) ? b[ru * P.b_sr + j] : (T)0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < kSkU; ++u) {
                T av[kSkCh];
                memcpy(av, &raw[u], 16);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float bj = to_f32(bv[u][j]);
#pragma unroll
                    for (int i = 0; i < kSkCh; ++i) acc[i][j] = fmaf(to_f32(av[i]), bj, acc[i][j]);
                }
            }
        }
    }
    // fold the rows_per_pass partial sums of every (column, channel) in row order; s_fold[j][m]
    for (int q = 0; q < P.rows_per_pass; ++q) {
        if (active && row == q) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (j >= nj) break;
#pragma unroll
                for (int i = 0; i < kSkCh; ++i) {
                    float* d = &s_fold[j * P.m + cg * kSkCh + i];
                    *d = q == 0 ? acc[i][j] : *d + acc[i][j];
                }
            }
        }
        __syncthreads();
    }
    for (int id = tid; id < nj * P.m; id += kBlock) {
        const int j = id / P.m, mm = id - j * P.m;
        P.part[((int64_t)blockIdx.x * P.n + j0 + j) * P.m + mm] = s_fold[id];
    }
}

}  // namespace segm

using namespace segm;

extern "C" size_t segm_skinny_tn_workspace_bytes(int32_t m, int32_t n, int64_t k) {
    if (m <= 0 || n <= 0 || k <= 0 || m % kSkCh != 0 || m / kSkCh > kBlock) return 0;
    int32_t tpr, rpp, nslab;
    int64_t slab_rows;
    skinny_geometry(m, n, k, &tpr, &rpp, &slab_rows, &nslab);
    return (size_t)nslab * n * m * sizeof(float);
}

extern "C" int segm_skinny_tn(const segm_skinny_tn_args* a) {
    if (!a) return SEGM_E_NULL;
    if (!a->wide || !a->skinny || !a->out || !a->workspace) return SEGM_E_NULL;
    if (a->m <= 0 || a->n <= 0 || a->n > kSkMaxN || a->k <= 0 || a->wide_stride_row < a->m || a->skinny_stride_row < a->n) return SEGM_E_SHAPE;
    if (a->m % kSkCh != 0 || a->m / kSkCh > kBlock || a->wide_stride_row % kSkCh != 0) return SEGM_E_SHAPE;     // 16-byte loads of whole rows
    if (((uintptr_t)a->wide & 15u) != 0) return SEGM_E_SHAPE;
    if (a->dtype != SEGM_BF16 && a->dtype != SEGM_F16) return SEGM_E_DTYPE;
    if (a->workspace_bytes < segm_skinny_tn_workspace_bytes(a->m, a->n, a->k)) return SEGM_E_WORKSPACE;
    SkinnyDev P;
    memset(&P, 0, sizeof(P));
    P.a = (const char*)a->wide; P.a_sr = a->wide_stride_row;
    P.b = (const char*)a->skinny; P.b_sr = a->skinny_stride_row;
    P.part = (float*)a->workspace;
    P.k = a->k; P.m = a->m; P.n = a->n;
    skinny_geometry(a->m, a->n, a->k, &P.threads_per_row, &P.rows_per_pass, &P.slab_rows, &P.nslab);
    // column tile: the largest whose fold buffer (m x tile floats) fits; 8 x tile running sums per thread
    int nt = skinny_tile(a->n);
    while (nt > 4 && (int64_t)nt * a->m > kSkMaxLds) nt >>= 1;
    if ((int64_t)nt * a->m > kSkMaxLds) return SEGM_E_SHAPE;
    hipStream_t st = (hipStream_t)a->stream;
    const dim3 grid(P.nslab, (a->n + nt - 1) / nt), block(kBlock);
    const bool f16 = a->dtype == SEGM_F16;
    // one vector load per row of b: every column tile's segment aligned to its size, and nt columns readable in every row (the
    // row stride covers the last tile); SEGM_SKINNY_BVEC=0: scalar loads (A/B)
    const char* be = getenv("SEGM_SKINNY_BVEC");
    const int ntiles = (a->n + nt - 1) / nt;
    const bool bvec = !(be && be[0] == '0') && ((uintptr_t)a->skinny % (size_t)(nt * 2)) == 0 && a->skinny_stride_row % nt == 0 &&
                      (int64_t)ntiles * nt <= a->skinny_stride_row;
#define SEGM_SK(NN, UU, BV)                                                                                 \
    do {                                                                                                 \
        if (f16) hipLaunchKernelGGL((skinny_tn_kernel<f16_t, NN, UU, BV>), grid, block, 0, st, P);       \
        else hipLaunchKernelGGL((skinny_tn_kernel<bf16_t, NN, UU, BV>), grid, block, 0, st, P);          \
    } while (0)
    if (nt == 4) { if (bvec) SEGM_SK(4, 8, true); else SEGM_SK(4, 8, false); }      // (rows beyond the slab are masked, so the passes need not divide the slab)
    else { if (bvec) SEGM_SK(8, 4, true); else SEGM_SK(8, 4, false); }
#undef SEGM_SK
    launch_reduce_partials(P.part, P.nslab, a->n, a->m, a->out, a->n, nullptr, nullptr, st);
    return (int)hipGetLastError();
}
