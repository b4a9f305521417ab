// Weight gradients of the projections: out (M, N) fp32 = sum over a long axis k of a[.][k] b[.][k]   (C ABI: segm_wgrad_gemm)
//
// The reference leaves these to cuBLAS through autograd: dW of `in_proj` / `out_proj` / `x_proj` / `dt_proj`
// (mamba/mamba_ssm/modules/mamba_simple.py:204-208,264; ops/selective_scan_interface.py:272-276) and of the 1x1x1 convolutions of
// the conv stem (monai/networks/blocks/dynunet_block.py:72-96,247-263).  They are "tall-skinny transposed" products: M, N <= a
// few hundred, K = batch x voxels = 65 536 .. 2 097 152.  Round 1 cut K into 4096-row slabs that became the batch dimension of a
// rocBLAS batched GEMM plus a sum (linear.py tn_matmul / nt_matmul_rows): 0.7 - 1.4 TB/s of operand traffic
// (profiles/r02_copy_shapes.log: 4.56 ms of bmm per training step).  The operands are streamed ONCE here:
//
//   layout TN  a (K, M), b (K, N) row-major (token-major activations: the Mamba projections).  MFMA wants 8 consecutive k per
//              lane but k is the row index, so a wave stages 32-row tiles [32][<= 64] of a and [32][<= 96] of b in a private
//              LDS strip (16-byte loads when the rows are 16-byte aligned, 2-byte loads otherwise - x_dbl has 35 columns) and
//              reads the fragments back TRANSPOSED with gfx950's ds_read_b64_tr_b16 (round 5; rounds 2 - 4 gathered them with
//              eight ds_read_u16 + packing each - 80 LDS reads per 32-row chunk - and lost to split-K rocBLAS).  What the
//              instruction does was measured (tools/experiments/tr16_probe.hip): with E[p] the four 16-bit values at lane p's
//              address, lane i of a 16-lane group receives E[4 j + (i >> 2)][i & 3], j = 0 .. 3.  Lane p = 4 r + c pointing at row
//              k0 + r, columns m0 + 4 c .. of the row-major strip, lane i gets column m0 + i of rows k0 .. k0 + 3: two reads are
//              one MFMA operand fragment.  Which four rows a lane group takes is free as long as both operands agree (the
//              contraction is a sum): group g reads rows 4 g .. and 16 + 4 g .., so the 32 lanes the LDS serves together
//              (groups 0 / 1, then 2 / 3) touch eight consecutive rows, and at pitches of 40 / 56 dwords (80 / 112 elements) those
//              are eight 8-dword windows on 64 distinct banks;
//   layout NT  a (B, M, K), b (B, N, K) with unit stride along K (channel-first volumes: the 1x1x1 convolutions): both operand
//              fragments are 16 contiguous bytes in memory - no staging at all.
//
// Every wave accumulates its share of K in registers (<= 24 resp. 36 tiles of 16 x 16) and writes one partial; a second launch
// adds the partials in a fixed order (no atomics: bitwise repeatable).
// v_mfma_f32_16x16x32: A[i][k]: lane l holds A[i = l & 15][8 (l >> 4) .. +7]; B[k][j]: lane l holds B[8 (l >> 4) .. +7][j = l & 15];
// D[row = 4 (l >> 4) + r][col = l & 15].  Here i = m, j = n.
#include <string.h>

#include "segm_device.h"

namespace segm {

typedef float wg_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t wg_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t wg_u32x2 __attribute__((ext_vector_type(2)));

constexpr int kGwWaves = 4;
constexpr int kTnMB = 64, kTnNB = 96;        // columns of a / b per workgroup
constexpr int kTnMT = kTnMB / 16, kTnNT = kTnNB / 16;
constexpr int kTnPA = kTnMB + 16, kTnPB = kTnNB + 16;    // LDS pitches (elements): 40 / 56 dwords - eight consecutive rows = 64 distinct banks for the transposing reads
constexpr int kNtMax = 6;                    // 16-row tiles per operand of the NT kernel (m, n <= 96)

struct GemmDev {
    const char* a;  int64_t a_sr, a_sb;      // element strides: row (TN: per k; NT: per m) and batch (NT)
    const char* b;  int64_t b_sr, b_sb;
    float* part;                             // [waves][m16][n16]
    int32_t m, n, m16, n16;
    int64_t k;                               // TN: rows; NT: K per batch
    int32_t batch;                           // NT
    int64_t chunks;                          // 32-wide k chunks in total (NT: batch * K / 32)
    int32_t kwaves;                          // waves that share the k range (per (m block, n block) in TN)
    int64_t chunks_per_wave;
};

// ------------------------------------------------------------------------------------------------------
// TN
// ------------------------------------------------------------------------------------------------------
// stage rows [k0, k0 + 32) x columns [c0, c0 + cw) of a row-major matrix into lds[32][PITCH]; rows >= K and columns >= cols are zero
// MODE 1: cols % 8 == 0 and 16-byte aligned rows - every piece is whole.  MODE 2 (round 6): 16-byte aligned rows whose STRIDE covers
// the columns rounded up to 8 (x_dbl / dx_dbl's padded layout: 36 of 40 columns) - the piece that straddles `cols` is read whole too:
// what it brings beyond `cols` only reaches output rows / columns beyond m / n, which nobody reads; the matrix's LAST row takes the
// masked element loads (its padding may lie outside the allocation when the operand is a column slice at the end of a buffer).
// MODE 0: element loads.
template <typename T, int PITCH, int CB, int MODE>
__device__ __forceinline__ void tn_fetch(const T* base, int64_t sr, int64_t k0, int64_t K, int c0, int cols, int lane, wg_u32x4 (&r)[CB * 32 / 8 / 64]) {
    constexpr int PIECES = CB / 8;           // 16-byte pieces per row
    constexpr int PER = CB * 32 / 8 / 64;    // pieces per lane
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int id = q * 64 + lane, row = id / PIECES, pc = id - row * PIECES;
        const int64_t kk = k0 + row;
        const int col = c0 + 8 * pc;
        wg_u32x4 v = {0u, 0u, 0u, 0u};
        if (kk < K && col < cols) {
            const T* src = base + kk * sr + col;
            if (MODE == 1 || (MODE == 2 && (kk + 1 < K || col + 8 <= cols))) {
                v = *reinterpret_cast<const wg_u32x4*>(src);
            } else {
                T e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = (col + j < cols) ? src[j] : from_f32<T>(0.f);
                memcpy(&v, e, 16);
            }
        }
        r[q] = v;
    }
}
template <typename T, int PITCH, int CB>
__device__ __forceinline__ void tn_park(T* lds, int lane, const wg_u32x4 (&r)[CB * 32 / 8 / 64]) {
    constexpr int PIECES = CB / 8;
    constexpr int PER = CB * 32 / 8 / 64;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int id = q * 64 + lane, row = id / PIECES, pc = id - row * PIECES;
        static_assert(PITCH % 8 == 0, "16-byte aligned strip rows");
        *reinterpret_cast<wg_u32x4*>(lds + row * PITCH + 8 * pc) = r[q];
    }
}

// Round 6: NB = columns of b per workgroup (48 or 96: half the accumulators and a smaller strip for the n <= 48 shapes - in_proj's
// and the 1x1x1 layers' - so that more waves fit a SIMD) and TWO chunks of both operands in flight per wave (registers set 0 / 1,
// the loop runs two chunks per trip): the kernel is a latency pipeline - a chunk's MFMAs are 200 cycles, its loads 2 000+ away.
template <typename T, int NB, int MODE_A, int MODE_B>
__global__ void __launch_bounds__(kGwWaves * 64) wgemm_tn_kernel(GemmDev P) {
    typedef typename Mfma16<T>::v8 frag8;
    constexpr int NT_ = NB / 16, PB = NB == 48 ? 80 : NB + 16;    // 40 / 56 dwords: eight consecutive rows on 64 distinct banks (32 dwords would not be)
    __shared__ __attribute__((aligned(16))) T s_a[kGwWaves][32 * kTnPA];
    __shared__ __attribute__((aligned(16))) T s_b[kGwWaves][32 * PB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int kw = blockIdx.x * kGwWaves + wave;                 // this wave's share of k
    const int m0 = blockIdx.y * kTnMB, n0 = blockIdx.z * NB;
    if (kw >= P.kwaves) return;                                   // whole waves; no workgroup barrier in this kernel
    const int mw = P.m - m0 < kTnMB ? P.m - m0 : kTnMB, nw = P.n - n0 < NB ? P.n - n0 : NB;
    const int mt_n = (mw + 15) / 16, nt_n = (nw + 15) / 16;
    const T* A = reinterpret_cast<const T*>(P.a);
    const T* B = reinterpret_cast<const T*>(P.b);
    T* la = &s_a[wave][0];
    T* lb = &s_b[wave][0];

    wg_f32x4 acc[kTnMT][NT_];
#pragma unroll
    for (int mt = 0; mt < kTnMT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT_; ++nt) acc[mt][nt] = wg_f32x4{0.f, 0.f, 0.f, 0.f};

    const int64_t c_begin = (int64_t)kw * P.chunks_per_wave;
    const int64_t c_end = c_begin + P.chunks_per_wave < P.chunks ? c_begin + P.chunks_per_wave : P.chunks;
    wg_u32x4 ra[2][kTnMB * 32 / 8 / 64], rb[2][NB * 32 / 8 / 64];
    auto fetch = [&](int set, int64_t c) {
        if (c < c_end) {
            tn_fetch<T, kTnPA, kTnMB, MODE_A>(A, P.a_sr, c * 32, P.k, m0, P.m, lane, ra[set]);
            tn_fetch<T, PB, NB, MODE_B>(B, P.b_sr, c * 32, P.k, n0, P.n, lane, rb[set]);
        }
    };
    auto chunk = [&](int set, int64_t c) {
        SEGM_WAVE_LDS_SYNC();                                     // the previous chunk's gathers are done
        tn_park<T, kTnPA, kTnMB>(la, lane, ra[set]);
        tn_park<T, PB, NB>(lb, lane, rb[set]);
        SEGM_WAVE_LDS_SYNC();
        fetch(set, c + 2);                                        // in flight during this chunk's and the next chunk's gathers and MFMAs
        // fragment k slots 0 .. 3 = strip rows 4 g .., slots 4 .. 7 = rows 16 + 4 g .. (the same assignment in both operands)
        frag8 bf[NT_];
#pragma unroll
        for (int nt = 0; nt < NT_; ++nt) {
            if (nt < nt_n) {
                const T* p0 = lb + (4 * g + (i16 >> 2)) * PB + 16 * nt + 4 * (i16 & 3);
                bf[nt] = tr16_fragment<frag8>(p0, p0 + 16 * PB);
            }
        }
#pragma unroll
        for (int mt = 0; mt < kTnMT; ++mt) {
            if (mt < mt_n) {
                const T* p0 = la + (4 * g + (i16 >> 2)) * kTnPA + 16 * mt + 4 * (i16 & 3);
                const frag8 af = tr16_fragment<frag8>(p0, p0 + 16 * kTnPA);
#pragma unroll
                for (int nt = 0; nt < NT_; ++nt)
                    if (nt < nt_n) acc[mt][nt] = Mfma16<T>::run(af, bf[nt], acc[mt][nt]);
            }
        }
    };
    fetch(0, c_begin);
    fetch(1, c_begin + 1);
    for (int64_t c = c_begin; c < c_end; c += 2) {
        chunk(0, c);
        if (c + 1 < c_end) chunk(1, c + 1);
    }
    // partial [kw][m16][n16]: D[m = 16 mt + 4 g + r][n = 16 nt + i16]
    float* pp = P.part + (int64_t)kw * P.m16 * P.n16;
#pragma unroll
    for (int mt = 0; mt < kTnMT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT_; ++nt)
            if (mt < mt_n && nt < nt_n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pp[(int64_t)(m0 + 16 * mt + 4 * g + r) * P.n16 + n0 + 16 * nt + i16] = acc[mt][nt][r];
}

// ------------------------------------------------------------------------------------------------------
// NT (unit stride along k in both operands)
// ------------------------------------------------------------------------------------------------------
template <typename T, int MT, int NT>
__global__ void __launch_bounds__(kGwWaves * 64) wgemm_nt_kernel(GemmDev P) {
    typedef typename Mfma16<T>::v8 frag8;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int kw = blockIdx.x * kGwWaves + wave;
    if (kw >= P.kwaves) return;
    const T* A = reinterpret_cast<const T*>(P.a);
    const T* B = reinterpret_cast<const T*>(P.b);
    const int64_t per_batch = P.k / 32;

    wg_f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = wg_f32x4{0.f, 0.f, 0.f, 0.f};

    // rows beyond m / n read row 0 and are zeroed
    int64_t arow[MT], brow[NT];
    bool alive[MT], blive[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { alive[mt] = 16 * mt + i16 < P.m; arow[mt] = (int64_t)(alive[mt] ? 16 * mt + i16 : 0) * P.a_sr + 8 * g; }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { blive[nt] = 16 * nt + i16 < P.n; brow[nt] = (int64_t)(blive[nt] ? 16 * nt + i16 : 0) * P.b_sr + 8 * g; }

    const int64_t c_begin = (int64_t)kw * P.chunks_per_wave;
    const int64_t c_end = c_begin + P.chunks_per_wave < P.chunks ? c_begin + P.chunks_per_wave : P.chunks;
    wg_u32x4 fa[2][MT], fb[2][NT];
    auto fetch = [&](int buf, int64_t c) {
        const int64_t bi = c / per_batch, k0 = (c - bi * per_batch) * 32;
        const T* ab = A + bi * P.a_sb + k0;
        const T* bb = B + bi * P.b_sb + k0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fa[buf][mt] = *reinterpret_cast<const wg_u32x4*>(ab + arow[mt]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) fb[buf][nt] = *reinterpret_cast<const wg_u32x4*>(bb + brow[nt]);
    };
    auto consume = [&](int buf) {
        const wg_u32x4 zero = {0u, 0u, 0u, 0u};
        frag8 bf[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = __builtin_bit_cast(frag8, blive[nt] ? fb[buf][nt] : zero);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const frag8 af = __builtin_bit_cast(frag8, alive[mt] ? fa[buf][mt] : zero);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = Mfma16<T>::run(af, bf[nt], acc[mt][nt]);
        }
    };
    if (c_begin < c_end) fetch(0, c_begin);
    int64_t c = c_begin;
    for (; c + 1 < c_end; c += 2) {                               // two chunks per trip: both register sets addressed statically
        fetch(1, c + 1);
        consume(0);
        if (c + 2 < c_end) fetch(0, c + 2);
        consume(1);
    }
    if (c < c_end) consume(0);

    float* pp = P.part + (int64_t)kw * P.m16 * P.n16;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                pp[(int64_t)(16 * mt + 4 * g + r) * P.n16 + 16 * nt + i16] = acc[mt][nt][r];
}

// out[m][n] = sum over waves of part[w][m][n].  A launch has only m * n outputs (a few thousand) but up to 2048 partials each:
// 32 outputs x 32 slices of the wave range per workgroup, eight loads in flight per thread, slices combined in order through
// LDS - a fixed summation order (bitwise repeatable) without a 2048-long dependent chain per thread.
constexpr int kRedOut = 32, kRedSlices = 32;
__global__ void __launch_bounds__(kRedOut * kRedSlices) wgemm_reduce_kernel(const float* __restrict__ part, int waves, int m, int n, int m16,
                                                                             int n16, float* __restrict__ out) {
    __shared__ float s_sum[kRedSlices][kRedOut];
    const int o = threadIdx.x % kRedOut, sl = threadIdx.x / kRedOut;
    const int64_t i = (int64_t)blockIdx.x * kRedOut + o;
    const bool live = i < (int64_t)m * n;
    const int64_t ii = live ? i : 0;
    const int r = (int)(ii / n), c = (int)(ii - (int64_t)r * n);
    const int64_t stride = (int64_t)m16 * n16;
    const float* p = part + (int64_t)r * n16 + c;
    const int per = (waves + kRedSlices - 1) / kRedSlices;
    const int w0 = sl * per, w1 = w0 + per < waves ? w0 + per : waves;
    float acc = 0.f;
    int w = w0;
    for (; w + 8 <= w1; w += 8) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = p[(int64_t)(w + q) * stride];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += v[q];
    }
    for (; w < w1; ++w) acc += p[(int64_t)w * stride];
    s_sum[sl][o] = acc;
    __syncthreads();
    if (sl == 0 && live) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < kRedSlices; ++q) t += s_sum[q][o];
        out[i] = t;
    }
}

// out[i] = sum over parts of part[p * n + i], parts added in a fixed order (shared with stem.hip)
void launch_partial_sum(const float* part, int parts, int64_t n, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(wgemm_reduce_kernel, dim3((unsigned)((n + kRedOut - 1) / kRedOut)), dim3(kRedOut * kRedSlices), 0, stream, part, parts, 1,
                       (int)n, 1, (int)n, out);
}

struct GemmPlan { int32_t kwaves, m16, n16, mb, nb; int64_t chunks, chunks_per_wave; };
static GemmPlan gemm_plan(int layout, int m, int n, int64_t k, int batch) {
    GemmPlan pl;
    pl.m16 = (m + 15) / 16 * 16; pl.n16 = (n + 15) / 16 * 16;
    if (layout == SEGM_WGEMM_NT) {           // the NT kernel is instantiated for 1, 3 or 6 tiles per operand
        const int mt = pl.m16 / 16, nt = pl.n16 / 16;
        pl.m16 = 16 * (mt <= 1 ? 1 : (mt <= 3 ? 3 : 6));
        pl.n16 = 16 * (nt <= 1 ? 1 : (nt <= 3 ? 3 : 6));
    }
    pl.mb = layout == SEGM_WGEMM_TN ? (m + kTnMB - 1) / kTnMB : 1;
    pl.nb = layout == SEGM_WGEMM_TN ? (n <= 48 ? 1 : (n + kTnNB - 1) / kTnNB) : 1;
    pl.chunks = layout == SEGM_WGEMM_TN ? (k + 31) / 32 : (int64_t)batch * (k / 32);
    // ~2048 waves on the chip (8 per CU), at least 8 chunks each, and at most 32 MB of partials
    int64_t want = 2048 / ((int64_t)pl.mb * pl.nb);
    const int64_t cap = ((int64_t)32 << 20) / ((int64_t)pl.m16 * pl.n16 * 4);
    if (want > cap) want = cap;
    if (want > pl.chunks / 8) want = pl.chunks / 8;
    if (want < 1) want = 1;
    pl.chunks_per_wave = (pl.chunks + want - 1) / want;
    pl.kwaves = (int32_t)((pl.chunks + pl.chunks_per_wave - 1) / pl.chunks_per_wave);
    return pl;
}

}  // namespace segm

using namespace segm;

static int wgemm_check(const segm_wgrad_gemm_args* a) {
    if (!a) return SEGM_E_NULL;
    if (a->layout != SEGM_WGEMM_TN && a->layout != SEGM_WGEMM_NT) return SEGM_E_SHAPE;
    if (a->m <= 0 || a->n <= 0 || a->k <= 0) return SEGM_E_SHAPE;
    if (a->dtype != SEGM_BF16 && a->dtype != SEGM_F16) return SEGM_E_DTYPE;
    if (a->layout == SEGM_WGEMM_NT) {
        if (a->batch <= 0 || a->m > 16 * kNtMax || a->n > 16 * kNtMax || a->k % 32) return SEGM_E_SHAPE;
        if (a->a_stride_row % 8 || a->b_stride_row % 8 || a->a_stride_batch % 8 || a->b_stride_batch % 8) return SEGM_E_SHAPE;
    } else {
        if (a->m > 1024 || a->n > 1024 || a->a_stride_row < a->m || a->b_stride_row < a->n) return SEGM_E_SHAPE;
    }
    return SEGM_OK;
}

extern "C" size_t segm_wgrad_gemm_workspace_bytes(int32_t layout, int32_t m, int32_t n, int64_t k, int32_t batch) {
    if (m <= 0 || n <= 0 || k <= 0 || (layout != SEGM_WGEMM_TN && layout != SEGM_WGEMM_NT)) return 0;
    const GemmPlan pl = gemm_plan(layout, m, n, k, batch < 1 ? 1 : batch);
    return (size_t)pl.kwaves * pl.m16 * pl.n16 * sizeof(float);
}

extern "C" int segm_wgrad_gemm(const segm_wgrad_gemm_args* a) {
    int rc = wgemm_check(a);
    if (rc != SEGM_OK) return rc;
    if (!a->a || !a->b || !a->out) return SEGM_E_NULL;
    if (a->layout == SEGM_WGEMM_NT && (((uintptr_t)a->a & 15) || ((uintptr_t)a->b & 15))) return SEGM_E_SHAPE;
    const GemmPlan pl = gemm_plan(a->layout, a->m, a->n, a->k, a->batch);
    const size_t need = (size_t)pl.kwaves * pl.m16 * pl.n16 * sizeof(float);
    if (!a->workspace || a->workspace_bytes < need) return SEGM_E_WORKSPACE;
    GemmDev P;
    memset(&P, 0, sizeof(P));
    P.a = (const char*)a->a; P.a_sr = a->a_stride_row; P.a_sb = a->a_stride_batch;
    P.b = (const char*)a->b; P.b_sr = a->b_stride_row; P.b_sb = a->b_stride_batch;
    P.part = (float*)a->workspace;
    P.m = a->m; P.n = a->n; P.m16 = pl.m16; P.n16 = pl.n16; P.k = a->k; P.batch = a->batch;
    P.chunks = pl.chunks; P.kwaves = pl.kwaves; P.chunks_per_wave = pl.chunks_per_wave;
    hipStream_t st = (hipStream_t)a->stream;
    const bool f16 = a->dtype == SEGM_F16;
    const dim3 block(kGwWaves * 64);
    const unsigned gx = (unsigned)((pl.kwaves + kGwWaves - 1) / kGwWaves);
    if (a->layout == SEGM_WGEMM_TN) {
        auto mode = [](const void* p, int64_t stride, int cols) {
            if (stride % 8 != 0 || ((uintptr_t)p & 15) != 0) return 0;
            if (cols % 8 == 0) return 1;
            return (cols + 7) / 8 * 8 <= stride ? 2 : 0;
        };
        const int ma = mode(a->a, a->a_stride_row, a->m), mb_ = mode(a->b, a->b_stride_row, a->n);
        const dim3 grid(gx, pl.mb, pl.nb);
#define SEGM_TN3(TT, NB_, MA_)                                                                                   \
        do {                                                                                                     \
            if (mb_ == 1) hipLaunchKernelGGL((wgemm_tn_kernel<TT, NB_, MA_, 1>), grid, block, 0, st, P);          \
            else if (mb_ == 2) hipLaunchKernelGGL((wgemm_tn_kernel<TT, NB_, MA_, 2>), grid, block, 0, st, P);     \
            else hipLaunchKernelGGL((wgemm_tn_kernel<TT, NB_, MA_, 0>), grid, block, 0, st, P);                   \
        } while (0)
#define SEGM_TN(TT, NB_)                                                                                         \
        do {                                                                                                     \
            if (ma == 1) SEGM_TN3(TT, NB_, 1); else if (ma == 2) SEGM_TN3(TT, NB_, 2); else SEGM_TN3(TT, NB_, 0);  \
        } while (0)
        if (a->n <= 48) { if (f16) SEGM_TN(f16_t, 48); else SEGM_TN(bf16_t, 48); }
        else { if (f16) SEGM_TN(f16_t, 96); else SEGM_TN(bf16_t, 96); }
#undef SEGM_TN3
#undef SEGM_TN
    } else {
        const int mc = pl.m16 / 16, nc = pl.n16 / 16;          // 1, 3 or 6 (gemm_plan)
        const dim3 grid(gx);
#define SEGM_NT(TT, MM, NN) hipLaunchKernelGGL((wgemm_nt_kernel<TT, MM, NN>), grid, block, 0, st, P)
#define SEGM_NT_N(TT, MM) do { if (nc == 1) SEGM_NT(TT, MM, 1); else if (nc == 3) SEGM_NT(TT, MM, 3); else SEGM_NT(TT, MM, 6); } while (0)
#define SEGM_NT_M(TT) do { if (mc == 1) SEGM_NT_N(TT, 1); else if (mc == 3) SEGM_NT_N(TT, 3); else SEGM_NT_N(TT, 6); } while (0)
        if (f16) SEGM_NT_M(f16_t); else SEGM_NT_M(bf16_t);
#undef SEGM_NT_M
#undef SEGM_NT_N
#undef SEGM_NT
    }
    const int64_t total = (int64_t)a->m * a->n;
    hipLaunchKernelGGL(wgemm_reduce_kernel, dim3((unsigned)((total + kRedOut - 1) / kRedOut)), dim3(kRedOut * kRedSlices), 0, st, P.part,
                       pl.kwaves, a->m, a->n, P.m16, P.n16, a->out);
    return (int)hipGetLastError();
}
