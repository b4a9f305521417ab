// Row-streaming projection  y[m, :] = x[m, :] W^T + bias  for tall activations (C ABI: segm_linear_rows).
//
// Replaces the `nn.Linear` / `F.linear` calls of the Mamba block on (batch * length) x channels activations - in_proj,
// x_proj, out_proj and their data gradients (reference mamba/mamba_ssm/modules/mamba_simple.py:204-208, 264;
// selective_scan_interface.py:185-205, 247-275) - which the reference hands to cuBLAS.  With M = B L up to 524 288 rows and
// K, N between 48 and 384 these are streaming operators: 9.7 GFLOP against 250 MB at stage 0.  The BLAS library runs them at
// ~1 TB/s (profiles/r01_bench_step_kernels_v11.txt: 13 launches of 225 us where the traffic needs ~45 us).
//
//   * W (N x K, at most 192 x 192 per block of output columns) is STATIONARY in registers as MFMA A-operand fragments;
//   * a wave walks 16-row tiles of x: each lane loads its B-operand fragments straight from global memory (16 bytes of
//     one row per fragment), the next tile's loads are in flight during the current tile's MFMAs;
//   * D = W x^T, so a lane ends up with four consecutive output columns of one row, no LDS anywhere.  Round 5 (WIDE): the rows of
//     W are dealt to the A operands of a PAIR of column tiles so that the lane's 4 + 4 columns are eight consecutive ones -
//     16-byte stores (and 16-byte reads of the old values in accumulate mode) where round 2 - 4 issued two 8-byte ones: output
//     bytes dominate these operators (in_proj writes four times what it reads), and 8-byte accesses run at 0.54 - 0.70 of the
//     16-byte rate on this part (MI355X_MICROARCH.md).
// v_mfma_f32_16x16x32: A[i][k]: lane l holds A[i = l & 15][8 (l >> 4) .. +7]; B[k][j]: lane l holds B[8 (l >> 4) .. +7][j = l & 15];
// D[row = 4 (l >> 4) + r][col = l & 15].  Here i = output column n, j = row m.
#include <stdlib.h>
#include <string.h>

#include "segm_device.h"

namespace segm {

typedef float lin_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t lin_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t lin_u32x2 __attribute__((ext_vector_type(2)));

constexpr int kLinWaves = 4;
constexpr int kLinTilesPerWave = 8;          // 16-row tiles a wave walks (amortises the weight fragment loads)

struct LinDev {
    const char* x;  int64_t ldx;             // element stride between rows
    const char* w;
    const float* bias;
    char* y;        int64_t ldy;
    int64_t rows;
    int32_t k, n;
    int32_t accumulate;                      // y += x W^T (+ bias) instead of y =
};

// KC = 32-wide chunks of K held per wave; NT = 16-column tiles of W per wave (at most 24 fragments = 96 VGPRs, which
// leaves room for two to three waves per SIMD - the kernel lives on memory-level parallelism)
constexpr int lin_tiles(int kc) { return kc <= 2 ? 12 : (kc == 3 ? 8 : (kc == 4 ? 6 : 4)); }
// WIDE: tile pair p = t / 2 covers columns n0 + 32 p .. + 31; A-operand row i of tile t holds output column
// n0 + 32 p + 8 (i >> 2) + 4 (t & 1) + (i & 3), so lane group g ends up with columns 8 g .. 8 g + 3 (tile 2 p) and 8 g + 4 .. 8 g + 7
// (tile 2 p + 1) of the pair: one 16-byte store.  (y rows 16-byte aligned: ldy % 8 == 0, checked by the launcher.)
template <typename T, int KC, bool ACCUM = false, bool WIDE = false>
__global__ void __launch_bounds__(kLinWaves * 64) linear_rows_kernel(LinDev P) {
    typedef typename Mfma16<T>::v8 frag8;
    constexpr int NT = lin_tiles(KC);
    static_assert(NT % 2 == 0, "column tiles come in pairs");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.y * NT * 16;
    int nt_live = (P.n - n0 + 15) / 16;
    if (WIDE) nt_live = (nt_live + 1) & ~1;               // whole pairs (columns past n are masked)
    nt_live = nt_live > NT ? NT : nt_live;
    const T* W = reinterpret_cast<const T*>(P.w);

    frag8 wf[NT][KC];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = WIDE ? n0 + 32 * (t >> 1) + 8 * (i16 >> 2) + 4 * (t & 1) + (i16 & 3) : n0 + 16 * t + i16;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int k = 32 * c + 8 * g;
            const bool live = t < nt_live && n < P.n && k < P.k;
            const lin_u32x4 v = *reinterpret_cast<const lin_u32x4*>(W + (int64_t)(live ? n : 0) * P.k + (live ? k : 0));
            const lin_u32x4 zero = {0u, 0u, 0u, 0u};
            wf[t][c] = __builtin_bit_cast(frag8, live ? v : zero);
        }
    }

    const int64_t ntiles = (P.rows + 15) / 16;
    const int64_t first = ((int64_t)blockIdx.x * kLinWaves + wave) * kLinTilesPerWave;
    auto load_x = [&](frag8 (&dst)[KC], int64_t tile) {
        int64_t m = tile * 16 + i16;
        m = m < P.rows ? m : P.rows - 1;                     // rows past the end are read from the last row and never stored
        const T* row = reinterpret_cast<const T*>(P.x) + m * P.ldx;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int k = 32 * c + 8 * g;
            const lin_u32x4 v = *reinterpret_cast<const lin_u32x4*>(row + (k < P.k ? k : 0));
            const lin_u32x4 zero = {0u, 0u, 0u, 0u};
            dst[c] = __builtin_bit_cast(frag8, k < P.k ? v : zero);
        }
    };
    auto compute = [&](const frag8 (&xf)[KC], int64_t tile) {
        const int64_t m = tile * 16 + i16;
        const bool row_ok = tile < ntiles && m < P.rows;
        if constexpr (WIDE) {
            T* yrow = reinterpret_cast<T*>(P.y) + (row_ok ? m : 0) * P.ldy + n0 + 8 * g;
            lin_u32x4 oldv[ACCUM ? NT / 2 : 1];
            if constexpr (ACCUM) {
#pragma unroll
                for (int p = 0; p < NT / 2; ++p)
                    if (2 * p < nt_live && row_ok && n0 + 32 * p + 8 * g < P.n) oldv[p] = *reinterpret_cast<const lin_u32x4*>(yrow + 32 * p);
            }
#pragma unroll
            for (int p = 0; p < NT / 2; ++p) {
                if (2 * p < nt_live) {                       // uniform
                    const int nb = n0 + 32 * p + 8 * g;      // this lane's eight output columns of the pair (n % 8 == 0: all or none)
                    const bool col_ok = nb < P.n;
                    lin_f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
                    if (P.bias) {
                        a0 = *reinterpret_cast<const lin_f32x4*>(P.bias + (col_ok ? nb : 0));
                        a1 = *reinterpret_cast<const lin_f32x4*>(P.bias + (col_ok ? nb + 4 : 0));
                    }
#pragma unroll
                    for (int c = 0; c < KC; ++c) {
                        a0 = Mfma16<T>::run(wf[2 * p][c], xf[c], a0);
                        a1 = Mfma16<T>::run(wf[2 * p + 1][c], xf[c], a1);
                    }
                    if (row_ok && col_ok) {
                        if constexpr (ACCUM) {
                            T o[8];
                            memcpy(o, &oldv[p], 16);
#pragma unroll
                            for (int q = 0; q < 4; ++q) { a0[q] += to_f32(o[q]); a1[q] += to_f32(o[4 + q]); }
                        }
                        lin_u32x4 pk;
                        pk[0] = pack2<T>(a0[0], a0[1]);
                        pk[1] = pack2<T>(a0[2], a0[3]);
                        pk[2] = pack2<T>(a1[0], a1[1]);
                        pk[3] = pack2<T>(a1[2], a1[3]);
                        *reinterpret_cast<lin_u32x4*>(yrow + 32 * p) = pk;
                    }
                }
            }
            return;
        }
        T* yrow = reinterpret_cast<T*>(P.y) + (row_ok ? m : 0) * P.ldy + n0 + 4 * g;
        // accumulate: what y holds is fetched for ALL column tiles before the first MFMA (round 3 loaded each tile's old values
        // after its MFMAs and waited for them: one memory round trip per column tile and row tile on the critical path)
        lin_u32x2 oldv[ACCUM ? NT : 1];
        if constexpr (ACCUM) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (t < nt_live && row_ok && n0 + 16 * t + 4 * g < P.n) oldv[t] = *reinterpret_cast<const lin_u32x2*>(yrow + 16 * t);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t < nt_live) {                               // uniform
                const int nb = n0 + 16 * t + 4 * g;          // this lane's four output columns of tile t
                const bool col_ok = nb < P.n;
                lin_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if (P.bias) acc = *reinterpret_cast<const lin_f32x4*>(P.bias + (col_ok ? nb : 0));
#pragma unroll
                for (int c = 0; c < KC; ++c) acc = Mfma16<T>::run(wf[t][c], xf[c], acc);
                if (row_ok && col_ok) {
                    if constexpr (ACCUM) {
                        T o[4];
                        memcpy(o, &oldv[t], 8);
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[q] += to_f32(o[q]);
                    }
                    lin_u32x2 pk;
                    pk[0] = pack2<T>(acc[0], acc[1]);
                    pk[1] = pack2<T>(acc[2], acc[3]);
                    *reinterpret_cast<lin_u32x2*>(yrow + 16 * t) = pk;
                }
            }
        }
    };
    if (first >= ntiles) return;
    auto clampt = [&](int64_t t) { return t < ntiles ? t : ntiles - 1; };
    frag8 xa[KC], xb[KC];
    load_x(xa, first);
    for (int i = 0; i < kLinTilesPerWave; i += 2) {          // two tiles per trip: the buffers swap roles without copies
        load_x(xb, clampt(first + i + 1));                   // in flight during this tile's MFMAs
        SEGM_SCHED_FENCE();
        compute(xa, first + i);
        SEGM_SCHED_FENCE();
        load_x(xa, clampt(first + i + 2));
        SEGM_SCHED_FENCE();
        compute(xb, first + i + 1);
        SEGM_SCHED_FENCE();
    }
}

// ------------------------------------------------------------------------------------------------------
// Round 6: the same product for K > 192 (the projections of stages 2 / 3: d_inner = 384 / 768, a few thousand rows).  W no longer
// fits a wave's registers; both operands are streamed - W is a few hundred KB and stays in L2 - as 16-byte fragments straight from
// memory, 32 rows x 96 columns per wave, four k chunks in flight.  Measured (profiles/r06_linear_small_gpu.log, GPU durations): 6 - 25 us
// on the stage-2 / 3 shapes where the vendor GEMM takes 5.6 - 12.7 us - plain NN products of this size are the vendor library's home
// ground; the host routing therefore keeps them there (linear._ROWS_MIN), and this kernel only widens what the entry point accepts.
// ------------------------------------------------------------------------------------------------------
constexpr int kLkNT = 6, kLkMT = 2;
template <typename T, bool ACCUM>
__global__ void __launch_bounds__(kLinWaves * 64) linear_rows_k_kernel(LinDev P) {
    typedef typename Mfma16<T>::v8 frag8;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.y * kLkNT * 16;
    const int64_t m0 = ((int64_t)blockIdx.x * kLinWaves + wave) * (kLkMT * 16);
    if (m0 >= P.rows) return;
    const T* W = reinterpret_cast<const T*>(P.w);
    const T* X = reinterpret_cast<const T*>(P.x);
    const lin_u32x4 zero = {0u, 0u, 0u, 0u};
    // fixed per lane: the W rows (output columns) and x rows of its fragments
    int64_t wrow[kLkNT], xrow[kLkMT];
    bool wlive[kLkNT];
#pragma unroll
    for (int t = 0; t < kLkNT; ++t) { const int n = n0 + 16 * t + i16; wlive[t] = n < P.n; wrow[t] = (int64_t)(wlive[t] ? n : 0) * P.k; }
#pragma unroll
    for (int mt = 0; mt < kLkMT; ++mt) { int64_t m = m0 + 16 * mt + i16; m = m < P.rows ? m : P.rows - 1; xrow[mt] = m * P.ldx; }
    lin_f32x4 acc[kLkMT][kLkNT];
#pragma unroll
    for (int mt = 0; mt < kLkMT; ++mt)
#pragma unroll
        for (int t = 0; t < kLkNT; ++t) {
            const int nb = n0 + 16 * t + 4 * g;
            acc[mt][t] = (P.bias && nb < P.n) ? *reinterpret_cast<const lin_f32x4*>(P.bias + nb) : lin_f32x4{0.f, 0.f, 0.f, 0.f};
        }
    const int chunks = (P.k + 31) / 32;
    // FOUR chunks in flight: the operands sit in L2 (~1 us away) and a chunk's MFMAs are ~200 cycles - with one chunk of prefetch
    // the wave's time was chunks x latency (24 chunks at K = 768: no faster than the vendor call)
    constexpr int DEPTH = 4;
    lin_u32x4 wf[DEPTH][kLkNT], xf[DEPTH][kLkMT];
    auto fetch = [&](int buf, int c) {
        const int k = 32 * c + 8 * g;
        const bool kin = c < chunks && k < P.k;
        const int kk = kin ? k : 0;
#pragma unroll
        for (int t = 0; t < kLkNT; ++t) wf[buf][t] = (kin && wlive[t]) ? *reinterpret_cast<const lin_u32x4*>(W + wrow[t] + kk) : zero;
#pragma unroll
        for (int mt = 0; mt < kLkMT; ++mt) xf[buf][mt] = kin ? *reinterpret_cast<const lin_u32x4*>(X + xrow[mt] + kk) : zero;
    };
    auto consume = [&](int buf) {
#pragma unroll
        for (int t = 0; t < kLkNT; ++t)
#pragma unroll
            for (int mt = 0; mt < kLkMT; ++mt)
                acc[mt][t] = Mfma16<T>::run(__builtin_bit_cast(frag8, wf[buf][t]), __builtin_bit_cast(frag8, xf[buf][mt]), acc[mt][t]);
    };
#pragma unroll
    for (int i = 0; i < DEPTH - 1; ++i) fetch(i, i);
    for (int c = 0; c < chunks; c += DEPTH) {                    // DEPTH chunks per trip: every register set addressed statically
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            fetch((i + DEPTH - 1) % DEPTH, c + i + DEPTH - 1);
            if (c + i < chunks) consume(i);
            SEGM_SCHED_FENCE();
        }
    }
    // D = W x^T: a lane holds four consecutive output columns of one row
#pragma unroll
    for (int mt = 0; mt < kLkMT; ++mt) {
        const int64_t m = m0 + 16 * mt + i16;
        if (m >= P.rows) continue;
        T* yrow = reinterpret_cast<T*>(P.y) + m * P.ldy + n0 + 4 * g;
#pragma unroll
        for (int t = 0; t < kLkNT; ++t) {
            if (n0 + 16 * t + 4 * g >= P.n) continue;
            lin_f32x4 a = acc[mt][t];
            if constexpr (ACCUM) {
                const lin_u32x2 old = *reinterpret_cast<const lin_u32x2*>(yrow + 16 * t);
                T o[4];
                memcpy(o, &old, 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] += to_f32(o[q]);
            }
            lin_u32x2 pk;
            pk[0] = pack2<T>(a[0], a[1]);
            pk[1] = pack2<T>(a[2], a[3]);
            *reinterpret_cast<lin_u32x2*>(yrow + 16 * t) = pk;
        }
    }
}

template <typename T>
static int launch_linear_k(const LinDev& P, hipStream_t st) {
    const int64_t gx = (P.rows + kLinWaves * kLkMT * 16 - 1) / (kLinWaves * kLkMT * 16);
    if (gx >= ((int64_t)1 << 31)) return SEGM_E_SHAPE;
    const dim3 grid((unsigned)gx, (unsigned)((P.n + kLkNT * 16 - 1) / (kLkNT * 16)));
    if (P.accumulate) hipLaunchKernelGGL((linear_rows_k_kernel<T, true>), grid, dim3(kLinWaves * 64), 0, st, P);
    else hipLaunchKernelGGL((linear_rows_k_kernel<T, false>), grid, dim3(kLinWaves * 64), 0, st, P);
    return (int)hipGetLastError();
}

template <typename T>
static int launch_linear(const LinDev& P, hipStream_t st) {
    const int kc = (P.k + 31) / 32;
    const int nt = lin_tiles(kc <= 2 ? 2 : kc);
    const int64_t ntiles = (P.rows + 15) / 16;
    const int64_t gx = (ntiles + kLinWaves * kLinTilesPerWave - 1) / (kLinWaves * kLinTilesPerWave);
    if (gx >= ((int64_t)1 << 31)) return SEGM_E_SHAPE;
    const dim3 grid((unsigned)gx, (unsigned)((P.n + nt * 16 - 1) / (nt * 16)));
    const dim3 block(kLinWaves * 64);
    // 16-byte stores when every output row segment is 16-byte aligned and whole (SEGM_LINEAR_WIDE=0: the 8-byte form, A/B)
    static const bool wide_on = [] { const char* e = getenv("SEGM_LINEAR_WIDE"); return !(e && e[0] == '0'); }();
    const bool wide = wide_on && P.n % 8 == 0 && P.ldy % 8 == 0 && ((uintptr_t)P.y & 15) == 0;
#define SEGM_LIN_LAUNCH(KC_, ACC_, WIDE_) hipLaunchKernelGGL((linear_rows_kernel<T, KC_, ACC_, WIDE_>), grid, block, 0, st, P)
#define SEGM_LIN_KC(ACC_, WIDE_)                                   \
    do {                                                           \
        if (kc <= 2) SEGM_LIN_LAUNCH(2, ACC_, WIDE_);              \
        else if (kc == 3) SEGM_LIN_LAUNCH(3, ACC_, WIDE_);         \
        else if (kc == 4) SEGM_LIN_LAUNCH(4, ACC_, WIDE_);         \
        else SEGM_LIN_LAUNCH(6, ACC_, WIDE_);                      \
    } while (0)
    if (P.accumulate) {
        if (wide) SEGM_LIN_KC(true, true); else SEGM_LIN_KC(true, false);
    } else {
        if (wide) SEGM_LIN_KC(false, true); else SEGM_LIN_KC(false, false);
    }
#undef SEGM_LIN_KC
#undef SEGM_LIN_LAUNCH
    return (int)hipGetLastError();
}

}  // namespace segm

using namespace segm;

extern "C" int segm_linear_rows(const segm_linear_args* a) {
    if (!a) return SEGM_E_NULL;
    if (a->rows <= 0 || a->k <= 0 || a->n <= 0) return SEGM_E_SHAPE;
    if (a->k % 8 != 0 || a->k > 2048 || a->n % 4 != 0) return SEGM_E_SHAPE;
    if (a->dtype != SEGM_BF16 && a->dtype != SEGM_F16) return SEGM_E_DTYPE;
    if (!a->x || !a->w || !a->y) return SEGM_E_NULL;
    if (a->x_stride_row % 8 != 0 || a->x_stride_row < a->k || a->y_stride_row % 4 != 0 || a->y_stride_row < a->n) return SEGM_E_SHAPE;
    if (((uintptr_t)a->x & 15) || ((uintptr_t)a->w & 15) || ((uintptr_t)a->y & 7) || (a->bias && ((uintptr_t)a->bias & 15)))
        return SEGM_E_SHAPE;
    LinDev P;
    P.x = (const char*)a->x; P.ldx = a->x_stride_row;
    P.w = (const char*)a->w; P.bias = a->bias;
    P.y = (char*)a->y; P.ldy = a->y_stride_row;
    P.rows = a->rows; P.k = a->k; P.n = a->n;
    P.accumulate = a->accumulate != 0;
    hipStream_t st = (hipStream_t)a->stream;
    if (a->k > 192) return a->dtype == SEGM_F16 ? launch_linear_k<f16_t>(P, st) : launch_linear_k<bf16_t>(P, st);
    return a->dtype == SEGM_F16 ? launch_linear<f16_t>(P, st) : launch_linear<bf16_t>(P, st);
}
