// Volume -> tokens with LayerNorm, forward and backward (C ABI: segm_layernorm_tokens_fwd / _bwd).
//
// A Mamba layer of SegMamba does `x.reshape(B, C, n).transpose(-1, -2)` -> `nn.LayerNorm(C)` before the mixer (reference
// model_segmamba/segmamba.py:60-66).  Done with library pieces that is a transposing copy, a LayerNorm that (under
// autocast) writes fp32 and a cast back to bf16: four full passes over a 400 MB tensor at stage 0, and three more ATen
// kernels in the backward.  Here the normalisation happens inside the transpose tile:
//
//   forward   x (B, C, S) channel-first -> LDS tile [C][64 tokens] -> per-token mean / rstd over C (two passes over the
//             tile: no E[x^2] - mean^2 cancellation) -> y (B, S, C) = (x - mean) rstd gamma + beta, coalesced 16-byte
//             stores; mean / rstd (B, S) fp32 are kept for the backward.
//   backward  dy (B, S, C), x (B, C, S) -> tiles of both in LDS -> m1 = mean_c(gamma dy), m2 = mean_c(gamma dy xhat)
//             -> dx (B, C, S) = rstd (gamma dy - m1 - xhat m2), coalesced channel-first stores; every workgroup walks
//             several tiles and leaves one partial (dgamma, dbeta) row, summed in a fixed order by reduce_partials.
// One read and one write of the tensor per direction (plus dy in the backward): HBM bound.
#include <string.h>

#include "segm_device.h"

namespace segm {

void launch_reduce_partials(const float* part, int64_t nrows, int K, int dim, float* out0, int K0, float* out1,
                            float* out2, hipStream_t stream);                  // conv1d.hip

constexpr int kLnTok = 64;                   // tokens per tile
constexpr int kLnPitch = kLnTok + 2;         // odd number of dwords per tile row for 2-byte elements

struct LnDev {
    const void* x; void* y; const float* gamma; const float* beta; float* mean; float* rstd;
    const void* dy; void* dx; float* part;
    int32_t B, C;
    int64_t S;
    float eps;
    int32_t tiles_per_wg, ntiles;            // backward: tiles of one batch element walked by a workgroup
};

// x[b][c][s0 .. s0+63] -> tile[c][.]   (16-byte global reads along s; tokens beyond S read as zero)
template <typename T, int CMAX>
__device__ __forceinline__ void ln_load_cf(T (&tile)[CMAX][kLnPitch], const T* xb, int C, int64_t S, int64_t s0) {
    constexpr int N = Vec<T>::N;
    constexpr int PKR = kLnTok / N;
    for (int id = threadIdx.x; id < C * PKR; id += kBlock) {
        const int c = id / PKR, s = (id - c * PKR) * N;
        Pack<T, true> v;
        if (s0 + s < S) v.load(xb + (int64_t)c * S + s0 + s);
        else {
#pragma unroll
            for (int i = 0; i < N; ++i) v.v[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < N; ++i) tile[c][s + i] = from_f32<T>(v.v[i]);
    }
}

// sum over the channels of `f(c, s)` for every token of the tile; result for token s in every thread with that s
template <typename F>
__device__ __forceinline__ float2 ln_token_sums(float2 (&red)[kWavesPerBlock][kLnTok], int C, F f) {
    const int s = threadIdx.x & (kLnTok - 1), part = threadIdx.x >> 6;
    float a = 0.f, b = 0.f;
    for (int c = part; c < C; c += kWavesPerBlock) {
        const float2 v = f(c, s);
        a += v.x; b += v.y;
    }
    __syncthreads();
    red[part][s] = make_float2(a, b);
    __syncthreads();
    float2 r = make_float2(0.f, 0.f);
#pragma unroll
    for (int p = 0; p < kWavesPerBlock; ++p) { r.x += red[p][s].x; r.y += red[p][s].y; }
    return r;
}

template <typename T, int CMAX>
__global__ void __launch_bounds__(kBlock) ln_tokens_fwd_kernel(LnDev P) {
    constexpr int N = Vec<T>::N;
    __shared__ T tile[CMAX][kLnPitch];
    __shared__ float2 red[kWavesPerBlock][kLnTok];
    __shared__ float2 stat[kLnTok];
    const int b = blockIdx.y;
    const int64_t s0 = (int64_t)blockIdx.x * kLnTok;
    const int C = P.C;
    ln_load_cf<T, CMAX>(tile, reinterpret_cast<const T*>(P.x) + (int64_t)b * C * P.S, C, P.S, s0);
    __syncthreads();
    const float inv_c = 1.0f / (float)C;
    const float2 s1 = ln_token_sums(red, C, [&](int c, int s) { return make_float2(to_f32(tile[c][s]), 0.f); });
    const float mean = s1.x * inv_c;
    const float2 s2 = ln_token_sums(red, C, [&](int c, int s) {
        const float d = to_f32(tile[c][s]) - mean;
        return make_float2(d * d, 0.f);
    });
    const float rstd = 1.0f / sqrtf(s2.x * inv_c + P.eps);
    if (threadIdx.x < kLnTok) {
        stat[threadIdx.x] = make_float2(mean, rstd);
        if (s0 + threadIdx.x < P.S) {
            P.mean[(int64_t)b * P.S + s0 + threadIdx.x] = mean;
            P.rstd[(int64_t)b * P.S + s0 + threadIdx.x] = rstd;
        }
    }
    __syncthreads();
    // y[b][s0 + s][c0 .. c0 + N): consecutive threads write consecutive 16-byte packets of the (64 tokens x C) block
    T* yb = reinterpret_cast<T*>(P.y) + ((int64_t)b * P.S + s0) * C;
    const int cg = C / N;
    for (int id = threadIdx.x; id < kLnTok * cg; id += kBlock) {
        const int s = id / cg, c0 = (id - s * cg) * N;
        if (s0 + s >= P.S) continue;
        const float2 st = stat[s];
        Pack<T, true> o;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float xh = (to_f32(tile[c0 + i][s]) - st.x) * st.y;
            o.v[i] = fmaf(xh, P.gamma[c0 + i], P.beta[c0 + i]);
        }
        o.store(yb + (int64_t)s * C + c0);
    }
}

template <typename T, int CMAX>
__global__ void __launch_bounds__(kBlock) ln_tokens_bwd_kernel(LnDev P) {
    constexpr int N = Vec<T>::N;
    constexpr int PKR = kLnTok / N;
    __shared__ T tx[CMAX][kLnPitch];                     // x, then xhat is formed on the fly
    __shared__ T tg[CMAX][kLnPitch];                     // dy, transposed to [c][s]
    __shared__ float2 red[kWavesPerBlock][kLnTok];
    __shared__ float2 stat[kLnTok];                      // mean, rstd
    __shared__ float2 msum[kLnTok];                      // m1, m2
    __shared__ float gam[CMAX];
    const int b = blockIdx.y;
    const int C = P.C;
    const int cg = C / N;
    for (int c = threadIdx.x; c < C; c += kBlock) gam[c] = P.gamma[c];
    const T* xb = reinterpret_cast<const T*>(P.x) + (int64_t)b * C * P.S;
    T* dxb = reinterpret_cast<T*>(P.dx) + (int64_t)b * C * P.S;
    float acc_g[(CMAX + kBlock - 1) / kBlock], acc_b[(CMAX + kBlock - 1) / kBlock];
#pragma unroll
    for (int k = 0; k < (CMAX + kBlock - 1) / kBlock; ++k) { acc_g[k] = 0.f; acc_b[k] = 0.f; }
    const float inv_c = 1.0f / (float)C;

    const int t_begin = blockIdx.x * P.tiles_per_wg;
    const int t_end = (t_begin + P.tiles_per_wg < P.ntiles) ? t_begin + P.tiles_per_wg : P.ntiles;
    for (int t = t_begin; t < t_end; ++t) {
        const int64_t s0 = (int64_t)t * kLnTok;
        __syncthreads();                                  // the previous tile is consumed
        ln_load_cf<T, CMAX>(tx, xb, C, P.S, s0);
        const T* dyb = reinterpret_cast<const T*>(P.dy) + ((int64_t)b * P.S + s0) * C;
        for (int id = threadIdx.x; id < kLnTok * cg; id += kBlock) {
            const int s = id / cg, c0 = (id - s * cg) * N;
            Pack<T, true> v;
            if (s0 + s < P.S) v.load(dyb + (int64_t)s * C + c0);
            else {
#pragma unroll
                for (int i = 0; i < N; ++i) v.v[i] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < N; ++i) tg[c0 + i][s] = from_f32<T>(v.v[i]);
        }
        if (threadIdx.x < kLnTok) {
            const bool in = s0 + threadIdx.x < P.S;
            stat[threadIdx.x] = make_float2(in ? P.mean[(int64_t)b * P.S + s0 + threadIdx.x] : 0.f,
                                            in ? P.rstd[(int64_t)b * P.S + s0 + threadIdx.x] : 0.f);
        }
        __syncthreads();
        const float2 m = ln_token_sums(red, C, [&](int c, int s) {
            const float g = gam[c] * to_f32(tg[c][s]);
            const float xh = (to_f32(tx[c][s]) - stat[s].x) * stat[s].y;
            return make_float2(g, g * xh);
        });
        if (threadIdx.x < kLnTok) msum[threadIdx.x] = make_float2(m.x * inv_c, m.y * inv_c);
        __syncthreads();
        // dx[b][c][s0 + 8 g ..): channel-first packets
        for (int id = threadIdx.x; id < C * PKR; id += kBlock) {
            const int c = id / PKR, s = (id - c * PKR) * N;
            if (s0 + s >= P.S) continue;
            Pack<T, true> o;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const float2 st = stat[s + i], mm = msum[s + i];
                const float xh = (to_f32(tx[c][s + i]) - st.x) * st.y;
                o.v[i] = st.y * (gam[c] * to_f32(tg[c][s + i]) - mm.x - xh * mm.y);
            }
            o.store(dxb + (int64_t)c * P.S + s0 + s);
        }
        // dgamma / dbeta contributions of this tile: one thread per channel (tokens beyond S hold dy = 0)
#pragma unroll
        for (int k = 0; k < (CMAX + kBlock - 1) / kBlock; ++k) {
            const int c = threadIdx.x + k * kBlock;
            if (c < C) {
                float ag = 0.f, ab = 0.f;
                for (int s = 0; s < kLnTok; ++s) {
                    const float d = to_f32(tg[c][s]);
                    ag = fmaf(d, (to_f32(tx[c][s]) - stat[s].x) * stat[s].y, ag);
                    ab += d;
                }
                acc_g[k] += ag; acc_b[k] += ab;
            }
        }
    }
    float* prow = P.part + ((int64_t)b * gridDim.x + blockIdx.x) * 2 * C;
#pragma unroll
    for (int k = 0; k < (CMAX + kBlock - 1) / kBlock; ++k) {
        const int c = threadIdx.x + k * kBlock;
        if (c < C) { prow[c] = acc_g[k]; prow[C + c] = acc_b[k]; }
    }
}

static int ln_wgs_per_batch(int64_t S, int* tiles_per_wg) {
    const int64_t ntiles = (S + kLnTok - 1) / kLnTok;
    int tpw = (int)((ntiles + 2047) / 2048);              // <= 2048 partial rows per batch element
    if (tpw < 1) tpw = 1;
    *tiles_per_wg = tpw;
    return (int)((ntiles + tpw - 1) / tpw);
}

static int ln_check(int32_t batch, int32_t channels, int64_t spatial, int32_t dtype) {
    if (batch <= 0 || channels <= 0 || spatial <= 0 || batch > 65535) return SEGM_E_SHAPE;
    if (dtype != SEGM_F32 && dtype != SEGM_F16 && dtype != SEGM_BF16) return SEGM_E_DTYPE;
    const int n = dtype == SEGM_F32 ? 4 : 8;
    if (channels % n != 0 || spatial % n != 0) return SEGM_E_SHAPE;
    if (channels > (dtype == SEGM_F32 ? 192 : 384)) return SEGM_E_SHAPE;      // the tile must fit LDS
    return SEGM_OK;
}

template <typename T>
static void ln_launch(const LnDev& P, bool bwd, dim3 grid, hipStream_t st) {
    if (P.C <= 96) {
        if (bwd) hipLaunchKernelGGL((ln_tokens_bwd_kernel<T, 96>), grid, dim3(kBlock), 0, st, P);
        else hipLaunchKernelGGL((ln_tokens_fwd_kernel<T, 96>), grid, dim3(kBlock), 0, st, P);
    } else {
        constexpr int CM = sizeof(T) == 4 ? 192 : 384;
        if (bwd) hipLaunchKernelGGL((ln_tokens_bwd_kernel<T, CM>), grid, dim3(kBlock), 0, st, P);
        else hipLaunchKernelGGL((ln_tokens_fwd_kernel<T, CM>), grid, dim3(kBlock), 0, st, P);
    }
}

}  // namespace segm

using namespace segm;

extern "C" size_t segm_layernorm_tokens_workspace_bytes(int32_t batch, int32_t channels, int64_t spatial) {
    if (batch <= 0 || channels <= 0 || spatial <= 0) return 0;
    int tpw;
    const int wgs = ln_wgs_per_batch(spatial, &tpw);
    return (size_t)batch * wgs * 2 * channels * sizeof(float);
}

extern "C" int segm_layernorm_tokens_fwd(const segm_layernorm_args* a) {
    if (!a) return SEGM_E_NULL;
    int rc = ln_check(a->batch, a->channels, a->spatial, a->dtype);
    if (rc != SEGM_OK) return rc;
    if (!a->x || !a->y || !a->gamma || !a->beta || !a->mean || !a->rstd) return SEGM_E_NULL;
    LnDev P;
    memset(&P, 0, sizeof(P));
    P.x = a->x; P.y = a->y; P.gamma = a->gamma; P.beta = a->beta; P.mean = a->mean; P.rstd = a->rstd;
    P.B = a->batch; P.C = a->channels; P.S = a->spatial; P.eps = a->eps;
    const dim3 grid((unsigned)((a->spatial + kLnTok - 1) / kLnTok), (unsigned)a->batch);
    hipStream_t st = (hipStream_t)a->stream;
    if (a->dtype == SEGM_F32) ln_launch<float>(P, false, grid, st);
    else if (a->dtype == SEGM_F16) ln_launch<f16_t>(P, false, grid, st);
    else ln_launch<bf16_t>(P, false, grid, st);
    return (int)hipGetLastError();
}

extern "C" int segm_layernorm_tokens_bwd(const segm_layernorm_args* a) {
    if (!a) return SEGM_E_NULL;
    int rc = ln_check(a->batch, a->channels, a->spatial, a->dtype);
    if (rc != SEGM_OK) return rc;
    if (!a->x || !a->dy || !a->dx || !a->gamma || !a->mean || !a->rstd || !a->dgamma || !a->dbeta) return SEGM_E_NULL;
    if (!a->workspace || a->workspace_bytes < segm_layernorm_tokens_workspace_bytes(a->batch, a->channels, a->spatial))
        return SEGM_E_WORKSPACE;
    LnDev P;
    memset(&P, 0, sizeof(P));
    P.x = a->x; P.dy = a->dy; P.dx = a->dx; P.gamma = a->gamma; P.mean = a->mean; P.rstd = a->rstd;
    P.part = (float*)a->workspace;
    P.B = a->batch; P.C = a->channels; P.S = a->spatial;
    P.ntiles = (int32_t)((a->spatial + kLnTok - 1) / kLnTok);
    const int wgs = ln_wgs_per_batch(a->spatial, &P.tiles_per_wg);
    const dim3 grid((unsigned)wgs, (unsigned)a->batch);
    hipStream_t st = (hipStream_t)a->stream;
    if (a->dtype == SEGM_F32) ln_launch<float>(P, true, grid, st);
    else if (a->dtype == SEGM_F16) ln_launch<f16_t>(P, true, grid, st);
    else ln_launch<bf16_t>(P, true, grid, st);
    // part rows: [batch * wgs][2][C]  ->  dgamma (k = 0), dbeta (k = 1)
    launch_reduce_partials(P.part, (int64_t)a->batch * wgs, 2, a->channels, nullptr, 0, a->dgamma, a->dbeta, st);
    return (int)hipGetLastError();
}
