// Causal depthwise conv1d (+SiLU), forward and backward (C ABI: segm_causal_conv1d_fwd / _bwd).
//
// Replaces reference causal-conv1d/csrc/causal_conv1d_fwd.cu:39-130 and causal_conv1d_bwd.cu:46-240
// (+ host causal_conv1d.cpp:130-268).  The reference assigns 128 threads to one (batch, channel) row and
// exchanges halo elements through shared memory; here a lane owns a channel and walks a chunk of logical
// time with the last width-1 inputs in registers, so there is no halo traffic at all and - with
// channel-last tensors - every access of a wave is one contiguous row segment.  Both kernels walk time
// upwards: the backward is written as "at index i: finish dy'[i], then emit dx[i-(W-1)]", which needs only
// the last W values of x and dy' (see conv1d_bwd_kernel).
//
//   o_t = bias + sum_w weight[d, w] * x_{t-(W-1-w)} ,  out = o * sigmoid(o)              (forward)
//   dy'_t = dout_t * silu'(o_t) ;  dx_t = sum_w weight[d, w] * dy'_{t+(W-1-w)} ;
//   dweight[d, w] = sum_{b,t} dy'_t x_{t-(W-1-w)} ;  dbias[d] = sum_{b,t} dy'_t            (backward)
#include <string.h>

#include "scan_common.h"

namespace segm {

struct ConvDev {
    Geom gm;             // nstate unused
    TimeMap tm;
    Seq x, out, dout, dx;
    const float* weight; // (dim, W)
    const float* bias;   // (dim) or null
    int32_t silu;
    float* part;         // backward: [batch][nchunks][W + 1][dim] partial dweight / dbias
};
struct ConvDevN { ConvDev d[kMaxDirs]; };


template <typename T, int W, int TS>
__global__ void __launch_bounds__(kBlock) conv1d_fwd_kernel(ConvDevN PP) {
    const ConvDev& P = PP.d[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Geom& gm = P.gm;
    const TimeMap tm = P.tm;
    const Item it = locate(gm, (int64_t)blockIdx.x * kWavesPerBlock + wave, lane);
    const int ub = uniform_batch(it);
    const bool item_ok = it.wave_valid && it.chunk < gm.nchunks;

    float wt[W];
#pragma unroll
    for (int w = 0; w < W; ++w) wt[w] = it.valid ? P.weight[(int64_t)it.d * W + w] : 0.f;
    const float bias = (it.valid && P.bias) ? P.bias[it.d] : 0.f;
    const RowPtr xp = make_rowptr<T>(P.x, ub, it.d, it.valid);
    const RowPtr op = make_rowptr<T>(P.out, ub, it.d, it.valid);

    const int32_t tau_begin = item_ok ? it.chunk * gm.chunk : 0;
    TimeIter ti;
    ti.seek(tm, tau_begin);

    // the W-1 inputs before the chunk: xh[k] = x[tau_begin - (W-1) + k]
    float xh[W - 1];
#pragma unroll
    for (int k = 0; k < W - 1; ++k) {
        const int back = (W - 1) - k;
        TimeIter tb = ti;
        tb.jump(tm, -back);
        const bool ok = it.valid && tau_begin - back >= 0;
        const float v = ld_row<T>(xp, ok ? tb.t : 0);
        xh[k] = ok ? v : 0.f;
    }

    float nx[TS];
    int32_t ntt[TS];
    uint32_t nok = row_indices<TS>(ntt, tm, ti, it.valid);
    fetch_rows<T, TS>(nx, xp, ntt, nok);
    for (int s0 = 0; s0 < gm.chunk; s0 += TS) {
        float cx[TS];
        int32_t ctt[TS];
#pragma unroll
        for (int j = 0; j < TS; ++j) { cx[j] = nx[j]; ctt[j] = ntt[j]; }
        const uint32_t cok = nok;
        ti.jump(tm, TS);
        nok = row_indices<TS>(ntt, tm, ti, it.valid);
        fetch_rows<T, TS>(nx, xp, ntt, nok);
#pragma unroll
        for (int j = 0; j < TS; ++j) {
            float o = fmaf(wt[W - 1], cx[j], bias);
#pragma unroll
            for (int k = 0; k < W - 1; ++k) o = fmaf(wt[k], xh[k], o);
#pragma unroll
            for (int k = 0; k + 1 < W - 1; ++k) xh[k] = xh[k + 1];
            xh[W - 2] = cx[j];
            if (P.silu) o = o * sigmoidf(o);
            if ((cok >> j) & 1u) st_row<T>(op, ctt[j], o);
        }
    }
}

// Backward.  Walks i = tau_begin .. tau_begin + chunk + W - 2.  At index i (all windows hold the last W values):
//   o_i   = bias + sum_k wt[k] * x[i-(W-1-k)]                      (x window)
//   g_i   = dout_i * silu'(o_i)                                     (0 for i >= L)
//   if i in this chunk:  dweight[k] += g_i * x[i-(W-1-k)],  dbias += g_i
//   tau = i-(W-1):  dx_tau = sum_k wt[k] * g_{tau+(W-1-k)} = sum_k wt[k] * gw[k]   with gw[k] = g_{i-k}... see below
template <typename T, int W, int TS>
__global__ void __launch_bounds__(kBlock) conv1d_bwd_kernel(ConvDevN PP) {
    const ConvDev& P = PP.d[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Geom& gm = P.gm;
    const TimeMap tm = P.tm;
    const Item it = locate(gm, (int64_t)blockIdx.x * kWavesPerBlock + wave, lane);
    const int ub = uniform_batch(it);
    const bool item_ok = it.wave_valid && it.chunk < gm.nchunks;

    float wt[W];
#pragma unroll
    for (int w = 0; w < W; ++w) wt[w] = it.valid ? P.weight[(int64_t)it.d * W + w] : 0.f;
    const float bias = (it.valid && P.bias) ? P.bias[it.d] : 0.f;
    const RowPtr xp = make_rowptr<T>(P.x, ub, it.d, it.valid);
    const RowPtr gp = make_rowptr<T>(P.dout, ub, it.d, it.valid);
    const RowPtr dxp = make_rowptr<T>(P.dx, ub, it.d, it.valid);

    const int32_t tau_begin = item_ok ? it.chunk * gm.chunk : 0;
    const int32_t tau_end = tau_begin + gm.chunk;          // exclusive; may exceed L
    TimeIter ti;
    ti.seek(tm, tau_begin);

    float xh[W - 1];                                        // xh[k] = x[i - (W-1) + k] before consuming x[i]
#pragma unroll
    for (int k = 0; k < W - 1; ++k) {
        const int back = (W - 1) - k;
        TimeIter tb = ti;
        tb.jump(tm, -back);
        const bool ok = it.valid && tau_begin - back >= 0;
        const float v = ld_row<T>(xp, ok ? tb.t : 0);
        xh[k] = ok ? v : 0.f;
    }
    float gh[W - 1];                                        // gh[k] = g[i - (W-1) + k]
    int32_t th[W - 1];                                      // physical row of step i - (W-1) + k
#pragma unroll
    for (int k = 0; k < W - 1; ++k) { gh[k] = 0.f; th[k] = 0; }
    float dw[W], db = 0.f;
#pragma unroll
    for (int w = 0; w < W; ++w) dw[w] = 0.f;

    float nx[TS], ng[TS];
    int32_t ntt[TS];
    uint32_t nok = row_indices<TS>(ntt, tm, ti, it.valid);
    fetch_rows<T, TS>(nx, xp, ntt, nok);
    fetch_rows<T, TS>(ng, gp, ntt, nok);
    // one extra sub-tile covers the W-1 indices past the chunk whose dy' the last dx of the chunk need
    for (int s0 = 0; s0 < gm.chunk + TS; s0 += TS) {
        float cx[TS], cg[TS];
        int32_t ctt[TS];
#pragma unroll
        for (int j = 0; j < TS; ++j) { cx[j] = nx[j]; cg[j] = ng[j]; ctt[j] = ntt[j]; }
        const uint32_t cok = nok;
        const int32_t i0 = ti.tau;
        ti.jump(tm, TS);
        nok = row_indices<TS>(ntt, tm, ti, it.valid);
        fetch_rows<T, TS>(nx, xp, ntt, nok);
        fetch_rows<T, TS>(ng, gp, ntt, nok);
#pragma unroll
        for (int j = 0; j < TS; ++j) {
            const int32_t i = i0 + j;
            float g = cg[j];                                 // already 0 for i >= L or channel-less lanes
            if (P.silu) {
                float o = fmaf(wt[W - 1], cx[j], bias);
#pragma unroll
                for (int k = 0; k < W - 1; ++k) o = fmaf(wt[k], xh[k], o);
                const float s = sigmoidf(o);
                g *= s * fmaf(o, 1.f - s, 1.f);
            }
            if (i < tau_end) {                               // wave-uniform per work item
                dw[W - 1] = fmaf(g, cx[j], dw[W - 1]);
#pragma unroll
                for (int k = 0; k < W - 1; ++k) dw[k] = fmaf(g, xh[k], dw[k]);
                db += g;
            }
            // dx at tau = i-(W-1): weight[k] pairs with g[tau + (W-1-k)] = g[i-k] -> wt[W-1] with g[tau], wt[0] with g[i]
            float dxv = wt[0] * g;
#pragma unroll
            for (int k = 1; k < W; ++k) dxv = fmaf(wt[k], gh[(W - 1) - k], dxv);
            const int32_t tau = i - (W - 1);
            if (it.valid && tau >= tau_begin && tau < tau_end && tau < tm.L) st_row<T>(dxp, th[0], dxv);
#pragma unroll
            for (int k = 0; k + 1 < W - 1; ++k) { xh[k] = xh[k + 1]; gh[k] = gh[k + 1]; th[k] = th[k + 1]; }
            xh[W - 2] = cx[j];
            gh[W - 2] = g;
            th[W - 2] = ctt[j];
            (void)cok;
        }
    }
    if (it.valid) {
        const int64_t row = ((int64_t)it.b * gm.nchunks + it.chunk) * (W + 1);
#pragma unroll
        for (int w = 0; w < W; ++w) P.part[(row + w) * gm.dim + it.d] = dw[w];
        P.part[(row + W) * gm.dim + it.d] = db;
    }
}

// ------------------------------------------------------------------------------------------------------
// out0[d*K0 + k] = sum_rows part[row][k][d] (k < K0) ; out1[d] = ... k == K0 ; out2[d] = ... k == K0+1
// grid (ceil(dim/64), K), block 16 waves: wave w sums rows w, w+16, ... then a 16-entry LDS fold.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void reduce_partials_body(const float* __restrict__ part, int64_t nrows, int K, int dim, float* out0, int K0,
                                                     float* out1, float* out2) {
    __shared__ float s_acc[kCarrySegs][64];
    const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + lane, k = blockIdx.y;
    const bool valid = d < dim;
    float acc = 0.f;
    if (valid) {
        // eight rows in flight per lane, added in row order (the sum is the same as a plain loop's, bit for bit)
        constexpr int U = 8;
        int64_t r = seg;
        for (; r + (U - 1) * kCarrySegs < nrows; r += U * kCarrySegs) {
            float v[U];
#pragma unroll
            for (int i = 0; i < U; ++i) v[i] = part[((r + i * kCarrySegs) * K + k) * dim + d];
#pragma unroll
            for (int i = 0; i < U; ++i) acc += v[i];
        }
        for (; r < nrows; r += kCarrySegs) acc += part[(r * K + k) * dim + d];
    }
    s_acc[seg][lane] = acc;
    __syncthreads();
    if (seg == 0 && valid) {
        float t = 0.f;
        for (int s = 0; s < kCarrySegs; ++s) t += s_acc[s][lane];
        if (k < K0) { if (out0) out0[(int64_t)d * K0 + k] = t; }
        else if (k == K0) { if (out1) out1[d] = t; }
        else if (out2) out2[d] = t;
    }
}
__global__ void __launch_bounds__(kCarrySegs * 64) reduce_partials_kernel(const float* __restrict__ part, int64_t nrows,
                                                                          int K, int dim, float* out0, int K0,
                                                                          float* out1, float* out2) {
    reduce_partials_body(part, nrows, K, dim, out0, K0, out1, out2);
}
// the same sums for up to four problems of one geometry in ONE launch (blockIdx.z: the directions of a Mamba layer - their partials
// come out of one scan / conv1d launch too; round 5: 48 launches of ~5 us less per step)
__global__ void __launch_bounds__(kCarrySegs * 64) reduce_partials_multi_kernel(ReduceN R, int64_t nrows, int K, int dim, int K0) {
    const int z = blockIdx.z;
    reduce_partials_body(R.part[z], nrows, K, dim, R.out0[z], K0, R.out1[z], R.out2[z]);
}

void launch_reduce_partials(const float* part, int64_t nrows, int K, int dim, float* out0, int K0, float* out1,
                            float* out2, hipStream_t stream) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((dim + 63) / 64, K), dim3(kCarrySegs * 64), 0, stream, part, nrows, K,
                       dim, out0, K0, out1, out2);
}
void launch_reduce_partials_multi(const ReduceN& R, int n, int64_t nrows, int K, int dim, int K0, hipStream_t stream) {
    hipLaunchKernelGGL(reduce_partials_multi_kernel, dim3((dim + 63) / 64, K, n), dim3(kCarrySegs * 64), 0, stream, R, nrows, K, dim, K0);
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
static int validate_conv(const segm_conv1d_args* a, bool bwd) {
    if (!a) return SEGM_E_NULL;
    if (a->batch <= 0 || a->dim <= 0 || a->seqlen <= 0 || a->seqlen >= ((int64_t)1 << 31)) return SEGM_E_SHAPE;
    if (a->width < 2 || a->width > 4) return SEGM_E_WIDTH;
    if (a->dtype != SEGM_F32 && a->dtype != SEGM_F16 && a->dtype != SEGM_BF16) return SEGM_E_DTYPE;
    if (a->time_order < SEGM_TIME_FORWARD || a->time_order > SEGM_TIME_INTERLEAVED) return SEGM_E_TIME_ORDER;
    if (a->time_order == SEGM_TIME_INTERLEAVED && (a->nslices <= 0 || a->nslices > 4096 || a->seqlen % a->nslices != 0)) return SEGM_E_SHAPE;
    if (!a->x.ptr || !a->weight) return SEGM_E_NULL;
    if (!bwd && !a->out.ptr) return SEGM_E_NULL;
    if (bwd && (!a->dout.ptr || !a->dx.ptr || !a->dweight)) return SEGM_E_NULL;
    const segm_seq* v[4] = {&a->x, &a->out, &a->dout, &a->dx};
    for (const segm_seq* s : v)
        if (s->ptr && (s->stride_t < 0 || s->stride_t >= ((int64_t)1 << 31))) return SEGM_E_SHAPE;
    return SEGM_OK;
}

static int conv_chunk(int batch, int dim, int64_t L) {
    int c = default_chunk(batch, dim, L);
    return c > 1024 ? 1024 : c;
}

static void fill_conv_dev(ConvDev& P, const segm_conv1d_args* a) {
    memset(&P, 0, sizeof(P));
    P.gm = make_geom(a->batch, a->dim, 1, a->seqlen, conv_chunk(a->batch, a->dim, a->seqlen));
    P.tm = make_timemap(a->time_order, a->nslices, a->seqlen);
    P.x = make_seq(a->x); P.out = make_seq(a->out); P.dout = make_seq(a->dout); P.dx = make_seq(a->dx);
    P.weight = a->weight; P.bias = a->bias; P.silu = a->silu;
}

// `ndir` argument blocks of ONE geometry (the three directions of a Mamba v3 layer) as one grid, blockIdx.y = direction
template <typename T, int W>
static int launch_conv(const ConvDevN& PP, int ndir, bool bwd, hipStream_t stream) {
    constexpr int TS = 8;
    const unsigned nblocks = (unsigned)((PP.d[0].gm.nwaves + kWavesPerBlock - 1) / kWavesPerBlock);
    if (bwd) hipLaunchKernelGGL((conv1d_bwd_kernel<T, W, TS>), dim3(nblocks, ndir), dim3(kBlock), 0, stream, PP);
    else hipLaunchKernelGGL((conv1d_fwd_kernel<T, W, TS>), dim3(nblocks, ndir), dim3(kBlock), 0, stream, PP);
    return (int)hipGetLastError();
}

template <typename T>
static int launch_conv_w(const ConvDevN& PP, int ndir, int width, bool bwd, hipStream_t stream) {
    if (width == 2) return launch_conv<T, 2>(PP, ndir, bwd, stream);
    if (width == 3) return launch_conv<T, 3>(PP, ndir, bwd, stream);
    return launch_conv<T, 4>(PP, ndir, bwd, stream);
}

static int launch_conv_t(const ConvDevN& PP, int ndir, int dtype, int width, bool bwd, hipStream_t stream) {
    if (dtype == SEGM_F32) return launch_conv_w<float>(PP, ndir, width, bwd, stream);
    if (dtype == SEGM_F16) return launch_conv_w<f16_t>(PP, ndir, width, bwd, stream);
    return launch_conv_w<bf16_t>(PP, ndir, width, bwd, stream);
}
static bool conv_same_launch(const segm_conv1d_args* a, const segm_conv1d_args* b) {
    return a->batch == b->batch && a->dim == b->dim && a->width == b->width && a->seqlen == b->seqlen && a->dtype == b->dtype &&
           a->stream == b->stream;
}

}  // namespace segm

using namespace segm;

extern "C" size_t segm_causal_conv1d_bwd_workspace_bytes(int32_t batch, int32_t dim, int32_t width, int64_t seqlen) {
    if (batch <= 0 || dim <= 0 || width <= 0 || seqlen <= 0) return 0;
    const int chunk = conv_chunk(batch, dim, seqlen);
    const int64_t nch = (seqlen + chunk - 1) / chunk;
    return align256((size_t)batch * nch * (width + 1) * dim * sizeof(float));
}

// `n` launches: one grid with a direction axis when they share geometry / dtype / stream (n <= 3), else one after the other
static int conv_multi(const segm_conv1d_args* args, int32_t n, bool bwd) {
    if (!args || n <= 0) return SEGM_E_NULL;
    for (int i = 0; i < n; ++i) {
        const int rc = validate_conv(&args[i], bwd);
        if (rc != SEGM_OK) return rc;
        if (bwd) {
            const size_t need = segm_causal_conv1d_bwd_workspace_bytes(args[i].batch, args[i].dim, args[i].width, args[i].seqlen);
            if (!args[i].workspace || args[i].workspace_bytes < need) return SEGM_E_WORKSPACE;
        }
    }
    int i = 0;
    while (i < n) {
        int m = 1;
        while (i + m < n && m < kMaxDirs && conv_same_launch(&args[i], &args[i + m])) ++m;
        ConvDevN PP;
        memset(&PP, 0, sizeof(PP));
        for (int k = 0; k < m; ++k) {
            fill_conv_dev(PP.d[k], &args[i + k]);
            if (bwd) PP.d[k].part = (float*)args[i + k].workspace;
        }
        hipStream_t stream = (hipStream_t)args[i].stream;
        const int rc = launch_conv_t(PP, m, args[i].dtype, args[i].width, bwd, stream);
        if (rc != 0) return rc;
        if (bwd) {                                         // (conv_same_launch: one geometry - one reduce launch for the m problems)
            ReduceN R;
            memset(&R, 0, sizeof(R));
            for (int k = 0; k < m; ++k) {
                R.part[k] = PP.d[k].part; R.out0[k] = args[i + k].dweight; R.out1[k] = args[i + k].dbias;
            }
            const segm_conv1d_args* a = &args[i];
            launch_reduce_partials_multi(R, m, (int64_t)a->batch * PP.d[0].gm.nchunks, a->width + 1, a->dim, a->width, stream);
        }
        i += m;
    }
    return (int)hipGetLastError();
}

extern "C" int segm_causal_conv1d_fwd(const segm_conv1d_args* a) { return conv_multi(a, a ? 1 : 0, false); }
extern "C" int segm_causal_conv1d_bwd(const segm_conv1d_args* a) { return conv_multi(a, a ? 1 : 0, true); }
extern "C" int segm_causal_conv1d_fwd_multi(const segm_conv1d_args* args, int32_t n) { return conv_multi(args, n, false); }
extern "C" int segm_causal_conv1d_bwd_multi(const segm_conv1d_args* args, int32_t n) { return conv_multi(args, n, true); }
