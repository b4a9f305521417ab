// InstanceNorm3d (+ residual) (+ ReLU / LeakyReLU), forward and backward (C ABI: segm_instnorm_fwd / _bwd).
//
// Replaces the torch.nn.InstanceNorm3d -> (+ residual) -> activation chains of the SegMamba stem and decoder
// (reference model_segmamba/segmamba.py:96-130 GSC, :147,169-187; monai/networks/blocks/dynunet_block.py:98-111):
// no affine parameters, no running statistics, biased variance, eps inside the square root.
//
// An instance is one (batch, channel) volume of S contiguous elements (2 M at stage 0).  Everything here is HBM
// bound byte work: 16-byte accesses, every instance cut into slabs so that the grid is a few thousand workgroups
// whatever B*C is, fp32 statistics merged with Chan's (count, mean, M2) update.
//
//   forward   K1 stats   per (instance, slab): sum, sum of squares            -> partials
//             K2 apply   merge the instance's partials -> mean, rstd; y = act((x - mean) rstd + residual)
//   backward  K3 stats   g = dy act'(v);  per (instance, slab): sum g, sum g xhat  (g is parked in dresidual if asked)
//             K4 apply   dx = rstd (g - mean(g) - xhat mean(g xhat))
// The activation mask is taken from y when a residual was added (v is not recomputable from x alone), else from xhat.
#include <stdlib.h>
#include <string.h>

#include "segm_device.h"

namespace segm {

constexpr int kNormMaxSplit = 64;

struct NormDev {
    const void* x; const void* res; void* y;
    const void* dy; const void* ymask; void* dx; void* dres;
    float* mean; float* rstd;
    float2* part;
    int64_t S;
    int64_t xs, rs, ys, dys, dxs, drs;   // elements between consecutive instances of x, residual, y (ymask), dy, dx, dres: S when
                                         // dense, larger for volumes with a padded channel stride (ops_raw.volume_empty)
    int64_t slab;          // elements per slab (multiple of 8 * 256)
    int32_t nsplit;
    int32_t act;           // 0 none, 1 relu, 2 leaky relu
    float slope, eps;
    const float4* ext;     // forward: {count, sum, sum of squares, -} partials summed by the producer of x (conv3d_fwd.hip), or null
    int32_t next;          // per instance
};

// sum over the workgroup of two values; result valid in every thread
__device__ __forceinline__ float2 block_sum2(float a, float b, float2* lds) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        a += __shfl_xor(a, m);
        b += __shfl_xor(b, m);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) lds[wave] = make_float2(a, b);
    __syncthreads();
    float2 r = make_float2(0.f, 0.f);
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) { r.x += lds[w].x; r.y += lds[w].y; }
    return r;
}

__device__ __forceinline__ void slab_range(const NormDev& P, int split, int64_t& e0, int64_t& e1) {
    e0 = (int64_t)split * P.slab;
    e1 = e0 + P.slab;
    if (e1 > P.S) e1 = P.S;
    if (e0 > P.S) e0 = P.S;
}

// merges the (sum, sumsq) partials of an instance into mean / rstd (every thread gets the result)
__device__ __forceinline__ float2 merge_stats(const NormDev& P, int inst, float2* lds) {
    if (threadIdx.x == 0) {
        float n = 0.f, mean = 0.f, m2 = 0.f;
        for (int s = 0; s < P.nsplit; ++s) {
            int64_t e0, e1;
            slab_range(P, s, e0, e1);
            const float ns = (float)(e1 - e0);
            if (ns <= 0.f) continue;
            const float2 p = P.part[(int64_t)inst * P.nsplit + s];
            const float ms = p.x / ns;
            const float m2s = fmaxf(p.y - p.x * ms, 0.f);
            const float d = ms - mean, nt = n + ns;
            mean += d * (ns / nt);
            m2 += m2s + d * d * (n * ns / nt);
            n = nt;
        }
        const float var = m2 / n;
        lds[0] = make_float2(mean, 1.0f / sqrtf(var + P.eps));
    }
    __syncthreads();
    const float2 r = lds[0];
    __syncthreads();
    return r;
}

// Chan's update of (count, mean, M2) by another such triple (either may be empty)
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float on, float omean, float om2) {
    if (on <= 0.f) return;
    if (n <= 0.f) { n = on; mean = omean; m2 = om2; return; }
    const float d = omean - mean, nt = n + on;
    mean += d * (on / nt);
    m2 += om2 + d * d * (n * on / nt);
    n = nt;
}

// the same merge over partials the PRODUCER of x summed (round 5: the statistics epilogue of the chained convolution kernels;
// hundreds of small partials per instance, not <= 64 slabs): every thread folds its strided share, the waves merge by a
// butterfly, thread 0 folds the waves - a fixed order, every thread gets the result
__device__ __forceinline__ float2 merge_ext(const NormDev& P, int inst, float2* lds) {
    __shared__ float wsum[kWavesPerBlock][3];
    float n = 0.f, mean = 0.f, m2 = 0.f;
    const float4* p = P.ext + (int64_t)inst * P.next;
    for (int i = threadIdx.x; i < P.next; i += kBlock) {
        const float4 v = p[i];
        if (v.x > 0.f) {
            const float ms = v.y / v.x;
            chan_merge(n, mean, m2, v.x, ms, fmaxf(v.z - v.y * ms, 0.f));
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float on = __shfl_xor(n, m), om = __shfl_xor(mean, m), o2 = __shfl_xor(m2, m);
        // both partners must fold in the SAME order (lower lane first), or the two halves of the butterfly drift apart by rounding
        if ((threadIdx.x & m) == 0) chan_merge(n, mean, m2, on, om, o2);
        else { float a = on, b = om, c = o2; chan_merge(a, b, c, n, mean, m2); n = a; mean = b; m2 = c; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { wsum[wave][0] = n; wsum[wave][1] = mean; wsum[wave][2] = m2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tn = 0.f, tm = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) chan_merge(tn, tm, t2, wsum[w][0], wsum[w][1], wsum[w][2]);
        lds[0] = make_float2(tm, 1.0f / sqrtf(t2 / tn + P.eps));
    }
    __syncthreads();
    const float2 r = lds[0];
    __syncthreads();
    return r;
}

// NT: non-temporal accesses for what is streamed once; U: packets a thread has in flight per tensor
template <typename T, bool VEC, bool NT = false, int U = 4>
__global__ void __launch_bounds__(kBlock) inorm_fwd_stats_kernel(NormDev P) {
    using Pk = Pack<T, VEC>;
    __shared__ float2 lds[kWavesPerBlock];
    const int split = blockIdx.x, inst = blockIdx.y;
    const T* x = reinterpret_cast<const T*>(P.x) + (int64_t)inst * P.xs;
    int64_t e0, e1;
    slab_range(P, split, e0, e1);
    float s[U], q[U];
#pragma unroll
    for (int k = 0; k < U; ++k) s[k] = q[k] = 0.f;
    const int64_t step = (int64_t)kBlock * Pk::N;
    int64_t i = e0 + (int64_t)threadIdx.x * Pk::N;
    for (; i + (U - 1) * step < e1; i += U * step) {       // U independent packets in flight per thread
        Pk p[U];
#pragma unroll
        for (int k = 0; k < U; ++k) p[k].template load<NT>(x + i + k * step);
#pragma unroll
        for (int k = 0; k < U; ++k)
#pragma unroll
            for (int j = 0; j < Pk::N; ++j) { s[k] += p[k].v[j]; q[k] = fmaf(p[k].v[j], p[k].v[j], q[k]); }
    }
    for (; i < e1; i += step) {
        Pk p;
        p.template load<NT>(x + i);
#pragma unroll
        for (int j = 0; j < Pk::N; ++j) { s[0] += p.v[j]; q[0] = fmaf(p.v[j], p.v[j], q[0]); }
    }
    float ss = 0.f, qq = 0.f;
    if constexpr (U == 4) { ss = (s[0] + s[1]) + (s[2] + s[3]); qq = (q[0] + q[1]) + (q[2] + q[3]); }
    else {
#pragma unroll
        for (int k = 0; k < U; ++k) { ss += s[k]; qq += q[k]; }
    }
    const float2 r = block_sum2(ss, qq, lds);
    if (threadIdx.x == 0) P.part[(int64_t)inst * P.nsplit + split] = r;
}

__device__ __forceinline__ float act_fwd(float v, int act, float slope) {
    if (act == 0) return v;
    return v > 0.f ? v : v * slope;                      // relu: slope == 0
}

template <typename T, bool VEC, bool NT = false, int U = 2>
__global__ void __launch_bounds__(kBlock) inorm_fwd_apply_kernel(NormDev P) {
    using Pk = Pack<T, VEC>;
    __shared__ float2 lds[kWavesPerBlock];
    const int split = blockIdx.x, inst = blockIdx.y;
    const float2 st = P.ext ? merge_ext(P, inst, lds) : merge_stats(P, inst, lds);
    const float mean = st.x, rstd = st.y;
    if (split == 0 && threadIdx.x == 0) { P.mean[inst] = mean; P.rstd[inst] = rstd; }
    const T* x = reinterpret_cast<const T*>(P.x) + (int64_t)inst * P.xs;
    const T* res = P.res ? reinterpret_cast<const T*>(P.res) + (int64_t)inst * P.rs : nullptr;
    T* y = reinterpret_cast<T*>(P.y) + (int64_t)inst * P.ys;
    const float slope = P.act == 1 ? 0.f : P.slope;
    const float shift = -mean * rstd;
    int64_t e0, e1;
    slab_range(P, split, e0, e1);
    const int64_t step = (int64_t)kBlock * Pk::N;
    auto one = [&](Pk& a, const Pk& r) {
#pragma unroll
        for (int j = 0; j < Pk::N; ++j) {
            float v = fmaf(a.v[j], rstd, shift);
            if (res) v += r.v[j];
            a.v[j] = act_fwd(v, P.act, slope);
        }
    };
    int64_t i = e0 + (int64_t)threadIdx.x * Pk::N;
    for (; i + (U - 1) * step < e1; i += U * step) {
        Pk a[U], r[U];
#pragma unroll
        for (int k = 0; k < U; ++k) a[k].template load<NT>(x + i + k * step);
        if (res) {
#pragma unroll
            for (int k = 0; k < U; ++k) r[k].template load<NT>(res + i + k * step);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            one(a[k], r[k]);
            a[k].template store<NT>(y + i + k * step);
        }
    }
    for (; i < e1; i += step) {
        Pk a, r;
        a.template load<NT>(x + i);
        if (res) r.template load<NT>(res + i);
        one(a, r);
        a.template store<NT>(y + i);
    }
}

// g = dy * act'(v) for one packet
template <typename Pk>
__device__ __forceinline__ void grad_through_act(Pk& g, const Pk& xh, const Pk& ym, bool use_y, int act, float slope) {
    if (act == 0) return;
#pragma unroll
    for (int j = 0; j < Pk::N; ++j) {
        const float v = use_y ? ym.v[j] : xh.v[j];
        g.v[j] = v > 0.f ? g.v[j] : g.v[j] * slope;
    }
}

template <typename T, bool VEC, bool NT = false, int U = 2>
__global__ void __launch_bounds__(kBlock) inorm_bwd_stats_kernel(NormDev P) {
    using Pk = Pack<T, VEC>;
    __shared__ float2 lds[kWavesPerBlock];
    const int split = blockIdx.x, inst = blockIdx.y;
    const T* x = reinterpret_cast<const T*>(P.x) + (int64_t)inst * P.xs;
    const T* dy = reinterpret_cast<const T*>(P.dy) + (int64_t)inst * P.dys;
    const T* ym = P.ymask ? reinterpret_cast<const T*>(P.ymask) + (int64_t)inst * P.ys : nullptr;
    T* dres = P.dres ? reinterpret_cast<T*>(P.dres) + (int64_t)inst * P.drs : nullptr;
    const float mean = P.mean[inst], rstd = P.rstd[inst];
    const float shift = -mean * rstd;
    const float slope = P.act == 1 ? 0.f : P.slope;
    const bool use_y = ym != nullptr;
    int64_t e0, e1;
    slab_range(P, split, e0, e1);
    float sg[U], sgx[U];
#pragma unroll
    for (int k = 0; k < U; ++k) sg[k] = sgx[k] = 0.f;
    const int64_t step = (int64_t)kBlock * Pk::N;
    // (x, dy and y are read again by the apply pass: never non-temporal; the parked g is)
    auto one = [&](Pk& xa, Pk& ga, const Pk& ya, float& a, float& b, int64_t at) {
#pragma unroll
        for (int j = 0; j < Pk::N; ++j) xa.v[j] = fmaf(xa.v[j], rstd, shift);
        grad_through_act(ga, xa, ya, use_y, P.act, slope);
#pragma unroll
        for (int j = 0; j < Pk::N; ++j) { a += ga.v[j]; b = fmaf(ga.v[j], xa.v[j], b); }
        if (dres) ga.store(dres + at);
    };
    int64_t i = e0 + (int64_t)threadIdx.x * Pk::N;
    for (; i + (U - 1) * step < e1; i += U * step) {
        Pk xa[U], ga[U], ya[U];
#pragma unroll
        for (int k = 0; k < U; ++k) { xa[k].template load<NT>(x + i + k * step); ga[k].template load<NT>(dy + i + k * step); }
        if (use_y && P.act) {
#pragma unroll
            for (int k = 0; k < U; ++k) ya[k].template load<NT>(ym + i + k * step);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) one(xa[k], ga[k], ya[k], sg[k], sgx[k], i + k * step);
    }
    for (; i < e1; i += step) {
        Pk xa, ga, ya;
        xa.template load<NT>(x + i); ga.template load<NT>(dy + i);
        if (use_y && P.act) ya.template load<NT>(ym + i);
        one(xa, ga, ya, sg[0], sgx[0], i);
    }
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < U; ++k) { a += sg[k]; b += sgx[k]; }
    const float2 r = block_sum2(a, b, lds);
    if (threadIdx.x == 0) P.part[(int64_t)inst * P.nsplit + split] = r;
}

template <typename T, bool VEC, bool NT = false, int U = 1>
__global__ void __launch_bounds__(kBlock) inorm_bwd_apply_kernel(NormDev P) {
    using Pk = Pack<T, VEC>;
    __shared__ float2 lds[kWavesPerBlock];
    const int split = blockIdx.x, inst = blockIdx.y;
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
        for (int s = 0; s < P.nsplit; ++s) {
            const float2 p = P.part[(int64_t)inst * P.nsplit + s];
            a += p.x; b += p.y;
        }
        lds[0] = make_float2(a / (float)P.S, b / (float)P.S);
    }
    __syncthreads();
    const float mg = lds[0].x, mgx = lds[0].y;
    const T* x = reinterpret_cast<const T*>(P.x) + (int64_t)inst * P.xs;
    const T* dy = reinterpret_cast<const T*>(P.dy) + (int64_t)inst * P.dys;
    const T* ym = P.ymask ? reinterpret_cast<const T*>(P.ymask) + (int64_t)inst * P.ys : nullptr;
    const T* gsrc = P.dres ? reinterpret_cast<const T*>(P.dres) + (int64_t)inst * P.drs : nullptr;     // g parked by the stats pass
    T* dx = reinterpret_cast<T*>(P.dx) + (int64_t)inst * P.dxs;
    const float mean = P.mean[inst], rstd = P.rstd[inst];
    const float shift = -mean * rstd;
    const float slope = P.act == 1 ? 0.f : P.slope;
    const bool use_y = ym != nullptr;
    int64_t e0, e1;
    slab_range(P, split, e0, e1);
    const int64_t step = (int64_t)kBlock * Pk::N;
    auto one = [&](Pk& xa, Pk& ga, const Pk& ya) {
#pragma unroll
        for (int j = 0; j < Pk::N; ++j) xa.v[j] = fmaf(xa.v[j], rstd, shift);
        if (!gsrc) grad_through_act(ga, xa, ya, use_y, P.act, slope);
#pragma unroll
        for (int j = 0; j < Pk::N; ++j) ga.v[j] = rstd * (ga.v[j] - mg - xa.v[j] * mgx);
    };
    int64_t i = e0 + (int64_t)threadIdx.x * Pk::N;
    for (; i + (U - 1) * step < e1; i += U * step) {
        Pk xa[U], ga[U], ya[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            xa[k].template load<NT>(x + i + k * step);
            if (gsrc) ga[k].load(gsrc + i + k * step);         // (the residual's gradient: read again by its consumer)
            else ga[k].template load<NT>(dy + i + k * step);
        }
        if (!gsrc && use_y && P.act) {
#pragma unroll
            for (int k = 0; k < U; ++k) ya[k].template load<NT>(ym + i + k * step);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            one(xa[k], ga[k], ya[k]);
            ga[k].template store<NT>(dx + i + k * step);
        }
    }
    if constexpr (U > 1) {
        for (; i < e1; i += step) {
            Pk xa, ga, ya;
            xa.template load<NT>(x + i);
            if (gsrc) ga.load(gsrc + i);
            else {
                ga.template load<NT>(dy + i);
                if (use_y && P.act) ya.template load<NT>(ym + i);
            }
            one(xa, ga, ya);
            ga.template store<NT>(dx + i);
        }
    }
}

// A/B knobs of these streaming passes (round 5; profiles/r05_inorm_tune.log), read per call
static int norm_knob(const char* name, int dflt) {
    const char* e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}

static void norm_plan(int instances, int64_t S, int vecn, int& nsplit, int64_t& slab) {
    const int64_t quantum = (int64_t)kBlock * vecn * 4;           // the stats loop's full stride
    const int target = norm_knob("SEGM_NORM_WGS", 4096);
    int64_t want = (target + instances - 1) / instances;          // ~`target` workgroups in total
    const int64_t maxs = (S + quantum - 1) / quantum;
    if (want > maxs) want = maxs;
    if (want > kNormMaxSplit) want = kNormMaxSplit;
    if (want < 1) want = 1;
    slab = (S + want - 1) / want;
    slab = (slab + quantum - 1) / quantum * quantum;
    nsplit = (int)((S + slab - 1) / slab);
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
// an instance stride argument: 0 = dense (the spatial size); never smaller than the spatial size
static bool norm_stride(int64_t given, int64_t spatial, int64_t& out) {
    out = given ? given : spatial;
    return out >= spatial;
}

// How a pass streams, by the bytes of ONE of its tensors (profiles/r05_inorm_tune.log, _tune2.log; 2 x 48 x 128^3 bf16 = 403 MB,
// 2 x 96 x 64^3 = 101 MB):
//   mode 0  < 64 MB    plain accesses, 2 / 1 packets in flight in the apply passes (rounds 1 - 4; the 256 MB last-level cache holds it)
//   mode 1  >= 64 MB   apply passes: non-temporal loads and stores, 4 packets in flight; backward statistics 4 in flight
//   mode 2  >= 192 MB  + non-temporal loads in the backward statistics pass (nothing survives between the passes anyway).  NOT in the
//                      forward statistics pass: plain loads leave part of x in the last-level cache for the apply pass behind it
//                      (403 MB: 0.200 ms with plain statistics loads, 0.212 ms without)
// 403 MB: forward 0.243 -> 0.200 ms (6.0 TB/s), with a residual 0.313 -> 0.259, backward 0.377 -> 0.326 (6.2 TB/s).
// SEGM_NORM_NT = 0 / 1 / 2 forces a mode (A/B).
static int norm_mode(int instances, int64_t S, size_t esize) {
    const int forced = norm_knob("SEGM_NORM_NT", -1);
    if (forced >= 0) return forced > 2 ? 2 : forced;
    const double bytes = (double)instances * (double)S * (double)esize;
    return bytes >= 192e6 ? 2 : (bytes >= 64e6 ? 1 : 0);
}

template <typename T>
static int launch_norm_fwd(NormDev& P, int instances, bool vec, hipStream_t st) {
    dim3 grid(P.nsplit, instances);
    const int mode = vec ? norm_mode(instances, P.S, sizeof(T)) : 0;
    if (vec) {
        if (!P.ext) hipLaunchKernelGGL((inorm_fwd_stats_kernel<T, true>), grid, dim3(kBlock), 0, st, P);
        if (mode) hipLaunchKernelGGL((inorm_fwd_apply_kernel<T, true, true, 4>), grid, dim3(kBlock), 0, st, P);
        else hipLaunchKernelGGL((inorm_fwd_apply_kernel<T, true>), grid, dim3(kBlock), 0, st, P);
    } else {
        if (!P.ext) hipLaunchKernelGGL((inorm_fwd_stats_kernel<T, false>), grid, dim3(kBlock), 0, st, P);
        hipLaunchKernelGGL((inorm_fwd_apply_kernel<T, false>), grid, dim3(kBlock), 0, st, P);
    }
    return (int)hipGetLastError();
}

template <typename T>
static int launch_norm_bwd(NormDev& P, int instances, bool vec, hipStream_t st) {
    dim3 grid(P.nsplit, instances);
    const int mode = vec ? norm_mode(instances, P.S, sizeof(T)) : 0;
    if (vec) {
        if (mode == 2) hipLaunchKernelGGL((inorm_bwd_stats_kernel<T, true, true, 4>), grid, dim3(kBlock), 0, st, P);
        else if (mode == 1) hipLaunchKernelGGL((inorm_bwd_stats_kernel<T, true, false, 4>), grid, dim3(kBlock), 0, st, P);
        else hipLaunchKernelGGL((inorm_bwd_stats_kernel<T, true>), grid, dim3(kBlock), 0, st, P);
        if (mode) hipLaunchKernelGGL((inorm_bwd_apply_kernel<T, true, true, 4>), grid, dim3(kBlock), 0, st, P);
        else hipLaunchKernelGGL((inorm_bwd_apply_kernel<T, true>), grid, dim3(kBlock), 0, st, P);
    } else {
        hipLaunchKernelGGL((inorm_bwd_stats_kernel<T, false>), grid, dim3(kBlock), 0, st, P);
        hipLaunchKernelGGL((inorm_bwd_apply_kernel<T, false>), grid, dim3(kBlock), 0, st, P);
    }
    return (int)hipGetLastError();
}

static int vec_width(int dtype) { return dtype == SEGM_F32 ? 4 : 8; }

}  // namespace segm

using namespace segm;

extern "C" size_t segm_instnorm_workspace_bytes(int32_t instances, int64_t spatial) {
    if (instances <= 0 || spatial <= 0) return 0;
    return (size_t)instances * kNormMaxSplit * sizeof(float2);
}

static int norm_common(int32_t instances, int64_t spatial, int32_t dtype, int32_t act, const void* ws, size_t ws_bytes) {
    if (instances <= 0 || spatial <= 0) return SEGM_E_SHAPE;
    if (dtype != SEGM_F32 && dtype != SEGM_F16 && dtype != SEGM_BF16) return SEGM_E_DTYPE;
    if (act < 0 || act > 2) return SEGM_E_SHAPE;
    if (!ws || ws_bytes < segm_instnorm_workspace_bytes(instances, spatial)) return SEGM_E_WORKSPACE;
    return SEGM_OK;
}

extern "C" int segm_instnorm_fwd(const segm_instnorm_fwd_args* a) {
    if (!a) return SEGM_E_NULL;
    int rc = norm_common(a->instances, a->spatial, a->dtype, a->act, a->workspace, a->workspace_bytes);
    if (rc != SEGM_OK) return rc;
    if (!a->x || !a->y || !a->mean || !a->rstd) return SEGM_E_NULL;
    NormDev P;
    memset(&P, 0, sizeof(P));
    P.x = a->x; P.res = a->residual; P.y = a->y; P.mean = a->mean; P.rstd = a->rstd;
    P.part = (float2*)a->workspace;
    P.S = a->spatial; P.act = a->act; P.slope = a->slope; P.eps = a->eps;
    if (a->stats_partials) {
        if (a->stats_nparts <= 0 || ((uintptr_t)a->stats_partials & 15)) return SEGM_E_SHAPE;
        P.ext = (const float4*)a->stats_partials; P.next = a->stats_nparts;
    }
    if (!norm_stride(a->x_instance_stride, a->spatial, P.xs) || !norm_stride(a->residual_instance_stride, a->spatial, P.rs) ||
        !norm_stride(a->y_instance_stride, a->spatial, P.ys))
        return SEGM_E_SHAPE;
    const int vn = vec_width(a->dtype);
    const bool vec = a->spatial % vn == 0 && P.xs % vn == 0 && P.rs % vn == 0 && P.ys % vn == 0 && aligned16(a->x) && aligned16(a->y) && (!a->residual || aligned16(a->residual));
    norm_plan(a->instances, a->spatial, vn, P.nsplit, P.slab);
    hipStream_t st = (hipStream_t)a->stream;
    if (a->dtype == SEGM_F32) return launch_norm_fwd<float>(P, a->instances, vec, st);
    if (a->dtype == SEGM_F16) return launch_norm_fwd<f16_t>(P, a->instances, vec, st);
    return launch_norm_fwd<bf16_t>(P, a->instances, vec, st);
}

extern "C" int segm_instnorm_bwd(const segm_instnorm_bwd_args* a) {
    if (!a) return SEGM_E_NULL;
    int rc = norm_common(a->instances, a->spatial, a->dtype, a->act, a->workspace, a->workspace_bytes);
    if (rc != SEGM_OK) return rc;
    if (!a->x || !a->dy || !a->dx || !a->mean || !a->rstd) return SEGM_E_NULL;
    NormDev P;
    memset(&P, 0, sizeof(P));
    P.x = a->x; P.dy = a->dy; P.ymask = a->act ? a->y : nullptr; P.dx = a->dx; P.dres = a->dresidual;
    P.mean = (float*)a->mean; P.rstd = (float*)a->rstd;
    P.part = (float2*)a->workspace;
    P.S = a->spatial; P.act = a->act; P.slope = a->slope; P.eps = 0.f;
    if (!norm_stride(a->x_instance_stride, a->spatial, P.xs) || !norm_stride(a->dy_instance_stride, a->spatial, P.dys) ||
        !norm_stride(a->y_instance_stride, a->spatial, P.ys) || !norm_stride(a->dx_instance_stride, a->spatial, P.dxs) ||
        !norm_stride(a->dresidual_instance_stride, a->spatial, P.drs))
        return SEGM_E_SHAPE;
    const int vn = vec_width(a->dtype);
    const bool vec = a->spatial % vn == 0 && P.xs % vn == 0 && P.dys % vn == 0 && P.ys % vn == 0 && P.dxs % vn == 0 && P.drs % vn == 0 &&
                     aligned16(a->x) && aligned16(a->dy) && aligned16(a->dx) &&
                     (!P.ymask || aligned16(P.ymask)) && (!a->dresidual || aligned16(a->dresidual));
    norm_plan(a->instances, a->spatial, vn, P.nsplit, P.slab);
    hipStream_t st = (hipStream_t)a->stream;
    if (a->dtype == SEGM_F32) return launch_norm_bwd<float>(P, a->instances, vec, st);
    if (a->dtype == SEGM_F16) return launch_norm_bwd<f16_t>(P, a->instances, vec, st);
    return launch_norm_bwd<bf16_t>(P, a->instances, vec, st);
}
