// Training-step glue on the device (C ABI: segm_sgd_clip_step, segm_cross_entropy).
//
// Replaces, per training step of the reference loop (light_training/trainer.py:461-470, 3_train.py:51-52,57-66):
//   torch.nn.utils.clip_grad_norm_(params, 12) + torch.optim.SGD(lr, momentum 0.99, nesterov, weight decay 3e-5).step()
//     - in ATen ~45 multi-tensor launches and seven passes over the 67 M fp32 parameters / gradients / momenta
//       (profiles/r01_bench_step_kernels_v11.txt: 1.5 ms) - by one pass that reads the gradients (squared norm) and one
//       that reads gradient, momentum and parameter and writes momentum and parameter;
//   nn.CrossEntropyLoss()(pred, label) and its backward - an fp32 copy of the logits, log-softmax, NLL, two backward
//       kernels - by one pass that reads the logits and labels and writes the loss partial sums and d(loss)/d(logits).
//
// Both are pure streaming kernels (HBM roofline).  Tensor lists travel in the kernel arguments (no device-side table,
// no host -> device copy): up to kMtMax tensors per launch.
#include <string.h>

#include "segm_device.h"

namespace segm {

constexpr int kMtMax = 96;                   // tensors per launch: 96 x 32 B of pointers + block map < 4 KB of kernarg
constexpr int kMtBlock = 256;
constexpr int kMtChunk = 16384;              // elements per workgroup

struct MtTensor {
    float* p;
    const float* g;
    float* m;
    int64_t n;
};
struct MtArgs {
    MtTensor t[kMtMax];
    int32_t blk0[kMtMax + 1];                // first workgroup of tensor i inside this launch
    int32_t count;
    int32_t part0;                           // index of this launch's first workgroup in the partial-sum array
    float* partial;                          // phase 1: one squared-norm partial per workgroup
    const float* coef;                       // phase 2: {clip coefficient}
    float lr, momentum, weight_decay;
    int32_t nesterov;
};

__device__ __forceinline__ int mt_find(const MtArgs& A, int blk) {
    int lo = 0, hi = A.count - 1;            // largest i with blk0[i] <= blk
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (A.blk0[mid] <= blk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// ---- phase 1: squared L2 norm of the gradients, one partial per workgroup (summed in a fixed order afterwards) ----------
__global__ void __launch_bounds__(kMtBlock) sgd_norm_kernel(MtArgs A) {
    __shared__ float red[kMtBlock / 64];
    const int ti = __builtin_amdgcn_readfirstlane(mt_find(A, blockIdx.x));
    const MtTensor T = A.t[ti];
    const int64_t e0 = (int64_t)(blockIdx.x - A.blk0[ti]) * kMtChunk;
    const int64_t e1 = e0 + kMtChunk < T.n ? e0 + kMtChunk : T.n;
    float s = 0.f;
    if ((reinterpret_cast<uintptr_t>(T.g) & 15) == 0) {
        const int64_t nv = (e1 - e0) / 4;
        const float4* gv = reinterpret_cast<const float4*>(T.g + e0);
        for (int64_t i = threadIdx.x; i < nv; i += kMtBlock) {
            const float4 v = gv[i];
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        for (int64_t i = e0 + nv * 4 + threadIdx.x; i < e1; i += kMtBlock) s += T.g[i] * T.g[i];
    } else {
        for (int64_t i = e0 + threadIdx.x; i < e1; i += kMtBlock) s += T.g[i] * T.g[i];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < kMtBlock / 64; ++w) tot += red[w];
        A.partial[A.part0 + blockIdx.x] = tot;
    }
}

// one workgroup: norm = sqrt(sum of partials), coefficient = min(1, max_norm / (norm + 1e-6))
// (torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1.0); max_norm <= 0 = no clipping.
// With a loss scale (fp16 autocast: the gradients are S x the true ones) the step is GradScaler's unscale_ -> clip -> step
// (reference light_training/trainer.py:461-466) without a pass over the gradients: the true norm is norm / S, the coefficient
// the update applies to the SCALED gradients is clip / S, and a non-finite norm (an inf / nan anywhere in the gradients) raises
// found_inf and makes the update kernels return without touching parameters or momenta - what GradScaler.step skips.
__global__ void __launch_bounds__(kMtBlock) sgd_coef_kernel(const float* partial, int n, float max_norm, float* out,
                                                            const float* loss_scale, float* found_inf) {
    __shared__ float red[kMtBlock];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += kMtBlock) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = kMtBlock / 2; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float raw = sqrtf(red[0]);
        const float inv = loss_scale ? 1.f / loss_scale[0] : 1.f;
        const float norm = raw * inv;
        float c = 1.f;
        if (max_norm > 0.f) {
            c = max_norm / (norm + 1e-6f);
            c = c > 1.f ? 1.f : c;           // a NaN norm gives a NaN coefficient, as in torch
        }
        const bool bad = loss_scale != nullptr && !(raw <= 3.4028234e38f);      // inf or nan
        out[0] = c * inv;
        out[1] = norm;
        out[2] = bad ? 1.f : 0.f;            // the update kernels skip the step (only ever set with a loss scale)
        out[3] = 0.f;
        if (found_inf) found_inf[0] = bad ? 1.f : 0.f;
    }
}

// ---- phase 2: g' = coef g + wd p;  m = mu m + g';  p -= lr (nesterov ? g' + mu m : m) ------------------------------------
// (torch.optim.SGD with dampening 0; a zero-initialised momentum buffer reproduces its first step, buf = g')
__device__ __forceinline__ void sgd_update(float& p, float g, float& m, float c, const MtArgs& A) {
    const float gg = c * g + A.weight_decay * p;
    const float mm = A.momentum * m + gg;
    m = mm;
    p -= A.lr * (A.nesterov ? gg + A.momentum * mm : mm);
}

__global__ void __launch_bounds__(kMtBlock) sgd_update_kernel(MtArgs A) {
    const int ti = __builtin_amdgcn_readfirstlane(mt_find(A, blockIdx.x));
    const MtTensor T = A.t[ti];
    const int64_t e0 = (int64_t)(blockIdx.x - A.blk0[ti]) * kMtChunk;
    const int64_t e1 = e0 + kMtChunk < T.n ? e0 + kMtChunk : T.n;
    const float c = A.coef[0];
    if (A.coef[2] != 0.f) return;                 // inf / nan gradients under a loss scale: the step is skipped (uniform)
    const bool vec = ((reinterpret_cast<uintptr_t>(T.g) | reinterpret_cast<uintptr_t>(T.p) | reinterpret_cast<uintptr_t>(T.m)) & 15) == 0;
    int64_t tail = e0;
    if (vec) {
        const int64_t nv = (e1 - e0) / 4;
        const float4* gv = reinterpret_cast<const float4*>(T.g + e0);
        float4* pv = reinterpret_cast<float4*>(T.p + e0);
        float4* mv = reinterpret_cast<float4*>(T.m + e0);
        for (int64_t i = threadIdx.x; i < nv; i += kMtBlock) {
            const float4 g = gv[i];
            float4 p = pv[i], m = mv[i];
            sgd_update(p.x, g.x, m.x, c, A);
            sgd_update(p.y, g.y, m.y, c, A);
            sgd_update(p.z, g.z, m.z, c, A);
            sgd_update(p.w, g.w, m.w, c, A);
            pv[i] = p;
            mv[i] = m;
        }
        tail = e0 + nv * 4;
    }
    for (int64_t i = tail + threadIdx.x; i < e1; i += kMtBlock) {
        float p = T.p[i], m = T.m[i];
        sgd_update(p, T.g[i], m, c, A);
        T.p[i] = p;
        T.m[i] = m;
    }
}

// ------------------------------------------------------------------------------------------------------
// cross entropy over the channel axis of (batch, classes, spatial) logits, with d(sum of losses)/d(logits)
// ------------------------------------------------------------------------------------------------------
constexpr int kCeMaxC = 16;
constexpr int kCeBlock = 256;

struct CeDev {
    const void* logits;
    const int64_t* labels;
    void* dlogits;
    float* loss_partial;                     // (workgroups)
    float* count_partial;                    // (workgroups)
    int64_t spatial, total;                  // total = batch * spatial
    int32_t classes;
    int64_t ignore_index;
};

template <typename T>
__global__ void __launch_bounds__(kCeBlock) cross_entropy_kernel(CeDev P) {
    __shared__ float red[2][kCeBlock / 64];
    const int64_t v = (int64_t)blockIdx.x * kCeBlock + threadIdx.x;
    float loss = 0.f, cnt = 0.f;
    if (v < P.total) {
        const int64_t b = v / P.spatial, s = v - b * P.spatial;
        const T* x = reinterpret_cast<const T*>(P.logits) + b * P.classes * P.spatial + s;
        T* d = reinterpret_cast<T*>(P.dlogits) + b * P.classes * P.spatial + s;
        const int64_t lab = P.labels[v];
        float xv[kCeMaxC];
        float mx = -3.0e38f;
#pragma unroll
        for (int c = 0; c < kCeMaxC; ++c) {
            if (c < P.classes) {
                xv[c] = to_f32(x[(int64_t)c * P.spatial]);
                mx = fmaxf(mx, xv[c]);
            }
        }
        float se = 0.f, xl = 0.f;                          // xl = x[label] - max
#pragma unroll
        for (int c = 0; c < kCeMaxC; ++c) {
            if (c < P.classes) {
                const float sh = xv[c] - mx;
                if ((int64_t)c == lab) xl = sh;
                xv[c] = fast_exp(sh);
                se += xv[c];
            }
        }
        const bool valid = lab != P.ignore_index;
        // a label outside [0, classes) that is not ignore_index is a caller error (ATen asserts on the device); here it
        // poisons the loss and this voxel's gradient with NaN instead of passing as "x[label] = 0": wrong labels stay loud
        const bool oob = valid && (lab < 0 || lab >= (int64_t)P.classes);
        const float inv = oob ? __builtin_nanf("") : 1.f / se;
#pragma unroll
        for (int c = 0; c < kCeMaxC; ++c) {
            if (c < P.classes) {
                const float pr = xv[c] * inv;
                d[(int64_t)c * P.spatial] = from_f32<T>(valid ? pr - ((int64_t)c == lab ? 1.f : 0.f) : 0.f);
            }
        }
        if (valid) {
            loss = oob ? __builtin_nanf("") : fast_log(se) - xl;          // logsumexp - x[label]
            cnt = 1.f;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        loss += __shfl_xor(loss, off, 64);
        cnt += __shfl_xor(cnt, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = loss;
        red[1][threadIdx.x >> 6] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float l = 0.f, n = 0.f;
#pragma unroll
        for (int w = 0; w < kCeBlock / 64; ++w) { l += red[0][w]; n += red[1][w]; }
        P.loss_partial[blockIdx.x] = l;
        P.count_partial[blockIdx.x] = n;
    }
}

static int64_t mt_blocks(int64_t n) { return (n + kMtChunk - 1) / kMtChunk; }

}  // namespace segm

using namespace segm;

extern "C" size_t segm_sgd_clip_step_workspace_bytes(int32_t ntensors, const int64_t* numel) {
    if (ntensors <= 0 || !numel) return 0;
    int64_t blocks = 0;
    for (int i = 0; i < ntensors; ++i) blocks += numel[i] > 0 ? mt_blocks(numel[i]) : 0;
    return (size_t)(blocks + 4) * sizeof(float);
}

extern "C" int segm_sgd_clip_step(const segm_sgd_args* a) {
    if (!a) return SEGM_E_NULL;
    if (a->ntensors < 0) return SEGM_E_SHAPE;
    if (a->ntensors == 0) return SEGM_OK;
    if (!a->params || !a->grads || !a->momenta || !a->numel) return SEGM_E_NULL;
    int64_t blocks = 0;
    for (int i = 0; i < a->ntensors; ++i) {
        if (a->numel[i] < 0) return SEGM_E_SHAPE;
        if (a->numel[i] > 0 && (!a->params[i] || !a->grads[i] || !a->momenta[i])) return SEGM_E_NULL;
        blocks += mt_blocks(a->numel[i]);
    }
    if (blocks >= ((int64_t)1 << 31)) return SEGM_E_SHAPE;
    if (!a->workspace || a->workspace_bytes < (size_t)(blocks + 4) * sizeof(float)) return SEGM_E_WORKSPACE;
    hipStream_t st = (hipStream_t)a->stream;
    float* coef = (float*)a->workspace;                   // {coefficient, norm, -, -}
    float* partial = coef + 4;

    for (int phase = 0; phase < 2; ++phase) {
        int i = 0;
        int64_t part0 = 0;
        while (i < a->ntensors) {
            MtArgs A;
            memset(&A, 0, sizeof(A));
            int cnt = 0, nb = 0;
            while (i < a->ntensors && cnt < kMtMax) {
                if (a->numel[i] > 0) {
                    A.t[cnt].p = a->params[i]; A.t[cnt].g = a->grads[i]; A.t[cnt].m = a->momenta[i]; A.t[cnt].n = a->numel[i];
                    A.blk0[cnt] = nb;
                    nb += (int)mt_blocks(a->numel[i]);
                    ++cnt;
                }
                ++i;
            }
            if (cnt == 0) break;
            A.blk0[cnt] = nb;
            A.count = cnt;
            A.part0 = (int32_t)part0;
            A.partial = partial;
            A.coef = coef;
            A.lr = a->lr; A.momentum = a->momentum; A.weight_decay = a->weight_decay; A.nesterov = a->nesterov;
            if (phase == 0) hipLaunchKernelGGL(sgd_norm_kernel, dim3(nb), dim3(kMtBlock), 0, st, A);
            else hipLaunchKernelGGL(sgd_update_kernel, dim3(nb), dim3(kMtBlock), 0, st, A);
            part0 += nb;
        }
        if (phase == 0)
            hipLaunchKernelGGL(sgd_coef_kernel, dim3(1), dim3(kMtBlock), 0, st, partial, (int)blocks, a->max_norm, coef,
                               a->loss_scale, a->found_inf);
    }
    return (int)hipGetLastError();
}

extern "C" int32_t segm_cross_entropy_partials(int32_t batch, int64_t spatial) {
    if (batch <= 0 || spatial <= 0) return 0;
    return (int32_t)(((int64_t)batch * spatial + kCeBlock - 1) / kCeBlock);
}

extern "C" int segm_cross_entropy(const segm_cross_entropy_args* a) {
    if (!a) return SEGM_E_NULL;
    if (a->batch <= 0 || a->spatial <= 0 || a->classes < 1 || a->classes > kCeMaxC) return SEGM_E_SHAPE;
    if (a->dtype != SEGM_F32 && a->dtype != SEGM_F16 && a->dtype != SEGM_BF16) return SEGM_E_DTYPE;
    if (!a->logits || !a->labels || !a->dlogits || !a->loss_partial || !a->count_partial) return SEGM_E_NULL;
    const int64_t total = (int64_t)a->batch * a->spatial;
    if ((total + kCeBlock - 1) / kCeBlock >= ((int64_t)1 << 31)) return SEGM_E_SHAPE;
    CeDev P;
    P.logits = a->logits; P.labels = a->labels; P.dlogits = a->dlogits;
    P.loss_partial = a->loss_partial; P.count_partial = a->count_partial;
    P.spatial = a->spatial; P.total = total; P.classes = a->classes; P.ignore_index = a->ignore_index;
    const dim3 grid((unsigned)((total + kCeBlock - 1) / kCeBlock));
    hipStream_t st = (hipStream_t)a->stream;
    if (a->dtype == SEGM_F32) hipLaunchKernelGGL((cross_entropy_kernel<float>), grid, dim3(kCeBlock), 0, st, P);
    else if (a->dtype == SEGM_F16) hipLaunchKernelGGL((cross_entropy_kernel<f16_t>), grid, dim3(kCeBlock), 0, st, P);
    else hipLaunchKernelGGL((cross_entropy_kernel<bf16_t>), grid, dim3(kCeBlock), 0, st, P);
    return (int)hipGetLastError();
}
