// out[c] (fp32) = sum over batch and voxels of x[b][c][s]  (C ABI: segm_channel_sum).
//
// The bias gradient of every convolution that has a bias - `dy.sum((0, 2, 3, 4))` where the reference leaves it to cuDNN's
// backward-bias / ATen (GSC.proj ... proj4, MlpChannel.fc1 / fc2, the stem and the output convolution; segmamba.py:95-131,
// 78-89, 141, 254).  ATen's generic reduction reads these (B, C, S) tensors at 0.3 - 0.7 TB/s (profiles/r03_copy_shapes.log: 0.16 ms
// for 50 MB); it is a plain HBM-bound stream: a workgroup sums one segment of one (b, c) row with 16-byte loads and four
// independent partial sums per thread, leaves one partial per (b, segment, c), and reduce_partials adds them in a fixed order
// (deterministic, no atomics).  Rows may carry a channel stride (padded 128^3 volumes); the voxel stride is 1.
#include <stdlib.h>
#include <string.h>

#include "segm_device.h"

namespace segm {

void launch_reduce_partials(const float* part, int64_t nrows, int K, int dim, float* out0, int K0, float* out1,
                            float* out2, hipStream_t stream);

constexpr int kCsTargetBlocks = 2048;      // eight workgroups per CU
constexpr int64_t kCsMinSeg = 8192;        // elements: two passes of 256 threads x 16 bytes of 16-bit data

struct ChSumDev {
    const char* x;
    int64_t sb, sc;        // batch / channel strides in elements
    int64_t S, seg_len;
    int32_t B, C, nseg;
    float* part;           // [b][segment][c]
};

static inline void channel_sum_geometry(int32_t B, int32_t C, int64_t S, int32_t* nseg, int64_t* seg_len) {
    int64_t n = (S + kCsMinSeg - 1) / kCsMinSeg;
    const int64_t want = (kCsTargetBlocks + (int64_t)B * C - 1) / ((int64_t)B * C);
    if (n > want) n = want;
    if (n < 1) n = 1;
    int64_t len = (S + n - 1) / n;
    len = (len + 7) / 8 * 8;               // segments start on 16-byte boundaries of a 16-bit row
    *seg_len = len;
    *nseg = (int32_t)((S + len - 1) / len);
}

template <typename T>
__global__ void __launch_bounds__(kBlock) channel_sum_kernel(ChSumDev P) {
    constexpr int N = Vec<T>::N;
    __shared__ float s_w[kWavesPerBlock];
    const int c = blockIdx.x, seg = blockIdx.y, b = blockIdx.z;
    const int64_t s0 = (int64_t)seg * P.seg_len;
    const int64_t n = (s0 + P.seg_len < P.S ? s0 + P.seg_len : P.S) - s0;
    const T* row = reinterpret_cast<const T*>(P.x) + (int64_t)b * P.sb + (int64_t)c * P.sc + s0;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int64_t done = 0;
    if ((reinterpret_cast<uintptr_t>(row) & 15u) == 0) {
        const int64_t step = (int64_t)kBlock * N;
        int64_t i = (int64_t)threadIdx.x * N;
        for (; i + 3 * step + N <= n; i += 4 * step) {     // four packets in flight per thread
            Pack<T, true> p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u].load(row + i + u * step);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < N; ++e) acc[u] += p[u].v[e];
        }
        for (; i + N <= n; i += step) {
            Pack<T, true> p;
            p.load(row + i);
#pragma unroll
            for (int e = 0; e < N; ++e) acc[0] += p.v[e];
        }
        done = n / N * N;
    }
    for (int64_t i = done + threadIdx.x; i < n; i += kBlock) acc[1] += to_f32(row[i]);
    float t = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = 0.f;
        for (int w = 0; w < kWavesPerBlock; ++w) r += s_w[w];
        P.part[((int64_t)b * P.nseg + seg) * P.C + c] = r;
    }
}

}  // namespace segm

using namespace segm;

extern "C" size_t segm_channel_sum_workspace_bytes(int32_t batch, int32_t channels, int64_t spatial) {
    if (batch <= 0 || channels <= 0 || spatial <= 0) return 0;
    int32_t nseg;
    int64_t seg_len;
    channel_sum_geometry(batch, channels, spatial, &nseg, &seg_len);
    return (size_t)batch * nseg * channels * sizeof(float);
}

extern "C" int segm_channel_sum(const segm_channel_sum_args* a) {
    if (!a) return SEGM_E_NULL;
    if (!a->x || !a->out || !a->workspace) return SEGM_E_NULL;
    if (a->batch <= 0 || a->channels <= 0 || a->spatial <= 0 || a->channels > 65535 || a->batch > 65535) return SEGM_E_SHAPE;
    if (a->stride_channel < a->spatial || a->stride_batch < 0) return SEGM_E_SHAPE;
    if (a->dtype != SEGM_F32 && a->dtype != SEGM_F16 && a->dtype != SEGM_BF16) return SEGM_E_DTYPE;
    if (a->workspace_bytes < segm_channel_sum_workspace_bytes(a->batch, a->channels, a->spatial)) return SEGM_E_WORKSPACE;
    ChSumDev P;
    memset(&P, 0, sizeof(P));
    P.x = (const char*)a->x; P.sb = a->stride_batch; P.sc = a->stride_channel;
    P.S = a->spatial; P.B = a->batch; P.C = a->channels;
    channel_sum_geometry(a->batch, a->channels, a->spatial, &P.nseg, &P.seg_len);
    P.part = (float*)a->workspace;
    hipStream_t st = (hipStream_t)a->stream;
    const dim3 grid(P.C, P.nseg, P.B), block(kBlock);
    if (a->dtype == SEGM_F32) hipLaunchKernelGGL(channel_sum_kernel<float>, grid, block, 0, st, P);
    else if (a->dtype == SEGM_F16) hipLaunchKernelGGL(channel_sum_kernel<f16_t>, grid, block, 0, st, P);
    else hipLaunchKernelGGL(channel_sum_kernel<bf16_t>, grid, block, 0, st, P);
    launch_reduce_partials(P.part, (int64_t)P.B * P.nseg, 1, P.C, a->out, 1, nullptr, nullptr, st);
    return (int)hipGetLastError();
}
