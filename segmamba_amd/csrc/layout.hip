// Channel-first <-> channel-last moves of the token tensors around a Mamba layer (C ABI: segm_transpose_add).
//
// The reference flattens (B, C, D, H, W) to tokens with `x.reshape(B, C, n).transpose(-1, -2)` and back with
// `.transpose(-1, -2).reshape(B, C, *dims)` + skip (model_segmamba/segmamba.py:60-75); both are transposing copies
// (the scan kernels want channel-last rows, the convolutions channel-first volumes).  ATen's strided copy moves them
// at ~0.3 TB/s (2.8 ms for 2 x 48 x 128^3 bf16); this is the LDS-tiled transpose: 64 x 64 tiles, 16-byte global
// accesses on both sides, conflict-free column reads (odd dword pitch), optional fused add of a tensor already in
// the output layout (the skip connection).
#include "segm_device.h"

namespace segm {

constexpr int kTile = 64;

struct TransDev {
    const void* in; const void* add; void* out;
    int32_t rows, cols;          // in: (batch, rows, cols) ; out: (batch, cols, rows)
    int32_t tiles_c;
};

template <typename T, bool VEC>
__global__ void __launch_bounds__(kBlock) transpose_add_kernel(TransDev P) {
    using Pk = Pack<T, VEC>;
    constexpr int N = Pk::N;
    constexpr int PITCH = kTile + (sizeof(T) == 2 ? 2 : 1);       // odd number of dwords per tile row
    __shared__ T tile[kTile][PITCH];
    const int tc = blockIdx.x % P.tiles_c, tr = blockIdx.x / P.tiles_c, b = blockIdx.y;
    const int r0 = tr * kTile, c0 = tc * kTile;
    const int64_t plane = (int64_t)P.rows * P.cols;
    const T* in = reinterpret_cast<const T*>(P.in) + (int64_t)b * plane;
    const T* add = P.add ? reinterpret_cast<const T*>(P.add) + (int64_t)b * plane : nullptr;
    T* out = reinterpret_cast<T*>(P.out) + (int64_t)b * plane;
    constexpr int PKR = kTile / N;                                  // packets per tile row
    for (int p = threadIdx.x; p < kTile * PKR; p += kBlock) {
        const int r = p / PKR, c = (p - r * PKR) * N;
        if (r0 + r < P.rows && c0 + c < P.cols) {
            Pk v;
            v.load(in + (int64_t)(r0 + r) * P.cols + c0 + c);
#pragma unroll
            for (int i = 0; i < N; ++i) tile[r][c + i] = from_f32<T>(v.v[i]);
        }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < kTile * PKR; p += kBlock) {
        const int c = p % kTile, r = (p / kTile) * N;               // lanes walk c: conflict-free column reads
        if (c0 + c < P.cols && r0 + r < P.rows) {
            Pk v;
#pragma unroll
            for (int i = 0; i < N; ++i) v.v[i] = to_f32(tile[r + i][c]);
            const int64_t o = (int64_t)(c0 + c) * P.rows + r0 + r;
            if (add) {
                Pk a;
                a.load(add + o);
#pragma unroll
                for (int i = 0; i < N; ++i) v.v[i] += a.v[i];
            }
            v.store(out + o);
        }
    }
}

template <typename T>
static int launch_transpose(const TransDev& P, int batch, bool vec, hipStream_t st) {
    const int tiles_r = (P.rows + kTile - 1) / kTile;
    dim3 grid((unsigned)(tiles_r * P.tiles_c), (unsigned)batch);
    if (vec) hipLaunchKernelGGL((transpose_add_kernel<T, true>), grid, dim3(kBlock), 0, st, P);
    else hipLaunchKernelGGL((transpose_add_kernel<T, false>), grid, dim3(kBlock), 0, st, P);
    return (int)hipGetLastError();
}

}  // namespace segm

using namespace segm;

extern "C" int segm_transpose_add(const segm_transpose_args* a) {
    if (!a) return SEGM_E_NULL;
    if (a->batch <= 0 || a->rows <= 0 || a->cols <= 0 || a->batch > 65535) return SEGM_E_SHAPE;
    if (a->dtype != SEGM_F32 && a->dtype != SEGM_F16 && a->dtype != SEGM_BF16) return SEGM_E_DTYPE;
    if (!a->in || !a->out) return SEGM_E_NULL;
    TransDev P;
    P.in = a->in; P.add = a->add; P.out = a->out;
    P.rows = a->rows; P.cols = a->cols;
    P.tiles_c = (a->cols + kTile - 1) / kTile;
    if ((int64_t)P.tiles_c * ((a->rows + kTile - 1) / kTile) >= ((int64_t)1 << 31)) return SEGM_E_SHAPE;
    const int vn = a->dtype == SEGM_F32 ? 4 : 8;
    auto al = [](const void* p) { return p == nullptr || ((uintptr_t)p & 15) == 0; };
    const bool vec = a->rows % vn == 0 && a->cols % vn == 0 && al(a->in) && al(a->out) && al(a->add);
    hipStream_t st = (hipStream_t)a->stream;
    if (a->dtype == SEGM_F32) return launch_transpose<float>(P, a->batch, vec, st);
    if (a->dtype == SEGM_F16) return launch_transpose<f16_t>(P, a->batch, vec, st);
    return launch_transpose<bf16_t>(P, a->batch, vec, st);
}
