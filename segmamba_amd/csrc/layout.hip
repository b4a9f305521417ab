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

// ------------------------------------------------------------------------------------------------------
// Depth-to-space / space-to-depth by 2 x 2 x 2 (C ABI: segm_depth_to_space2).
// A ConvTranspose3d with kernel_size = stride = 2 (reference unetr_block.py:52-60) is a GEMM to (B, Cout * 8, D, H, W) followed by
//     vol[b, c, 2z+i, 2y+j, 2x+k] = blk[b, c, i, j, k, z, y, x]
// which the host code left to ATen's strided copy (profiles/r04_copy_shapes.log: 0.38 ms for 2 x 48 x 128^3 = 2.1 TB/s, four such
// copies per step with the backward's inverse).  Here a thread reads 16 bytes of the k = 0 row and 16 bytes of the k = 1 row,
// interleaves the 16-bit elements with two v_perm per dword pair and writes 32 contiguous bytes of the output row (and the
// reverse for space-to-depth): every access is 16 bytes wide and contiguous over the lanes of a row on both sides.
// ------------------------------------------------------------------------------------------------------
struct D2sDev {
    char* blk; char* vol;
    int64_t sb, sc, sz, sy;      // element strides of vol
    int32_t C, D, H, W;          // of blk
    int64_t total;               // threads: B * C * 4 * D * H * (W / 8)
};
typedef uint32_t d2s_u32x4 __attribute__((ext_vector_type(4)));

template <int DIR>
__global__ void __launch_bounds__(256) depth_to_space2_kernel(D2sDev P) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= P.total) return;
    const int gw = P.W / 8;
    int64_t r = id;
    const int gx = (int)(r % gw);  r /= gw;
    const int y = (int)(r % P.H);  r /= P.H;
    const int z = (int)(r % P.D);  r /= P.D;
    const int j = (int)(r & 1), i = (int)((r >> 1) & 1);  r >>= 2;
    const int c = (int)(r % P.C);
    const int64_t b = r / P.C;
    const int64_t plane = (int64_t)P.D * P.H * P.W;
    // blk[b][c][i][j][k][z][y][x]: the k = 0 row, k = 1 one plane further
    char* bp = P.blk + ((((((b * P.C + c) * 2 + i) * 2 + j) * 2) * P.D + z) * P.H + y) * (int64_t)P.W * 2 + gx * 16;
    char* vp = P.vol + (b * P.sb + c * P.sc + (int64_t)(2 * z + i) * P.sz + (int64_t)(2 * y + j) * P.sy) * 2 + gx * 32;
    if (DIR == 0) {
        const d2s_u32x4 a = *reinterpret_cast<const d2s_u32x4*>(bp);
        const d2s_u32x4 k1 = *reinterpret_cast<const d2s_u32x4*>(bp + plane * 2);
        d2s_u32x4 lo, hi;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            lo[2 * q] = __builtin_amdgcn_perm(k1[q], a[q], 0x05040100u);
            lo[2 * q + 1] = __builtin_amdgcn_perm(k1[q], a[q], 0x07060302u);
            hi[2 * q] = __builtin_amdgcn_perm(k1[2 + q], a[2 + q], 0x05040100u);
            hi[2 * q + 1] = __builtin_amdgcn_perm(k1[2 + q], a[2 + q], 0x07060302u);
        }
        *reinterpret_cast<d2s_u32x4*>(vp) = lo;
        *reinterpret_cast<d2s_u32x4*>(vp + 16) = hi;
    } else {
        const d2s_u32x4 lo = *reinterpret_cast<const d2s_u32x4*>(vp);
        const d2s_u32x4 hi = *reinterpret_cast<const d2s_u32x4*>(vp + 16);
        d2s_u32x4 a, k1;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            a[q] = __builtin_amdgcn_perm(lo[2 * q + 1], lo[2 * q], 0x05040100u);
            k1[q] = __builtin_amdgcn_perm(lo[2 * q + 1], lo[2 * q], 0x07060302u);
            a[2 + q] = __builtin_amdgcn_perm(hi[2 * q + 1], hi[2 * q], 0x05040100u);
            k1[2 + q] = __builtin_amdgcn_perm(hi[2 * q + 1], hi[2 * q], 0x07060302u);
        }
        *reinterpret_cast<d2s_u32x4*>(bp) = a;
        *reinterpret_cast<d2s_u32x4*>(bp + plane * 2) = k1;
    }
}

// ------------------------------------------------------------------------------------------------------
// out = a + b + c (C ABI: segm_add3; round 6).  The v3 Mamba block sums its three directions twice per layer - `out + out_b + out_s`
// in front of out_proj (reference mamba_simple.py:160 / :264) and the three `dxz` contributions in the backward pass - and
// ATen does each as two binary adds: six tensor passes (and a 16-bit rounding in between) where four do.  One streaming pass, fp32
// sum, one rounding; `out` may be `a`.
// ------------------------------------------------------------------------------------------------------
struct Add3Dev { const void* a; const void* b; const void* c; void* out; int64_t npack; };

template <typename T>
__global__ void __launch_bounds__(kBlock) add3_kernel(Add3Dev P) {
    using Pk = Pack<T, true>;
    constexpr int N = Pk::N;
    const T* a = reinterpret_cast<const T*>(P.a);
    const T* b = reinterpret_cast<const T*>(P.b);
    const T* c = reinterpret_cast<const T*>(P.c);
    T* out = reinterpret_cast<T*>(P.out);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < P.npack; i += 2 * stride) {     // two packets of each in flight
        const int64_t j = i + stride;
        const bool two = j < P.npack;
        Pk va, vb, vc, wa, wb, wc;
        va.load(a + i * N); vb.load(b + i * N); vc.load(c + i * N);
        if (two) { wa.load(a + j * N); wb.load(b + j * N); wc.load(c + j * N); }
#pragma unroll
        for (int q = 0; q < N; ++q) va.v[q] = (va.v[q] + vb.v[q]) + vc.v[q];
        va.store(out + i * N);
        if (two) {
#pragma unroll
            for (int q = 0; q < N; ++q) wa.v[q] = (wa.v[q] + wb.v[q]) + wc.v[q];
            wa.store(out + j * N);
        }
    }
}

}  // namespace segm

using namespace segm;

extern "C" int segm_add3(const segm_add3_args* p) {
    if (!p) return SEGM_E_NULL;
    if (!p->a || !p->b || !p->c || !p->out) return SEGM_E_NULL;
    if (p->dtype != SEGM_F32 && p->dtype != SEGM_F16 && p->dtype != SEGM_BF16) return SEGM_E_DTYPE;
    const int n = p->dtype == SEGM_F32 ? 4 : 8;
    if (p->count <= 0 || p->count % n != 0) return SEGM_E_SHAPE;
    if (((uintptr_t)p->a | (uintptr_t)p->b | (uintptr_t)p->c | (uintptr_t)p->out) & 15) return SEGM_E_SHAPE;
    Add3Dev P{p->a, p->b, p->c, p->out, p->count / n};
    int64_t blocks = (P.npack + 2 * kBlock - 1) / (2 * kBlock);
    blocks = blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks);
    hipStream_t st = (hipStream_t)p->stream;
    if (p->dtype == SEGM_F32) hipLaunchKernelGGL((add3_kernel<float>), dim3((unsigned)blocks), dim3(kBlock), 0, st, P);
    else if (p->dtype == SEGM_F16) hipLaunchKernelGGL((add3_kernel<f16_t>), dim3((unsigned)blocks), dim3(kBlock), 0, st, P);
    else hipLaunchKernelGGL((add3_kernel<bf16_t>), dim3((unsigned)blocks), dim3(kBlock), 0, st, P);
    return (int)hipGetLastError();
}

extern "C" int segm_depth_to_space2(const segm_d2s_args* a) {
    if (!a) return SEGM_E_NULL;
    if (!a->blk || !a->vol) return SEGM_E_NULL;
    if (a->batch <= 0 || a->channels <= 0 || a->depth <= 0 || a->height <= 0 || a->width <= 0 || a->width % 8 != 0) return SEGM_E_SHAPE;
    if (a->dtype != SEGM_F16 && a->dtype != SEGM_BF16) return SEGM_E_DTYPE;
    if (a->direction != 0 && a->direction != 1) return SEGM_E_SHAPE;
    const int64_t st[4] = {a->vol_stride_b, a->vol_stride_c, a->vol_stride_z, a->vol_stride_y};
    for (int64_t s : st)
        if (s % 8 != 0 || s <= 0) return SEGM_E_SHAPE;      // 16-byte aligned rows
    if (((uintptr_t)a->blk & 15) || ((uintptr_t)a->vol & 15)) return SEGM_E_SHAPE;
    D2sDev P;
    P.blk = (char*)a->blk; P.vol = (char*)a->vol;
    P.sb = a->vol_stride_b; P.sc = a->vol_stride_c; P.sz = a->vol_stride_z; P.sy = a->vol_stride_y;
    P.C = a->channels; P.D = a->depth; P.H = a->height; P.W = a->width;
    P.total = (int64_t)a->batch * a->channels * 4 * a->depth * a->height * (a->width / 8);
    const int64_t blocks = (P.total + 255) / 256;
    if (blocks >= ((int64_t)1 << 31)) return SEGM_E_SHAPE;
    hipStream_t stm = (hipStream_t)a->stream;
    if (a->direction == 0) hipLaunchKernelGGL((depth_to_space2_kernel<0>), dim3((unsigned)blocks), dim3(256), 0, stm, P);
    else hipLaunchKernelGGL((depth_to_space2_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, stm, P);
    return (int)hipGetLastError();
}

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

// ---- out[i] = src[map(i)] for 16-bit elements (C ABI: segm_gather16; round 6) -------------------------------------------------
// The parameter bank refreshes every re-arranged copy of a weight (the convolution kernels' fragment images and packed blocks) with
// ONE gather per training step (param_bank.refresh) - torch.index_select with a 4-byte index per element there: 0.48 ms for the
// 118 M elements of the benchmarked network, index traffic twice the payload.  Every group of eight consecutive packed elements is
// eight input channels of one (output channel, tap), i.e. an arithmetic progression in the source - mode 1 stores (first index,
// step) per group: 1 byte of map per element.  A thread produces eight elements = one 16-byte store.
namespace segm {
struct Gather16Dev { const uint16_t* src; const int32_t* map; uint16_t* out; int64_t groups; };

template <int MODE>
__global__ void __launch_bounds__(256) gather16_kernel(Gather16Dev P) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
    typedef int32_t i32x2 __attribute__((ext_vector_type(2)));
    for (int64_t gI = (int64_t)blockIdx.x * 256 + threadIdx.x; gI < P.groups; gI += (int64_t)gridDim.x * 256) {
        int32_t ix[8];
        if (MODE == 0) {
            const i32x4 a = reinterpret_cast<const i32x4*>(P.map)[2 * gI], b = reinterpret_cast<const i32x4*>(P.map)[2 * gI + 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) { ix[e] = a[e]; ix[4 + e] = b[e]; }
        } else {
            const i32x2 bs = reinterpret_cast<const i32x2*>(P.map)[gI];
#pragma unroll
            for (int e = 0; e < 8; ++e) ix[e] = bs[0] + e * bs[1];
        }
        uint32_t v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = P.src[ix[e]];
        reinterpret_cast<u32x4*>(P.out)[gI] = u32x4{v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)};
    }
}
}  // namespace segm

extern "C" int segm_gather16(const segm_gather16_args* p) {
    if (!p) return SEGM_E_NULL;
    if (p->count == 0) return SEGM_OK;
    if (!p->src || !p->map || !p->out) return SEGM_E_NULL;
    if (p->count < 0 || p->count % 8 != 0 || (p->mode != 0 && p->mode != 1)) return SEGM_E_SHAPE;
    if (((uintptr_t)p->out & 15) || ((uintptr_t)p->map & 15)) return SEGM_E_SHAPE;
    segm::Gather16Dev P{(const uint16_t*)p->src, (const int32_t*)p->map, (uint16_t*)p->out, p->count / 8};
    int64_t blocks = (P.groups + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipStream_t st = (hipStream_t)p->stream;
    if (p->mode == 0) hipLaunchKernelGGL((segm::gather16_kernel<0>), dim3((unsigned)blocks), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((segm::gather16_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, st, P);
    return (int)hipGetLastError();
}
