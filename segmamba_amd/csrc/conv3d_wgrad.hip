// Weight gradient of a 3x3x3, stride-1, pad-1 convolution (C ABI: segm_conv3d_k3_wgrad) - the stem's hottest op.
//
// SegMamba's encoder/decoder (reference monai/networks/blocks/dynunet_block.py:44-111 via torch.nn.Conv3d ->
// cuDNN) spends most of its backward pass in the weight gradients of 48-channel 3x3x3 convolutions on 128^3 volumes.
// MIOpen's solver for them is im2col + GEMM: 22 ms per call, 31 % of a training step
// (profiles/r01_bench_step_kernels_v4.txt).  This kernel is a direct MFMA formulation with no im2col buffer:
//
//   dW[co, ci, kz, ky, kx] = sum_{b, z, y, x} dY[b, co, z, y, x] * X[b, ci, z+kz-1, y+ky-1, x+kx-1]
//
// For a fixed row (b, z, y) and a fixed (kz, ky) this is three 48x48x128 GEMMs (kx = 0, 1, 2) whose reduction index is
// x - contiguous in memory for both operands in NCDHW, which is exactly what an MFMA fragment wants (8 consecutive k
// per lane = one 16-byte load).  The three x-shifts are built from ONE aligned 16-byte load plus two halo dwords with
// v_alignbyte, so every global access stays aligned.
//
//   workgroup = 9 waves: wave (kz, ct) owns output rows co in [16 ct, 16 ct + 16) of tap plane kz for a 48 x 48
//               (co, ci) block: 3 (ky) x 3 (kx) x 3 (ci tiles) accumulators of 16x16 (108 VGPRs, fp32)
//   grid      = (row slabs, co blocks of 48, ci blocks of 48); each workgroup walks its slab of (b, z, y) rows and
//               writes its partial 27 x 48 x 48 block; a second kernel sums the slabs in a fixed order
//               (deterministic, no atomics) and converts to the weight dtype.
//
// v_mfma_f32_16x16x32_bf16 operand layout (cdna_hip_programming.md §3): lane l holds A[i = l & 15][k = 8 (l >> 4) .. +7],
// B[k = 8 (l >> 4) .. +7][j = l & 15]; result D[row = 4 (l >> 4) + r][col = l & 15], r = 0..3.
#include <string.h>

#include "segm_device.h"

namespace segm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWgBlock = 48;                 // channels per (co, ci) block
constexpr int kWgWaves = 9;                  // 3 tap planes x 3 co tiles

struct WgradDev {
    const char* x;   int64_t x_sb, x_sc, x_sz, x_sy;      // element strides, x contiguous
    const char* dy;  int64_t dy_sb, dy_sc, dy_sz, dy_sy;
    float* part;                                          // [co blk][ci blk][slab][27][48][48]
    int32_t B, D, H, W;
    int32_t rows_per_slab, nslab;
    int32_t ncob, ncib;
};

__global__ void __launch_bounds__(kWgWaves * 64) conv3d_k3_wgrad_kernel(WgradDev P) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kz = wave / 3, ct = wave - kz * 3;
    const int i16 = lane & 15, g = lane >> 4;
    const int slab = blockIdx.x, cob = blockIdx.y, cib = blockIdx.z;
    const int nrows = P.B * P.D * P.H;
    const int r0 = slab * P.rows_per_slab;
    const int r1 = (r0 + P.rows_per_slab < nrows) ? r0 + P.rows_per_slab : nrows;
    const int nq = P.W / 32;                              // 32-wide reduction chunks per row

    f32x4 acc[3][3][3];                                   // [ky][kx][ci tile]
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[a][b][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const __bf16* dyp = reinterpret_cast<const __bf16*>(P.dy) + (int64_t)(cob * kWgBlock + ct * 16 + i16) * P.dy_sc + 8 * g;
    const __bf16* xp = reinterpret_cast<const __bf16*>(P.x) + (int64_t)(cib * kWgBlock + i16) * P.x_sc + 8 * g;

    for (int row = r0; row < r1; ++row) {
        const int y = row % P.H;
        const int bz = row / P.H;
        const int z = bz % P.D, b = bz / P.D;
        const int zz = z + kz - 1;
        if (zz < 0 || zz >= P.D) continue;                // zero padding in z (uniform over the wave)
        const __bf16* dyr = dyp + (int64_t)b * P.dy_sb + (int64_t)z * P.dy_sz + (int64_t)y * P.dy_sy;
        for (int q = 0; q < nq; ++q) {
            const bf16x8 afrag = *reinterpret_cast<const bf16x8*>(dyr + 32 * q);
            const int xoff = 32 * q + 8 * g;              // first x of this lane's 8 reduction elements
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = y + ky - 1;
                if (yy < 0 || yy >= P.H) continue;        // zero padding in y (uniform)
                const __bf16* xr = xp + (int64_t)b * P.x_sb + (int64_t)zz * P.x_sz + (int64_t)yy * P.x_sy + 32 * q;
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const __bf16* xc = xr + (int64_t)(16 * ci) * P.x_sc;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(xc);
                    // halo dwords: elements (x-2, x-1) and (x+8, x+9); zero outside the row (padding in x)
                    const uint32_t hl = (xoff > 0) ? *reinterpret_cast<const uint32_t*>(xc - 2) : 0u;
                    const uint32_t hr = (xoff + 8 < P.W) ? *reinterpret_cast<const uint32_t*>(xc + 8) : 0u;
                    const uint32_t s1 = __builtin_amdgcn_alignbyte(v[1], v[0], 2);
                    const uint32_t s2 = __builtin_amdgcn_alignbyte(v[2], v[1], 2);
                    const uint32_t s3 = __builtin_amdgcn_alignbyte(v[3], v[2], 2);
                    const u32x4 vl = {__builtin_amdgcn_alignbyte(v[0], hl, 2), s1, s2, s3};   // X[x-1 ..]  (kx = 0)
                    const u32x4 vr = {s1, s2, s3, __builtin_amdgcn_alignbyte(hr, v[3], 2)};   // X[x+1 ..]  (kx = 2)
                    acc[ky][0][ci] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, __builtin_bit_cast(bf16x8, vl), acc[ky][0][ci], 0, 0, 0);
                    acc[ky][1][ci] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, __builtin_bit_cast(bf16x8, v), acc[ky][1][ci], 0, 0, 0);
                    acc[ky][2][ci] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, __builtin_bit_cast(bf16x8, vr), acc[ky][2][ci], 0, 0, 0);
                }
            }
        }
    }
    // partial block: part[((cob * ncib + cib) * nslab + slab)][tap = kz*9 + ky*3 + kx][co][ci]
    float* out = P.part + ((((int64_t)cob * P.ncib + cib) * P.nslab + slab) * 27) * (kWgBlock * kWgBlock);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = ct * 16 + g * 4 + r, cin = ci * 16 + i16;
                    out[((int64_t)(kz * 9 + ky * 3 + kx) * kWgBlock + co) * kWgBlock + cin] = acc[ky][kx][ci][r];
                }
}

// dW[co][ci][tap] (contiguous (Cout, Cin, 3, 3, 3)) = sum over slabs of the partial blocks, fixed order.
template <typename T>
__global__ void __launch_bounds__(256) conv3d_k3_wgrad_reduce_kernel(const float* __restrict__ part, T* __restrict__ dw,
                                                                      int nslab, int ncib, int cout, int cin) {
    const int idx = blockIdx.x * 256 + threadIdx.x;       // over cout * cin * 27, ci fastest within a block row
    const int total = cout * cin * 27;
    if (idx >= total) return;
    const int ci = idx % cin;
    const int rest = idx / cin;
    const int co = rest % cout;
    const int tap = rest / cout;
    const int cob = co / kWgBlock, cib = ci / kWgBlock;
    const float* p = part + ((((int64_t)cob * ncib + cib) * nslab) * 27 + tap) * (kWgBlock * kWgBlock) +
                     (co - cob * kWgBlock) * kWgBlock + (ci - cib * kWgBlock);
    float s = 0.f;
    for (int k = 0; k < nslab; ++k) s += p[(int64_t)k * 27 * kWgBlock * kWgBlock];
    dw[((int64_t)co * cin + ci) * 27 + tap] = from_f32<T>(s);
}

static int wgrad_slabs(int nrows) {
    int ns = 512;                                         // ~2 workgroups per CU
    if (ns > nrows) ns = nrows;
    return ns;
}

}  // namespace segm

using namespace segm;

extern "C" size_t segm_conv3d_k3_wgrad_workspace_bytes(int32_t batch, int32_t cin, int32_t cout, int32_t d, int32_t h, int32_t w) {
    if (batch <= 0 || cin <= 0 || cout <= 0 || d <= 0 || h <= 0 || w <= 0) return 0;
    const int ns = wgrad_slabs(batch * d * h);
    return (size_t)(cout / kWgBlock) * (cin / kWgBlock) * ns * 27 * kWgBlock * kWgBlock * sizeof(float);
}

extern "C" int segm_conv3d_k3_wgrad(const segm_conv3d_wgrad_args* a) {
    if (!a) return SEGM_E_NULL;
    if (!a->x || !a->dy || !a->dw || !a->workspace) return SEGM_E_NULL;
    if (a->batch <= 0 || a->depth <= 0 || a->height <= 0 || a->width <= 0) return SEGM_E_SHAPE;
    if (a->cin % kWgBlock != 0 || a->cout % kWgBlock != 0 || a->cin <= 0 || a->cout <= 0) return SEGM_E_SHAPE;
    if (a->width % 32 != 0) return SEGM_E_SHAPE;
    if (a->dtype != SEGM_BF16) return SEGM_E_DTYPE;
    if (a->dw_dtype != SEGM_BF16 && a->dw_dtype != SEGM_F32) return SEGM_E_DTYPE;
    // 16-byte aligned rows: every stride a multiple of 8 elements, base pointers 16-byte aligned
    const int64_t st[8] = {a->x_stride_b, a->x_stride_c, a->x_stride_z, a->x_stride_y,
                           a->dy_stride_b, a->dy_stride_c, a->dy_stride_z, a->dy_stride_y};
    for (int64_t s : st)
        if (s % 8 != 0) return SEGM_E_SHAPE;
    if (((uintptr_t)a->x & 15) || ((uintptr_t)a->dy & 15)) return SEGM_E_SHAPE;
    const size_t need = segm_conv3d_k3_wgrad_workspace_bytes(a->batch, a->cin, a->cout, a->depth, a->height, a->width);
    if (a->workspace_bytes < need) return SEGM_E_WORKSPACE;

    WgradDev P;
    memset(&P, 0, sizeof(P));
    P.x = (const char*)a->x; P.x_sb = a->x_stride_b; P.x_sc = a->x_stride_c; P.x_sz = a->x_stride_z; P.x_sy = a->x_stride_y;
    P.dy = (const char*)a->dy; P.dy_sb = a->dy_stride_b; P.dy_sc = a->dy_stride_c; P.dy_sz = a->dy_stride_z; P.dy_sy = a->dy_stride_y;
    P.part = (float*)a->workspace;
    P.B = a->batch; P.D = a->depth; P.H = a->height; P.W = a->width;
    const int nrows = a->batch * a->depth * a->height;
    P.nslab = wgrad_slabs(nrows);
    P.rows_per_slab = (nrows + P.nslab - 1) / P.nslab;
    P.ncob = a->cout / kWgBlock; P.ncib = a->cin / kWgBlock;
    hipStream_t stream = (hipStream_t)a->stream;
    hipLaunchKernelGGL(conv3d_k3_wgrad_kernel, dim3(P.nslab, P.ncob, P.ncib), dim3(kWgWaves * 64), 0, stream, P);
    const int total = a->cout * a->cin * 27;
    if (a->dw_dtype == SEGM_F32)
        hipLaunchKernelGGL((conv3d_k3_wgrad_reduce_kernel<float>), dim3((total + 255) / 256), dim3(256), 0, stream,
                           (const float*)P.part, (float*)a->dw, P.nslab, P.ncib, a->cout, a->cin);
    else
        hipLaunchKernelGGL((conv3d_k3_wgrad_reduce_kernel<bf16_t>), dim3((total + 255) / 256), dim3(256), 0, stream,
                           (const float*)P.part, (bf16_t*)a->dw, P.nslab, P.ncib, a->cout, a->cin);
    return (int)hipGetLastError();
}
