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
//   workgroup = 3 waves: wave ky of workgroup (slab, co block, kz, ci block) owns the three kx taps of tap row
//               (kz, ky) for a 32 x 48 (co, ci) block: 2 (co tiles) x 3 (kx) x 3 (ci tiles) accumulators of 16x16
//               (72 VGPRs, fp32).  One step = one row (b, z, y) of dY times the row (z+kz-1, y+ky-1) of X over 32 x
//               positions: 2 + 3 aligned 16-byte loads and 6 halo dwords (all unconditional: clamped addresses,
//               masked values, issued together) feed 18 MFMAs.  Small workgroups and <= 128 VGPRs keep 4 waves per
//               SIMD resident, which is what hides the latency of the channel-strided NCDHW rows; the three ky
//               waves of a workgroup read the same dY rows and X rows one / two steps apart (L1 hits).
//               An odd last co tile (cout = 48) is a second launch with one co tile per workgroup.
//   grid      = (row slabs, co blocks of 32, ci blocks of 48); each workgroup walks its slab of (b, z, y) rows and
//               writes its partial 27 x 32 x 48 block; a second kernel sums the slabs in a fixed order
//               (deterministic, no atomics) and converts to the weight dtype.
// v_mfma_f32_16x16x32_bf16 operand layout (cdna_hip_programming.md §3): lane l holds A[i = l & 15][k = 8 (l >> 4) .. +7],
// B[k = 8 (l >> 4) .. +7][j = l & 15]; result D[row = 4 (l >> 4) + r][col = l & 15], r = 0..3.
#include <string.h>

#include "segm_device.h"

namespace segm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWgBlock = 48;                 // input channels per block; cin and cout must be multiples of it
constexpr int kWgCo = 32;                    // output channels per workgroup (2 MFMA tiles)
constexpr int kWgWaves = 3;                  // ky

struct WgradDev {
    const char* x;   int64_t x_sb, x_sc, x_sz, x_sy;      // element strides, x contiguous
    const char* dy;  int64_t dy_sb, dy_sc, dy_sz, dy_sy;
    float* part;                                          // [co blk][ci blk][slab][27][32][48]
    int32_t B, D, H, W, cob0;
    int32_t rows_per_slab, nslab;
    int32_t ncob, ncib;
};

// operands of one step, as loaded
struct WgStep {
    u32x4 a[2];                // dY fragments, one per co tile
    u32x4 v[3];                // X fragments (aligned), one per ci tile
    uint32_t hl[3], hr[3];     // halo dwords: elements (x-2, x-1) and (x+8, x+9), unmasked as loaded
    bool has_l, has_r;         // whether those exist (zero padding in x otherwise): applied when the step is used,
                               // so that nothing waits on the loads at issue time
    bool ok;                   // the step contributes (wave-uniform): its X row is not in the z / y zero padding
};

// position of a step inside the volume (all wave-uniform)
struct WgPos {
    int b, z, y, q;
    __device__ __forceinline__ void advance(const WgradDev& P, int nq) {
        if (++q < nq) return;
        q = 0;
        if (++y < P.H) return;
        y = 0;
        if (++z < P.D) return;
        z = 0; ++b;
    }
};

__device__ __forceinline__ void wg_load(WgStep& f, const WgradDev& P, const WgPos& p, int kz, int ky, int g,
                                        const __bf16* dyp, const __bf16* xp, int64_t dy_t1) {
    const int zz = p.z + kz - 1, yy = p.y + ky - 1;
    const bool ok = zz >= 0 && zz < P.D && yy >= 0 && yy < P.H;       // zero padding in z / y: the step contributes nothing
    const int zc = ok ? zz : p.z, yc = ok ? yy : p.y;
    const __bf16* dyr = dyp + (int64_t)p.b * P.dy_sb + (int64_t)p.z * P.dy_sz + (int64_t)p.y * P.dy_sy + 32 * p.q;
    const __bf16* xr = xp + (int64_t)p.b * P.x_sb + (int64_t)zc * P.x_sz + (int64_t)yc * P.x_sy + 32 * p.q;
    const int xoff = 32 * p.q + 8 * g;                                // first x of this lane's 8 reduction elements
    f.has_l = xoff > 0;
    f.has_r = xoff + 8 < P.W;
    f.ok = ok;
    f.a[0] = *reinterpret_cast<const u32x4*>(dyr);
    f.a[1] = *reinterpret_cast<const u32x4*>(dyr + dy_t1);           // dy_t1 = 0 when the block has a single co tile
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const __bf16* xc = xr + (int64_t)(16 * t) * P.x_sc;
        f.v[t] = *reinterpret_cast<const u32x4*>(xc);
        f.hl[t] = *reinterpret_cast<const uint32_t*>(xc + (f.has_l ? -2 : 0));
        f.hr[t] = *reinterpret_cast<const uint32_t*>(xc + (f.has_r ? 8 : 6));
    }
}

// the 18 (NCO = 2) or 9 MFMAs of one step.  Straight-line code: every loaded register is consumed on every path
// (a step in the z / y padding multiplies by a zeroed dY fragment instead of being skipped), otherwise the compiler
// has to drain all outstanding loads before the next look-ahead load may overwrite the buffer.
template <int NCO>
__device__ __forceinline__ void wg_compute(f32x4 (&acc)[2][3][3], const WgStep& f) {
    const u32x4 zero = {0u, 0u, 0u, 0u};
    bf16x8 a[NCO];
#pragma unroll
    for (int co = 0; co < NCO; ++co) a[co] = __builtin_bit_cast(bf16x8, f.ok ? f.a[co] : zero);
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
        const u32x4 v = f.v[ci];
        const uint32_t hl = f.has_l ? f.hl[ci] : 0u, hr = f.has_r ? f.hr[ci] : 0u;
        const uint32_t s1 = __builtin_amdgcn_alignbyte(v[1], v[0], 2);
        const uint32_t s2 = __builtin_amdgcn_alignbyte(v[2], v[1], 2);
        const uint32_t s3 = __builtin_amdgcn_alignbyte(v[3], v[2], 2);
        const u32x4 vl = {__builtin_amdgcn_alignbyte(v[0], hl, 2), s1, s2, s3};   // X[x-1 ..]  (kx = 0)
        const u32x4 vr = {s1, s2, s3, __builtin_amdgcn_alignbyte(hr, v[3], 2)};   // X[x+1 ..]  (kx = 2)
#pragma unroll
        for (int co = 0; co < NCO; ++co) {
            acc[co][0][ci] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[co], __builtin_bit_cast(bf16x8, vl), acc[co][0][ci], 0, 0, 0);
            acc[co][1][ci] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[co], __builtin_bit_cast(bf16x8, v), acc[co][1][ci], 0, 0, 0);
            acc[co][2][ci] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[co], __builtin_bit_cast(bf16x8, vr), acc[co][2][ci], 0, 0, 0);
        }
    }
}

template <int NCO>
__global__ void __launch_bounds__(kWgWaves * 64, 4) conv3d_k3_wgrad_kernel(WgradDev P) {
    const int lane = threadIdx.x & 63;
    const int ky = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // scalar: row / tap arithmetic stays on the SALU
    const int kz = blockIdx.y % 3;
    const int i16 = lane & 15, g = lane >> 4;
    const int slab = blockIdx.x, cob = blockIdx.y / 3 + P.cob0, cib = blockIdx.z;
    const int nrows = P.B * P.D * P.H;
    const int r0 = slab * P.rows_per_slab;
    const int r1 = (r0 + P.rows_per_slab < nrows) ? r0 + P.rows_per_slab : nrows;
    const int nq = P.W / 32;                              // 32-wide reduction chunks per row
    const int nsteps = (r1 > r0 ? r1 - r0 : 0) * nq;

    f32x4 acc[2][3][3];                                   // [co tile][kx][ci tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[a][b][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const __bf16* dyp = reinterpret_cast<const __bf16*>(P.dy) + (int64_t)(cob * kWgCo + i16) * P.dy_sc + 8 * g;
    const int64_t dy_t1 = NCO == 2 ? 16 * P.dy_sc : 0;
    const __bf16* xp = reinterpret_cast<const __bf16*>(P.x) + (int64_t)(cib * kWgBlock + i16) * P.x_sc + 8 * g;

    WgPos pos;
    pos.q = 0;
    pos.y = r0 % P.H;
    pos.z = (r0 / P.H) % P.D;
    pos.b = r0 / (P.H * P.D);
    for (int s = 0; s < nsteps; ++s) {
        WgStep f;
        wg_load(f, P, pos, kz, ky, g, dyp, xp, dy_t1);
        SEGM_SCHED_FENCE();        // all 11 loads are issued before the first MFMA (the scheduler otherwise trickles them)
        wg_compute<NCO>(acc, f);
        pos.advance(P, nq);
    }
    // partial block: part[((cob * ncib + cib) * nslab + slab)][tap = kz*9 + ky*3 + kx][co (32)][ci (48)]
    float* out = P.part + ((((int64_t)cob * P.ncib + cib) * P.nslab + slab) * 27) * (kWgCo * kWgBlock);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = ct * 16 + g * 4 + r, cin = ci * 16 + i16;
                    if (ct < NCO) out[((int64_t)(kz * 9 + ky * 3 + kx) * kWgCo + co) * kWgBlock + cin] = acc[ct][kx][ci][r];
                }
}

// dW[co][ci][tap] (contiguous (Cout, Cin, 3, 3, 3)) = sum over slabs of the partial blocks, fixed order:
// wave w of a workgroup adds slabs w, w + 4, ...; the four partial sums are then added in wave order.
template <typename T>
__global__ void __launch_bounds__(256) conv3d_k3_wgrad_reduce_kernel(const float* __restrict__ part, T* __restrict__ dw,
                                                                      int nslab, int ncib, int cout, int cin) {
    __shared__ float s_sum[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + lane;               // over 27 * cout * cin, ci fastest
    const int total = cout * cin * 27;
    const bool in = idx < total;
    const int ii = in ? idx : 0;
    const int ci = ii % cin;
    const int rest = ii / cin;
    const int co = rest % cout;
    const int tap = rest / cout;
    const int cob = co / kWgCo, cib = ci / kWgBlock;
    const float* p = part + ((((int64_t)cob * ncib + cib) * nslab) * 27 + tap) * (kWgCo * kWgBlock) +
                     (co - cob * kWgCo) * kWgBlock + (ci - cib * kWgBlock);
    float s = 0.f;
    for (int k = w; k < nslab; k += 4) s += p[(int64_t)k * 27 * kWgCo * kWgBlock];
    s_sum[w][lane] = s;
    __syncthreads();
    if (w == 0 && in) {
        const float t = (s_sum[0][lane] + s_sum[1][lane]) + (s_sum[2][lane] + s_sum[3][lane]);
        dw[((int64_t)co * cin + ci) * 27 + tap] = from_f32<T>(t);
    }
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
    return (size_t)((cout + kWgCo - 1) / kWgCo) * (cin / kWgBlock) * ns * 27 * kWgCo * kWgBlock * sizeof(float);
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
    P.ncob = (a->cout + kWgCo - 1) / kWgCo; P.ncib = a->cin / kWgBlock;
    hipStream_t stream = (hipStream_t)a->stream;
    const int full = a->cout / kWgCo;                    // blocks with two co tiles; cout % 32 == 16 leaves one with a single tile
    P.cob0 = 0;
    if (full > 0)
        hipLaunchKernelGGL((conv3d_k3_wgrad_kernel<2>), dim3(P.nslab, full * 3, P.ncib), dim3(kWgWaves * 64), 0, stream, P);
    if (P.ncob > full) {
        P.cob0 = full;
        hipLaunchKernelGGL((conv3d_k3_wgrad_kernel<1>), dim3(P.nslab, 3, P.ncib), dim3(kWgWaves * 64), 0, stream, P);
    }
    const int total = a->cout * a->cin * 27;
    if (a->dw_dtype == SEGM_F32)
        hipLaunchKernelGGL((conv3d_k3_wgrad_reduce_kernel<float>), dim3((total + 63) / 64), dim3(256), 0, stream,
                           (const float*)P.part, (float*)a->dw, P.nslab, P.ncib, a->cout, a->cin);
    else
        hipLaunchKernelGGL((conv3d_k3_wgrad_reduce_kernel<bf16_t>), dim3((total + 63) / 64), dim3(256), 0, stream,
                           (const float*)P.part, (bf16_t*)a->dw, P.nslab, P.ncib, a->cout, a->cin);
    return (int)hipGetLastError();
}
