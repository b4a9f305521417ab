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
// What bounds it is operand traffic, not arithmetic: with only 48 channels every X element feeds 27 x 32 MACs, so
// fetching operands per tap from L2 (a first version: 15.6 GB of L1 fills for 0.8 GB of tensors, 4 ms) leaves the MFMA
// pipe idle.  Operands are therefore staged once per workgroup in LDS:
//
//   workgroup = 3 waves (ky = wave) for one tap plane kz, one 48 x 48 (co, ci) block and one work item
//               (batch b, depth z, 64-wide x block, y range).  It walks y; an LDS ring of 4 rows holds
//               X[ci block][z+kz-1][y-1 .. y+2][x block + 8 halo columns each side] (each X row is fetched ONCE and
//               used by the three ky waves on three consecutive steps), a double buffer holds the dY row.  While the
//               waves run the 54 MFMAs of step y (2 k-chunks x 3 co tiles x 3 kx x 3 ci tiles), the row y+2 and the
//               next dY row are already in flight global -> registers; they are parked in LDS after the MFMAs; one
//               barrier per step.
//   fragments   A (dY) and B (X) fragments are 16-byte LDS reads; the kx = 0 / 2 operands are built from the aligned
//               read plus two halo dwords with v_alignbyte.  Zero padding in x / y / z is materialised as zero rows /
//               columns in LDS, so the inner loop has no masks.  Row pitch 88 elements (44 dwords): the 16 lanes of a
//               fragment read hit 16 distinct 4-bank groups.
//   grid      = (co blocks of 48 x work items x 3 kz in an XCD-aware order, ci blocks of 48); every workgroup writes its 9 x 48 x 48 partial
//               taps; a second kernel sums the work items in a fixed order (deterministic, no atomics) and converts
//               to the weight dtype.
// v_mfma_f32_16x16x32_bf16 operand layout (cdna_hip_programming.md §3): lane l holds A[i = l & 15][k = 8 (l >> 4) .. +7],
// B[k = 8 (l >> 4) .. +7][j = l & 15]; result D[row = 4 (l >> 4) + r][col = l & 15], r = 0..3.
#include <string.h>

#include "segm_device.h"

namespace segm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWgBlock = 48;                 // input channels per block; cin and cout must be multiples of it
constexpr int kWgCo = 48;                    // output channels per workgroup (3 MFMA tiles)
constexpr int kWgWaves = 3;                  // ky
constexpr int kWgThreads = kWgWaves * 64;
constexpr int kPitch = 88;                   // LDS row pitch (elements): 8 halo + 64 + 8 halo + 8 (bank spread)
constexpr int kCopies = 5;                   // granules per thread per step: ceil((48 * 10 + 48 * 8) / 192)
constexpr int wg_copies(int qs) { return qs == 1 ? kCopies : 3; }     // six waves: ceil(864 / 384)

struct WgradDev {
    const char* x;   int64_t x_sb, x_sc, x_sz, x_sy;      // element strides, x contiguous
    const char* dy;  int64_t dy_sb, dy_sc, dy_sz, dy_sy;
    float* part;                                          // [co blk][ci blk][item][27][32][48]
    int32_t B, D, H, W, cin;
    int32_t nxb, ysplit, rows_per_part, nitems;
    int32_t ncob, ncib;
};

// One thread's share of the per-step global -> LDS copy: up to NC 16-byte granules.
template <int NC> struct WgCopy {
    int64_t src[NC];           // element offset from the row base (X granules: from the X row, dY granules: from the dY row)
    int32_t dst[NC];           // element offset inside an X ring slot / a dY buffer
    bool is_x[NC], live[NC], inside[NC];                  // inside: the granule's x range is inside the volume
};

// QS = 1: three waves (ky), each walks both 32-wide halves (q) of a 64-wide x block.  QS = 2 (64-wide blocks only): SIX waves =
// ky x q, each with the 27 accumulator tiles of its ky over its own half of x, summed through LDS after the row loop.  Why: two
// three-wave workgroups put 2, 2, 1, 1 waves on a CU's four SIMDs and the row loop runs at the speed of the SIMDs that hold two
// (a SIMD's matrix pipe and issue slots are shared by its waves, MI355X_MICROARCH.md "two waves per SIMD"); two six-wave
// workgroups of half the work per wave are 3, 3, 3, 3.  Needs <= 168 registers (three waves per SIMD): the half-width wave holds
// the same 108 accumulator registers and fewer fragments.
template <typename T, int NQ, int QS>
__global__ void __launch_bounds__(kWgThreads * QS, QS == 2 ? 3 : 2) conv3d_k3_wgrad_kernel(WgradDev P) {
    typedef typename Mfma16<T>::v8 frag8;
    static_assert(QS == 1 || NQ == 2, "the six-wave layout splits a 64-wide block");
    constexpr int NCO = kWgCo / 16;
    constexpr int NC = wg_copies(QS), NTHR = kWgThreads * QS;
    __shared__ __attribute__((aligned(16))) T xs[4][kWgBlock][kPitch];
    __shared__ __attribute__((aligned(16))) T dys[2][kWgCo][kPitch];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // scalar: row / tap arithmetic stays on the SALU
    const int ky = wave % 3, qw = wave / 3;                           // QS = 2: this wave's half of the x block
    const int i16 = lane & 15, g = lane >> 4;
    // blockIdx.x runs over (co block, item, kz) with kz fastest, re-ordered so that every XCD owns a contiguous range: the three
    // tap planes of an item read the same dY rows, items that are neighbours in z the same X rows - one L2 fetches them once
    // (before: the kz planes were grid rows, dispatched thousands of workgroups apart: X and dY came from HBM three times)
    const int vid = xcd_item(blockIdx.x, gridDim.x);
    const int kz = vid % 3, cib = blockIdx.y;
    const int item_id = (vid / 3) % P.nitems, cob = vid / (3 * P.nitems);
    int item = item_id;
    const int ypart = item % P.ysplit;  item /= P.ysplit;
    const int xb = item % P.nxb;        item /= P.nxb;
    const int z = item % P.D, b = item / P.D;
    const int zz = z + kz - 1;
    const bool plane_ok = zz >= 0 && zz < P.D;                       // else: the whole tap plane reads z padding -> zeros
    const int y0 = ypart * P.rows_per_part;
    const int y1 = (y0 + P.rows_per_part < P.H) ? y0 + P.rows_per_part : P.H;
    constexpr int XB = 32 * NQ;
    const int x0 = xb * XB;

    f32x4 acc[NCO][3][3];                                 // [co tile][kx][ci tile]
#pragma unroll
    for (int a = 0; a < NCO; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d) acc[a][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (plane_ok && y1 > y0) {
        // ---- copy plan -------------------------------------------------------------------------------------------
        constexpr int XG = 4 * NQ + 2;                    // granules per X row: left halo, data, right halo
        constexpr int DG = 4 * NQ;                        // granules per dY row
        constexpr int NX = kWgBlock * XG, ND = NCO * 16 * DG;
        WgCopy<NC> cp;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int id = tid + k * NTHR;
            cp.is_x[k] = id < NX;
            cp.live[k] = id < NX + ND;
            if (cp.is_x[k]) {
                const int ci = id / XG, gr = id - ci * XG;
                const int xg = x0 - 8 + 8 * gr;           // first x of the granule (W % 8 == 0: all inside or all outside)
                const int cg = cib * kWgBlock + ci;       // channels at or beyond cin (a narrow first layer) are zero rows
                cp.inside[k] = xg >= 0 && xg < P.W && cg < P.cin;
                cp.src[k] = (int64_t)(cg < P.cin ? cg : 0) * P.x_sc + (xg >= 0 && xg < P.W ? xg : 0);
                cp.dst[k] = ci * kPitch + 8 * gr;
            } else {
                const int j = cp.live[k] ? id - NX : 0;
                const int co = j / DG, gr = j - co * DG;
                cp.inside[k] = x0 + 8 * gr < P.W;         // widths below 32: the rest of the k-chunk is zero
                cp.src[k] = (int64_t)(cob * kWgCo + co) * P.dy_sc + (cp.inside[k] ? x0 + 8 * gr : 0);
                cp.dst[k] = co * kPitch + 8 * gr;
            }
        }
        const T* xplane = reinterpret_cast<const T*>(P.x) + (int64_t)b * P.x_sb + (int64_t)zz * P.x_sz;
        const T* dyplane = reinterpret_cast<const T*>(P.dy) + (int64_t)b * P.dy_sb + (int64_t)z * P.dy_sz;
        const u32x4 zero4 = {0u, 0u, 0u, 0u};

        // fetch X row yy and dY row yd into registers (rows outside the volume / range read a valid row and are zeroed)
        auto fetch = [&](u32x4 (&r)[NC], int yy, int yd) {
            const bool x_ok = yy >= 0 && yy < P.H;
            const T* xr = xplane + (int64_t)(x_ok ? yy : 0) * P.x_sy;
            const T* dr = dyplane + (int64_t)(yd < P.H ? yd : 0) * P.dy_sy;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const T* src = (cp.is_x[k] ? xr : dr) + cp.src[k];
                r[k] = *reinterpret_cast<const u32x4*>(src);
            }
        };
        auto park = [&](const u32x4 (&r)[NC], int yy, int slot, int buf) {
            const bool x_ok = yy >= 0 && yy < P.H;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                if (!cp.live[k]) continue;
                const bool keep = cp.is_x[k] ? (x_ok && cp.inside[k]) : cp.inside[k];
                T* dst = (cp.is_x[k] ? &xs[slot][0][0] : &dys[buf][0][0]) + cp.dst[k];
                *reinterpret_cast<u32x4*>(dst) = keep ? r[k] : zero4;
            }
        };

        // ---- prologue: rows y0 - 1, y0, y0 + 1 and dY row y0 ---------------------------------------------------------
        {
            u32x4 r[NC];
#pragma unroll
            for (int d = -1; d <= 1; ++d) {
                fetch(r, y0 + d, y0);
                park(r, y0 + d, (y0 + d + 4) & 3, 0);     // the dY row is parked three times (same data): harmless
            }
        }
        __syncthreads();

        // ---- main loop ------------------------------------------------------------------------------------------------
        for (int y = y0; y < y1; ++y) {
            u32x4 r[NC];
            fetch(r, y + 2, y + 1);                       // in flight during this step's MFMAs
            SEGM_SCHED_FENCE();
            const int slot = (y + ky - 1 + 4) & 3, buf = (y - y0) & 1;
#pragma unroll
            for (int qi = 0; qi < NQ / QS; ++qi) {
                const int q = QS == 2 ? qw : qi;
                u32x4 av[NCO];
#pragma unroll
                for (int co = 0; co < NCO; ++co)
                    av[co] = *reinterpret_cast<const u32x4*>(&dys[buf][co * 16 + i16][32 * q + 8 * g]);
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const T* xc = &xs[slot][ci * 16 + i16][8 + 32 * q + 8 * g];
                    const u32x4 v = *reinterpret_cast<const u32x4*>(xc);
                    const uint32_t hl = *reinterpret_cast<const uint32_t*>(xc - 2);     // elements (x-2, x-1)
                    const uint32_t hr = *reinterpret_cast<const uint32_t*>(xc + 8);     // elements (x+8, x+9)
                    const uint32_t s1 = __builtin_amdgcn_alignbyte(v[1], v[0], 2);
                    const uint32_t s2 = __builtin_amdgcn_alignbyte(v[2], v[1], 2);
                    const uint32_t s3 = __builtin_amdgcn_alignbyte(v[3], v[2], 2);
                    const u32x4 vl = {__builtin_amdgcn_alignbyte(v[0], hl, 2), s1, s2, s3};   // X[x-1 ..]  (kx = 0)
                    const u32x4 vr = {s1, s2, s3, __builtin_amdgcn_alignbyte(hr, v[3], 2)};   // X[x+1 ..]  (kx = 2)
#pragma unroll
                    for (int co = 0; co < NCO; ++co) {
                        const frag8 a = __builtin_bit_cast(frag8, av[co]);
                        acc[co][0][ci] = Mfma16<T>::run(a, __builtin_bit_cast(frag8, vl), acc[co][0][ci]);
                        acc[co][1][ci] = Mfma16<T>::run(a, __builtin_bit_cast(frag8, v), acc[co][1][ci]);
                        acc[co][2][ci] = Mfma16<T>::run(a, __builtin_bit_cast(frag8, vr), acc[co][2][ci]);
                    }
                }
            }
            SEGM_SCHED_FENCE();
            park(r, y + 2, (y + 2) & 3, buf ^ 1);
            __syncthreads();
        }
    }
    if constexpr (QS == 2) {
        // the two halves of x: the q = 1 waves hand their tiles to the q = 0 waves through the (now idle) X ring, one co tile per
        // round (3 waves x 9 tiles x 1 KB = 27 KB of the ring's 33 KB)
        f32x4* lf = reinterpret_cast<f32x4*>(&xs[0][0][0]);
#pragma unroll
        for (int ct = 0; ct < NCO; ++ct) {
            __syncthreads();                              // the ring is free (first round: every wave is out of the row loop)
            if (qw == 1) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci) lf[((ky * 3 + kx) * 3 + ci) * 64 + lane] = acc[ct][kx][ci];
            }
            __syncthreads();
            if (qw == 0) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci) acc[ct][kx][ci] += lf[((ky * 3 + kx) * 3 + ci) * 64 + lane];
            }
        }
        if (qw == 1) return;
    }
    // partial block: part[((cob * ncib + cib) * nitems + item)][tap = kz*9 + ky*3 + kx][co (32)][ci (48)]
    float* out = P.part + ((((int64_t)cob * P.ncib + cib) * P.nitems + item_id) * 27) * (kWgCo * kWgBlock);
#pragma unroll
    for (int ct = 0; ct < NCO; ++ct)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = ct * 16 + g * 4 + r, cin = ci * 16 + i16;
                    out[((int64_t)(kz * 9 + ky * 3 + kx) * kWgCo + co) * kWgBlock + cin] = acc[ct][kx][ci][r];
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
    // eight independent partial sums: eight loads in flight per lane (a single running sum waits for one 250 KB-strided load
    // per addition: 78 us per launch for 127 MB of partials that stream in 25 us); fixed order, deterministic
    const int64_t slab = (int64_t)27 * kWgCo * kWgBlock;
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int k = w;
    for (; k + 28 < nslab; k += 32) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a8[u] += p[(int64_t)(k + 4 * u) * slab];
    }
    for (int u = 0; k < nslab; k += 4, ++u) a8[u & 7] += p[(int64_t)k * slab];
    const float s = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    s_sum[w][lane] = s;
    __syncthreads();
    if (w == 0 && in) {
        const float t = (s_sum[0][lane] + s_sum[1][lane]) + (s_sum[2][lane] + s_sum[3][lane]);
        dw[((int64_t)co * cin + ci) * 27 + tap] = from_f32<T>(t);
    }
}

// The same sum with 16-byte loads (cin a multiple of 4: every SegMamba layer): a lane owns four consecutive ci of one (tap, co),
// a wave reads 1 KB contiguous per slab instead of 256 B, eight waves take slabs w, w + 8, ... with eight loads in flight each -
// four times the bytes in flight per workgroup for the same fixed summation order per element class (wave-strided partial sums,
// then waves in order).  NOT bit-identical to the kernel above (eight strided classes instead of four): both are deterministic.
template <typename T>
__global__ void __launch_bounds__(512) conv3d_k3_wgrad_reduce4_kernel(const float* __restrict__ part, T* __restrict__ dw,
                                                                       int nslab, int ncib, int cout, int cin) {
    __shared__ f32x4 s_sum[8][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + lane;               // over 27 * cout * cin / 4, ci fastest
    const int total4 = cout * cin * 27 / 4;
    const bool in = idx < total4;
    const int ii = (in ? idx : 0) * 4;
    const int ci = ii % cin;
    const int rest = ii / cin;
    const int co = rest % cout;
    const int tap = rest / cout;
    const int cob = co / kWgCo, cib = ci / kWgBlock;
    const f32x4* p = reinterpret_cast<const f32x4*>(part + ((((int64_t)cob * ncib + cib) * nslab) * 27 + tap) * (kWgCo * kWgBlock) +
                                                   (co - cob * kWgCo) * kWgBlock + (ci - cib * kWgBlock));
    const int64_t slab4 = (int64_t)27 * kWgCo * kWgBlock / 4;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 a8[8] = {z4, z4, z4, z4, z4, z4, z4, z4};
    int k = w;
    for (; k + 56 < nslab; k += 64) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a8[u] += p[(int64_t)(k + 8 * u) * slab4];
    }
    for (int u = 0; k < nslab; k += 8, ++u) a8[u & 7] += p[(int64_t)k * slab4];
    s_sum[w][lane] = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    __syncthreads();
    if (w == 0 && in) {
        const f32x4 t = ((s_sum[0][lane] + s_sum[1][lane]) + (s_sum[2][lane] + s_sum[3][lane])) +
                        ((s_sum[4][lane] + s_sum[5][lane]) + (s_sum[6][lane] + s_sum[7][lane]));
#pragma unroll
        for (int q = 0; q < 4; ++q) dw[((int64_t)co * cin + ci + q) * 27 + tap] = from_f32<T>(t[q]);
    }
}

// work decomposition: items = (batch, depth) planes x 64-wide (32 if W % 64) x blocks x y parts
struct WgPlan { int nq, nxb, ysplit, rows_per_part, nitems; };
static WgPlan wgrad_plan(int batch, int cin, int cout, int d, int h, int w) {
    WgPlan p;
    p.nq = (w % 64 == 0) ? 2 : 1;
    p.nxb = (w + 32 * p.nq - 1) / (32 * p.nq);
    const int64_t wgs = (int64_t)batch * d * p.nxb * 3 * ((cout + kWgCo - 1) / kWgCo) * ((cin + kWgBlock - 1) / kWgBlock);
    int split = 1;                                        // cut y when there are too few workgroups to fill 256 CUs x 3
    while (wgs * split < 1536 && h / (split * 2) >= 8) split *= 2;
    p.ysplit = split;
    p.rows_per_part = (h + split - 1) / split;
    p.nitems = batch * d * p.nxb * split;
    return p;
}

}  // namespace segm

using namespace segm;

extern "C" size_t segm_conv3d_k3_wgrad_workspace_bytes(int32_t batch, int32_t cin, int32_t cout, int32_t d, int32_t h, int32_t w) {
    if (batch <= 0 || cin <= 0 || cout <= 0 || d <= 0 || h <= 0 || w <= 0) return 0;
    const WgPlan pl = wgrad_plan(batch, cin, cout, d, h, w);
    return (size_t)((cout + kWgCo - 1) / kWgCo) * ((cin + kWgBlock - 1) / kWgBlock) * pl.nitems * 27 * kWgCo * kWgBlock * sizeof(float);
}

extern "C" int segm_conv3d_k3_wgrad(const segm_conv3d_wgrad_args* a) {
    if (!a) return SEGM_E_NULL;
    if (!a->x || !a->dy || !a->dw || !a->workspace) return SEGM_E_NULL;
    if (a->batch <= 0 || a->depth <= 0 || a->height <= 0 || a->width <= 0) return SEGM_E_SHAPE;
    if (a->cin <= 0 || a->cout <= 0 || a->cout % kWgBlock != 0) return SEGM_E_SHAPE;
    if (a->cin % kWgBlock != 0 && a->cin > kWgBlock) return SEGM_E_SHAPE;      // a multiple of 48, or one narrow block
    if (a->width % 8 != 0) return SEGM_E_SHAPE;
    if (a->dtype != SEGM_BF16 && a->dtype != SEGM_F16) return SEGM_E_DTYPE;
    if (a->dw_dtype != SEGM_BF16 && a->dw_dtype != SEGM_F16 && a->dw_dtype != SEGM_F32) return SEGM_E_DTYPE;
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
    P.B = a->batch; P.D = a->depth; P.H = a->height; P.W = a->width; P.cin = a->cin;
    const WgPlan pl = wgrad_plan(a->batch, a->cin, a->cout, a->depth, a->height, a->width);
    P.nxb = pl.nxb; P.ysplit = pl.ysplit; P.rows_per_part = pl.rows_per_part; P.nitems = pl.nitems;
    P.ncob = (a->cout + kWgCo - 1) / kWgCo; P.ncib = (a->cin + kWgBlock - 1) / kWgBlock;
    hipStream_t stream = (hipStream_t)a->stream;
    {
        const dim3 grid(P.nitems * 3 * P.ncob, P.ncib);
        static const bool six = [] { const char* e = getenv("SEGM_WGRAD_SIX"); return e && atoi(e) == 1; }();     // A/B: six-wave layout
        if (a->dtype == SEGM_F16) {
            if (pl.nq == 2 && six) hipLaunchKernelGGL((conv3d_k3_wgrad_kernel<f16_t, 2, 2>), grid, dim3(kWgThreads * 2), 0, stream, P);
            else if (pl.nq == 2) hipLaunchKernelGGL((conv3d_k3_wgrad_kernel<f16_t, 2, 1>), grid, dim3(kWgThreads), 0, stream, P);
            else hipLaunchKernelGGL((conv3d_k3_wgrad_kernel<f16_t, 1, 1>), grid, dim3(kWgThreads), 0, stream, P);
        } else {
            if (pl.nq == 2 && six) hipLaunchKernelGGL((conv3d_k3_wgrad_kernel<bf16_t, 2, 2>), grid, dim3(kWgThreads * 2), 0, stream, P);
            else if (pl.nq == 2) hipLaunchKernelGGL((conv3d_k3_wgrad_kernel<bf16_t, 2, 1>), grid, dim3(kWgThreads), 0, stream, P);
            else hipLaunchKernelGGL((conv3d_k3_wgrad_kernel<bf16_t, 1, 1>), grid, dim3(kWgThreads), 0, stream, P);
        }
    }
    const int total = a->cout * a->cin * 27;
    // 16-byte loads where they measured faster (profiles/r04_call9_ab.log: 96 -> 96 @64^3 0.368 -> 0.336 ms; 48 -> 48 @128^3 0.69 ->
    // 0.72 ms - one (co, ci) block pair has too few 1 KB pieces per slab to fill the chip); SEGM_WGRAD_REDUCE4 = 0 / 1 forces a side
    static const int wide = [] { const char* e = getenv("SEGM_WGRAD_REDUCE4"); return e ? atoi(e) : -1; }();
    if (a->cin % 4 == 0 && (wide == 1 || (wide < 0 && (int64_t)a->cin * a->cout >= 96 * 96))) {
        const dim3 g4((total / 4 + 63) / 64);
        if (a->dw_dtype == SEGM_F32)
            hipLaunchKernelGGL((conv3d_k3_wgrad_reduce4_kernel<float>), g4, dim3(512), 0, stream, (const float*)P.part, (float*)a->dw, P.nitems, P.ncib, a->cout, a->cin);
        else if (a->dw_dtype == SEGM_F16)
            hipLaunchKernelGGL((conv3d_k3_wgrad_reduce4_kernel<f16_t>), g4, dim3(512), 0, stream, (const float*)P.part, (f16_t*)a->dw, P.nitems, P.ncib, a->cout, a->cin);
        else
            hipLaunchKernelGGL((conv3d_k3_wgrad_reduce4_kernel<bf16_t>), g4, dim3(512), 0, stream, (const float*)P.part, (bf16_t*)a->dw, P.nitems, P.ncib, a->cout, a->cin);
        return (int)hipGetLastError();
    }
    if (a->dw_dtype == SEGM_F32)
        hipLaunchKernelGGL((conv3d_k3_wgrad_reduce_kernel<float>), dim3((total + 63) / 64), dim3(256), 0, stream,
                           (const float*)P.part, (float*)a->dw, P.nitems, P.ncib, a->cout, a->cin);
    else if (a->dw_dtype == SEGM_F16)
        hipLaunchKernelGGL((conv3d_k3_wgrad_reduce_kernel<f16_t>), dim3((total + 63) / 64), dim3(256), 0, stream,
                           (const float*)P.part, (f16_t*)a->dw, P.nitems, P.ncib, a->cout, a->cin);
    else
        hipLaunchKernelGGL((conv3d_k3_wgrad_reduce_kernel<bf16_t>), dim3((total + 63) / 64), dim3(256), 0, stream,
                           (const float*)P.part, (bf16_t*)a->dw, P.nitems, P.ncib, a->cout, a->cin);
    return (int)hipGetLastError();
}
