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

struct WgradDev {
    const char* x;   int64_t x_sb, x_sc, x_sz, x_sy;      // element strides, x contiguous
    const char* dy;  int64_t dy_sb, dy_sc, dy_sz, dy_sy;
    float* part;                                          // [co blk][ci blk][item][27][32][48]
    int32_t B, D, H, W, cin;
    int32_t nxb, ysplit, rows_per_part, nitems;
    int32_t ncob, ncib;
    int32_t ipw, nslab;                                   // work items per workgroup (round 5), partial slabs = ceil(nitems / ipw)
};

// One thread's share of the per-step global -> LDS copy: up to kCopies 16-byte granules.
struct WgCopy {
    int64_t src[kCopies];      // element offset from the row base (X granules: from the X row, dY granules: from the dY row)
    int32_t dst[kCopies];      // element offset inside an X ring slot / a dY buffer
    bool is_x[kCopies], live[kCopies], inside[kCopies];   // inside: the granule's x range is inside the volume
};

template <typename T, int NQ>
__global__ void __launch_bounds__(kWgThreads, 2) conv3d_k3_wgrad_v1_kernel(WgradDev P) {
    typedef typename Mfma16<T>::v8 frag8;
    constexpr int NCO = kWgCo / 16;
    __shared__ __attribute__((aligned(16))) T xs[4][kWgBlock][kPitch];
    __shared__ __attribute__((aligned(16))) T dys[2][kWgCo][kPitch];
    const int tid = threadIdx.x, lane = tid & 63;
    const int ky = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: row / tap arithmetic stays on the SALU
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
        WgCopy cp;
#pragma unroll
        for (int k = 0; k < kCopies; ++k) {
            const int id = tid + k * kWgThreads;
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
        auto fetch = [&](u32x4 (&r)[kCopies], int yy, int yd) {
            const bool x_ok = yy >= 0 && yy < P.H;
            const T* xr = xplane + (int64_t)(x_ok ? yy : 0) * P.x_sy;
            const T* dr = dyplane + (int64_t)(yd < P.H ? yd : 0) * P.dy_sy;
#pragma unroll
            for (int k = 0; k < kCopies; ++k) {
                const T* src = (cp.is_x[k] ? xr : dr) + cp.src[k];
                r[k] = *reinterpret_cast<const u32x4*>(src);
            }
        };
        auto park = [&](const u32x4 (&r)[kCopies], int yy, int slot, int buf) {
            const bool x_ok = yy >= 0 && yy < P.H;
#pragma unroll
            for (int k = 0; k < kCopies; ++k) {
                if (!cp.live[k]) continue;
                const bool keep = cp.is_x[k] ? (x_ok && cp.inside[k]) : cp.inside[k];
                T* dst = (cp.is_x[k] ? &xs[slot][0][0] : &dys[buf][0][0]) + cp.dst[k];
                *reinterpret_cast<u32x4*>(dst) = keep ? r[k] : zero4;
            }
        };

        // ---- prologue: rows y0 - 1, y0, y0 + 1 and dY row y0 ---------------------------------------------------------
        {
            u32x4 r[kCopies];
#pragma unroll
            for (int d = -1; d <= 1; ++d) {
                fetch(r, y0 + d, y0);
                park(r, y0 + d, (y0 + d + 4) & 3, 0);     // the dY row is parked three times (same data): harmless
            }
        }
        __syncthreads();

        // ---- main loop ------------------------------------------------------------------------------------------------
        for (int y = y0; y < y1; ++y) {
            u32x4 r[kCopies];
            fetch(r, y + 2, y + 1);                       // in flight during this step's MFMAs
            SEGM_SCHED_FENCE();
            const int slot = (y + ky - 1 + 4) & 3, buf = (y - y0) & 1;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
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

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: conv3d_k3_wgrad_kernel - the shipped kernel.  Same problem decomposition as v1 above (one tap plane kz, one 48 x 48
// channel block, one (batch, depth, x block, y range) work item per workgroup of three waves that walks y with the X rows and the
// dY row staged in LDS), a different row loop.  Three designs were built and measured on the way (the sources are kept as text in
// tools/experiments/conv3d_wgrad_r5_variants.hip.txt, the numbers in profiles/r05_wgrad_abl*.log, r05_wgrad_pmc*.log):
//   1. v1's loop by INSTRUCTION COUNT (~300 -> ~180 per row step: buffer loads whose range check does the zero padding, uniform copy
//      rounds, scalar row offsets): 0.686 -> 0.57 ms at 48 -> 48 @128^3.  Its ablations showed no single bound - without the row
//      fetch, without the MFMAs or without the fragment reads a launch still took 0.48 - 0.50 of 0.57 ms, a third workgroup per CU
//      gained 13 - 22 % - i.e. every wave sits through a serial chain per step: barrier -> fragment reads -> MFMAs -> park -> barrier.
//   2. the same loop software-pipelined ACROSS the barrier (rows parked at the top of the step after their fetch, the next step's
//      first fragments read before the barrier): 0.52 - 0.55 ms; parked-wave time halves (WAIT_ANY 0.32 -> 0.18 of the wave
//      cycles) but the LDS pipe stays 55 % busy, half of it BANK CONFLICTS - the two halo dwords of every X fragment are 4-way
//      conflicts no row pitch can avoid (all 32 lanes of a ds_read_b32 group address dword 3, or dword 0, of a granule).
//   3. ONE WAVE = ONE ci TILE, all three ky (this kernel): with ky = wave (v1) every X row is read from LDS by three waves, on
//      three consecutive steps, and shifted three times.  Here wave w owns ci tile w: an X row is read ONCE, by the one wave that
//      needs it (16 of the 48 channels), kept in registers for the three steps it is used (ky = 2, then 1, then 0 - the compiler
//      keeps the shifted operands too), and only the dY fragments (all 48 co, every wave) are re-read per step:
//          LDS reads per wave and step   6 dY + 6 X fragments + 6 halo pairs  ->  6 dY + 2 X fragments + 2 halo pairs
//          vector instructions           48 funnel shifts / moves             ->  16
//          X ring                        4 rows                               ->  2 (the row being read, the row being parked)
//      acc[ky][kx][co tile] for the wave's ci tile: the same 27 accumulators and 54 MFMAs per step; ~125 instructions per step.
//      0.47 ms (1108 TF/s) at 48 -> 48 @128^3, 0.23 ms at 96 -> 96 @64^3 (v1: 0.675 / 0.312); LDS busy 25 %, conflicts 10 %.
// Common to all three: every row is a BUFFER load (descriptor = the (batch, plane, channel block) base, lane = a constant 32-bit
// byte offset computed once per work item, row = a scalar byte offset); a granule outside the volume (x halo beyond the edge,
// channel >= cin, pad granule) carries an offset beyond num_records and reads zeros, a row outside the volume is fetched through
// the same descriptor with num_records = 0 - park is one ds_write_b128 per granule, no select, no branch; flat granule ids make
// the LDS destination of copy round k an immediate offset from ONE address register.  The grid is 1-D over (item group, co block,
// ci block, kz) with kz fastest, mapped so that every XCD owns a contiguous range: the workgroups that read the same rows - the
// three tap planes and the ci blocks (dY), the co blocks and the z neighbours (X) - are resident together in one L2 (hit rate 0.71).
// A workgroup may accumulate `ipw` work items (strided by the number of groups, so that resident workgroups stay on neighbouring
// items) before it writes its 27 x 48 x 48 partial block: fewer partial slabs for the reduce kernel when items are short.
// What bounds it now (profiles/r05_wgrad_pmc3.log): WAIT_INST_ANY 0.49 of the wave cycles with the matrix pipe busy 0.39 per wave -
// six waves on four SIMDs (2, 2, 1, 1): the two shared SIMDs run at ~77 % of the matrix pipe, the other two at half of that.
// Two attempts to even that out were measured and lost (tools/experiments/README.md): 32-wide blocks with three / four workgroups
// per CU (168 registers: spills, 1.10 ms) and ONE WAVE PER WORKGROUP with every fragment loaded straight from memory - no LDS, no
// barrier, two independent waves per SIMD (conv3d_wgrad_direct_r5.hip.txt: 1.12 ms against 0.52 - twelve 1-KB vector loads per
// wave and step in place of nine LDS reads are what the CU's load path cannot feed).
// SEGM_WGRAD_V1=1 launches the round-1..4 kernel (A/B, and the fallback for strides beyond 32-bit byte offsets).
// ---------------------------------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t wg_rsrc_t;
constexpr uint32_t kWgNumRec = 0xFFFFF000u;        // bytes a descriptor covers; a lane offset at / beyond it reads zeros
constexpr uint32_t kWgOob = 0xFFFFF000u;

template <int NQ> struct WgGeo {
    static constexpr int PG = 4 * NQ + 3;          // granules per LDS row: left halo, 4 NQ data, right halo, one pad (bank spread)
    static constexpr int PITCH = 8 * PG;           // elements: 88 (44 dwords) / 56 (28 dwords) - 16 fragment lanes hit 16 distinct 4-bank groups
    static constexpr int XG = 4 * NQ + 2;          // granules fetched per X row
    static constexpr int DG = 4 * NQ;              // granules per dY row
    static constexpr int RX = (48 * PG + kWgThreads - 1) / kWgThreads;    // fetch rounds per X row: 3 / 2
    static constexpr int RD = 48 * DG / kWgThreads;                       // per dY row: 2 / 1
    static constexpr int XSLOT = RX * kWgThreads * 8;     // elements per ring slot = whole rounds of flat granule ids (tail unused)
    static constexpr int DSLOT = 48 * PITCH;
    static constexpr int NR = RX + RD;
};

template <typename T, int NQ, int OCC = 2>
__global__ void __launch_bounds__(kWgThreads, OCC) conv3d_k3_wgrad_kernel(WgradDev P) {
    typedef typename Mfma16<T>::v8 frag8;
    typedef WgGeo<NQ> G;
    constexpr int NCO = kWgCo / 16;
    __shared__ __attribute__((aligned(16))) T xs[2 * G::XSLOT];
    __shared__ __attribute__((aligned(16))) T dys[3 * G::DSLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);          // ci tile of this wave
    const int i16 = lane & 15, g = lane >> 4;
    const int vid = xcd_item(blockIdx.x, gridDim.x);
    const int kz = vid % 3;
    const int cib = (vid / 3) % P.ncib, cob = (vid / (3 * P.ncib)) % P.ncob;
    const int grp = vid / (3 * P.ncib * P.ncob);
    constexpr int XB = 32 * NQ;

    f32x4 acc[3][3][NCO];                                 // [ky][kx][co tile]
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int d = 0; d < NCO; ++d) acc[a][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int xrd = (wv * 16 + i16) * G::PITCH + 8 + 8 * g;            // this lane's X fragment (chunk 0) inside a ring slot
    const int yrd = i16 * G::PITCH + 8 * g;                          // its dY fragment of (co tile 0, chunk 0) inside a dY buffer
    const int dco = tid / G::DG, dgr = tid - dco * G::DG;
    const int ddst = dco * G::PITCH + 8 * dgr;
    const uint32_t x_sy2 = (uint32_t)(P.x_sy * 2), dy_sy2 = (uint32_t)(P.dy_sy * 2);

    struct XRow { u32x4 v[NQ]; uint32_t hl[NQ], hr[NQ]; };            // one X row of the wave's ci tile: fragments + halo dwords
    auto read_xrow = [&](XRow& r_, int slot) {
        const T* p = xs + slot * G::XSLOT + xrd;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            r_.v[q] = *reinterpret_cast<const u32x4*>(p + 32 * q);
            r_.hl[q] = *reinterpret_cast<const uint32_t*>(p + 32 * q - 2);
            r_.hr[q] = *reinterpret_cast<const uint32_t*>(p + 32 * q + 8);
        }
    };
    auto read_a = [&](u32x4 (&a)[NCO], int buf, int q) {
        const T* p = dys + buf * G::DSLOT + yrd + 32 * q;
#pragma unroll
        for (int co = 0; co < NCO; ++co) a[co] = *reinterpret_cast<const u32x4*>(p + co * 16 * G::PITCH);
    };
    auto mm = [&](const u32x4 (&a)[NCO], const XRow& r_, int q, int kyi) {
        const u32x4 v = r_.v[q];
        const uint32_t s1 = __builtin_amdgcn_alignbyte(v[1], v[0], 2);
        const uint32_t s2 = __builtin_amdgcn_alignbyte(v[2], v[1], 2);
        const uint32_t s3 = __builtin_amdgcn_alignbyte(v[3], v[2], 2);
        const u32x4 vl = {__builtin_amdgcn_alignbyte(v[0], r_.hl[q], 2), s1, s2, s3};   // X[x-1 ..]  (kx = 0)
        const u32x4 vr = {s1, s2, s3, __builtin_amdgcn_alignbyte(r_.hr[q], v[3], 2)};   // X[x+1 ..]  (kx = 2)
#pragma unroll
        for (int co = 0; co < NCO; ++co) {
            const frag8 af = __builtin_bit_cast(frag8, a[co]);
            acc[kyi][0][co] = Mfma16<T>::run(af, __builtin_bit_cast(frag8, vl), acc[kyi][0][co]);
            acc[kyi][1][co] = Mfma16<T>::run(af, __builtin_bit_cast(frag8, v), acc[kyi][1][co]);
            acc[kyi][2][co] = Mfma16<T>::run(af, __builtin_bit_cast(frag8, vr), acc[kyi][2][co]);
        }
    };

    for (int it = 0; it < P.ipw; ++it) {
        const int item_id = it * P.nslab + grp;
        if (item_id >= P.nitems) break;
        int item = item_id;
        const int ypart = item % P.ysplit;  item /= P.ysplit;
        const int xb = item % P.nxb;        item /= P.nxb;
        const int z = item % P.D, b = item / P.D;
        const int zz = z + kz - 1;
        const int y0 = ypart * P.rows_per_part;
        const int y1 = (y0 + P.rows_per_part < P.H) ? y0 + P.rows_per_part : P.H;
        if (zz < 0 || zz >= P.D || y1 <= y0) continue;
        const int x0 = xb * XB;

        uint32_t xv[G::RX], dv[G::RD];
#pragma unroll
        for (int k = 0; k < G::RX; ++k) {
            const int id = tid + k * kWgThreads;
            const int ci = id / G::PG, gr = id - ci * G::PG;
            const int xg = x0 - 8 + 8 * gr;
            const bool ok = id < 48 * G::PG && gr < G::XG && xg >= 0 && xg < P.W && cib * kWgBlock + ci < P.cin;
            xv[k] = ok ? (uint32_t)(((int64_t)ci * P.x_sc + xg) * 2) : kWgOob;
        }
#pragma unroll
        for (int k = 0; k < G::RD; ++k) {
            const int co = dco + k * (kWgThreads / G::DG);
            const bool ok = x0 + 8 * dgr < P.W;
            dv[k] = ok ? (uint32_t)(((int64_t)co * P.dy_sc + x0 + 8 * dgr) * 2) : kWgOob;
        }
        const T* xbase = reinterpret_cast<const T*>(P.x) + (int64_t)b * P.x_sb + (int64_t)zz * P.x_sz + (int64_t)cib * kWgBlock * P.x_sc;
        const T* dbase = reinterpret_cast<const T*>(P.dy) + (int64_t)b * P.dy_sb + (int64_t)z * P.dy_sz + (int64_t)cob * kWgCo * P.dy_sc;

        auto fetch = [&](u32x4 (&r)[G::NR], int yy, int yd) {
            const bool x_ok = yy >= 0 && yy < P.H, d_ok = yd < P.H;
            const wg_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(xbase), 0, x_ok ? (int)kWgNumRec : 0, 0x00020000);
            const wg_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(dbase), 0, d_ok ? (int)kWgNumRec : 0, 0x00020000);
            const uint32_t sx = (uint32_t)(x_ok ? yy : 0) * x_sy2, sd = (uint32_t)(d_ok ? yd : 0) * dy_sy2;
#pragma unroll
            for (int k = 0; k < G::RX; ++k) r[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, xv[k], sx, 0);
#pragma unroll
            for (int k = 0; k < G::RD; ++k) r[G::RX + k] = __builtin_amdgcn_raw_buffer_load_b128(rd, dv[k], sd, 0);
        };
        auto park = [&](const u32x4 (&r)[G::NR], int slot, int buf) {
            T* xd = xs + slot * G::XSLOT + tid * 8;
#pragma unroll
            for (int k = 0; k < G::RX; ++k) *reinterpret_cast<u32x4*>(xd + k * kWgThreads * 8) = r[k];
            T* dd = dys + buf * G::DSLOT + ddst;
#pragma unroll
            for (int k = 0; k < G::RD; ++k) *reinterpret_cast<u32x4*>(dd + k * (kWgThreads / G::DG) * G::PITCH) = r[G::RX + k];
        };

        // ---- prologue: rows y0 - 1, y0 -> registers (through slots 0, 1); row y0 + 1 -> slot 0; row y0 + 2 in flight;
        //      dY rows y0, y0 + 1 -> buffers 0, 1; dY row y0 + 2 in flight --------------------------------------------------------
        XRow rows[3];                                     // rows[(rel + ky) % 3] = X row y + ky - 1 at step rel = y - y0
        u32x4 a0[NCO], a1[NCO];                           // dY fragments of the step: chunk 0 (read before the barrier), chunk 1
        u32x4 r[G::NR];
        fetch(r, y0 - 1, y0);     park(r, 0, 0);
        fetch(r, y0, y0 + 1);     park(r, 1, 1);
        __syncthreads();
        read_xrow(rows[0], 0);
        read_xrow(rows[1], 1);
        read_a(a0, 0, 0);
        fetch(r, y0 + 1, y0 + 1);
        __syncthreads();                                  // (drains the reads above) slots 0, 1 may be overwritten
        park(r, 0, 1);
        fetch(r, y0 + 2, y0 + 2);
        __syncthreads();

        for (int yb = y0; yb < y1; yb += 3) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int y = yb + u;
                if (y < y1) {                             // uniform
                    const int rel = y - y0;               // rel % 3 == u
                    park(r, (rel + 1) & 1, (u + 2) % 3);  // X row y + 2, dY row y + 2: fetched during the previous step
                    fetch(r, y + 3, y + 3);
                    read_xrow(rows[(u + 2) % 3], rel & 1);                // X row y + 1: visible since the previous barrier
                    if constexpr (NQ == 2) read_a(a1, u, 1);
                    SEGM_SCHED_FENCE();
                    mm(a0, rows[u], 0, 0);
                    mm(a0, rows[(u + 1) % 3], 0, 1);
                    SEGM_SCHED_FENCE();                   // row y + 1 arrives behind these 18 MFMAs
                    mm(a0, rows[(u + 2) % 3], 0, 2);
                    SEGM_SCHED_FENCE();
                    read_a(a0, (u + 1) % 3, 0);           // the next step's first chunk (dY row y + 1: parked a step ago)
                    SEGM_SCHED_FENCE();
                    if constexpr (NQ == 2) {
                        mm(a1, rows[u], 1, 0);
                        mm(a1, rows[(u + 1) % 3], 1, 1);
                        mm(a1, rows[(u + 2) % 3], 1, 2);
                    }
                    SEGM_SCHED_FENCE();
                    __syncthreads();
                }
            }
        }
    }
    // partial block: part[((cob * ncib + cib) * nslab + grp)][tap = kz*9 + ky*3 + kx][co (48)][ci (48)], this wave: ci tile wv
    float* out = P.part + ((((int64_t)cob * P.ncib + cib) * P.nslab + grp) * 27) * (kWgCo * kWgBlock);
#pragma unroll
    for (int kyi = 0; kyi < 3; ++kyi)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ct = 0; ct < NCO; ++ct)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int co = ct * 16 + g * 4 + rr, cin = wv * 16 + i16;
                    out[((int64_t)(kz * 9 + kyi * 3 + kx) * kWgCo + co) * kWgBlock + cin] = acc[kyi][kx][ct][rr];
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

// work decomposition: items = (batch, depth) planes x 64-wide (32 if W % 64) x blocks x y parts; a workgroup of the round-5 kernel
// accumulates `ipw` consecutive items (one resident round of workgroups: 256 CUs x 2)
struct WgPlan { int nq, nxb, ysplit, rows_per_part, nitems, ipw, nslab; };
static bool wgrad_v1() {
    const char* e = getenv("SEGM_WGRAD_V1");           // read per call (tests switch it): a getenv is nothing next to a launch
    return e && atoi(e) == 1;
}
static WgPlan wgrad_plan(int batch, int cin, int cout, int d, int h, int w) {
    WgPlan p;
    p.nq = (w % 64 == 0) ? 2 : 1;                         // (32-wide blocks everywhere, three workgroups per CU: measured, 0.63 / 1.10 ms
                                                          // against 0.50 ms at 48 -> 48 @128^3 - profiles/r05_wgrad_ab3.log)
    p.nxb = (w + 32 * p.nq - 1) / (32 * p.nq);
    const int64_t blocks = (int64_t)3 * ((cout + kWgCo - 1) / kWgCo) * ((cin + kWgBlock - 1) / kWgBlock);
    const int64_t wgs = (int64_t)batch * d * p.nxb * blocks;
    int split = 1;                                        // cut y when there are too few workgroups to fill 256 CUs x 3
    while (wgs * split < 1536 && h / (split * 2) >= 8) split *= 2;
    p.ysplit = split;
    p.rows_per_part = (h + split - 1) / split;
    p.nitems = batch * d * p.nxb * split;
    const char* ipw_s = getenv("SEGM_WGRAD_IPW");
    const int ipw_env = ipw_s ? atoi(ipw_s) : 0;
    // items per workgroup (profiles/r05_wgrad_abl3.log): many short-lived workgroups balance themselves - one resident round of
    // long-lived ones loses to its own tail (513 workgroups on 512 slots = two rounds) - so only short items (H < 128 rows: the
    // prologue and the 108-store epilogue weigh more) are paired up, and never below ~768 workgroups
    int64_t ipw = ipw_env > 0 ? ipw_env : 128 / (p.rows_per_part > 0 ? p.rows_per_part : 1);
    if (ipw_env <= 0 && ipw > ((int64_t)p.nitems * blocks) / 768) ipw = ((int64_t)p.nitems * blocks) / 768;
    if (wgrad_v1() || ipw < 1) ipw = 1;
    if (ipw > p.nitems) ipw = p.nitems;
    p.ipw = (int)ipw;
    p.nslab = (p.nitems + p.ipw - 1) / p.ipw;
    return p;
}

}  // namespace segm

using namespace segm;

extern "C" size_t segm_conv3d_k3_wgrad_workspace_bytes(int32_t batch, int32_t cin, int32_t cout, int32_t d, int32_t h, int32_t w) {
    if (batch <= 0 || cin <= 0 || cout <= 0 || d <= 0 || h <= 0 || w <= 0) return 0;
    const WgPlan pl = wgrad_plan(batch, cin, cout, d, h, w);
    return (size_t)((cout + kWgCo - 1) / kWgCo) * ((cin + kWgBlock - 1) / kWgBlock) * pl.nslab * 27 * kWgCo * kWgBlock * sizeof(float);
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
    P.ipw = pl.ipw; P.nslab = pl.nslab;
    hipStream_t stream = (hipStream_t)a->stream;
    // the round-5 kernel addresses a (batch, plane, 48-channel block) through 32-bit byte offsets: 48 channel strides and a plane's
    // rows must stay below the descriptor's 0xFFFFF000 bytes (volumes up to ~350^3 elements per channel); beyond: the v1 kernel
    const uint64_t lim = kWgNumRec - 4096;
    const bool fits32 = (uint64_t)a->x_stride_c * 96 + 2 * (uint64_t)a->width < lim && (uint64_t)a->dy_stride_c * 96 + 2 * (uint64_t)a->width < lim &&
                        (uint64_t)a->x_stride_y * 2 * (uint64_t)a->height < lim && (uint64_t)a->dy_stride_y * 2 * (uint64_t)a->height < lim;
    if (!fits32 && !wgrad_v1()) return SEGM_E_SHAPE;
    {
        const dim3 grid = wgrad_v1() ? dim3(P.nslab * 3 * P.ncob, P.ncib) : dim3(P.nslab * 3 * P.ncob * P.ncib);
        if (wgrad_v1()) {
            if (a->dtype == SEGM_F16) {
                if (pl.nq == 2) hipLaunchKernelGGL((conv3d_k3_wgrad_v1_kernel<f16_t, 2>), grid, dim3(kWgThreads), 0, stream, P);
                else hipLaunchKernelGGL((conv3d_k3_wgrad_v1_kernel<f16_t, 1>), grid, dim3(kWgThreads), 0, stream, P);
            } else {
                if (pl.nq == 2) hipLaunchKernelGGL((conv3d_k3_wgrad_v1_kernel<bf16_t, 2>), grid, dim3(kWgThreads), 0, stream, P);
                else hipLaunchKernelGGL((conv3d_k3_wgrad_v1_kernel<bf16_t, 1>), grid, dim3(kWgThreads), 0, stream, P);
            }
        } else if (a->dtype == SEGM_F16) {
            if (pl.nq == 2) hipLaunchKernelGGL((conv3d_k3_wgrad_kernel<f16_t, 2>), grid, dim3(kWgThreads), 0, stream, P);
            else hipLaunchKernelGGL((conv3d_k3_wgrad_kernel<f16_t, 1>), grid, dim3(kWgThreads), 0, stream, P);
        } else {
            if (pl.nq == 2) hipLaunchKernelGGL((conv3d_k3_wgrad_kernel<bf16_t, 2>), grid, dim3(kWgThreads), 0, stream, P);
            else hipLaunchKernelGGL((conv3d_k3_wgrad_kernel<bf16_t, 1>), grid, dim3(kWgThreads), 0, stream, P);
        }
    }
    const int total = a->cout * a->cin * 27;
    // 16-byte loads where they measured faster (profiles/r04_call9_ab.log: 96 -> 96 @64^3 0.368 -> 0.336 ms; 48 -> 48 @128^3 0.69 ->
    // 0.72 ms - one (co, ci) block pair has too few 1 KB pieces per slab to fill the chip); SEGM_WGRAD_REDUCE4 = 0 / 1 forces a side
    static const int wide = [] { const char* e = getenv("SEGM_WGRAD_REDUCE4"); return e ? atoi(e) : -1; }();
    if (a->cin % 4 == 0 && (wide == 1 || (wide < 0 && (int64_t)a->cin * a->cout >= 96 * 96))) {
        const dim3 g4((total / 4 + 63) / 64);
        if (a->dw_dtype == SEGM_F32)
            hipLaunchKernelGGL((conv3d_k3_wgrad_reduce4_kernel<float>), g4, dim3(512), 0, stream, (const float*)P.part, (float*)a->dw, P.nslab, P.ncib, a->cout, a->cin);
        else if (a->dw_dtype == SEGM_F16)
            hipLaunchKernelGGL((conv3d_k3_wgrad_reduce4_kernel<f16_t>), g4, dim3(512), 0, stream, (const float*)P.part, (f16_t*)a->dw, P.nslab, P.ncib, a->cout, a->cin);
        else
            hipLaunchKernelGGL((conv3d_k3_wgrad_reduce4_kernel<bf16_t>), g4, dim3(512), 0, stream, (const float*)P.part, (bf16_t*)a->dw, P.nslab, P.ncib, a->cout, a->cin);
        return (int)hipGetLastError();
    }
    if (a->dw_dtype == SEGM_F32)
        hipLaunchKernelGGL((conv3d_k3_wgrad_reduce_kernel<float>), dim3((total + 63) / 64), dim3(256), 0, stream,
                           (const float*)P.part, (float*)a->dw, P.nslab, P.ncib, a->cout, a->cin);
    else if (a->dw_dtype == SEGM_F16)
        hipLaunchKernelGGL((conv3d_k3_wgrad_reduce_kernel<f16_t>), dim3((total + 63) / 64), dim3(256), 0, stream,
                           (const float*)P.part, (f16_t*)a->dw, P.nslab, P.ncib, a->cout, a->cin);
    else
        hipLaunchKernelGGL((conv3d_k3_wgrad_reduce_kernel<bf16_t>), dim3((total + 63) / 64), dim3(256), 0, stream,
                           (const float*)P.part, (bf16_t*)a->dw, P.nslab, P.ncib, a->cout, a->cin);
    return (int)hipGetLastError();
}
