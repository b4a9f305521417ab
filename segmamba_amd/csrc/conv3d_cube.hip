// 3x3x3 stride-1 pad-1 convolution of WIDE layers on SMALL volumes (C ABI: segm_conv3d_k3_cube_fwd): NCDHW 16-bit activations,
// Cin % 32 == 0, Cout a multiple of 64 or 96, depth / height / width multiples of 8 - the 16^3 and 8^3 levels of SegMamba's encoder / decoder
// (192 ... 768 channels; reference model_segmamba/segmamba.py:91-132 GSC blocks, monai/networks/blocks/dynunet_block.py:44-111
// UnetResBlock, unetr_block.py:82-84 decoder blocks; torch.nn.Conv3d -> cuDNN there).  Forward and, with the image of
// flip(W)^T, data gradient.
//
// Why another kernel: csrc/conv3d_fwd.hip walks ROWS of 32 / 64 voxels with one 48-channel input block per launch - built for the
// 128-wide rows where 60 % of the convolution flops are.  At 16^3 a 384 -> 384 layer is 8 launches that each re-read and re-write
// the whole output for 1/8 of the contraction, on rows half as wide as the kernel's tile: 120 - 300 TF/s; the 8^3 layers went to
// the vendor's im2col + GEMM (60 - 340 TF/s, one GEMM per sample).  Together 5.4 ms of a 53 ms step for 7 % of its flops
// (profiles/r06_conv_layers.txt).  Here the unit of work is a CUBE:
//
//   * a workgroup (8 waves) owns 8 x 8 x 8 output voxels x NB output channels and walks the input channels in rounds of 32:
//     per round the 10 x 10 x 10 halo cube of 32 channels is staged in LDS VOXEL-major (channels of a voxel contiguous - what
//     the MFMA operand wants; NCDHW has x contiguous, so the transposition happens once per staged element in the LDS write, two
//     channels per 32-bit write), zeros where the cube leaves the volume, and then serves all 27 taps: the fragment of tap
//     (kz, ky, kx) is the same LDS image at a voxel offset - an immediate in the ds_read_b128;
//   * implicit GEMM per tap and round: M = 512 voxels (32 tiles of 4 x 4 in a z plane), N = NB, K = 32.  Wave (wm, wn) owns the two
//     z planes 2 wm, 2 wm + 1 (8 tiles) x NT column tiles: 8 input fragments from LDS and NT weight fragments straight from global
//     memory (host-arranged as fragments: 1 KB contiguous per wave and load) feed 8 NT MFMAs;
//   * LDS row / plane pitch 12 / 120 voxels of 96 bytes: with 4 x 4 tiles every ds_read_b128 of every tap is conflict-free
//     (tools/lds_conflicts.py model: 4.0 cycles); 115 KB, one workgroup per CU;
//   * small volumes have few cubes, so the contraction is SPLIT over workgroups (`splits` ranges of rounds): every workgroup stores
//     fp32 partial sums, and a second launch adds them in a fixed order, adds the bias (and, in accumulate mode, the existing
//     values - the later parts of a concatenated input) and rounds once.  No atomics: results are run-to-run identical.
//
// v_mfma_f32_16x16x32: A[i][k]: lane l holds A[i = l & 15][8 (l >> 4) .. +7]; B[k][j]: lane l holds B[8 (l >> 4) .. +7][j = l & 15];
// D[row = 4 (l >> 4) + r][col = l & 15].  Here row = voxel (4 x 4 tile: row = 4 y + x), col = output channel: a lane ends up with
// four x-consecutive voxels of one channel = one 16-byte store into the partial sums.
#include <stdlib.h>
#include <string.h>

#include "segm_device.h"

namespace segm {

typedef uint32_t cube_u32x4 __attribute__((ext_vector_type(4)));

#ifndef SEGM_CUBE_BRING
#define SEGM_CUBE_BRING 3
#endif
constexpr int kCubeP = 96;                    // LDS bytes per staged voxel: 32 channels + 32 bytes of padding
constexpr int kCubeRow = 12;                  // voxels between rows of the staged cube (10 used)
constexpr int kCubePlane = 120;               // voxels between planes
constexpr int kCubeLds = 10 * kCubePlane * kCubeP;
constexpr int kCubePairRows = 10 * 10 * 16;   // (z', y', channel pair) rows of one round
constexpr int kCubeIters = (kCubePairRows + 511) / 512;

struct CubeDev {
    const char* x;  int64_t x_sb, x_sc, x_sz, x_sy;      // element strides; x contiguous
    const char* wimg;                                    // [Cout / 16][R][27][64][8] fragments (segm_conv3d_k3_cube_pack_index)
    float* part;                                         // [S][B][Cout][D][H][W]
    int32_t B, Cin, Cout, D, H, W;
    int32_t R, S, tz, ty, tx, ncb;
    // EPI 1 / 2 (one split: no partial sums, no second launch): the finished values go straight to y
    char* y;  int64_t y_sb, y_sc, y_sz, y_sy;
    const float* bias;
    float* stats;                                        // (B * Cout, vol / 128, 4) {count, sum, sum of squares, -} per wave, or null
};

template <typename T> struct CubeStage {                 // one thread's share of a round on its way from global memory to LDS
    cube_u32x4 a[kCubeIters], b[kCubeIters];             // channels 2 cp and 2 cp + 1: eight x of the row
    uint32_t hl[kCubeIters], hr[kCubeIters];             // the voxels left and right of them, both channels packed
};

// NT = 16-channel tiles per wave; NB = 32 NT output channels per workgroup (two waves side by side in N, four in M)
// EPI: 0 = fp32 partial sums of this split (the reduction launch finishes them); 1 = the only split: + bias, rounded, stored to y
// (+ the InstanceNorm partials of what is stored, per wave = 128 voxels of a channel); 2 = the same added to what y holds
template <typename T, int NT, int EPI>
__global__ void __launch_bounds__(512, 1) conv3d_k3_cube_kernel(CubeDev P) {
    typedef typename Mfma16<T>::v8 frag8;
    constexpr int NB = 32 * NT;
    __shared__ __attribute__((aligned(16))) unsigned char s_x[kCubeLds];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;

    // work item: spatial cube fastest (the cubes of one (channel block, split) share its weights in one L2), then block, then split
    const int item = xcd_item(blockIdx.x, gridDim.x);
    const int ncube = P.tz * P.ty * P.tx;
    const int nsp = ncube * P.B;
    const int sp = item % nsp, rest = item / nsp;
    const int cb = rest % P.ncb, s = rest / P.ncb;
    const int b = sp / ncube, cube = sp - b * ncube;
    const int cz = cube / (P.ty * P.tx), cy = (cube / P.tx) % P.ty, cx = cube % P.tx;
    const int z0 = 8 * cz, y0 = 8 * cy, x0 = 8 * cx;
    const int r0 = (int)((int64_t)s * P.R / P.S), r1 = (int)((int64_t)(s + 1) * P.R / P.S);
    const bool has_left = x0 > 0, has_right = x0 + 8 < P.W;

    // ---- staging: pair row q = (z', y', cp) -> thread; the same rows every round, only the channel base moves
    int64_t src_off[kCubeIters];                          // element offset of (b, channel 2 cp, row) from P.x at round 0; < 0: outside
    uint32_t dst_off[kCubeIters];
#pragma unroll
    for (int it = 0; it < kCubeIters; ++it) {
        const int q = it * 512 + tid;
        const int cp = q & 15, yz = q >> 4, yp = yz % 10, zp = yz / 10;
        const int gz = z0 - 1 + zp, gy = y0 - 1 + yp;
        const bool live = q < kCubePairRows && gz >= 0 && gz < P.D && gy >= 0 && gy < P.H;
        src_off[it] = live ? (int64_t)b * P.x_sb + (int64_t)(2 * cp) * P.x_sc + (int64_t)gz * P.x_sz + (int64_t)gy * P.x_sy + x0 : -1;
        dst_off[it] = q < kCubePairRows ? (uint32_t)((zp * kCubePlane + yp * kCubeRow) * kCubeP + cp * 4) : 0xffffffffu;
    }
    CubeStage<T> st;
    auto load_round = [&](int r) {
        const int64_t cbase = (int64_t)r * 32 * P.x_sc;
#pragma unroll
        for (int it = 0; it < kCubeIters; ++it) {
            st.a[it] = cube_u32x4{0, 0, 0, 0}; st.b[it] = cube_u32x4{0, 0, 0, 0}; st.hl[it] = 0; st.hr[it] = 0;
            if (src_off[it] >= 0) {
                const uint16_t* pa = reinterpret_cast<const uint16_t*>(P.x) + src_off[it] + cbase;
                const uint16_t* pb = pa + P.x_sc;
                st.a[it] = *reinterpret_cast<const cube_u32x4*>(pa);
                st.b[it] = *reinterpret_cast<const cube_u32x4*>(pb);
                if (has_left) st.hl[it] = (uint32_t)pa[-1] | ((uint32_t)pb[-1] << 16);
                if (has_right) st.hr[it] = (uint32_t)pa[8] | ((uint32_t)pb[8] << 16);
            }
        }
    };
    auto store_round = [&]() {
#pragma unroll
        for (int it = 0; it < kCubeIters; ++it) {
            if (dst_off[it] != 0xffffffffu) {
                unsigned char* d = s_x + dst_off[it];
                *reinterpret_cast<uint32_t*>(d) = st.hl[it];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t a = st.a[it][j], bb = st.b[it][j];
                    *reinterpret_cast<uint32_t*>(d + (1 + 2 * j) * kCubeP) = (a & 0xffffu) | (bb << 16);
                    *reinterpret_cast<uint32_t*>(d + (2 + 2 * j) * kCubeP) = (a >> 16) | (bb & 0xffff0000u);
                }
                *reinterpret_cast<uint32_t*>(d + 9 * kCubeP) = st.hr[it];
            }
        }
    };

    // ---- fragments: lane (m = voxel of the 4 x 4 tile, g = channel group of 8)
    const int m16 = lane & 15, g = lane >> 4;
    const unsigned char* a_base = s_x + (2 * wm * kCubePlane + (m16 >> 2) * kCubeRow + (m16 & 3)) * kCubeP + g * 16;
    const char* w_base = P.wimg + ((int64_t)(cb * (NB / 16) + wn * NT) * P.R * 27 * 64 + lane) * 16;
    const int64_t w_tile = (int64_t)P.R * 27 * 1024;       // bytes between column tiles

    mfma_f32x4 acc[8][NT];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma_f32x4{0.f, 0.f, 0.f, 0.f};

    if (r0 < r1) load_round(r0);
    for (int r = r0; r < r1; ++r) {
        __syncthreads();                                   // the previous round's fragments are all read
        store_round();
        __syncthreads();
        if (r + 1 < r1) load_round(r + 1);                 // in flight under this round's MFMAs
        const char* wr = w_base + (int64_t)r * 27 * 1024;
        // software pipeline over the 27 taps: the input fragments of tap + 1 (LDS) and the weight fragments of tap + 2 (global / L2)
        // are requested before the MFMAs of tap; the fence keeps the compiler from hoisting more than that (27 x 8 reads spill)
        constexpr int BR = SEGM_CUBE_BRING;               // weight fragments of BR - 1 taps in flight (global memory / L2: the latency the counters show the waves waiting on)
        frag8 av[2][8], bw[BR][NT];
        auto read_a = [&](int tap, frag8* dst) {
            const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int vox = (i >> 2) * kCubePlane + ((i >> 1) & 1) * 4 * kCubeRow + (i & 1) * 4 + kz * kCubePlane + ky * kCubeRow + kx;
                dst[i] = *reinterpret_cast<const frag8*>(a_base + vox * kCubeP);
            }
        };
        auto read_b = [&](int tap, frag8* dst) {
#pragma unroll
            for (int j = 0; j < NT; ++j) dst[j] = *reinterpret_cast<const frag8*>(wr + tap * 1024 + j * w_tile);
        };
#pragma unroll
        for (int t = 0; t + 1 < BR; ++t) read_b(t, bw[t]);
        constexpr bool PIPE_A = NT < 4;                    // 128 accumulator registers leave no room for a second set of input fragments
        if (PIPE_A) read_a(0, av[0]);
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            if (tap + BR - 1 < 27) read_b(tap + BR - 1, bw[(tap + BR - 1) % BR]);
            if (PIPE_A) { if (tap + 1 < 27) read_a(tap + 1, av[(tap + 1) & 1]); }
            else read_a(tap, av[0]);
            const int cur = PIPE_A ? (tap & 1) : 0;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = Mfma16<T>::run(av[cur][i], bw[tap % BR][j], acc[i][j]);
            SEGM_SCHED_FENCE();
        }
    }

    // ---- lane = (channel m16 of the column tile, row g of the 4 x 4 tile): four x-consecutive voxels
    const int64_t plane = (int64_t)P.H * P.W, vol = (int64_t)P.D * plane;
    if (EPI == 0) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int co = cb * NB + (wn * NT + j) * 16 + m16;
            float* pc = P.part + (((int64_t)s * P.B + b) * P.Cout + co) * vol;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int z = z0 + 2 * wm + (i >> 2), y = y0 + ((i >> 1) & 1) * 4 + g, x = x0 + (i & 1) * 4;
                *reinterpret_cast<mfma_f32x4*>(pc + z * plane + (int64_t)y * P.W + x) = acc[i][j];
            }
        }
    } else {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int co = cb * NB + (wn * NT + j) * 16 + m16;
            const float bv = P.bias ? P.bias[co] : 0.f;
            T* yc = reinterpret_cast<T*>(P.y) + (int64_t)b * P.y_sb + (int64_t)co * P.y_sc;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int z = z0 + 2 * wm + (i >> 2), y = y0 + ((i >> 1) & 1) * 4 + g, x = x0 + (i & 1) * 4;
                T* dst = yc + (int64_t)z * P.y_sz + (int64_t)y * P.y_sy + x;
                T o[4];
                float v[4];
                if (EPI == 2) {
                    const u32x2 old = *reinterpret_cast<const u32x2*>(dst);
                    memcpy(o, &old, 8);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] = acc[i][j][q] + bv + (EPI == 2 ? to_f32(o[q]) : 0.f);
                    s1 += v[q]; s2 = fmaf(v[q], v[q], s2);
                    o[q] = from_f32<T>(v[q]);
                }
                u32x2 pk;
                memcpy(&pk, o, 8);
                *reinterpret_cast<u32x2*>(dst) = pk;
            }
            if (P.stats) {                                 // sum over the four rows of the tiles (lane groups); a wave = 128 voxels
                s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
                s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                if (g == 0)
                    reinterpret_cast<float4*>(P.stats)[((int64_t)b * P.Cout + co) * (vol / 128) + (int64_t)cube * 4 + wm] = float4{128.f, s1, s2, 0.f};
            }
        }
    }
}

struct CubeReduceDev {
    const float* part;                                   // [S][B][Cout][vol]
    const float* bias;
    char* y;  int64_t y_sb, y_sc, y_sz, y_sy;            // element strides
    int32_t S, B, Cout, D, H, W;
    int64_t n8;                                          // B * Cout * vol / 8
    float* stats;                                        // STATS: (B * Cout, vol / 512, 4) {count, sum, sum of squares, -} of what is stored
};

// y = sum over splits (fixed order) + bias (+ y): eight x-consecutive voxels per thread (W % 8 == 0: never across a row)
// STATS: the InstanceNorm behind the convolution merges the per-wave sums (a wave = 512 consecutive voxels of one (batch, channel)
// instance, vol % 512 == 0) instead of reading the volume again - the epilogue csrc/conv3d_fwd.hip's chained kernels have
template <typename T, bool ACC, bool STATS>
__global__ void __launch_bounds__(256) conv3d_k3_cube_reduce_kernel(CubeReduceDev P) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P.n8) return;
    const int64_t vol = (int64_t)P.D * P.H * P.W, e = i * 8;
    const int64_t bc = e / vol, v = e - bc * vol;
    const int bi = (int)(bc / P.Cout), co = (int)(bc - (int64_t)bi * P.Cout);
    const int z = (int)(v / ((int64_t)P.H * P.W)), yx = (int)(v - (int64_t)z * P.H * P.W), yy = yx / P.W, xx = yx - yy * P.W;
    const int64_t total = (int64_t)P.B * P.Cout * vol;
    float sum[8];
    const float bv = P.bias ? P.bias[co] : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum[k] = 0.f;
    for (int s = 0; s < P.S; ++s) {
        const mfma_f32x4 lo = *reinterpret_cast<const mfma_f32x4*>(P.part + s * total + e);
        const mfma_f32x4 hi = *reinterpret_cast<const mfma_f32x4*>(P.part + s * total + e + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { sum[k] += lo[k]; sum[4 + k] += hi[k]; }
    }
    T* dst = reinterpret_cast<T*>(P.y) + (int64_t)bi * P.y_sb + (int64_t)co * P.y_sc + (int64_t)z * P.y_sz + (int64_t)yy * P.y_sy + xx;
    T out[8];
    if (ACC) {
        const cube_u32x4 old = *reinterpret_cast<const cube_u32x4*>(dst);
        memcpy(out, &old, 16);
#pragma unroll
        for (int k = 0; k < 8; ++k) sum[k] += to_f32(out[k]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { sum[k] += bv; out[k] = from_f32<T>(sum[k]); }
    if (STATS) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { s1 += sum[k]; s2 = fmaf(sum[k], sum[k], s2); }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
        if ((threadIdx.x & 63) == 0)
            reinterpret_cast<float4*>(P.stats)[bc * (vol / 512) + v / 512] = float4{512.f, s1, s2, 0.f};
    }
    cube_u32x4 packed;
    memcpy(&packed, out, 16);
    *reinterpret_cast<cube_u32x4*>(dst) = packed;
}

// ---- weight gradient ---------------------------------------------------------------------------------------------------------
// dW[co][ci][kz][ky][kx] = sum over (b, z, y, x) of dY[b, co, z, y, x] * X[b, ci, z + kz - 1, y + ky - 1, x + kx - 1] - per tap a GEMM whose
// contraction runs over VOXELS, and NCDHW has x contiguous: an operand fragment (8 consecutive k per lane) is one x octet of a
// row, for both operands - no transposition anywhere.  A workgroup (8 waves) owns 64 output x 32 input channels for ALL 27 taps
// and walks a range of 8 x 8 x 8 cubes: per cube the dY cube (64 channels x 64 rows of 16 bytes) and the X halo cube (32 channels x
// 10 x 10 rows) are staged in LDS in their native row layout (channel pitches 1 056 / 1 632 bytes: conflict-free ds_read_b128 by
// the tools/lds_conflicts.py model; + the two x neighbours of every dY row), the next cube's rows already in registers.  Wave
// (cot, cit) owns one 16 x 16 (co, ci) tile = 27 accumulator tiles; a k-step is four y-consecutive rows (lane group g = row): ONE dY
// fragment, shifted one element left / right in registers for the kx = 2 / 0 taps (v_alignbyte with the neighbour elements), and
// nine X fragments - rows (z + kz, y + ky) of the halo cube as they lie -, 27 MFMAs.  Partial sums per cube range in
// [split][tap][co][ci] (64-byte segments per store); a second launch adds them in a fixed order into the weight's (co, ci, 27)
// layout.  (Round 6 first tried both operands straight from global memory - correct, but ten scattered 1-KB loads per 27 MFMAs and
// wave are more than a CU's load path delivers, as round 5's direct variant of the row kernel had already shown.)
// The row kernels of csrc/conv3d_wgrad.hip want rows of >= 16 voxels and 48-channel blocks: 260 - 370 TF/s at 16^3; the vendor
// route ran the 8^3 layers at 90 - 200 TF/s.
constexpr int kWgXP = 1632, kWgDP = 1056, kWgHP = 260;          // LDS bytes per channel: X rows, dY rows, dY row neighbours
constexpr int kWgLdsX = 32 * kWgXP, kWgLdsD = 64 * kWgDP, kWgLdsH = 64 * kWgHP;
constexpr int kWgXIters = 7, kWgDIters = 8;               // 16 x 7 >= 100 halo rows per channel; 8 planes of the dY cube

struct CubeWgDev {
    const char* x;   int64_t x_sb, x_sc, x_sz, x_sy;
    const char* dy;  int64_t d_sb, d_sc, d_sz, d_sy;
    float* part;                                         // [S][27][Cout][Cin]
    int32_t B, Cin, Cout, D, H, W;
    int32_t ncib, ntasks, S, ncubes, tz, ty, tx;         // 32-channel input blocks; (64 co, 32 ci) blocks; splits; cubes over the batch
};

struct CubeWgStage {
    cube_u32x4 xr[kWgXIters];
    cube_u32x4 dr[kWgDIters];  uint32_t dh[kWgDIters];
};

template <typename T, bool HALO>
__global__ void __launch_bounds__(512, 1) conv3d_k3_cube_wgrad_kernel(CubeWgDev P) {
    typedef typename Mfma16<T>::v8 frag8;
    __shared__ __attribute__((aligned(16))) unsigned char s_x[kWgLdsX];
    __shared__ __attribute__((aligned(16))) unsigned char s_d[kWgLdsD];
    __shared__ __attribute__((aligned(16))) unsigned char s_h[HALO ? kWgLdsH : 16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cot = wave & 3, cit = wave >> 2;
    const int item = xcd_item(blockIdx.x, gridDim.x);
    const int s = item / P.ntasks, t2 = item - s * P.ntasks;       // the blocks of one split next to each other: they read the same cubes
    const int cob = t2 / P.ncib, cib = t2 - cob * P.ncib;
    const int c0 = (int)((int64_t)s * P.ncubes / P.S), c1 = (int)((int64_t)(s + 1) * P.ncubes / P.S);
    const int n16 = lane & 15, g = lane >> 4;
    const int per_b = P.tz * P.ty * P.tx;

    const uint16_t* xg = reinterpret_cast<const uint16_t*>(P.x) + (int64_t)(cib * 32) * P.x_sc;
    const uint16_t* dg = reinterpret_cast<const uint16_t*>(P.dy) + (int64_t)(cob * 64) * P.d_sc;
    CubeWgStage st;
    // staging map (index arithmetic kept trivial - a general (channel, row) = divmod(q, 100) map cost 500 instructions per thread and cube):
    // X: thread -> channel tid / 16, rows tid % 16 + 16 it of the 100 halo rows;  dY: thread -> channel tid / 8, row y = tid % 8 of plane it
    const int x_ci = tid >> 4, x_r0 = tid & 15, d_co = tid >> 3, d_y = tid & 7;
    const uint16_t* xt = xg + (int64_t)x_ci * P.x_sc;
    const uint16_t* dt = dg + (int64_t)d_co * P.d_sc + (int64_t)d_y * P.d_sy;
    auto load_cube = [&](int c) {
        const int b = c / per_b, cc = c - b * per_b;
        const int z0 = 8 * (cc / (P.ty * P.tx)), y0 = 8 * ((cc / P.tx) % P.ty), x0 = 8 * (cc % P.tx);
        const uint16_t* xc = xt + (int64_t)b * P.x_sb + (int64_t)(z0 - 1) * P.x_sz + (int64_t)(y0 - 1) * P.x_sy + x0;
#pragma unroll
        for (int it = 0; it < kWgXIters; ++it) {
            const int row = x_r0 + 16 * it, zp = row / 10, yp = row - zp * 10;
            const int gz = z0 - 1 + zp, gy = y0 - 1 + yp;
            st.xr[it] = cube_u32x4{0, 0, 0, 0};
            if (row < 100 && gz >= 0 && gz < P.D && gy >= 0 && gy < P.H)
                st.xr[it] = *reinterpret_cast<const cube_u32x4*>(xc + (int64_t)zp * P.x_sz + (int64_t)yp * P.x_sy);
        }
        const uint16_t* dc = dt + (int64_t)b * P.d_sb + (int64_t)z0 * P.d_sz + (int64_t)y0 * P.d_sy + x0;
#pragma unroll
        for (int it = 0; it < kWgDIters; ++it) {
            const uint16_t* r = dc + (int64_t)it * P.d_sz;
            st.dr[it] = *reinterpret_cast<const cube_u32x4*>(r);
            if (HALO) st.dh[it] = (x0 > 0 ? (uint32_t)r[-1] : 0u) | (x0 + 8 < P.W ? (uint32_t)r[8] << 16 : 0u);
        }
    };
    auto store_cube = [&]() {
#pragma unroll
        for (int it = 0; it < kWgXIters; ++it) {
            const int row = x_r0 + 16 * it;
            if (row < 100) *reinterpret_cast<cube_u32x4*>(s_x + x_ci * kWgXP + row * 16) = st.xr[it];
        }
#pragma unroll
        for (int it = 0; it < kWgDIters; ++it) {
            const int row = it * 8 + d_y;
            *reinterpret_cast<cube_u32x4*>(s_d + d_co * kWgDP + row * 16) = st.dr[it];
            if (HALO) *reinterpret_cast<uint32_t*>(s_h + d_co * kWgHP + row * 4) = st.dh[it];
        }
    };

    mfma_f32x4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = mfma_f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* a_base = s_d + (cot * 16 + n16) * kWgDP + g * 16;
    const unsigned char* b_base = s_x + (cit * 16 + n16) * kWgXP + g * 16;
    const unsigned char* h_base = s_h + (HALO ? (cot * 16 + n16) * kWgHP + g * 4 : 0);

#ifndef SEGM_WG_NOPF
    if (c0 < c1) load_cube(c0);
#endif
    for (int c = c0; c < c1; ++c) {
#ifdef SEGM_WG_NOPF                                       /* experiment: no register prefetch of the next cube (67 registers less) */
        load_cube(c);
#endif
        __syncthreads();
        store_cube();
        __syncthreads();
#if !defined(SEGM_WG_NOPF) && !defined(SEGM_WG_ABL_NOLOAD)
        if (c + 1 < c1) load_cube(c + 1);
#endif
        // 16 k-steps (plane z = j / 2, rows y = 4 (j & 1) + g): the dY fragment and its two one-element shifts (the kx = 2 / 0 taps:
        // sum_x dY[x] X[x + kx - 1] = sum_x' dY[x' - kx + 1] X[x'] - shifting the ONE dY fragment of a k-step instead of its nine X
        // fragments: five funnel shifts per 27 MFMAs instead of 45), then nine X fragments - rows (z + kz, y + ky) of the halo cube,
        // read as they lie - with three MFMAs each.  X fragments are requested two reads ahead (ring of three), the dY fragment one
        // k-step ahead; the fences keep the compiler from hoisting more reads than the rings hold.
        cube_u32x4 ring_o[3];
        auto rd = [&](int j, int kk, int slot) {
            const int row = ((j >> 1) + kk / 3) * 10 + 4 * (j & 1) + kk % 3;
            ring_o[slot] = *reinterpret_cast<const cube_u32x4*>(b_base + row * 16);
        };
        cube_u32x4 a_next;
        uint32_t h_next = 0;
        auto rd_a = [&](int j) {
            const int row = (j >> 1) * 8 + 4 * (j & 1);
            a_next = *reinterpret_cast<const cube_u32x4*>(a_base + row * 16);
            if (HALO) h_next = *reinterpret_cast<const uint32_t*>(h_base + row * 4);
        };
        rd_a(0);
        rd(0, 0, 0);
        rd(0, 1, 1);
#ifdef SEGM_WG_ABL_NOMFMA
        for (int j = 0; j < 0; ++j) {
#else
#pragma unroll
        for (int j = 0; j < 16; ++j) {
#endif
            const cube_u32x4 o = a_next;
            const uint32_t lw = HALO ? h_next << 16 : 0u, rw = HALO ? h_next >> 16 : 0u;
            if (j + 1 < 16) rd_a(j + 1);
            const uint32_t s01 = __builtin_amdgcn_alignbyte(o[1], o[0], 2), s12 = __builtin_amdgcn_alignbyte(o[2], o[1], 2),
                           s23 = __builtin_amdgcn_alignbyte(o[3], o[2], 2);
            const cube_u32x4 l = cube_u32x4{__builtin_amdgcn_alignbyte(o[0], lw, 2), s01, s12, s23};       // element i = dY[i - 1]: kx = 2
            const cube_u32x4 r = cube_u32x4{s01, s12, s23, __builtin_amdgcn_alignbyte(rw, o[3], 2)};       // element i = dY[i + 1]: kx = 0
            const frag8 a0 = __builtin_bit_cast(frag8, r), a1 = __builtin_bit_cast(frag8, o), a2 = __builtin_bit_cast(frag8, l);
#pragma unroll
            for (int kk = 0; kk < 9; ++kk) {
                const int q = j * 9 + kk;
                if (q + 2 < 144) rd((q + 2) / 9, (q + 2) % 9, (q + 2) % 3);
                const frag8 bf = __builtin_bit_cast(frag8, ring_o[q % 3]);
                acc[3 * kk] = Mfma16<T>::run(a0, bf, acc[3 * kk]);
                acc[3 * kk + 1] = Mfma16<T>::run(a1, bf, acc[3 * kk + 1]);
                acc[3 * kk + 2] = Mfma16<T>::run(a2, bf, acc[3 * kk + 2]);
#ifndef SEGM_WG_NOFENCE
                SEGM_SCHED_FENCE();
#endif
            }
        }
    }
    // D[row = co 4 g + i][col = ci n16]
    float* pp = P.part + (int64_t)s * 27 * P.Cout * P.Cin + (int64_t)(cob * 64 + cot * 16 + 4 * g) * P.Cin + cib * 32 + cit * 16 + n16;
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) pp[(int64_t)t * P.Cout * P.Cin + (int64_t)i * P.Cin] = acc[t][i];
}

// partial sums [S][27][n] (n = Cout * Cin pairs) -> dw[n][27]: a block sums 256 pairs tap by tap (coalesced along the pairs), turns the
// (tap, pair) tile through LDS (pitch 27 dwords: odd, conflict-free) and writes 256 * 27 contiguous values.  (A thread per pair writing its
// 27 values directly put 4-byte pieces 108 bytes apart: the reduction then took 4 - 10 times as long as the MFMA launch.)
template <typename OUT>
__global__ void __launch_bounds__(256) conv3d_k3_cube_wgrad_reduce_kernel(const float* __restrict__ part, OUT* __restrict__ dw, int S, int64_t n) {
    __shared__ float s_t[256 * 27];
    const int64_t i0 = (int64_t)blockIdx.x * 256;
    const int tid = threadIdx.x;
    const int cnt = n - i0 < 256 ? (int)(n - i0) : 256;
    if (tid < cnt) {
#pragma unroll 3
        for (int t = 0; t < 27; ++t) {
            float v = 0.f;
            for (int s = 0; s < S; ++s) v += part[((int64_t)s * 27 + t) * n + i0 + tid];
            s_t[tid * 27 + t] = v;
        }
    }
    __syncthreads();
    OUT* o = dw + i0 * 27;
    for (int e = tid; e < cnt * 27; e += 256) o[e] = from_f32<OUT>(s_t[e]);
}

// ---- the weight images of a whole network in ONE launch (segm_conv3d_k3_cube_pack_multi) ---------------------------------------
// A training step changes every weight, so every fragment image is rebuilt once per step from the step's 16-bit weight copy.  As
// packs of the parameter bank's generic gather (an index per element, eight 2-byte gathers per 16-byte store) the 112 M image
// elements of the benchmarked network took 0.45 ms per step.  Here a block takes the (16 output x 32 input channel) tile of one
// image - 16 (forward) or 32 (data gradient) contiguous runs of the weight - through LDS with coalesced loads and writes its 27
// fragments with coalesced 16-byte stores; a descriptor table deals the blocks of all images to one grid.
struct CubePackDev { const uint16_t* src; uint16_t* out; const segm_cube_pack_desc* descs; int32_t ndesc; };

__global__ void __launch_bounds__(256) conv3d_k3_cube_pack_kernel(CubePackDev P) {
    __shared__ uint16_t s_t[16 * 866];                    // forward: [16 co][864 (+2)]; flipped: [32 co][432 (+1)] - odd dword pitches
    __shared__ int s_d;
    const int tid = threadIdx.x;
    if (tid == 0) {
        int d = 0;
        while (d + 1 < P.ndesc && P.descs[d + 1].first_block <= (int)blockIdx.x) ++d;
        s_d = d;
    }
    __syncthreads();
    const segm_cube_pack_desc D = P.descs[s_d];
    const int lb = (int)blockIdx.x - D.first_block;
    const uint16_t* src = P.src + D.src_off;
    const bool fl = D.flipped != 0;
    const int R = (fl ? D.cout_w : D.cin_w) / 32;          // rounds of the convolution the image is for
    const int ct = lb / R, r = lb - ct * R;
    const int rows = fl ? 32 : 16, run = fl ? 432 : 864, pitch = fl ? 433 : 866;
    // row i of the tile: forward: w[ct 16 + i][r 32 .. + 31][27]; flipped: w[r 32 + i][ct 16 .. + 15][27]
    const int64_t base = fl ? (int64_t)(r * 32) * D.co_stride + (int64_t)ct * 16 * 27 : (int64_t)(ct * 16) * D.co_stride + (int64_t)r * 32 * 27;
    for (int e = tid; e < rows * run; e += 256) {
        const int i = e / run, j = e - i * run;
        s_t[i * pitch + j] = src[base + (int64_t)i * D.co_stride + j];
    }
    __syncthreads();
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4* out = reinterpret_cast<u32x4*>(P.out + D.out_off) + (int64_t)lb * 27 * 64;
    for (int gI = tid; gI < 27 * 64; gI += 256) {
        const int tap = gI >> 6, lane = gI & 63, n = lane & 15, kg = lane >> 4;
        uint32_t v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = fl ? s_t[(kg * 8 + e) * pitch + n * 27 + (26 - tap)] : s_t[n * pitch + (kg * 8 + e) * 27 + tap];
        out[gI] = u32x4{v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)};
    }
}

static bool cube_direct() {                                // one split: the main launch finishes the values itself (SEGM_CUBE_DIRECT=0: always two launches)
    static const bool v = [] { const char* e = getenv("SEGM_CUBE_DIRECT"); return !(e && e[0] == '0'); }();
    return v;
}

static int cube_nt(int cout, int forced) {                 // column tiles per wave the layer can use: 4 (NB = 128), 3 (96), 2 (64)
    if (forced >= 2 && forced <= 4 && cout % (32 * forced) == 0) return forced;
    return 0;
}

}  // namespace segm

using namespace segm;

// The launch plan of a layer: NT column tiles per wave (NB = 32 NT channels per workgroup) and the number of splits of the
// contraction, chosen so that the grid fills the 256 CUs with as few partial sums as that takes; workspace = fp32 elements of the
// partial sums.  nt / splits > 0 on entry are kept if the layer can use them (experiments, tests).
extern "C" int segm_conv3d_k3_cube_plan(int32_t batch, int32_t cin, int32_t cout, int32_t depth, int32_t height, int32_t width,
                                        int32_t* nt, int32_t* splits, int64_t* workspace_elems) {
    if (!nt || !splits || !workspace_elems) return SEGM_E_NULL;
    if (batch <= 0 || cin <= 0 || cout <= 0 || depth <= 0 || height <= 0 || width <= 0) return SEGM_E_SHAPE;
    if (cin % 32 != 0 || (cout % 64 != 0 && cout % 96 != 0) || depth % 8 != 0 || height % 8 != 0 || width % 8 != 0) return SEGM_E_SHAPE;
    const int R = cin / 32;
    const int64_t cubes = (int64_t)(depth / 8) * (height / 8) * (width / 8) * batch;
    static const int env_nt = [] { const char* e = getenv("SEGM_CUBE_NT"); return e ? atoi(e) : 0; }();
    static const int env_s = [] { const char* e = getenv("SEGM_CUBE_SPLITS"); return e ? atoi(e) : 0; }();
    static const int target = [] { const char* e = getenv("SEGM_CUBE_WORKGROUPS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();
    if ((*nt > 0 && !cube_nt(cout, *nt)) || *nt < 0 || *splits < 0 || *splits > R) return SEGM_E_SHAPE;     // an explicit request the layer cannot run
    int want_nt = *nt > 0 ? *nt : env_nt, want_s = *splits > 0 ? *splits : env_s;
    // every (column tiles, divisor of the rounds) by a cost model fitted to profiles/r06_conv_cube_v2.txt: a round is 6 912 MFMA cycles per
    // column tile + ~3 000 of staging and barriers (the 128-channel variant has no room to pipeline its fragment reads: + 5 000), a
    // workgroup ~12 000 of prologue and partial-sum stores; more than `target` workgroups run as a second wave of workgroups
    // behind the first; every split costs a pass over the output's partial sums in the reduction
    int best_nt = 0, best_s = 0;
    double best_cost = 0;
    const double red = 0.0028 * (double)batch * cout * depth * height * width;
    for (int c = 4; c >= 2; --c) {
        if (cout % (32 * c) != 0 || (want_nt > 0 && cube_nt(cout, want_nt) && c != want_nt)) continue;
        const int64_t base = cubes * (cout / (32 * c));
        for (int sp = 1; sp <= R; ++sp) {
            if (R % sp != 0 || (want_s > 0 && want_s <= R && sp != want_s && R % want_s == 0)) continue;
            const int64_t wg = base * sp, waves = (wg + target - 1) / target;
            const double cost = (double)waves * ((double)(R / sp) * (6912.0 * c + (c == 4 ? 8000.0 : 3000.0)) + 12000.0) + (sp == 1 && cube_direct() ? 0.0 : red * sp);
            if (best_nt == 0 || cost < best_cost) { best_nt = c; best_s = sp; best_cost = cost; }
        }
    }
    if (best_nt == 0) return SEGM_E_SHAPE;
    if (want_s > 0 && want_s <= R && R % want_s != 0) best_s = want_s;      // an uneven split on request (the kernel deals the rounds floor / ceil)
    *nt = best_nt; *splits = best_s;
    // one split: the main launch stores the finished values itself - no partial sums
    *workspace_elems = best_s == 1 && cube_direct() ? 0 : (int64_t)best_s * batch * cout * depth * height * width;
    return SEGM_OK;
}

// Index map of the weight image: out[i] = flat index into w (cout_w, cin_w, 3, 3, 3) contiguous for i over
// [Cout / 16][Cin / 32][27][64][8] of the convolution the image is FOR: flipped = 0: the forward convolution (Cout = cout_w,
// Cin = cin_w); flipped = 1: its data gradient as a convolution of dy (Cout = cin_w, Cin = cout_w, taps mirrored).
extern "C" int segm_conv3d_k3_cube_pack_index(int32_t* out, int64_t n, int32_t cout_w, int32_t cin_w, int32_t flipped) {
    if (!out) return SEGM_E_NULL;
    const int Cout = flipped ? cin_w : cout_w, Cin = flipped ? cout_w : cin_w;
    if (cout_w <= 0 || cin_w <= 0 || Cout % 16 != 0 || Cin % 32 != 0) return SEGM_E_SHAPE;
    if (n != (int64_t)Cout * Cin * 27 || n >= ((int64_t)1 << 31)) return SEGM_E_SHAPE;
    const int R = Cin / 32;
    int64_t o = 0;
    for (int ct = 0; ct < Cout / 16; ++ct)
        for (int r = 0; r < R; ++r)
            for (int tap = 0; tap < 27; ++tap)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e) {
                        const int co = ct * 16 + (l & 15), ci = r * 32 + (l >> 4) * 8 + e;
                        out[o++] = flipped ? (int32_t)(((int64_t)ci * cin_w + co) * 27 + (26 - tap))
                                           : (int32_t)(((int64_t)co * cin_w + ci) * 27 + tap);
                    }
    return SEGM_OK;
}

// partials per (batch, channel) instance a launch with `splits` splits writes: per wave of the launch that stores the finished values -
// the main launch (128 voxels per wave) with one split, the reduction launch (512) otherwise
extern "C" int32_t segm_conv3d_k3_cube_stats_parts(int32_t depth, int32_t height, int32_t width, int32_t splits) {
    const int64_t vol = (int64_t)depth * height * width;
    return (int32_t)(splits == 1 && cube_direct() ? vol / 128 : vol / 512);
}

extern "C" int segm_conv3d_k3_cube_fwd(const segm_conv3d_cube_args* a) {
    if (!a) return SEGM_E_NULL;
    if (!a->x || !a->y || !a->w_image) return SEGM_E_NULL;
    if (a->dtype != SEGM_BF16 && a->dtype != SEGM_F16) return SEGM_E_DTYPE;
    if (a->flags & ~SEGM_CONV_CUBE_ACCUMULATE) return SEGM_E_SHAPE;
    int32_t nt = a->nt, splits = a->splits;
    int64_t need = 0;
    const int rc = segm_conv3d_k3_cube_plan(a->batch, a->cin, a->cout, a->depth, a->height, a->width, &nt, &splits, &need);
    if (rc != SEGM_OK) return rc;
    if ((a->nt > 0 && nt != a->nt) || (a->splits > 0 && splits != a->splits)) return SEGM_E_SHAPE;
    const bool direct = splits == 1 && cube_direct();
    if (need > 0 && (!a->workspace || a->workspace_elems < need)) return SEGM_E_WORKSPACE;
    if (a->stats_partials && a->stats_nparts != segm_conv3d_k3_cube_stats_parts(a->depth, a->height, a->width, splits)) return SEGM_E_WORKSPACE;
    const int64_t st[8] = {a->x_stride_b, a->x_stride_c, a->x_stride_z, a->x_stride_y, a->y_stride_b, a->y_stride_c, a->y_stride_z, a->y_stride_y};
    for (int64_t s : st)
        if (s % 8 != 0 || s <= 0) return SEGM_E_SHAPE;      // 16-byte aligned rows
    if (((uintptr_t)a->x & 15) || ((uintptr_t)a->y & 15) || ((uintptr_t)a->w_image & 15) || (need > 0 && ((uintptr_t)a->workspace & 15))) return SEGM_E_SHAPE;

    CubeDev P;
    memset(&P, 0, sizeof(P));
    P.x = (const char*)a->x; P.x_sb = a->x_stride_b; P.x_sc = a->x_stride_c; P.x_sz = a->x_stride_z; P.x_sy = a->x_stride_y;
    P.wimg = (const char*)a->w_image; P.part = (float*)a->workspace;
    P.B = a->batch; P.Cin = a->cin; P.Cout = a->cout; P.D = a->depth; P.H = a->height; P.W = a->width;
    P.R = a->cin / 32; P.S = splits; P.tz = a->depth / 8; P.ty = a->height / 8; P.tx = a->width / 8; P.ncb = a->cout / (32 * nt);
    const int64_t nwg = (int64_t)P.tz * P.ty * P.tx * P.B * P.ncb * P.S;
    if (nwg >= ((int64_t)1 << 31)) return SEGM_E_SHAPE;
    hipStream_t stream = (hipStream_t)a->stream;
    const bool f16 = a->dtype == SEGM_F16;
    const bool acc = (a->flags & SEGM_CONV_CUBE_ACCUMULATE) != 0;
    if (direct) {
        P.y = (char*)a->y; P.y_sb = a->y_stride_b; P.y_sc = a->y_stride_c; P.y_sz = a->y_stride_z; P.y_sy = a->y_stride_y;
        P.bias = a->bias; P.stats = a->stats_partials;
    }
    const int epi = direct ? (acc ? 2 : 1) : 0;
#define SEGM_CUBE_LAUNCH2(T, NT_)                                                                                                                  \
    do {                                                                                                                                            \
        if (epi == 0) hipLaunchKernelGGL((conv3d_k3_cube_kernel<T, NT_, 0>), dim3((unsigned)nwg), dim3(512), 0, stream, P);                          \
        else if (epi == 1) hipLaunchKernelGGL((conv3d_k3_cube_kernel<T, NT_, 1>), dim3((unsigned)nwg), dim3(512), 0, stream, P);                     \
        else hipLaunchKernelGGL((conv3d_k3_cube_kernel<T, NT_, 2>), dim3((unsigned)nwg), dim3(512), 0, stream, P);                                   \
    } while (0)
#define SEGM_CUBE_LAUNCH(NT_) do { if (f16) SEGM_CUBE_LAUNCH2(f16_t, NT_); else SEGM_CUBE_LAUNCH2(bf16_t, NT_); } while (0)
    if (nt == 4) SEGM_CUBE_LAUNCH(4); else if (nt == 3) SEGM_CUBE_LAUNCH(3); else SEGM_CUBE_LAUNCH(2);
#undef SEGM_CUBE_LAUNCH
#undef SEGM_CUBE_LAUNCH2
    int err = (int)hipGetLastError();
    if (err || direct) return err;

    CubeReduceDev Q;
    memset(&Q, 0, sizeof(Q));
    Q.part = (const float*)a->workspace; Q.bias = a->bias;
    Q.y = (char*)a->y; Q.y_sb = a->y_stride_b; Q.y_sc = a->y_stride_c; Q.y_sz = a->y_stride_z; Q.y_sy = a->y_stride_y;
    Q.S = splits; Q.B = a->batch; Q.Cout = a->cout; Q.D = a->depth; Q.H = a->height; Q.W = a->width;
    Q.n8 = (int64_t)a->batch * a->cout * a->depth * a->height * a->width / 8;
    const unsigned nb = (unsigned)((Q.n8 + 255) / 256);
    Q.stats = a->stats_partials;
#define SEGM_CUBE_RED(T, A_)                                                                                                     \
    do {                                                                                                                          \
        if (Q.stats) hipLaunchKernelGGL((conv3d_k3_cube_reduce_kernel<T, A_, true>), dim3(nb), dim3(256), 0, stream, Q);           \
        else hipLaunchKernelGGL((conv3d_k3_cube_reduce_kernel<T, A_, false>), dim3(nb), dim3(256), 0, stream, Q);                  \
    } while (0)
    if (f16) { if (acc) SEGM_CUBE_RED(f16_t, true); else SEGM_CUBE_RED(f16_t, false); }
    else { if (acc) SEGM_CUBE_RED(bf16_t, true); else SEGM_CUBE_RED(bf16_t, false); }
#undef SEGM_CUBE_RED
    return (int)hipGetLastError();
}

static void cube_wgrad_plan(int batch, int cin, int cout, int depth, int height, int width, int* splits, int* ncubes) {
    const int64_t cubes = (int64_t)batch * (depth / 8) * (height / 8) * (width / 8);
    const int64_t tasks = (int64_t)(cout / 64) * (cin / 32);
    static const int target = [] { const char* e = getenv("SEGM_CUBE_WGRAD_WORKGROUPS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();
    // as many splits as fill the CUs once, never more than cubes; with more tasks than CUs no split
    int64_t sp = target / tasks;
    if (sp > cubes) sp = cubes;
    if (sp < 1) sp = 1;
    *splits = (int)sp;
    *ncubes = (int)cubes;
}

static bool cube_wgrad_shape_ok(int batch, int cin, int cout, int depth, int height, int width) {
    return batch > 0 && cin > 0 && cout > 0 && depth > 0 && height > 0 && width > 0 && cin % 32 == 0 && cout % 64 == 0 && depth % 8 == 0 &&
           height % 8 == 0 && width % 8 == 0 && (int64_t)batch * (depth / 8) * (height / 8) * (width / 8) < ((int64_t)1 << 24) &&
           (int64_t)cin * cout * 27 < ((int64_t)1 << 31);
}

extern "C" size_t segm_conv3d_k3_cube_wgrad_workspace_bytes(int32_t batch, int32_t cin, int32_t cout, int32_t depth, int32_t height, int32_t width) {
    if (!cube_wgrad_shape_ok(batch, cin, cout, depth, height, width)) return 0;
    int sp, nc;
    cube_wgrad_plan(batch, cin, cout, depth, height, width, &sp, &nc);
    return (size_t)sp * 27 * cout * cin * sizeof(float);
}

// the weight gradient of the same layers (same argument block as segm_conv3d_k3_wgrad): cin % 32 == 0, cout % 64 == 0, depth /
// height / width multiples of 8
extern "C" int segm_conv3d_k3_cube_wgrad(const segm_conv3d_wgrad_args* a) {
    if (!a) return SEGM_E_NULL;
    if (!a->x || !a->dy || !a->dw) return SEGM_E_NULL;
    if (a->dtype != SEGM_BF16 && a->dtype != SEGM_F16) return SEGM_E_DTYPE;
    if (a->dw_dtype != SEGM_BF16 && a->dw_dtype != SEGM_F16 && a->dw_dtype != SEGM_F32) return SEGM_E_DTYPE;
    if (!cube_wgrad_shape_ok(a->batch, a->cin, a->cout, a->depth, a->height, a->width)) return SEGM_E_SHAPE;
    const int64_t st[8] = {a->x_stride_b, a->x_stride_c, a->x_stride_z, a->x_stride_y, a->dy_stride_b, a->dy_stride_c, a->dy_stride_z, a->dy_stride_y};
    for (int64_t v : st)
        if (v % 8 != 0 || v <= 0) return SEGM_E_SHAPE;
    if (((uintptr_t)a->x & 15) || ((uintptr_t)a->dy & 15) || ((uintptr_t)a->workspace & 15)) return SEGM_E_SHAPE;
    int sp, nc;
    cube_wgrad_plan(a->batch, a->cin, a->cout, a->depth, a->height, a->width, &sp, &nc);
    const size_t need = (size_t)sp * 27 * a->cout * a->cin * sizeof(float);
    if (!a->workspace || a->workspace_bytes < need) return SEGM_E_WORKSPACE;
    CubeWgDev P;
    memset(&P, 0, sizeof(P));
    P.x = (const char*)a->x; P.x_sb = a->x_stride_b; P.x_sc = a->x_stride_c; P.x_sz = a->x_stride_z; P.x_sy = a->x_stride_y;
    P.dy = (const char*)a->dy; P.d_sb = a->dy_stride_b; P.d_sc = a->dy_stride_c; P.d_sz = a->dy_stride_z; P.d_sy = a->dy_stride_y;
    P.part = (float*)a->workspace;
    P.B = a->batch; P.Cin = a->cin; P.Cout = a->cout; P.D = a->depth; P.H = a->height; P.W = a->width;
    P.ncib = a->cin / 32; P.ntasks = (a->cout / 64) * P.ncib; P.S = sp; P.ncubes = nc;
    P.tz = a->depth / 8; P.ty = a->height / 8; P.tx = a->width / 8;
    hipStream_t stream = (hipStream_t)a->stream;
    const unsigned nwg = (unsigned)((int64_t)P.ntasks * sp);
    const bool halo = a->width > 8, f16 = a->dtype == SEGM_F16;
    if (f16) {
        if (halo) hipLaunchKernelGGL((conv3d_k3_cube_wgrad_kernel<f16_t, true>), dim3(nwg), dim3(512), 0, stream, P);
        else hipLaunchKernelGGL((conv3d_k3_cube_wgrad_kernel<f16_t, false>), dim3(nwg), dim3(512), 0, stream, P);
    } else {
        if (halo) hipLaunchKernelGGL((conv3d_k3_cube_wgrad_kernel<bf16_t, true>), dim3(nwg), dim3(512), 0, stream, P);
        else hipLaunchKernelGGL((conv3d_k3_cube_wgrad_kernel<bf16_t, false>), dim3(nwg), dim3(512), 0, stream, P);
    }
    int err = (int)hipGetLastError();
    if (err) return err;
    const int64_t n = (int64_t)a->cout * a->cin;
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (a->dw_dtype == SEGM_F32) hipLaunchKernelGGL((conv3d_k3_cube_wgrad_reduce_kernel<float>), dim3(nb), dim3(256), 0, stream, (const float*)a->workspace, (float*)a->dw, sp, n);
    else if (a->dw_dtype == SEGM_F16) hipLaunchKernelGGL((conv3d_k3_cube_wgrad_reduce_kernel<f16_t>), dim3(nb), dim3(256), 0, stream, (const float*)a->workspace, (f16_t*)a->dw, sp, n);
    else hipLaunchKernelGGL((conv3d_k3_cube_wgrad_reduce_kernel<bf16_t>), dim3(nb), dim3(256), 0, stream, (const float*)a->workspace, (bf16_t*)a->dw, sp, n);
    return (int)hipGetLastError();
}

// descs: DEVICE array of ndesc descriptors with ascending first_block; nblocks = sum over the images of (Cout / 16) * (Cin / 32) of the
// convolution each image is for.  src / out: the 16-bit weight buffer and the image buffer the descriptors' element offsets refer to.
extern "C" int segm_conv3d_k3_cube_pack_multi(const void* src, void* out, const segm_cube_pack_desc* descs, int32_t ndesc, int32_t nblocks, void* stream) {
    if (ndesc == 0 || nblocks == 0) return SEGM_OK;
    if (!src || !out || !descs) return SEGM_E_NULL;
    if (ndesc < 0 || nblocks < 0 || ((uintptr_t)out & 15)) return SEGM_E_SHAPE;
    CubePackDev P{(const uint16_t*)src, (uint16_t*)out, descs, ndesc};
    hipLaunchKernelGGL(conv3d_k3_cube_pack_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, P);
    return (int)hipGetLastError();
}
