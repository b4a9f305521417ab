// The stem convolution of the encoder: Conv3d(kernel 7, stride 2, padding 3) on a few input channels   (C ABI: segm_stem_conv_fwd)
//
// model_segmamba/segmamba.py:141 (`nn.Conv3d(in_chans, dims[0], kernel_size=7, stride=2, padding=3)`), cuDNN in the reference.
// MIOpen runs it as im2col + GEMM: a 1.4 GB column matrix for a 2 x 4 x 128^3 input, 1.37 ms forward (profiles/r02_torch_prof.log)
// for 69 GFLOP and 84 MB of real traffic.  Here it is an implicit GEMM on MFMA whose K axis is laid out so that every operand
// fragment is one contiguous piece of memory:
//   * the input is channel-last with 4 channels, x4[b][z][y][x][4] (one transposing copy of the small input, made by the
//     host); K = (kz, ky, kx slot 0..7, ci 0..3): one K = 32 chunk per (kz, ky) pair = 49 MFMA steps (slot 7 has zero weights);
//   * A fragment of output voxel (z, y, x) for kx slots 2 g, 2 g + 1 = the 4 channels of input positions 2 x + 2 g - 3 and
//     2 x + 2 g - 2 on row (2 z + kz - 3, 2 y + ky - 3): two 8-byte raw buffer loads whose range check supplies the zero padding
//     (round 5: uniform row descriptor + fixed lane offsets, see the kernel);
//   * weights packed as wp[co][kz][ky][slot][ci]: a B fragment is 16 contiguous bytes; a wave keeps the three 16-channel
//     fragments of a (kz, ky) step for EIGHT 16-voxel tiles (96 accumulator registers), so weight traffic is 1/8 of the
//     activation traffic and both come from L1 / L2;
//   * D[voxel][co]: a lane holds four consecutive x of one output channel; x-adjacent tiles exchange halves (v_permlane16_swap_b32)
//     and store 16 bytes per lane; bias in the accumulators.
// v_mfma_f32_16x16x32: A[i][k]: lane l holds A[i = l & 15][8 (l >> 4) .. +7]; B[k][j]: lane l holds B[8 (l >> 4) .. +7][j = l & 15];
// D[row = 4 (l >> 4) + r][col = l & 15].  Here i = output voxel (16 consecutive x), k = (kx slot, ci), j = output channel.
#include <stdlib.h>
#include <string.h>

#include "segm_device.h"

namespace segm {

typedef float st_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t st_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t st_u32x2 __attribute__((ext_vector_type(2)));

constexpr int kStemWaves = 4;
constexpr int kStemTiles = 8;                // 16-voxel tiles per wave
constexpr int kStemK = 7;                    // the stem proper; the same kernels run the 3^3 stride-1 first layer (KSZ = 3, STR = 1)

struct StemDev {
    const char* x4;                          // (B, Din, Hin, Win, 4) contiguous
    const char* wp;                          // (cout, 7, 7, 8, 4) contiguous
    const float* bias;
    char* y;                                 // (B, cout, Dout, Hout, Wout), dense inside a channel
    int64_t y_sc;                            // elements between channels (batch stride = cout * y_sc)
    int32_t batch, cout;
    int32_t din, hin, win, dout, hout, wout;
    int64_t blocks;                          // wave blocks: batch * dout * (hout / TY) * (wout / 16 / TX), TX * TY = kStemTiles
};

// A wave owns a block of TX x TY = 8 tiles: TX 16-voxel tiles along x on each of TY consecutive output rows (TX = min(8, tiles
// per row)), so its tiles differ only by compile-time offsets from one (batch, z, y0, x0).
template <typename T, int NT, int TX, int KSZ, int STR, bool WIDE = false>     // KSZ^3 taps, stride STR, padding KSZ / 2
__global__ void __launch_bounds__(kStemWaves * 64, 2) stem_conv_fwd_kernel(StemDev P) {
    typedef typename Mfma16<T>::v8 frag8;
    constexpr int TY = kStemTiles / TX;
    constexpr int PAD = KSZ / 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    int64_t id = (int64_t)blockIdx.x * kStemWaves + wave;
    if (id >= P.blocks) return;
    const int xb = P.wout / (16 * TX), yb = P.hout / TY;
    const int x0 = (int)(id % xb) * 16 * TX;
    id /= xb;
    const int y0 = (int)(id % yb) * TY;
    id /= yb;
    const int z0 = (int)(id % P.dout);
    const int b0 = (int)(id / P.dout);
    st_f32x4 acc[kStemTiles][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float b = (P.bias && 16 * nt + i16 < P.cout) ? P.bias[16 * nt + i16] : 0.f;
#pragma unroll
        for (int t = 0; t < kStemTiles; ++t) acc[t][nt] = st_f32x4{b, b, b, b};
    }
    const T* X = reinterpret_cast<const T*>(P.x4);
    const T* W = reinterpret_cast<const T*>(P.wp);
    const int64_t row_el = (int64_t)P.win * 4, plane_el = (int64_t)P.hin * row_el, vol_el = (int64_t)P.din * plane_el;

    // One (kz, ky) step = the three weight fragments + the eight tiles' input fragments.  Round 5, two changes:
    //  * the input fragments are RAW BUFFER loads: a tile row's descriptor (uniform: base of input row (iz, iy), num_records = the
    //    row's bytes, or 0 for a padding row) + a lane offset that is fixed for the whole kernel (the lane's two input columns of the
    //    tile; a column outside the row or a kx slot >= KSZ carries an offset beyond num_records).  The range check returns the
    //    zeros: no per-step address arithmetic, masks or selects in vector registers - rounds 2 - 4 spent ~25 vector instructions per
    //    tile and step on them, 1 800 per wave against 216 MFMAs (the 3^3 first layer ran 0.40 ms for 0.44 GB at 128^3);
    //  * the NEXT step's loads are issued before this step's MFMAs (two register sets, alternating).
    typedef __amdgpu_buffer_rsrc_t rsrc_t;
    constexpr uint32_t kOob = 0xFFFFF000u;
    uint32_t vo0[kStemTiles], vo1[kStemTiles];
#pragma unroll
    for (int t = 0; t < kStemTiles; ++t) {
        const int ix = STR * (x0 + 16 * (t % TX) + i16) + 2 * g - PAD;               // first of the lane's two input columns
        vo0[t] = (ix >= 0 && ix < P.win && 2 * g < KSZ) ? (uint32_t)ix * 8u : kOob;
        vo1[t] = (ix + 1 >= 0 && ix + 1 < P.win && 2 * g + 1 < KSZ) ? (uint32_t)(ix + 1) * 8u : kOob;
    }
    const T* wlane[NT];                                                               // this lane's weight fragment of step 0
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = 16 * nt + i16;
        wlane[nt] = W + (int64_t)(co < P.cout ? co : 0) * KSZ * KSZ * 32 + 8 * g;
    }
    st_u32x4 wfa[NT], wfb[NT];
    st_u32x4 ava[kStemTiles], avb[kStemTiles];
    auto load_step = [&](int it, st_u32x4 (&wr)[NT], st_u32x4 (&av)[kStemTiles]) {
        const int kz = it / KSZ, ky = it - kz * KSZ;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wr[nt] = *reinterpret_cast<const st_u32x4*>(wlane[nt] + it * 32);
        const int iz = STR * z0 + kz - PAD;                                          // wave-uniform
        const bool z_ok = iz >= 0 && iz < P.din;
        const T* planep = X + (int64_t)b0 * vol_el + (int64_t)(z_ok ? iz : 0) * plane_el;
#pragma unroll
        for (int ty = 0; ty < TY; ++ty) {
            const int iy = STR * (y0 + ty) + ky - PAD;
            const bool row_ok = z_ok && iy >= 0 && iy < P.hin;
            const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(planep + (int64_t)(row_ok ? iy : 0) * row_el), 0,
                                                                row_ok ? P.win * 8 : 0, 0x00020000);
#pragma unroll
            for (int tx = 0; tx < TX; ++tx) {
                const int t = ty * TX + tx;
                const st_u32x2 v0 = __builtin_amdgcn_raw_buffer_load_b64(rs, vo0[t], 0, 0);
                const st_u32x2 v1 = __builtin_amdgcn_raw_buffer_load_b64(rs, vo1[t], 0, 0);
                av[t] = st_u32x4{v0[0], v0[1], v1[0], v1[1]};
            }
        }
    };
    auto mma_step = [&](const st_u32x4 (&wr)[NT], const st_u32x4 (&av)[kStemTiles]) {
        frag8 wf[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const st_u32x4 zero = {0u, 0u, 0u, 0u};
            wf[nt] = __builtin_bit_cast(frag8, 16 * nt + i16 < P.cout ? wr[nt] : zero);
        }
#pragma unroll
        for (int t = 0; t < kStemTiles; ++t) {
            const frag8 af = __builtin_bit_cast(frag8, av[t]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[t][nt] = Mfma16<T>::run(af, wf[nt], acc[t][nt]);
        }
    };
    constexpr int STEPS = KSZ * KSZ;
    load_step(0, wfa, ava);
#pragma unroll 1
    for (int it = 0; it < STEPS; it += 2) {
        if (it + 1 < STEPS) load_step(it + 1, wfb, avb);
        SEGM_SCHED_FENCE();
        mma_step(wfa, ava);
        SEGM_SCHED_FENCE();
        if (it + 2 < STEPS) load_step(it + 2, wfa, ava);
        SEGM_SCHED_FENCE();
        if (it + 1 < STEPS) mma_step(wfb, avb);
        SEGM_SCHED_FENCE();
    }
    T* Y = reinterpret_cast<T*>(P.y);
    const int64_t oplane = (int64_t)P.hout * P.wout, ovol = P.y_sc;
    if constexpr (WIDE) {
        // x-adjacent tiles (t, t + 1) of a row: v_permlane16_swap_b32 on the packed values leaves lane group g with EIGHT consecutive x
        // - 8 (g >> 1) .. + 7 of tile t + (g & 1) - one 16-byte store instead of two 8-byte ones (conv3d_fwd.hip has the derivation)
        static_assert(!WIDE || TX >= 2, "pairs are tiles along x");
#pragma unroll
        for (int t = 0; t < kStemTiles; t += 2) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = 16 * nt + i16;
                st_u32x2 p0, p1;
                p0[0] = pack2<T>(acc[t][nt][0], acc[t][nt][1]);         p0[1] = pack2<T>(acc[t][nt][2], acc[t][nt][3]);
                p1[0] = pack2<T>(acc[t + 1][nt][0], acc[t + 1][nt][1]); p1[1] = pack2<T>(acc[t + 1][nt][2], acc[t + 1][nt][3]);
                const st_u32x2 s0 = __builtin_amdgcn_permlane16_swap(p0[0], p1[0], false, false);
                const st_u32x2 s1 = __builtin_amdgcn_permlane16_swap(p0[1], p1[1], false, false);
                const st_u32x4 pk = {s0[0], s1[0], s0[1], s1[1]};
                if (co < P.cout) {
                    T* yp = Y + ((int64_t)b0 * P.cout + co) * ovol + (int64_t)z0 * oplane + (int64_t)(y0 + t / TX) * P.wout + x0 + 16 * (t % TX) +
                            16 * (g & 1) + 8 * (g >> 1);
                    *reinterpret_cast<st_u32x4*>(yp) = pk;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < kStemTiles; ++t) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = 16 * nt + i16;
            if (co < P.cout) {
                T* yp = Y + ((int64_t)b0 * P.cout + co) * ovol + (int64_t)z0 * oplane + (int64_t)(y0 + t / TX) * P.wout + x0 + 16 * (t % TX) + 4 * g;
                st_u32x2 pk;
                pk[0] = pack2<T>(acc[t][nt][0], acc[t][nt][1]);
                pk[1] = pack2<T>(acc[t][nt][2], acc[t][nt][3]);
                *reinterpret_cast<st_u32x2*>(yp) = pk;
            }
        }
    }
}

template <typename T, int TX, int KSZ, int STR>
static int launch_stem_tx(const StemDev& P, hipStream_t st) {
    const int64_t gx = (P.blocks + kStemWaves - 1) / kStemWaves;
    if (gx >= ((int64_t)1 << 31)) return SEGM_E_SHAPE;
    const dim3 grid((unsigned)gx), block(kStemWaves * 64);
    const int nt = (P.cout + 15) / 16;
    // 16-byte stores: tiles pair along x and every row segment is 16-byte aligned (SEGM_STEM_WIDE=0: the 8-byte form, A/B)
    const char* we = getenv("SEGM_STEM_WIDE");
    const bool wide = TX >= 2 && !(we && we[0] == '0') && P.wout % 8 == 0 && P.y_sc % 8 == 0 && ((uintptr_t)P.y & 15) == 0;
    if constexpr (TX >= 2) {
        if (wide) {
            if (nt == 1) hipLaunchKernelGGL((stem_conv_fwd_kernel<T, 1, TX, KSZ, STR, true>), grid, block, 0, st, P);
            else if (nt == 2) hipLaunchKernelGGL((stem_conv_fwd_kernel<T, 2, TX, KSZ, STR, true>), grid, block, 0, st, P);
            else hipLaunchKernelGGL((stem_conv_fwd_kernel<T, 3, TX, KSZ, STR, true>), grid, block, 0, st, P);
            return (int)hipGetLastError();
        }
    }
    if (nt == 1) hipLaunchKernelGGL((stem_conv_fwd_kernel<T, 1, TX, KSZ, STR>), grid, block, 0, st, P);
    else if (nt == 2) hipLaunchKernelGGL((stem_conv_fwd_kernel<T, 2, TX, KSZ, STR>), grid, block, 0, st, P);
    else hipLaunchKernelGGL((stem_conv_fwd_kernel<T, 3, TX, KSZ, STR>), grid, block, 0, st, P);
    return (int)hipGetLastError();
}

// tiles per output row -> TX (a divisor of 8 that divides it); TY = 8 / TX must divide the output height
static int stem_tx(int wout, int hout) {
    const int xt = wout / 16;
    for (int tx = 8; tx >= 1; tx >>= 1)
        if (xt % tx == 0 && hout % (kStemTiles / tx) == 0) return tx;
    return 0;
}

template <typename T, int KSZ, int STR>
static int launch_stem(StemDev& P, hipStream_t st) {
    const int tx = stem_tx(P.wout, P.hout);
    if (!tx) return SEGM_E_SHAPE;
    P.blocks = (int64_t)P.batch * P.dout * (P.hout / (kStemTiles / tx)) * (P.wout / (16 * tx));
    if (tx == 8) return launch_stem_tx<T, 8, KSZ, STR>(P, st);
    if (tx == 4) return launch_stem_tx<T, 4, KSZ, STR>(P, st);
    if (tx == 2) return launch_stem_tx<T, 2, KSZ, STR>(P, st);
    return launch_stem_tx<T, 1, KSZ, STR>(P, st);
}

// ------------------------------------------------------------------------------------------------------
// weight gradient of the stem convolution  (C ABI: segm_stem_conv_wgrad)
// ------------------------------------------------------------------------------------------------------
// dW[co][kz][ky][slot][ci] = sum over (b, z, y, x) dy[b][co][z][y][x] * x4[b][2 z + kz - 3][2 y + ky - 3][2 x + slot - 3][ci]
// MIOpen builds the 1.4 GB column matrix again (3.0 ms, profiles/r02_stem_bwd.log).  Here every output row (b, z, y) is a
// correlation along x on MFMA with the contraction over the row's voxels:
//   A[i = co][k = x]          a lane holds 8 consecutive x of one channel of dy: 16 contiguous bytes, loaded ONCE per row and kept
//                             for the 7 ky taps a wave owns (kz is the wave's, from the grid);
//   B[k = x][j = (slot, ci)]  the input row (2 z + kz - 3, 2 y + ky - 3), channel-last-4, staged into a wave-private LDS
//                             strip (one 16-byte load per lane) and gathered at stride 2 positions: 8 ds_read_u16 per fragment;
//   acc[ky][co tile][j tile]  7 x 3 x 2 tiles = 168 registers, written once per wave as a partial [slab][kz][ky][co][32];
// a second launch adds the slabs in a fixed order (deterministic, no atomics).
constexpr int kStemRowsPerWave = 64;

struct StemWgDev {
    const char* x4;                          // (B, Din, Hin, Win, 4)
    const char* dy;                          // (B, cout, Dout, Hout, Wout), dense inside a channel
    int64_t dy_sc;                           // elements between channels of dy (batch stride = cout * dy_sc)
    float* part;                             // [slabs][KSZ kz][KSZ ky][cout16][16 NJ]
    int32_t batch, cout, cout16;
    int32_t din, hin, win, dout, hout, wout;
    int64_t rows;                            // batch * dout * hout
    int32_t slabs;
};

// NT = 16-channel tiles of cout, KS = 32-voxel steps per output row, KSZ^3 taps, stride STR, padding KSZ / 2
template <typename T, int NT, int KS, int KSZ, int STR>
__global__ void __launch_bounds__(kStemWaves * 64) stem_conv_wgrad_kernel(StemWgDev P) {
    typedef typename Mfma16<T>::v8 frag8;
    constexpr int PAD = KSZ / 2;
    constexpr int NJ = KSZ > 4 ? 2 : 1;                           // 16-column tiles of (kx slot, ci): slots 0 - 3 / 4 - 7
    constexpr int WIN = KS * 32 * STR;                            // input row length (positions)
    constexpr int STRIP = (WIN + 8) * 4;                          // elements: 4 + 4 padding positions around the row (16-byte aligned stores)
    __shared__ __attribute__((aligned(16))) T s_row[kStemWaves][STRIP];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int slab = blockIdx.x * kStemWaves + wave;
    const int kz = blockIdx.y;
    if (slab >= P.slabs) return;
    T* strip = &s_row[wave][0];
    // padding positions -4 .. -1 and WIN .. WIN + 3 stay zero for the whole kernel
    if (lane < 16) strip[lane] = from_f32<T>(0.f);
    else if (lane < 32) strip[(4 + WIN) * 4 + lane - 16] = from_f32<T>(0.f);

    st_f32x4 acc[KSZ][NT][NJ];
#pragma unroll
    for (int ky = 0; ky < KSZ; ++ky)
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) acc[ky][mt][jt] = st_f32x4{0.f, 0.f, 0.f, 0.f};

    const T* X = reinterpret_cast<const T*>(P.x4);
    const T* DY = reinterpret_cast<const T*>(P.dy);
    const int64_t row_el = (int64_t)P.win * 4, plane_el = (int64_t)P.hin * row_el, vol_el = (int64_t)P.din * plane_el;
    const int64_t oplane = (int64_t)P.hout * P.wout, ovol = P.dy_sc;
    // the lane's gather offsets: column j = i16 -> slot 4 jt + i16 / 4, channel i16 % 4; voxel x = 32 ks + 8 g + e
    const int ci = i16 & 3, sl = i16 >> 2;

    const int64_t r0 = (int64_t)slab * kStemRowsPerWave;
    for (int rr = 0; rr < kStemRowsPerWave; ++rr) {
        const int64_t row = r0 + rr;
        if (row >= P.rows) break;                                // uniform
        const int y = (int)(row % P.hout);
        const int z = (int)((row / P.hout) % P.dout);
        const int b = (int)(row / ((int64_t)P.hout * P.dout));
        const int iz = STR * z + kz - PAD;
        if (iz < 0 || iz >= P.din) continue;                     // uniform: a padding plane contributes nothing
        frag8 af[NT][KS];
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
            const int co = 16 * mt + i16;
            const T* dp = DY + ((int64_t)b * P.cout + (co < P.cout ? co : 0)) * ovol + (int64_t)z * oplane + (int64_t)y * P.wout;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const st_u32x4 zero = {0u, 0u, 0u, 0u};
                const st_u32x4 v = *reinterpret_cast<const st_u32x4*>(dp + 32 * ks + 8 * g);
                af[mt][ks] = __builtin_bit_cast(frag8, co < P.cout ? v : zero);
            }
        }
#pragma unroll
        for (int ky = 0; ky < KSZ; ++ky) {
            const int iy = STR * y + ky - PAD;
            if (iy < 0 || iy >= P.hin) continue;                 // uniform
            const T* rowp = X + (int64_t)b * vol_el + (int64_t)iz * plane_el + (int64_t)iy * row_el;
            SEGM_WAVE_LDS_SYNC();                                // the previous row's gathers are done
#pragma unroll
            for (int q = 0; q < (WIN + 127) / 128; ++q)          // 64 lanes x 16 bytes = 128 positions per load
                if ((q * 64 + lane) * 8 < WIN * 4)
                    *reinterpret_cast<st_u32x4*>(strip + 16 + (q * 64 + lane) * 8) = *reinterpret_cast<const st_u32x4*>(rowp + (q * 64 + lane) * 8);
            SEGM_WAVE_LDS_SYNC();
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    // B fragment: x4row[STR (32 ks + 8 g + e) + (4 jt + sl) - PAD][ci], e = 0 .. 7   (strip index = position + 4)
                    const T* gp = strip + (STR * (32 * ks + 8 * g) + 4 * jt + sl + 4 - PAD) * 4 + ci;
                    T e8[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) e8[e] = gp[e * STR * 4];
                    frag8 bf;
                    memcpy(&bf, e8, 16);
#pragma unroll
                    for (int mt = 0; mt < NT; ++mt) acc[ky][mt][jt] = Mfma16<T>::run(af[mt][ks], bf, acc[ky][mt][jt]);
                }
            }
        }
    }
    // partial: D[co = 16 mt + 4 g + r][j = 16 jt + i16]
    float* pp = P.part + (((int64_t)slab * KSZ + kz) * KSZ) * P.cout16 * (16 * NJ);
#pragma unroll
    for (int ky = 0; ky < KSZ; ++ky)
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pp[((int64_t)ky * P.cout16 + 16 * mt + 4 * g + r) * (16 * NJ) + 16 * jt + i16] = acc[ky][mt][jt][r];
}

// dwp[n] = sum over slabs of part[slab][n], slabs added in a fixed order (wgrad_gemm.hip)
void launch_partial_sum(const float* part, int parts, int64_t n, float* out, hipStream_t stream);

}  // namespace segm

using namespace segm;

// (kernel size, stride) of a call: 0 / 0 = the stem proper
static bool stem_geometry(int32_t ksz, int32_t str, int& k, int& s) {
    k = ksz ? ksz : 7; s = str ? str : 2;
    return (k == 7 && s == 2) || (k == 3 && s == 1);
}

extern "C" int segm_stem_conv_fwd(const segm_stem_args* a) {
    if (!a) return SEGM_E_NULL;
    int ksz, str;
    if (!stem_geometry(a->kernel_size, a->stride, ksz, str)) return SEGM_E_SHAPE;
    if (a->batch <= 0 || a->cout <= 0 || a->cout > 48 || a->din <= 0 || a->hin <= 0 || a->win <= 0) return SEGM_E_SHAPE;
    if (a->din % str || a->hin % str || (a->win / str) % 16 || a->win % str) return SEGM_E_SHAPE;      // output rows in 16-voxel tiles
    if (!stem_tx(a->win / str, a->hin / str)) return SEGM_E_SHAPE;                   // 8 tiles = TX along x times TY rows
    if (a->dtype != SEGM_BF16 && a->dtype != SEGM_F16) return SEGM_E_DTYPE;
    if (!a->x4 || !a->w_packed || !a->y) return SEGM_E_NULL;
    if (((uintptr_t)a->x4 & 7) || ((uintptr_t)a->w_packed & 15) || ((uintptr_t)a->y & 7)) return SEGM_E_SHAPE;
    StemDev P;
    P.x4 = (const char*)a->x4; P.wp = (const char*)a->w_packed; P.bias = a->bias; P.y = (char*)a->y;
    P.batch = a->batch; P.cout = a->cout;
    P.din = a->din; P.hin = a->hin; P.win = a->win;
    P.dout = a->din / str; P.hout = a->hin / str; P.wout = a->win / str;
    const int64_t dense = (int64_t)P.dout * P.hout * P.wout;
    P.y_sc = a->y_channel_stride ? a->y_channel_stride : dense;
    if (P.y_sc < dense || P.y_sc % 4) return SEGM_E_SHAPE;
    hipStream_t st = (hipStream_t)a->stream;
    if (ksz == 7) return a->dtype == SEGM_F16 ? launch_stem<f16_t, 7, 2>(P, st) : launch_stem<bf16_t, 7, 2>(P, st);
    return a->dtype == SEGM_F16 ? launch_stem<f16_t, 3, 1>(P, st) : launch_stem<bf16_t, 3, 1>(P, st);
}

extern "C" size_t segm_stem_conv_wgrad_workspace_bytes2(int32_t batch, int32_t cout, int32_t din, int32_t hin, int32_t kernel_size, int32_t stride) {
    int ksz, str;
    if (batch <= 0 || cout <= 0 || din <= 0 || hin <= 0 || !stem_geometry(kernel_size, stride, ksz, str)) return 0;
    const int64_t rows = (int64_t)batch * (din / str) * (hin / str);
    const int64_t slabs = (rows + kStemRowsPerWave - 1) / kStemRowsPerWave;
    return (size_t)slabs * ksz * ksz * ((cout + 15) / 16 * 16) * (ksz > 4 ? 32 : 16) * sizeof(float);
}

extern "C" size_t segm_stem_conv_wgrad_workspace_bytes(int32_t batch, int32_t cout, int32_t din, int32_t hin) {
    return segm_stem_conv_wgrad_workspace_bytes2(batch, cout, din, hin, 7, 2);
}

template <typename T, int KSZ, int STR>
static void launch_stem_wgrad(const StemWgDev& P, int nt, int ks, dim3 grid, dim3 block, hipStream_t st) {
#define SEGM_STEM_WG(NN)                                                                                              \
    do {                                                                                                              \
        if (ks == 1) hipLaunchKernelGGL((stem_conv_wgrad_kernel<T, NN, 1, KSZ, STR>), grid, block, 0, st, P);          \
        else if (ks == 2) hipLaunchKernelGGL((stem_conv_wgrad_kernel<T, NN, 2, KSZ, STR>), grid, block, 0, st, P);     \
        else hipLaunchKernelGGL((stem_conv_wgrad_kernel<T, NN, 4, KSZ, STR>), grid, block, 0, st, P);                  \
    } while (0)
    if (nt == 1) SEGM_STEM_WG(1); else if (nt == 2) SEGM_STEM_WG(2); else SEGM_STEM_WG(3);
#undef SEGM_STEM_WG
}

extern "C" int segm_stem_conv_wgrad(const segm_stem_wgrad_args* a) {
    if (!a) return SEGM_E_NULL;
    int ksz, str;
    if (!stem_geometry(a->kernel_size, a->stride, ksz, str)) return SEGM_E_SHAPE;
    if (a->batch <= 0 || a->cout <= 0 || a->cout > 48 || a->din <= 0 || a->hin <= 0 || a->win <= 0) return SEGM_E_SHAPE;
    const int wout = a->win / str;
    if (a->din % str || a->hin % str || a->win % str || (wout != 32 && wout != 64 && wout != 128)) return SEGM_E_SHAPE;   // rows of 1, 2 or 4 k-steps
    if (a->dtype != SEGM_BF16 && a->dtype != SEGM_F16) return SEGM_E_DTYPE;
    if (!a->x4 || !a->dy || !a->dw_packed) return SEGM_E_NULL;
    if (((uintptr_t)a->x4 & 15) || ((uintptr_t)a->dy & 15)) return SEGM_E_SHAPE;
    const size_t need = segm_stem_conv_wgrad_workspace_bytes2(a->batch, a->cout, a->din, a->hin, ksz, str);
    if (!a->workspace || a->workspace_bytes < need) return SEGM_E_WORKSPACE;
    StemWgDev P;
    P.x4 = (const char*)a->x4; P.dy = (const char*)a->dy; P.part = (float*)a->workspace;
    P.batch = a->batch; P.cout = a->cout; P.cout16 = (a->cout + 15) / 16 * 16;
    P.din = a->din; P.hin = a->hin; P.win = a->win;
    P.dout = a->din / str; P.hout = a->hin / str; P.wout = wout;
    const int64_t dense = (int64_t)P.dout * P.hout * P.wout;
    P.dy_sc = a->dy_channel_stride ? a->dy_channel_stride : dense;
    if (P.dy_sc < dense || P.dy_sc % 8) return SEGM_E_SHAPE;
    P.rows = (int64_t)P.batch * P.dout * P.hout;
    P.slabs = (int32_t)((P.rows + kStemRowsPerWave - 1) / kStemRowsPerWave);
    hipStream_t st = (hipStream_t)a->stream;
    const dim3 grid((unsigned)((P.slabs + kStemWaves - 1) / kStemWaves), ksz), block(kStemWaves * 64);
    const int nt = P.cout16 / 16, ks = wout / 32;
    if (ksz == 7) {
        if (a->dtype == SEGM_F16) launch_stem_wgrad<f16_t, 7, 2>(P, nt, ks, grid, block, st);
        else launch_stem_wgrad<bf16_t, 7, 2>(P, nt, ks, grid, block, st);
    } else {
        if (a->dtype == SEGM_F16) launch_stem_wgrad<f16_t, 3, 1>(P, nt, ks, grid, block, st);
        else launch_stem_wgrad<bf16_t, 3, 1>(P, nt, ks, grid, block, st);
    }
    const int64_t n = (int64_t)ksz * ksz * P.cout16 * (ksz > 4 ? 32 : 16);
    launch_partial_sum(P.part, P.slabs, n, a->dw_packed, st);
    return (int)hipGetLastError();
}
