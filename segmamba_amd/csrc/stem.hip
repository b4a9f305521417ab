// The stem convolution of the encoder: Conv3d(kernel 7, stride 2, padding 3) on a few input channels   (C ABI: segm_stem_conv_fwd)
//
// model_segmamba/segmamba.py:141 (`nn.Conv3d(in_chans, dims[0], kernel_size=7, stride=2, padding=3)`), cuDNN in the reference.
// MIOpen runs it as im2col + GEMM: a 1.4 GB column matrix for a 2 x 4 x 128^3 input, 1.37 ms forward (profiles/r02_torch_prof.log)
// for 69 GFLOP and 84 MB of real traffic.  Here it is an implicit GEMM on MFMA whose K axis is laid out so that every operand
// fragment is one contiguous piece of memory:
//   * the input is channel-last with 4 channels, x4[b][z][y][x][4] (one transposing copy of the small input, made by the
//     host); K = (kz, ky, kx slot 0..7, ci 0..3): one K = 32 chunk per (kz, ky) pair = 49 MFMA steps (slot 7 has zero weights);
//   * A fragment of output voxel (z, y, x) for kx slots 2 g, 2 g + 1 = the 4 channels of input positions 2 x + 2 g - 3 and
//     2 x + 2 g - 2 on row (2 z + kz - 3, 2 y + ky - 3): two 8-byte loads with their own bounds masks (padding = zero);
//   * weights packed as wp[co][kz][ky][slot][ci]: a B fragment is 16 contiguous bytes; a wave keeps the three 16-channel
//     fragments of a (kz, ky) step for EIGHT 16-voxel tiles (96 accumulator registers), so weight traffic is 1/8 of the
//     activation traffic and both come from L1 / L2;
//   * D[voxel][co]: a lane holds four consecutive x of one output channel: 8-byte NCDHW stores, bias in the accumulators.
// v_mfma_f32_16x16x32: A[i][k]: lane l holds A[i = l & 15][8 (l >> 4) .. +7]; B[k][j]: lane l holds B[8 (l >> 4) .. +7][j = l & 15];
// D[row = 4 (l >> 4) + r][col = l & 15].  Here i = output voxel (16 consecutive x), k = (kx slot, ci), j = output channel.
#include <string.h>

#include "segm_device.h"

namespace segm {

typedef float st_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t st_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t st_u32x2 __attribute__((ext_vector_type(2)));

constexpr int kStemWaves = 4;
constexpr int kStemTiles = 8;                // 16-voxel tiles per wave
constexpr int kStemK = 7;

struct StemDev {
    const char* x4;                          // (B, Din, Hin, Win, 4) contiguous
    const char* wp;                          // (cout, 7, 7, 8, 4) contiguous
    const float* bias;
    char* y;                                 // (B, cout, Dout, Hout, Wout) contiguous
    int32_t batch, cout;
    int32_t din, hin, win, dout, hout, wout;
    int64_t blocks;                          // wave blocks: batch * dout * (hout / TY) * (wout / 16 / TX), TX * TY = kStemTiles
};

// A wave owns a block of TX x TY = 8 tiles: TX 16-voxel tiles along x on each of TY consecutive output rows (TX = min(8, tiles
// per row)), so its tiles differ only by compile-time offsets from one (batch, z, y0, x0).
template <typename T, int NT, int TX>
__global__ void __launch_bounds__(kStemWaves * 64, 2) stem_conv_fwd_kernel(StemDev P) {
    typedef typename Mfma16<T>::v8 frag8;
    constexpr int TY = kStemTiles / TX;
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

#pragma unroll 1
    for (int kz = 0; kz < kStemK; ++kz) {
#pragma unroll 1
        for (int ky = 0; ky < kStemK; ++ky) {
            frag8 wf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = 16 * nt + i16;
                const st_u32x4 zero = {0u, 0u, 0u, 0u};
                const st_u32x4 v = *reinterpret_cast<const st_u32x4*>(W + (((int64_t)(co < P.cout ? co : 0) * kStemK + kz) * kStemK + ky) * 32 + 8 * g);
                wf[nt] = __builtin_bit_cast(frag8, co < P.cout ? v : zero);
            }
            const int iz = 2 * z0 + kz - 3;                                          // wave-uniform
            const bool z_ok = iz >= 0 && iz < P.din;
            const T* planep = X + (int64_t)b0 * vol_el + (int64_t)(z_ok ? iz : 0) * plane_el;
#pragma unroll
            for (int t = 0; t < kStemTiles; ++t) {
                const int iy = 2 * (y0 + t / TX) + ky - 3;
                const bool row_ok = z_ok && iy >= 0 && iy < P.hin;
                const int ix = 2 * (x0 + 16 * (t % TX) + i16) + 2 * g - 3;           // first of the lane's two input columns
                const T* rowp = planep + (int64_t)(row_ok ? iy : 0) * row_el;
                const bool ok0 = row_ok && ix >= 0 && ix < P.win, ok1 = row_ok && ix + 1 >= 0 && ix + 1 < P.win && !(g == 3);   // slot 7 = no tap
                const st_u32x2 z2 = {0u, 0u};
                const st_u32x2 v0 = *reinterpret_cast<const st_u32x2*>(rowp + (int64_t)(ok0 ? ix : 0) * 4);
                const st_u32x2 v1 = *reinterpret_cast<const st_u32x2*>(rowp + (int64_t)(ok1 ? ix + 1 : 0) * 4);
                const st_u32x2 a0 = ok0 ? v0 : z2, a1 = ok1 ? v1 : z2;
                const st_u32x4 av = {a0[0], a0[1], a1[0], a1[1]};
                const frag8 af = __builtin_bit_cast(frag8, av);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[t][nt] = Mfma16<T>::run(af, wf[nt], acc[t][nt]);
            }
        }
    }
    T* Y = reinterpret_cast<T*>(P.y);
    const int64_t oplane = (int64_t)P.hout * P.wout, ovol = (int64_t)P.dout * oplane;
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

template <typename T, int TX>
static int launch_stem_tx(const StemDev& P, hipStream_t st) {
    const int64_t gx = (P.blocks + kStemWaves - 1) / kStemWaves;
    if (gx >= ((int64_t)1 << 31)) return SEGM_E_SHAPE;
    const dim3 grid((unsigned)gx), block(kStemWaves * 64);
    const int nt = (P.cout + 15) / 16;
    if (nt == 1) hipLaunchKernelGGL((stem_conv_fwd_kernel<T, 1, TX>), grid, block, 0, st, P);
    else if (nt == 2) hipLaunchKernelGGL((stem_conv_fwd_kernel<T, 2, TX>), grid, block, 0, st, P);
    else hipLaunchKernelGGL((stem_conv_fwd_kernel<T, 3, TX>), grid, block, 0, st, P);
    return (int)hipGetLastError();
}

// tiles per output row -> TX (a divisor of 8 that divides it); TY = 8 / TX must divide the output height
static int stem_tx(int wout, int hout) {
    const int xt = wout / 16;
    for (int tx = 8; tx >= 1; tx >>= 1)
        if (xt % tx == 0 && hout % (kStemTiles / tx) == 0) return tx;
    return 0;
}

template <typename T>
static int launch_stem(StemDev& P, hipStream_t st) {
    const int tx = stem_tx(P.wout, P.hout);
    if (!tx) return SEGM_E_SHAPE;
    P.blocks = (int64_t)P.batch * P.dout * (P.hout / (kStemTiles / tx)) * (P.wout / (16 * tx));
    if (tx == 8) return launch_stem_tx<T, 8>(P, st);
    if (tx == 4) return launch_stem_tx<T, 4>(P, st);
    if (tx == 2) return launch_stem_tx<T, 2>(P, st);
    return launch_stem_tx<T, 1>(P, st);
}

}  // namespace segm

using namespace segm;

extern "C" int segm_stem_conv_fwd(const segm_stem_args* a) {
    if (!a) return SEGM_E_NULL;
    if (a->batch <= 0 || a->cout <= 0 || a->cout > 48 || a->din <= 0 || a->hin <= 0 || a->win <= 0) return SEGM_E_SHAPE;
    if (a->din % 2 || a->hin % 2 || a->win % 32) return SEGM_E_SHAPE;             // even extents, output rows in 16-voxel tiles
    if (!stem_tx(a->win / 2, a->hin / 2)) return SEGM_E_SHAPE;                       // 8 tiles = TX along x times TY rows
    if (a->dtype != SEGM_BF16 && a->dtype != SEGM_F16) return SEGM_E_DTYPE;
    if (!a->x4 || !a->w_packed || !a->y) return SEGM_E_NULL;
    if (((uintptr_t)a->x4 & 7) || ((uintptr_t)a->w_packed & 15) || ((uintptr_t)a->y & 7)) return SEGM_E_SHAPE;
    StemDev P;
    P.x4 = (const char*)a->x4; P.wp = (const char*)a->w_packed; P.bias = a->bias; P.y = (char*)a->y;
    P.batch = a->batch; P.cout = a->cout;
    P.din = a->din; P.hin = a->hin; P.win = a->win;
    P.dout = a->din / 2; P.hout = a->hin / 2; P.wout = a->win / 2;
    hipStream_t st = (hipStream_t)a->stream;
    return a->dtype == SEGM_F16 ? launch_stem<f16_t>(P, st) : launch_stem<bf16_t>(P, st);
}
