// Channel-first 1x1x1 convolution  y[b, co, s] = sum_ci W[co, ci] x[b, ci, s] + bias[co]   (C ABI: segm_pointwise_cf)
//
// The 1x1x1 convolutions of the conv stem on NCDHW activations - the residual branches of MONAI's UnetResBlock
// (monai/networks/blocks/dynunet_block.py:72-96, conv3), UnetOutBlock (:247-263), GSC.proj3 / proj4 and MlpChannel.fc1 / fc2
// (model_segmamba/segmamba.py:78-131) - which the reference hands to cuDNN.  Round 1 ran them as batched BLAS GEMMs on
// strided views: 48 -> 48 at 128^3 took 0.5 ms for 0.8 GB of traffic (1.6 TB/s, profiles/r02_torch_prof.log) and the bias
// was a further elementwise pass.  They are streaming operators (4.6 KB of weights against gigabytes of activations), so:
//   * W^T is STATIONARY in registers as MFMA B-operand fragments (at most 96 x 96: 18 fragments = 72 VGPRs);
//   * a wave walks 64-voxel strips: 16-byte loads along s (the contiguous axis) of two adjacent channel rows per lane,
//     paired with v_perm_b32 into {ci, ci + 1} dwords and written to a wave-private LDS tile [64 s][ci] - the transpose the
//     MFMA operand needs (K = ci must be contiguous per lane); one ds_read_b128 per A fragment; the next strip's loads are
//     in flight during the current strip's MFMAs;
//   * D[s][co] = x^T W^T: a lane ends up with four consecutive voxels of one output channel; bias in the accumulator
//     initialisation, optional accumulation into y (the second half of a concatenated input).  Round 5 (WIDE): the tile rows of a
//     PAIR of 16-voxel blocks are dealt so that the lane's 4 + 4 voxels are eight consecutive ones - 16-byte stores (and 16-byte
//     reads of the old values) instead of two 8-byte ones, which run at 0.54 - 0.70 of the 16-byte rate (MI355X_MICROARCH.md); only
//     the park's row index changes (and its bank conflicts drop from 4-way to 2-way), the fragment reads stay consecutive rows.
// v_mfma_f32_16x16x32: A[i][k]: lane l holds A[i = l & 15][8 (l >> 4) .. +7]; B[k][j]: lane l holds B[8 (l >> 4) .. +7][j = l & 15];
// D[row = 4 (l >> 4) + r][col = l & 15].  Here i = voxel, k = input channel, j = output channel.
#include <stdlib.h>
#include <string.h>

#include "segm_device.h"

namespace segm {

typedef float pw_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t pw_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t pw_u32x2 __attribute__((ext_vector_type(2)));

constexpr int kPwWaves = 4;
constexpr int kPwStrip = 64;                 // voxels per strip

struct PwDev {
    const char* x;  int64_t x_sb, x_sc;      // element strides of batch / channel (voxels contiguous)
    const char* w;  int32_t w_ld;            // (cout, w_ld) row-major, w_ld >= cin a multiple of 8 (padding columns are zero)
    const float* bias;
    char* y;        int64_t y_sb, y_sc;
    int64_t S;                               // voxels per (batch, channel): a multiple of kPwStrip
    int64_t strips;                          // batch * S / kPwStrip
    int32_t cin, cout;
    int32_t accumulate;
    int32_t strips_per_wave;
};

// KT = 32-wide chunks of cin, NT = 16-wide tiles of cout
// WIDE: tile row 32 pb + 16 h + 4 q + r holds voxel 32 pb + 8 q + 4 h + r (pb: pair of 16-voxel blocks, h: block of the pair), so MFMA
// output row 4 g + r of block 2 pb + h is voxel 32 pb + 8 g + 4 h + r: lane group g owns voxels 8 g .. 8 g + 7 of the pair.
template <typename T, int KT, int NT, bool WIDE = false>
__global__ void __launch_bounds__(kPwWaves * 64) pointwise_cf_kernel(PwDev P) {
    typedef typename Mfma16<T>::v8 frag8;
    constexpr int PITCH = KT * 32 + 8;       // elements per LDS row: 16-byte aligned rows, consecutive rows 4 banks apart
    constexpr int G16 = KT * 2;              // 16-channel groups staged per strip
    __shared__ __attribute__((aligned(16))) T s_x[kPwWaves][kPwStrip * PITCH];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const T* W = reinterpret_cast<const T*>(P.w);

    // stationary weights: fragment (kt, nt) = W[co = 16 nt + i16][ci = 32 kt + 8 g .. +7]
    frag8 wf[KT][NT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = 16 * nt + i16, ci = 32 * kt + 8 * g;
            const bool live = co < P.cout && ci < P.w_ld;
            const pw_u32x4 v = *reinterpret_cast<const pw_u32x4*>(W + (int64_t)(live ? co : 0) * P.w_ld + (live ? ci : 0));
            const pw_u32x4 zero = {0u, 0u, 0u, 0u};
            wf[kt][nt] = __builtin_bit_cast(frag8, live ? v : zero);
        }
    }
    float bias4[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias4[nt] = (P.bias && 16 * nt + i16 < P.cout) ? P.bias[16 * nt + i16] : 0.f;

    // staging roles: lane moves 8 voxels (sg) of the channel pair (2 rp, 2 rp + 1) of every 16-channel group
    const int sg = lane & 7, rp = lane >> 3;
    T* lt = &s_x[wave][0];
    const int64_t strips_per_batch = P.S / kPwStrip;
    const int64_t first = ((int64_t)blockIdx.x * kPwWaves + wave) * P.strips_per_wave;
    if (first >= P.strips) return;                               // whole waves; no workgroup barrier in this kernel

    pw_u32x4 ra[G16], rc[G16];
    auto load_strip = [&](int64_t strip) {
        const int64_t b = strip / strips_per_batch, s0 = (strip - b * strips_per_batch) * kPwStrip;
        const T* xb = reinterpret_cast<const T*>(P.x) + b * P.x_sb + s0 + 8 * sg;
#pragma unroll
        for (int q = 0; q < G16; ++q) {
            const int c0 = 16 * q + 2 * rp;
            const pw_u32x4 zero = {0u, 0u, 0u, 0u};
            const pw_u32x4 va = *reinterpret_cast<const pw_u32x4*>(xb + (int64_t)(c0 < P.cin ? c0 : 0) * P.x_sc);
            const pw_u32x4 vc = *reinterpret_cast<const pw_u32x4*>(xb + (int64_t)(c0 + 1 < P.cin ? c0 + 1 : 0) * P.x_sc);
            ra[q] = c0 < P.cin ? va : zero;
            rc[q] = c0 + 1 < P.cin ? vc : zero;
        }
    };
    auto park_strip = [&]() {
#pragma unroll
        for (int q = 0; q < G16; ++q) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {                         // {channel c0, channel c0 + 1} of voxel 8 sg + e
                const int row = WIDE ? 32 * (sg >> 2) + 16 * (e >> 2) + 4 * (sg & 3) + (e & 3) : 8 * sg + e;
                *reinterpret_cast<uint32_t*>(lt + row * PITCH + 16 * q + 2 * rp) =
                    __builtin_amdgcn_perm(rc[q][e >> 1], ra[q][e >> 1], (e & 1) ? 0x07060302u : 0x05040100u);
            }
        }
    };
    auto compute_strip = [&](int64_t strip) {
        const int64_t b = strip / strips_per_batch, s0 = (strip - b * strips_per_batch) * kPwStrip;
        T* yb = reinterpret_cast<T*>(P.y) + b * P.y_sb + s0;
        if constexpr (WIDE) {
#pragma unroll
            for (int pb = 0; pb < kPwStrip / 32; ++pb) {          // pairs of 16-voxel blocks
                frag8 a0[KT], a1[KT];
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    a0[kt] = __builtin_bit_cast(frag8, *reinterpret_cast<const pw_u32x4*>(lt + (32 * pb + i16) * PITCH + 32 * kt + 8 * g));
                    a1[kt] = __builtin_bit_cast(frag8, *reinterpret_cast<const pw_u32x4*>(lt + (32 * pb + 16 + i16) * PITCH + 32 * kt + 8 * g));
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    pw_f32x4 c0 = {bias4[nt], bias4[nt], bias4[nt], bias4[nt]}, c1 = c0;
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
                        c0 = Mfma16<T>::run(a0[kt], wf[kt][nt], c0);
                        c1 = Mfma16<T>::run(a1[kt], wf[kt][nt], c1);
                    }
                    const int co = 16 * nt + i16;
                    if (co < P.cout) {                            // this lane: voxels 32 pb + 8 g .. + 7 of output channel co
                        T* yp = yb + (int64_t)co * P.y_sc + 32 * pb + 8 * g;
                        if (P.accumulate) {                       // uniform
                            const pw_u32x4 old = *reinterpret_cast<const pw_u32x4*>(yp);
                            T o[8];
                            memcpy(o, &old, 16);
#pragma unroll
                            for (int r = 0; r < 4; ++r) { c0[r] += to_f32(o[r]); c1[r] += to_f32(o[4 + r]); }
                        }
                        pw_u32x4 pk;
                        pk[0] = pack2<T>(c0[0], c0[1]);
                        pk[1] = pack2<T>(c0[2], c0[3]);
                        pk[2] = pack2<T>(c1[0], c1[1]);
                        pk[3] = pack2<T>(c1[2], c1[3]);
                        *reinterpret_cast<pw_u32x4*>(yp) = pk;
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int vb = 0; vb < kPwStrip / 16; ++vb) {              // 16-voxel blocks of the strip
            frag8 af[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
                af[kt] = __builtin_bit_cast(frag8, *reinterpret_cast<const pw_u32x4*>(lt + (16 * vb + i16) * PITCH + 32 * kt + 8 * g));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                pw_f32x4 acc = {bias4[nt], bias4[nt], bias4[nt], bias4[nt]};
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) acc = Mfma16<T>::run(af[kt], wf[kt][nt], acc);
                const int co = 16 * nt + i16;
                if (co < P.cout) {                                // this lane: voxels 16 vb + 4 g .. + 3 of output channel co
                    T* yp = yb + (int64_t)co * P.y_sc + 16 * vb + 4 * g;
                    if (P.accumulate) {                           // uniform
                        const pw_u32x2 old = *reinterpret_cast<const pw_u32x2*>(yp);
                        T o[4];
                        memcpy(o, &old, 8);
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r] += to_f32(o[r]);
                    }
                    pw_u32x2 pk;
                    pk[0] = pack2<T>(acc[0], acc[1]);
                    pk[1] = pack2<T>(acc[2], acc[3]);
                    *reinterpret_cast<pw_u32x2*>(yp) = pk;
                }
            }
        }
    };

    const int64_t last = (first + P.strips_per_wave < P.strips ? first + P.strips_per_wave : P.strips) - 1;
    load_strip(first);
    for (int64_t st = first; st <= last; ++st) {
        SEGM_WAVE_LDS_SYNC();                                     // the previous strip's fragment reads are done
        park_strip();
        SEGM_WAVE_LDS_SYNC();
        load_strip(st < last ? st + 1 : st);                      // in flight during this strip's MFMAs
        SEGM_SCHED_FENCE();
        compute_strip(st);
        SEGM_SCHED_FENCE();
    }
}

template <typename T, int KT, bool WIDE>
static int launch_pw_nt(const PwDev& P, hipStream_t st, dim3 grid) {
    const int nt = (P.cout + 15) / 16;
    const dim3 block(kPwWaves * 64);
    switch (nt) {
        case 1: hipLaunchKernelGGL((pointwise_cf_kernel<T, KT, 1, WIDE>), grid, block, 0, st, P); break;
        case 2: hipLaunchKernelGGL((pointwise_cf_kernel<T, KT, 2, WIDE>), grid, block, 0, st, P); break;
        case 3: hipLaunchKernelGGL((pointwise_cf_kernel<T, KT, 3, WIDE>), grid, block, 0, st, P); break;
        case 4: hipLaunchKernelGGL((pointwise_cf_kernel<T, KT, 4, WIDE>), grid, block, 0, st, P); break;
        case 5: hipLaunchKernelGGL((pointwise_cf_kernel<T, KT, 5, WIDE>), grid, block, 0, st, P); break;
        default: hipLaunchKernelGGL((pointwise_cf_kernel<T, KT, 6, WIDE>), grid, block, 0, st, P); break;
    }
    return (int)hipGetLastError();
}

template <typename T>
static int launch_pw(PwDev& P, hipStream_t st) {
    // enough waves to fill the chip (>= 4 per SIMD) while a wave amortises its weight fragments over several strips
    int spw = 16;
    while (spw > 1 && P.strips / spw < 4096) spw >>= 1;
    P.strips_per_wave = spw;
    const int64_t waves = (P.strips + spw - 1) / spw;
    const int64_t gx = (waves + kPwWaves - 1) / kPwWaves;
    if (gx >= ((int64_t)1 << 31)) return SEGM_E_SHAPE;
    const dim3 grid((unsigned)gx);
    const int kt = (P.cin + 31) / 32;
    // 16-byte stores when every output row is 16-byte aligned (SEGM_POINTWISE_WIDE=0: the 8-byte form, A/B)
    static const bool wide_on = [] { const char* e = getenv("SEGM_POINTWISE_WIDE"); return !(e && e[0] == '0'); }();
    const bool wide = wide_on && P.y_sb % 8 == 0 && P.y_sc % 8 == 0 && ((uintptr_t)P.y & 15) == 0;
    if (wide) {
        if (kt == 1) return launch_pw_nt<T, 1, true>(P, st, grid);
        if (kt == 2) return launch_pw_nt<T, 2, true>(P, st, grid);
        return launch_pw_nt<T, 3, true>(P, st, grid);
    }
    if (kt == 1) return launch_pw_nt<T, 1, false>(P, st, grid);
    if (kt == 2) return launch_pw_nt<T, 2, false>(P, st, grid);
    return launch_pw_nt<T, 3, false>(P, st, grid);
}

}  // namespace segm

using namespace segm;

extern "C" int segm_pointwise_cf(const segm_pointwise_args* a) {
    if (!a) return SEGM_E_NULL;
    if (a->batch <= 0 || a->cin <= 0 || a->cout <= 0 || a->spatial <= 0) return SEGM_E_SHAPE;
    if (a->cin > 96 || a->cout > 96 || a->w_stride < a->cin || a->w_stride % 8 != 0 || a->spatial % kPwStrip != 0) return SEGM_E_SHAPE;
    if (a->dtype != SEGM_BF16 && a->dtype != SEGM_F16) return SEGM_E_DTYPE;
    if (!a->x || !a->w || !a->y) return SEGM_E_NULL;
    if (a->x_stride_b % 8 != 0 || a->x_stride_c % 8 != 0 || a->y_stride_b % 4 != 0 || a->y_stride_c % 4 != 0) return SEGM_E_SHAPE;
    if (((uintptr_t)a->x & 15) || ((uintptr_t)a->w & 15) || ((uintptr_t)a->y & 7)) return SEGM_E_SHAPE;
    PwDev P;
    P.x = (const char*)a->x; P.x_sb = a->x_stride_b; P.x_sc = a->x_stride_c;
    P.w = (const char*)a->w; P.w_ld = a->w_stride; P.bias = a->bias;
    P.y = (char*)a->y; P.y_sb = a->y_stride_b; P.y_sc = a->y_stride_c;
    P.S = a->spatial;
    P.strips = (int64_t)a->batch * (a->spatial / kPwStrip);
    P.cin = a->cin; P.cout = a->cout;
    P.accumulate = a->accumulate != 0;
    hipStream_t st = (hipStream_t)a->stream;
    return a->dtype == SEGM_F16 ? launch_pw<f16_t>(P, st) : launch_pw<bf16_t>(P, st);
}
