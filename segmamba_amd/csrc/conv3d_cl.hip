// Forward (and, with flipped / transposed weights, data-gradient) 3x3x3 stride-1 pad-1 convolution on CHANNEL-LAST 16-bit volumes,
// 48 input and 48 output channels (C ABI: segm_conv3d_k3_fwd_cl).  Round 6 prototype of VERDICT r05 item 1.
//
// Replaces torch.nn.Conv3d -> cuDNN for the 48-channel 3x3x3 layers of SegMamba's stem / decoder (reference
// model_segmamba/segmamba.py:91-132, monai/networks/blocks/dynunet_block.py:44-111) like csrc/conv3d_fwd.hip does, for activations
// stored (batch, depth, height, width, channel).  The NCDHW kernel spends its time on a per-row chain barrier -> LDS hand-off ->
// MFMA -> transposing park -> barrier: MFMA wants the contraction index (ci) contiguous per lane and NCDHW has x
// contiguous.  With channels last an operand fragment is 16 contiguous bytes of global memory, and the layout allows a different
// decomposition altogether:
//
//   THE kx TAPS MOVE FROM K TO N.   P[kx][co][v] = sum_{kz, ky, ci} W[co, ci, kz, ky, kx] * X[v + (kz - 1, ky - 1, 0)][ci]
//                                   Y[co][x]     = P[0][co][x - 1] + P[1][co][x] + P[2][co][x + 1]
//   i.e. one GEMM with K = 9 * 48 = 432 (13.5 MFMA k-steps) and N' = 3 * 48 = 144 rows of weights (9 tiles), followed by a shift-and-add
//   of fp32 accumulators along x.  The flop count is that of the direct form (27 * 48 * 48 per voxel); what changes is that an
//   input fragment needs NO x shift, is loaded once per (row, k-step) and feeds 9 MFMAs, and the shift happens once per output
//   tile on accumulators (DPP row shifts: the voxel index is the lane index inside a row of 16) instead of 27 times on operands.
//
//   * weights (144 x 448 bf16 = 126 KB, pre-arranged by the host as MFMA A-operand fragments [k-step][tile][lane][8]) live in LDS for
//     the whole launch: one ds_read_b128 per fragment, conflict-free, shared by the M voxel tiles a wave works on;
//   * input fragments come straight from global memory: lane (j = voxel of the tile, g = k group) loads the 16 bytes
//     X[row(kz, ky)][x0 + j][ci0 .. ci0 + 7] with a raw buffer load whose lane offset is fixed per k-step and whose scalar offset
//     walks the row; a (kz, ky) row outside the volume is a lane offset beyond num_records - the range check returns the zeros of
//     the padding (as in stem.hip / conv3d_wgrad.hip).  x padding needs no loads at all: P[0][-1] and P[2][W] are simply absent;
//   * no LDS traffic for activations, no barrier after the prologue: every wave owns whole output rows and walks them in groups
//     of M tiles (16 M voxels), the last tile of a group waiting in registers for P[2] of the next group's first voxel;
//   * persistent workgroups (one per CU, the weights are staged once), rows dealt so that the rows of a plane and the planes
//     of an XCD are neighbours (segm_device.h xcd_item).
//
// v_mfma_f32_16x16x32: A[i][k]: lane l holds A[i = l & 15][8 (l >> 4) .. +7]; B[k][j]: lane l holds B[8 (l >> 4) .. +7][j = l & 15];
// D[row = 4 (l >> 4) + r][col = l & 15].  Here i / row = weight row (kx, co), j / col = voxel: a lane ends up with four output
// channels of one voxel per tile.  The weight rows of a kx are dealt to the three tiles so that lane group g holds channels
// 8 g .. 8 g + 7 (tiles 0, 1) and 32 + 4 g .. + 3 (tile 2): one 16-byte and one 8-byte store per voxel and lane.
#include <stdlib.h>
#include <string.h>

#include "segm_device.h"

namespace segm {

typedef float cl_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t cl_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t cl_u32x2 __attribute__((ext_vector_type(2)));

constexpr int kClC = 48;                       // channels, in and out
constexpr int kClK = 9 * kClC;                 // contraction length: (kz, ky, ci)
constexpr int kClKS = (kClK + 31) / 32;        // 14 k-steps, the last one half empty
constexpr int kClNT = 9;                       // 16-row tiles of the weight operand: kx * 3 + co tile
constexpr int kClFrags = kClKS * kClNT;        // 126 fragments of 1 KB
constexpr uint32_t kClNumRec = 0x80000000u;    // bytes the input descriptor covers; a lane offset at / beyond it reads zeros

struct ConvClDev {
    const char* x;  int64_t x_sb, x_sz, x_sy, x_sx;       // element strides; channels contiguous
    char* y;        int64_t y_sb, y_sz, y_sy, y_sx;
    const void* wimg;                                     // [14][9][64][8] fragments (segm_conv3d_k3_cl_pack_index)
    const float* bias;                                    // (48) or null
    int32_t B, D, H, W;
    int32_t nrows, rows_per_wg;
};

// the channel a lane holds in register r of co tile ct (g = lane >> 4)
__host__ __device__ __forceinline__ int cl_channel(int ct, int g, int r) { return ct < 2 ? 8 * g + 4 * ct + r : 32 + 4 * g + r; }

template <int CTRL, bool ZERO_FILL>
__device__ __forceinline__ float cl_dpp(float old, float src) {      // ZERO_FILL: lanes without a source read 0; otherwise they keep `old`
    return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(old), __float_as_uint(src), CTRL, 0xf, 0xf, ZERO_FILL));
}
// v[j - 1] of this tile; lane 0 of every row of 16 takes lane 15 of `left` (the tile to the left)
__device__ __forceinline__ float cl_from_left(float left, float v) {
    const float edge = cl_dpp<0x121, true>(0.f, left);      // row_ror:1 - lane 0 <- lane 15 (every lane has a source: no `old`)
    return cl_dpp<0x111, false>(edge, v);                  // row_shr:1 - lanes 1 .. 15 <- lane - 1, lane 0 keeps the edge
}
// v[j + 1] of this tile; lane 15 takes lane 0 of `right`
__device__ __forceinline__ float cl_from_right(float right, float v) {
    const float edge = cl_dpp<0x12f, true>(0.f, right);     // row_ror:15 - lane 15 <- lane 0
    return cl_dpp<0x101, false>(edge, v);                  // row_shl:1 - lanes 0 .. 14 <- lane + 1, lane 15 keeps the edge
}

#ifdef SEGM_CL_TIMELINE
// experiments only (tools/gpu_conv_cl_timeline.py): per wave, the cycles spent in the k loops, in the epilogues and in all, [wave][8] u64
__device__ unsigned long long* g_cl_timeline = nullptr;
#define CL_T(v) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : : "memory"); } while (0)
#else
#define CL_T(v) ((void)0)
#endif

constexpr int kClRing = 6;                     // weight fragments in flight LDS -> registers (divides 126: the ring phase is the same in every group)

// M = voxel tiles of 16 a wave works on at once (W % (16 M) == 0); WAVES per workgroup (4 = one per SIMD, 8 = two); PF = how many
// k-steps ahead of the MFMAs the input fragments are requested (an L2 miss is 2 000+ cycles away, a k-step 16 x 9 M cycles)
template <typename T, int M, int WAVES, bool ACC, int PF>
__global__ void __launch_bounds__(WAVES * 64, WAVES / 4) conv3d_k3_fwd_cl_kernel(ConvClDev P) {
    typedef typename Mfma16<T>::v8 frag8;
    constexpr int RX = PF + 1;                             // input fragment slots: k-step s lives in slot s % RX
    static_assert(kClFrags % kClRing == 0, "ring phase");
    __shared__ __attribute__((aligned(16))) cl_u32x4 s_w[kClFrags * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j16 = lane & 15, g = lane >> 4;
    unsigned long long tl_t0 = 0, tl_a = 0, tl_b = 0, tl_c = 0, tl_k = 0, tl_e = 0, tl_p = 0, tl_n = 0;
    (void)tl_t0; (void)tl_a; (void)tl_b; (void)tl_c; (void)tl_k; (void)tl_e; (void)tl_p; (void)tl_n;
    CL_T(tl_t0);
    {
        const cl_u32x4* src = reinterpret_cast<const cl_u32x4*>(P.wimg);
        for (int i = tid; i < kClFrags * 64; i += WAVES * 64) s_w[i] = src[i];
    }
    __syncthreads();
    CL_T(tl_a);
    tl_p = tl_a - tl_t0;

    // per k-step: which (kz, ky) row and which eight channels this lane's fragment slice is, as a byte offset from the row
    // (z - 1, y - 1) of the batch the descriptor is based at.  Lane groups 0 / 1 and 2 / 3 of a k-step may sit on different rows.
    uint32_t cvoff[kClKS];
#pragma unroll
    for (int s = 0; s < kClKS; ++s) {
        const int k0 = 32 * s + 8 * g;
        const int r9 = k0 / kClC, ci0 = k0 - r9 * kClC;
        const int kz = r9 / 3, ky = r9 - 3 * kz;
        cvoff[s] = (uint32_t)(((int64_t)kz * P.x_sz + (int64_t)ky * P.x_sy + (int64_t)j16 * P.x_sx + ci0) * 2);
    }
    // the lane offset of k-step s under the row mask (bit kz * 3 + ky: that input row exists): a row outside the volume is an
    // offset beyond num_records.  s is a compile-time number at every call: the row of a lane is (32 s + 8 g) / 48.
    auto lane_off = [&](int s, uint32_t mask9) -> uint32_t {
        const int lo = (32 * s) / kClC, hi = (32 * s + 24) / kClC;          // rows of lane groups 0 and 3
        const bool ok_lo = (mask9 >> lo) & 1u, ok_hi = hi < 9 && ((mask9 >> hi) & 1u);
        const bool in_hi = (32 * s + 8 * g) / kClC != lo;
        return (in_hi ? ok_hi : ok_lo) ? cvoff[s] : kClNumRec;
    };
    float bias_r[3][4];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias_r[ct][r] = P.bias ? P.bias[cl_channel(ct, g, r)] : 0.f;

    const int wg = xcd_item(blockIdx.x, gridDim.x);
    const int row_begin = wg * P.rows_per_wg;
    const int row_end = row_begin + P.rows_per_wg < P.nrows ? row_begin + P.rows_per_wg : P.nrows;
    const uint32_t tile_bytes_x = (uint32_t)(16 * P.x_sx * 2), tile_bytes_y = (uint32_t)(16 * P.y_sx * 2);
    const uint32_t yvoff = (uint32_t)(j16 * P.y_sx * 2) + 16u * (uint32_t)g, yvoff2 = (uint32_t)(j16 * P.y_sx * 2) + 64u + 8u * (uint32_t)g;
    const int ngroups = P.W / (16 * M);

    // what is uniform over a row: batch descriptor, scalar row offset, row mask
    struct RowAt { __amdgpu_buffer_rsrc_t rx; uint32_t xrow, mask9; int b, z, y; };
    auto row_at = [&](int row) {
        RowAt r;
        r.y = row % P.H;
        const int bz = row / P.H;
        r.z = bz % P.D;
        r.b = bz / P.D;
        uint32_t m = 0;
#pragma unroll
        for (int kz = 0; kz < 3; ++kz)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const bool ok = r.z + kz - 1 >= 0 && r.z + kz - 1 < P.D && r.y + ky - 1 >= 0 && r.y + ky - 1 < P.H;
                m |= (ok ? 1u : 0u) << (kz * 3 + ky);
            }
        r.mask9 = m;
        // descriptor based one plane and one row BEFORE the batch (never dereferenced there: those rows are masked), so that every
        // part of an address is non-negative: lane offset (kz, ky, j, ci0) + scalar offset (z, y, x0)
        r.rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(P.x) + ((int64_t)r.b * P.x_sb - P.x_sz - P.x_sy) * 2, 0, (int)kClNumRec, 0x00020000);
        r.xrow = (uint32_t)(((int64_t)r.z * P.x_sz + (int64_t)r.y * P.x_sy) * 2);
        return r;
    };
    auto load_step = [&](frag8 (&dst)[M], const RowAt& ra, int s, uint32_t xg) {
        const uint32_t vo = lane_off(s, ra.mask9);
#pragma unroll
        for (int t = 0; t < M; ++t)
            dst[t] = __builtin_bit_cast(frag8, __builtin_amdgcn_raw_buffer_load_b128(ra.rx, vo, ra.xrow + xg + (uint32_t)t * tile_bytes_x, 0));
    };

    frag8 xf[RX][M];                                       // input fragments of k-steps s .. s + PF
    frag8 wr[kClRing];                                     // weight fragments idx .. idx + kClRing - 1 (idx = s * 9 + tile)
    if (row_begin + wave < row_end) {
        const RowAt r0 = row_at(row_begin + wave);
#pragma unroll
        for (int i = 0; i < PF; ++i) load_step(xf[i], r0, i, 0u);
#pragma unroll
        for (int i = 0; i < kClRing - 1; ++i) wr[i] = __builtin_bit_cast(frag8, s_w[i * 64 + lane]);
    }
    for (int row = row_begin + wave; row < row_end; row += WAVES) {
        const RowAt ra = row_at(row);
        const bool more_rows = row + WAVES < row_end;
        const RowAt rn = row_at(more_rows ? row + WAVES : row);    // where the prefetch behind the last group goes
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
            P.y + ((int64_t)ra.b * P.y_sb + (int64_t)ra.z * P.y_sz + (int64_t)ra.y * P.y_sy) * 2, 0, (int)kClNumRec, 0x00020000);

        float pend[3][4], left0[3][4];                     // the waiting last tile of the previous group; its P[0] (kx = 0) values
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) { pend[ct][r] = 0.f; left0[ct][r] = 0.f; }

        auto store_tile = [&](const float (&o)[3][4], uint32_t ybytes) {
            float v[3][4];
#pragma unroll
            for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[ct][r] = o[ct][r];
            if (ACC) {
                const cl_u32x4 o4 = __builtin_amdgcn_raw_buffer_load_b128(ry, yvoff, ybytes, 0);
                const cl_u32x2 o2 = __builtin_amdgcn_raw_buffer_load_b64(ry, yvoff2, ybytes, 0);
                T t8[8], t4[4];
                memcpy(t8, &o4, 16);
                memcpy(t4, &o2, 8);
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[0][r] += to_f32(t8[r]); v[1][r] += to_f32(t8[4 + r]); v[2][r] += to_f32(t4[r]); }
            }
            const cl_u32x4 p4 = {pack2<T>(v[0][0], v[0][1]), pack2<T>(v[0][2], v[0][3]), pack2<T>(v[1][0], v[1][1]), pack2<T>(v[1][2], v[1][3])};
            const cl_u32x2 p2 = {pack2<T>(v[2][0], v[2][1]), pack2<T>(v[2][2], v[2][3])};
            __builtin_amdgcn_raw_buffer_store_b128(p4, ry, yvoff, ybytes, 0);
            __builtin_amdgcn_raw_buffer_store_b64(p2, ry, yvoff2, ybytes, 0);
        };

        for (int gi = 0; gi < ngroups; ++gi) {
            const uint32_t xg = (uint32_t)gi * (uint32_t)M * tile_bytes_x;
            const bool last_group = gi + 1 == ngroups;
            CL_T(tl_a);
            cl_f32x4 acc[M][kClNT];
#pragma unroll
            for (int t = 0; t < M; ++t)
#pragma unroll
                for (int nt = 0; nt < kClNT; ++nt) {
                    const int ct = nt % 3;
                    acc[t][nt] = nt / 3 == 1 ? cl_f32x4{bias_r[ct][0], bias_r[ct][1], bias_r[ct][2], bias_r[ct][3]} : cl_f32x4{0.f, 0.f, 0.f, 0.f};
                }
            // One flat stream over the 126 weight fragments.  Per fragment: the LDS read of the fragment kClRing - 1 ahead (beyond
            // the last one: the next group's first ones - the weights are the same for every group), then the M MFMAs on it; per
            // k-step also the input fragments two k-steps ahead (behind the last two of a group: the first two of the next group, or of
            // this wave's next row).  The order written is the order issued (scheduling fences): nothing waits for a read it just issued.
#pragma unroll
            for (int s = 0; s < kClKS; ++s) {
                if (s + PF < kClKS) load_step(xf[(s + PF) % RX], ra, s + PF, xg);
                else if (!last_group) load_step(xf[(s + PF) % RX], ra, s + PF - kClKS, xg + (uint32_t)M * tile_bytes_x);
                else load_step(xf[(s + PF) % RX], rn, s + PF - kClKS, 0u);
                SEGM_SCHED_FENCE();
#pragma unroll
                for (int nt = 0; nt < kClNT; ++nt) {
                    const int idx = s * kClNT + nt;
                    wr[(idx + kClRing - 1) % kClRing] = __builtin_bit_cast(frag8, s_w[((idx + kClRing - 1) % kClFrags) * 64 + lane]);
#pragma unroll
                    for (int t = 0; t < M; ++t) acc[t][nt] = Mfma16<T>::run(wr[idx % kClRing], xf[s % RX][t], acc[t][nt]);
                    SEGM_SCHED_FENCE();
                }
            }
            CL_T(tl_b);
            // the next group's k-steps 0 .. PF - 1 sit in slots (14 + i) % RX: to slots i (nothing to do when RX divides 14)
            if constexpr (kClKS % RX != 0) {
                frag8 nx[PF][M];
#pragma unroll
                for (int i = 0; i < PF; ++i)
#pragma unroll
                    for (int t = 0; t < M; ++t) nx[i][t] = xf[(kClKS + i) % RX][t];
#pragma unroll
                for (int i = 0; i < PF; ++i)
#pragma unroll
                    for (int t = 0; t < M; ++t) xf[i][t] = nx[i][t];
            }

            // ---- shift-and-add along x, then the stores; tile M - 1 waits for the next group's first voxel -------------------------
            // the tile that waited: its last voxel gets P[2] of this group's first voxel
            if (gi > 0) {
                float o[3][4];
#pragma unroll
                for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = cl_dpp<0x12f, true>(0.f, acc[0][6 + ct][r]);      // lane 15 <- lane 0
                        o[ct][r] = pend[ct][r] + (j16 == 15 ? e : 0.f);
                    }
                store_tile(o, (uint32_t)(gi * M - 1) * tile_bytes_y);
            }
#pragma unroll
            for (int t = 0; t < M; ++t) {
                float o[3][4];
#pragma unroll
                for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float lf = t == 0 ? left0[ct][r] : acc[t - 1][ct][r];
                        const float a = cl_from_left(lf, acc[t][ct][r]);
                        const float rt = t + 1 < M ? acc[t + 1][6 + ct][r] : 0.f;
                        const float c = cl_from_right(rt, acc[t][6 + ct][r]);
                        o[ct][r] = acc[t][3 + ct][r] + a + c;
                    }
                if (t + 1 < M) {
                    store_tile(o, (uint32_t)gi * (uint32_t)M * tile_bytes_y + (uint32_t)t * tile_bytes_y);
                } else {
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { pend[ct][r] = o[ct][r]; left0[ct][r] = acc[M - 1][ct][r]; }
                }
            }
            CL_T(tl_c);
            tl_k += tl_b - tl_a; tl_e += tl_c - tl_b; tl_n += 1;
        }
        store_tile(pend, (uint32_t)(ngroups * M - 1) * tile_bytes_y);      // the row's last tile: nothing to its right
    }
#ifdef SEGM_CL_TIMELINE
    CL_T(tl_c);
    if (lane == 0 && g_cl_timeline) {
        unsigned long long* o = g_cl_timeline + ((size_t)blockIdx.x * WAVES + wave) * 8;
        o[0] = tl_c - tl_t0; o[1] = tl_k; o[2] = tl_e; o[3] = tl_n; o[4] = tl_p; o[5] = tl_t0; o[6] = tl_c;
        o[7] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
#endif
}

}  // namespace segm

using namespace segm;

// Index map of the weight image: out[i] = flat index into w (cout = 48, cin = 48, 3, 3, 3) contiguous, or -1 for a zero, for
// i over [14][9][64][8].  The host gathers the image with it (ops_raw.conv3d_cl_weight_image); exported so that the arrangement has
// one definition.
extern "C" int segm_conv3d_k3_cl_pack_index(int32_t* out, int64_t n) {
    if (!out || n != (int64_t)kClFrags * 64 * 8) return SEGM_E_SHAPE;
    for (int s = 0; s < kClKS; ++s)
        for (int nt = 0; nt < kClNT; ++nt)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int i16 = l & 15, g = l >> 4, kx = nt / 3, ct = nt % 3;
                    const int co = cl_channel(ct, i16 >> 2, i16 & 3);
                    const int k = 32 * s + 8 * g + e;
                    int32_t idx = -1;
                    if (k < kClK) {
                        const int r9 = k / kClC, ci = k % kClC, kz = r9 / 3, ky = r9 % 3;
                        idx = (((co * kClC + ci) * 3 + kz) * 3 + ky) * 3 + kx;
                    }
                    out[(((int64_t)s * kClNT + nt) * 64 + l) * 8 + e] = idx;
                }
    return SEGM_OK;
}

extern "C" int segm_conv3d_k3_fwd_cl(const segm_conv3d_cl_args* a) {
    if (!a) return SEGM_E_NULL;
    if (!a->x || !a->y || !a->w_image) return SEGM_E_NULL;
    if (a->batch <= 0 || a->depth <= 0 || a->height <= 0 || a->width <= 0) return SEGM_E_SHAPE;
    if (a->channels != kClC || a->width % 16 != 0) return SEGM_E_SHAPE;
    if (a->dtype != SEGM_BF16 && a->dtype != SEGM_F16) return SEGM_E_DTYPE;
    if (a->flags & ~(SEGM_CONV_CL_ACCUMULATE | SEGM_CONV_CL_WAVES8)) return SEGM_E_SHAPE;
    const int64_t st[8] = {a->x_stride_b, a->x_stride_z, a->x_stride_y, a->x_stride_x, a->y_stride_b, a->y_stride_z, a->y_stride_y, a->y_stride_x};
    for (int64_t s : st)
        if (s % 8 != 0 || s < kClC) return SEGM_E_SHAPE;     // 16-byte aligned voxels
    if (((uintptr_t)a->x & 15) || ((uintptr_t)a->y & 15) || ((uintptr_t)a->w_image & 15)) return SEGM_E_SHAPE;
    // 32-bit byte offsets inside a batch: the lane part (two planes + a row of the input) and the scalar part (z, y, x) stay below 2^31
    const int64_t span_x = ((int64_t)(a->depth + 1) * a->x_stride_z + (int64_t)(a->height + 1) * a->x_stride_y + (int64_t)a->width * a->x_stride_x) * 2;
    const int64_t span_y = ((int64_t)a->depth * a->y_stride_z + (int64_t)a->height * a->y_stride_y + (int64_t)a->width * a->y_stride_x) * 2;
    if (span_x >= (int64_t)kClNumRec || span_y >= (int64_t)kClNumRec) return SEGM_E_SHAPE;

    ConvClDev P;
    memset(&P, 0, sizeof(P));
    P.x = (const char*)a->x; P.x_sb = a->x_stride_b; P.x_sz = a->x_stride_z; P.x_sy = a->x_stride_y; P.x_sx = a->x_stride_x;
    P.y = (char*)a->y; P.y_sb = a->y_stride_b; P.y_sz = a->y_stride_z; P.y_sy = a->y_stride_y; P.y_sx = a->y_stride_x;
    P.wimg = a->w_image; P.bias = a->bias;
    P.B = a->batch; P.D = a->depth; P.H = a->height; P.W = a->width;
    const int64_t nrows = (int64_t)a->batch * a->depth * a->height;
    if (nrows >= ((int64_t)1 << 31)) return SEGM_E_SHAPE;
    P.nrows = (int32_t)nrows;
    const bool w8 = (a->flags & SEGM_CONV_CL_WAVES8) != 0;
    const int waves = w8 ? 8 : 4;
    static const int ncu = [] { const char* e = getenv("SEGM_CL_WORKGROUPS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();
    int nwg = (int)((nrows + waves - 1) / waves);
    nwg = nwg < ncu ? nwg : ncu;
    // whole multiples of the wave count per workgroup, so that the rows of one wave round are neighbours in y
    int64_t rpw = (nrows + nwg - 1) / nwg;
    rpw = (rpw + waves - 1) / waves * waves;
    nwg = (int)((nrows + rpw - 1) / rpw);
    P.rows_per_wg = (int32_t)rpw;
    hipStream_t stream = (hipStream_t)a->stream;
    const bool f16 = a->dtype == SEGM_F16;
    const bool acc = (a->flags & SEGM_CONV_CL_ACCUMULATE) != 0;
    const int m = a->width % 64 == 0 ? 4 : (a->width % 32 == 0 ? 2 : 1);
    static const int pf = [] { const char* e = getenv("SEGM_CL_PREFETCH"); const int v = e ? atoi(e) : 6; return v == 2 ? 2 : 6; }();
#define SEGM_CL_LAUNCH(T, M_, W_, A_)                                                                                                   \
    do {                                                                                                                                 \
        if (pf == 2) hipLaunchKernelGGL((conv3d_k3_fwd_cl_kernel<T, M_, W_, A_, 2>), dim3(nwg), dim3(W_ * 64), 0, stream, P);             \
        else hipLaunchKernelGGL((conv3d_k3_fwd_cl_kernel<T, M_, W_, A_, 6>), dim3(nwg), dim3(W_ * 64), 0, stream, P);                    \
    } while (0)
#define SEGM_CL_T(M_, W_)                                                                                   \
    do {                                                                                                     \
        if (f16) { if (acc) SEGM_CL_LAUNCH(f16_t, M_, W_, true); else SEGM_CL_LAUNCH(f16_t, M_, W_, false); } \
        else { if (acc) SEGM_CL_LAUNCH(bf16_t, M_, W_, true); else SEGM_CL_LAUNCH(bf16_t, M_, W_, false); }   \
    } while (0)
    if (w8) {                                              // two waves per SIMD: at most two tiles per wave (256 registers)
        if (m >= 2) SEGM_CL_T(2, 8); else SEGM_CL_T(1, 8);
    } else {
        if (m == 4) SEGM_CL_T(4, 4); else if (m == 2) SEGM_CL_T(2, 4); else SEGM_CL_T(1, 4);
    }
#undef SEGM_CL_T
#undef SEGM_CL_LAUNCH
    return (int)hipGetLastError();
}

#ifdef SEGM_CL_TIMELINE
extern "C" int segm_debug_set_cl_timeline(void* buf) {
    unsigned long long* p = (unsigned long long*)buf;
    return hipMemcpyToSymbol(HIP_SYMBOL(segm::g_cl_timeline), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif
