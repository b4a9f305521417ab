// Forward (and, with flipped weights, data-gradient) 3x3x3 stride-1 pad-1 convolution on channel-first bf16 volumes
// (C ABI: segm_conv3d_k3_fwd).
//
// Replaces torch.nn.Conv3d -> cuDNN for the 48-channel 3x3x3 layers of SegMamba's stem / GSC / decoder (reference
// model_segmamba/segmamba.py:95-131, monai/networks/blocks/dynunet_block.py:44-111).  MIOpen's best solver for them is a
// CK implicit GEMM on channel-last data: two layout-transposing kernels around a 2.2 ms GEMM per 48 -> 48 @128^3 call
// (profiles/r01_bench_step_kernels_v8.txt: 44 ms of a 145 ms training step).  This kernel works on NCDHW directly:
//
//   Y[b, co, z, y, x] = bias[co] + sum_{ci, kz, ky, kx} W[co, ci, kz, ky, kx] * X[b, ci, z+kz-1, y+ky-1, x+kx-1]
//
// as the GEMM  D[x][co] = sum_k A[x][k] B[k][co],  k = (kz, ky, kx, ci), per output row (b, z, y) and 64-wide x block.
// MFMA fragments want 8 consecutive k per lane; in NCDHW the contiguous index is x, so X rows are TRANSPOSED while they are
// staged: the LDS ring holds X as [row][x][ci] (ci contiguous), and an A fragment is one aligned 16-byte LDS read at
// [x + kx][ci0] - the three kx taps are just three row offsets, no shifting.
//
//   workgroup = 12 waves = 3 (kz) x 4 (16-wide x tiles) for one work item (b, z, x block, y range) and a block of 32
//               output channels.  Wave (kz, xt) keeps its slice of the weights - k in (ky, kx, ci), 432 values per output
//               channel, 13.5 MFMA k-chunks - STATIONARY in registers (2 co tiles x 14 fragments), so per step it only
//               reads 14 A fragments from LDS for 28 MFMAs (16x16x32 bf16).  The three kz partial sums are added
//               through LDS; wave (0, xt) adds the bias, converts and stores 4 consecutive x per lane.
//   LDS         ring of 4 rows x 3 planes (z-1, z, z+1) of [66 x][56 ci] bf16 (each X row is fetched once per work item
//               and used on three consecutive steps), double-buffered 16 KB reduction buffer: 121 KB, one workgroup per CU.
//               The next row of every plane is in flight global -> registers during the MFMAs of the current step and is
//               parked (transposed, two channels per dword) afterwards; one barrier per step.
//   padding     zero rows / columns are materialised in LDS; a tap plane outside the volume is skipped by its waves.
//
// v_mfma_f32_16x16x32_bf16: lane l holds A[i = l & 15][k = 8 (l >> 4) .. +7], B[k = 8 (l >> 4) .. +7][j = l & 15];
// result D[row = 4 (l >> 4) + r][col = l & 15].  Here row = x, col = co.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "segm_device.h"

namespace segm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kFwCi = 48;                    // input channels (K = 27 * 48)
constexpr int kFwCo = 32;                    // output channels per workgroup (2 MFMA tiles)
constexpr int kFwXB = 64;                    // x positions per workgroup
constexpr int kFwWaves = 12;                 // 3 kz x 4 x tiles
constexpr int kFwThreads = kFwWaves * 64;
constexpr int kFwCP = 56;                    // ci pitch (elements) of an LDS x position: 28 dwords -> conflict-free b128 reads
constexpr int kFwXP = kFwXB + 2;             // x positions per ring row: x0 - 1 .. x0 + 64
constexpr int kFwSlot = kFwXP * kFwCP;       // elements per ring row
constexpr int kFwKSlice = 9 * kFwCi;         // k values per kz: (ky, kx, ci) = 432
constexpr int kFwChunks = (kFwKSlice + 31) / 32;     // 14 (the last one half empty)
constexpr int kFwGran = kFwXB / 8 + 2;       // 16-byte granules fetched per (ci, row): 8 data + one halo granule each side
constexpr int kFwTasks = 3 * (kFwCi / 2) * kFwGran;  // copy tasks per step: (plane, ci pair, granule) = 720

struct ConvFwdDev {
    const char* x;  int64_t x_sb, x_sc, x_sz, x_sy;       // element strides, x contiguous
    char* y;        int64_t y_sb, y_sc, y_sz, y_sy;
    const void* wp;                                       // packed weights [cout][27 * 48], k = ((kz*3 + ky)*3 + kx)*48 + ci
    const float* bias;                                    // (cout) or null
    int32_t B, D, H, W, cout, cob0, cin;
    int32_t nxb, ysplit, rows_per_part, ncob;
    float* stats;                                         // STATS variants: per-workgroup partial sums {sum y, sum y^2} per output channel
};

// four consecutive outputs of one lane: convert and store, optionally on top of what y holds (a later 48-channel block
// of a wider layer)
template <typename T, bool ACC>
__device__ __forceinline__ void store4(T* dst, const float (&v)[4]) {
    u32x2 pk;
    if (ACC) {
        const u32x2 old = *reinterpret_cast<const u32x2*>(dst);
        T o[4];
        memcpy(o, &old, 8);
        pk[0] = pack2<T>(v[0] + to_f32(o[0]), v[1] + to_f32(o[1]));
        pk[1] = pack2<T>(v[2] + to_f32(o[2]), v[3] + to_f32(o[3]));
    } else {
        pk[0] = pack2<T>(v[0], v[1]);
        pk[1] = pack2<T>(v[2], v[3]);
    }
    *reinterpret_cast<u32x2*>(dst) = pk;
}

// Round 5 (WIDE epilogue of the chained kernels): a lane holds four consecutive x of BOTH x tiles of its wave (lane group g: x = 4 g ..
// of tile 0 and of tile 1).  v_permlane16_swap_b32 trades the odd lane groups' tile-0 values for the even groups' tile-1 values:
// afterwards group g holds EIGHT consecutive x - 8 (g >> 1) .. + 7 of tile (g & 1) - one 16-byte store (and one 16-byte read of the
// old values in the accumulate variants) instead of two 8-byte ones; a store instruction then writes 64 contiguous bytes per
// channel row instead of 32.  (Dealing the tile rows in pairs, as the 1x1x1 kernel does, would put fragment rows r and r + 16 of the
// ring on the same banks.)  The exchange runs on the PACKED values (two per tile), the arithmetic stays in the accumulators' layout:
// exchanging fp32 values first kept eight more registers live per co tile and spilled (57 in the 32-wide accumulate + statistics kernel).
// packed values of the two tiles (p0: tile 0, p1: tile 1; two dwords = four x each) -> the lane's eight consecutive x
__device__ __forceinline__ u32x4 pair_swap(const u32x2& p0, const u32x2& p1) {
    const u32x2 s0 = __builtin_amdgcn_permlane16_swap(p0[0], p1[0], false, false);
    const u32x2 s1 = __builtin_amdgcn_permlane16_swap(p0[1], p1[1], false, false);
    const u32x4 r = {s0[0], s1[0], s0[1], s1[1]};
    return r;
}
// ... and back (the exchange is its own inverse): what a 16-byte read of y returned -> the old values in the accumulators' layout
__device__ __forceinline__ void pair_unswap(const u32x4& o, u32x2& p0, u32x2& p1) {
    const u32x2 s0 = __builtin_amdgcn_permlane16_swap(o[0], o[2], false, false);
    const u32x2 s1 = __builtin_amdgcn_permlane16_swap(o[1], o[3], false, false);
    p0[0] = s0[0]; p0[1] = s1[0];
    p1[0] = s0[1]; p1[1] = s1[1];
}
// the wide epilogue of one co tile: (+ old values) (+ statistics) -> 16-byte store
template <typename T, bool ACC, bool STATS>
__device__ __forceinline__ void store_pair(char* dst, const f32x4& a0, const f32x4& a1, const u32x4& old, float& ssum, float& ssq) {
    float v0[4], v1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { v0[q] = a0[q]; v1[q] = a1[q]; }
    if (ACC) {
        u32x2 o0, o1;
        pair_unswap(old, o0, o1);
        T t0[4], t1[4];
        memcpy(t0, &o0, 8);
        memcpy(t1, &o1, 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) { v0[q] += to_f32(t0[q]); v1[q] += to_f32(t1[q]); }
    }
    if (STATS) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { ssum += v0[q]; ssq = fmaf(v0[q], v0[q], ssq); }
#pragma unroll
        for (int q = 0; q < 4; ++q) { ssum += v1[q]; ssq = fmaf(v1[q], v1[q], ssq); }
    }
    u32x2 p0, p1;
    p0[0] = pack2<T>(v0[0], v0[1]); p0[1] = pack2<T>(v0[2], v0[3]);
    p1[0] = pack2<T>(v1[0], v1[1]); p1[1] = pack2<T>(v1[2], v1[3]);
    *reinterpret_cast<u32x4*>(dst) = pair_swap(p0, p1);
}

template <typename T, int NCO>
__global__ void __launch_bounds__(kFwThreads) conv3d_k3_fwd_kernel(ConvFwdDev P) {
    typedef typename Mfma16<T>::v8 frag8;
    __shared__ __attribute__((aligned(16))) T xs[3][4][kFwSlot];
    __shared__ __attribute__((aligned(16))) float red[2][8][2][4][64];      // [buffer][(kz - 1) * 4 + xt][co tile][r][lane]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kz = wave >> 2, xt = wave & 3;
    const int i16 = lane & 15, g = lane >> 4;
    const int cob = blockIdx.y + P.cob0;
    int item = xcd_item(blockIdx.x, gridDim.x);
    const int ypart = item % P.ysplit;  item /= P.ysplit;
    const int xb = item % P.nxb;        item /= P.nxb;
    const int z = item % P.D, b = item / P.D;
    const int y0 = ypart * P.rows_per_part;
    const int y1 = (y0 + P.rows_per_part < P.H) ? y0 + P.rows_per_part : P.H;
    const int x0 = xb * kFwXB;
    const int zz = z + kz - 1;
    const bool plane_ok = zz >= 0 && zz < P.D;            // this wave's tap plane exists (else it contributes zero)

    // ---- stationary weights: B[k][co] fragments of this wave's kz slice -----------------------------------------------
    // chunk c, lane group g: k = 32 c + 8 g .. + 7 inside the slice -> tap (ky, kx) = k / 48, ci0 = k % 48
    frag8 wf[NCO][kFwChunks];
    int32_t aoff[kFwChunks];                              // LDS element offset of the matching A fragment, without the row slot
    int32_t aky[kFwChunks];
#pragma unroll
    for (int c = 0; c < kFwChunks; ++c) {
        const int k = 32 * c + 8 * g;
        const bool live = k < kFwKSlice;
        const int kk = live ? k : 0;
        const int tap = kk / kFwCi, ci0 = kk - tap * kFwCi;
        const int ky = tap / 3, kx = tap - ky * 3;
        aky[c] = ky;
        aoff[c] = (xt * 16 + i16 + kx) * kFwCP + ci0;
#pragma unroll
        for (int t = 0; t < NCO; ++t) {
            const int co = cob * kFwCo + t * 16 + i16;
            const u32x4 w = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(P.wp) + ((int64_t)co * 27 + kz * 9) * kFwCi + kk);
            const u32x4 zero = {0u, 0u, 0u, 0u};
            wf[t][c] = __builtin_bit_cast(frag8, (live && plane_ok) ? w : zero);
        }
    }
    float bias[NCO];
#pragma unroll
    for (int t = 0; t < NCO; ++t) bias[t] = P.bias ? P.bias[cob * kFwCo + t * 16 + i16] : 0.f;

    // ---- copy plan: task = (plane, ci pair, granule), ci pair fastest (adjacent LDS dwords) ---------------------------------
    const int task = tid;                                 // kFwTasks = 720 <= 768 threads: at most one task per thread
    const bool has_task = task < kFwTasks;
    const int tpl = has_task ? task / (24 * kFwGran) : 0;
    const int trem = has_task ? task - tpl * (24 * kFwGran) : 0;
    const int tgr = trem / 24, tcp = trem - tgr * 24;     // granule 0 = left halo (x0-8 .. x0-1), 1..8 data, 9 = right halo
    const int txg = x0 - 8 + 8 * tgr;                     // first x of the granule
    const bool t_inside = txg >= 0 && txg < P.W;          // W % 8 == 0: entirely inside or outside
    const int tzz = z + tpl - 1;
    const bool t_plane = tzz >= 0 && tzz < P.D;
    // channels at or beyond cin (a narrow first layer) are zero: read a valid channel, drop the value
    const bool t_c0 = 2 * tcp < P.cin, t_c1 = 2 * tcp + 1 < P.cin;
    const T* tsrc = reinterpret_cast<const T*>(P.x) + (int64_t)b * P.x_sb + (int64_t)(t_plane ? tzz : z) * P.x_sz +
                         (int64_t)(t_c0 ? 2 * tcp : 0) * P.x_sc + (t_inside ? txg : 0);
    const int64_t tc1 = t_c1 ? P.x_sc : 0;
    // ring positions p = x - (x0 - 1): the granule covers p = 8 tgr - 7 .. 8 tgr; only 0 <= p < kFwXP is stored
    const int tp0 = 8 * tgr - 7;

    auto fetch = [&](u32x4 (&r)[2], int yy) {
        const bool ok = yy >= 0 && yy < P.H;
        const T* s = tsrc + (int64_t)(ok ? yy : 0) * P.x_sy;
        r[0] = *reinterpret_cast<const u32x4*>(s);
        r[1] = *reinterpret_cast<const u32x4*>(s + tc1);
    };
    auto park = [&](const u32x4 (&r)[2], int yy, int slot) {
        if (!has_task) return;
        const bool keep = yy >= 0 && yy < P.H && t_inside && t_plane;
        T* row = &xs[tpl][slot][0];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int p = tp0 + e;
            if (p < 0 || p >= kFwXP) continue;
            // element e of channel 2 tcp (low half) and 2 tcp + 1 (high half)
            const uint32_t a = r[0][e >> 1], c = r[1][e >> 1];
            const uint32_t lo = t_c0 ? ((e & 1) ? (a >> 16) : (a & 0xffffu)) : 0u;
            const uint32_t hi = t_c1 ? ((e & 1) ? (c & 0xffff0000u) : (c << 16)) : 0u;
            *reinterpret_cast<uint32_t*>(row + p * kFwCP + 2 * tcp) = keep ? (lo | hi) : 0u;
        }
    };

    if (y1 > y0) {
        {   // prologue: rows y0 - 1, y0, y0 + 1 of the three planes
            u32x4 r[2];
#pragma unroll
            for (int d = -1; d <= 1; ++d) {
                fetch(r, y0 + d);
                park(r, y0 + d, (y0 + d + 4) & 3);
            }
        }
        __syncthreads();
        for (int y = y0; y < y1; ++y) {
            u32x4 r[2];
            fetch(r, y + 2);                              // in flight during this step's MFMAs
            SEGM_SCHED_FENCE();
            f32x4 acc[NCO];
#pragma unroll
            for (int t = 0; t < NCO; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (plane_ok) {
                const T* pl = &xs[kz][0][0];
#pragma unroll
                for (int c = 0; c < kFwChunks; ++c) {
                    const int slot = (y + aky[c] - 1 + 4) & 3;
                    const frag8 a = *reinterpret_cast<const frag8*>(pl + slot * kFwSlot + aoff[c]);
#pragma unroll
                    for (int t = 0; t < NCO; ++t)
                        acc[t] = Mfma16<T>::run(a, wf[t][c], acc[t]);
                }
            }
            SEGM_SCHED_FENCE();
            const int rb = (y - y0) & 1;
            if (kz > 0) {
#pragma unroll
                for (int t = 0; t < NCO; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) red[rb][(kz - 1) * 4 + xt][t][q][lane] = acc[t][q];
            }
            park(r, y + 2, (y + 2) & 3);
            __syncthreads();
            if (kz == 0) {
                const int xg = x0 + xt * 16 + 4 * g;      // this lane's 4 output positions
                if (xg < P.W) {
#pragma unroll
                    for (int t = 0; t < NCO; ++t) {
                        float v[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            v[q] = acc[t][q] + red[rb][xt][t][q][lane] + red[rb][4 + xt][t][q][lane] + bias[t];
                        const int co = cob * kFwCo + t * 16 + i16;
                        T* dst = reinterpret_cast<T*>(P.y) + (int64_t)b * P.y_sb + (int64_t)co * P.y_sc +
                                      (int64_t)z * P.y_sz + (int64_t)y * P.y_sy + xg;
                        store4<T, false>(dst, v);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Row staging of the two 48-channel kernels below.
// The copy plan of the kernel above keeps per-thread task tables (plane, granule, channel pair) and packs each element
// with shifts, masks and selects: ~850 VALU instructions (86 of them quarter-rate 32 x 32 multiplies) per wave and output
// row in the 48-channel kernel, against 66 MFMAs.  Here one "slot" moves one X row (b, plane, row) into one ring slot
// with four waves, and everything that depends on the row or the plane is wave-uniform (SGPR): waves 0 - 2 of the slot
// own the 192 granule tasks (8 granules x 24 channel pairs: two 16-byte loads at [uniform row base + per-lane 32-bit
// offset], eight v_perm_b32 + ds_write_b32 with immediate offsets), wave 3 owns the 48 halo tasks (columns x0 - 1 and
// x0 + 64: two 2-byte loads, one ds_write_b32).  Per lane: two global byte offsets, one LDS byte offset, three flags.
// The loop body of the 48-channel kernel went from ~1500 to 636 instructions per wave; measured 0.871 -> 0.799 ms at
// 48 -> 48 @128^3 x 2 (profiles/r01_conv_chain_ab.log) - the kernel is bound by instruction issue and LDS / barrier
// latency at two waves per SIMD, not by the MFMA pipe (2 100 of ~6 900 cycles per output row).
// ------------------------------------------------------------------------------------------------------
struct CopyLane {
    uint32_t goff0, goff1;      // byte offsets of channels 2 tcp and 2 tcp + 1 from the row base
    uint32_t loff;              // byte offset of the lane's first ring position inside a ring row
    bool has, live0, live1;     // lane owns a task; its x range is inside the volume and the channel exists
};
struct RowRegs { u32x4 a, c; };  // 8 x values of channels 2 tcp / 2 tcp + 1 (halo wave: one value each, in a[0] / c[0])

template <typename T, int CP = kFwCP>
__device__ __forceinline__ CopyLane copy_lane(const ConvFwdDev& P, int wslot, int lane, int x0) {
    CopyLane L;
    int tcp, x, pos;
    if (wslot < 3) {
        const int j = wslot * 64 + lane, gr = j / 24;
        tcp = j - gr * 24;
        x = x0 + 8 * gr;
        pos = 1 + 8 * gr;
        L.has = true;
    } else {
        // lanes 48 - 63 repeat the tasks of lanes 0 - 15 (same address, same value): no lane-divergent branch around the
        // park, so every path through a step waits for its loads before the output stores are issued
        const int l48 = lane >= 48 ? lane - 48 : lane;
        const int side = l48 >= 24 ? 1 : 0;
        tcp = l48 - side * 24;
        L.has = true;
        x = side ? x0 + kFwXB : x0 - 1;
        pos = side ? kFwXB + 1 : 0;
    }
    const bool inside = L.has && x >= 0 && x < P.W;
    L.live0 = inside && 2 * tcp < P.cin;                  // channels at or beyond cin (a narrow first layer) are zero
    L.live1 = inside && 2 * tcp + 1 < P.cin;
    const int64_t xs = inside ? x : 0;
    L.goff0 = (uint32_t)(((int64_t)(L.live0 ? 2 * tcp : 0) * P.x_sc + xs) * (int64_t)sizeof(T));
    L.goff1 = (uint32_t)(((int64_t)(L.live1 ? 2 * tcp + 1 : 0) * P.x_sc + xs) * (int64_t)sizeof(T));
    L.loff = (uint32_t)((pos * CP + 2 * tcp) * (int)sizeof(T));
    return L;
}
// uniform base of row (b, zz, yy); null when the row lies outside the volume (a zero row)
template <typename T>
__device__ __forceinline__ const char* row_base(const ConvFwdDev& P, int b, int zz, int yy) {
    const bool ok = zz >= 0 && zz < P.D && yy >= 0 && yy < P.H;
    return ok ? P.x + ((int64_t)b * P.x_sb + (int64_t)zz * P.x_sz + (int64_t)yy * P.x_sy) * (int64_t)sizeof(T) : nullptr;
}
template <typename T>
__device__ __forceinline__ void row_fetch(RowRegs& r, const char* rb, const CopyLane& L, bool halo_wave) {
    const u32x4 zero = {0u, 0u, 0u, 0u};
    r.a = zero; r.c = zero;
    if (rb == nullptr) return;                            // uniform
    if (!halo_wave) {
        r.a = *reinterpret_cast<const u32x4*>(rb + L.goff0);
        r.c = *reinterpret_cast<const u32x4*>(rb + L.goff1);
    } else {
        r.a[0] = *reinterpret_cast<const uint16_t*>(rb + L.goff0);
        r.c[0] = *reinterpret_cast<const uint16_t*>(rb + L.goff1);
    }
}
// true when no lane of the wave has to zero what it fetched (interior x range, all 48 input channels): wave-uniform
__device__ __forceinline__ bool park_unmasked(const CopyLane& L) { return __ballot(L.has && !(L.live0 && L.live1)) == 0ull; }

template <typename T, int CP = kFwCP, bool GUARD = false>   // GUARD: granule waves may hold lanes without a task
__device__ __forceinline__ void row_park(const RowRegs& r, char* ring_row, const CopyLane& L, bool halo_wave, bool unmasked = false) {
    char* dst = ring_row + L.loff;
    if (!halo_wave) {
        if (GUARD && !L.has) return;
        u32x4 a = r.a, c = r.c;
        if (!unmasked) {                                  // uniform
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = L.live0 ? a[i] : 0u; c[i] = L.live1 ? c[i] : 0u; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)                       // {channel 2 tcp, channel 2 tcp + 1} at x position e
            *reinterpret_cast<uint32_t*>(dst + e * CP * (int)sizeof(T)) =
                __builtin_amdgcn_perm(c[e >> 1], a[e >> 1], (e & 1) ? 0x07060302u : 0x05040100u);
    } else if (L.has) {
        const uint32_t lo = L.live0 ? r.a[0] : 0u, hi = L.live1 ? r.c[0] : 0u;
        *reinterpret_cast<uint32_t*>(dst) = lo | (hi << 16);
    }
}

// ------------------------------------------------------------------------------------------------------
// 48 output channels per workgroup (cout % 48 == 0: every SegMamba layer this kernel family is picked for).
// With the kz split above, 48 channels are a 32 + 16 pair of workgroups that stage the same X rows twice.  Here the
// reduction index is cut four ways instead - k in (kz, ky, kx, ci) order, 1296 values = 40.5 MFMA chunks, parts of
// 11 / 10 / 10 / 10 chunks - so one wave's stationary weights for THREE co tiles fit in registers (3 x 11 fragments) and
// every A fragment read from LDS feeds 3 MFMAs.  8 waves = 4 K parts x 2 pairs of x tiles; the K parts are added through
// a single 36 KB LDS buffer (two barriers per step: ring + partials ready, partials consumed).
// ------------------------------------------------------------------------------------------------------
constexpr int kF48Waves = 8;
constexpr int kF48Threads = kF48Waves * 64;
constexpr int kF48Chunks = 11;                    // chunks per K part (parts 1..3 use 10; the last chunk of all is half empty)
constexpr int kF48K = 27 * kFwCi;                 // 1296

template <typename T, bool ACC>
__global__ void __launch_bounds__(kF48Threads, 2) conv3d_k3_fwd48_kernel(ConvFwdDev P) {
    typedef typename Mfma16<T>::v8 frag8;
    __shared__ __attribute__((aligned(16))) T xs[3][4][kFwSlot];
    __shared__ __attribute__((aligned(16))) float red[6][3][2][4][64];      // [(part - 1) * 2 + xp][co tile][x tile][r][lane]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int part = wave >> 1, xp = wave & 1;            // K part, pair of x tiles (x0 + 32 xp .. + 31)
    const int i16 = lane & 15, g = lane >> 4;
    // blockIdx.x runs over (item, block of 48 output channels) with the channel block fastest, XCD-aware: the co blocks of an
    // item stage the same input rows and meet in one L2 (as grid rows they ran thousands of workgroups apart)
    const int vid = xcd_item(blockIdx.x, gridDim.x);
    const int cob = vid % P.ncob;
    int item = vid / P.ncob;
    const int ypart = item % P.ysplit;  item /= P.ysplit;
    const int xb = item % P.nxb;        item /= P.nxb;
    const int z = item % P.D, b = item / P.D;
    const int y0 = ypart * P.rows_per_part;
    const int y1 = (y0 + P.rows_per_part < P.H) ? y0 + P.rows_per_part : P.H;
    const int x0 = xb * kFwXB;
    const int c_begin = part == 0 ? 0 : 11 + 10 * (part - 1);     // first chunk of this K part
    const int c_count = part == 0 ? 11 : 10;

    // ---- stationary weights and the matching A fragment offsets ------------------------------------------------------
    frag8 wf[3][kF48Chunks];
    int32_t aoff[kF48Chunks];                             // LDS element offset without the row slot; ky in bits 28..29
#pragma unroll
    for (int c = 0; c < kF48Chunks; ++c) {
        const int k = 32 * (c_begin + c) + 8 * g;
        const bool live = c < c_count && k < kF48K;
        const int kk = live ? k : 0;
        const int tap = kk / kFwCi, ci0 = kk - tap * kFwCi;
        const int tz = tap / 9, ty = (tap - tz * 9) / 3, tx = tap - tz * 9 - ty * 3;
        aoff[c] = (tz * 4 * kFwSlot + (xp * 32 + i16 + tx) * kFwCP + ci0) | (ty << 28);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int co = cob * 48 + t * 16 + i16;
            const u32x4 w = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(P.wp) + (int64_t)co * kF48K + kk);
            const u32x4 zero = {0u, 0u, 0u, 0u};
            wf[t][c] = __builtin_bit_cast(frag8, live ? w : zero);
        }
    }
    float bias[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) bias[t] = P.bias ? P.bias[cob * 48 + t * 16 + i16] : 0.f;

    // ---- row staging: slot A = plane 0 (waves 0 - 3) and plane 1 (waves 4 - 7), slot B = plane 2 (waves 0 - 3) -----------------
    const int wslot = wave & 3;
    const bool halo_wave = wslot == 3, second = wave < 4;
    const int plane_a = wave >> 2;
    const CopyLane cl = copy_lane<T>(P, wslot, lane, x0);
    char* ring = reinterpret_cast<char*>(&xs[0][0][0]);
    auto fetch = [&](RowRegs (&r)[2], int yy) {
        row_fetch<T>(r[0], row_base<T>(P, b, z + plane_a - 1, yy), cl, halo_wave);
        if (second) row_fetch<T>(r[1], row_base<T>(P, b, z + 1, yy), cl, halo_wave);
    };
    auto park = [&](const RowRegs (&r)[2], int yy) {
        const int slot = (yy + 4) & 3;
        row_park<T>(r[0], ring + (plane_a * 4 + slot) * kFwSlot * (int)sizeof(T), cl, halo_wave);
        if (second) row_park<T>(r[1], ring + (2 * 4 + slot) * kFwSlot * (int)sizeof(T), cl, halo_wave);
    };

    if (y1 > y0) {
        {
            RowRegs r[2];
#pragma unroll
            for (int d = -1; d <= 1; ++d) {
                fetch(r, y0 + d);
                park(r, y0 + d);
            }
        }
        __syncthreads();
        for (int y = y0; y < y1; ++y) {
            RowRegs r[2];
            fetch(r, y + 2);                              // in flight during this step's MFMAs
            SEGM_SCHED_FENCE();
            f32x4 acc[3][2];
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
            const T* pl = &xs[0][0][0];
#pragma unroll
            for (int c = 0; c < kF48Chunks; ++c) {
                const int slot = (y + (aoff[c] >> 28) + 3) & 3;
                const T* ap = pl + slot * kFwSlot + (aoff[c] & 0x0fffffff);
                const frag8 a0 = *reinterpret_cast<const frag8*>(ap);
                const frag8 a1 = *reinterpret_cast<const frag8*>(ap + 16 * kFwCP);       // the pair's second x tile
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    acc[t][0] = Mfma16<T>::run(a0, wf[t][c], acc[t][0]);
                    acc[t][1] = Mfma16<T>::run(a1, wf[t][c], acc[t][1]);
                }
            }
            SEGM_SCHED_FENCE();
            if (part > 0) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int q = 0; q < 4; ++q) red[(part - 1) * 2 + xp][t][u][q][lane] = acc[t][u][q];
            }
            park(r, y + 2);
            __syncthreads();                              // ring row y + 2 and the partial sums are in LDS
            if (part == 0) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int xg = x0 + xp * 32 + u * 16 + 4 * g;      // this lane's 4 output positions
                    if (xg >= P.W) continue;
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        float v[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            v[q] = acc[t][u][q] + red[xp][t][u][q][lane] + red[2 + xp][t][u][q][lane] +
                                   red[4 + xp][t][u][q][lane] + bias[t];
                        const int co = cob * 48 + t * 16 + i16;
                        T* dst = reinterpret_cast<T*>(P.y) + (int64_t)b * P.y_sb + (int64_t)co * P.y_sc +
                                      (int64_t)z * P.y_sz + (int64_t)y * P.y_sy + xg;
                        store4<T, ACC>(dst, v);
                    }
                }
            }
            __syncthreads();                              // the partial sums are consumed: the buffer may be rewritten
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// 48 output channels per workgroup, K parts chained in time (SEGM_CONV_FWD_CHAIN).
// The kernel above adds its four K parts through LDS once per output row: six waves write, two waves read, sum,
// convert and store while the other six wait at the barrier (two barriers per row).  Here the same four K parts form a
// pipeline instead:
//   * at step s part p works on output row s - p: it starts from the partial sums part p - 1 left for that row one step
//     earlier (double-buffered LDS hand-off, float4 per lane and tile), adds its own chunks and either hands the
//     tiles on or - part 3 - converts and stores (part 0 starts from the bias).  All waves do the same MFMA work between
//     two barriers, one barrier per step, three drain steps per work item;
//   * the skew makes each tap plane's ring run at its own row offset (the K parts are in (kz, ky, kx, ci) order:
//     plane 0 is read by parts 0 - 1, plane 1 by parts 1 - 2, plane 2 by parts 2 - 3): during step s the incoming rows
//     are s + 2 (plane 0), s (plane 1) and s - 1 (plane 2); four slots per plane as before;
//   * the A fragments of the next chunk are read while the current chunk's MFMAs run.
// LDS: 88.7 KB ring + 2 x 36.9 KB hand-off = 162 432 B of the 163 840 B a workgroup may declare.
// Measured (profiles/r01_conv_chain_ab.log): 0.70 ms against 0.83 ms at 48 -> 48 @128^3 x 2, ahead on the 128^3 / 64^3
// layers, 4 % behind at 32^3 (drain steps); the dispatcher times both per shape.  A first version with ONE wave per SIMD
// (4 waves x 4 x tiles, 512 registers each) lost - 0.94 ms: nothing overlapped that wave's ~680 non-MFMA instructions
// per step with its 132 MFMAs.
// ------------------------------------------------------------------------------------------------------
// Eight waves = 4 K parts x 2 x pairs (two per SIMD, the layout of the kernel above with the hand-off pipeline in place of its
// reduce + epilogue phase).  CP = ci pitch of a ring position in elements: 56 as above, or 48 (no padding) - by the LDS
// bank model of MI355X_MICROARCH.md (ds_read_b128: four groups of 16 lanes over 64 banks) the unpadded rows make the A
// fragment reads conflict-free where the padded ones are 2-way (tools/lds_conflicts.py); SEGM_CONV_FWD_PITCH48.
// VAR: bit 0 - K parts 1 - 3 skip their eleventh (all-zero) chunk instead of running it (41 chunks of 32 split 11 + 10 + 10 + 10;
// the loop is unrolled over 11); bits 1 - 2 - the A fragments are read PF = 1 + that many chunks ahead of their MFMAs.  Shipped:
// VAR 3 (skip, two chunks ahead) on the unpadded-pitch kernel and the 32-wide one; SEGM_CONV_CHAIN_VAR=0 launches the round-3
// schedule (VAR 0).  Measured in round 4 (profiles/r04_conv_chain_var_time.log): 48 -> 48 @128^3 0.626 -> 0.588 ms, the 32-wide
// kernel 0.776 -> 0.680 ms there and 0.369 -> 0.355 ms at 96 -> 96 @64^3; three chunks ahead spills (256 registers at two waves
// per SIMD) and loses; whole step 61.97 -> 61.6 ms.  The 64-wide kernel ships VAR 11: bit 3 instantiates its row loop once per K
// part (profiles/r04_conv_chain_perpart.log: 96 -> 96 @64^3 0.345 -> 0.327 ms, 192 -> 192 @32^3 0.377 -> 0.364, 48 -> 48 @128^3 equal).  `s_setprio 1` for the second-dispatched half of the workgroup (the guide's
// two-waves-per-SIMD section, item 4) measured nothing here (0.605 / 0.612 vs 0.618 / 0.631 ms, profiles/r04_call9_ab.log).
// STATS (round 5): the storing K part also sums y and y^2 of what it stores (the fp32 values, before they are
// rounded to 16 bits), per output channel, and writes one {count, sum, sum of squares} partial per workgroup and x pair; the
// InstanceNorm behind this convolution merges D x H / rows x W / 32 partials per instance (Chan's update, segm_instnorm_fwd's
// `stats_partials`) instead of reading the volume once more.  MEASURED price (profiles/r05_inorm_epilogue.log): none - 48 -> 48
// @128^3 0.569 vs 0.546 ms, 96 -> 48 @128^3 1.225 vs 1.241 ms with / without; the statistics pass it replaces: ~0.08 ms per 128^3 layer.
template <typename T, bool ACC, int CP, int VAR = 0, bool STATS = false, bool WIDE = false>
__global__ void __launch_bounds__(512, 2) conv3d_k3_fwd48_chain_kernel(ConvFwdDev P) {
    typedef typename Mfma16<T>::v8 frag8;
    constexpr bool SKIP = (VAR & 1) != 0;
    constexpr int PF = 1 + ((VAR >> 1) & 3);
    constexpr int XT = 2;                                 // x tiles per wave
    constexpr int XP = 2;                                 // waves per K part
    constexpr int kSlot = kFwXP * CP;                     // elements per ring row
    __shared__ __attribute__((aligned(16))) T xs[3][4][kSlot];
    __shared__ __attribute__((aligned(16))) f32x4 hand[2][3][XP][3 * XT][64];     // [buffer][link p -> p + 1][x part][co tile * XT + x tile][lane]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int part = wave / XP, xp = wave % XP;
    const int i16 = lane & 15, g = lane >> 4;
    // blockIdx.x runs over (item, block of 48 output channels) with the channel block fastest, XCD-aware: the co blocks of an
    // item stage the same input rows and meet in one L2 (as grid rows they ran thousands of workgroups apart)
    const int vid = xcd_item(blockIdx.x, gridDim.x);
    const int cob = vid % P.ncob;
    int item = vid / P.ncob;
    const int ypart = item % P.ysplit;  item /= P.ysplit;
    const int xb = item % P.nxb;        item /= P.nxb;
    const int z = item % P.D, b = item / P.D;
    const int y0 = ypart * P.rows_per_part;
    const int y1 = (y0 + P.rows_per_part < P.H) ? y0 + P.rows_per_part : P.H;
    const int x0 = xb * kFwXB;
    const int c_begin = part == 0 ? 0 : 11 + 10 * (part - 1);     // first chunk of this K part
    const int c_count = part == 0 ? 11 : 10;

    // ---- stationary weights and the matching A fragment offsets (the wave's first x tile) ------------------------------------
    frag8 wf[3][kF48Chunks];
    int32_t aoff[kF48Chunks];                             // LDS element offset without the row slot; ky in bits 28..29
#pragma unroll
    for (int c = 0; c < kF48Chunks; ++c) {
        const int k = 32 * (c_begin + c) + 8 * g;
        const bool live = c < c_count && k < kF48K;
        // a dead chunk (zero weights) still issues its LDS reads: aim them at rows this part reads anyway, not at tap 0 of
        // plane 0, whose slot another wave may be refilling in this step
        const int kk = live ? k : 32 * c_begin + 8 * g;
        const int tap = kk / kFwCi, ci0 = kk - tap * kFwCi;
        const int tz = tap / 9, ty = (tap - tz * 9) / 3, tx = tap - tz * 9 - ty * 3;
        aoff[c] = (tz * 4 * kSlot + (xp * XT * 16 + i16 + tx) * CP + ci0) | (ty << 28);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int co = cob * 48 + t * 16 + i16;
            const u32x4 w = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(P.wp) + (int64_t)co * kF48K + kk);
            const u32x4 zero = {0u, 0u, 0u, 0u};
            wf[t][c] = __builtin_bit_cast(frag8, live ? w : zero);
        }
    }
    float bias[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) bias[t] = P.bias ? P.bias[cob * 48 + t * 16 + i16] : 0.f;

    // ---- row staging: plane q's incoming row during step s is s + {2, 0, -1}[q]: slot A = plane 0 (waves 0 - 3) / plane 1
    //      (waves 4 - 7), slot B = plane 2 (waves 0 - 3) -------------------------------------------------------------------------
    // the halo task goes to waves 0 and 4 (K parts 0 and 2), never to the storing part: a halo wave's fetch is followed by
    // a wait that would also cover the output stores it issued one step earlier
    const int wslot = ((wave & 3) + 3) & 3;
    const bool halo_wave = wslot == 3, second = wave < 4;
    const int plane_a = wave >> 2, soff_a = plane_a == 0 ? 2 : 0;
    const CopyLane cl = copy_lane<T, CP>(P, wslot, lane, x0);
    char* ring = reinterpret_cast<char*>(&xs[0][0][0]);
    auto fetch = [&](RowRegs (&r)[2], int base, bool skewed) {
        row_fetch<T>(r[0], row_base<T>(P, b, z + plane_a - 1, base + (skewed ? soff_a : 0)), cl, halo_wave);
        if (second) row_fetch<T>(r[1], row_base<T>(P, b, z + 1, base + (skewed ? -1 : 0)), cl, halo_wave);
    };
    auto park = [&](const RowRegs (&r)[2], int base, bool skewed) {
        const int ya = base + (skewed ? soff_a : 0), yb = base + (skewed ? -1 : 0);
        row_park<T, CP>(r[0], ring + (plane_a * 4 + ((ya + 8) & 3)) * kSlot * (int)sizeof(T), cl, halo_wave);
        if (second) row_park<T, CP>(r[1], ring + (2 * 4 + ((yb + 8) & 3)) * kSlot * (int)sizeof(T), cl, halo_wave);
    };

    if (y1 <= y0) {                                       // an empty y part (ysplit does not divide H): nothing to compute, but its
        if (P.stats && part == 3 && g == 0) {             // statistics slot must hold count 0 - the merge folds every slot
            const int nparts = P.D * P.ysplit * P.nxb * XP, pid = ((z * P.ysplit + ypart) * P.nxb + xb) * XP + xp;
#pragma unroll
            for (int t = 0; t < 3; ++t)
                reinterpret_cast<float4*>(P.stats)[((int64_t)b * P.cout + cob * 48 + t * 16 + i16) * nparts + pid] = float4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    {   // prologue: rows y0 - 1, y0, y0 + 1 of every plane (planes 1 and 2 re-park theirs on schedule; same slot, same data)
        RowRegs r[2];
#pragma unroll
        for (int d = -1; d <= 1; ++d) {
            fetch(r, y0 + d, false);
            park(r, y0 + d, false);
        }
    }
    __syncthreads();
    // output addresses = uniform base (row, co tile, x tile: scalar arithmetic) + one 32-bit lane offset (co and x within the tile;
    // the host checks that 16 channel strides fit 32 bits)
    const uint32_t ylane = (uint32_t)(((int64_t)i16 * P.y_sc + 4 * g) * (int64_t)sizeof(T));
    // WIDE: after pair_swap this lane stores x = 8 (g >> 1) .. + 7 of x tile (g & 1), relative to ybase(row, t, 0)
    const uint32_t ylane16 = (uint32_t)(((int64_t)i16 * P.y_sc + 16 * (g & 1) + 8 * (g >> 1)) * (int64_t)sizeof(T));
    auto ybase = [&](int row, int t, int u) {
        return P.y + ((int64_t)b * P.y_sb + (int64_t)(cob * 48 + t * 16) * P.y_sc + (int64_t)z * P.y_sz + (int64_t)row * P.y_sy + x0 +
                      (xp * XT + u) * 16) * (int64_t)sizeof(T);
    };
    int32_t aoffs[kF48Chunks];                            // aoff without the ky bits
#pragma unroll
    for (int c = 0; c < kF48Chunks; ++c) aoffs[c] = aoff[c] & 0x0fffffff;
    auto chunk_begin = [](int pt) { return pt == 0 ? 0 : 11 + 10 * (pt - 1); };
    auto chunk_count = [](int pt) { return pt == 0 ? 11 : 10; };
    // VAR bit 3: the row loop is instantiated once per K part (PART >= 0) and entered through a switch - which part starts from the
    // bias, reads / writes hand-off tiles, stores, stages which planes and has an eleventh chunk are then compile-time facts
    // instead of scalar branches in every step (profiles/r04_conv_pmc.log: 1.3 scalar instructions per MFMA)
    auto steps = [&](auto pc) {
    constexpr int PART = decltype(pc)::value;
    const int prt = PART >= 0 ? PART : part;
    const bool second_ = PART >= 0 ? PART < 2 : second;
    const int plane_a_ = PART >= 0 ? (PART >> 1) : plane_a, soff_a_ = plane_a_ == 0 ? 2 : 0;
    const bool halo_ = PART >= 0 ? ((PART & 1) == 0 && xp == 0) : halo_wave;
    auto fetch_s = [&](RowRegs (&r)[2], int base) {
        row_fetch<T>(r[0], row_base<T>(P, b, z + plane_a_ - 1, base + soff_a_), cl, halo_);
        if (second_) row_fetch<T>(r[1], row_base<T>(P, b, z + 1, base - 1), cl, halo_);
    };
    const bool unm = (VAR & 8) != 0 && park_unmasked(cl);
    auto park_s = [&](const RowRegs (&r)[2], int base) {
        const int ya = base + soff_a_, yb = base - 1;
        row_park<T, CP>(r[0], ring + (plane_a_ * 4 + ((ya + 8) & 3)) * kSlot * (int)sizeof(T), cl, halo_, unm);
        if (second_) row_park<T, CP>(r[1], ring + (2 * 4 + ((yb + 8) & 3)) * kSlot * (int)sizeof(T), cl, halo_, unm);
    };
    float st_s[3] = {0.f, 0.f, 0.f}, st_q[3] = {0.f, 0.f, 0.f};       // STATS: this lane's sums over its co of every tile
    for (int s = y0; s < y1 + 3; ++s) {
        const int row = s - prt;                          // this part's output row
        const bool active = row >= y0 && row < y1;
        // the partial sums of the previous K part for this row, then the first chunk's A fragments: issued before the
        // global fetches so that their latency is covered by the address arithmetic
        f32x4 acc[3][XT];
        frag8 a[PF + 1][XT];
        const T* pl = &xs[0][0][0];
        auto live = [&](int c) { return !SKIP || c + 1 < kF48Chunks || prt == 0; };
        auto load_a = [&](frag8 (&dst)[XT], int c) {
            const T* ap;
            if constexpr (PART >= 0) {
                // Round 5: with the K part a compile-time fact the tap row ky of every lane group of every chunk is one too (once
                // the chunk loop is unrolled): 37 of the 41 chunks lie inside one ky - the ring slot is then SCALAR arithmetic
                // shared by the chunks of that ky - and the rest split at a known lane group.  Per chunk: one v_add (two selects
                // more in a split chunk) where the per-lane form below spends five vector instructions (shift, add, and,
                // multiply-add, and): ~45 of the ~125 vector instructions of a step.
                const int cb = chunk_begin(PART);
                int ty[4];
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    const int k = 32 * (cb + c) + 8 * gg;
                    const bool lv = c < chunk_count(PART) && k < kF48K;
                    const int tap = (lv ? k : 32 * cb + 8 * gg) / kFwCi;
                    ty[gg] = (tap % 9) / 3;
                }
                int off = ((row + ty[0] + 7) & 3) * kSlot;                 // scalar
#pragma unroll
                for (int gg = 1; gg < 4; ++gg)
                    if (ty[gg] != ty[gg - 1]) off = g >= gg ? ((row + ty[gg] + 7) & 3) * kSlot : off;
                ap = pl + off + aoffs[c];
            } else {
                const int slot = (row + (aoff[c] >> 28) + 7) & 3;          // input row = row + ky - 1
                ap = pl + slot * kSlot + (aoff[c] & 0x0fffffff);
            }
#pragma unroll
            for (int u = 0; u < XT; ++u) dst[u] = *reinterpret_cast<const frag8*>(ap + u * 16 * CP);
        };
        if (active) {
            if (prt == 0) {                               // the chain starts from the bias (a lane's four results share a co)
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int u = 0; u < XT; ++u) acc[t][u] = f32x4{bias[t], bias[t], bias[t], bias[t]};
            } else {
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int u = 0; u < XT; ++u) acc[t][u] = hand[(s + 1) & 1][prt - 1][xp][t * XT + u][lane];
            }
#pragma unroll
            for (int c = 0; c < PF; ++c) load_a(a[c], c);
        }
        RowRegs r[2];
        fetch_s(r, s);                                    // in flight during this step's MFMAs
        // accumulate variant: what y holds for this row is fetched now and added after the MFMAs (see the 32-wide kernel)
        u32x2 oldy[XT][3];
        u32x4 oldw[3];
        if (ACC && WIDE && active && prt == 3) {
#pragma unroll
            for (int t = 0; t < 3; ++t) oldw[t] = *reinterpret_cast<const u32x4*>(ybase(row, t, 0) + ylane16);
        }
        if (ACC && !WIDE && active && prt == 3) {
#pragma unroll
            for (int u = 0; u < XT; ++u) {
                const int xg = x0 + (xp * XT + u) * 16 + 4 * g;
                // lanes beyond W neither load nor store (a tile beyond W starts past the row's end); a block inside W: no lane mask
                if (x0 + kFwXB <= P.W || xg < P.W) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) oldy[u][t] = *reinterpret_cast<const u32x2*>(ybase(row, t, u) + ylane);
                }
            }
        }
        SEGM_SCHED_FENCE();
        if (active) {
#pragma unroll
            for (int c = 0; c < kF48Chunks; ++c) {
                if (c + PF < kF48Chunks && live(c + PF)) load_a(a[(c + PF) % (PF + 1)], c + PF);     // PF chunks (6 MFMAs each) ahead of its use
                SEGM_SCHED_FENCE();
                if (live(c)) {
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int u = 0; u < XT; ++u) acc[t][u] = Mfma16<T>::run(a[c % (PF + 1)][u], wf[t][c], acc[t][u]);
                }
                SEGM_SCHED_FENCE();
            }
            if (prt < 3) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int u = 0; u < XT; ++u) hand[s & 1][prt][xp][t * XT + u][lane] = acc[t][u];
            }
        }
        SEGM_SCHED_FENCE();
        park_s(r, s);
        // the last K part stores AFTER the rows are parked: vmcnt counts loads and stores in order, so a park behind the stores
        // would wait for their write acknowledgements (a memory round trip on the critical path of every step); here the
        // stores drain during the next step
        if (WIDE && active && prt == 3) {                 // (the launcher: W a multiple of the 64-wide block, nothing ragged)
            static_assert(!WIDE || XT == 2, "the wide epilogue pairs the wave's two x tiles");
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if constexpr (STATS) store_pair<T, ACC, true>(ybase(row, t, 0) + ylane16, acc[t][0], acc[t][1], oldw[t], st_s[t], st_q[t]);
                else { float d0 = 0.f, d1 = 0.f; store_pair<T, ACC, false>(ybase(row, t, 0) + ylane16, acc[t][0], acc[t][1], oldw[t], d0, d1); }
            }
        }
        if (!WIDE && active && prt == 3) {
#pragma unroll
            for (int u = 0; u < XT; ++u) {
                const int xg = x0 + (xp * XT + u) * 16 + 4 * g;            // this lane's 4 output positions
                if (xg >= P.W) continue;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = acc[t][u][q];
                    T* dst = reinterpret_cast<T*>(ybase(row, t, u) + ylane);
                    if (ACC) {
                        T o[4];
                        memcpy(o, &oldy[u][t], 8);
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] += to_f32(o[q]);
                    }
                    if constexpr (STATS) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) { st_s[t] += v[q]; st_q[t] = fmaf(v[q], v[q], st_q[t]); }
                    }
                    store4<T, false>(dst, v);
                }
            }
        }
        __syncthreads();                                  // incoming rows and the hand-off tiles are in LDS
    }
    if constexpr (STATS) {
        if (prt == 3) {                                   // sum over the four lane groups (x within the tiles), lane group 0 writes
            const int nparts = P.D * P.ysplit * P.nxb * XP, pid = ((z * P.ysplit + ypart) * P.nxb + xb) * XP + xp;
            const int xlo = x0 + xp * XT * 16;
            const int nx = P.W - xlo < XT * 16 ? (P.W - xlo > 0 ? P.W - xlo : 0) : XT * 16;
            const float cnt = (float)((y1 - y0) * nx);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                float a_ = st_s[t], q_ = st_q[t];
                a_ += __shfl_xor(a_, 16, 64); q_ += __shfl_xor(q_, 16, 64);
                a_ += __shfl_xor(a_, 32, 64); q_ += __shfl_xor(q_, 32, 64);
                if (g == 0)
                    reinterpret_cast<float4*>(P.stats)[((int64_t)b * P.cout + cob * 48 + t * 16 + i16) * nparts + pid] = float4{cnt, a_, q_, 0.f};
            }
        }
    }
    };
    if constexpr ((VAR & 8) != 0) {
        switch (part) {
            case 0: steps(std::integral_constant<int, 0>{}); break;
            case 1: steps(std::integral_constant<int, 1>{}); break;
            case 2: steps(std::integral_constant<int, 2>{}); break;
            default: steps(std::integral_constant<int, 3>{}); break;
        }
    } else {
        steps(std::integral_constant<int, -1>{});
    }
}

// ------------------------------------------------------------------------------------------------------
// The chained kernel on 32-wide x blocks (SEGM_CONV_FWD_CHAIN32): four waves = the four K parts, one pair of x tiles, unpadded
// LDS rows.  Ring 12 x 34 x 48 x 2 B = 39.2 KB + hand-off 36.9 KB = 76 KB, so TWO workgroups share a CU: the same eight
// waves per CU as above, but in two independent barrier domains - one workgroup's hand-off / staging / barrier phases can
// overlap the other's MFMA phase instead of all eight waves walking through the phases in lockstep.  (Written after the GPU
// budget of round 1 was spent: parity-tested on the emulator, not yet timed; the dispatcher times it per shape.)
// Staging per step: wave p < 3 moves plane p's 96 granule tasks (two slots: 64 + 32 lanes), wave 3 - the storing wave - the
// 48 halo tasks of each of the three planes.
// ------------------------------------------------------------------------------------------------------
constexpr int kC32XB = 32;                        // x positions per workgroup
constexpr int kC32XP = kC32XB + 2;                // ring positions per row
constexpr int kC32CP = 48;                        // ci pitch (elements): unpadded

template <typename T>
__device__ __forceinline__ CopyLane copy_lane32(const ConvFwdDev& P, bool halo, int j, int x0) {
    CopyLane L;
    int tcp, x, pos;
    if (!halo) {
        const int gr = j / 24;
        tcp = j - gr * 24;
        L.has = j < 4 * 24;
        if (!L.has) tcp = 0;
        x = x0 + 8 * (L.has ? gr : 0);
        pos = 1 + 8 * (L.has ? gr : 0);
    } else {
        const int side = j >= 24 ? 1 : 0;
        tcp = j - side * 24;
        L.has = j < 48;
        if (!L.has) tcp = 0;
        x = side ? x0 + kC32XB : x0 - 1;
        pos = side ? kC32XB + 1 : 0;
    }
    const bool inside = L.has && x >= 0 && x < P.W;
    L.live0 = inside && 2 * tcp < P.cin;
    L.live1 = inside && 2 * tcp + 1 < P.cin;
    const int64_t xs = inside ? x : 0;
    L.goff0 = (uint32_t)(((int64_t)(L.live0 ? 2 * tcp : 0) * P.x_sc + xs) * (int64_t)sizeof(T));
    L.goff1 = (uint32_t)(((int64_t)(L.live1 ? 2 * tcp + 1 : 0) * P.x_sc + xs) * (int64_t)sizeof(T));
    L.loff = (uint32_t)((pos * kC32CP + 2 * tcp) * (int)sizeof(T));
    return L;
}

template <typename T, bool ACC, int VAR = 0, bool STATS = false, bool WIDE = false>
__global__ void __launch_bounds__(256, 2) conv3d_k3_fwd48_chain32_kernel(ConvFwdDev P) {
    typedef typename Mfma16<T>::v8 frag8;
    constexpr bool SKIP = (VAR & 1) != 0;
    constexpr int PF = 1 + ((VAR >> 1) & 3);
    constexpr int XT = 2, CP = kC32CP;
    constexpr int kSlot = kC32XP * CP;                    // elements per ring row
    __shared__ __attribute__((aligned(16))) T xs[3][4][kSlot];
    __shared__ __attribute__((aligned(16))) f32x4 hand[2][3][3 * XT][64];         // [buffer][link p -> p + 1][co tile * XT + x tile][lane]
    const int tid = threadIdx.x, lane = tid & 63;
    const int part = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    // blockIdx.x runs over (item, block of 48 output channels) with the channel block fastest, XCD-aware: the co blocks of an
    // item stage the same input rows and meet in one L2 (as grid rows they ran thousands of workgroups apart)
    const int vid = xcd_item(blockIdx.x, gridDim.x);
    const int cob = vid % P.ncob;
    int item = vid / P.ncob;
    const int ypart = item % P.ysplit;  item /= P.ysplit;
    const int xb = item % P.nxb;        item /= P.nxb;
    const int z = item % P.D, b = item / P.D;
    const int y0 = ypart * P.rows_per_part;
    const int y1 = (y0 + P.rows_per_part < P.H) ? y0 + P.rows_per_part : P.H;
    const int x0 = xb * kC32XB;
    const int c_begin = 10 * part;                                // first chunk of this K part: 41 chunks = 10 + 10 + 10 + 11 - the eleven
    const int c_count = part == 3 ? 11 : 10;                      // go to the wave that stages nothing (registers: 120 + staging vs 132)

    // ---- stationary weights and the matching A fragment offsets (x tile 0) ---------------------------------------------------
    frag8 wf[3][kF48Chunks];
    int32_t aoff[kF48Chunks];                             // LDS element offset without the row slot; ky in bits 28..29
#pragma unroll
    for (int c = 0; c < kF48Chunks; ++c) {
        const int k = 32 * (c_begin + c) + 8 * g;
        const bool live = c < c_count && k < kF48K;
        // a dead chunk (zero weights) still issues its LDS reads: aim them at rows this part reads anyway, not at tap 0 of
        // plane 0, whose slot another wave may be refilling in this step
        const int kk = live ? k : 32 * c_begin + 8 * g;
        const int tap = kk / kFwCi, ci0 = kk - tap * kFwCi;
        const int tz = tap / 9, ty = (tap - tz * 9) / 3, tx = tap - tz * 9 - ty * 3;
        aoff[c] = (tz * 4 * kSlot + (i16 + tx) * CP + ci0) | (ty << 28);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int co = cob * 48 + t * 16 + i16;
            const u32x4 w = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(P.wp) + (int64_t)co * kF48K + kk);
            const u32x4 zero = {0u, 0u, 0u, 0u};
            wf[t][c] = __builtin_bit_cast(frag8, live ? w : zero);
        }
    }
    float bias[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) bias[t] = P.bias ? P.bias[cob * 48 + t * 16 + i16] : 0.f;

    // ---- row staging: plane q's incoming row during step s is s + {2, 0, -1}[q] -----------------------------------------------
    // waves 0 - 2: slots 0 / 1 = granule tasks lane / 64 + lane of plane `part`; wave 3: slots 0 - 2 = halo tasks of planes 0 - 2
    // waves 0 - 2 stage plane `part` completely: slots 0 / 1 = its 96 granule tasks (lane, 64 + lane), slot 2 = its 48 halo tasks.
    // Wave 3 - the storing wave - stages nothing: in round 3's plan it carried the halo tasks of all three planes, and its loads sat
    // behind `s_waitcnt vmcnt(0)` for the output stores of the step before (their source registers are reused: the compiler waits for
    // the stores to complete) - a memory round trip at the head of every step of the wave the chain ends in.
    const bool stager = part < 3;
    const CopyLane cl0 = copy_lane32<T>(P, false, lane, x0);
    const CopyLane cl1 = copy_lane32<T>(P, false, 64 + lane, x0);
    const CopyLane clh = copy_lane32<T>(P, true, lane, x0);
    char* ring = reinterpret_cast<char*>(&xs[0][0][0]);
    auto srow = [&](int pl, int base, bool skewed) { return base + (skewed ? (pl == 0 ? 2 : (pl == 1 ? 0 : -1)) : 0); };
    auto fetch = [&](RowRegs (&r)[3], int base, bool skewed) {
        if (stager) {
            const char* rb = row_base<T>(P, b, z + part - 1, srow(part, base, skewed));
            row_fetch<T>(r[0], rb, cl0, false);
            row_fetch<T>(r[1], rb, cl1, false);
            row_fetch<T>(r[2], rb, clh, true);
        }
    };
    auto park = [&](const RowRegs (&r)[3], int base, bool skewed) {
        if (stager) {
            char* row = ring + (part * 4 + ((srow(part, base, skewed) + 8) & 3)) * kSlot * (int)sizeof(T);
            row_park<T, CP, true>(r[0], row, cl0, false);
            row_park<T, CP, true>(r[1], row, cl1, false);
            row_park<T, CP>(r[2], row, clh, true);
        }
    };

    if (y1 <= y0) {                                       // an empty y part: its statistics slot must hold count 0 (see the 64-wide kernel)
        if (P.stats && part == 3 && g == 0) {
            const int nparts = P.D * P.ysplit * P.nxb, pid = (z * P.ysplit + ypart) * P.nxb + xb;
#pragma unroll
            for (int t = 0; t < 3; ++t)
                reinterpret_cast<float4*>(P.stats)[((int64_t)b * P.cout + cob * 48 + t * 16 + i16) * nparts + pid] = float4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    {   // prologue: rows y0 - 1, y0, y0 + 1 of every plane (planes 1 and 2 re-park theirs on schedule; same slot, same data)
        RowRegs r[3];
#pragma unroll
        for (int d = -1; d <= 1; ++d) {
            fetch(r, y0 + d, false);
            park(r, y0 + d, false);
        }
    }
    __syncthreads();
    // output addresses = uniform base (row, co tile, x tile: scalar arithmetic) + one 32-bit lane offset (co within the tile, x
    // within the tile; the host checks that 16 channel strides fit 32 bits).  (As a buffer resource + scalar offset the accumulate
    // variant's last K part - eleven chunks of weights, the old outputs in flight - spilled 22 registers; this form spills 3.)
    const uint32_t ylane = (uint32_t)(((int64_t)i16 * P.y_sc + 4 * g) * (int64_t)sizeof(T));
    // WIDE: after pair_swap this lane stores x = 8 (g >> 1) .. + 7 of x tile (g & 1), relative to ybase(row, t, 0)
    const uint32_t ylane16 = (uint32_t)(((int64_t)i16 * P.y_sc + 16 * (g & 1) + 8 * (g >> 1)) * (int64_t)sizeof(T));
    auto ybase = [&](int row, int t, int u) {
        return P.y + ((int64_t)b * P.y_sb + (int64_t)(cob * 48 + t * 16) * P.y_sc + (int64_t)z * P.y_sz + (int64_t)row * P.y_sy + x0 + u * 16) *
                         (int64_t)sizeof(T);
    };
    int32_t aoffs[kF48Chunks];                            // aoff without the ky bits
#pragma unroll
    for (int c = 0; c < kF48Chunks; ++c) aoffs[c] = aoff[c] & 0x0fffffff;
    auto chunk_begin = [](int pt) { return 10 * pt; };
    auto chunk_count = [](int pt) { return pt == 3 ? 11 : 10; };
    // VAR bit 3: the row loop once per K part, entered through a switch (see the 64-wide kernel)
    auto steps = [&](auto pc) {
    constexpr int PART = decltype(pc)::value;
    const int prt = PART >= 0 ? PART : part;
    const bool stager_ = PART >= 0 ? PART < 3 : stager;
    const bool unm0 = (VAR & 8) != 0 && park_unmasked(cl0), unm1 = (VAR & 8) != 0 && park_unmasked(cl1);
    auto fetch_s = [&](RowRegs (&r)[3], int base) {
        if (stager_) {
            const char* rb = row_base<T>(P, b, z + prt - 1, srow(prt, base, true));
            row_fetch<T>(r[0], rb, cl0, false);
            row_fetch<T>(r[1], rb, cl1, false);
            row_fetch<T>(r[2], rb, clh, true);
        }
    };
    auto park_s = [&](const RowRegs (&r)[3], int base) {
        if (stager_) {
            char* row = ring + (prt * 4 + ((srow(prt, base, true) + 8) & 3)) * kSlot * (int)sizeof(T);
            row_park<T, CP, true>(r[0], row, cl0, false, unm0);
            row_park<T, CP, true>(r[1], row, cl1, false, unm1);
            row_park<T, CP>(r[2], row, clh, true);
        }
    };
    float st_s[3] = {0.f, 0.f, 0.f}, st_q[3] = {0.f, 0.f, 0.f};       // STATS: this lane's sums over its co of every tile
    for (int s = y0; s < y1 + 3; ++s) {
        const int row = s - prt;                          // this part's output row
        const bool active = row >= y0 && row < y1;
        f32x4 acc[3][XT];
        frag8 a[PF + 1][XT];
        const T* pl = &xs[0][0][0];
        auto live = [&](int c) { return !SKIP || c + 1 < kF48Chunks || prt == 3; };
        auto load_a = [&](frag8 (&dst)[XT], int c) {
            const T* ap;
            if constexpr (PART >= 0) {
                // Round 5: with the K part a compile-time fact the tap row ky of every lane group of every chunk is one too (once
                // the chunk loop is unrolled): 37 of the 41 chunks lie inside one ky - the ring slot is then SCALAR arithmetic
                // shared by the chunks of that ky - and the rest split at a known lane group.  Per chunk: one v_add (two selects
                // more in a split chunk) where the per-lane form below spends five vector instructions (shift, add, and,
                // multiply-add, and): ~45 of the ~125 vector instructions of a step.
                const int cb = chunk_begin(PART);
                int ty[4];
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    const int k = 32 * (cb + c) + 8 * gg;
                    const bool lv = c < chunk_count(PART) && k < kF48K;
                    const int tap = (lv ? k : 32 * cb + 8 * gg) / kFwCi;
                    ty[gg] = (tap % 9) / 3;
                }
                int off = ((row + ty[0] + 7) & 3) * kSlot;                 // scalar
#pragma unroll
                for (int gg = 1; gg < 4; ++gg)
                    if (ty[gg] != ty[gg - 1]) off = g >= gg ? ((row + ty[gg] + 7) & 3) * kSlot : off;
                ap = pl + off + aoffs[c];
            } else {
                const int slot = (row + (aoff[c] >> 28) + 7) & 3;          // input row = row + ky - 1
                ap = pl + slot * kSlot + (aoff[c] & 0x0fffffff);
            }
#pragma unroll
            for (int u = 0; u < XT; ++u) dst[u] = *reinterpret_cast<const frag8*>(ap + u * 16 * CP);
        };
        if (active) {
            if (prt == 0) {                               // the chain starts from the bias (a lane's four results share a co)
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int u = 0; u < XT; ++u) acc[t][u] = f32x4{bias[t], bias[t], bias[t], bias[t]};
            } else {
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int u = 0; u < XT; ++u) acc[t][u] = hand[(s + 1) & 1][prt - 1][t * XT + u][lane];
            }
#pragma unroll
            for (int c = 0; c < PF; ++c) load_a(a[c], c);
        }
        RowRegs r[3];
        fetch_s(r, s);                                    // in flight during this step's MFMAs
        // accumulate variant: what y holds for this row is fetched NOW and added after the MFMAs (round 3 loaded it in the epilogue:
        // a memory round trip between the last MFMA and the stores of every row - profiles/r04_conv_pmc.log: the `_Accum`
        // launches took twice the wave cycles of the plain ones)
        u32x2 oldy[XT][3];
        u32x4 oldw[3];
        if (ACC && WIDE && active && prt == 3) {
#pragma unroll
            for (int t = 0; t < 3; ++t) oldw[t] = *reinterpret_cast<const u32x4*>(ybase(row, t, 0) + ylane16);
        }
        if (ACC && !WIDE && active && prt == 3) {
#pragma unroll
            for (int u = 0; u < XT; ++u) {
                const int xg = x0 + u * 16 + 4 * g;
                // lanes beyond W neither load nor store (a tile beyond W starts past the row's end); a block inside W: no lane mask
                if (x0 + kC32XB <= P.W || xg < P.W) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) oldy[u][t] = *reinterpret_cast<const u32x2*>(ybase(row, t, u) + ylane);
                }
            }
        }
        SEGM_SCHED_FENCE();
        if (active) {
#pragma unroll
            for (int c = 0; c < kF48Chunks; ++c) {
                if (c + PF < kF48Chunks && live(c + PF)) load_a(a[(c + PF) % (PF + 1)], c + PF);     // PF chunks (6 MFMAs each) ahead of its use
                SEGM_SCHED_FENCE();
                if (live(c)) {
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int u = 0; u < XT; ++u) acc[t][u] = Mfma16<T>::run(a[c % (PF + 1)][u], wf[t][c], acc[t][u]);
                }
                SEGM_SCHED_FENCE();
            }
            if (prt < 3) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int u = 0; u < XT; ++u) hand[s & 1][prt][t * XT + u][lane] = acc[t][u];
            } else if constexpr (WIDE) {                  // (the launcher: W a multiple of the 32-wide block, nothing ragged)
                static_assert(!WIDE || XT == 2, "the wide epilogue pairs the wave's two x tiles");
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    if constexpr (STATS) store_pair<T, ACC, true>(ybase(row, t, 0) + ylane16, acc[t][0], acc[t][1], oldw[t], st_s[t], st_q[t]);
                    else { float d0 = 0.f, d1 = 0.f; store_pair<T, ACC, false>(ybase(row, t, 0) + ylane16, acc[t][0], acc[t][1], oldw[t], d0, d1); }
                }
            } else {
#pragma unroll
                for (int u = 0; u < XT; ++u) {
                    const int xg = x0 + u * 16 + 4 * g;   // this lane's 4 output positions
                    if (xg >= P.W) continue;
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        float v[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = acc[t][u][q];
                        if (ACC) {
                            T o[4];
                            memcpy(o, &oldy[u][t], 8);
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] += to_f32(o[q]);
                        }
                        if constexpr (STATS) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) { st_s[t] += v[q]; st_q[t] = fmaf(v[q], v[q], st_q[t]); }
                        }
                        store4<T, false>(reinterpret_cast<T*>(ybase(row, t, u) + ylane), v);
                    }
                }
            }
        }
        SEGM_SCHED_FENCE();
        park_s(r, s);
        __syncthreads();                                  // incoming rows and the hand-off tiles are in LDS
    }
    if constexpr (STATS) {
        if (prt == 3) {
            const int nparts = P.D * P.ysplit * P.nxb, pid = (z * P.ysplit + ypart) * P.nxb + xb;
            const int nx = P.W - x0 < kC32XB ? (P.W - x0 > 0 ? P.W - x0 : 0) : kC32XB;
            const float cnt = (float)((y1 - y0) * nx);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                float a_ = st_s[t], q_ = st_q[t];
                a_ += __shfl_xor(a_, 16, 64); q_ += __shfl_xor(q_, 16, 64);
                a_ += __shfl_xor(a_, 32, 64); q_ += __shfl_xor(q_, 32, 64);
                if (g == 0)
                    reinterpret_cast<float4*>(P.stats)[((int64_t)b * P.cout + cob * 48 + t * 16 + i16) * nparts + pid] = float4{cnt, a_, q_, 0.f};
            }
        }
    }
    };
    if constexpr ((VAR & 8) != 0) {
        switch (part) {
            case 0: steps(std::integral_constant<int, 0>{}); break;
            case 1: steps(std::integral_constant<int, 1>{}); break;
            case 2: steps(std::integral_constant<int, 2>{}); break;
            default: steps(std::integral_constant<int, 3>{}); break;
        }
    } else {
        steps(std::integral_constant<int, -1>{});
    }
}

struct FwPlan { int nxb, ysplit, rows_per_part, nitems; };
static FwPlan fwd_plan(int batch, int cout, int d, int h, int w, bool chain = false, int xb = kFwXB) {
    FwPlan p;
    p.nxb = (w + xb - 1) / xb;
    // cut y when there are too few workgroups for 256 CUs.  The chained kernels pay three drain steps per cut, so they stop
    // at one workgroup per slot (one per CU for 64-wide x blocks, two for 32-wide ones); the other kernels aim for two rounds.
    const int64_t wgs = (int64_t)batch * d * p.nxb * (chain ? (cout + 47) / 48 : (cout + kFwCo - 1) / kFwCo);
    const int64_t target = chain ? (xb == kFwXB ? 256 : 512) : 512;
    int split = 1;
    while (wgs * split < target && h / (split * 2) >= 8) split *= 2;
    p.ysplit = split;
    p.rows_per_part = (h + split - 1) / split;
    p.nitems = batch * d * p.nxb * split;
    return p;
}

// the wide epilogue of the chained kernels (pair_swap + 16-byte stores): every x block inside the row (nothing ragged to mask);
// SEGM_CONV_WIDE=0: the 8-byte form (A/B)
static bool conv_wide(int width, int xb) {
    const char* e = getenv("SEGM_CONV_WIDE");             // (per launch: the tests switch it)
    return !(e && e[0] == '0') && width % xb == 0;
}

static int chain_var() {                                // 3 = the shipped schedule, 0 = round 3's (A/B timing)
    static const int v = [] { const char* e = getenv("SEGM_CONV_CHAIN_VAR"); return e && atoi(e) == 0 ? 0 : 3; }();
    return v;
}

template <int CHAIN>                                    // 0: reduce-per-row kernel; else the chained kernel with that ci pitch
static void launch48(const ConvFwdDev& P, dim3 grid, bool f16, bool acc, hipStream_t stream) {
    if (CHAIN == 48 && chain_var() != 0) {
    // the accumulate variant keeps the old outputs of a row in flight during its MFMAs: fragments ONE chunk ahead there (V & ~2).
    // W_: the wide epilogue (16-byte stores; W a multiple of the x block - conv_wide())
#define SEGM_LV(T, V, S_, W_) do { if (acc) hipLaunchKernelGGL((conv3d_k3_fwd48_chain_kernel<T, true, 48, (V) & ~2, S_, W_>), grid, dim3(512), 0, stream, P); \
                        else hipLaunchKernelGGL((conv3d_k3_fwd48_chain_kernel<T, false, 48, V, S_, W_>), grid, dim3(512), 0, stream, P); } while (0)
#define SEGM_LW(T, S_) do { if (conv_wide(P.W, kFwXB)) SEGM_LV(T, 11, S_, true); else SEGM_LV(T, 11, S_, false); } while (0)
        if (P.stats) {                                  // + the statistics epilogue of the storing K part
            if (f16) SEGM_LW(f16_t, true); else SEGM_LW(bf16_t, true);
            return;
        }
        if (f16) SEGM_LW(f16_t, false); else SEGM_LW(bf16_t, false);       // the 64-wide kernel: also instantiated per K part (bit 3)
#undef SEGM_LW
#undef SEGM_LV
        return;
    }
#define SEGM_L48(T, A)                                                                                          \
    do {                                                                                                        \
        if (CHAIN) hipLaunchKernelGGL((conv3d_k3_fwd48_chain_kernel<T, A, CHAIN ? CHAIN : kFwCP>), grid, dim3(512), 0, stream, P); \
        else hipLaunchKernelGGL((conv3d_k3_fwd48_kernel<T, A>), grid, dim3(kF48Threads), 0, stream, P);         \
    } while (0)
    if (f16) { if (acc) SEGM_L48(f16_t, true); else SEGM_L48(f16_t, false); }
    else { if (acc) SEGM_L48(bf16_t, true); else SEGM_L48(bf16_t, false); }
#undef SEGM_L48
}

}  // namespace segm

using namespace segm;

extern "C" int32_t segm_conv3d_k3_fwd_stats_parts(int32_t depth, int32_t height, int32_t width, int32_t batch, int32_t cout, int32_t flags) {
    if (depth <= 0 || height <= 0 || width <= 0 || batch <= 0 || cout <= 0) return 0;
    const bool chain32 = (flags & SEGM_CONV_FWD_CHAIN32) != 0;
    if (!chain32 && !(flags & SEGM_CONV_FWD_CHAIN)) return 0;
    if (!chain32 && chain_var() == 0) return 0;           // SEGM_CONV_CHAIN_VAR=0: the 64-wide schedule without the statistics epilogue
    const FwPlan pl = fwd_plan(batch, cout, depth, height, width, true, chain32 ? kC32XB : kFwXB);
    return depth * pl.ysplit * pl.nxb * (chain32 ? 1 : 2);
}

extern "C" int segm_conv3d_k3_fwd(const segm_conv3d_fwd_args* a) {
    if (!a) return SEGM_E_NULL;
    if (!a->x || !a->y || !a->w_packed) return SEGM_E_NULL;
    if (a->batch <= 0 || a->depth <= 0 || a->height <= 0 || a->width <= 0) return SEGM_E_SHAPE;
    if (a->cin < 1 || a->cin > kFwCi || a->cout <= 0 || a->cout % 16 != 0) return SEGM_E_SHAPE;
    if (a->width % 8 != 0) return SEGM_E_SHAPE;
    if (a->dtype != SEGM_BF16 && a->dtype != SEGM_F16) return SEGM_E_DTYPE;
    if (a->flags & ~(SEGM_CONV_FWD_ACCUMULATE | SEGM_CONV_FWD_CHAIN | SEGM_CONV_FWD_PITCH48 | SEGM_CONV_FWD_CHAIN32)) return SEGM_E_SHAPE;
    if ((a->flags & (SEGM_CONV_FWD_CHAIN | SEGM_CONV_FWD_CHAIN32)) && a->cout % 48 != 0) return SEGM_E_SHAPE;
    if ((a->flags & SEGM_CONV_FWD_CHAIN32) && (a->flags & (SEGM_CONV_FWD_CHAIN | SEGM_CONV_FWD_PITCH48))) return SEGM_E_SHAPE;
    if ((a->flags & SEGM_CONV_FWD_PITCH48) && !(a->flags & SEGM_CONV_FWD_CHAIN)) return SEGM_E_SHAPE;
    const int64_t st[8] = {a->x_stride_b, a->x_stride_c, a->x_stride_z, a->x_stride_y,
                           a->y_stride_b, a->y_stride_c, a->y_stride_z, a->y_stride_y};
    for (int64_t s : st)
        if (s % 8 != 0) return SEGM_E_SHAPE;              // 16-byte aligned rows
    if (((uintptr_t)a->x & 15) || ((uintptr_t)a->y & 15) || ((uintptr_t)a->w_packed & 15)) return SEGM_E_SHAPE;

    ConvFwdDev P;
    memset(&P, 0, sizeof(P));
    P.x = (const char*)a->x; P.x_sb = a->x_stride_b; P.x_sc = a->x_stride_c; P.x_sz = a->x_stride_z; P.x_sy = a->x_stride_y;
    P.y = (char*)a->y; P.y_sb = a->y_stride_b; P.y_sc = a->y_stride_c; P.y_sz = a->y_stride_z; P.y_sy = a->y_stride_y;
    P.wp = a->w_packed;
    P.bias = a->bias;
    P.B = a->batch; P.D = a->depth; P.H = a->height; P.W = a->width; P.cout = a->cout; P.cin = a->cin;
    const bool chain32 = (a->flags & SEGM_CONV_FWD_CHAIN32) != 0;
    const FwPlan pl = fwd_plan(a->batch, a->cout, a->depth, a->height, a->width, (a->flags & (SEGM_CONV_FWD_CHAIN | SEGM_CONV_FWD_CHAIN32)) != 0,
                               chain32 ? kC32XB : kFwXB);
    P.nxb = pl.nxb; P.ysplit = pl.ysplit; P.rows_per_part = pl.rows_per_part; P.ncob = a->cout / 48 > 0 ? a->cout / 48 : 1;
    hipStream_t stream = (hipStream_t)a->stream;
    const bool f16 = a->dtype == SEGM_F16;
    const bool acc = (a->flags & SEGM_CONV_FWD_ACCUMULATE) != 0;
    if (a->stats_partials) {
        // the statistics epilogue exists in the shipped schedules of the two chained kernels (unpadded 64-wide, 32-wide)
        const bool c64 = (a->flags & SEGM_CONV_FWD_CHAIN) && (a->flags & SEGM_CONV_FWD_PITCH48) && chain_var() != 0;
        if (!(c64 || chain32) || a->cout % 48 != 0) return SEGM_E_SHAPE;
        if (a->stats_nparts != segm_conv3d_k3_fwd_stats_parts(a->depth, a->height, a->width, a->batch, a->cout, a->flags)) return SEGM_E_WORKSPACE;
        P.stats = a->stats_partials;
    }
    // the 48-channel kernels address a row as [uniform base + 32-bit lane offset]: 48 channel strides must fit
    const bool off32 = ((int64_t)47 * a->x_stride_c + a->width) * 2 < ((int64_t)1 << 32);
    if ((a->flags & (SEGM_CONV_FWD_CHAIN | SEGM_CONV_FWD_CHAIN32)) && !off32) return SEGM_E_SHAPE;
    if ((a->flags & (SEGM_CONV_FWD_CHAIN | SEGM_CONV_FWD_CHAIN32)) && ((int64_t)16 * a->y_stride_c + a->width) * 2 >= ((int64_t)1 << 32)) return SEGM_E_SHAPE;
    if (chain32) {
        const dim3 grid(pl.nitems * (a->cout / 48));
#define SEGM_L32(T, A) hipLaunchKernelGGL((conv3d_k3_fwd48_chain32_kernel<T, A>), grid, dim3(256), 0, stream, P)
#define SEGM_LV(T, V, S_, W_) do { if (acc) hipLaunchKernelGGL((conv3d_k3_fwd48_chain32_kernel<T, true, V, S_, W_>), grid, dim3(256), 0, stream, P); \
                        else hipLaunchKernelGGL((conv3d_k3_fwd48_chain32_kernel<T, false, V, S_, W_>), grid, dim3(256), 0, stream, P); } while (0)
#define SEGM_LW(T, S_) do { if (conv_wide(P.W, kC32XB)) SEGM_LV(T, 9, S_, true); else SEGM_LV(T, 9, S_, false); } while (0)
        if (P.stats) {                                  // + the statistics epilogue of the storing K part
            if (f16) SEGM_LW(f16_t, true); else SEGM_LW(bf16_t, true);
            return (int)hipGetLastError();
        }
        if (chain_var() != 0) {
            // 9: skip + per-part loop, fragments one chunk ahead (two ahead measured nothing and spills here: profiles/r04_conv_chain32_pf2.log)
            if (f16) SEGM_LW(f16_t, false); else SEGM_LW(bf16_t, false);
            return (int)hipGetLastError();
        }
#undef SEGM_LW
#undef SEGM_LV
        if (f16) { if (acc) SEGM_L32(f16_t, true); else SEGM_L32(f16_t, false); }
        else { if (acc) SEGM_L32(bf16_t, true); else SEGM_L32(bf16_t, false); }
#undef SEGM_L32
        return (int)hipGetLastError();
    }
    if (a->flags & SEGM_CONV_FWD_CHAIN) {
        if (a->flags & SEGM_CONV_FWD_PITCH48) launch48<48>(P, dim3(pl.nitems * (a->cout / 48)), f16, acc, stream);
        else launch48<kFwCP>(P, dim3(pl.nitems * (a->cout / 48)), f16, acc, stream);
        return (int)hipGetLastError();
    }
    if (a->cout % 48 == 0 && off32 && (acc || !getenv("SEGM_CONV_FWD_KZ_SPLIT"))) {     // the env switch forces the 32 + 16 kernels (A/B timing)
        launch48<0>(P, dim3(pl.nitems * (a->cout / 48)), f16, acc, stream);
        return (int)hipGetLastError();
    }
    if (acc) return SEGM_E_SHAPE;                         // in-place accumulation is a feature of the 48-channel kernels
    const int full = a->cout / kFwCo;                     // blocks with two co tiles; cout % 32 == 16 leaves one with a single tile
    P.cob0 = 0;
    if (full > 0) {
        const dim3 grid(pl.nitems, full);
        if (f16) hipLaunchKernelGGL((conv3d_k3_fwd_kernel<f16_t, 2>), grid, dim3(kFwThreads), 0, stream, P);
        else hipLaunchKernelGGL((conv3d_k3_fwd_kernel<bf16_t, 2>), grid, dim3(kFwThreads), 0, stream, P);
    }
    if (a->cout % kFwCo) {
        P.cob0 = full;
        const dim3 grid(pl.nitems, 1);
        if (f16) hipLaunchKernelGGL((conv3d_k3_fwd_kernel<f16_t, 1>), grid, dim3(kFwThreads), 0, stream, P);
        else hipLaunchKernelGGL((conv3d_k3_fwd_kernel<bf16_t, 1>), grid, dim3(kFwThreads), 0, stream, P);
    }
    return (int)hipGetLastError();
}
