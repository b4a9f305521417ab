// Selective scan backward, regular shapes: the main kernel on 8-step windows (round 4; the default main kernel).
//
// What bounds this operator on the MI355X is the number of instructions a SIMD issues (~4 cycles per vector
// instruction, 8 per transcendental, at any occupancy).  The round-2 / round-3 main kernels spent a third of their ~510
// instructions per step on bookkeeping: 16-step windows needed 256 VGPRs + ~210 AGPRs (v_accvgpr moves), every 2-byte access
// formed a 64-bit address, the dB / dC channel sums went through move + select + add stages, and dB / dC left the kernel as
// fp32 atomics on a zeroed buffer (one partial per d-tile: nondeterministic in the last bits) that the host then copied into
// the x_proj gradient operand.  This kernel:
//   * works on 8-step windows from checkpoints every 8 steps (kCkpt): a pair's a_t / h_t are 32 registers, no AGPRs, and the
//     kernel fits three waves per SIMD;
//   * dB / dC leave without atomics: a wave's channel sums of a window go to the fp32 slab of its d-tile as 16-byte stores, and a
//     small second kernel adds the slabs in tile order - deterministic, no zero-initialised buffer - and writes the result in
//     the element type and strides the caller asked for, e.g. straight into the columns of the x_proj gradient operand.
//     (A first round-4 version put all d-tiles of a chunk group into ONE workgroup and summed them through LDS: 96 channels are
//     three waves, which tile a CU's four SIMDs only at three waves per SIMD, i.e. 168 registers; at that budget the compiler
//     spills ~90 registers and the kernel took 1.29 ms, at two waves per SIMD - one SIMD in four half empty - 0.73 ms;
//     profiles/r04_scan_ab1.log.)
//   * the channel sums of one state PAIR and window (32 values per lane) are one reduce-scatter whose first stages exchange
//     registers with v_permlane32_swap / v_permlane16_swap and add packed pairs;
//   * rows are addressed as buffer resource + constant lane offset + scalar row offset (scan_fast.h).
// Per (step, state pair): forward a = exp2(delta A2), h = a h_prev + (delta u) B (3 packed + 2 v_exp); backward dh = g C + e,
// e' = a dh, t2 = e' h_prev, dA += t2 delta, q += dh B, ddelta += t2 A, dB_c = dh (delta u), dC_c = g h (8 packed).
#include <stdlib.h>

#include "scan_fast.h"

#ifndef SEGM_BWD_CKPT_AUX
#define SEGM_BWD_CKPT_AUX 0            // cache policy of the checkpoint loads (2 = nt: read once)
#endif

namespace segm {

constexpr int kW8 = 8;                 // steps per window = spacing of the forward checkpoints
static_assert(kW8 == kCkpt && kW8 == kFT, "one window per checkpoint and sub-tile");

#ifndef SEGM_W8_MIN_WAVES
#define SEGM_W8_MIN_WAVES 2            // waves per SIMD the register allocator must leave room for
#endif

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ void stream_fetch_raw(uint32_t (&dst)[kFT], const Stream& st, int32_t rows, int32_t dT) {
    uint32_t so = (uint32_t)rows * (uint32_t)st.stb;
    const uint32_t inc = (uint32_t)(dT * st.stb);
#pragma unroll
    for (int j = 0; j < kFT; ++j) {
        dst[j] = BufIO<T>::ld_raw(st.rs, st.voff, so);
        so += inc;
    }
}
template <typename T>
__device__ __forceinline__ void stream_store(const float (&src)[kFT], const Stream& st, int32_t rows, int32_t dT) {
    uint32_t so = (uint32_t)rows * (uint32_t)st.stb;
    const uint32_t inc = (uint32_t)(dT * st.stb);
#pragma unroll
    for (int j = 0; j < kFT; ++j) {
        BufIO<T>::st(st.rs, st.voff, so, src[j]);
        so += inc;
    }
}

// the raw rows of u kept for the window epilogue: two 16-bit elements per register
template <typename T> struct RawKeep {
    static constexpr int N = sizeof(T) == 2 ? kW8 / 2 : kW8;
    uint32_t v[N];
    __device__ __forceinline__ void put(const uint32_t (&r)[kW8]) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = sizeof(T) == 2 ? (r[2 * i] | (r[2 * i + 1] << 16)) : r[i];
    }
    __device__ __forceinline__ float get(int j) const {
        if (sizeof(T) == 2) return BufIO<T>::cvt_raw((j & 1) ? (v[j >> 1] >> 16) : (v[j >> 1] & 0xffffu));
        return BufIO<T>::cvt_raw(v[j]);
    }
};

// LDS image of a window's B / C: [step][state pair][B_2p, B_2p+1, C_2p, C_2p+1] - one 16-byte broadcast read per step and pair
__device__ __forceinline__ int bc_index(int m, int j, int n) { return (j * (kFS / 2) + (n >> 1)) * 4 + m * 2 + (n & 1); }

// ---- reduce-scatter stages inside a row of 16 lanes ---------------------------------------------------------------------------
// A stage with partner lane ^ m: a lane whose bit is clear ends with lo + (partner's lo), a lane whose bit is set with
// hi + (partner's hi), in lo's register.  For the two stages whose bit is a BANK bit (bit 3: partner row_ror:8, bit 2: partner
// row_half_mirror = lane ^ 7) that is two instructions per element - v_add_f32_dpp with a bank mask writes only the lanes of one
// class and leaves the others as they are - where the intrinsic form needs move, move, select, add.  The compiler cannot emit a
// DPP add whose untouched lanes keep a value that is not one of its sources, hence inline assembly; the hardware wants two wait
// states between a vector write of a register and a DPP read of it, which the compiler does not track through asm operands:
// every block starts with s_nop 1 (its DPP sources are inputs), and no block reads through DPP what it wrote itself.
#ifndef SEGM_W8_DPP_ASM
#define SEGM_W8_DPP_ASM 1
#endif
template <int CTRL> __device__ __forceinline__ float row_pick(float lo, float hi, bool up) {
    const float slo = lo + dpp_get<CTRL>(lo), shi = hi + dpp_get<CTRL>(hi);
    return up ? shi : slo;
}
// BIT = 3 (partner lane ^ 8) or 2 (partner lane ^ 7); `up` = the lane's bit
// ONE list of operations per stage feeds both builds: the GPU build pastes it into the assembly text, the CPU emulation build
// runs the same (destination, DPP source, plain source, control, bank mask) tuples through tests/emu's model of v_add_f32_dpp - so
// the lane masks and controls of the assembly are checked off the GPU too (round 6; the hazards, s_nop placement, stay GPU-only).
//   OP(d, a, b, ctl, bank):  v_add_f32_dpp %d, %a, %b <ctl> row_mask:0xf bank_mask:<bank>   =   r[d] = r[a](dpp lane) + r[b]  on the
//   lanes of the enabled banks, the others keep r[d]
#define SEGM_W8_CTL_STR_ROR8 "row_ror:8"
#define SEGM_W8_CTL_NUM_ROR8 0x128
#define SEGM_W8_CTL_STR_HMIR "row_half_mirror"
#define SEGM_W8_CTL_NUM_HMIR 0x141
#define SEGM_W8_ASM(d, a, b, ctl, bank) "v_add_f32_dpp %" #d ", %" #a ", %" #b " " SEGM_W8_CTL_STR_##ctl " row_mask:0xf bank_mask:" #bank "\n\t"
#define SEGM_W8_EMU(d, a, b, ctl, bank) r[d] = hipemu_v_add_f32_dpp(r[d], r[a], r[b], SEGM_W8_CTL_NUM_##ctl, 0xf, bank);
// four (two) elements: first the lanes whose bit is clear keep lo + partner's lo, then the lanes whose bit is set take hi + partner's hi
#define SEGM_W8_LIST4(OP, ctl, bk0, bk1) OP(0, 0, 0, ctl, bk0) OP(1, 1, 1, ctl, bk0) OP(2, 2, 2, ctl, bk0) OP(3, 3, 3, ctl, bk0) \
                                         OP(0, 4, 4, ctl, bk1) OP(1, 5, 5, ctl, bk1) OP(2, 6, 6, ctl, bk1) OP(3, 7, 7, ctl, bk1)
#define SEGM_W8_LIST2(OP, ctl, bk0, bk1) OP(0, 0, 0, ctl, bk0) OP(1, 1, 1, ctl, bk0) OP(0, 2, 2, ctl, bk1) OP(1, 3, 3, ctl, bk1)
template <int BIT>
__device__ __forceinline__ void row_stage4(float& x0, float& x1, float& x2, float& x3, float y0, float y1, float y2, float y3, bool up) {
#if SEGM_W8_DPP_ASM && !defined(SEGM_EMU)
    (void)up;
    if constexpr (BIT == 3)
        asm volatile("s_nop 1\n\t" SEGM_W8_LIST4(SEGM_W8_ASM, ROR8, 0x3, 0xc)
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y0), "v"(y1), "v"(y2), "v"(y3));
    else
        asm volatile("s_nop 1\n\t" SEGM_W8_LIST4(SEGM_W8_ASM, HMIR, 0x5, 0xa)
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y0), "v"(y1), "v"(y2), "v"(y3));
#elif SEGM_W8_DPP_ASM
    (void)up;
    float r[8] = {x0, x1, x2, x3, y0, y1, y2, y3};
    if constexpr (BIT == 3) { SEGM_W8_LIST4(SEGM_W8_EMU, ROR8, 0x3, 0xc) }
    else { SEGM_W8_LIST4(SEGM_W8_EMU, HMIR, 0x5, 0xa) }
    x0 = r[0]; x1 = r[1]; x2 = r[2]; x3 = r[3];
#else
    constexpr int CTRL = BIT == 3 ? 0x128 : 0x141;
    x0 = row_pick<CTRL>(x0, y0, up); x1 = row_pick<CTRL>(x1, y1, up); x2 = row_pick<CTRL>(x2, y2, up); x3 = row_pick<CTRL>(x3, y3, up);
#endif
}
template <int BIT>
__device__ __forceinline__ void row_stage2(float& x0, float& x1, float y0, float y1, bool up) {
#if SEGM_W8_DPP_ASM && !defined(SEGM_EMU)
    (void)up;
    if constexpr (BIT == 3)
        asm volatile("s_nop 1\n\t" SEGM_W8_LIST2(SEGM_W8_ASM, ROR8, 0x3, 0xc) : "+v"(x0), "+v"(x1) : "v"(y0), "v"(y1));
    else
        asm volatile("s_nop 1\n\t" SEGM_W8_LIST2(SEGM_W8_ASM, HMIR, 0x5, 0xa) : "+v"(x0), "+v"(x1) : "v"(y0), "v"(y1));
#elif SEGM_W8_DPP_ASM
    (void)up;
    float r[4] = {x0, x1, y0, y1};
    if constexpr (BIT == 3) { SEGM_W8_LIST2(SEGM_W8_EMU, ROR8, 0x3, 0xc) }
    else { SEGM_W8_LIST2(SEGM_W8_EMU, HMIR, 0x5, 0xa) }
    x0 = r[0]; x1 = r[1];
#else
    constexpr int CTRL = BIT == 3 ? 0x128 : 0x141;
    x0 = row_pick<CTRL>(x0, y0, up); x1 = row_pick<CTRL>(x1, y1, up);
#endif
}
// the same stages on packed pairs: (x0, x1) keep-low registers, (y0, y1) the high halves
template <int BIT> __device__ __forceinline__ void row_stage_pairs(f2& x0, f2& x1, f2 y0, f2 y1, bool up) {
    float a = x0.x, b = x0.y, c = x1.x, d = x1.y;
    row_stage4<BIT>(a, b, c, d, y0.x, y0.y, y1.x, y1.y, up);
    x0 = f2{a, b};
    x1 = f2{c, d};
}
template <int BIT> __device__ __forceinline__ void row_stage_pair(f2& x0, f2 y0, bool up) {
    float a = x0.x, b = x0.y;
    row_stage2<BIT>(a, b, y0.x, y0.y, up);
    x0 = f2{a, b};
}
// the compiler's own DPP instructions may follow: two wait states behind the last assembly write
__device__ __forceinline__ void row_stage_done() {
#if SEGM_W8_DPP_ASM && !defined(SEGM_EMU)
    asm volatile("s_nop 1");
#endif
}

// Sum over the RW lanes (channels) of a work item of the 32 values (a[j].x, a[j].y, h[j].x, h[j].y), j < 8, of every lane:
// a reduce-scatter - each stage halves the values a lane carries and adds the partner's copy of the half it keeps (the lane's
// own bit picks the half: low half on a clear bit).  Partners: lane ^ 32 (v_permlane32_swap), ^ 16 (v_permlane16_swap), then inside
// a row of 16 as DPP operands: ^ 8 (row_ror:8), ^ 7 (row_half_mirror: one control reaches a lane with the other bit 2), ^ 2 and
// ^ 1 (quad_perm) - the four masks span the row, so every lane's contribution reaches the owner.  On return lane r holds in a[0].x
// (RW 16: a[0], both states) the finished sum of element (m, j, state 2p + s) with, from r's high bit down:
//   RW 64: m, j bit 2, j bit 1, j bit 0, s, (bit 0: both lanes hold it)   RW 32: m, j2, j1, j0, s   RW 16: m, j2, j1, j0
// (m = 0: dB, 1: dC; j = step of the window).
__device__ __forceinline__ f2 swap_add32(f2 x, f2 y) {     // (x.lo + x.hi | y.lo + y.hi) over the two 32-lane halves
    const u32x2_t sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(x.x), __float_as_uint(y.x), false, false);
    const u32x2_t sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(x.y), __float_as_uint(y.y), false, false);
    return f2{__uint_as_float(sx.x), __uint_as_float(sy.x)} + f2{__uint_as_float(sx.y), __uint_as_float(sy.y)};
}
__device__ __forceinline__ f2 swap_add16(f2 x, f2 y) {     // even rows: x summed over the row pair, odd rows: y
    const u32x2_t sx = __builtin_amdgcn_permlane16_swap(__float_as_uint(x.x), __float_as_uint(y.x), false, false);
    const u32x2_t sy = __builtin_amdgcn_permlane16_swap(__float_as_uint(x.y), __float_as_uint(y.y), false, false);
    return f2{__uint_as_float(sx.x), __uint_as_float(sy.x)} + f2{__uint_as_float(sx.y), __uint_as_float(sy.y)};
}
template <int RW>
__device__ __forceinline__ void reduce_scatter_pair(f2 (&a)[kW8], f2 (&h)[kW8], int r) {
    const bool up8 = (r & 8) != 0, up4 = (r & 4) != 0, up2 = (r & 2) != 0, up1 = (r & 1) != 0;
    if constexpr (RW == 64) {
#pragma unroll
        for (int j = 0; j < kW8; ++j) a[j] = swap_add32(a[j], h[j]);          // lower half: dB, upper half: dC
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = swap_add16(a[j], a[j + 4]);         // even rows: steps j, odd rows: steps j + 4
        row_stage_pairs<3>(a[0], a[1], a[2], a[3], up8);                       // steps j | j + 2
        row_stage_pair<2>(a[0], a[1], up4);                                    // steps j | j + 1
        row_stage_done();
        a[0].x = row_pick<0x4e>(a[0].x, a[0].y, up2);                         // state 2p | 2p + 1
        a[0].x += dpp_get<0xb1>(a[0].x);                                       // lanes l ^ 1: both end with the sum
    } else if constexpr (RW == 32) {
#pragma unroll
        for (int j = 0; j < kW8; ++j) a[j] = swap_add16(a[j], h[j]);          // even row: dB, odd row: dC
        row_stage_pairs<3>(a[0], a[1], a[4], a[5], up8);                       // steps j | j + 4
        row_stage_pairs<3>(a[2], a[3], a[6], a[7], up8);
        row_stage_pairs<2>(a[0], a[1], a[2], a[3], up4);                       // steps j | j + 2
        row_stage_done();
        a[0].x = row_pick<0x4e>(a[0].x, a[1].x, up2);                         // steps j | j + 1
        a[0].y = row_pick<0x4e>(a[0].y, a[1].y, up2);
        a[0].x = row_pick<0xb1>(a[0].x, a[0].y, up1);                         // state 2p | 2p + 1
    } else {                                               // RW == 16: one row of 16 lanes per item, a lane ends with both states
#pragma unroll
        for (int j = 0; j < kW8; j += 2)                                       // dB | dC
            row_stage_pairs<3>(a[j], a[j + 1], h[j], h[j + 1], up8);
        row_stage_pairs<2>(a[0], a[1], a[4], a[5], up4);                       // steps j | j + 4
        row_stage_pairs<2>(a[2], a[3], a[6], a[7], up4);
        row_stage_done();
#pragma unroll
        for (int j = 0; j < 2; ++j) {                                          // steps j | j + 2
            a[j].x = row_pick<0x4e>(a[j].x, a[j + 2].x, up2);
            a[j].y = row_pick<0x4e>(a[j].y, a[j + 2].y, up2);
        }
        a[0].x = row_pick<0xb1>(a[0].x, a[1].x, up1);                         // steps j | j + 1
        a[0].y = row_pick<0xb1>(a[0].y, a[1].y, up1);
    }
}

// ------------------------------------------------------------------------------------------------------
// K3 (regular shapes): main backward kernel.  One wave = G work items (consecutive chunks) x RW channels of one d-tile, waves are
// independent (no workgroup barrier); grid.y = direction.
// ------------------------------------------------------------------------------------------------------
// MODE 1: delta_softplus and a gate z are known at compile time (what Mamba launches); 0: read from the arguments - as
// wave-uniform branches around every step's softplus / gate arithmetic and stores, which the compiler does not unswitch
// SAME: u, delta, dout, out, du, ddelta have one set of strides and z, dz another (what Mamba launches: contiguous (B, L, D)
// tensors and the halves of xz / dxz) - two lane offsets instead of eight (the kernel is built at the register limit of two waves
// per SIMD: every register it keeps out of the state-pair loop is a spill it does not take there)
template <typename T, int RW, int MODE, bool SAME>
__global__ void __launch_bounds__(kBlock, SEGM_W8_MIN_WAVES) scan_bwd_main_w8_kernel(ScanDevN PP) {
    constexpr int G = 64 / RW, EPL = StageStream<RW>::EPL;
    constexpr int ITEM = 2 * kW8 * kFS;                   // floats of one item's dB + dC tile of a window: [j][m][n]
    // Nothing inside the state-pair loop comes from memory: vmcnt is ONE in-order counter, so a load issued there could only be
    // waited for together with the 40 row loads of the next window that are in flight around it (measured: the loop then starts
    // every window by waiting out that prefetch).  What a pair needs is in LDS - 18 KB per wave, two workgroups per CU:
    __shared__ __attribute__((aligned(16))) float s_bc[kWavesPerBlock][G][kW8 * (kFS / 2) * 4];        // the window's B, C
    __shared__ __attribute__((aligned(16))) float s_dbc[kWavesPerBlock][G][ITEM];                      // channel sums of dB / dC
    __shared__ f2 s_e[kFS / 2][kBlock];                   // adjoint entering from the right, per thread and state pair
    __shared__ f2 s_dA[kFS / 2][kBlock];
    __shared__ f2 s_hp[kFS / 2][kBlock];                  // state entering the window (forward checkpoint)
    __shared__ f2 s_A[kFS / 2][kWavesPerBlock * RW];      // A[d][2p], A[d][2p + 1] of the wave's channels
    const ScanDev& P = PP.d[blockIdx.y];
    const Geom& gm = P.gm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Item it = locate(gm, (int64_t)blockIdx.x * kWavesPerBlock + wave, lane);
    if (!it.wave_valid) return;                           // the last workgroup may have spare waves (no workgroup barriers here)
    const int ub = uniform_batch(it);
    const bool softplus_on = MODE == 1 || P.delta_softplus != 0;
    const bool has_z = MODE == 1 || P.z.p != nullptr;
    const WaveRows wr = wave_rows(P.tm, gm, it);
    const int32_t chunk0 = __builtin_amdgcn_readfirstlane(it.chunk - it.gi);
    const int32_t dtile = __builtin_amdgcn_readfirstlane(it.dt);

    const int64_t crow = (int64_t)it.b * gm.nchunks + it.chunk;
#pragma unroll
    for (int p = 0; p < kFS / 2; ++p) {
        s_e[p][tid] = f2{P.carry[(crow * kFS + 2 * p) * gm.dim + it.d], P.carry[(crow * kFS + 2 * p + 1) * gm.dim + it.d]};
        s_dA[p][tid] = f2{0.f, 0.f};
    }
    {
        const f2* Arow = reinterpret_cast<const f2*>(P.A + (int64_t)it.d * kFS);
#pragma unroll
        for (int p = 0; p < kFS / 2; ++p) s_A[p][wave * RW + it.r] = Arow[p];     // the items of a wave share the channels: same value
    }
    const float bias = P.delta_bias ? P.delta_bias[it.d] : 0.f;
    const float Dv = P.D ? P.D[it.d] : 0.f;
    float dD_acc = 0.f, dbias_acc = 0.f;

    const Stream up = make_stream<T>(P.u, ub, wr, it.d);
    Stream dp = make_stream<T>(P.delta, ub, wr, it.d);
    Stream gp = make_stream<T>(P.dout, ub, wr, it.d);
    const Stream zp = make_stream<T>(has_z ? P.z : P.dout, ub, wr, it.d);
    Stream yp = make_stream<T>(has_z ? P.out : P.dout, ub, wr, it.d);
    Stream dup = make_stream<T>(P.du, ub, wr, it.d);
    Stream ddp = make_stream<T>(P.ddelta, ub, wr, it.d);
    Stream dzp = make_stream<T>(has_z ? P.dz : P.du, ub, wr, it.d);
    if constexpr (SAME) {
        dp.voff = gp.voff = yp.voff = dup.voff = ddp.voff = up.voff;
        dp.stb = gp.stb = yp.stb = dup.stb = ddp.stb = up.stb;
        dzp.voff = zp.voff;
        dzp.stb = zp.stb;
    }
    const StageStream<RW> sb = make_stage<T, RW>(P.Bm, ub, wr, it.r);
    const StageStream<RW> sc = make_stage<T, RW>(P.Cm, ub, wr, it.r);
    int lds_b0, lds_c0, lds_binc, lds_cinc;               // where this lane's staged elements go in the [step][pair][4] image
    {
        const bool tfast_b = P.Bm.st <= P.Bm.sn, tfast_c = P.Cm.st <= P.Cm.sn;
        const int jb = tfast_b ? it.r % kFT : it.r / kFS, nb = tfast_b ? it.r / kFT : it.r % kFS;
        const int jc = tfast_c ? it.r % kFT : it.r / kFS, nc = tfast_c ? it.r / kFT : it.r % kFS;
        lds_b0 = bc_index(0, jb, nb);
        lds_c0 = bc_index(1, jc, nc);
        // time fastest: consecutive elements are RW / 8 states apart (an even number); state fastest: RW / 16 steps (at least one)
        lds_binc = tfast_b ? (RW / kFT / 2) * 4 : (RW >= kFS ? RW / kFS : 1) * (kFS / 2) * 4;
        lds_cinc = tfast_c ? (RW / kFT / 2) * 4 : (RW >= kFS ? RW / kFS : 1) * (kFS / 2) * 4;
    }
    // checkpoints [batch][nck][8 pairs][dim][2]: a buffer based at the wave's lowest chunk; a pair's entering state is one 8-byte load
    const rsrc_t ckr = make_rsrc(P.ckpt + (((int64_t)ub * P.nck + (int64_t)chunk0 * (gm.chunk / kCkpt)) * kFS) * gm.dim);
    const uint32_t ck_voff = ((uint32_t)(it.gi * (gm.chunk / kCkpt)) * kFS * (uint32_t)gm.dim + 2u * (uint32_t)it.d) * 4u;
    const int32_t ck_pair = gm.dim * 8;                   // bytes between consecutive state pairs of one checkpoint
    const int32_t ck_row = ck_pair * (kFS / 2);           // bytes of one checkpoint (all channels)
    // dB / dC of a window: one d-tile -> finished values straight to the destination (fp32, or T when the caller asked for the
    // tensors' own type); several d-tiles -> this tile's sums to its fp32 slab [d-tile][batch][row][dB 16 | dC 16], a row = 128
    // bytes, added over the d-tiles in a fixed order by scan_bwd_dbc_sum_kernel
    const bool direct = gm.ndt == 1;
    const bool bc16 = P.dbc_native != 0;
    const int32_t dbc_es = bc16 ? (int32_t)sizeof(T) : 4;
    const rsrc_t dBr = make_rsrc(reinterpret_cast<char*>(P.dB) + ((int64_t)ub * P.dB_sb + (int64_t)wr.row_lo * P.dB_st) * dbc_es);
    const rsrc_t dCr = make_rsrc(reinterpret_cast<char*>(P.dC) + ((int64_t)ub * P.dC_sb + (int64_t)wr.row_lo * P.dC_st) * dbc_es);
    const rsrc_t slabr = make_rsrc(direct ? nullptr : reinterpret_cast<char*>(P.dbc_part) +
                                   (((int64_t)dtile * gm.batch + ub) * gm.L + wr.row_lo) * (int64_t)(2 * kFS * 4));
    // slab flush: lane q of an item's 64 moves 16 bytes: row j = q / 8 of the window, bytes 16 (q % 8) of the row
    uint32_t slab_voff[G];
#pragma unroll
    for (int k = 0; k < G; ++k) {
        const int32_t item_row = fast_item_row(P.tm, (chunk0 + k) * gm.chunk) - wr.bias - wr.row_lo;     // >= 0, uniform
        const int j = lane >> 3;
        const int32_t steps = wr.dT < 0 ? (kW8 - 1 - j) * (-wr.dT) : j * wr.dT;
        slab_voff[k] = (uint32_t)(item_row + steps) * (uint32_t)(2 * kFS * 4) + (uint32_t)(lane & 7) * 16u;
    }

    const int nwin = gm.chunk / kW8;
    uint32_t ru[kW8], rd[kW8], rg[kW8], rz[kW8], ry[kW8], nbv[EPL], ncv[EPL];
    u32x2_t rh[kFS / 2];                                  // a window's checkpoint, raw
    float* const img = &s_bc[wave][it.gi][0];
    {                                                      // the first window's inputs (the only exposed fetch of the chunk)
        const int32_t U = wr.bias + fast_U_of(P.tm, nwin - 1);
#pragma unroll
        for (int p = 0; p < kFS / 2; ++p) rh[p] = __builtin_amdgcn_raw_buffer_load_b64(ckr, ck_voff, (uint32_t)((nwin - 1) * ck_row + p * ck_pair), SEGM_BWD_CKPT_AUX);
        stage_fetch_raw<T, RW>(nbv, sb, U, wr.dT);
        stage_fetch_raw<T, RW>(ncv, sc, U, wr.dT);
        stream_fetch_raw<T>(rz, zp, U, wr.dT);
        stream_fetch_raw<T>(ry, yp, U, wr.dT);
        stream_fetch_raw<T>(rg, gp, U, wr.dT);
        stream_fetch_raw<T>(rd, dp, U, wr.dT);
        stream_fetch_raw<T>(ru, up, U, wr.dT);
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            img[lds_b0 + i * lds_binc] = BufIO<T>::cvt_raw(nbv[i]);
            img[lds_c0 + i * lds_cinc] = BufIO<T>::cvt_raw(ncv[i]);
        }
#pragma unroll
        for (int p = 0; p < kFS / 2; ++p) s_hp[p][tid] = f2{__uint_as_float(rh[p].x), __uint_as_float(rh[p].y)};
    }

    for (int w = nwin - 1; w >= 0; --w) {
        const int32_t Uw = wr.bias + fast_U_of(P.tm, w);
        const int32_t Un = wr.bias + fast_U_of(P.tm, w > 0 ? w - 1 : 0);
        // ---- window prologue: this lane's 8 steps ------------------------------------------------------------------------------
        float wd[kW8], wg[kW8], wdu[kW8];
        RawKeep<T> ku;
        f2 qs[kW8], ddA[kW8];                              // sum over states of dh B and of t2 A, two partial sums each
        {
            float dzv[kW8];
            ku.put(ru);
#pragma unroll
            for (int j = 0; j < kW8; ++j) {
                wg[j] = BufIO<T>::cvt_raw(rg[j]);
                if (has_z) {
                    const float zz = BufIO<T>::cvt_raw(rz[j]), yy = BufIO<T>::cvt_raw(ry[j]);
                    const float sg = sigmoidf(zz);
                    dzv[j] = wg[j] * yy * sg * fmaf(zz, 1.f - sg, 1.f);
                    wg[j] *= zz * sg;
                }
                const float uu = BufIO<T>::cvt_raw(ru[j]);
                const float dl = BufIO<T>::cvt_raw(rd[j]) + bias;
                wd[j] = softplus_on ? softplus20(dl) : dl;
                wdu[j] = wd[j] * uu;
                qs[j] = f2{0.f, 0.f};
                ddA[j] = f2{0.f, 0.f};
                dD_acc = fmaf(wg[j], uu, dD_acc);
            }
            if (has_z) stream_store<T>(dzv, dzp, Uw, wr.dT);
        }
        // ---- the next window's inputs: in flight during the state loop (raw bits: nothing here is a use of a loaded value) --------
        {
            const uint32_t so = (uint32_t)((w > 0 ? w - 1 : 0) * ck_row);
#pragma unroll
            for (int p = 0; p < kFS / 2; ++p) rh[p] = __builtin_amdgcn_raw_buffer_load_b64(ckr, ck_voff, so + (uint32_t)(p * ck_pair), SEGM_BWD_CKPT_AUX);
        }
        stage_fetch_raw<T, RW>(nbv, sb, Un, wr.dT);
        stage_fetch_raw<T, RW>(ncv, sc, Un, wr.dT);
        if (has_z) {
            stream_fetch_raw<T>(rz, zp, Un, wr.dT);
            stream_fetch_raw<T>(ry, yp, Un, wr.dT);
        }
        stream_fetch_raw<T>(rg, gp, Un, wr.dT);
        stream_fetch_raw<T>(rd, dp, Un, wr.dT);
        stream_fetch_raw<T>(ru, up, Un, wr.dT);

        SEGM_WAVE_LDS_SYNC();                             // this window's B / C image is parked, the dB / dC tile has been flushed
        float* part = &s_dbc[wave][it.gi][0];
#pragma unroll 1
        for (int p = 0; p < kFS / 2; ++p) {               // runtime loop over state pairs
            const f2 An = s_A[p][wave * RW + it.r];
            const f2 A2n = An * kLog2e;
            const f2 hp = s_hp[p][tid];
            f2 en = s_e[p][tid];
            f2 dAn = s_dA[p][tid];
            const float* bcp = img + p * 4;
            f2 a[kW8], h[kW8];
#pragma unroll
            for (int j = 0; j < kW8; ++j) {
                const f2 bb = *reinterpret_cast<const f2*>(bcp + j * (kFS / 2) * 4);
                const f2 da = A2n * wd[j];
                a[j] = f2{fast_exp2(da.x), fast_exp2(da.y)};
                h[j] = a[j] * (j ? h[j - 1] : hp) + bb * wdu[j];
            }
#pragma unroll
            for (int jj = 0; jj < kW8; ++jj) {
                const int j = kW8 - 1 - jj;
                const float4 q4 = *reinterpret_cast<const float4*>(bcp + j * (kFS / 2) * 4);
                const f2 bb = {q4.x, q4.y}, cc = {q4.z, q4.w};
                const f2 dh = cc * wg[j] + en;
                en = a[j] * dh;
                const f2 t2 = en * (j ? h[j - 1] : hp);
                dAn = t2 * wd[j] + dAn;
                qs[j] = dh * bb + qs[j];
                ddA[j] = t2 * An + ddA[j];
                a[j] = dh * wdu[j];                        // dB contribution of (j, pair), in a's registers
                h[j] = h[j] * wg[j];                       // dC contribution, in h's registers
            }
            s_e[p][tid] = en;
            s_dA[p][tid] = dAn;
            reduce_scatter_pair<RW>(a, h, it.r);          // tile layout [j][m][n]
            if constexpr (RW == 64) {
                if ((it.r & 1) == 0) part[(((it.r >> 2) & 7) * 2 + (it.r >> 5)) * kFS + 2 * p + ((it.r >> 1) & 1)] = a[0].x;
            } else if constexpr (RW == 32) {
                part[(((it.r >> 1) & 7) * 2 + (it.r >> 4)) * kFS + 2 * p + (it.r & 1)] = a[0].x;
            } else {
                *reinterpret_cast<f2*>(&part[((it.r & 7) * 2 + (it.r >> 3)) * kFS + 2 * p]) = a[0];
            }
        }
        // ---- window epilogue -----------------------------------------------------------------------------------------------------
        {
            float du[kW8], ddl[kW8];
#pragma unroll
            for (int j = 0; j < kW8; ++j) {
                const float q = qs[j].x + qs[j].y;
                du[j] = fmaf(wd[j], q, Dv * wg[j]);
                float ddv = fmaf(ku.get(j), q, ddA[j].x + ddA[j].y);
                ddv *= softplus_on ? 1.f - fast_exp(-wd[j]) : 1.f;       // sigmoid(raw) = 1 - exp(-softplus(raw))
                dbias_acc += ddv;
                ddl[j] = ddv;
            }
            stream_store<T>(du, dup, Uw, wr.dT);
            stream_store<T>(ddl, ddp, Uw, wr.dT);
        }
        // the next window's B / C and checkpoint (this window's pairs are done with theirs)
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            img[lds_b0 + i * lds_binc] = BufIO<T>::cvt_raw(nbv[i]);
            img[lds_c0 + i * lds_cinc] = BufIO<T>::cvt_raw(ncv[i]);
        }
#pragma unroll
        for (int p = 0; p < kFS / 2; ++p) s_hp[p][tid] = f2{__uint_as_float(rh[p].x), __uint_as_float(rh[p].y)};
        SEGM_WAVE_LDS_SYNC();                             // the dB / dC tile of every item of the wave is complete
        // ---- dB / dC of the window ---------------------------------------------------------------------------------------------
        if (!direct) {
            const uint32_t soff = (uint32_t)(Uw + (wr.dT < 0 ? (kW8 - 1) * wr.dT : 0)) * (uint32_t)(2 * kFS * 4);
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(&s_dbc[wave][k][lane * 4]);
                __builtin_amdgcn_raw_buffer_store_b128(v, slabr, slab_voff[k], soff, 0);
            }
        } else {
#pragma unroll
            for (int k = 0; k < G * ITEM / 64; ++k) {     // 64 consecutive elements = 2 steps x (dB 16 | dC 16) of one item
                const int e0 = k * 64, gk = e0 / ITEM, j = ((e0 % ITEM) / (2 * kFS)) + (lane >> 5), mk = (lane >> 4) & 1, n = lane & (kFS - 1);
                const float s = s_dbc[wave][gk][(e0 % ITEM) + lane];
                const int32_t item_row = fast_item_row(P.tm, (chunk0 + gk) * gm.chunk) - wr.bias - wr.row_lo;       // >= 0, uniform
                const int32_t steps = wr.dT < 0 ? (kW8 - 1 - j) * (-wr.dT) : j * wr.dT;                            // >= 0
                const uint32_t vb = ((uint32_t)(item_row + steps) * (uint32_t)P.dB_st + (uint32_t)n * (uint32_t)P.dB_sn) * (uint32_t)dbc_es;
                const uint32_t vc = ((uint32_t)(item_row + steps) * (uint32_t)P.dC_st + (uint32_t)n * (uint32_t)P.dC_sn) * (uint32_t)dbc_es;
                const uint32_t rows = (uint32_t)(Uw + (wr.dT < 0 ? (kW8 - 1) * wr.dT : 0));
                const uint32_t sbo = rows * (uint32_t)P.dB_st * (uint32_t)dbc_es, sco = rows * (uint32_t)P.dC_st * (uint32_t)dbc_es;
                if (mk == 0) {
                    if (bc16) BufIO<T>::st(dBr, vb, sbo, s); else BufIO<float>::st(dBr, vb, sbo, s);
                } else {
                    if (bc16) BufIO<T>::st(dCr, vc, sco, s); else BufIO<float>::st(dCr, vc, sco, s);
                }
            }
        }
    }
    const int64_t row = crow * (kFS + 2);
#pragma unroll
    for (int p = 0; p < kFS / 2; ++p) {
        const f2 dA = s_dA[p][tid];
        P.part[(row + 2 * p) * gm.dim + it.d] = dA.x;
        P.part[(row + 2 * p + 1) * gm.dim + it.d] = dA.y;
    }
    P.part[(row + kFS) * gm.dim + it.d] = dD_acc;
    P.part[(row + kFS + 1) * gm.dim + it.d] = dbias_acc;
}

// dB / dC = sum over the d-tiles of their slabs, in tile order (deterministic), converted to the destination's element type and
// stored through its strides: a thread owns four consecutive states of one (batch, row, matrix)
template <typename T>
__global__ void __launch_bounds__(256) scan_bwd_dbc_sum_kernel(ScanDevN PP) {
    const ScanDev& P = PP.d[blockIdx.z];
    const Geom& gm = P.gm;
    const int b = blockIdx.y;
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;           // (row, m, n / 4)
    if (q >= (int64_t)gm.L * 8) return;
    const int64_t t = q >> 3;
    const int m = (int)(q >> 2) & 1, n0 = (int)(q & 3) * 4;
    const int64_t slab = (int64_t)gm.batch * gm.L * (2 * kFS);           // floats per d-tile
    const float* src = P.dbc_part + ((int64_t)b * gm.L + t) * (2 * kFS) + m * kFS + n0;
    float4 s = *reinterpret_cast<const float4*>(src);
    for (int v = 1; v < gm.ndt; ++v) {
        const float4 x = *reinterpret_cast<const float4*>(src + v * slab);
        s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
    }
    const float r[4] = {s.x, s.y, s.z, s.w};
    const int64_t sb = m ? P.dC_sb : P.dB_sb, st = m ? P.dC_st : P.dB_st, sn = m ? P.dC_sn : P.dB_sn;
    char* base = reinterpret_cast<char*>(m ? P.dC : P.dB);
    const int64_t off = (int64_t)b * sb + t * st + (int64_t)n0 * sn;
    if (sn == 1) {                                         // the states are adjacent (dx_dbl's B | C columns): one store, not four
        if (P.dbc_native) {
            if constexpr (sizeof(T) == 2) {
                T* d = reinterpret_cast<T*>(base) + off;
                if ((reinterpret_cast<uintptr_t>(d) & 7u) == 0) {
                    u32x2_t pk;
                    pk[0] = pack2<T>(r[0], r[1]);
                    pk[1] = pack2<T>(r[2], r[3]);
                    *reinterpret_cast<u32x2_t*>(d) = pk;
                    return;
                }
            }
        } else {
            float* d = reinterpret_cast<float*>(base) + off;
            if ((reinterpret_cast<uintptr_t>(d) & 15u) == 0) {
                *reinterpret_cast<float4*>(d) = s;
                return;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (P.dbc_native) reinterpret_cast<T*>(base)[off + i * sn] = from_f32<T>(r[i]);
        else reinterpret_cast<float*>(base)[off + i * sn] = r[i];
    }
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
bool scan_bwd_w8_shape(const ScanDev& P, size_t esize) {
    (void)esize;
    return scan_fast_shape(P);
}
// bytes of the per-d-tile dB / dC slabs of one launch (none when a single d-tile covers the channels)
size_t scan_bwd_w8_slab_bytes(int batch, int dim, int nstate, int64_t L) {
    const Geom gm = make_geom(batch, dim, nstate, L, kChunkQuantum);
    return gm.ndt > 1 && nstate == kFS ? (size_t)gm.ndt * batch * L * (2 * kFS) * sizeof(float) : 0;
}

template <typename T, int RW>
static void launch_w8_rw(const ScanDevN& PP, int ndir, hipStream_t stream) {
    const Geom& gm = PP.d[0].gm;
    const unsigned nblocks = (unsigned)((gm.nwaves + kWavesPerBlock - 1) / kWavesPerBlock);
    bool mamba = true, same = true;                        // every direction: softplus on delta, gated by z; the two stride sets
    for (int i = 0; i < ndir; ++i) {
        const ScanDev& P = PP.d[i];
        mamba = mamba && P.delta_softplus != 0 && P.z.p != nullptr;
        const Seq* six[5] = {&P.delta, &P.dout, &P.out, &P.du, &P.ddelta};
        for (const Seq* s : six) same = same && s->p && s->sb == P.u.sb && s->st == P.u.st && s->sd == P.u.sd;
        same = same && P.z.p && P.dz.p && P.dz.sb == P.z.sb && P.dz.st == P.z.st && P.dz.sd == P.z.sd;
    }
    if (mamba && same) hipLaunchKernelGGL((scan_bwd_main_w8_kernel<T, RW, 1, true>), dim3(nblocks, ndir), dim3(kBlock), 0, stream, PP);
    else if (mamba) hipLaunchKernelGGL((scan_bwd_main_w8_kernel<T, RW, 1, false>), dim3(nblocks, ndir), dim3(kBlock), 0, stream, PP);
    else hipLaunchKernelGGL((scan_bwd_main_w8_kernel<T, RW, 0, false>), dim3(nblocks, ndir), dim3(kBlock), 0, stream, PP);
    if (gm.ndt > 1)
        hipLaunchKernelGGL((scan_bwd_dbc_sum_kernel<T>), dim3((unsigned)(((int64_t)gm.L * 8 + 255) / 256), gm.batch, ndir), dim3(256), 0, stream, PP);
}
template <typename T>
static void launch_w8_t(const ScanDevN& PP, int ndir, hipStream_t stream) {
    if (PP.d[0].gm.rw == 64) launch_w8_rw<T, 64>(PP, ndir, stream);
    else if (PP.d[0].gm.rw == 32) launch_w8_rw<T, 32>(PP, ndir, stream);
    else launch_w8_rw<T, 16>(PP, ndir, stream);
}
void launch_scan_bwd_main_w8(const ScanDevN& PP, int ndir, int dtype, hipStream_t stream) {
    if (dtype == SEGM_F32) launch_w8_t<float>(PP, ndir, stream);
    else if (dtype == SEGM_F16) launch_w8_t<f16_t>(PP, ndir, stream);
    else launch_w8_t<bf16_t>(PP, ndir, stream);
}

}  // namespace segm
