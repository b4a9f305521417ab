// Selective scan forward, the "regular shape" kernels (same algorithm and workspace as scan_fwd.hip's K1 / K3).
//
// scan_fwd.hip handles every shape: ragged chunk tails, padded channel tiles, arbitrary slice counts.  That generality
// costs instructions the SegMamba shapes never need - per-lane time-index arithmetic, bounds masks and the selects that
// apply them are ~45 % of the issue slots of its apply kernel (tools/isa_mix.py), and the kernel is VALU-issue bound
// (nothing co-issues with a v_exp_f32).  When
//     nstate == 16,  dim % RW == 0,  L % chunk == 0,  nchunks % (64/RW) == 0,  and the time order is affine inside a
//     sub-tile and identical for every work item  (FORWARD / REVERSED always; INTERLEAVED when nslices % 8 == 0 and
//     chunk % nslices == 0)
// the host launches these kernels instead:
//   * address = wave-uniform base (SGPR arithmetic: batch, sub-tile, step) + one per-lane byte offset that is constant
//     for the whole kernel - no per-access VALU address math, no masks;
//   * the 16-state update runs on packed fp32 (v_pk_mul_f32 / v_pk_fma_f32: two states per instruction), which halves
//     the non-transcendental issue slots of a step:  per state pair  2 v_exp + 2 v_pk_mul + 1 (agg) or 2 (apply) v_pk_fma;
//   * (round 3) the row streams u / delta / z are rings of 8 registers each: a step consumes its value and refills the register
//     with the same step of the next sub-tile (no second copy of the sub-tile: apply 134 -> 94 VGPRs), the B / C stage loads are
//     the oldest loads on both paths into the loop (no vmcnt(0) at the loop head), the per-step flags are template arguments
//     for the two combinations training and inference use, and broadcast operands of packed instructions are real pairs (an
//     op_sel broadcast reads the odd register of its pair, which may be a ring load in flight).  None of this changed the time:
//     the kernels are bound by the instructions a SIMD issues, ~4.4 cycles each (9 for v_exp / v_log / v_rcp) at any occupancy -
//     340 / 520 cycles per wave-step for the 72 / 118 instructions of the two passes (profiles/r03_scan_occupancy.log,
//     r03_probe_valu3.log, r03_scan_ablations.log).
// Results are bit-identical in structure to the general kernels (same operation order per state), so the backward pass
// and the checkpoint format are unchanged.
#include <stdlib.h>

#include "scan_fast.h"

// Round 6: the state checkpoints (402 MB of fp32 per direction at stage 0) are written once and read by the backward pass tens of
// milliseconds later: non-temporal stores keep them from evicting the rows the apply pass is about to read.  Measured
// (profiles/r06_scan_nt_policy.log, r06_scan_ckpt_forms.log): three directions per launch 1.27 -> 1.155 ms, one direction unchanged
// (0.44 ms); 16-byte stores of a [quad][dim][4] layout instead of 8-byte pairs: no gain (the cost is bytes, not store width);
// nt on the un-gated y and on the backward's checkpoint loads: nothing.  -DSEGM_CKPT_AUX=0 restores the default policy.
#ifndef SEGM_CKPT_AUX
#define SEGM_CKPT_AUX 2                 // cache policy of the checkpoint stores (2 = nt)
#endif
#ifndef SEGM_OUT_AUX
#define SEGM_OUT_AUX 0                  // ... of the un-gated output y, which only the backward pass reads
#endif

namespace segm {

#ifdef SEGM_SCAN_TIMELINE
__device__ unsigned long long* g_segm_timeline = nullptr;
__device__ unsigned int g_segm_timeline_waves = 0;
#endif

// ------------------------------------------------------------------------------------------------------
// K1 (regular shapes): chunk aggregates.  grid.y = direction (up to kMaxDirs launches of identical geometry in one).
// ------------------------------------------------------------------------------------------------------
// MODE: what the per-step flags are known to be at compile time (they are wave-uniform branches in every step otherwise):
//   0 = read from the arguments,  1 = softplus + gate z + un-gated output kept (training),  2 = softplus + gate z (inference)
// FUSED: `u` holds the INPUT of the causal depthwise conv1d; u_t = SiLU(conv(x)_t + b) is formed here, step by step, from the last
// four inputs (conv_step below) - the north star's "causal depthwise conv1d fused into the same launch".  Opt-in: it adds ~9 vector
// and 2 transcendental instructions per step to passes that are bound by instructions issued, and the conv output is needed in
// memory anyway (x_proj reads all of its channels before any scan step can start).
struct ConvTaps { float w0, w1, w2, w3, b; };            // taps of x[t-3], x[t-2], x[t-1], x[t] (leading ones zero for width < 4)
template <typename T>
__device__ __forceinline__ float conv_step(const ConvTaps& k, float xt, float& xm1, float& xm2, float& xm3) {
    float o = fmaf(k.w3, xt, k.b);                         // the operation order of conv1d_fwd_kernel: bit-identical results
    o = fmaf(k.w0, xm3, o);
    o = fmaf(k.w1, xm2, o);
    o = fmaf(k.w2, xm1, o);
    xm3 = xm2; xm2 = xm1; xm1 = xt;
    o = o * sigmoidf(o);
    return to_f32(from_f32<T>(o));                         // as stored by the separate launch
}
// taps of channel d and the three inputs in front of the chunk that starts at logical step tau0 (zero before the sequence)
template <typename T>
__device__ __forceinline__ void conv_init(const ScanDev& P, const Item& it, int32_t tau0, ConvTaps& k, float& xm1, float& xm2, float& xm3) {
    const int W = P.conv_width;
    const float* wr = P.conv_w + (int64_t)it.d * W;
    k.w3 = wr[W - 1];
    k.w2 = W >= 2 ? wr[W - 2] : 0.f;
    k.w1 = W >= 3 ? wr[W - 3] : 0.f;
    k.w0 = W >= 4 ? wr[W - 4] : 0.f;
    k.b = P.conv_b ? P.conv_b[it.d] : 0.f;
    TimeIter ti;
    ti.seek(P.tm, tau0);
    float h[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ti.prev(P.tm);
        const bool ok = tau0 - 1 - i >= 0;
        const int64_t off = (int64_t)it.b * P.u.sb + row_off(ok ? ti.t : 0, P.u.st) + (int64_t)it.d * P.u.sd;
        const float v = to_f32(reinterpret_cast<const T*>(P.u.p)[off]);
        h[i] = ok ? v : 0.f;
    }
    xm1 = h[0]; xm2 = h[1]; xm3 = h[2];
}

// DTR > 0: delta_t = sum_r dt_w[d][r] dt_x[t][r] formed here (DTR = 4 or 8 = the rank rounded up; columns at or beyond the rank
// count as zero) - the north star's "fused into the same launch" taken one operator further: `delta = dt_proj(x_dbl[:, :R])`
// (reference selective_scan_interface.py:181-182) is no launch and no tensor read any more.  The dt rows of a sub-tile are staged
// like B / C (fetched by the item's own lanes a sub-tile ahead, parked in LDS as fp32, read per step as wave-uniform 16-byte
// reads); the sum is rounded to the element type as the stored tensor would be, so that both passes and the backward (which
// reads the delta the apply pass stores) see the same value.  Opt-in, measured: DESIGN.md section 0, row N1.
template <int RW, int DTR> struct DtStage {
    static constexpr int NE = kFT * (DTR > 0 ? DTR : 1);          // elements of a sub-tile's dt block per item: [step][column]
    static constexpr int E = (NE + RW - 1) / RW;                   // per lane
    rsrc_t rs;
    uint32_t voff[E];
    int32_t lds[E];
    bool ok[E];
    int32_t stb;
};
template <typename T, int RW, int DTR>
__device__ __forceinline__ DtStage<RW, DTR> make_dt_stage(const ScanDev& P, int b_uniform, const WaveRows& w, int r) {
    DtStage<RW, DTR> st;
    st.stb = (int32_t)(P.dt_x.st * (int64_t)sizeof(T));
    st.rs = make_rsrc(P.dt_x.p + ((int64_t)b_uniform * P.dt_x.sb + (int64_t)w.row_lo * P.dt_x.st) * (int64_t)sizeof(T));
#pragma unroll
    for (int i = 0; i < DtStage<RW, DTR>::E; ++i) {
        const int e = r + i * RW;
        const int j = (e / DTR) % kFT, c = e % DTR;
        st.ok[i] = e < DtStage<RW, DTR>::NE && c < P.dt_rank;
        const int32_t lane_steps = w.dT < 0 ? (kFT - 1 - j) * (-w.dT) : j * w.dT;
        st.voff[i] = st.ok[i] ? (uint32_t)(w.lane_row + lane_steps) * (uint32_t)st.stb + (uint32_t)(c * (int)sizeof(T)) : 0u;
        st.lds[i] = j * DTR + c;
    }
    return st;
}
template <typename T, int RW, int DTR, int E>
__device__ __forceinline__ void dt_fetch(uint32_t (&v)[E], const DtStage<RW, DTR>& st, int32_t rows, int32_t dT) {
    const uint32_t so = (uint32_t)(rows + (dT < 0 ? (kFT - 1) * dT : 0)) * (uint32_t)st.stb;
#pragma unroll
    for (int i = 0; i < DtStage<RW, DTR>::E; ++i) v[i] = BufIO<T>::ld_raw(st.rs, st.voff[i], so);
}
template <typename T, int RW, int DTR, int E>
__device__ __forceinline__ void dt_park(const uint32_t (&v)[E], const DtStage<RW, DTR>& st, float* lds_item, int r) {
#pragma unroll
    for (int i = 0; i < DtStage<RW, DTR>::E; ++i)
        if (r + i * RW < DtStage<RW, DTR>::NE) lds_item[st.lds[i]] = st.ok[i] ? BufIO<T>::cvt_raw(v[i]) : 0.f;
}
// delta of one step from its staged dt row (wave-uniform reads) and the lane's weights, rounded as the stored tensor
template <typename T, int DTR>
__device__ __forceinline__ float dt_delta(const float* row, const float (&w)[DTR > 0 ? DTR : 1]) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < DTR / 4; ++q) {
        const float4 x = reinterpret_cast<const float4*>(row)[q];
        s = fmaf(w[4 * q], x.x, s);
        s = fmaf(w[4 * q + 1], x.y, s);
        s = fmaf(w[4 * q + 2], x.z, s);
        s = fmaf(w[4 * q + 3], x.w, s);
    }
    return to_f32(from_f32<T>(s));
}

template <typename T, int RW, int MODE, bool FUSED, int DTR>
__global__ void __launch_bounds__(kBlock) scan_fwd_agg_fast_kernel(ScanDevN PP) {
    constexpr int G = 64 / RW, EPL = StageStream<RW>::EPL;
    __shared__ __attribute__((aligned(16))) float s_b[2][kWavesPerBlock][G][kFT * kFS];
    __shared__ __attribute__((aligned(16))) float s_dt[2][kWavesPerBlock][G][DTR > 0 ? kFT * DTR : 4];
    SEGM_TL_DECL();
    const ScanDev& P = PP.d[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Geom& gm = P.gm;
    const Item it = locate(gm, (int64_t)blockIdx.x * kWavesPerBlock + wave, lane);
    if (!it.wave_valid) return;                           // the last workgroup may have spare waves (no workgroup barriers here)
    const int ub = uniform_batch(it);
    const bool softplus_on = MODE != 0 || P.delta_softplus != 0;
    FastClock ck;
    ck.init(P.tm);
    const WaveRows wr = wave_rows(P.tm, gm, it);

    f2 A2[kFS / 2], h[kFS / 2];
#pragma unroll
    for (int n = 0; n < kFS / 2; ++n) {
        A2[n] = f2{P.A[(int64_t)it.d * kFS + 2 * n] * kLog2e, P.A[(int64_t)it.d * kFS + 2 * n + 1] * kLog2e};
        h[n] = f2{0.f, 0.f};
    }
    const float bias = P.delta_bias ? P.delta_bias[it.d] : 0.f;
    const Stream up = make_stream<T>(P.u, ub, wr, it.d);
    const Stream dp = make_stream<T>(P.delta, ub, wr, it.d);
    const StageStream<RW> sb = make_stage<T, RW>(P.Bm, ub, wr, it.r);
    ConvTaps ck4 = {0.f, 0.f, 0.f, 0.f, 0.f};
    float xm1 = 0.f, xm2 = 0.f, xm3 = 0.f;
    if constexpr (FUSED) conv_init<T>(P, it, it.chunk * gm.chunk, ck4, xm1, xm2, xm3);
    typedef DtStage<RW, DTR> DS;
    DS ds;
    float wdt[DTR > 0 ? DTR : 1];
    uint32_t ndt[DS::E];
    if constexpr (DTR > 0) {
        ds = make_dt_stage<T, RW, DTR>(P, ub, wr, it.r);
#pragma unroll
        for (int c = 0; c < DTR; ++c) wdt[c] = c < P.dt_rank ? P.dt_w[(int64_t)it.d * P.dt_rank + c] : 0.f;
    }

    // the B rows first: at the loop head they are then the oldest loads on both paths into it (vmcnt is in order, and the
    // compiler merges the pending-load state of the prologue with that of the back edge - with the stage loads issued last here
    // every sub-tile would begin with s_waitcnt vmcnt(0))
    float nu[kFT], nd[kFT], nb[EPL];
    stage_fetch_buf<T, RW>(nb, sb, wr.bias + ck.U, wr.dT);
    if constexpr (DTR > 0) dt_fetch<T>(ndt, ds, wr.bias + ck.U, wr.dT);
    __builtin_amdgcn_sched_barrier(0);                    // keep that order
    stream_fetch<T>(nu, up, wr.bias + ck.U, wr.dT);
    if constexpr (DTR == 0) stream_fetch<T>(nd, dp, wr.bias + ck.U, wr.dT);

    float sumd = 0.f;
    int buf = 0;
    const int nsub = gm.chunk / kFT;
    SEGM_TL_STAMP(1);
    for (int s = 0; s < nsub; ++s) {
#ifdef SEGM_SCAN_TIMELINE
        if (s == 1) SEGM_TL_STAMP(2);
#endif
        float* lb = &s_b[buf][wave][it.gi][0];
        float* ldt = &s_dt[buf][wave][it.gi][0];
        stage_park_buf<RW>(nb, sb, lb);
        if constexpr (DTR > 0) dt_park<T>(ndt, ds, ldt, it.r);
        SEGM_WAVE_LDS_SYNC();
        // the row streams are a ring of kFT registers each: step j's value is consumed and its register refilled with step j of
        // the NEXT sub-tile in the same step (after the last sub-tile: re-read this one, never past the chunk)
        const int32_t Un = wr.bias + ((s + 1 < nsub) ? ck.next_U() : ck.U);
        stage_fetch_buf<T, RW>(nb, sb, Un, wr.dT);
        if constexpr (DTR > 0) dt_fetch<T>(ndt, ds, Un, wr.dT);
        ck.advance();
        uint32_t su = (uint32_t)Un * (uint32_t)up.stb, sd = (uint32_t)Un * (uint32_t)dp.stb;
        const uint32_t iu = (uint32_t)(wr.dT * up.stb), id = (uint32_t)(wr.dT * dp.stb);
        float4 bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = reinterpret_cast<const float4*>(lb)[q];
#pragma unroll
        for (int j = 0; j < kFT; ++j) {
            float uu = nu[j];
            float dl;
            if constexpr (DTR > 0) dl = dt_delta<T, DTR>(ldt + j * DTR, wdt) + bias;
            else dl = nd[j] + bias;
            nu[j] = BufIO<T>::ld(up.rs, up.voff, su);
            if constexpr (DTR == 0) nd[j] = BufIO<T>::ld(dp.rs, dp.voff, sd);
            su += iu;
            sd += id;
            if constexpr (FUSED) uu = conv_step<T>(ck4, uu, xm1, xm2, xm3);
            dl = softplus_on ? softplus20(dl) : dl;
            sumd += dl;
            // both halves written: a packed operand with op_sel broadcast reads the odd register of its pair too, and that register
            // may be the destination of a ring load in flight (a false dependency the compiler then waits on)
            f2 dl2 = {dl, dl}, dlu2 = {dl * uu, dl * uu};
            SEGM_PIN_F2(dl2);
            SEGM_PIN_F2(dlu2);
            float4 bn[4];
            if (j + 1 < kFT) {
#pragma unroll
                for (int q = 0; q < 4; ++q) bn[q] = reinterpret_cast<const float4*>(lb + (j + 1) * kFS)[q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f2 b0 = {bq[q].x, bq[q].y}, b1 = {bq[q].z, bq[q].w};
                const f2 da0 = A2[2 * q] * dl2, da1 = A2[2 * q + 1] * dl2;
                const f2 a0 = {fast_exp2(da0.x), fast_exp2(da0.y)};
                const f2 a1 = {fast_exp2(da1.x), fast_exp2(da1.y)};
                h[2 * q] = a0 * h[2 * q] + b0 * dlu2;
                h[2 * q + 1] = a1 * h[2 * q + 1] + b1 * dlu2;
            }
#pragma unroll
            for (int n = 0; n < kFS / 2; ++n) SEGM_PIN_F2(h[n]);      // finish this step before the LDS reads two steps ahead
            if (j + 1 < kFT) {
#pragma unroll
                for (int q = 0; q < 4; ++q) bq[q] = bn[q];
            }
        }
        buf ^= 1;
    }
    SEGM_TL_STAMP(3);
    const int64_t row = (int64_t)it.b * gm.nchunks + it.chunk;
    P.agg_sd[row * gm.dim + it.d] = sumd;
#pragma unroll
    for (int n = 0; n < kFS / 2; ++n) {
        P.agg_h[(row * kFS + 2 * n) * gm.dim + it.d] = h[n].x;
        P.agg_h[(row * kFS + 2 * n + 1) * gm.dim + it.d] = h[n].y;
    }
    SEGM_TL_END(0, (blockIdx.y * gridDim.x + blockIdx.x) * kWavesPerBlock + wave, lane);
}

// ------------------------------------------------------------------------------------------------------
// K3 (regular shapes): apply
// ------------------------------------------------------------------------------------------------------
template <typename T, int RW, int MODE, bool FUSED, int DTR>
__global__ void __launch_bounds__(kBlock, SEGM_FAST_MIN_WAVES) scan_fwd_apply_fast_kernel(ScanDevN PP) {
    constexpr int G = 64 / RW, EPL = StageStream<RW>::EPL;
    __shared__ __attribute__((aligned(16))) float s_bc[2][kWavesPerBlock][G][2][kFT * kFS];
    __shared__ __attribute__((aligned(16))) float s_dt[2][kWavesPerBlock][G][DTR > 0 ? kFT * DTR : 4];
    SEGM_TL_DECL();
    const ScanDev& P = PP.d[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Geom& gm = P.gm;
    const Item it = locate(gm, (int64_t)blockIdx.x * kWavesPerBlock + wave, lane);
    if (!it.wave_valid) return;                           // the last workgroup may have spare waves (no workgroup barriers here)
    const int ub = uniform_batch(it);
    const bool softplus_on = MODE != 0 || P.delta_softplus != 0;
    const bool has_z = MODE != 0 || P.z.p != nullptr, has_out = MODE == 1 || (MODE == 0 && P.out.p != nullptr);
    FastClock ck;
    ck.init(P.tm);
    const int32_t tau0 = it.chunk * gm.chunk;
    const WaveRows wr = wave_rows(P.tm, gm, it);

    f2 A2[kFS / 2], h[kFS / 2];
    const int64_t crow = (int64_t)it.b * gm.nchunks + it.chunk;
#pragma unroll
    for (int n = 0; n < kFS / 2; ++n) {
        A2[n] = f2{P.A[(int64_t)it.d * kFS + 2 * n] * kLog2e, P.A[(int64_t)it.d * kFS + 2 * n + 1] * kLog2e};
        h[n] = f2{P.carry[(crow * kFS + 2 * n) * gm.dim + it.d], P.carry[(crow * kFS + 2 * n + 1) * gm.dim + it.d]};
    }
    const float bias = P.delta_bias ? P.delta_bias[it.d] : 0.f;
    const float Dv = P.D ? P.D[it.d] : 0.f;
    ConvTaps ck4 = {0.f, 0.f, 0.f, 0.f, 0.f};
    float xm1 = 0.f, xm2 = 0.f, xm3 = 0.f;
    if constexpr (FUSED) conv_init<T>(P, it, tau0, ck4, xm1, xm2, xm3);
    const Stream up = make_stream<T>(P.u, ub, wr, it.d);
    const Stream dp = make_stream<T>(P.delta, ub, wr, it.d);
    const Stream zp = make_stream<T>(has_z ? P.z : P.u, ub, wr, it.d);            // without a gate: aliases u, unused
    const Stream op = make_stream<T>(has_out ? P.out : P.u, ub, wr, it.d);
    const Stream ozp = make_stream<T>(has_z ? P.out_z : P.u, ub, wr, it.d);
    const StageStream<RW> sb = make_stage<T, RW>(P.Bm, ub, wr, it.r);
    const StageStream<RW> sc = make_stage<T, RW>(P.Cm, ub, wr, it.r);
    // checkpoints: state entering step kCkpt k of the sequence, [batch][nck][8 state pairs][dim][2]; the wave's window as a
    // buffer (base = first checkpoint row of the wave's lowest chunk), one scalar offset per checkpoint and pair: eight stores of
    // 64 lanes x 8 contiguous bytes.  (Round 4 measured the alternatives at one checkpoint per 8 steps, stage 0: sixteen 4-byte
    // stores per lane 280 us, four 16-byte stores of a state-fastest layout - 64-byte lane stride - 334 us;
    // profiles/r04_scan_kernels_*.txt)
    const int32_t chunk0 = __builtin_amdgcn_readfirstlane(it.chunk - it.gi);
    const rsrc_t ckr = make_rsrc(P.ckpt ? P.ckpt + (((int64_t)ub * P.nck + (int64_t)chunk0 * (gm.chunk / kCkpt)) * kFS) * gm.dim : nullptr);
    const uint32_t ck_voff = ((uint32_t)(it.gi * (gm.chunk / kCkpt)) * kFS * (uint32_t)gm.dim + 2u * (uint32_t)it.d) * 4u;
    const int32_t ck_pair = gm.dim * 8;                   // bytes between consecutive state pairs of one checkpoint

    typedef DtStage<RW, DTR> DS;
    DS ds;
    float wdt[DTR > 0 ? DTR : 1];
    uint32_t ndt[DS::E];
    if constexpr (DTR > 0) {
        ds = make_dt_stage<T, RW, DTR>(P, ub, wr, it.r);
#pragma unroll
        for (int c = 0; c < DTR; ++c) wdt[c] = c < P.dt_rank ? P.dt_w[(int64_t)it.d * P.dt_rank + c] : 0.f;
    }
    float nu[kFT], nd[kFT], nz[kFT], nb[EPL], nc[EPL];
    stage_fetch_buf<T, RW>(nb, sb, wr.bias + ck.U, wr.dT);        // oldest loads at the loop head (see the aggregate kernel)
    stage_fetch_buf<T, RW>(nc, sc, wr.bias + ck.U, wr.dT);
    if constexpr (DTR > 0) dt_fetch<T>(ndt, ds, wr.bias + ck.U, wr.dT);
    __builtin_amdgcn_sched_barrier(0);                    // keep that order
    stream_fetch<T>(nu, up, wr.bias + ck.U, wr.dT);
    if constexpr (DTR == 0) stream_fetch<T>(nd, dp, wr.bias + ck.U, wr.dT);
    stream_fetch<T>(nz, zp, wr.bias + ck.U, wr.dT);

    int buf = 0;
    const int nsub = gm.chunk / kFT;
    SEGM_TL_STAMP(1);
    for (int s = 0; s < nsub; ++s) {
#ifdef SEGM_SCAN_TIMELINE
        if (s == 1) SEGM_TL_STAMP(2);
#endif
        float* lb = &s_bc[buf][wave][it.gi][0][0];
        float* lc = &s_bc[buf][wave][it.gi][1][0];
        float* ldt = &s_dt[buf][wave][it.gi][0];
        stage_park_buf<RW>(nb, sb, lb);
        stage_park_buf<RW>(nc, sc, lc);
        if constexpr (DTR > 0) dt_park<T>(ndt, ds, ldt, it.r);
        SEGM_WAVE_LDS_SYNC();
        const int32_t Uc = wr.bias + ck.U;
        const int32_t Un = wr.bias + ((s + 1 < nsub) ? ck.next_U() : ck.U);
        stage_fetch_buf<T, RW>(nb, sb, Un, wr.dT);
        stage_fetch_buf<T, RW>(nc, sc, Un, wr.dT);
        if constexpr (DTR > 0) dt_fetch<T>(ndt, ds, Un, wr.dT);
        ck.advance();
        // row streams: rings of kFT registers, refilled step by step with the next sub-tile's rows (see the aggregate kernel)
        uint32_t su = (uint32_t)Un * (uint32_t)up.stb, sd = (uint32_t)Un * (uint32_t)dp.stb, sz = (uint32_t)Un * (uint32_t)zp.stb;
        const uint32_t iu = (uint32_t)(wr.dT * up.stb), id = (uint32_t)(wr.dT * dp.stb), iz = (uint32_t)(wr.dT * zp.stb);
        if (P.ckpt) {                                      // kCkpt = one sub-tile
#ifdef SEGM_CKPT_QUAD_EXPERIMENT
            // timing experiment only (the backward still reads pairs): [quad][dim][4], four 16-byte stores of 64 lanes x 16 bytes
            uint32_t kso = (uint32_t)(s * (kFS / 2) * ck_pair);
#pragma unroll
            for (int n = 0; n < kFS / 4; ++n) {
                typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                const u32x4_t v = {__float_as_uint(h[2 * n].x), __float_as_uint(h[2 * n].y), __float_as_uint(h[2 * n + 1].x), __float_as_uint(h[2 * n + 1].y)};
                __builtin_amdgcn_raw_buffer_store_b128(v, ckr, ck_voff + 8u * (uint32_t)it.d, kso, SEGM_CKPT_AUX);
                kso += 2u * (uint32_t)ck_pair;
            }
#else
            uint32_t kso = (uint32_t)(s * (kFS / 2) * ck_pair);
#pragma unroll
            for (int n = 0; n < kFS / 2; ++n) {
                typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
                const u32x2_t v = {__float_as_uint(h[n].x), __float_as_uint(h[n].y)};
                __builtin_amdgcn_raw_buffer_store_b64(v, ckr, ck_voff, kso, SEGM_CKPT_AUX);
                kso += (uint32_t)ck_pair;
            }
#endif
        }
        uint32_t oso = (uint32_t)Uc * (uint32_t)op.stb, ozso = (uint32_t)Uc * (uint32_t)ozp.stb;     // running scalar offsets
        const uint32_t oinc = (uint32_t)(wr.dT * op.stb), ozinc = (uint32_t)(wr.dT * ozp.stb);
        uint32_t dso = (uint32_t)Uc * (uint32_t)dp.stb;    // DTR: delta is written here, for the backward
#pragma unroll
        for (int j = 0; j < kFT; ++j) {
            float uu = nu[j];
            const float zz = nz[j];
            float dl;
            if constexpr (DTR > 0) {
                const float draw = dt_delta<T, DTR>(ldt + j * DTR, wdt);
                BufIO<T>::st(dp.rs, dp.voff, dso, draw);
                dso += id;
                dl = draw + bias;
            } else {
                dl = nd[j] + bias;
            }
            nu[j] = BufIO<T>::ld(up.rs, up.voff, su);
            if constexpr (DTR == 0) nd[j] = BufIO<T>::ld(dp.rs, dp.voff, sd);
            nz[j] = BufIO<T>::ld(zp.rs, zp.voff, sz);          // without a gate the stream aliases u: loaded, never used
            su += iu;
            sd += id;
            sz += iz;
            if constexpr (FUSED) uu = conv_step<T>(ck4, uu, xm1, xm2, xm3);
            dl = softplus_on ? softplus20(dl) : dl;
            f2 dl2 = {dl, dl}, dlu2 = {dl * uu, dl * uu};      // real pairs, not op_sel broadcasts (see the aggregate kernel)
            SEGM_PIN_F2(dl2);
            SEGM_PIN_F2(dlu2);
            // this step's B / C rows (wave-uniform addresses: LDS broadcast reads, 4 LDS cycles each); the read latency is covered
            // by the other resident waves, so nothing is prefetched into registers
            float4 bq[4], cq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bq[q] = reinterpret_cast<const float4*>(lb + j * kFS)[q];
                cq[q] = reinterpret_cast<const float4*>(lc + j * kFS)[q];
            }
            f2 ya = {Dv * uu, 0.f}, yb = {0.f, 0.f};       // two independent accumulation chains
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f2 b0 = {bq[q].x, bq[q].y}, b1 = {bq[q].z, bq[q].w};
                const f2 c0 = {cq[q].x, cq[q].y}, c1 = {cq[q].z, cq[q].w};
                const f2 da0 = A2[2 * q] * dl2, da1 = A2[2 * q + 1] * dl2;
                const f2 a0 = {fast_exp2(da0.x), fast_exp2(da0.y)};
                const f2 a1 = {fast_exp2(da1.x), fast_exp2(da1.y)};
                h[2 * q] = a0 * h[2 * q] + b0 * dlu2;
                h[2 * q + 1] = a1 * h[2 * q + 1] + b1 * dlu2;
                ya = c0 * h[2 * q] + ya;
                yb = c1 * h[2 * q + 1] + yb;
            }
            const float y = (ya.x + yb.x) + (ya.y + yb.y);
            if (has_out) BufIO<T>::template st<SEGM_OUT_AUX>(op.rs, op.voff, oso, y);
            if (has_z) BufIO<T>::st(ozp.rs, ozp.voff, ozso, y * zz * sigmoidf(zz));
            oso += oinc;
            ozso += ozinc;
#pragma unroll
            for (int n = 0; n < kFS / 2; ++n) SEGM_PIN_F2(h[n]);      // keeps the LDS reads of later steps from being hoisted here
        }
        buf ^= 1;
    }
    SEGM_TL_STAMP(3);
    SEGM_TL_END(1, (blockIdx.y * gridDim.x + blockIdx.x) * kWavesPerBlock + wave, lane);
    (void)tau0;
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
bool scan_fast_shape(const ScanDev& P) {
    const Geom& gm = P.gm;
    if (gm.nstate != kFS) return false;
    if (gm.dim % gm.rw != 0 || gm.L % gm.chunk != 0 || gm.nchunks % gm.g != 0) return false;
    if (gm.chunk % kFT != 0) return false;
    if (P.tm.ns > 1 && (P.tm.ns % kFT != 0 || gm.chunk % P.tm.ns != 0)) return false;
    return true;
}

template <typename T, int RW, int MODE, bool FUSED, int DTR = 0>
static void launch_fast_mode(const ScanDevN& PP, int ndir, bool apply, hipStream_t stream) {
    const unsigned nblocks = (unsigned)((PP.d[0].gm.nwaves + kWavesPerBlock - 1) / kWavesPerBlock);
    if (apply) hipLaunchKernelGGL((scan_fwd_apply_fast_kernel<T, RW, MODE, FUSED, DTR>), dim3(nblocks, ndir), dim3(kBlock), 0, stream, PP);
    else hipLaunchKernelGGL((scan_fwd_agg_fast_kernel<T, RW, MODE, FUSED, DTR>), dim3(nblocks, ndir), dim3(kBlock), 0, stream, PP);
}
// the compile-time flag set every direction of the launch agrees with (0 = none: flags read per step)
static int fast_mode(const ScanDevN& PP, int ndir, bool apply) {
    static const bool off = [] { const char* e = getenv("SEGM_SCAN_FLAGS"); return e && atoi(e) == 0; }();     // experiments only
    if (off) return 0;
    int mode = -1;
    for (int i = 0; i < ndir; ++i) {
        const ScanDev& P = PP.d[i];
        int m = 0;
        if (P.delta_softplus != 0 && (!apply || P.z.p != nullptr)) m = (!apply || P.out.p != nullptr) ? 1 : 2;
        if (mode >= 0 && m != mode) return 0;
        mode = m;
    }
    return mode < 0 ? 0 : mode;
}
template <typename T, int RW>
static void launch_fast_rw(const ScanDevN& PP, int ndir, bool apply, hipStream_t stream) {
    const int mode = fast_mode(PP, ndir, apply);
    if (PP.d[0].dt_rank != 0) {                            // dt_proj inside the passes (flags read per step unless Mamba's training set)
        const bool m1 = mode == 1 || (mode == 2 && !apply);
        if (PP.d[0].dt_rank <= 4) { if (m1) launch_fast_mode<T, RW, 1, false, 4>(PP, ndir, apply, stream); else launch_fast_mode<T, RW, 0, false, 4>(PP, ndir, apply, stream); }
        else { if (m1) launch_fast_mode<T, RW, 1, false, 8>(PP, ndir, apply, stream); else launch_fast_mode<T, RW, 0, false, 8>(PP, ndir, apply, stream); }
        return;
    }
    if (PP.d[0].conv_width != 0) {                         // conv1d inside the passes
        if (mode == 1 || (mode == 2 && !apply)) launch_fast_mode<T, RW, 1, true>(PP, ndir, apply, stream);
        else if (mode == 2) launch_fast_mode<T, RW, 2, true>(PP, ndir, apply, stream);
        else launch_fast_mode<T, RW, 0, true>(PP, ndir, apply, stream);
        return;
    }
    if (mode == 1) launch_fast_mode<T, RW, 1, false>(PP, ndir, apply, stream);
    else if (mode == 2 && apply) launch_fast_mode<T, RW, 2, false>(PP, ndir, apply, stream);
    else if (mode == 2) launch_fast_mode<T, RW, 1, false>(PP, ndir, apply, stream);          // the aggregate pass only has the softplus flag
    else launch_fast_mode<T, RW, 0, false>(PP, ndir, apply, stream);
}
template <typename T>
static void launch_fast_t(const ScanDevN& PP, int ndir, bool apply, hipStream_t stream) {
    if (PP.d[0].gm.rw == 64) launch_fast_rw<T, 64>(PP, ndir, apply, stream);
    else if (PP.d[0].gm.rw == 32) launch_fast_rw<T, 32>(PP, ndir, apply, stream);
    else launch_fast_rw<T, 16>(PP, ndir, apply, stream);
}
// launches K1 (apply == false) or K3 (apply == true) of the regular-shape path for `ndir` argument blocks of one geometry
void launch_scan_fwd_fast(const ScanDevN& PP, int ndir, int dtype, bool apply, hipStream_t stream) {
    if (dtype == SEGM_F32) launch_fast_t<float>(PP, ndir, apply, stream);
    else if (dtype == SEGM_F16) launch_fast_t<f16_t>(PP, ndir, apply, stream);
    else launch_fast_t<bf16_t>(PP, ndir, apply, stream);
}

}  // namespace segm

#ifdef SEGM_SCAN_TIMELINE
// experiments only (tools/gpu_scan_timeline.py): installs the side buffer of the per-wave stamps, [2 kernels][waves][8] u64
extern "C" int segm_debug_set_timeline(void* buf, unsigned int waves) {
    unsigned long long* p = (unsigned long long*)buf;
    if (hipMemcpyToSymbol(HIP_SYMBOL(segm::g_segm_timeline), &p, sizeof(p)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(segm::g_segm_timeline_waves), &waves, sizeof(waves)) != hipSuccess) return -1;
    return 0;
}
#endif
