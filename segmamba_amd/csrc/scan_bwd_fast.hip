// Selective scan backward, the "regular shape" kernels (same algorithm, workspace and checkpoints as scan_bwd.hip's
// K1 / K3; see scan_fwd_fast.hip for what "regular" means - here additionally B, C, dB, dC must be state-fastest and the
// sequence tensors channel-contiguous with 16-byte aligned rows: the channel-last production layout).
//
// Round 3 rewrite of the main kernel.  Round 2's version needed 256 VGPRs + 208 AGPRs (one wave per SIMD, a quarter of its
// instructions v_accvgpr moves, another tenth 64-bit address arithmetic) and fetched every window behind exposed latency:
// VALU busy 64 % of a launch whose arithmetic is 55 % of the instructions.  This one is built to fit two waves per SIMD:
//   * addressing through buffer resources (scan_fast.h): a scalar offset per row, no vector address arithmetic;
//   * the five row streams of the NEXT window (u, delta, dout, z, out: 16 steps x RW channels each) are fetched during the
//     current window's state loop as 16-byte pieces by all lanes of the wave - two loads per lane and stream instead of 80
//     two-byte loads - and parked in an LDS tile the next window's prologue reads its 16 steps from; B / C of the next window
//     likewise (raw, per-lane elements).  Nothing of a window's input is waited for inside the window;
//   * the adjoint carried between windows and the dA accumulators (16 + 16 floats per lane) are register arrays indexed by the
//     state-pair loop counter (s_set_gpr_idx), not LDS: the LDS footprint is 16 KB per wave (two workgroups per CU);
//   * q = <dh, B> and the delta-gradient sums are scalars (two plain fma per step and pair instead of two packed ones on
//     pair accumulators: 32 registers less);
//   * the channel sums of dB / dC leave the reduce-scatter straight to memory (the lane that ends with the finished value
//     stores it): no LDS tile, no flush phase.
// Per (step, state): forward  a = exp2(delta A2), h = a h_prev + (delta u) B;  backward  dh = g C + e, en = a dh,
//   t2 = en h_prev, dA += t2 delta, q += dh B, ddelta += t2 A, dB_c = dh (delta u), dC_c = g h - 11 packed half-ops + 1 v_exp.
#include <stdlib.h>

#include "scan_fast.h"

namespace segm {

constexpr int kFW = 16;     // window = spacing of the forward checkpoints

// ------------------------------------------------------------------------------------------------------
// K1 (regular shapes): reverse chunk aggregates.  grid.y = direction.
// ------------------------------------------------------------------------------------------------------
template <typename T, int RW>
__global__ void __launch_bounds__(kBlock) scan_bwd_agg_fast_kernel(ScanDevN PP) {
    constexpr int G = 64 / RW, EPL = StageStream<RW>::EPL;
    __shared__ __attribute__((aligned(16))) float s_c[2][kWavesPerBlock][G][kFT * kFS];
    const ScanDev& P = PP.d[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Geom& gm = P.gm;
    const Item it = locate(gm, (int64_t)blockIdx.x * kWavesPerBlock + wave, lane);
    if (!it.wave_valid) return;                           // the last workgroup may have spare waves (no workgroup barriers here)
    const int ub = uniform_batch(it);
    const bool softplus_on = P.delta_softplus != 0;
    const bool has_z = P.z.p != nullptr;
    const WaveRows wr = wave_rows(P.tm, gm, it);

    f2 A2[kFS / 2], e[kFS / 2];
#pragma unroll
    for (int n = 0; n < kFS / 2; ++n) {
        A2[n] = f2{P.A[(int64_t)it.d * kFS + 2 * n] * kLog2e, P.A[(int64_t)it.d * kFS + 2 * n + 1] * kLog2e};
        e[n] = f2{0.f, 0.f};
    }
    const float bias = P.delta_bias ? P.delta_bias[it.d] : 0.f;
    const Stream dp = make_stream<T>(P.delta, ub, wr, it.d);
    const Stream gp = make_stream<T>(P.dout, ub, wr, it.d);
    const Stream zp = make_stream<T>(has_z ? P.z : P.dout, ub, wr, it.d);
    const StageStream<RW> sc = make_stage<T, RW>(P.Cm, ub, wr, it.r);

    const int nsub = gm.chunk / kFT;
    float nd[kFT], ng[kFT], nz[kFT], nc[EPL];
    {
        const int32_t U = wr.bias + fast_U_of(P.tm, nsub - 1);
        stream_fetch<T>(nd, dp, U, wr.dT);
        stream_fetch<T>(ng, gp, U, wr.dT);
        stream_fetch<T>(nz, zp, U, wr.dT);
        stage_fetch_buf<T, RW>(nc, sc, U, wr.dT);
    }
    float sumd = 0.f;
    int buf = 0;
    for (int s = nsub - 1; s >= 0; --s) {
        float* lc = &s_c[buf][wave][it.gi][0];
        stage_park_buf<RW>(nc, sc, lc);
        SEGM_WAVE_LDS_SYNC();
        float cd[kFT], cg[kFT], cz[kFT];
#pragma unroll
        for (int j = 0; j < kFT; ++j) { cd[j] = nd[j]; cg[j] = ng[j]; cz[j] = nz[j]; }
        const int32_t Un = wr.bias + fast_U_of(P.tm, s > 0 ? s - 1 : 0);     // prefetch the next (lower) sub-tile
        stream_fetch<T>(nd, dp, Un, wr.dT);
        stream_fetch<T>(ng, gp, Un, wr.dT);
        stream_fetch<T>(nz, zp, Un, wr.dT);
        stage_fetch_buf<T, RW>(nc, sc, Un, wr.dT);
#pragma unroll
        for (int jj = 0; jj < kFT; ++jj) {
            const int j = kFT - 1 - jj;
            float dl = cd[j] + bias;
            dl = softplus_on ? softplus20(dl) : dl;
            sumd += dl;
            float g = cg[j];
            if (has_z) { const float zz = cz[j]; g *= zz * sigmoidf(zz); }
            float4 cq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) cq[q] = reinterpret_cast<const float4*>(lc + j * kFS)[q];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f2 c0 = {cq[q].x, cq[q].y}, c1 = {cq[q].z, cq[q].w};
                const f2 da0 = A2[2 * q] * dl, da1 = A2[2 * q + 1] * dl;
                const f2 a0 = {fast_exp2(da0.x), fast_exp2(da0.y)};
                const f2 a1 = {fast_exp2(da1.x), fast_exp2(da1.y)};
                e[2 * q] = a0 * (c0 * g + e[2 * q]);
                e[2 * q + 1] = a1 * (c1 * g + e[2 * q + 1]);
            }
#pragma unroll
            for (int n = 0; n < kFS / 2; ++n) SEGM_PIN_F2(e[n]);
        }
        buf ^= 1;
    }
    const int64_t row = (int64_t)it.b * gm.nchunks + it.chunk;
    P.agg_sd[row * gm.dim + it.d] = sumd;
#pragma unroll
    for (int n = 0; n < kFS / 2; ++n) {
        P.agg_h[(row * kFS + 2 * n) * gm.dim + it.d] = e[n].x;
        P.agg_h[(row * kFS + 2 * n + 1) * gm.dim + it.d] = e[n].y;
    }
}

// ------------------------------------------------------------------------------------------------------
// K3 (regular shapes): main backward kernel
// ------------------------------------------------------------------------------------------------------
// One row stream of a wave as 16-byte pieces: a half window (8 steps x RW channels x G items = 64 channels-rows of 8 steps)
// is 64 * sizeof(T) / 2 pieces, i.e. sizeof(T) / 2 loads per lane.  Piece pc of a half window: item = pc / (8 PPR),
// step = (pc % (8 PPR)) / PPR, 16-byte part = pc % PPR, with PPR = RW sizeof(T) / 16 pieces per row.  In LDS a half window is
// the same pieces in order, i.e. [item][step][RW channels].
template <typename T, int RW> struct TileStream {
    static constexpr int NLH = (int)sizeof(T) / 2;        // loads per lane and half window
    static constexpr int PPR = RW * (int)sizeof(T) / 16;
    static constexpr int HALF_BYTES = 64 * NLH * 16;      // one half window of one stream in LDS
    rsrc_t rs;
    uint32_t voff[NLH];
    int32_t stb;
};
template <typename T, int RW>
__device__ __forceinline__ TileStream<T, RW> make_tile(const Seq& s, int b_uniform, const TimeMap& tm, const Geom& gm,
                                                        const WaveRows& w, int32_t chunk0, int dt, int lane) {
    typedef TileStream<T, RW> TS;
    TS t;
    t.stb = (int32_t)(s.st * (int64_t)sizeof(T));
    t.rs = make_rsrc(s.p + ((int64_t)b_uniform * s.sb + (int64_t)w.row_lo * s.st + (int64_t)dt * RW * s.sd) * (int64_t)sizeof(T));
#pragma unroll
    for (int q = 0; q < TS::NLH; ++q) {
        const int pc = lane + 64 * q;
        const int item = pc / (8 * TS::PPR), rem = pc % (8 * TS::PPR), jl = rem / TS::PPR, part = rem % TS::PPR;
        const int32_t item_row = fast_item_row(tm, (chunk0 + item) * gm.chunk) - w.bias - w.row_lo;     // >= 0
        const int32_t lane_steps = w.dT < 0 ? (kFT - 1 - jl) * (-w.dT) : jl * w.dT;
        t.voff[q] = (uint32_t)(item_row + lane_steps) * (uint32_t)t.stb + (uint32_t)part * 16u;
    }
    return t;
}
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
// issue the loads of half window `h` (uniform rows of its first step: `rows`) ...
template <typename T, int RW, int N>
__device__ __forceinline__ void tile_issue(u32x4_t (&v)[N], const TileStream<T, RW>& t, int32_t rows, int32_t dT) {
    const uint32_t s0 = (uint32_t)(rows + (dT < 0 ? (kFT - 1) * dT : 0)) * (uint32_t)t.stb;
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = __builtin_amdgcn_raw_buffer_load_b128(t.rs, t.voff[q], s0, 0);
}
// ... and park them: `half` points at the half window's LDS bytes
template <int N>
__device__ __forceinline__ void tile_park(const u32x4_t (&v)[N], char* half, int lane) {
#pragma unroll
    for (int q = 0; q < N; ++q) *reinterpret_cast<u32x4_t*>(half + (lane + 64 * q) * 16) = v[q];
}
template <typename T> __device__ __forceinline__ float lds_elem(const char* p) { return to_f32(*reinterpret_cast<const T*>(p)); }
// the raw bits of an element (BufIO::ld_raw) into LDS
__device__ __forceinline__ void put_raw(float* p, uint32_t bits) { *reinterpret_cast<uint32_t*>(p) = bits; }
template <typename T> __device__ __forceinline__ void put_raw(T* p, uint32_t bits) { *reinterpret_cast<unsigned short*>(p) = (unsigned short)bits; }

#ifndef SEGM_BWD_MAIN_WAVES
#define SEGM_BWD_MAIN_WAVES 2
#endif

template <typename T, int RW>
__global__ void __launch_bounds__(kBlock, SEGM_BWD_MAIN_WAVES) scan_bwd_main_fast_kernel(ScanDevN PP) {
    constexpr int G = 64 / RW, EPL = StageStream<RW>::EPL;
    constexpr int V = RW < 32 ? RW : 32;
    typedef TileStream<T, RW> TS;
    constexpr int NSTREAM = 5;                             // u, delta, dout, z, out
    constexpr int STREAM_BYTES = 2 * TS::HALF_BYTES;      // a window of one stream
    __shared__ __attribute__((aligned(16))) char s_tile[kWavesPerBlock][NSTREAM * STREAM_BYTES];           // next window's rows
    __shared__ __attribute__((aligned(16))) T s_raw[kWavesPerBlock][G][2][kFW * kFS];                        // next window's B, C (raw)
    __shared__ __attribute__((aligned(16))) float s_bc[kWavesPerBlock][G][2][kFW * kFS];                     // this window's B, C
    __shared__ __attribute__((aligned(16))) float s_dbc[kWavesPerBlock][G][2][kFW * (kFS / 2)];              // dB, dC of 8 states: [step][state % 8]
    const ScanDev& P = PP.d[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Geom& gm = P.gm;
    const Item it = locate(gm, (int64_t)blockIdx.x * kWavesPerBlock + wave, lane);
    if (!it.wave_valid) return;                           // the last workgroup may have spare waves (no workgroup barriers here)
    const int ub = uniform_batch(it);
    const bool softplus_on = P.delta_softplus != 0;
    const bool has_z = P.z.p != nullptr;
    const WaveRows wr = wave_rows(P.tm, gm, it);
    const int32_t chunk0 = __builtin_amdgcn_readfirstlane(it.chunk - it.gi);
    const int32_t dtile = __builtin_amdgcn_readfirstlane(it.dt);
    const int32_t tau0 = it.chunk * gm.chunk;

    const int64_t crow = (int64_t)it.b * gm.nchunks + it.chunk;
    float en[kFS], dAn[kFS];                              // adjoint entering from the right / dA sums, per state
#pragma unroll
    for (int n = 0; n < kFS; ++n) {
        en[n] = P.carry[(crow * kFS + n) * gm.dim + it.d];
        dAn[n] = 0.f;
    }
    const float* Arow = P.A + (int64_t)it.d * kFS;
    const float bias = P.delta_bias ? P.delta_bias[it.d] : 0.f;
    const float Dv = P.D ? P.D[it.d] : 0.f;
    float dD_acc = 0.f, dbias_acc = 0.f;

    // inputs: cooperative 16-byte tiles (whole d-tile of the wave's items); outputs: per-lane rows
    const Seq* in_seq[NSTREAM] = {&P.u, &P.delta, &P.dout, has_z ? &P.z : &P.dout, has_z ? &P.out : &P.dout};
    TS tiles[NSTREAM];
#pragma unroll
    for (int i = 0; i < NSTREAM; ++i) tiles[i] = make_tile<T, RW>(*in_seq[i], ub, P.tm, gm, wr, chunk0, dtile, lane);
    const Stream dup = make_stream<T>(P.du, ub, wr, it.d);
    const Stream ddp = make_stream<T>(P.ddelta, ub, wr, it.d);
    const Stream dzp = make_stream<T>(has_z ? P.dz : P.du, ub, wr, it.d);
    const StageStream<RW> sb = make_stage<T, RW>(P.Bm, ub, wr, it.r);
    const StageStream<RW> sc = make_stage<T, RW>(P.Cm, ub, wr, it.r);
    // dB / dC (fp32, state-fastest): the lane that ends a reduce-scatter with the sum of step j stores element (j, n)
    BC dBv = {reinterpret_cast<char*>(P.dB), P.dB_sb, P.dB_st, P.dB_sn};
    BC dCv = {reinterpret_cast<char*>(P.dC), P.dC_sb, P.dC_st, P.dC_sn};
    char* const dBbase = dBv.p + ((int64_t)ub * dBv.sb + (int64_t)wr.row_lo * dBv.st) * 4;     // wave-uniform
    char* const dCbase = dCv.p + ((int64_t)ub * dCv.sb + (int64_t)wr.row_lo * dCv.st) * 4;
    const rsrc_t dBr = make_rsrc(dBbase);
    const rsrc_t dCr = make_rsrc(dCbase);
    const int32_t dB_stb = (int32_t)(P.dB_st * 4), dC_stb = (int32_t)(P.dC_st * 4);
    const int32_t dB_snb = (int32_t)(P.dB_sn * 4), dC_snb = (int32_t)(P.dC_sn * 4);

    char* tile = &s_tile[wave][0];
    // this lane's element of a tile row: [item][step][RW channels]
    const int32_t elem_off = (it.gi * kFT * RW + it.r) * (int)sizeof(T);
    T* raw_b = &s_raw[wave][it.gi][0][0];
    T* raw_c = &s_raw[wave][it.gi][1][0];
    float* ldb = &s_dbc[wave][it.gi][0][0];
    float* ldc = &s_dbc[wave][it.gi][1][0];
    float* lb = &s_bc[wave][it.gi][0][0];
    float* lc = &s_bc[wave][it.gi][1][0];
    // checkpoints [batch][nck][16][dim]: buffer based at the wave's lowest chunk
    const rsrc_t ckr = make_rsrc(P.ckpt + (((int64_t)ub * P.nck + (int64_t)chunk0 * (gm.chunk / kCkpt)) * kFS) * gm.dim);
    const uint32_t ck_voff = ((uint32_t)(it.gi * (gm.chunk / kCkpt)) * kFS * (uint32_t)gm.dim + (uint32_t)it.d) * 4u;
    const int32_t ck_state = gm.dim * 4;

    const int nwin = gm.chunk / kFW;
    // ---- first window's inputs: fetched and parked up front (the only exposed fetch of the chunk) -----------------------
    {
        const int32_t U0 = wr.bias + fast_U_of(P.tm, 2 * (nwin - 1)), U1 = wr.bias + fast_U_of(P.tm, 2 * (nwin - 1) + 1);
#pragma unroll
        for (int i = 0; i < NSTREAM; ++i) {
            u32x4_t v0[TS::NLH], v1[TS::NLH];
            tile_issue(v0, tiles[i], U0, wr.dT);
            tile_issue(v1, tiles[i], U1, wr.dT);
            tile_park(v0, tile + i * STREAM_BYTES, lane);
            tile_park(v1, tile + i * STREAM_BYTES + TS::HALF_BYTES, lane);
        }
        uint32_t vb0[EPL], vb1[EPL], vc0[EPL], vc1[EPL];
        stage_fetch_raw<T, RW>(vb0, sb, U0, wr.dT);
        stage_fetch_raw<T, RW>(vb1, sb, U1, wr.dT);
        stage_fetch_raw<T, RW>(vc0, sc, U0, wr.dT);
        stage_fetch_raw<T, RW>(vc1, sc, U1, wr.dT);
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            put_raw(raw_b + sb.lds0 + i * sb.ldsinc, vb0[i]);
            put_raw(raw_b + kFT * kFS + sb.lds0 + i * sb.ldsinc, vb1[i]);
            put_raw(raw_c + sc.lds0 + i * sc.ldsinc, vc0[i]);
            put_raw(raw_c + kFT * kFS + sc.lds0 + i * sc.ldsinc, vc1[i]);
        }
    }
    float hp_next;                                        // checkpoint of the next (state, window) of the walk, one state ahead
    hp_next = BufIO<float>::ld(ckr, ck_voff, (uint32_t)((nwin - 1) * kFS * ck_state));
    float A_next = Arow[0];                               // likewise A[d][n]: a load per state, never waited for where it is issued

    for (int w = nwin - 1; w >= 0; --w) {
        SEGM_WAVE_LDS_SYNC();                             // the tiles of this window are parked (by this wave)
        // ---- window prologue: the 16 steps of this lane from the tiles, B / C to fp32 -----------------------------------
        float wd[kFW], wg[kFW], wdu[kFW], q[kFW], ddA[kFW];
        {
            float dzv[kFW];
#pragma unroll
            for (int j = 0; j < kFW; ++j) {
                const int off = (j / kFT) * TS::HALF_BYTES + (j % kFT) * RW * (int)sizeof(T) + elem_off;
                const float uu = lds_elem<T>(tile + 0 * STREAM_BYTES + off);
                float dl = lds_elem<T>(tile + 1 * STREAM_BYTES + off) + bias;
                wd[j] = softplus_on ? softplus20(dl) : dl;
                wg[j] = lds_elem<T>(tile + 2 * STREAM_BYTES + off);
                if (has_z) {
                    const float zz = lds_elem<T>(tile + 3 * STREAM_BYTES + off), yy = lds_elem<T>(tile + 4 * STREAM_BYTES + off);
                    const float sg = sigmoidf(zz);
                    dzv[j] = wg[j] * yy * sg * fmaf(zz, 1.f - sg, 1.f);
                    wg[j] *= zz * sg;
                }
                wdu[j] = wd[j] * uu;
                q[j] = 0.f;
                ddA[j] = 0.f;
                dD_acc = fmaf(wg[j], uu, dD_acc);
            }
            if (has_z) {
                const int32_t U0 = wr.bias + fast_U_of(P.tm, 2 * w), U1 = wr.bias + fast_U_of(P.tm, 2 * w + 1);
                const uint32_t inc = (uint32_t)(wr.dT * dzp.stb);
                uint32_t so = (uint32_t)U0 * (uint32_t)dzp.stb;
#pragma unroll
                for (int j = 0; j < kFW; ++j) {
                    if (j == kFT) so = (uint32_t)U1 * (uint32_t)dzp.stb;
                    BufIO<T>::st(dzp.rs, dzp.voff, so, dzv[j]);
                    so += inc;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2 * EPL; ++i) {              // raw [step][state] -> fp32 [state][step]: a state's 16 steps are 64 bytes
            const int e = it.r + i * RW;
            const int t = (e & (kFS - 1)) * kFW + (e >> 4);
            lb[t] = to_f32(raw_b[e]);
            lc[t] = to_f32(raw_c[e]);
        }
        SEGM_WAVE_LDS_SYNC();                             // tiles and raw B / C are consumed: the state loop may refill them

        const int32_t Un0 = wr.bias + fast_U_of(P.tm, w > 0 ? 2 * (w - 1) : 0), Un1 = wr.bias + fast_U_of(P.tm, w > 0 ? 2 * (w - 1) + 1 : 1);
        const int32_t Uw0 = wr.bias + fast_U_of(P.tm, 2 * w), Uw1 = wr.bias + fast_U_of(P.tm, 2 * w + 1);
#pragma unroll 1
        for (int n = 0; n < kFS; ++n) {                   // runtime loop over the states
            // ---- the next window's inputs, a slice per state: delta, dout, z, out in n = 0..3, B / C halves in n = 4..7 (u: when
            //      the window closes).  Raw bits only: nothing here is a use of a loaded value (BufIO::ld_raw) ---------------
            u32x4_t pv0[TS::NLH], pv1[TS::NLH];
            uint32_t pbc[EPL];
            if (n == 0) { tile_issue(pv0, tiles[1], Un0, wr.dT); tile_issue(pv1, tiles[1], Un1, wr.dT); }
            else if (n == 1) { tile_issue(pv0, tiles[2], Un0, wr.dT); tile_issue(pv1, tiles[2], Un1, wr.dT); }
            else if (n == 2) { tile_issue(pv0, tiles[3], Un0, wr.dT); tile_issue(pv1, tiles[3], Un1, wr.dT); }
            else if (n == 3) { tile_issue(pv0, tiles[4], Un0, wr.dT); tile_issue(pv1, tiles[4], Un1, wr.dT); }
            else if (n == 4) stage_fetch_raw<T, RW>(pbc, sb, Un0, wr.dT);
            else if (n == 5) stage_fetch_raw<T, RW>(pbc, sb, Un1, wr.dT);
            else if (n == 6) stage_fetch_raw<T, RW>(pbc, sc, Un0, wr.dT);
            else if (n == 7) stage_fetch_raw<T, RW>(pbc, sc, Un1, wr.dT);
            const float hp = hp_next;
            {                                             // checkpoint of the state after this one (next window after the last)
                const int nn = n + 1 < kFS ? n + 1 : 0;
                const int wn = n + 1 < kFS ? w : (w > 0 ? w - 1 : 0);
                hp_next = BufIO<float>::ld(ckr, ck_voff, (uint32_t)__builtin_amdgcn_readfirstlane((wn * kFS + nn) * ck_state));
            }
            const float A2n = A_next * kLog2e;
            const float An = A2n * 0.6931471805599453f;
            A_next = Arow[(n + 1) & (kFS - 1)];
            float e1 = en[n];
            float dA1 = dAn[n];
            // this state's B_t / C_t of the window: eight broadcast 16-byte reads up front (per-step reads from a [step][state] tile
            // put one LDS round trip on the critical path of each of the 32 steps of the two loops: half the kernel's time)
            float bbv[kFW], ccv[kFW];
#pragma unroll
            for (int q4 = 0; q4 < kFW / 4; ++q4) {
                const float4 b4 = reinterpret_cast<const float4*>(lb + n * kFW)[q4];
                const float4 c4 = reinterpret_cast<const float4*>(lc + n * kFW)[q4];
                bbv[4 * q4] = b4.x; bbv[4 * q4 + 1] = b4.y; bbv[4 * q4 + 2] = b4.z; bbv[4 * q4 + 3] = b4.w;
                ccv[4 * q4] = c4.x; ccv[4 * q4 + 1] = c4.y; ccv[4 * q4 + 2] = c4.z; ccv[4 * q4 + 3] = c4.w;
            }
            float a[kFW], h[kFW];
#pragma unroll
            for (int j = 0; j < kFW; ++j) {
                const float bb = bbv[j];
                a[j] = fast_exp2(A2n * wd[j]);
                h[j] = fmaf(a[j], j ? h[j - 1] : hp, bb * wdu[j]);
            }
#pragma unroll
            for (int jj = 0; jj < kFW; ++jj) {
                const int j = kFW - 1 - jj;
                const float bb = bbv[j];
                const float cc = ccv[j];
                const float dh = fmaf(cc, wg[j], e1);
                e1 = a[j] * dh;
                const float t2 = e1 * (j ? h[j - 1] : hp);
                dA1 = fmaf(t2, wd[j], dA1);
                q[j] = fmaf(dh, bb, q[j]);
                ddA[j] = fmaf(t2, An, ddA[j]);
                a[j] = dh * wdu[j];                        // dB contribution of (j, n), in a's register
                h[j] = h[j] * wg[j];                       // dC contribution, in h's register
            }
            en[n] = e1;
            dAn[n] = dA1;
            // ---- sum the dB / dC contributions over the channels (lanes) of the work item; the lane that ends with the sum of
            //      step j parks it in the [step][state % 8] tile, which leaves for memory after states 7 and 15 as rows of 8
            //      consecutive states (scattered 4-byte atomics from here cost 3 x the whole kernel: one L2 transaction each) -----
            if constexpr (RW >= 32) {
                float v[2 * kFW];
#pragma unroll
                for (int j = 0; j < kFW; ++j) { v[j] = a[j]; v[kFW + j] = h[j]; }
                reduce_scatter<RW, V>(v, it.r);
                if (it.r < 32) (it.r < kFW ? ldb : ldc)[(it.r & (kFW - 1)) * (kFS / 2) + (n & 7)] = v[0];     // lanes 0..15 dB_j, 16..31 dC_j
            } else {
                reduce_scatter<RW, kFW>(a, it.r);
                reduce_scatter<RW, kFW>(h, it.r);
                ldb[it.r * (kFS / 2) + (n & 7)] = a[0];
                ldc[it.r * (kFS / 2) + (n & 7)] = h[0];
            }
            if ((n & 7) == 7) {
                SEGM_WAVE_LDS_SYNC();
                constexpr int FE = kFW * (kFS / 2) / RW;   // elements per lane and matrix: e = r + i RW -> step e / 8, state e % 8
#pragma unroll
                for (int i = 0; i < FE; ++i) {
                    const int e = it.r + i * RW, j = e >> 3, st8 = (n & 8) + (e & 7);
                    const uint32_t rows = (uint32_t)((j < kFT ? Uw0 : Uw1) + (j & (kFT - 1)) * wr.dT + wr.lane_row);
                    const uint32_t ob = rows * (uint32_t)dB_stb + (uint32_t)(st8 * dB_snb);
                    const uint32_t oc = rows * (uint32_t)dC_stb + (uint32_t)(st8 * dC_snb);
                    const float xb = ldb[e], xc = ldc[e];
                    if (P.atomic_bc) {
                        atomicAdd(reinterpret_cast<float*>(dBbase + ob), xb);
                        atomicAdd(reinterpret_cast<float*>(dCbase + oc), xc);
                    } else {
                        BufIO<float>::st(dBr, ob, 0u, xb);
                        BufIO<float>::st(dCr, oc, 0u, xc);
                    }
                }
                SEGM_WAVE_LDS_SYNC();                     // the tile is free for the next eight states
            }
            // ---- park what this state's slice fetched (it arrived during the arithmetic) ---------------------------------------
            if (n < 4) {
                tile_park(pv0, tile + (n + 1) * STREAM_BYTES, lane);
                tile_park(pv1, tile + (n + 1) * STREAM_BYTES + TS::HALF_BYTES, lane);
            } else if (n < 8) {
                const int k = n - 4;                       // 0: B first half, 1: B second half, 2: C first half, 3: C second half
                T* dst = (k < 2 ? raw_b : raw_c) + (k & 1) * kFT * kFS;
                const StageStream<RW>& ss = k < 2 ? sb : sc;
#pragma unroll
                for (int i = 0; i < EPL; ++i) put_raw(dst + ss.lds0 + i * ss.ldsinc, pbc[i]);
            }
        }
        // the next window's u: fetched while this window closes (its own u is still in the tile), parked behind the epilogue
        u32x4_t pu0[TS::NLH], pu1[TS::NLH];
        tile_issue(pu0, tiles[0], Un0, wr.dT);
        tile_issue(pu1, tiles[0], Un1, wr.dT);
        {
            float du[kFW], ddl[kFW];
#pragma unroll
            for (int j = 0; j < kFW; ++j) {
                const float uu = lds_elem<T>(tile + (j / kFT) * TS::HALF_BYTES + (j % kFT) * RW * (int)sizeof(T) + elem_off);
                du[j] = fmaf(wd[j], q[j], Dv * wg[j]);
                float ddv = fmaf(uu, q[j], ddA[j]);
                ddv *= softplus_on ? 1.f - fast_exp(-wd[j]) : 1.f;       // sigmoid(raw) = 1 - exp(-softplus(raw))
                dbias_acc += ddv;
                ddl[j] = ddv;
            }
            const uint32_t uinc = (uint32_t)(wr.dT * dup.stb), dinc = (uint32_t)(wr.dT * ddp.stb);
            uint32_t uso = (uint32_t)Uw0 * (uint32_t)dup.stb, dso = (uint32_t)Uw0 * (uint32_t)ddp.stb;
#pragma unroll
            for (int j = 0; j < kFW; ++j) {
                if (j == kFT) { uso = (uint32_t)Uw1 * (uint32_t)dup.stb; dso = (uint32_t)Uw1 * (uint32_t)ddp.stb; }
                BufIO<T>::st(dup.rs, dup.voff, uso, du[j]);
                BufIO<T>::st(ddp.rs, ddp.voff, dso, ddl[j]);
                uso += uinc;
                dso += dinc;
            }
        }
        SEGM_WAVE_LDS_SYNC();                             // every lane has read this window's u
        tile_park(pu0, tile, lane);
        tile_park(pu1, tile + TS::HALF_BYTES, lane);
    }
    const int64_t row = crow * (kFS + 2);
#pragma unroll
    for (int n = 0; n < kFS; ++n) P.part[(row + n) * gm.dim + it.d] = dAn[n];
    P.part[(row + kFS) * gm.dim + it.d] = dD_acc;
    P.part[(row + kFS + 1) * gm.dim + it.d] = dbias_acc;
    (void)tau0;
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
bool scan_bwd_fast_shape(const ScanDev& P, size_t esize) {
    if (!scan_fast_shape(P)) return false;
    // state-fastest B, C, dB, dC (the channel-last production layout)
    if (!(P.Bm.sn < P.Bm.st && P.Cm.sn < P.Cm.st && P.dB_sn < P.dB_st && P.dC_sn < P.dC_st)) return false;
    // the input streams are fetched as 16-byte pieces of channel-contiguous rows
    const Seq* in[5] = {&P.u, &P.delta, &P.dout, &P.z, &P.out};
    for (const Seq* s : in) {
        if (!s->p) continue;
        if (s->sd != 1 || (s->st * (int64_t)esize) % 16 != 0 || (s->sb * (int64_t)esize) % 16 != 0 ||
            (reinterpret_cast<uintptr_t>(s->p) % 16) != 0)
            return false;
    }
    return ((int64_t)P.gm.rw * (int64_t)esize) % 16 == 0;
}

template <typename T, int RW>
static void launch_bwd_fast_rw(const ScanDevN& PP, int ndir, bool main, hipStream_t stream) {
    const unsigned nblocks = (unsigned)((PP.d[0].gm.nwaves + kWavesPerBlock - 1) / kWavesPerBlock);
    if (main) hipLaunchKernelGGL((scan_bwd_main_fast_kernel<T, RW>), dim3(nblocks, ndir), dim3(kBlock), 0, stream, PP);
    else hipLaunchKernelGGL((scan_bwd_agg_fast_kernel<T, RW>), dim3(nblocks, ndir), dim3(kBlock), 0, stream, PP);
}
template <typename T>
static void launch_bwd_fast_t(const ScanDevN& PP, int ndir, bool main, hipStream_t stream) {
    if (PP.d[0].gm.rw == 64) launch_bwd_fast_rw<T, 64>(PP, ndir, main, stream);
    else if (PP.d[0].gm.rw == 32) launch_bwd_fast_rw<T, 32>(PP, ndir, main, stream);
    else launch_bwd_fast_rw<T, 16>(PP, ndir, main, stream);
}
bool scan_full_span_fits(const ScanDev& P, size_t esize) {
    const int64_t lim = (int64_t)1 << 32, L = P.gm.L, es = (int64_t)esize;
    const Seq* sq[9] = {&P.u, &P.delta, &P.z, &P.out, &P.dout, &P.du, &P.ddelta, &P.dz, nullptr};
    for (const Seq* s : sq)
        if (s && s->p && ((L - 1) * s->st + (int64_t)(P.gm.dim - 1) * s->sd + 1) * es >= lim) return false;
    const BC* bc[2] = {&P.Bm, &P.Cm};
    for (const BC* m : bc)
        if (((L - 1) * m->st + (int64_t)(P.gm.nstate - 1) * m->sn + 1) * es >= lim) return false;
    return ((L - 1) * P.dB_st + (P.gm.nstate - 1) * P.dB_sn + 1) * 4 < lim && ((L - 1) * P.dC_st + (P.gm.nstate - 1) * P.dC_sn + 1) * 4 < lim;
}
// main kernel choice: "r2" = the pair kernel (scan_bwd_pair.hip: 16-step windows, one wave per SIMD, 32-bit offsets from the batch
// base), "r3" = the kernel above (wave-local bases: any span).  The pair kernel is the default wherever it can address the tensors:
// what bounds these kernels is the NUMBER of instructions a SIMD issues (~4.4 cycles each, v_exp_f32 ~9, whatever the number of
// resident waves - profiles/r03_scan_occupancy.log), and both multi-wave kernels of round 3 issue more of them (this one: LDS
// tiles and per-state loops; the half-window kernel of tools/experiments: +50 % recurrence exponentials and spills).
static bool use_pair_kernel(const ScanDevN& PP, int ndir, int dtype) {
    const char* e = getenv("SEGM_BWD_MAIN");               // read per launch: tests switch it inside one process
    if (e && e[0] == 'r' && e[1] == '3') return false;
    const size_t es = dtype == SEGM_F32 ? 4 : 2;
    for (int i = 0; i < ndir; ++i)
        if (!scan_full_span_fits(PP.d[i], es)) return false;
    return true;
}
// launches K1 (main == false) or K3 (main == true) of the regular-shape backward for `ndir` blocks of one geometry
void launch_scan_bwd_fast(const ScanDevN& PP, int ndir, int dtype, bool main, hipStream_t stream) {
    if (main && use_pair_kernel(PP, ndir, dtype)) {
        launch_scan_bwd_main_pair(PP, ndir, dtype, stream);
        return;
    }
    if (dtype == SEGM_F32) launch_bwd_fast_t<float>(PP, ndir, main, stream);
    else if (dtype == SEGM_F16) launch_bwd_fast_t<f16_t>(PP, ndir, main, stream);
    else launch_bwd_fast_t<bf16_t>(PP, ndir, main, stream);
}

}  // namespace segm
