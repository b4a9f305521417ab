// Selective scan backward, regular shapes: main kernel on 8-step HALF windows, three waves per SIMD.
//
// What bounds the backward main kernel is the number of VALU instructions a SIMD has to issue, and how many waves it can
// choose them from: the 16-step-window kernel (scan_bwd_pair.hip) keeps a[16], h[16] and six 16-entry row arrays alive -
// 256 VGPRs + AGPR copies, ONE wave per SIMD, and a lone wave issues a dependent VALU instruction only every ~7 cycles
// (profiles/r02_probe_valu2.log).  Here a window is walked as two halves of 8 steps, upper half first:
//     state entering the upper half = checkpoint advanced through the lower half (recurrence only: 8 more exponentials per
//     state and window, +8 % of the kernel's VALU cycles), then  forward 8 steps keeping a_j, h_j  ->  backward 8 steps;
//     the lower half starts from the checkpoint itself.
// The per-step arrays halve (<= 168 VGPRs, no AGPRs), three workgroups of four waves fit a CU (52 KB of LDS each), and the
// loads of one wave's half window are covered by the other two waves instead of a register prefetch.
// The adjoint entering from the right and the dA sums of a state pair live in LDS between the halves (runtime loop over the
// pairs), dB / dC contributions are summed over the channels of a work item by the DPP reduce-scatter - one call per state
// pair: 8 steps x {dB, dC} x 2 states = 32 values - and leave through an LDS tile per half.
// Same arguments, checkpoints, workspace and results as the other main kernels (reference selective_scan_bwd_kernel.cuh:75-489).
#include "scan_fast.h"

namespace segm {

constexpr int kHW = kFT;    // steps of a half window
#ifndef SEGM_BWD_HALF_WAVES
#define SEGM_BWD_HALF_WAVES 3
#endif

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

// One state pair of one half window.  HH = 1: upper half (steps 8 .. 15 of the window, entered through the lower half).
//   wd / wdu: softplus(delta) and softplus(delta) u of the WINDOW's 16 steps; wg: gated dout of the half; lb: B of the window,
//   lc: C of the half ([step][state] fp32 in LDS); qs / ddA: per-step sums over the states.
template <int HH, int RW>
__device__ __forceinline__ void half_pair(int p, const f2 A2n, const f2 An, const f2 hp, f2& en, f2& dAn, const float (&wd)[2 * kHW],
                                          const float (&wdu)[2 * kHW], const float (&wg)[kHW], float (&qs)[kHW], float (&ddA)[kHW],
                                          const float* lb, const float* lc, float* ldb, float* ldc, int r) {
    f2 hin = hp;
    if (HH == 1) {
#pragma unroll
        for (int j = 0; j < kHW; ++j) {
            const f2 bb = *reinterpret_cast<const f2*>(lb + j * kFS + 2 * p);
            const f2 da = A2n * wd[j];
            const f2 a = {fast_exp2(da.x), fast_exp2(da.y)};
            hin = a * hin + bb * wdu[j];
        }
    }
    f2 a[kHW], h[kHW];
#pragma unroll
    for (int j = 0; j < kHW; ++j) {
        const int jj = HH * kHW + j;
        const f2 bb = *reinterpret_cast<const f2*>(lb + jj * kFS + 2 * p);
        const f2 da = A2n * wd[jj];
        a[j] = f2{fast_exp2(da.x), fast_exp2(da.y)};
        h[j] = a[j] * (j ? h[j - 1] : hin) + bb * wdu[jj];
    }
#pragma unroll
    for (int q = 0; q < kHW; ++q) {
        const int j = kHW - 1 - q, jj = HH * kHW + j;
        const f2 bb = *reinterpret_cast<const f2*>(lb + jj * kFS + 2 * p);
        const f2 cc = *reinterpret_cast<const f2*>(lc + j * kFS + 2 * p);
        const f2 dh = cc * wg[j] + en;
        en = a[j] * dh;                                    // adjoint leaving step j to the left
        const f2 t2 = en * (j ? h[j - 1] : hin);           // dL/d(delta A) of (j, pair)
        dAn = t2 * wd[jj] + dAn;
        qs[j] = fmaf(dh.y, bb.y, fmaf(dh.x, bb.x, qs[j]));       // scalar sums: 16 registers fewer than packed partial sums, +1 cycle
        ddA[j] = fmaf(t2.y, An.y, fmaf(t2.x, An.x, ddA[j]));
        a[j] = dh * wdu[jj];                               // dB contribution, in a's registers
        h[j] = h[j] * wg[j];                               // dC contribution, in h's registers
    }
    // sum the contributions over the channels (lanes) of the work item
    if constexpr (RW >= 32) {
        float v[4 * kHW];                                  // [dB x | dC x | dB y | dC y][step]
#pragma unroll
        for (int j = 0; j < kHW; ++j) {
            v[j] = a[j].x;
            v[kHW + j] = h[j].x;
            v[2 * kHW + j] = a[j].y;
            v[3 * kHW + j] = h[j].y;
        }
        reduce_scatter<RW, 32>(v, r);
        if (r < 32) {
            const int grp = r >> 3, j = r & (kHW - 1);
            ((grp & 1) ? ldc : ldb)[j * kFS + 2 * p + (grp >> 1)] = v[0];
        }
    } else {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float v[2 * kHW];                              // [dB | dC][step] of one state
#pragma unroll
            for (int j = 0; j < kHW; ++j) {
                v[j] = half ? a[j].y : a[j].x;
                v[kHW + j] = half ? h[j].y : h[j].x;
            }
            reduce_scatter<RW, 16>(v, r);
            ((r & kHW) ? ldc : ldb)[(r & (kHW - 1)) * kFS + 2 * p + half] = v[0];
        }
    }
}

template <typename T, int RW>
__global__ void __launch_bounds__(kBlock, SEGM_BWD_HALF_WAVES) scan_bwd_main_half_kernel(ScanDevN PP) {
    constexpr int G = 64 / RW, EPL = StageStream<RW>::EPL;
    __shared__ __attribute__((aligned(16))) float s_b[kWavesPerBlock][G][2 * kHW * kFS];       // B of the window, [step][state]
    __shared__ __attribute__((aligned(16))) float s_c[kWavesPerBlock][G][kHW * kFS];           // C of the half
    __shared__ __attribute__((aligned(16))) float s_dbc[kWavesPerBlock][G][2][kHW * kFS];      // dB, dC of the half
    __shared__ f2 s_e[kFS / 2][kBlock];                  // adjoint entering from the right, per thread and state pair
    __shared__ f2 s_dA[kFS / 2][kBlock];
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
    const int32_t tau0 = it.chunk * gm.chunk;
    const int32_t t_item = fast_item_row(P.tm, tau0);
    const int32_t dT = wr.dT;

    const int64_t crow = (int64_t)it.b * gm.nchunks + it.chunk;
#pragma unroll
    for (int p = 0; p < kFS / 2; ++p) {
        s_e[p][threadIdx.x] = f2{P.carry[(crow * kFS + 2 * p) * gm.dim + it.d], P.carry[(crow * kFS + 2 * p + 1) * gm.dim + it.d]};
        s_dA[p][threadIdx.x] = f2{0.f, 0.f};
    }
    const float* Arow = P.A + (int64_t)it.d * kFS;
    const float bias = P.delta_bias ? P.delta_bias[it.d] : 0.f;
    const float Dv = P.D ? P.D[it.d] : 0.f;
    float dD_acc = 0.f, dbias_acc = 0.f;

    const Stream up = make_stream<T>(P.u, ub, wr, it.d);
    const Stream dp = make_stream<T>(P.delta, ub, wr, it.d);
    const Stream gp = make_stream<T>(P.dout, ub, wr, it.d);
    const Stream zp = make_stream<T>(has_z ? P.z : P.dout, ub, wr, it.d);
    const Stream yp = make_stream<T>(has_z ? P.out : P.dout, ub, wr, it.d);
    const Stream dup = make_stream<T>(P.du, ub, wr, it.d);
    const Stream ddp = make_stream<T>(P.ddelta, ub, wr, it.d);
    const Stream dzp = make_stream<T>(has_z ? P.dz : P.du, ub, wr, it.d);
    const StageStream<RW> sb = make_stage<T, RW>(P.Bm, ub, wr, it.r);
    const StageStream<RW> sc = make_stage<T, RW>(P.Cm, ub, wr, it.r);
    // dB / dC flush (state-fastest fp32): lane r writes elements el = r + i RW of the half's tile: j = el / 16, n = el % 16
    const int fj = it.r / kFS, fn = it.r % kFS;
    constexpr int FJ = RW >= kFS ? RW / kFS : 1;          // steps between a lane's consecutive elements
    char* dBb = reinterpret_cast<char*>(P.dB) + (int64_t)ub * P.dB_sb * 4;
    char* dCb = reinterpret_cast<char*>(P.dC) + (int64_t)ub * P.dC_sb * 4;
    const uint32_t dB_loff = (uint32_t)(t_item + fj * dT) * (uint32_t)(P.dB_st * 4) + (uint32_t)fn * (uint32_t)(P.dB_sn * 4);
    const uint32_t dC_loff = (uint32_t)(t_item + fj * dT) * (uint32_t)(P.dC_st * 4) + (uint32_t)fn * (uint32_t)(P.dC_sn * 4);
    float* lb = &s_b[wave][it.gi][0];
    float* lc = &s_c[wave][it.gi][0];
    float* ldb = &s_dbc[wave][it.gi][0][0];
    float* ldc = &s_dbc[wave][it.gi][1][0];
    // checkpoints [batch][nck][16][dim]: buffer based at the wave's lowest chunk
    const rsrc_t ckr = make_rsrc(P.ckpt + (((int64_t)ub * P.nck + (int64_t)chunk0 * (gm.chunk / kCkpt)) * kFS) * gm.dim);
    const uint32_t ck_voff = ((uint32_t)(it.gi * (gm.chunk / kCkpt)) * kFS * (uint32_t)gm.dim + (uint32_t)it.d) * 4u;
    const int32_t ck_state = gm.dim * 4;

    const int nwin = gm.chunk / (2 * kHW);
    for (int w = nwin - 1; w >= 0; --w) {
        const int32_t Uq[2] = {fast_U_of(P.tm, 2 * w), fast_U_of(P.tm, 2 * w + 1)};       // row offsets of the two halves
        float wd[2 * kHW], wdu[2 * kHW];
        // ---- the window's delta, u (both halves) and B ---------------------------------------------------------------
        {
            float vb0[EPL], vb1[EPL], x0[kHW], x1[kHW], d0[kHW], d1[kHW];
            stage_fetch_buf<T, RW>(vb0, sb, wr.bias + Uq[0], dT);
            stage_fetch_buf<T, RW>(vb1, sb, wr.bias + Uq[1], dT);
            stream_fetch<T>(d0, dp, wr.bias + Uq[0], dT);
            stream_fetch<T>(d1, dp, wr.bias + Uq[1], dT);
            stream_fetch<T>(x0, up, wr.bias + Uq[0], dT);
            stream_fetch<T>(x1, up, wr.bias + Uq[1], dT);
            SEGM_WAVE_LDS_SYNC();                         // the previous window is done with s_b
            stage_park_buf<RW>(vb0, sb, lb);
            stage_park_buf<RW>(vb1, sb, lb + kHW * kFS);
#pragma unroll
            for (int j = 0; j < kHW; ++j) {
                const float a0 = d0[j] + bias, a1 = d1[j] + bias;
                wd[j] = softplus_on ? softplus20(a0) : a0;
                wd[kHW + j] = softplus_on ? softplus20(a1) : a1;
                wdu[j] = wd[j] * x0[j];
                wdu[kHW + j] = wd[kHW + j] * x1[j];
            }
        }
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) {
            const int hh = 1 - hx;                        // upper half first
            const int32_t Uh = wr.bias + Uq[hh];
            float wg[kHW], qs[kHW], ddA[kHW];
            // ---- the half's dout, z, out and C ------------------------------------------------------------------------
            {
                float vc[EPL];
                stage_fetch_buf<T, RW>(vc, sc, Uh, dT);
                stream_fetch<T>(wg, gp, Uh, dT);
                SEGM_WAVE_LDS_SYNC();                     // the previous half is done with s_c / s_dbc
                stage_park_buf<RW>(vc, sc, lc);
                if (has_z) {
                    float wz[kHW], wy[kHW];
                    stream_fetch<T>(wz, zp, Uh, dT);
                    stream_fetch<T>(wy, yp, Uh, dT);
#pragma unroll
                    for (int j = 0; j < kHW; ++j) {
                        const float zz = wz[j], sg = sigmoidf(zz);
                        wz[j] = wg[j] * wy[j] * sg * fmaf(zz, 1.f - sg, 1.f);       // dz
                        wg[j] *= zz * sg;
                    }
                    stream_store<T>(wz, dzp, Uh, dT);
                }
#pragma unroll
                for (int j = 0; j < kHW; ++j) {
                    qs[j] = 0.f;
                    ddA[j] = 0.f;
                }
            }
            SEGM_WAVE_LDS_SYNC();                         // B / C tiles visible to the item's lanes
#pragma unroll 1
            for (int p = 0; p < kFS / 2; ++p) {           // runtime loop over state pairs
                const f2 A2n = f2{Arow[2 * p], Arow[2 * p + 1]} * kLog2e;
                const f2 An = A2n * 0.6931471805599453f;
                const uint32_t cko = (uint32_t)((w * kFS + 2 * p) * ck_state);
                const f2 hp = {BufIO<float>::ld(ckr, ck_voff, cko), BufIO<float>::ld(ckr, ck_voff, cko + (uint32_t)ck_state)};
                f2 en = s_e[p][threadIdx.x];
                f2 dAn = s_dA[p][threadIdx.x];
                if (hh) half_pair<1, RW>(p, A2n, An, hp, en, dAn, wd, wdu, wg, qs, ddA, lb, lc, ldb, ldc, it.r);
                else half_pair<0, RW>(p, A2n, An, hp, en, dAn, wd, wdu, wg, qs, ddA, lb, lc, ldb, ldc, it.r);
                s_e[p][threadIdx.x] = en;
                s_dA[p][threadIdx.x] = dAn;
            }
            float wu[kHW];                                // u of the half again (not kept across the pair loop: 8 registers)
            stream_fetch<T>(wu, up, Uh, dT);
            SEGM_WAVE_LDS_SYNC();                         // the dB / dC tile of every item is complete
            {
                char* pb = dBb + (int64_t)Uq[hh] * (P.dB_st * 4);
                char* pc = dCb + (int64_t)Uq[hh] * (P.dC_st * 4);
#pragma unroll
                for (int i = 0; i < EPL; ++i) {
                    const int j = fj + i * FJ;
                    float* ob = reinterpret_cast<float*>(pb + (int64_t)(i * FJ * dT) * (P.dB_st * 4) + dB_loff);
                    float* oc = reinterpret_cast<float*>(pc + (int64_t)(i * FJ * dT) * (P.dC_st * 4) + dC_loff);
                    const float xb = ldb[j * kFS + fn], xc = ldc[j * kFS + fn];
                    if (P.atomic_bc) { atomicAdd(ob, xb); atomicAdd(oc, xc); }
                    else { *ob = xb; *oc = xc; }
                }
            }
            {
                float du[kHW], ddl[kHW];
#pragma unroll
                for (int j = 0; j < kHW; ++j) {
                    const int jj = hh * kHW + j;
                    const float q = qs[j];
                    dD_acc = fmaf(wg[j], wu[j], dD_acc);
                    du[j] = fmaf(wd[jj], q, Dv * wg[j]);
                    float ddv = fmaf(wu[j], q, ddA[j]);
                    ddv *= softplus_on ? 1.f - fast_exp(-wd[jj]) : 1.f;      // sigmoid(raw) = 1 - exp(-softplus(raw))
                    dbias_acc += ddv;
                    ddl[j] = ddv;
                }
                stream_store<T>(du, dup, Uh, dT);
                stream_store<T>(ddl, ddp, Uh, dT);
            }
        }
    }
    const int64_t row = crow * (kFS + 2);
#pragma unroll
    for (int p = 0; p < kFS / 2; ++p) {
        const f2 dA = s_dA[p][threadIdx.x];
        P.part[(row + 2 * p) * gm.dim + it.d] = dA.x;
        P.part[(row + 2 * p + 1) * gm.dim + it.d] = dA.y;
    }
    P.part[(row + kFS) * gm.dim + it.d] = dD_acc;
    P.part[(row + kFS + 1) * gm.dim + it.d] = dbias_acc;
}

template <typename T, int RW>
static void launch_half_rw(const ScanDevN& PP, int ndir, hipStream_t stream) {
    const unsigned nblocks = (unsigned)((PP.d[0].gm.nwaves + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL((scan_bwd_main_half_kernel<T, RW>), dim3(nblocks, ndir), dim3(kBlock), 0, stream, PP);
}
template <typename T>
static void launch_half_t(const ScanDevN& PP, int ndir, hipStream_t stream) {
    if (PP.d[0].gm.rw == 64) launch_half_rw<T, 64>(PP, ndir, stream);
    else if (PP.d[0].gm.rw == 32) launch_half_rw<T, 32>(PP, ndir, stream);
    else launch_half_rw<T, 16>(PP, ndir, stream);
}
void launch_scan_bwd_main_half(const ScanDevN& PP, int ndir, int dtype, hipStream_t stream) {
    if (dtype == SEGM_F32) launch_half_t<float>(PP, ndir, stream);
    else if (dtype == SEGM_F16) launch_half_t<f16_t>(PP, ndir, stream);
    else launch_half_t<bf16_t>(PP, ndir, stream);
}

}  // namespace segm
