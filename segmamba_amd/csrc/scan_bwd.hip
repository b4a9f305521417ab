// Selective scan, backward (C ABI: segm_selective_scan_bwd).
//
// Replaces reference mamba/csrc/selective_scan/selective_scan_bwd_kernel.cuh:75-489, reverse_scan.cuh and
// selective_scan.cpp:338-492.  Same decomposition as the forward (one lane per channel, L cut into chunks):
//
//   K1  scan_bwd_agg_kernel   per (batch, chunk, channel): walk the chunk right-to-left with a zero adjoint
//                             entering from the right; emits E = adjoint leaving on the left, and sum(delta)
//   K2  scan_carry_kernel<1>  compose E over chunks right-to-left -> adjoint entering every chunk
//   K3  scan_bwd_main_kernel  per (batch, chunk, channel): 16-step windows right-to-left; each is re-run forward
//                             from the forward pass' checkpoint (state entering step 16k), state by state,
//                             keeping a_t, h_t of ONE state in registers
//                             (the reference keeps them for one state per thread tile too, :235-273), then
//                             walked backwards:
//        dh_t  = g_t C_t + e_{t+1}            e_t = a_t dh_t           g = dout * silu(z)
//        du_t += delta_t <dh_t, B_t> (+ D g_t)     ddelta_t = <dh_t, B_t u_t + A a_t h_{t-1}>
//        dA   += dh_t a_t h_{t-1} delta_t          dB_t = sum_d dh_t delta_t u_t      dC_t = sum_d g_t h_t
//      (reference :330-478).  dB / dC sum over channels = over the lanes of a work
//      item: a reduce-scatter butterfly (log2(RW) shuffle stages, each lane ends with one finished value),
//      parked in LDS and flushed once per window - no per-element global atomics from every channel as in
//      the reference (:297-316); atomics are only used between d-tiles.
//   K4  reduce_partials_kernel  per-item partial dA / dD / ddelta_bias -> final (deterministic, no atomics)
#include <stdlib.h>
#include <string.h>

#include "scan_fast.h"

namespace segm {

void launch_reduce_partials(const float* part, int64_t nrows, int K, int dim, float* out0, int K0, float* out1,
                            float* out2, hipStream_t stream);
void launch_reduce_partials_multi(const ReduceN& R, int n, int64_t nrows, int K, int dim, int K0, hipStream_t stream);

constexpr int kWin = 16;   // window length of the backward main kernel = spacing of the forward checkpoints

// ------------------------------------------------------------------------------------------------------
// K1: reverse chunk aggregates
// ------------------------------------------------------------------------------------------------------
template <typename T, int NS, int TS, int RW>
__global__ void __launch_bounds__(kBlock) scan_bwd_agg_kernel(ScanDev P) {
    constexpr int G = 64 / RW;
    __shared__ __attribute__((aligned(16))) float s_c[2][kWavesPerBlock][G][TS * NS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Geom& gm = P.gm;
    const TimeMap tm = P.tm;
    const Item it = locate(gm, (int64_t)blockIdx.x * kWavesPerBlock + wave, lane);
    const int ub = uniform_batch(it);
    const bool item_ok = it.wave_valid && it.chunk < gm.nchunks;
    const int nstate = gm.nstate;
    const bool softplus_on = P.delta_softplus != 0;
    const bool has_z = P.z.p != nullptr;

    float A2[NS], e[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        A2[n] = (it.valid && n < nstate) ? P.A[(int64_t)it.d * nstate + n] * kLog2e : 0.f;
        e[n] = 0.f;
    }
    const float bias = (it.valid && P.delta_bias) ? P.delta_bias[it.d] : 0.f;
    const RowPtr dp = make_rowptr<T>(P.delta, ub, it.d, it.valid);
    const RowPtr gp = make_rowptr<T>(P.dout, ub, it.d, it.valid);
    const Seq& zs = has_z ? P.z : P.dout;
    const RowPtr zp = make_rowptr<T>(zs, ub, it.d, it.valid);
    const bool t_fastest = P.Cm.st <= P.Cm.sn;

    TimeIter ti;
    ti.seek(tm, item_ok ? it.chunk * gm.chunk : 0);
    ti.jump(tm, gm.chunk > 64 ? 64 : gm.chunk);          // to the chunk's last sub-tile, in hops of <= 64
    for (int done = 64; done < gm.chunk; done += 64) ti.jump(tm, gm.chunk - done > 64 ? 64 : gm.chunk - done);
    ti.jump(tm, -TS);

    float nd[TS], ng[TS], nz[TS];
    int32_t ntt[TS];
    StageRegs<TS, NS, RW> sc;
    uint32_t nok = row_indices<TS>(ntt, tm, ti, it.valid);
    fetch_rows<T, TS>(nd, dp, ntt, nok);
    fetch_rows<T, TS>(ng, gp, ntt, nok);
    fetch_rows<T, TS>(nz, zp, ntt, nok);
    stage_fetch<T, TS, NS, RW>(sc, P.Cm, tm, ti, ub, nstate, it.r, item_ok);

    float sumd = 0.f;
    int buf = 0;
    for (int s0 = gm.chunk - TS; s0 >= 0; s0 -= TS) {
        float* lc = &s_c[buf][wave][it.gi][0];
        stage_park<TS, NS, RW, true>(sc, lc, t_fastest, it.r);
        SEGM_WAVE_LDS_SYNC();
        float cd[TS], cg[TS], cz[TS];
#pragma unroll
        for (int j = 0; j < TS; ++j) { cd[j] = nd[j]; cg[j] = ng[j]; cz[j] = nz[j]; }
        const uint32_t cok = nok;
        ti.jump(tm, -TS);                                  // next (lower) sub-tile; below the chunk start rows are
        nok = row_indices<TS>(ntt, tm, ti, it.valid);      // the neighbour's or masked (tau < 0): prefetch only
        fetch_rows<T, TS>(nd, dp, ntt, nok);
        fetch_rows<T, TS>(ng, gp, ntt, nok);
        fetch_rows<T, TS>(nz, zp, ntt, nok);
        stage_fetch<T, TS, NS, RW>(sc, P.Cm, tm, ti, ub, nstate, it.r, item_ok);
        float4 cq[NS / 4];
#pragma unroll
        for (int q = 0; q < NS / 4; ++q) cq[q] = reinterpret_cast<const float4*>(lc + (TS - 1) * NS)[q];
#pragma unroll
        for (int jj = 0; jj < TS; ++jj) {
            const int j = TS - 1 - jj;
            const bool ok = (cok >> j) & 1u;
            float dl = cd[j] + bias;
            dl = softplus_on ? softplus20(dl) : dl;       // select, not a branch: keeps each step one basic block
            dl = ok ? dl : 0.f;
            sumd += dl;
            float g = cg[j];
            if (has_z) { const float zz = cz[j]; g *= zz * sigmoidf(zz); }
            float4 cn[NS / 4];
            if (j > 0) {                                   // next (lower) step's C row, one step ahead of its use
#pragma unroll
                for (int q = 0; q < NS / 4; ++q) cn[q] = reinterpret_cast<const float4*>(lc + (j - 1) * NS)[q];
            }
#pragma unroll
            for (int q = 0; q < NS / 4; ++q) {
                const float cc[4] = {cq[q].x, cq[q].y, cq[q].z, cq[q].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = q * 4 + i;
                    const float a = fast_exp2(dl * A2[n]);
                    e[n] = a * fmaf(g, cc[i], e[n]);
                }
            }
#pragma unroll
            for (int n = 0; n < NS; ++n) SEGM_PIN_F32(e[n]);     // finish this step before the LDS reads two steps ahead
            if (j > 0) {
#pragma unroll
                for (int q = 0; q < NS / 4; ++q) cq[q] = cn[q];
            }
        }
        buf ^= 1;
    }
    if (it.valid) {
        const int64_t row = (int64_t)it.b * gm.nchunks + it.chunk;
        P.agg_sd[row * gm.dim + it.d] = sumd;
#pragma unroll
        for (int n = 0; n < NS; ++n)
            if (n < nstate) P.agg_h[(row * nstate + n) * gm.dim + it.d] = e[n];
    }
}

// ------------------------------------------------------------------------------------------------------
// K3: main backward kernel
// ------------------------------------------------------------------------------------------------------
#ifndef SEGM_BWD_MIN_WAVES
#define SEGM_BWD_MIN_WAVES 1          // waves per SIMD the register allocator must leave room for
#endif
template <typename T, int NS, int RW>
__global__ void __launch_bounds__(kBlock, SEGM_BWD_MIN_WAVES) scan_bwd_main_kernel(ScanDev P) {
    static_assert(kWin % kCkpt == 0 && kChunkQuantum % kWin == 0, "a window starts at a forward checkpoint");
    constexpr int G = 64 / RW;
    constexpr int V = RW < 32 ? RW : 32;
    __shared__ __attribute__((aligned(16))) float s_bc[kWavesPerBlock][G][2][NS * kWin];    // [n][s]: B then C
    __shared__ __attribute__((aligned(16))) float s_dbc[kWavesPerBlock][G][2][kWin * NS];   // [j][n]: dB then dC
    __shared__ int32_t s_rows[kWavesPerBlock][G][kWin];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Geom& gm = P.gm;
    const TimeMap tm = P.tm;
    const Item it = locate(gm, (int64_t)blockIdx.x * kWavesPerBlock + wave, lane);
    const int ub = uniform_batch(it);
    const bool item_ok = it.wave_valid && it.chunk < gm.nchunks;
    const int nstate = gm.nstate;
    const bool softplus_on = P.delta_softplus != 0;
    const bool has_z = P.z.p != nullptr;
    const int dsafe = it.valid ? it.d : 0;

    // per-state values that live across windows - the adjoint e and the dA accumulator - sit in LDS, one private
    // slot per thread and state ([n][thread]: conflict-free), because the loop over states is a runtime loop
    // (unrolling it 16x would not fit the instruction cache, indexing a register array by n would spill)
    __shared__ float s_e[NS][kBlock];
    __shared__ float s_dA[NS][kBlock];
    const int64_t crow = (int64_t)it.b * gm.nchunks + it.chunk;
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        const bool on = it.valid && n < nstate;
        s_e[n][threadIdx.x] = on ? P.carry[(crow * nstate + n) * gm.dim + it.d] : 0.f;
        s_dA[n][threadIdx.x] = 0.f;
    }
    const float* Arow = P.A + (int64_t)dsafe * nstate;       // A[d][.]: re-read per state (L1-resident)
    const float bias = (it.valid && P.delta_bias) ? P.delta_bias[it.d] : 0.f;
    const float Dv = (it.valid && P.D) ? P.D[it.d] : 0.f;
    float dD_acc = 0.f, dbias_acc = 0.f;

    const RowPtr up = make_rowptr<T>(P.u, ub, it.d, it.valid);
    const RowPtr dp = make_rowptr<T>(P.delta, ub, it.d, it.valid);
    const RowPtr gp = make_rowptr<T>(P.dout, ub, it.d, it.valid);
    const Seq& zs = has_z ? P.z : P.dout;
    const Seq& ys = has_z ? P.out : P.dout;
    const RowPtr zp = make_rowptr<T>(zs, ub, it.d, it.valid);
    const RowPtr yp = make_rowptr<T>(ys, ub, it.d, it.valid);
    const RowPtr dup = make_rowptr<T>(P.du, ub, it.d, it.valid);
    const RowPtr ddp = make_rowptr<T>(P.ddelta, ub, it.d, it.valid);
    const RowPtr dzp = make_rowptr<T>(P.dz, ub, it.d, it.valid && has_z);
    const bool bt_fastest = P.Bm.st <= P.Bm.sn, ct_fastest = P.Cm.st <= P.Cm.sn;
    float* lb = &s_bc[wave][it.gi][0][0];
    float* lc = &s_bc[wave][it.gi][1][0];
    float* ldb = &s_dbc[wave][it.gi][0][0];
    float* ldc = &s_dbc[wave][it.gi][1][0];
    int32_t* lrows = &s_rows[wave][it.gi][0];

    const int32_t tau_begin = item_ok ? it.chunk * gm.chunk : 0;
    const int nwin = gm.chunk / kWin;
    TimeIter tw;
    tw.seek(tm, tau_begin);
    for (int done = 0; done < gm.chunk - kWin; done += kWin) tw.jump(tm, kWin);     // -> the chunk's last window

    for (int w = nwin - 1; w >= 0; --w) {
        // ---- window data -------------------------------------------------------------------------------------
        StageRegs<kWin, NS, RW> sb, scc;
        stage_fetch<T, kWin, NS, RW>(sb, P.Bm, tm, tw, ub, nstate, it.r, item_ok);
        stage_fetch<T, kWin, NS, RW>(scc, P.Cm, tm, tw, ub, nstate, it.r, item_ok);
        float wu[kWin], wd[kWin], wg[kWin], du[kWin], dd[kWin], wdu[kWin];     // wdu = delta * u, shared by all states
        uint32_t okm;
        {
            int32_t tt[kWin];
            okm = row_indices<kWin>(tt, tm, tw, it.valid);
            fetch_rows<T, kWin>(wu, up, tt, okm);
            fetch_rows<T, kWin>(wd, dp, tt, okm);
            fetch_rows<T, kWin>(wg, gp, tt, okm);
            float wz[kWin], wy[kWin];
            fetch_rows<T, kWin>(wz, zp, tt, okm);
            fetch_rows<T, kWin>(wy, yp, tt, okm);
#pragma unroll
            for (int j = 0; j < kWin; ++j) {
                const bool ok = (okm >> j) & 1u;
                float dl = wd[j] + bias;
                dl = softplus_on ? softplus20(dl) : dl;       // select, not a branch: keeps each step one basic block
                wd[j] = ok ? dl : 0.f;
                if (has_z) {
                    const float zz = wz[j], sg = sigmoidf(zz);
                    const float dzv = wg[j] * wy[j] * sg * fmaf(zz, 1.f - sg, 1.f);
                    if (ok) st_row<T>(dzp, tt[j], dzv);
                    wg[j] *= zz * sg;
                }
                du[j] = Dv * wg[j];
                dd[j] = 0.f;
                wdu[j] = wd[j] * wu[j];
                dD_acc = fmaf(wg[j], wu[j], dD_acc);
            }
        }
        SEGM_WAVE_LDS_SYNC();                                    // the previous window is done with s_bc / s_dbc / s_rows
        stage_park<kWin, NS, RW, false>(sb, lb, bt_fastest, it.r);
        stage_park<kWin, NS, RW, false>(scc, lc, ct_fastest, it.r);
        if (it.r < kWin) lrows[it.r] = (item_ok && tw.tau + it.r < tm.L) ? tw.ahead(tm, it.r) : -1;
        SEGM_WAVE_LDS_SYNC();

        // state entering the window = forward checkpoint (zero past the end of the sequence)
        const bool ck_ok = it.valid && tw.tau < tm.L;
        const float* ck = P.ckpt + ((int64_t)it.b * P.nck + (ck_ok ? tw.tau / kCkpt : 0)) * ((nstate + 1) / 2) * gm.dim * 2 + (int64_t)dsafe * 2;
        float A2n_next = Arow[0] * kLog2e;
        float hp_next = ck[0];                               // state n of the checkpoint: ck[(n / 2) * 2 dim + (n & 1)]
#pragma unroll 1
        for (int n = 0; n < NS; ++n) {                      // runtime loop over states
            const bool on = it.valid && n < nstate;
            const float A2n = on ? A2n_next : 0.f;
            const float hp = (on && ck_ok) ? hp_next : 0.f;
            {
                const int nn = (n + 1 < nstate) ? n + 1 : nstate - 1;          // prefetch the next state's scalars
                A2n_next = Arow[nn] * kLog2e;
                hp_next = ck[(int64_t)(nn >> 1) * gm.dim * 2 + (nn & 1)];
            }
            const float An = A2n * 0.6931471805599453f;
            float en = s_e[n][threadIdx.x];
            float dAn = s_dA[n][threadIdx.x];
            float a[kWin], h[kWin];
            const float4* B4 = reinterpret_cast<const float4*>(lb + n * kWin);
            const float4* C4 = reinterpret_cast<const float4*>(lc + n * kWin);
#pragma unroll
            for (int q = 0; q < kWin / 4; ++q) {
                const float4 bv = B4[q];
                const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = q * 4 + i;
                    a[j] = fast_exp2(wd[j] * A2n);
                    h[j] = fmaf(a[j], j ? h[j - 1] : hp, wdu[j] * bb[i]);
                }
            }
            float v[2 * kWin];                              // [0,16): dB_j ; [16,32): dC_j   (this state)
#pragma unroll
            for (int qq = 0; qq < kWin / 4; ++qq) {
                const int q = kWin / 4 - 1 - qq;
                const float4 bv = B4[q], cv = C4[q];
                const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
                const float cc[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int i = 3 - ii, j = q * 4 + i;
                    const float dh = fmaf(wg[j], cc[i], en);
                    const float t2 = dh * (j ? h[j - 1] : hp) * a[j];
                    dAn = fmaf(t2, wd[j], dAn);
                    const float qv = dh * bb[i];
                    dd[j] = fmaf(t2, An, fmaf(qv, wu[j], dd[j]));
                    du[j] = fmaf(qv, wd[j], du[j]);
                    v[j] = dh * wdu[j];
                    v[kWin + j] = wg[j] * h[j];
                    en = a[j] * dh;
                }
            }
            // sum the dB / dC contributions over the channels (lanes) of the work item
            if constexpr (RW >= 32) {
                reduce_scatter<RW, V>(v, it.r);
                if (it.r < 32) {
                    const int j = it.r & (kWin - 1);        // lanes 0..15 -> dB_j, 16..31 -> dC_j
                    (it.r < kWin ? ldb : ldc)[j * NS + n] = v[0];
                }
            } else {
                float vb[kWin], vc[kWin];
#pragma unroll
                for (int j = 0; j < kWin; ++j) { vb[j] = v[j]; vc[j] = v[kWin + j]; }
                reduce_scatter<RW, kWin>(vb, it.r);
                reduce_scatter<RW, kWin>(vc, it.r);
                ldb[it.r * NS + n] = vb[0];
                ldc[it.r * NS + n] = vc[0];
            }
            s_e[n][threadIdx.x] = en;
            s_dA[n][threadIdx.x] = dAn;
        }
        SEGM_WAVE_LDS_SYNC();                                    // the dB / dC tile of every item is complete
        // flush the window's dB / dC tile: contiguous in whichever of (t, n) has the smaller stride
        {
            const bool n_fast = P.dB_sn <= P.dB_st;
#pragma unroll
            for (int i = 0; i < kWin * NS / RW; ++i) {
                const int el = it.r + i * RW;
                int j, n;
                if (n_fast) { j = el / NS; n = el - j * NS; } else { n = el / kWin; j = el - n * kWin; }
                const int32_t row = lrows[j];
                if (row >= 0 && n < nstate) {
                    const int64_t ob = (int64_t)it.b * P.dB_sb + row_off(row, P.dB_st) + (int64_t)n * P.dB_sn;
                    const int64_t oc = (int64_t)it.b * P.dC_sb + row_off(row, P.dC_st) + (int64_t)n * P.dC_sn;
                    const float xb = ldb[j * NS + n], xc = ldc[j * NS + n];
                    if (P.atomic_bc) { atomicAdd(P.dB + ob, xb); atomicAdd(P.dC + oc, xc); }
                    else { P.dB[ob] = xb; P.dC[oc] = xc; }
                }
            }
        }
        {
            TimeIter ts = tw;
#pragma unroll
            for (int j = 0; j < kWin; ++j) {
                if ((okm >> j) & 1u) {
                    float ddv = dd[j];
                    ddv *= softplus_on ? 1.f - fast_exp(-wd[j]) : 1.f;       // sigmoid(raw) = 1 - exp(-softplus(raw))
                    dbias_acc += ddv;
                    st_row<T>(dup, ts.t, du[j]);
                    st_row<T>(ddp, ts.t, ddv);
                }
                ts.next(tm);
            }
        }
        tw.jump(tm, -kWin);
    }
    if (it.valid) {
        const int64_t row = crow * (nstate + 2);
#pragma unroll
        for (int n = 0; n < NS; ++n)
            if (n < nstate) P.part[(row + n) * gm.dim + it.d] = s_dA[n][threadIdx.x];
        P.part[(row + nstate) * gm.dim + it.d] = dD_acc;
        P.part[(row + nstate + 1) * gm.dim + it.d] = dbias_acc;
    }
}

// zero a strided (batch, time, state) fp32 view (dB / dC accumulate atomically across d-tiles)
__global__ void __launch_bounds__(256) clear_bc_kernel(float* p, int64_t sb, int64_t st, int64_t sn, int32_t L, int nstate) {
    const int b = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)L * nstate;
    if (i < total) {
        const int32_t t = (int32_t)(i / nstate);
        const int n = (int)(i - (int64_t)t * nstate);
        p[(int64_t)b * sb + row_off(t, st) + (int64_t)n * sn] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
struct BwdWs { size_t sd, e, carry, part, seg, slab, total; };
static BwdWs bwd_ws_layout(int batch, int dim, int nstate, int64_t L, int chunk) {
    const int64_t nch = (L + chunk - 1) / chunk;
    BwdWs w;
    w.sd = 0;
    w.e = align256((size_t)batch * nch * dim * sizeof(float));
    w.carry = w.e + align256((size_t)batch * nch * nstate * dim * sizeof(float));
    w.part = w.carry + align256((size_t)batch * nch * nstate * dim * sizeof(float));
    w.seg = w.part + align256((size_t)batch * nch * (nstate + 2) * dim * sizeof(float));
    w.slab = w.seg + align256(scan_carry_scratch_bytes(batch, dim, nstate, nch));
    w.total = w.slab + scan_bwd_w8_slab_bytes(batch, dim, nstate, L);      // one B / C group (the regular-shape kernels take no other)
    return w;
}

template <typename T, int NS, int RW>
static int launch_bwd_rw(const ScanDev& P, hipStream_t stream) {
    constexpr int TS = 8;
    const Geom& gm = P.gm;
    const unsigned nblocks = (unsigned)((gm.nwaves + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL((scan_bwd_agg_kernel<T, NS, TS, RW>), dim3(nblocks), dim3(kBlock), 0, stream, P);
    {
        ScanDevN PP;
        memset(&PP, 0, sizeof(PP));
        PP.d[0] = P;
        launch_scan_carry(PP, 1, true, stream);
    }
    hipLaunchKernelGGL((scan_bwd_main_kernel<T, NS, RW>), dim3(nblocks), dim3(kBlock), 0, stream, P);
    return (int)hipGetLastError();
}

template <typename T, int NS>
static int launch_bwd(const ScanDev& P, hipStream_t stream) {
    if (P.gm.rw == 64) return launch_bwd_rw<T, NS, 64>(P, stream);
    if (P.gm.rw == 32) return launch_bwd_rw<T, NS, 32>(P, stream);
    return launch_bwd_rw<T, NS, 16>(P, stream);
}

// SEGM_SCAN_FAST=0 forces the general kernels (A/B timing, and the tests that must exercise them on regular shapes)
static bool use_fast_bwd() {
    const char* e = getenv("SEGM_SCAN_FAST");
    return !(e && e[0] == '0');
}

template <typename T>
static int launch_bwd_ns(const ScanDev& P, hipStream_t stream) {
    if (P.gm.nstate <= 4) return launch_bwd<T, 4>(P, stream);
    if (P.gm.nstate <= 8) return launch_bwd<T, 8>(P, stream);
    return launch_bwd<T, 16>(P, stream);
}

}  // namespace segm

using namespace segm;

extern "C" size_t segm_selective_scan_bwd_workspace_bytes(int32_t batch, int32_t dim, int32_t dstate, int64_t seqlen,
                                                          int32_t chunk) {
    if (batch <= 0 || dim <= 0 || dstate <= 0 || seqlen <= 0) return 0;
    if (chunk <= 0) chunk = default_chunk(batch, dim, seqlen);
    return bwd_ws_layout(batch, dim, dstate, seqlen, chunk).total;
}

// one backward launch (see scan_fwd.hip's scan_fwd_one): `batched` != null hands back the argument block of a launch the
// regular-shape kernels take instead of launching (the dB / dC clears, if any, are enqueued either way)
static int scan_bwd_one(const segm_scan_bwd_args* b, ScanDev* batched) {
    if (!b) return SEGM_E_NULL;
    const segm_scan_fwd_args* a = &b->f;
    int rc = validate_scan_common(a);
    if (rc != SEGM_OK) return rc;
    if (a->chunk <= 0) return SEGM_E_SHAPE;                 // must be the forward's chunk
    if (a->conv_width != 0) return SEGM_E_SHAPE;            // the backward takes the conv OUTPUT as u (forward-only option)
    if (!a->ckpt || !b->dout.ptr || !b->du.ptr || !b->ddelta.ptr || !b->dA || !b->dB.ptr || !b->dC.ptr) return SEGM_E_NULL;
    if (a->z.ptr && (!b->dz.ptr || !a->out.ptr)) return SEGM_E_NULL;
    const segm_seq* sv[4] = {&b->dout, &b->du, &b->ddelta, &b->dz};
    for (const segm_seq* s : sv)
        if (s->ptr && (s->stride_t < 0 || s->stride_t >= ((int64_t)1 << 31))) return SEGM_E_SHAPE;
    if (b->dB.stride_t < 0 || b->dB.stride_t >= ((int64_t)1 << 31) || b->dC.stride_t < 0 ||
        b->dC.stride_t >= ((int64_t)1 << 31))
        return SEGM_E_SHAPE;
    const int chunk = a->chunk;
    const BwdWs ws = bwd_ws_layout(a->batch, a->dim, a->dstate, a->seqlen, chunk);
    if (!b->workspace || b->workspace_bytes < ws.total) return SEGM_E_WORKSPACE;

    const int G = a->n_groups, Dg = a->dim / G, N = a->dstate;
    const size_t es = dtype_size(a->dtype);
    const int64_t nch = (a->seqlen + chunk - 1) / chunk;
    hipStream_t stream = (hipStream_t)a->stream;
    char* wsb = (char*)b->workspace;
    const segm_seq* all[8] = {&a->u, &a->delta, &a->z, &a->out, &b->dout, &b->du, &b->ddelta, &b->dz};
    const segm_bc* bv[2] = {&a->B, &a->C};
    const segm_bc* gv[2] = {&b->dB, &b->dC};                            // fp32, or the tensors' type (dbc_native)

    for (int g = 0; g < G; ++g) {
        const int64_t d0 = (int64_t)g * Dg;
        ScanDev P;
        fill_scan_dev(P, a, g, chunk);
        P.agg_sd = (float*)(wsb + ws.sd) + (size_t)g * a->batch * nch * Dg;
        P.agg_h = (float*)(wsb + ws.e) + (size_t)g * a->batch * nch * N * Dg;
        P.carry = (float*)(wsb + ws.carry) + (size_t)g * a->batch * nch * N * Dg;
        P.part = (float*)(wsb + ws.part) + (size_t)g * a->batch * nch * (N + 2) * Dg;
        P.carry_seg = (float*)(wsb + ws.seg) + (size_t)g * a->batch * ((nch + kCarrySeg - 1) / kCarrySeg) * (N + 1) * Dg;
        P.dout = seq_at(b->dout, d0, es); P.du = seq_at(b->du, d0, es);
        P.ddelta = seq_at(b->ddelta, d0, es); P.dz = seq_at(b->dz, d0, es);
        P.dB = (float*)b->dB.ptr + (int64_t)g * b->dB.stride_g;
        P.dB_sb = b->dB.stride_b; P.dB_st = b->dB.stride_t; P.dB_sn = b->dB.stride_n;
        P.dC = (float*)b->dC.ptr + (int64_t)g * b->dC.stride_g;
        P.dC_sb = b->dC.stride_b; P.dC_st = b->dC.stride_t; P.dC_sn = b->dC.stride_n;
        bool fast = use_fast_bwd() && G == 1 && scan_bwd_fast_shape(P, es);
        if (fast) {
            // ONE predicate for "the regular-shape kernels run" (ADVICE r04): the shape AND the 32-bit spans their buffer addressing
            // needs - a view beyond 4 GiB per batch element falls back to the general kernels (64-bit addresses) instead of failing
            const int64_t fspan = fast_span_rows(P);
            if (validate_spans(all, 8, bv, 2, a->dim, a->dstate, a->seqlen, es, fspan) != SEGM_OK ||
                validate_spans(nullptr, 0, gv, 2, a->dim, a->dstate, a->seqlen, b->dbc_native ? es : sizeof(float), fspan) != SEGM_OK)
                fast = false;
        }
        P.dbc_part = (float*)(wsb + ws.slab);
        // the regular-shape main kernel leaves one fp32 slab of dB / dC per d-tile, added in a fixed order by a second kernel; the
        // general kernels add one partial per d-tile atomically onto a zeroed fp32 buffer
        P.atomic_bc = !fast && P.gm.ndt > 1;
        P.dbc_native = b->dbc_native != 0;
        if (P.dbc_native && !fast) return SEGM_E_SHAPE;
        const int64_t span = fast ? fast_span_rows(P) : 0;
        rc = validate_spans(all, 8, bv, 2, a->dim, a->dstate, a->seqlen, es, span);
        if (rc != SEGM_OK) return rc;
        rc = validate_spans(nullptr, 0, gv, 2, a->dim, a->dstate, a->seqlen, P.dbc_native ? es : sizeof(float), span);
        if (rc != SEGM_OK) return rc;
        if (batched && (!fast || G != 1)) return SEGM_E_SHAPE;          // the caller falls back to one launch per block
        if (P.atomic_bc) {
            const int64_t total = a->seqlen * N;
            dim3 cg((unsigned)((total + 255) / 256), a->batch);
            hipLaunchKernelGGL(clear_bc_kernel, cg, dim3(256), 0, stream, P.dB, P.dB_sb, P.dB_st, P.dB_sn, (int32_t)a->seqlen, N);
            hipLaunchKernelGGL(clear_bc_kernel, cg, dim3(256), 0, stream, P.dC, P.dC_sb, P.dC_st, P.dC_sn, (int32_t)a->seqlen, N);
        }
        if (batched) {
            *batched = P;
            return SEGM_OK;
        }
        if (fast) {                                        // regular shapes (every SegMamba stage): scan_bwd_fast.hip
            ScanDevN PP;
            memset(&PP, 0, sizeof(PP));
            PP.d[0] = P;
            launch_scan_bwd_fast(PP, 1, a->dtype, false, stream);
            launch_scan_carry(PP, 1, true, stream);
            launch_scan_bwd_fast(PP, 1, a->dtype, true, stream);
            rc = (int)hipGetLastError();
        } else if (a->dtype == SEGM_F32) rc = launch_bwd_ns<float>(P, stream);
        else if (a->dtype == SEGM_F16) rc = launch_bwd_ns<f16_t>(P, stream);
        else rc = launch_bwd_ns<bf16_t>(P, stream);
        if (rc != 0) return rc;
        launch_reduce_partials(P.part, (int64_t)a->batch * nch, N + 2, Dg, b->dA + d0 * N, N,
                               b->dD ? b->dD + d0 : nullptr, b->ddelta_bias ? b->ddelta_bias + d0 : nullptr, stream);
    }
    return (int)hipGetLastError();
}

extern "C" int segm_selective_scan_bwd(const segm_scan_bwd_args* b) { return scan_bwd_one(b, nullptr); }

extern "C" int segm_selective_scan_bwd_deterministic(const segm_scan_bwd_args* b) {
    if (!b || !use_fast_bwd()) return 0;
    const segm_scan_fwd_args* a = &b->f;
    if (validate_scan_common(a) != SEGM_OK || a->chunk <= 0 || a->n_groups <= 0) return 0;
    if (a->n_groups != 1) return 0;
    ScanDev P;
    fill_scan_dev(P, a, 0, a->chunk);
    const size_t es = dtype_size(a->dtype);
    if (!scan_bwd_fast_shape(P, es)) return 0;
    // the same span bound scan_bwd_one applies, on the tensors this query can see (the forward's; the gradient tensors are laid
    // out like them by every caller in this repository - if they are not, segm_selective_scan_bwd answers SEGM_E_SHAPE for
    // dbc_native and falls back to the general kernels otherwise)
    const segm_seq* fw[4] = {&a->u, &a->delta, &a->z, &a->out};
    const segm_bc* bv[2] = {&a->B, &a->C};
    return validate_spans(fw, 4, bv, 2, a->dim, a->dstate, a->seqlen, es, fast_span_rows(P)) == SEGM_OK ? 1 : 0;
}

extern "C" int segm_selective_scan_bwd_multi(const segm_scan_bwd_args* args, int32_t n) {
    if (!args || n <= 0) return SEGM_E_NULL;
    bool batch = n > 1 && n <= kMaxDirs && use_fast_bwd();
    for (int i = 1; batch && i < n; ++i) batch = scan_same_launch(&args[0].f, &args[i].f);
    if (batch) {                                           // probe WITHOUT side effects first: all blocks must be regular
        ScanDevN PP;
        memset(&PP, 0, sizeof(PP));
        int rc = SEGM_OK;
        for (int i = 0; i < n && rc == SEGM_OK; ++i) rc = scan_bwd_one(&args[i], &PP.d[i]);
        if (rc == SEGM_OK) {
            const segm_scan_fwd_args* a = &args[0].f;
            hipStream_t stream = (hipStream_t)a->stream;
            launch_scan_bwd_fast(PP, n, a->dtype, false, stream);
            launch_scan_carry(PP, n, true, stream);
            launch_scan_bwd_fast(PP, n, a->dtype, true, stream);
            const int64_t nch = (a->seqlen + a->chunk - 1) / a->chunk;
            ReduceN R;                                     // one launch for the n directions (scan_same_launch: one geometry)
            memset(&R, 0, sizeof(R));
            for (int i = 0; i < n; ++i) {
                R.part[i] = PP.d[i].part; R.out0[i] = args[i].dA; R.out1[i] = args[i].dD; R.out2[i] = args[i].ddelta_bias;
            }
            launch_reduce_partials_multi(R, n, (int64_t)a->batch * nch, a->dstate + 2, a->dim, a->dstate, stream);
            return (int)hipGetLastError();
        }
        if (rc != SEGM_E_SHAPE) return rc;                  // (a clear enqueued for an earlier block is repeated below: harmless)
    }
    for (int i = 0; i < n; ++i) {
        const int rc = scan_bwd_one(&args[i], nullptr);
        if (rc != SEGM_OK) return rc;
    }
    return SEGM_OK;
}
