// Selective scan backward, regular shapes: the main kernel on 16-step windows ("pair" kernel: two states per iteration in packed
// fp32, one wave per SIMD, per-batch base + 32-bit row offsets).  The default main kernel.
//
// Round 3 built two alternatives with more waves per SIMD (scan_bwd_fast.hip: LDS-tile prefetch and a per-state loop at two waves;
// tools/experiments/scan_bwd_halfwindow_r3.hip.txt: 8-step half windows at three).  Both are correct and SLOWER on the MI355X
// (1.2 - 1.5 ms against 0.90 ms at the stage-0 shape, profiles/r03_scan_ab*.log), and round 3 also found why: a SIMD retires about
// one instruction per 4.4 cycles (v_exp_f32: 9) however many waves it picks them from - this kernel runs at 2 300 cycles per
// wave-step for its ~510 instructions with ONE wave per SIMD, and exactly as fast per wave with one to eight waves of work per SIMD
// (profiles/r03_scan_occupancy.log).  More resident waves buy nothing; only fewer instructions per step do, and the multi-wave
// kernels issue more (recomputed exponentials, LDS staging, spills).  SEGM_BWD_MAIN=r3 selects the other kernel, and spans beyond
// 4 GiB (fp32 at 2^24 steps) always take it (it addresses from the wave's lowest row).
#include "scan_fast.h"

namespace segm {

constexpr int kFW = 16;     // window = spacing of the forward checkpoints

// ------------------------------------------------------------------------------------------------------
// K3 (regular shapes): main backward kernel
// ------------------------------------------------------------------------------------------------------
// the 16 rows of a window: two affine halves of 8
template <typename T>
__device__ __forceinline__ void win_fetch(float (&dst)[kFW], const FastRow& r, int32_t U0, int32_t U1, int32_t dT) {
    const char* p0 = r.base + (int64_t)U0 * r.stb;
    const char* p1 = r.base + (int64_t)U1 * r.stb;
    const int64_t inc = (int64_t)dT * r.stb;
#pragma unroll
    for (int j = 0; j < kFT; ++j) {
        dst[j] = to_f32(*reinterpret_cast<const T*>(p0 + (int64_t)j * inc + r.loff));
        dst[kFT + j] = to_f32(*reinterpret_cast<const T*>(p1 + (int64_t)j * inc + r.loff));
    }
}
template <typename T>
__device__ __forceinline__ void win_store(const float (&src)[kFW], const FastRow& r, int32_t U0, int32_t U1, int32_t dT) {
    char* p0 = const_cast<char*>(r.base) + (int64_t)U0 * r.stb;
    char* p1 = const_cast<char*>(r.base) + (int64_t)U1 * r.stb;
    const int64_t inc = (int64_t)dT * r.stb;
#pragma unroll
    for (int j = 0; j < kFT; ++j) {
        *reinterpret_cast<T*>(p0 + (int64_t)j * inc + r.loff) = from_f32<T>(src[j]);
        *reinterpret_cast<T*>(p1 + (int64_t)j * inc + r.loff) = from_f32<T>(src[kFT + j]);
    }
}

template <typename T, int RW>
__global__ void __launch_bounds__(kBlock, 1) scan_bwd_main_pair_kernel(ScanDevN PP) {
    const ScanDev& P = PP.d[blockIdx.y];
    constexpr int G = 64 / RW, EPL = FastStage<RW>::EPL;
    constexpr int V = RW < 32 ? RW : 32;
    __shared__ __attribute__((aligned(16))) float s_bc[kWavesPerBlock][G][2][kFW * kFS];     // [s][n]: B then C
    __shared__ __attribute__((aligned(16))) float s_dbc[kWavesPerBlock][G][2][kFW * kFS];    // [j][n]: dB then dC
    __shared__ f2 s_e[kFS / 2][kBlock];                  // adjoint entering from the right, per thread and state pair
    __shared__ f2 s_dA[kFS / 2][kBlock];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Geom& gm = P.gm;
    const Item it = locate(gm, (int64_t)blockIdx.x * kWavesPerBlock + wave, lane);
    if (!it.wave_valid) return;                           // the last workgroup may have spare waves (no workgroup barriers here)
    const int ub = uniform_batch(it);
    const bool softplus_on = P.delta_softplus != 0;
    const bool has_z = P.z.p != nullptr;
    const int32_t dT = P.tm.ns > 1 ? P.tm.sA : P.tm.sA + P.tm.sW;
    const int32_t tau0 = it.chunk * gm.chunk;
    const int32_t t_item = fast_item_row(P.tm, tau0);

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

    const FastRow up = fast_row<T>(P.u, ub, t_item, it.d);
    const FastRow dp = fast_row<T>(P.delta, ub, t_item, it.d);
    const FastRow gp = fast_row<T>(P.dout, ub, t_item, it.d);
    const FastRow zp = fast_row<T>(has_z ? P.z : P.dout, ub, t_item, it.d);
    const FastRow yp = fast_row<T>(has_z ? P.out : P.dout, ub, t_item, it.d);
    const FastRow dup = fast_row<T>(P.du, ub, t_item, it.d);
    const FastRow ddp = fast_row<T>(P.ddelta, ub, t_item, it.d);
    const FastRow dzp = fast_row<T>(has_z ? P.dz : P.du, ub, t_item, it.d);
    const FastStage<RW> sb = fast_stage<T, RW>(P.Bm, ub, t_item, dT, it.r);
    const FastStage<RW> sc = fast_stage<T, RW>(P.Cm, ub, t_item, dT, it.r);
    // dB / dC flush (state-fastest fp32): lane r writes elements el = r + i RW of each 8-step half: j = el / 16, n = el % 16
    const int fj = it.r / kFS, fn = it.r % kFS;
    constexpr int FJ = RW >= kFS ? RW / kFS : 1;          // steps between a lane's consecutive elements
    char* dBb = reinterpret_cast<char*>(P.dB) + (int64_t)ub * P.dB_sb * 4;
    char* dCb = reinterpret_cast<char*>(P.dC) + (int64_t)ub * P.dC_sb * 4;
    const uint32_t dB_loff = (uint32_t)(t_item + fj * dT) * (uint32_t)(P.dB_st * 4) + (uint32_t)fn * (uint32_t)(P.dB_sn * 4);
    const uint32_t dC_loff = (uint32_t)(t_item + fj * dT) * (uint32_t)(P.dC_st * 4) + (uint32_t)fn * (uint32_t)(P.dC_sn * 4);
    float* lb = &s_bc[wave][it.gi][0][0];
    float* lc = &s_bc[wave][it.gi][1][0];
    float* ldb = &s_dbc[wave][it.gi][0][0];
    float* ldc = &s_dbc[wave][it.gi][1][0];
    const float* ckbase = P.ckpt + (((int64_t)it.b * P.nck + tau0 / kCkpt) * kFS) * gm.dim + it.d;

    const int nwin = gm.chunk / kFW;
    for (int w = nwin - 1; w >= 0; --w) {
        const int32_t U0 = fast_U_of(P.tm, 2 * w), U1 = fast_U_of(P.tm, 2 * w + 1);
        // ---- window data -------------------------------------------------------------------------------------
        float vb0[EPL], vb1[EPL], vc0[EPL], vc1[EPL];
        fast_stage_fetch<T, RW>(vb0, sb, U0);
        fast_stage_fetch<T, RW>(vb1, sb, U1);
        fast_stage_fetch<T, RW>(vc0, sc, U0);
        fast_stage_fetch<T, RW>(vc1, sc, U1);
        float wu[kFW], wd[kFW], wg[kFW], wdu[kFW];
        f2 qs[kFW], ddA[kFW];                             // sum over states of dh B and of t2 A, two partial sums each
        win_fetch<T>(wu, up, U0, U1, dT);
        win_fetch<T>(wd, dp, U0, U1, dT);
        win_fetch<T>(wg, gp, U0, U1, dT);
        {
            float wz[kFW], wy[kFW];
            if (has_z) {
                win_fetch<T>(wz, zp, U0, U1, dT);
                win_fetch<T>(wy, yp, U0, U1, dT);
            }
#pragma unroll
            for (int j = 0; j < kFW; ++j) {
                float dl = wd[j] + bias;
                wd[j] = softplus_on ? softplus20(dl) : dl;
                if (has_z) {
                    const float zz = wz[j], sg = sigmoidf(zz);
                    wz[j] = wg[j] * wy[j] * sg * fmaf(zz, 1.f - sg, 1.f);       // dz
                    wg[j] *= zz * sg;
                }
                wdu[j] = wd[j] * wu[j];
                qs[j] = f2{0.f, 0.f};
                ddA[j] = f2{0.f, 0.f};
                dD_acc = fmaf(wg[j], wu[j], dD_acc);
            }
            if (has_z) win_store<T>(wz, dzp, U0, U1, dT);
        }
        SEGM_WAVE_LDS_SYNC();                             // the previous window is done with s_bc / s_dbc
        fast_stage_park<RW>(vb0, sb, lb);
        fast_stage_park<RW>(vb1, sb, lb + kFT * kFS);
        fast_stage_park<RW>(vc0, sc, lc);
        fast_stage_park<RW>(vc1, sc, lc + kFT * kFS);
        SEGM_WAVE_LDS_SYNC();

        const float* ck = ckbase + (int64_t)w * kFS * gm.dim;      // state entering the window
#pragma unroll 1
        for (int p = 0; p < kFS / 2; ++p) {               // runtime loop over state pairs
            const f2 A2n = f2{Arow[2 * p], Arow[2 * p + 1]} * kLog2e;
            const f2 An = A2n * 0.6931471805599453f;
            const f2 hp = {ck[(int64_t)(2 * p) * gm.dim], ck[(int64_t)(2 * p + 1) * gm.dim]};
            f2 en = s_e[p][threadIdx.x];
            f2 dAn = s_dA[p][threadIdx.x];
            f2 a[kFW], h[kFW];
#pragma unroll
            for (int j = 0; j < kFW; ++j) {
                const f2 bb = *reinterpret_cast<const f2*>(lb + j * kFS + 2 * p);
                const f2 da = A2n * wd[j];
                a[j] = f2{fast_exp2(da.x), fast_exp2(da.y)};
                h[j] = a[j] * (j ? h[j - 1] : hp) + bb * wdu[j];
            }
#pragma unroll
            for (int jj = 0; jj < kFW; ++jj) {
                const int j = kFW - 1 - jj;
                const f2 bb = *reinterpret_cast<const f2*>(lb + j * kFS + 2 * p);
                const f2 cc = *reinterpret_cast<const f2*>(lc + j * kFS + 2 * p);
                const f2 dh = cc * wg[j] + en;
                const f2 t2 = dh * (j ? h[j - 1] : hp) * a[j];
                dAn = t2 * wd[j] + dAn;
                qs[j] = dh * bb + qs[j];
                ddA[j] = t2 * An + ddA[j];
                en = a[j] * dh;
                a[j] = dh * wdu[j];                        // dB contribution of (j, pair), in a's registers
                h[j] = h[j] * wg[j];                       // dC contribution, in h's registers
            }
            // sum the dB / dC contributions over the channels (lanes) of the work item, one state at a time
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int n = 2 * p + half;
                if constexpr (RW >= 32) {
                    float v[2 * kFW];
#pragma unroll
                    for (int j = 0; j < kFW; ++j) {
                        v[j] = half ? a[j].y : a[j].x;
                        v[kFW + j] = half ? h[j].y : h[j].x;
                    }
                    reduce_scatter<RW, V>(v, it.r);
                    if (it.r < 32) {
                        const int j = it.r & (kFW - 1);   // lanes 0..15 -> dB_j, 16..31 -> dC_j
                        (it.r < kFW ? ldb : ldc)[j * kFS + n] = v[0];
                    }
                } else {
                    float vb[kFW], vc[kFW];
#pragma unroll
                    for (int j = 0; j < kFW; ++j) { vb[j] = half ? a[j].y : a[j].x; vc[j] = half ? h[j].y : h[j].x; }
                    reduce_scatter<RW, kFW>(vb, it.r);
                    reduce_scatter<RW, kFW>(vc, it.r);
                    ldb[it.r * kFS + n] = vb[0];
                    ldc[it.r * kFS + n] = vc[0];
                }
            }
            s_e[p][threadIdx.x] = en;
            s_dA[p][threadIdx.x] = dAn;
        }
        SEGM_WAVE_LDS_SYNC();                             // the dB / dC tile of every item is complete
        {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int32_t Uh = hh ? U1 : U0;
                char* pb = dBb + (int64_t)Uh * (P.dB_st * 4);
                char* pc = dCb + (int64_t)Uh * (P.dC_st * 4);
#pragma unroll
                for (int i = 0; i < EPL; ++i) {
                    const int j = hh * kFT + fj + i * FJ;
                    float* ob = reinterpret_cast<float*>(pb + (int64_t)(i * FJ * dT) * (P.dB_st * 4) + dB_loff);
                    float* oc = reinterpret_cast<float*>(pc + (int64_t)(i * FJ * dT) * (P.dC_st * 4) + dC_loff);
                    const float xb = ldb[j * kFS + fn], xc = ldc[j * kFS + fn];
                    if (P.atomic_bc) { atomicAdd(ob, xb); atomicAdd(oc, xc); }
                    else { *ob = xb; *oc = xc; }
                }
            }
        }
        {
            float du[kFW], ddl[kFW];
#pragma unroll
            for (int j = 0; j < kFW; ++j) {
                const float q = qs[j].x + qs[j].y;
                du[j] = fmaf(wd[j], q, Dv * wg[j]);
                float ddv = fmaf(wu[j], q, ddA[j].x + ddA[j].y);
                ddv *= softplus_on ? 1.f - fast_exp(-wd[j]) : 1.f;       // sigmoid(raw) = 1 - exp(-softplus(raw))
                dbias_acc += ddv;
                ddl[j] = ddv;
            }
            win_store<T>(du, dup, U0, U1, dT);
            win_store<T>(ddl, ddp, U0, U1, dT);
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
static void launch_pair_rw(const ScanDevN& PP, int ndir, hipStream_t stream) {
    const unsigned nblocks = (unsigned)((PP.d[0].gm.nwaves + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL((scan_bwd_main_pair_kernel<T, RW>), dim3(nblocks, ndir), dim3(kBlock), 0, stream, PP);
}
template <typename T>
static void launch_pair_t(const ScanDevN& PP, int ndir, hipStream_t stream) {
    if (PP.d[0].gm.rw == 64) launch_pair_rw<T, 64>(PP, ndir, stream);
    else if (PP.d[0].gm.rw == 32) launch_pair_rw<T, 32>(PP, ndir, stream);
    else launch_pair_rw<T, 16>(PP, ndir, stream);
}
void launch_scan_bwd_main_pair(const ScanDevN& PP, int ndir, int dtype, hipStream_t stream) {
    if (dtype == SEGM_F32) launch_pair_t<float>(PP, ndir, stream);
    else if (dtype == SEGM_F16) launch_pair_t<f16_t>(PP, ndir, stream);
    else launch_pair_t<bf16_t>(PP, ndir, stream);
}

}  // namespace segm
