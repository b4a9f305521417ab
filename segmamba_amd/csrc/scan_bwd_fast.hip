// Selective scan backward, the "regular shape" kernels (same algorithm, workspace and checkpoints as scan_bwd.hip's
// K1 / K3; see scan_fwd_fast.hip for what "regular" means - here additionally B, C, dB, dC must be state-fastest, the
// channel-last production layout).
//
//   * addressing: wave-uniform base + constant per-lane offset, no masks, no per-lane time iterators;
//   * the state loop handles two states per iteration in packed fp32.  Per (step, state) that is 6 packed
//     instructions + 1 v_exp instead of 17 scalar ones + 1:
//         forward   a = exp2(delta A2)            h = a h_prev + (delta u) B
//         backward  dh = g C + e                  t2 = dh h_prev a          dA += t2 delta     q += dh B
//                   ddA += t2 A                   dB_c = dh (delta u)       dC_c = g h         e = a dh
//     (du = D g + delta q and ddelta = ddA + u q are formed once per window from q = sum_n dh B);
//   * the dB / dC contributions overwrite the registers of a / h as the backward walk frees them, so the pair loop
//     needs no additional arrays, and are summed over the channels with the LDS-free reduce-scatter of scan_bwd.hip.
#include "scan_fast.h"

namespace segm {

constexpr int kFW = 16;     // window = spacing of the forward checkpoints

// ------------------------------------------------------------------------------------------------------
// K1 (regular shapes): reverse chunk aggregates
// ------------------------------------------------------------------------------------------------------
template <typename T, int RW>
__global__ void __launch_bounds__(kBlock) scan_bwd_agg_fast_kernel(ScanDev P) {
    constexpr int G = 64 / RW, EPL = FastStage<RW>::EPL;
    __shared__ __attribute__((aligned(16))) float s_c[2][kWavesPerBlock][G][kFT * kFS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Geom& gm = P.gm;
    const Item it = locate(gm, (int64_t)blockIdx.x * kWavesPerBlock + wave, lane);
    if (!it.wave_valid) return;                           // the last workgroup may have spare waves (no workgroup barriers here)
    const int ub = uniform_batch(it);
    const bool softplus_on = P.delta_softplus != 0;
    const bool has_z = P.z.p != nullptr;
    const int32_t dT = P.tm.ns > 1 ? P.tm.sA : P.tm.sA + P.tm.sW;
    const int32_t t_item = fast_item_row(P.tm, it.chunk * gm.chunk);

    f2 A2[kFS / 2], e[kFS / 2];
#pragma unroll
    for (int n = 0; n < kFS / 2; ++n) {
        A2[n] = f2{P.A[(int64_t)it.d * kFS + 2 * n] * kLog2e, P.A[(int64_t)it.d * kFS + 2 * n + 1] * kLog2e};
        e[n] = f2{0.f, 0.f};
    }
    const float bias = P.delta_bias ? P.delta_bias[it.d] : 0.f;
    const FastRow dp = fast_row<T>(P.delta, ub, t_item, it.d);
    const FastRow gp = fast_row<T>(P.dout, ub, t_item, it.d);
    const FastRow zp = fast_row<T>(has_z ? P.z : P.dout, ub, t_item, it.d);
    const FastStage<RW> sc = fast_stage<T, RW>(P.Cm, ub, t_item, dT, it.r);

    const int nsub = gm.chunk / kFT;
    float nd[kFT], ng[kFT], nz[kFT], nc[EPL];
    {
        const int32_t U = fast_U_of(P.tm, nsub - 1);
        fast_fetch<T>(nd, dp, U, dT);
        fast_fetch<T>(ng, gp, U, dT);
        fast_fetch<T>(nz, zp, U, dT);
        fast_stage_fetch<T, RW>(nc, sc, U);
    }
    float sumd = 0.f;
    int buf = 0;
    for (int s = nsub - 1; s >= 0; --s) {
        float* lc = &s_c[buf][wave][it.gi][0];
        fast_stage_park<RW>(nc, sc, lc);
        SEGM_WAVE_LDS_SYNC();
        float cd[kFT], cg[kFT], cz[kFT];
#pragma unroll
        for (int j = 0; j < kFT; ++j) { cd[j] = nd[j]; cg[j] = ng[j]; cz[j] = nz[j]; }
        const int32_t Un = fast_U_of(P.tm, s > 0 ? s - 1 : 0);     // prefetch the next (lower) sub-tile
        fast_fetch<T>(nd, dp, Un, dT);
        fast_fetch<T>(ng, gp, Un, dT);
        fast_fetch<T>(nz, zp, Un, dT);
        fast_stage_fetch<T, RW>(nc, sc, Un);
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
__global__ void __launch_bounds__(kBlock, 1) scan_bwd_main_fast_kernel(ScanDev P) {
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

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
bool scan_bwd_fast_shape(const ScanDev& P) {
    if (!scan_fast_shape(P)) return false;
    // state-fastest B, C, dB, dC (the channel-last production layout)
    return P.Bm.sn < P.Bm.st && P.Cm.sn < P.Cm.st && P.dB_sn < P.dB_st && P.dC_sn < P.dC_st;
}

template <typename T, int RW>
static void launch_bwd_fast_rw(const ScanDev& P, bool main, hipStream_t stream) {
    const unsigned nblocks = (unsigned)((P.gm.nwaves + kWavesPerBlock - 1) / kWavesPerBlock);
    if (main) hipLaunchKernelGGL((scan_bwd_main_fast_kernel<T, RW>), dim3(nblocks), dim3(kBlock), 0, stream, P);
    else hipLaunchKernelGGL((scan_bwd_agg_fast_kernel<T, RW>), dim3(nblocks), dim3(kBlock), 0, stream, P);
}
template <typename T>
static void launch_bwd_fast_t(const ScanDev& P, bool main, hipStream_t stream) {
    if (P.gm.rw == 64) launch_bwd_fast_rw<T, 64>(P, main, stream);
    else if (P.gm.rw == 32) launch_bwd_fast_rw<T, 32>(P, main, stream);
    else launch_bwd_fast_rw<T, 16>(P, main, stream);
}
// launches K1 (main == false) or K3 (main == true) of the regular-shape backward
void launch_scan_bwd_fast(const ScanDev& P, int dtype, bool main, hipStream_t stream) {
    if (dtype == SEGM_F32) launch_bwd_fast_t<float>(P, main, stream);
    else if (dtype == SEGM_F16) launch_bwd_fast_t<f16_t>(P, main, stream);
    else launch_bwd_fast_t<bf16_t>(P, main, stream);
}

}  // namespace segm
