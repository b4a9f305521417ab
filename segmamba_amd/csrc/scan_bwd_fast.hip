// Selective scan backward, the "regular shape" kernels (same algorithm and workspace as scan_bwd.hip's K1 / K3; see
// scan_fwd_fast.hip for what "regular" means): the reverse aggregate pass here, the main kernel in scan_bwd_w8.hip.
// (Rounds 2 and 3 kept two more main kernels in this directory - 16-step windows at one wave per SIMD, and an LDS-tile version at
// two; both are superseded by the 8-step-window kernel and live on in the history and in tools/experiments/.)
#include <stdlib.h>

#include "scan_fast.h"

namespace segm {

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
// host side
// ------------------------------------------------------------------------------------------------------
bool scan_bwd_fast_shape(const ScanDev& P, size_t esize) { return scan_bwd_w8_shape(P, esize); }

template <typename T, int RW>
static void launch_bwd_agg_rw(const ScanDevN& PP, int ndir, hipStream_t stream) {
    const unsigned nblocks = (unsigned)((PP.d[0].gm.nwaves + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL((scan_bwd_agg_fast_kernel<T, RW>), dim3(nblocks, ndir), dim3(kBlock), 0, stream, PP);
}
template <typename T>
static void launch_bwd_agg_t(const ScanDevN& PP, int ndir, hipStream_t stream) {
    if (PP.d[0].gm.rw == 64) launch_bwd_agg_rw<T, 64>(PP, ndir, stream);
    else if (PP.d[0].gm.rw == 32) launch_bwd_agg_rw<T, 32>(PP, ndir, stream);
    else launch_bwd_agg_rw<T, 16>(PP, ndir, stream);
}
// launches K1 (main == false) or K3 (main == true) of the regular-shape backward for `ndir` blocks of one geometry
void launch_scan_bwd_fast(const ScanDevN& PP, int ndir, int dtype, bool main, hipStream_t stream) {
    if (main) {
        launch_scan_bwd_main_w8(PP, ndir, dtype, stream);
        return;
    }
    if (dtype == SEGM_F32) launch_bwd_agg_t<float>(PP, ndir, stream);
    else if (dtype == SEGM_F16) launch_bwd_agg_t<f16_t>(PP, ndir, stream);
    else launch_bwd_agg_t<bf16_t>(PP, ndir, stream);
}

}  // namespace segm
