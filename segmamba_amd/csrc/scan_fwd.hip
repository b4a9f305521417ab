// Selective scan, forward (C ABI: segm_selective_scan_fwd).
//
// Replaces reference mamba/csrc/selective_scan/selective_scan_fwd_kernel.cuh:67-303 + selective_scan.cpp:226-336.
// The reference walks L serially inside one block per (batch, channel) and block-scans 2048-step
// tiles per state; on 256 CUs that is < 1 block per CU at SegMamba's stage 0.
// Here L is cut into chunks handled by independent lanes (one lane = one channel, state in registers):
//
//   K1  scan_fwd_agg_kernel    per (batch, chunk, channel): chunk end state from a zero start, sum(delta)
//   K2  scan_carry_kernel      per (batch, channel, state): compose the chunk aggregates into the state
//                              entering every chunk      (h_in[c+1] = exp(A * sum_delta[c]) * h_in[c] + H[c])
//   K3  scan_fwd_apply_kernel  per (batch, chunk, channel): re-run the chunk from its true entering state,
//                              y = <C, h> + D u, out_z = y silu(z); optionally checkpoint h every 16 steps
//
// The monoid (a1, b1) o (a0, b0) = (a1 a0, a1 b0 + b1) is the reference's SSMScanOp
// (selective_scan_common.h:110-115); the product of a's over a chunk is exp(A * sum of delta), so only the
// sum is stored.  All state / accumulation is fp32; I/O is fp32, fp16 or bf16.
#include <stdlib.h>
#include <string.h>

#include "scan_common.h"

namespace segm {

// ------------------------------------------------------------------------------------------------------
// K1: chunk aggregates
// ------------------------------------------------------------------------------------------------------
template <typename T, int NS, int TS, int RW>
__global__ void __launch_bounds__(kBlock) scan_fwd_agg_kernel(ScanDev P) {
    constexpr int G = 64 / RW;
    __shared__ __attribute__((aligned(16))) float s_b[2][kWavesPerBlock][G][TS * NS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Geom& gm = P.gm;
    const TimeMap tm = P.tm;
    const Item it = locate(gm, (int64_t)blockIdx.x * kWavesPerBlock + wave, lane);
    const int ub = uniform_batch(it);
    const bool item_ok = it.wave_valid && it.chunk < gm.nchunks;
    const int nstate = gm.nstate;
    const bool softplus_on = P.delta_softplus != 0;

    float A2[NS], h[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        A2[n] = (it.valid && n < nstate) ? P.A[(int64_t)it.d * nstate + n] * kLog2e : 0.f;
        h[n] = 0.f;
    }
    const float bias = (it.valid && P.delta_bias) ? P.delta_bias[it.d] : 0.f;
    const RowPtr up = make_rowptr<T>(P.u, ub, it.d, it.valid);
    const RowPtr dp = make_rowptr<T>(P.delta, ub, it.d, it.valid);
    const bool t_fastest = P.Bm.st <= P.Bm.sn;

    TimeIter ti;
    ti.seek(tm, item_ok ? it.chunk * gm.chunk : 0);

    float nu[TS], nd[TS];
    int32_t ntt[TS];
    StageRegs<TS, NS, RW> sb;
    uint32_t nok = row_indices<TS>(ntt, tm, ti, it.valid);
    fetch_rows<T, TS>(nu, up, ntt, nok);
    fetch_rows<T, TS>(nd, dp, ntt, nok);
    stage_fetch<T, TS, NS, RW>(sb, P.Bm, tm, ti, ub, nstate, it.r, item_ok);

    float sumd = 0.f;
    int buf = 0;
    for (int s0 = 0; s0 < gm.chunk; s0 += TS) {
        float* lb = &s_b[buf][wave][it.gi][0];
        stage_park<TS, NS, RW, true>(sb, lb, t_fastest, it.r);
        SEGM_WAVE_LDS_SYNC();
        float cu[TS], cd[TS];
#pragma unroll
        for (int j = 0; j < TS; ++j) { cu[j] = nu[j]; cd[j] = nd[j]; }
        const uint32_t cok = nok;
        ti.jump(tm, TS);
        // prefetch the next sub-tile (past the chunk end this reads the neighbour's / masked rows: harmless)
        nok = row_indices<TS>(ntt, tm, ti, it.valid);
        fetch_rows<T, TS>(nu, up, ntt, nok);
        fetch_rows<T, TS>(nd, dp, ntt, nok);
        stage_fetch<T, TS, NS, RW>(sb, P.Bm, tm, ti, ub, nstate, it.r, item_ok);
        // B rows are read from LDS one step ahead of their use so the LDS latency overlaps the previous step's math
        float4 bq[NS / 4];
#pragma unroll
        for (int q = 0; q < NS / 4; ++q) bq[q] = reinterpret_cast<const float4*>(lb)[q];
#pragma unroll
        for (int j = 0; j < TS; ++j) {
            const bool ok = (cok >> j) & 1u;
            float dl = cd[j] + bias;
            dl = softplus_on ? softplus20(dl) : dl;       // select, not a branch: keeps each step one basic block
            dl = ok ? dl : 0.f;
            const float dlu = dl * cu[j];
            sumd += dl;
            float4 bn[NS / 4];
            if (j + 1 < TS) {
#pragma unroll
                for (int q = 0; q < NS / 4; ++q) bn[q] = reinterpret_cast<const float4*>(lb + (j + 1) * NS)[q];
            }
#pragma unroll
            for (int q = 0; q < NS / 4; ++q) {
                const float bb[4] = {bq[q].x, bq[q].y, bq[q].z, bq[q].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = q * 4 + i;
                    const float a = fast_exp2(dl * A2[n]);
                    h[n] = fmaf(a, h[n], dlu * bb[i]);
                }
            }
#pragma unroll
            for (int n = 0; n < NS; ++n) SEGM_PIN_F32(h[n]);     // finish this step before the LDS reads two steps ahead
            if (j + 1 < TS) {
#pragma unroll
                for (int q = 0; q < NS / 4; ++q) bq[q] = bn[q];
            }
        }
        buf ^= 1;
    }
    if (it.valid) {
        const int64_t row = (int64_t)it.b * gm.nchunks + it.chunk;
        P.agg_sd[row * gm.dim + it.d] = sumd;
#pragma unroll
        for (int n = 0; n < NS; ++n)
            if (n < nstate) P.agg_h[(row * nstate + n) * gm.dim + it.d] = h[n];
    }
}

// ------------------------------------------------------------------------------------------------------
// K2: compose chunk aggregates into the state entering every chunk (REVERSE: chunks from last to first - the backward
// pass's adjoint entering from the right).  A latency problem, not a bandwidth one: per (batch, channel, state) a chain of
// one fma per chunk behind two loads.  Two launches of single-wave workgroups (lane = channel), one per
// (batch, channel tile, state, segment of kCarrySeg chunks), each holding its whole segment in registers behind ONE batch of
// loads:  A) fold the segment from a zero start -> (sum of delta, end state);  B) fold the composites of the segments
// before this one (again one batch), then walk the segment's chunks from that state, storing the state entering each.
// (Round 1's kernel walked 64 chunks per wave twice behind 16-deep single-buffered loads: 47 us at 1024 chunks.)
// ------------------------------------------------------------------------------------------------------
// grid: x = channel tile + (channel tiles) * segment, y = state, z = batch + batch * direction (segments on x: nstate * nseg
// on y overflowed 65535 for an explicit small chunk on a long sequence).
template <bool REVERSE, bool PHASE_B>
__global__ void __launch_bounds__(64) scan_carry_kernel(ScanDevN PP) {
    const int lane = threadIdx.x;
    const int nbatch = PP.d[0].gm.batch;
    const ScanDev& P = PP.d[blockIdx.z / nbatch];
    const float* __restrict__ agg_sd = P.agg_sd;
    const float* __restrict__ agg_h = P.agg_h;
    float* __restrict__ carry = P.carry;
    float* __restrict__ seg = P.carry_seg;
    const Geom& gm = P.gm;
    const int nstate = gm.nstate, nch = gm.nchunks;
    const int nseg = (nch + kCarrySeg - 1) / kCarrySeg;
    const int ndblk = (gm.dim + 63) / 64;
    const int d = (blockIdx.x % ndblk) * 64 + lane, b = blockIdx.z % nbatch;
    const int n = blockIdx.y, sg = blockIdx.x / ndblk;
    const bool valid = d < gm.dim;
    const int dd = valid ? d : 0;
    const float A2 = valid ? P.A[(int64_t)d * nstate + n] * kLog2e : 0.f;
    const float* sd_base = agg_sd + (int64_t)b * nch * gm.dim + dd;
    const float* h_base = agg_h + ((int64_t)b * nch * nstate + n) * gm.dim + dd;
    const int64_t h_stride = (int64_t)nstate * gm.dim;
    // segment composites: [batch][segment][nstate + 1][dim], row nstate = sum of delta
    float* seg_b = seg + (int64_t)b * nseg * (nstate + 1) * gm.dim + dd;
    const int64_t seg_stride = (int64_t)(nstate + 1) * gm.dim;

    float cin = 0.f;
    float psd[kCarrySeg], ph[kCarrySeg];                          // PHASE_B: composites of the earlier segments
    if (PHASE_B) {
#pragma unroll
        for (int i = 0; i < kCarrySeg; ++i) {
            const int s = i < sg ? i : 0;
            psd[i] = seg_b[(int64_t)s * seg_stride + (int64_t)nstate * gm.dim];
            ph[i] = seg_b[(int64_t)s * seg_stride + (int64_t)n * gm.dim];
        }
    }
    float sd[kCarrySeg], hh[kCarrySeg];
    const int q0 = sg * kCarrySeg;
#pragma unroll
    for (int i = 0; i < kCarrySeg; ++i) {
        const int q = q0 + i < nch ? q0 + i : nch - 1;            // clamped: a re-read, masked below
        const int c = REVERSE ? nch - 1 - q : q;
        sd[i] = sd_base[(int64_t)c * gm.dim];
        hh[i] = h_base[(int64_t)c * h_stride];
    }
    if (PHASE_B) {
#pragma unroll
        for (int i = 0; i < kCarrySeg; ++i)
            if (i < sg) cin = fmaf(fast_exp2(A2 * psd[i]), cin, ph[i]);
        for (int s = kCarrySeg; s < sg; ++s)                       // more than kCarrySeg^2 chunks: the remaining composites, in order
            cin = fmaf(fast_exp2(A2 * seg_b[(int64_t)s * seg_stride + (int64_t)nstate * gm.dim]), cin,
                       seg_b[(int64_t)s * seg_stride + (int64_t)n * gm.dim]);
    }
    float sds = 0.f;
#pragma unroll
    for (int i = 0; i < kCarrySeg; ++i) {
        const int q = q0 + i;
        if (q < nch) {
            const int c = REVERSE ? nch - 1 - q : q;
            if (PHASE_B && valid) carry[(((int64_t)b * nch + c) * nstate + n) * gm.dim + d] = cin;
            cin = fmaf(fast_exp2(A2 * sd[i]), cin, hh[i]);
            sds += sd[i];
        }
    }
    if (!PHASE_B && valid) {
        seg_b[(int64_t)sg * seg_stride + (int64_t)n * gm.dim] = cin;
        if (n == 0) seg_b[(int64_t)sg * seg_stride + (int64_t)nstate * gm.dim] = sds;
    }
    if (PHASE_B && !REVERSE && P.last_state && valid && sg == nseg - 1)
        P.last_state[(int64_t)b * P.last_state_sb + (int64_t)d * nstate + n] = cin;
}

size_t scan_carry_scratch_bytes(int batch, int dim, int nstate, int64_t nchunks) {
    const int64_t nseg = (nchunks + kCarrySeg - 1) / kCarrySeg;
    return align256((size_t)batch * nseg * (nstate + 1) * dim * sizeof(float));
}

void launch_scan_carry(const ScanDevN& PP, int ndir, bool reverse, hipStream_t stream) {
    const Geom& gm = PP.d[0].gm;
    const int nseg = (gm.nchunks + kCarrySeg - 1) / kCarrySeg;
    dim3 cgrid(((gm.dim + 63) / 64) * nseg, gm.nstate, gm.batch * ndir);
    if (reverse) {
        if (nseg > 1) hipLaunchKernelGGL((scan_carry_kernel<true, false>), cgrid, dim3(64), 0, stream, PP);
        hipLaunchKernelGGL((scan_carry_kernel<true, true>), cgrid, dim3(64), 0, stream, PP);
    } else {
        if (nseg > 1) hipLaunchKernelGGL((scan_carry_kernel<false, false>), cgrid, dim3(64), 0, stream, PP);
        hipLaunchKernelGGL((scan_carry_kernel<false, true>), cgrid, dim3(64), 0, stream, PP);
    }
}
static inline ScanDevN one_dir(const ScanDev& P) { ScanDevN PP; memset(&PP, 0, sizeof(PP)); PP.d[0] = P; return PP; }

// ------------------------------------------------------------------------------------------------------
// K3: apply
// ------------------------------------------------------------------------------------------------------
template <typename T, int NS, int TS, int RW>
__global__ void __launch_bounds__(kBlock) scan_fwd_apply_kernel(ScanDev P) {
    constexpr int G = 64 / RW;
    __shared__ __attribute__((aligned(16))) float s_bc[2][kWavesPerBlock][G][2][TS * NS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Geom& gm = P.gm;
    const TimeMap tm = P.tm;
    const Item it = locate(gm, (int64_t)blockIdx.x * kWavesPerBlock + wave, lane);
    const int ub = uniform_batch(it);
    const bool item_ok = it.wave_valid && it.chunk < gm.nchunks;
    const int nstate = gm.nstate;
    const bool softplus_on = P.delta_softplus != 0;
    const bool has_z = P.z.p != nullptr, has_out = P.out.p != nullptr;

    float A2[NS], h[NS];
    const int64_t crow = (int64_t)it.b * gm.nchunks + it.chunk;
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        const bool on = it.valid && n < nstate;
        A2[n] = on ? P.A[(int64_t)it.d * nstate + n] * kLog2e : 0.f;
        h[n] = on ? P.carry[(crow * nstate + n) * gm.dim + it.d] : 0.f;
    }
    const float bias = (it.valid && P.delta_bias) ? P.delta_bias[it.d] : 0.f;
    const float Dv = (it.valid && P.D) ? P.D[it.d] : 0.f;
    const RowPtr up = make_rowptr<T>(P.u, ub, it.d, it.valid);
    const RowPtr dp = make_rowptr<T>(P.delta, ub, it.d, it.valid);
    const Seq& zs = has_z ? P.z : P.u;                          // without a gate the z stream aliases u (unused)
    const RowPtr zp = make_rowptr<T>(zs, ub, it.d, it.valid);
    const RowPtr op = make_rowptr<T>(P.out, ub, it.d, it.valid && has_out);
    const RowPtr ozp = make_rowptr<T>(P.out_z, ub, it.d, it.valid && has_z);
    const bool t_fastest = P.Bm.st <= P.Bm.sn;

    TimeIter ti;
    ti.seek(tm, item_ok ? it.chunk * gm.chunk : 0);

    float nu[TS], nd[TS], nz[TS];
    int32_t ntt[TS];
    StageRegs<TS, NS, RW> sb, sc;
    uint32_t nok = row_indices<TS>(ntt, tm, ti, it.valid);
    fetch_rows<T, TS>(nu, up, ntt, nok);
    fetch_rows<T, TS>(nd, dp, ntt, nok);
    fetch_rows<T, TS>(nz, zp, ntt, nok);
    stage_fetch<T, TS, NS, RW>(sb, P.Bm, tm, ti, ub, nstate, it.r, item_ok);
    stage_fetch<T, TS, NS, RW>(sc, P.Cm, tm, ti, ub, nstate, it.r, item_ok);

    int buf = 0;
    for (int s0 = 0; s0 < gm.chunk; s0 += TS) {
        float* lb = &s_bc[buf][wave][it.gi][0][0];
        float* lc = &s_bc[buf][wave][it.gi][1][0];
        stage_park<TS, NS, RW, true>(sb, lb, t_fastest, it.r);
        stage_park<TS, NS, RW, true>(sc, lc, t_fastest, it.r);
        SEGM_WAVE_LDS_SYNC();
        float cu[TS], cd[TS], cz[TS];
        int32_t ctt[TS];               // rows of this sub-tile, for the stores
#pragma unroll
        for (int j = 0; j < TS; ++j) { cu[j] = nu[j]; cd[j] = nd[j]; cz[j] = nz[j]; ctt[j] = ntt[j]; }
        const uint32_t cok = nok;
        const int32_t tau0 = ti.tau;
        ti.jump(tm, TS);
        nok = row_indices<TS>(ntt, tm, ti, it.valid);
        fetch_rows<T, TS>(nu, up, ntt, nok);
        fetch_rows<T, TS>(nd, dp, ntt, nok);
        fetch_rows<T, TS>(nz, zp, ntt, nok);
        stage_fetch<T, TS, NS, RW>(sb, P.Bm, tm, ti, ub, nstate, it.r, item_ok);
        stage_fetch<T, TS, NS, RW>(sc, P.Cm, tm, ti, ub, nstate, it.r, item_ok);
        // state entering step tau0, every kCkpt steps (kept for the backward pass)
        if (P.ckpt && (tau0 % kCkpt) == 0 && it.valid && tau0 < tm.L) {
            const int64_t krow = (int64_t)it.b * P.nck + tau0 / kCkpt;
#pragma unroll
            for (int n = 0; n < NS; ++n)
                if (n < nstate) P.ckpt[((krow * ((nstate + 1) / 2) + n / 2) * gm.dim + it.d) * 2 + (n & 1)] = h[n];
        }
        float4 bq[NS / 4], cq[NS / 4];
#pragma unroll
        for (int q = 0; q < NS / 4; ++q) {
            bq[q] = reinterpret_cast<const float4*>(lb)[q];
            cq[q] = reinterpret_cast<const float4*>(lc)[q];
        }
#pragma unroll
        for (int j = 0; j < TS; ++j) {
            const bool ok = (cok >> j) & 1u;
            float dl = cd[j] + bias;
            dl = softplus_on ? softplus20(dl) : dl;       // select, not a branch: keeps each step one basic block
            dl = ok ? dl : 0.f;
            const float uu = cu[j];
            const float dlu = dl * uu;
            float y = Dv * uu;
            float4 bn[NS / 4], cn[NS / 4];
            if (j + 1 < TS) {                              // next step's B / C rows, one step ahead of their use
#pragma unroll
                for (int q = 0; q < NS / 4; ++q) {
                    bn[q] = reinterpret_cast<const float4*>(lb + (j + 1) * NS)[q];
                    cn[q] = reinterpret_cast<const float4*>(lc + (j + 1) * NS)[q];
                }
            }
#pragma unroll
            for (int q = 0; q < NS / 4; ++q) {
                const float bb[4] = {bq[q].x, bq[q].y, bq[q].z, bq[q].w};
                const float cc[4] = {cq[q].x, cq[q].y, cq[q].z, cq[q].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = q * 4 + i;
                    const float a = fast_exp2(dl * A2[n]);
                    h[n] = fmaf(a, h[n], dlu * bb[i]);
                    y = fmaf(cc[i], h[n], y);
                }
            }
            if (ok) {
                if (has_out) st_row<T>(op, ctt[j], y);
                if (has_z) {
                    const float zz = cz[j];
                    st_row<T>(ozp, ctt[j], y * zz * sigmoidf(zz));
                }
            }
#pragma unroll
            for (int n = 0; n < NS; ++n) SEGM_PIN_F32(h[n]);     // finish this step before the LDS reads two steps ahead
            if (j + 1 < TS) {
#pragma unroll
                for (int q = 0; q < NS / 4; ++q) { bq[q] = bn[q]; cq[q] = cn[q]; }
            }
        }
        buf ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
static int pick_rw(int dim) {
    if (dim % 64 == 0) return 64;
    if (dim % 32 == 0) return 32;
    if (dim % 16 == 0) return 16;
    int best = 64, best_pad = ((dim + 63) / 64) * 64;
    const int cand[2] = {32, 16};
    for (int c : cand) {
        int pad = ((dim + c - 1) / c) * c;
        if (pad < best_pad) { best = c; best_pad = pad; }
    }
    return best;
}

Geom make_geom(int batch, int dim, int nstate, int64_t L, int chunk) {   // L < 2^31 (validated by the caller)
    Geom g;
    g.batch = batch; g.dim = dim; g.nstate = nstate; g.L = (int32_t)L;
    g.rw = pick_rw(dim);
    // the lanes of a work item stage the 8 x NS block of B / C of a sub-tile: an item must not be wider than that block
    const int ns_pad = nstate <= 4 ? 4 : (nstate <= 8 ? 8 : 16);
    if (g.rw > 8 * ns_pad) g.rw = 8 * ns_pad;
    g.g = 64 / g.rw;
    g.ndt = (dim + g.rw - 1) / g.rw;
    g.chunk = chunk;
    g.nchunks = (int32_t)((L + chunk - 1) / chunk);
    g.ncg = (g.nchunks + g.g - 1) / g.g;
    g.nwaves = (int64_t)batch * g.ndt * g.ncg;
    return g;
}

int32_t default_chunk(int32_t batch, int32_t dim, int64_t L) {
    static int forced = [] { const char* e = getenv("SEGM_CHUNK"); return e ? atoi(e) : 0; }();   // experiments only
    if (forced >= kChunkQuantum && forced % kChunkQuantum == 0) return forced;
    // aim at ~3 waves per SIMD (256 CUs x 4 SIMDs) while keeping <= 4096 chunks for the carry kernel
    const double lanes_steps = (double)batch * (double)dim * (double)L;
    const int64_t c = (int64_t)(lanes_steps / (64.0 * 3072.0));
    int32_t chunk = kChunkQuantum;
    while ((int64_t)chunk * 2 <= c && chunk < 4096) chunk *= 2;
    while ((L + chunk - 1) / chunk > 4096 && chunk < (1 << 20)) chunk *= 2;
    return chunk;
}

struct FwdWs { size_t sd, h, carry, seg, total; };
static FwdWs fwd_ws_layout(int batch, int dim, int nstate, int64_t L, int chunk) {
    const int64_t nch = (L + chunk - 1) / chunk;
    FwdWs w;
    w.sd = 0;
    w.h = align256((size_t)batch * nch * dim * sizeof(float));
    w.carry = w.h + align256((size_t)batch * nch * nstate * dim * sizeof(float));
    w.seg = w.carry + align256((size_t)batch * nch * nstate * dim * sizeof(float));
    w.total = w.seg + scan_carry_scratch_bytes(batch, dim, nstate, nch);
    return w;
}

// sub-tile length (steps whose rows are prefetched together) of the apply kernel: 8, or 4 (fewer live registers,
// one more wave per SIMD, twice the barriers).  SEGM_APPLY_TS overrides for experiments.
static int apply_subtile() {
    static int ts = [] { const char* e = getenv("SEGM_APPLY_TS"); return (e && atoi(e) == 4) ? 4 : 8; }();
    return ts;
}

// SEGM_SCAN_FAST=0 forces the general kernels (A/B timing, and the tests that must exercise them on regular shapes)
static bool use_fast_path() {
    const char* e = getenv("SEGM_SCAN_FAST");
    return !(e && e[0] == '0');
}

template <typename T, int NS, int RW>
static int launch_fwd_rw(const ScanDev& P, hipStream_t stream) {
    constexpr int TS = 8;
    const Geom& gm = P.gm;
    const unsigned nblocks = (unsigned)((gm.nwaves + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL((scan_fwd_agg_kernel<T, NS, TS, RW>), dim3(nblocks), dim3(kBlock), 0, stream, P);
    launch_scan_carry(one_dir(P), 1, false, stream);
    if (apply_subtile() == 4)
        hipLaunchKernelGGL((scan_fwd_apply_kernel<T, NS, 4, RW>), dim3(nblocks), dim3(kBlock), 0, stream, P);
    else
        hipLaunchKernelGGL((scan_fwd_apply_kernel<T, NS, TS, RW>), dim3(nblocks), dim3(kBlock), 0, stream, P);
    return (int)hipGetLastError();
}

template <typename T, int NS>
static int launch_fwd(const ScanDev& P, hipStream_t stream) {
    if (P.gm.rw == 64) return launch_fwd_rw<T, NS, 64>(P, stream);
    if (P.gm.rw == 32) return launch_fwd_rw<T, NS, 32>(P, stream);
    return launch_fwd_rw<T, NS, 16>(P, stream);
}

template <typename T>
static int launch_fwd_ns(const ScanDev& P, hipStream_t stream) {
    if (P.gm.nstate <= 4) return launch_fwd<T, 4>(P, stream);
    if (P.gm.nstate <= 8) return launch_fwd<T, 8>(P, stream);
    return launch_fwd<T, 16>(P, stream);
}

// time strides are multiplied as uint32 (row_off)
static bool strides_ok(const segm_seq& s) { return !s.ptr || (s.stride_t >= 0 && s.stride_t < ((int64_t)1 << 31)); }
static bool bc_strides_ok(const segm_bc& m) { return !m.ptr || (m.stride_t >= 0 && m.stride_t < ((int64_t)1 << 31)); }

int validate_scan_common(const segm_scan_fwd_args* a) {
    if (!a) return SEGM_E_NULL;
    if (a->batch <= 0 || a->dim <= 0 || a->seqlen <= 0 || a->n_groups <= 0) return SEGM_E_SHAPE;
    if (a->seqlen >= ((int64_t)1 << 31)) return SEGM_E_SHAPE;          // 32-bit time indices
    if (a->dim % a->n_groups != 0) return SEGM_E_SHAPE;
    if (a->dstate < 1 || a->dstate > kMaxState) return SEGM_E_DSTATE;
    if (a->dtype != SEGM_F32 && a->dtype != SEGM_F16 && a->dtype != SEGM_BF16) return SEGM_E_DTYPE;
    if (a->time_order < SEGM_TIME_FORWARD || a->time_order > SEGM_TIME_INTERLEAVED) return SEGM_E_TIME_ORDER;
    if (a->time_order == SEGM_TIME_INTERLEAVED && (a->nslices <= 0 || a->nslices > 4096 || a->seqlen % a->nslices != 0)) return SEGM_E_SHAPE;
    if (a->chunk < 0 || (a->chunk % kChunkQuantum) != 0) return SEGM_E_SHAPE;
    if (!a->u.ptr || !a->delta.ptr || !a->B.ptr || !a->C.ptr || !a->A) return SEGM_E_NULL;
    if (!strides_ok(a->u) || !strides_ok(a->delta) || !strides_ok(a->z) || !strides_ok(a->out) ||
        !strides_ok(a->out_z) || !bc_strides_ok(a->B) || !bc_strides_ok(a->C))
        return SEGM_E_SHAPE;
    return SEGM_OK;
}

// `span_rows`: the rows whose byte offsets must fit 32 bits together with the channel / state offset - the whole sequence for
// the general kernels (per-batch base + 32-bit offset) and for an INTERLEAVED order (a chunk visits every slice), the rows of
// ONE wave (items per wave x chunk) for the regular-shape kernels in FORWARD / REVERSED order, whose buffer base is the wave's
// lowest row (scan_fast.h): a per-batch span beyond 4 GiB (2^24 rows of 96 fp32 channels) is then addressable.
int validate_spans(const segm_seq* const* seqs, int nseq, const segm_bc* const* bcs, int nbc, int dim, int dstate,
                   int64_t L, size_t esize, int64_t span_rows) {
    const int64_t es = (int64_t)esize, lim24 = (int64_t)1 << 24, lim32 = (int64_t)1 << 32;
    if (L > lim24) return SEGM_E_SHAPE;                    // time indices 0 .. L-1 must fit 24 bits
    if (span_rows <= 0 || span_rows > L) span_rows = L;
    for (int i = 0; i < nseq; ++i) {
        const segm_seq* s = seqs[i];
        if (!s || !s->ptr) continue;
        if (s->stride_t < 0 || s->stride_d < 0 || s->stride_t * es >= lim24) return SEGM_E_SHAPE;
        if (((span_rows - 1) * s->stride_t + (int64_t)(dim - 1) * s->stride_d + 1) * es >= lim32) return SEGM_E_SHAPE;
    }
    for (int i = 0; i < nbc; ++i) {
        const segm_bc* m = bcs[i];
        if (!m || !m->ptr) continue;
        if (m->stride_t < 0 || m->stride_n < 0 || m->stride_t * es >= lim24) return SEGM_E_SHAPE;
        if (((span_rows - 1) * m->stride_t + (int64_t)(dstate - 1) * m->stride_n + 1) * es >= lim32) return SEGM_E_SHAPE;
    }
    return SEGM_OK;
}
// rows one wave of the regular-shape kernels touches (0 = the whole sequence)
int64_t fast_span_rows(const ScanDev& P) { return P.tm.ns > 1 ? 0 : (int64_t)P.gm.g * P.gm.chunk; }

TimeMap make_timemap(int time_order, int nslices, int64_t L) {
    TimeMap tm;
    tm.L = (int32_t)L;
    if (time_order == SEGM_TIME_INTERLEAVED) { tm.ns = nslices; tm.sA = (int32_t)(L / nslices); tm.sW = -(int32_t)(L - 1); tm.base = 0; }
    else if (time_order == SEGM_TIME_REVERSED) { tm.ns = 1; tm.sA = 0; tm.sW = -1; tm.base = (int32_t)(L - 1); }
    else { tm.ns = 1; tm.sA = 0; tm.sW = 1; tm.base = 0; }
    const uint64_t magic = ((uint64_t)1 << 32) / (uint64_t)tm.ns + 1;
    tm.magic_lo = (uint32_t)magic;
    tm.magic_hi = (uint32_t)(magic >> 32);
    return tm;
}

// offsets a sequence view to channel d0
Seq seq_at(const segm_seq& s, int64_t d0, size_t esize) {
    Seq r = make_seq(s);
    if (r.p) r.p += d0 * s.stride_d * (int64_t)esize;
    return r;
}
BC bc_at(const segm_bc& m, int g, size_t esize) {
    BC r;
    r.p = (char*)m.ptr + (int64_t)g * m.stride_g * (int64_t)esize;
    r.sb = m.stride_b; r.st = m.stride_t; r.sn = m.stride_n;
    return r;
}
size_t dtype_size(int dtype) { return dtype == SEGM_F32 ? 4 : 2; }

// fills the parts of the device argument block shared by forward and backward for B/C group g
void fill_scan_dev(ScanDev& P, const segm_scan_fwd_args* a, int g, int chunk) {
    const int G = a->n_groups, Dg = a->dim / G, N = a->dstate;
    const size_t es = dtype_size(a->dtype);
    const int64_t d0 = (int64_t)g * Dg;
    memset(&P, 0, sizeof(P));
    P.gm = make_geom(a->batch, Dg, N, a->seqlen, chunk);
    P.tm = make_timemap(a->time_order, a->nslices, a->seqlen);
    P.u = seq_at(a->u, d0, es); P.delta = seq_at(a->delta, d0, es); P.z = seq_at(a->z, d0, es);
    P.out = seq_at(a->out, d0, es); P.out_z = seq_at(a->out_z, d0, es);
    P.Bm = bc_at(a->B, g, es); P.Cm = bc_at(a->C, g, es);
    P.A = a->A + d0 * N;
    P.D = a->D ? a->D + d0 : nullptr;
    P.delta_bias = a->delta_bias ? a->delta_bias + d0 : nullptr;
    P.delta_softplus = a->delta_softplus;
    P.nck = (int32_t)((a->seqlen + kCkpt - 1) / kCkpt);
    P.ckpt = a->ckpt ? a->ckpt + (size_t)g * a->batch * P.nck * (2 * ((N + 1) / 2)) * Dg : nullptr;
}

}  // namespace segm

using namespace segm;

extern "C" int32_t segm_selective_scan_default_chunk(int32_t batch, int32_t dim, int64_t seqlen) {
    if (batch <= 0 || dim <= 0 || seqlen <= 0) return kChunkQuantum;
    return default_chunk(batch, dim, seqlen);
}

extern "C" int32_t segm_selective_scan_regular_shape(int32_t batch, int32_t dim, int32_t dstate, int64_t seqlen, int32_t chunk,
                                                     int32_t time_order, int32_t nslices) {
    if (batch <= 0 || dim <= 0 || dstate <= 0 || seqlen <= 0 || seqlen >= ((int64_t)1 << 31)) return 0;
    if (time_order < SEGM_TIME_FORWARD || time_order > SEGM_TIME_INTERLEAVED) return 0;
    if (time_order == SEGM_TIME_INTERLEAVED && (nslices <= 0 || seqlen % nslices != 0)) return 0;
    ScanDev P;
    memset(&P, 0, sizeof(P));
    P.gm = make_geom(batch, dim, dstate, seqlen, chunk > 0 ? chunk : default_chunk(batch, dim, seqlen));
    P.tm = make_timemap(time_order, time_order == SEGM_TIME_INTERLEAVED ? nslices : 1, seqlen);
    return use_fast_path() && scan_fast_shape(P) ? 1 : 0;
}

extern "C" size_t segm_selective_scan_fwd_workspace_bytes(int32_t batch, int32_t dim, int32_t dstate, int64_t seqlen,
                                                          int32_t chunk) {
    if (batch <= 0 || dim <= 0 || dstate <= 0 || seqlen <= 0) return 0;
    if (chunk <= 0) chunk = default_chunk(batch, dim, seqlen);
    return fwd_ws_layout(batch, dim, dstate, seqlen, chunk).total;
}

extern "C" size_t segm_selective_scan_ckpt_bytes(int32_t batch, int32_t dim, int32_t dstate, int64_t seqlen) {
    if (batch <= 0 || dim <= 0 || dstate <= 0 || seqlen <= 0) return 0;
    const int64_t nck = (seqlen + kCkpt - 1) / kCkpt;
    return (size_t)batch * nck * (2 * ((dstate + 1) / 2)) * dim * sizeof(float);      // state pairs (an odd dstate pads one)
}

// one forward launch: validation, workspace slices, regular-shape or general kernels.  `PPout` (optional): instead of
// launching, hand back the argument block of a launch the regular-shape kernels take (n_groups == 1) for a batched launch.
static int scan_fwd_one(const segm_scan_fwd_args* a, ScanDev* batched) {
    int rc = validate_scan_common(a);
    if (rc != SEGM_OK) return rc;
    if (a->z.ptr && !a->out_z.ptr) return SEGM_E_NULL;
    if (!a->z.ptr && !a->out.ptr) return SEGM_E_NULL;
    const int chunk = a->chunk > 0 ? a->chunk : default_chunk(a->batch, a->dim, a->seqlen);
    const FwdWs ws = fwd_ws_layout(a->batch, a->dim, a->dstate, a->seqlen, chunk);
    if (!a->workspace || a->workspace_bytes < ws.total) return SEGM_E_WORKSPACE;

    const int G = a->n_groups, Dg = a->dim / G, N = a->dstate;
    const int64_t nch = (a->seqlen + chunk - 1) / chunk;
    hipStream_t stream = (hipStream_t)a->stream;
    char* wsb = (char*)a->workspace;
    const segm_seq* sv[5] = {&a->u, &a->delta, &a->z, &a->out, &a->out_z};
    const segm_bc* bv[2] = {&a->B, &a->C};

    for (int g = 0; g < G; ++g) {
        ScanDev P;
        fill_scan_dev(P, a, g, chunk);
        // per-group slices of the workspace: every region is [batch][rows][Dg]-shaped, group-major
        P.agg_sd = (float*)(wsb + ws.sd) + (size_t)g * a->batch * nch * Dg;
        P.agg_h = (float*)(wsb + ws.h) + (size_t)g * a->batch * nch * N * Dg;
        P.carry = (float*)(wsb + ws.carry) + (size_t)g * a->batch * nch * N * Dg;
        P.carry_seg = (float*)(wsb + ws.seg) + (size_t)g * a->batch * ((nch + kCarrySeg - 1) / kCarrySeg) * (N + 1) * Dg;
        P.last_state = a->last_state ? a->last_state + (int64_t)g * Dg * N : nullptr;
        P.last_state_sb = (int64_t)a->dim * N;
        const bool fast = use_fast_path() && scan_fast_shape(P);
        rc = validate_spans(sv, 5, bv, 2, a->dim, a->dstate, a->seqlen, dtype_size(a->dtype), fast ? fast_span_rows(P) : 0);
        if (rc != SEGM_OK) return rc;
        if (a->conv_width != 0) {                          // conv1d + SiLU inside the passes: regular-shape kernels, training / inference flags
            if (a->conv_width < 2 || a->conv_width > 4) return SEGM_E_WIDTH;
            if (!a->conv_weight) return SEGM_E_NULL;
            if (!fast || G != 1 || !a->delta_softplus || !a->z.ptr) return SEGM_E_SHAPE;
            P.conv_w = a->conv_weight; P.conv_b = a->conv_bias; P.conv_width = a->conv_width;
        }
        if (a->dt_rank != 0) {                             // dt_proj inside the passes: regular-shape kernels only
            if (a->dt_rank < 1 || a->dt_rank > 8) return SEGM_E_SHAPE;
            if (!a->dt_x || !a->dt_weight) return SEGM_E_NULL;
            if (!fast || G != 1 || a->conv_width != 0) return SEGM_E_SHAPE;
            if (a->dt_stride_t < a->dt_rank || a->dt_stride_t >= ((int64_t)1 << 31)) return SEGM_E_SHAPE;
            P.dt_x = BC{(char*)a->dt_x, a->dt_stride_b, a->dt_stride_t, 1};
            P.dt_w = a->dt_weight;
            P.dt_rank = a->dt_rank;
            const segm_bc dtb = {const_cast<void*>(a->dt_x), a->dt_stride_b, 0, a->dt_stride_t, 1};
            const segm_bc* dv[1] = {&dtb};
            rc = validate_spans(nullptr, 0, dv, 1, a->dim, a->dt_rank, a->seqlen, dtype_size(a->dtype), fast_span_rows(P));
            if (rc != SEGM_OK) return rc;
        }
        if (batched) {
            if (!fast || G != 1) return SEGM_E_SHAPE;       // the caller falls back to one launch per block
            *batched = P;
            return SEGM_OK;
        }
        if (fast) {                                        // regular shapes (every SegMamba stage): scan_fwd_fast.hip
            const ScanDevN PP = one_dir(P);
            launch_scan_fwd_fast(PP, 1, a->dtype, false, stream);
            launch_scan_carry(PP, 1, false, stream);
            launch_scan_fwd_fast(PP, 1, a->dtype, true, stream);
            rc = (int)hipGetLastError();
        } else if (a->dtype == SEGM_F32) rc = launch_fwd_ns<float>(P, stream);
        else if (a->dtype == SEGM_F16) rc = launch_fwd_ns<f16_t>(P, stream);
        else rc = launch_fwd_ns<bf16_t>(P, stream);
        if (rc != 0) return rc;
    }
    return SEGM_OK;
}

extern "C" int segm_selective_scan_fwd(const segm_scan_fwd_args* a) { return scan_fwd_one(a, nullptr); }

namespace segm {
// true when the launches share geometry, element type and stream, i.e. can be ONE grid with a direction axis
bool scan_same_launch(const segm_scan_fwd_args* a, const segm_scan_fwd_args* b) {
    return a->batch == b->batch && a->dim == b->dim && a->dstate == b->dstate && a->n_groups == 1 && b->n_groups == 1 &&
           a->seqlen == b->seqlen && a->dtype == b->dtype && a->stream == b->stream && a->conv_width == b->conv_width &&
           a->dt_rank == b->dt_rank &&
           (a->chunk > 0 ? a->chunk : default_chunk(a->batch, a->dim, a->seqlen)) ==
               (b->chunk > 0 ? b->chunk : default_chunk(b->batch, b->dim, b->seqlen));
}
}  // namespace segm

extern "C" int segm_selective_scan_fwd_multi(const segm_scan_fwd_args* args, int32_t n) {
    if (!args || n <= 0) return SEGM_E_NULL;
    bool batch = n > 1 && n <= kMaxDirs && use_fast_path();
    for (int i = 1; batch && i < n; ++i) batch = scan_same_launch(&args[0], &args[i]);
    ScanDevN PP;
    memset(&PP, 0, sizeof(PP));
    for (int i = 0; batch && i < n; ++i) {
        const int rc = scan_fwd_one(&args[i], &PP.d[i]);
        if (rc == SEGM_E_SHAPE) batch = false;              // not a regular shape: one launch per block below
        else if (rc != SEGM_OK) return rc;
    }
    if (batch) {
        hipStream_t stream = (hipStream_t)args[0].stream;
        launch_scan_fwd_fast(PP, n, args[0].dtype, false, stream);
        launch_scan_carry(PP, n, false, stream);
        launch_scan_fwd_fast(PP, n, args[0].dtype, true, stream);
        return (int)hipGetLastError();
    }
    for (int i = 0; i < n; ++i) {
        const int rc = scan_fwd_one(&args[i], nullptr);
        if (rc != SEGM_OK) return rc;
    }
    return SEGM_OK;
}
