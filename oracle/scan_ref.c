/* CPU ORACLE (test infrastructure, never a product path) - plain C, fp64, OpenMP.
 *
 * Restatement of the reference's selective scan for sizes the PyTorch oracle (oracle/ref_ops.py) cannot walk:
 *   forward   mamba/mamba_ssm/ops/selective_scan_interface.py:86-152   (selective_scan_ref)
 *   backward  the closed form of SURVEY.md Appendix A, i.e. what mamba/csrc/selective_scan/selective_scan_bwd_kernel.cuh:161-478
 *             computes (autograd through selective_scan_ref gives the same numbers; pinned by tests/test_oracle_golden.py
 *             against the fixtures the reference's own functions produced, tests/golden/scan_L*_G*.npz)
 * and of the depthwise causal conv1d (+SiLU)   causal-conv1d/causal_conv1d/causal_conv1d_interface.py:49-65.
 * real A, input-dependent B / C with G groups, optional D / z / delta_bias, softplus with the reference's threshold 20.
 *
 * Layout = the reference's: u, delta, z, dout (B, D, L) fp32 contiguous; A (D, N); Bm, Cm (B, G, N, L); outputs fp64.
 * All arithmetic in double.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object (oracle/_build/libscan_ref.so, built by oracle/build_oracle.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BLK 2048            /* recompute block of the backward walk (state history kept per block only) */

static inline double softplus20(double x) { return x > 20.0 ? x : log1p(exp(x)); }
static inline double sigmoid(double x) { return 1.0 / (1.0 + exp(-x)); }

int segm_oracle_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* out (B, D, L): y * silu(z) when z != NULL else y;  y_noz (optional): y;  last_state (optional) (B, D, N) */
int segm_oracle_scan_fwd(int B, int D, int N, int G, int64_t L, const float* u, const float* delta, const float* A,
                         const float* Bm, const float* Cm, const float* Dv, const float* z, const float* dbias,
                         int softplus, double* out, double* y_noz, double* last_state) {
    if (N > 256 || D % G != 0) return -1;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int b = 0; b < B; ++b)
        for (int d = 0; d < D; ++d) {
            const int g = d / (D / G);
            const float* ur = u + ((int64_t)b * D + d) * L;
            const float* dr = delta + ((int64_t)b * D + d) * L;
            const float* zr = z ? z + ((int64_t)b * D + d) * L : NULL;
            const float* Bg = Bm + ((int64_t)b * G + g) * N * L;
            const float* Cg = Cm + ((int64_t)b * G + g) * N * L;
            double h[256];
            for (int n = 0; n < N; ++n) h[n] = 0.0;
            for (int64_t t = 0; t < L; ++t) {
                double dl = (double)dr[t] + (dbias ? (double)dbias[d] : 0.0);
                if (softplus) dl = softplus20(dl);
                const double uu = ur[t], dlu = dl * uu;
                double y = 0.0;
                for (int n = 0; n < N; ++n) {
                    h[n] = exp(dl * (double)A[d * N + n]) * h[n] + dlu * (double)Bg[(int64_t)n * L + t];
                    y += h[n] * (double)Cg[(int64_t)n * L + t];
                }
                if (Dv) y += (double)Dv[d] * uu;
                const int64_t o = ((int64_t)b * D + d) * L + t;
                if (y_noz) y_noz[o] = y;
                out[o] = zr ? y * (double)zr[t] * sigmoid((double)zr[t]) : y;
            }
            if (last_state)
                for (int n = 0; n < N; ++n) last_state[((int64_t)b * D + d) * N + n] = h[n];
        }
    return 0;
}

/* One (b, d) row, states [n0, n1): forward states with block checkpoints, then the reverse walk.
 * row outputs (may be NULL): du, ddelta, dz (L);  scalars: dA[n] (N), dD, ddb;  dBrow / dCrow: (N, L) accumulated (+=). */
static void bwd_row(int N, int n0, int n1, int64_t L, const float* ur, const float* dr, const float* zr, const float* gr,
                    const float* Arow, const float* Bg, const float* Cg, double Dd, int hasD, double bias, int softplus,
                    double* du, double* ddelta, double* dz, double* dA, double* dD, double* ddb, double* dBrow, double* dCrow) {
    const int ns = n1 - n0;
    const int64_t nblk = (L + BLK - 1) / BLK;
    double* ck = (double*)malloc(sizeof(double) * (size_t)nblk * ns);          /* state entering each block */
    double* hh = (double*)malloc(sizeof(double) * (size_t)BLK * ns);           /* h_t inside the block       */
    double* dls = (double*)malloc(sizeof(double) * BLK);
    double h[256], e[256];
    for (int k = 0; k < ns; ++k) { h[k] = 0.0; e[k] = 0.0; }
    for (int64_t t = 0; t < L; ++t) {
        if (t % BLK == 0) memcpy(ck + (t / BLK) * ns, h, sizeof(double) * ns);
        double dl = (double)dr[t] + bias;
        if (softplus) dl = softplus20(dl);
        const double dlu = dl * (double)ur[t];
        for (int k = 0; k < ns; ++k)
            h[k] = exp(dl * (double)Arow[n0 + k]) * h[k] + dlu * (double)Bg[(int64_t)(n0 + k) * L + t];
    }
    double dDa = 0.0, dba = 0.0;
    for (int64_t blk = nblk - 1; blk >= 0; --blk) {
        const int64_t t0 = blk * BLK, t1 = (t0 + BLK < L) ? t0 + BLK : L;
        memcpy(h, ck + blk * ns, sizeof(double) * ns);
        for (int64_t t = t0; t < t1; ++t) {
            double dl = (double)dr[t] + bias;
            if (softplus) dl = softplus20(dl);
            dls[t - t0] = dl;
            const double dlu = dl * (double)ur[t];
            for (int k = 0; k < ns; ++k) {
                h[k] = exp(dl * (double)Arow[n0 + k]) * h[k] + dlu * (double)Bg[(int64_t)(n0 + k) * L + t];
                hh[(t - t0) * ns + k] = h[k];
            }
        }
        for (int64_t t = t1 - 1; t >= t0; --t) {
            const double dl = dls[t - t0], uu = ur[t];
            double g = gr[t], y = 0.0;
            /* y (needed for dz) from the states of THIS call only: callers that want dz pass all states */
            if (zr) {
                const double zz = zr[t], sg = sigmoid(zz);
                if (dz) {
                    for (int k = 0; k < ns; ++k) y += hh[(t - t0) * ns + k] * (double)Cg[(int64_t)(n0 + k) * L + t];
                    if (hasD) y += Dd * uu;
                    dz[t] = g * y * sg * (1.0 + zz * (1.0 - sg));
                }
                g *= zz * sg;
            }
            double q = 0.0, ddl = 0.0;
            for (int k = 0; k < ns; ++k) {
                const int n = n0 + k;
                const double An = Arow[n], a = exp(dl * An);
                const double hprev = (t > t0) ? hh[(t - 1 - t0) * ns + k] : ck[blk * ns + k];
                const double Bv = Bg[(int64_t)n * L + t], Cv = Cg[(int64_t)n * L + t];
                const double dh = Cv * g + e[k];                     /* e = a_{t+1} dh_{t+1} */
                const double t2 = dh * hprev * a;                     /* dh * (h_t - b_t)     */
                if (dA) dA[n] += t2 * dl;
                q += dh * Bv;
                ddl += t2 * An;
                if (dBrow) dBrow[(int64_t)n * L + t] += dh * dl * uu;
                if (dCrow) dCrow[(int64_t)n * L + t] += g * hh[(t - t0) * ns + k];
                e[k] = a * dh;
            }
            if (du) du[t] = dl * q + (hasD ? Dd * g : 0.0);
            if (ddelta) {
                double dd = ddl + uu * q;
                if (softplus) { const double raw = (double)dr[t] + bias; dd *= raw > 20.0 ? 1.0 : sigmoid(raw); }
                ddelta[t] = dd;
                dba += dd;
            }
            dDa += g * uu;
        }
    }
    if (dD) *dD += dDa;
    if (ddb) *ddb += dba;
    free(ck); free(hh); free(dls);
}

/* Gradients of sum(out * dout) with out as in segm_oracle_scan_fwd.  du, ddelta, dz (B, D, L); dA (D, N); dB, dC (B, G, N, L);
 * dD, ddbias (D).  Any output pointer may be NULL.  dB / dC: pass 2 below, race-free by ownership of (b, n). */
int segm_oracle_scan_bwd(int B, int D, int N, int G, int64_t L, const float* u, const float* delta, const float* A,
                         const float* Bm, const float* Cm, const float* Dv, const float* z, const float* dbias,
                         int softplus, const float* dout, double* du, double* ddelta, double* dA, double* dB, double* dC,
                         double* dD, double* dz, double* ddbias) {
    if (N > 256 || D % G != 0) return -1;
    double* pA = (double*)calloc((size_t)B * D * N, sizeof(double));
    double* pD = (double*)calloc((size_t)B * D, sizeof(double));
    double* pb = (double*)calloc((size_t)B * D, sizeof(double));
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int b = 0; b < B; ++b)
        for (int d = 0; d < D; ++d) {
            const int g = d / (D / G);
            const int64_t r = ((int64_t)b * D + d) * L;
            bwd_row(N, 0, N, L, u + r, delta + r, z ? z + r : NULL, dout + r, A + (int64_t)d * N,
                    Bm + ((int64_t)b * G + g) * N * L, Cm + ((int64_t)b * G + g) * N * L, Dv ? (double)Dv[d] : 0.0, Dv != NULL,
                    dbias ? (double)dbias[d] : 0.0, softplus, du ? du + r : NULL, ddelta ? ddelta + r : NULL,
                    dz ? dz + r : NULL, pA + ((int64_t)b * D + d) * N, pD + (int64_t)b * D + d, pb + (int64_t)b * D + d, NULL,
                    NULL);
        }
    for (int d = 0; d < D; ++d) {
        double sD = 0.0, sb = 0.0;
        for (int b = 0; b < B; ++b) { sD += pD[(int64_t)b * D + d]; sb += pb[(int64_t)b * D + d]; }
        if (dD) dD[d] = sD;
        if (ddbias) ddbias[d] = sb;
        if (dA)
            for (int n = 0; n < N; ++n) {
                double s = 0.0;
                for (int b = 0; b < B; ++b) s += pA[((int64_t)b * D + d) * N + n];
                dA[(int64_t)d * N + n] = s;
            }
    }
    free(pA); free(pD); free(pb);
    if (dB || dC) {
        /* work item (b, n, c): one state over the channels of chunk c of a group, into a private (L) row pair; chunks are
         * summed afterwards in fixed order.  NC chunks per group = the largest of 8, 4, 2, 1 dividing D / G. */
        const int dg = D / G;
        const int NC = dg % 8 == 0 ? 8 : dg % 4 == 0 ? 4 : dg % 2 == 0 ? 2 : 1;
        const int64_t items = (int64_t)B * G * N * NC;
        double* pB = dB ? (double*)calloc((size_t)items * L, sizeof(double)) : NULL;
        double* pC = dC ? (double*)calloc((size_t)items * L, sizeof(double)) : NULL;
        if ((dB && !pB) || (dC && !pC)) { free(pB); free(pC); return -2; }
#pragma omp parallel for schedule(dynamic, 1)
        for (int64_t it = 0; it < items; ++it) {
            const int c = (int)(it % NC), n = (int)((it / NC) % N), g = (int)((it / NC / N) % G), b = (int)(it / NC / N / G);
            const int64_t o = ((int64_t)b * G + g) * N * L;
            /* bwd_row indexes its dB / dC rows as [n * L + t]: shift the private row so that state n lands at offset 0 */
            double* rb = pB ? pB + it * L - (int64_t)n * L : NULL;
            double* rc = pC ? pC + it * L - (int64_t)n * L : NULL;
            for (int d = g * dg + c * (dg / NC); d < g * dg + (c + 1) * (dg / NC); ++d) {
                const int64_t r = ((int64_t)b * D + d) * L;
                bwd_row(N, n, n + 1, L, u + r, delta + r, z ? z + r : NULL, dout + r, A + (int64_t)d * N, Bm + o, Cm + o, 0.0, 0,
                        dbias ? (double)dbias[d] : 0.0, softplus, NULL, NULL, NULL, NULL, NULL, NULL, rb, rc);
            }
        }
#pragma omp parallel for schedule(static)
        for (int64_t row = 0; row < (int64_t)B * G * N; ++row)
            for (int64_t t = 0; t < L; ++t) {
                double sb = 0.0, sc = 0.0;
                for (int c = 0; c < NC; ++c) {
                    if (pB) sb += pB[(row * NC + c) * L + t];
                    if (pC) sc += pC[(row * NC + c) * L + t];
                }
                if (dB) dB[row * L + t] = sb;
                if (dC) dC[row * L + t] = sc;
            }
        free(pB); free(pC);
    }
    return 0;
}

/* out[b, d, t] = act(bias[d] + sum_w W[d, w] x[b, d, t - (width-1-w)])   (zero left pad) */
int segm_oracle_conv1d_fwd(int B, int D, int W, int64_t L, const float* x, const float* weight, const float* bias, int silu,
                           double* out) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int d = 0; d < D; ++d) {
            const float* xr = x + ((int64_t)b * D + d) * L;
            double* orow = out + ((int64_t)b * D + d) * L;
            for (int64_t t = 0; t < L; ++t) {
                double acc = bias ? (double)bias[d] : 0.0;
                for (int w = 0; w < W; ++w) {
                    const int64_t s = t - (W - 1 - w);
                    if (s >= 0) acc += (double)weight[d * W + w] * (double)xr[s];
                }
                orow[t] = silu ? acc * sigmoid(acc) : acc;
            }
        }
    return 0;
}
