// Shared pieces of the selective-scan forward / backward kernels.
#pragma once
#include "segm_device.h"

namespace segm {

constexpr int kCarrySegs = 16;        // waves per workgroup of the partial-sum reduction kernel (conv1d.hip)
constexpr int kCarrySeg = 64;         // chunks per segment of the carry kernels (held in registers by one wave)
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// Device-side argument block of the scan kernels: one B/C group, channel offset already applied.
struct ScanDev {
    Geom gm;
    TimeMap tm;
    Seq u, delta, z, out, out_z;          // z.p / out.p / out_z.p may be null
    BC Bm, Cm;
    const float* A;                        // (dim, nstate)
    const float* D;                        // (dim) or null
    const float* delta_bias;               // (dim) or null
    int32_t delta_softplus;
    // forward workspace (fp32), all laid out [batch][chunk][...][dim] so that lanes (channels) are contiguous
    float* agg_sd;                         // [batch][nchunks][dim]          sum of delta over the chunk
    float* agg_h;                          // [batch][nchunks][nstate][dim]  chunk end state from a zero start
    float* carry;                          // [batch][nchunks][nstate][dim]  state entering the chunk
    float* carry_seg;                      // [batch][nseg][nstate + 1][dim] scratch of the carry kernels (segment composites)
    float* ckpt;                           // [batch][nck][state pair][dim][2]  state entering step kCkpt*k (or null): a wave's
                                           // access to a pair is 64 lanes x 8 contiguous bytes, forward (store) and backward (load)
    int32_t nck;
    float* last_state;                     // (batch, dim, nstate) or null
    int64_t last_state_sb;                 // = full_dim * nstate
    // backward only
    Seq dout, du, ddelta, dz;
    float* dB;  int64_t dB_sb, dB_st, dB_sn;   // fp32 (one group)
    float* dC;  int64_t dC_sb, dC_st, dC_sn;
    float* part;                           // [batch][nchunks][nstate + 2][dim]  per-item dA / dD / ddelta_bias partials
    int32_t atomic_bc;                     // more than one d-tile contributes to dB / dC -> accumulate atomically
    int32_t dbc_native;                    // dB / dC have the tensors' element type (scan_bwd_w8.hip only), else fp32
    float* dbc_part;                       // [ndt][batch][L][dB 16 | dC 16]  per-d-tile sums (scan_bwd_w8.hip, ndt > 1)
    // forward, optional: u = SiLU(conv1d(x) + b) formed inside the passes (`u` then holds x); conv_width 0 = off
    const float* conv_w;                   // (dim, conv_width)
    const float* conv_b;                   // (dim) or null
    int32_t conv_width;
    // forward, optional: delta = dt_weight . dt_x formed inside the passes (`delta` is then written by the apply pass); dt_rank 0 = off
    BC dt_x;                               // rows of dt_rank consecutive elements (sn = 1)
    const float* dt_w;                     // (dim, dt_rank)
    int32_t dt_rank;
};

// up to three launches of ONE geometry (the three directions of a Mamba v3 layer: same shapes, different time order,
// tensors and parameters) travel in one kernel argument; blockIdx.y picks the block
constexpr int kMaxDirs = 3;
struct ScanDevN { ScanDev d[kMaxDirs]; };

// --- staging of the rows shared by all channels (B_t, C_t) ---------------------------------------------
// Each work item's own RW lanes fetch the TS x NS block of one matrix for the TS consecutive logical steps
// tau0 .. tau0+TS-1 (`it0` = the item's iterator positioned at tau0) into registers (`stage_fetch`) and
// later park them in LDS as fp32 (`stage_park`).  LDS layout per item: [s][n] (ROWMAJOR) for the forward
// (all states of one step are read together), [n][s] for the backward (all steps of one state are read
// together).  Rows outside [0, L) and states >= nstate are zero.  Loads are unconditional (the address is
// clamped to element 0 of the batch, the value masked) so no divergent branches are generated.
template <int TS, int NS, int RW> struct StageRegs { float v[TS * NS / RW]; };

template <typename T, int TS, int NS, int RW>
__device__ __forceinline__ void stage_fetch(StageRegs<TS, NS, RW>& rg, const BC& m, const TimeMap& tm, const TimeIter& it0,
                                            int b, int nstate, int r, bool item_ok) {
    const bool t_fastest = m.st <= m.sn;     // pick the lane -> element order that is contiguous in memory
    const char* base = m.p + (int64_t)b * m.sb * (int64_t)sizeof(T);          // wave-uniform
    const uint32_t stb = (uint32_t)m.st * (uint32_t)sizeof(T), snb = (uint32_t)m.sn * (uint32_t)sizeof(T);
#pragma unroll
    for (int i = 0; i < TS * NS / RW; ++i) {
        const int e = r + i * RW;
        int s, n;
        if (t_fastest) { n = e / TS; s = e - n * TS; } else { s = e / NS; n = e - s * NS; }
        const bool ok = item_ok && n < nstate && (it0.tau + s) < tm.L && (it0.tau + s) >= 0;
        const uint32_t off = ok ? __umul24((uint32_t)it0.ahead(tm, s), stb) + (uint32_t)n * snb : 0u;
        const float v = to_f32(*reinterpret_cast<const T*>(base + off));
        rg.v[i] = ok ? v : 0.f;
    }
}

template <int TS, int NS, int RW, bool ROWMAJOR>
__device__ __forceinline__ void stage_park(const StageRegs<TS, NS, RW>& rg, float* lds_item, bool t_fastest, int r) {
#pragma unroll
    for (int i = 0; i < TS * NS / RW; ++i) {
        const int e = r + i * RW;
        int s, n;
        if (t_fastest) { n = e / TS; s = e - n * TS; } else { s = e / NS; n = e - s * NS; }
        lds_item[ROWMAJOR ? (s * NS + n) : (n * TS + s)] = rg.v[i];
    }
}

// --- per-lane access to the rows of a sequence tensor ----------------------------------------------------
// address = wave-uniform batch base (SGPRs) + 32-bit byte offset  t * row_stride_bytes + lane_offset_bytes,
// i.e. one v_mad_u32_u24 per access and the saddr + voffset addressing mode.  The C ABI guarantees
// t < 2^24, row stride < 2^24 bytes and a per-batch span < 4 GiB (validate_spans()).
struct RowPtr {
    char* base;        // p + b * stride_b   (uniform over the wave)
    uint32_t loff;     // d * stride_d       in bytes (0 for lanes without a channel)
    uint32_t stb;      // stride_t           in bytes
};
template <typename T> __device__ __forceinline__ RowPtr make_rowptr(const Seq& s, int b_uniform, int d, bool lane_ok) {
    RowPtr r;
    r.base = s.p + (int64_t)b_uniform * s.sb * (int64_t)sizeof(T);
    r.loff = lane_ok ? (uint32_t)d * (uint32_t)(s.sd * (int64_t)sizeof(T)) : 0u;
    r.stb = (uint32_t)(s.st * (int64_t)sizeof(T));
    return r;
}
template <typename T> __device__ __forceinline__ float ld_row(const RowPtr& r, int32_t t) {
    return to_f32(*reinterpret_cast<const T*>(r.base + (uint32_t)(__umul24((uint32_t)t, r.stb) + r.loff)));
}
template <typename T> __device__ __forceinline__ void st_row(const RowPtr& r, int32_t t, float v) {
    *reinterpret_cast<T*>(r.base + (uint32_t)(__umul24((uint32_t)t, r.stb) + r.loff)) = from_f32<T>(v);
}

// Physical row index of the TS consecutive logical steps starting at `tj` (0 where the step is outside [0, L) or the
// lane owns no channel: such rows are loaded from row 0 and masked) and the mask of real rows.
template <int TS>
__device__ __forceinline__ uint32_t row_indices(int32_t (&tt)[TS], const TimeMap& tm, TimeIter tj, bool lane_ok) {
    uint32_t okm = 0;
#pragma unroll
    for (int j = 0; j < TS; ++j) {
        const bool ok = lane_ok && tj.tau < tm.L && tj.tau >= 0;
        tt[j] = ok ? tj.t : 0;
        okm |= ok ? (1u << j) : 0u;
        tj.next(tm);
    }
    return okm;
}

// Per-lane rows of a sequence tensor.  Unconditional loads, masked values.
template <typename T, int TS>
__device__ __forceinline__ void fetch_rows(float (&dst)[TS], const RowPtr& rp, const int32_t (&tt)[TS], uint32_t okm) {
#pragma unroll
    for (int j = 0; j < TS; ++j) {
        const float v = ld_row<T>(rp, tt[j]);
        dst[j] = ((okm >> j) & 1u) ? v : 0.f;
    }
}

// the batch index of a wave's work items as a scalar (all items of a wave share it by construction)
__device__ __forceinline__ int uniform_batch(const Item& it) { return __builtin_amdgcn_readfirstlane(it.b); }

// ---- host helpers shared by scan_fwd.hip / scan_bwd.hip -------------------------------------------------
Geom make_geom(int batch, int dim, int nstate, int64_t L, int chunk);
int32_t default_chunk(int32_t batch, int32_t dim, int64_t L);
int validate_scan_common(const segm_scan_fwd_args* a);
TimeMap make_timemap(int time_order, int nslices, int64_t L);
Seq seq_at(const segm_seq& s, int64_t d0, size_t esize);
BC bc_at(const segm_bc& m, int g, size_t esize);
size_t dtype_size(int dtype);
void fill_scan_dev(ScanDev& P, const segm_scan_fwd_args* a, int g, int chunk);
// the kernels' 32-bit offset arithmetic: L <= 2^24, |stride_t| * esize < 2^24, per-batch span of every view < 4 GiB
int validate_spans(const segm_seq* const* seqs, int nseq, const segm_bc* const* bcs, int nbc, int dim, int dstate,
                   int64_t L, size_t esize, int64_t span_rows);
int64_t fast_span_rows(const ScanDev& P);
bool scan_same_launch(const segm_scan_fwd_args* a, const segm_scan_fwd_args* b);
bool scan_fast_shape(const ScanDev& P);                        // scan_fwd_fast.hip: shapes its kernels take
void launch_scan_fwd_fast(const ScanDevN& PP, int ndir, int dtype, bool apply, hipStream_t stream);
bool scan_bwd_fast_shape(const ScanDev& P, size_t esize);      // scan_bwd_fast.hip
void launch_scan_bwd_fast(const ScanDevN& PP, int ndir, int dtype, bool main, hipStream_t stream);
bool scan_bwd_w8_shape(const ScanDev& P, size_t esize);        // scan_bwd_w8.hip: launches its main kernel takes
void launch_scan_bwd_main_w8(const ScanDevN& PP, int ndir, int dtype, hipStream_t stream);
size_t scan_bwd_w8_slab_bytes(int batch, int dim, int nstate, int64_t L);
// the whole sequence of every view of the launch fits a 32-bit byte offset from its batch base (what the pair kernel needs)
bool scan_full_span_fits(const ScanDev& P, size_t esize);
// carry composition of `ndir` argument blocks of one geometry; forward: agg_* / carry / carry_seg of each block, reverse: the
// backward's workspace (its agg_* / carry / carry_seg fields hold the reverse aggregates)
void launch_scan_carry(const ScanDevN& PP, int ndir, bool reverse, hipStream_t stream);
size_t scan_carry_scratch_bytes(int batch, int dim, int nstate, int64_t nchunks);

}  // namespace segm
