// Device-side building blocks shared by the scan / conv1d kernels (gfx950, wave64).
//
// Work decomposition used by every kernel in this library ("one lane per channel"):
//   * a LANE owns one channel d of one (batch, chunk) work item and walks `chunk` consecutive logical
//     time steps sequentially with its recurrence state in registers - no cross-lane scan;
//   * a WAVE (64 lanes) = G = 64/RW work items x RW adjacent channels (RW in {64, 32, 16}, the largest
//     that divides the channel count well), the G items being consecutive chunks of one (batch, d-tile);
//   * the reference parallelises only over (batch, channel) and walks L serially inside a block
//     (selective_scan_fwd_kernel.cuh:132,318); here L is split into chunks whose carries are
//     composed by a separate tiny kernel, so the grid is (B * D/RW * L/chunk) / G waves.
// With channel-last tensors (stride_d == 1) every per-step access of a wave is RW*esize contiguous
// bytes; quantities shared by all channels (B_t, C_t) are staged once per wave in LDS and read
// with wave-uniform (broadcast) addresses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/segmamba_hip.h"

namespace segm {

constexpr int kWave = 64;
constexpr int kBlock = 256;           // 4 waves per workgroup
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kMaxState = 16;
constexpr int kCkpt = 8;              // spacing (steps) of the forward state checkpoints kept for backward (round 4: 8 - the
                                      // backward's windows are 8 steps, scan_bwd_w8.hip; 2 x the checkpoint bytes of rounds 1 - 3)
constexpr int kChunkQuantum = 16;      // chunk lengths are multiples of this (the general backward kernel walks 16-step windows)
constexpr float kLog2e = 1.4426950408889634f;

// Pins a value to "computed here": an empty asm that reads and writes the register, so the compiler can neither
// sink the computation producing it below this point nor hoist later memory reads above it.  Used once per time
// step in the scan loops: without it the optimiser batches the LDS reads of a whole sub-tile ahead of the
// arithmetic and the kernels need > 200 VGPRs.  (The CPU emulation build defines it away.)
#ifndef SEGM_PIN_F32
#define SEGM_PIN_F32(x) asm volatile("" : "+v"(x) : : "memory")
#endif

// Nothing is scheduled across this point (machine scheduler fence); defined away in the CPU emulation build.
#ifndef SEGM_SCHED_FENCE
#define SEGM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// Orders a wave's LDS writes before its later LDS reads of data written by OTHER lanes of the same wave.  Every LDS
// region of the scan kernels is private to one wave, so no workgroup barrier is needed: the LDS unit executes one
// wave's operations in order; this only stops the compiler from reordering across it (and drains the counter).
// (The CPU emulation build maps it to its wave-level barrier.)
#ifndef SEGM_WAVE_LDS_SYNC
#define SEGM_WAVE_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif

typedef _Float16 f16_t;
typedef __bf16 bf16_t;

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }
// two fp32 -> one dword of two 16-bit elements (a in the low half): a single v_cvt_pk_* where the target has one
template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b) {
    typedef T t2 __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, t2));
}

// gfx950 LDS transpose read (ds_read_b64_tr_b16), measured semantics (tools/experiments/tr16_probe.hip, profiles/r05_tr16_probe.txt):
// every lane supplies an 8-byte aligned LDS address; with E[p] the four 16-bit values at lane p's address, lane i of a 16-lane group
// receives E[4 j + (i >> 2)][i & 3] for j = 0 .. 3.  tr16_fragment: two such reads = the eight k values of one MFMA operand fragment.
typedef short tr16_v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ tr16_v4 lds_read_tr16(const void* p) {
    typedef __attribute__((address_space(3))) tr16_v4 lds_v4;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p));
}
template <typename F>
__device__ __forceinline__ F tr16_fragment(const void* lo, const void* hi) {
    typedef short tr16_v8 __attribute__((ext_vector_type(8)));
    const tr16_v4 a = lds_read_tr16(lo), b = lds_read_tr16(hi);
    const tr16_v8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(F, v);
}

// up to four fixed-order partial reductions of one geometry in one launch (conv1d.hip reduce_partials_multi_kernel)
struct ReduceN {
    const float* part[4];
    float* out0[4];
    float* out1[4];
    float* out2[4];
};

// ---- XCD-aware work item order (conv3d_fwd.hip, conv3d_wgrad.hip) -----------------------------------------------------
// Workgroups are handed to the 8 XCDs round robin (workgroup i -> XCD i % 8), each XCD with its own L2.  Work items that are
// neighbours in z read the same input rows (the convolution kernels stage every row for the three planes around it), so
// consecutive items should meet in ONE L2: XCD k takes the k-th contiguous eighth of the item range.  Without this every row is fetched from HBM /
// Infinity Cache by three different L2s.
__device__ __forceinline__ int xcd_item(int bid, int nitems) {
#ifdef SEGM_NO_XCD_MAP
    return bid;
#else
    // any grid size (round 5: a 513-workgroup grid fell through to "no mapping" and lost every L2 hit - profiles/r05_wgrad_pmc.log):
    // XCD x runs workgroups x, x + 8, ... = per or per + 1 of them; it takes the range that starts at x * per + min(x, rem)
    const int per = nitems >> 3, rem = nitems & 7;
    if (per == 0) return bid;
    const int x = bid & 7, j = bid >> 3;
    return x * per + (x < rem ? x : rem) + j;
#endif
}

// ---- v_mfma_f32_16x16x32 for the two 16-bit element types (conv3d_fwd.hip, conv3d_wgrad.hip) -----------------------
typedef float mfma_f32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct Mfma16;
template <> struct Mfma16<bf16_t> {
    typedef bf16_t v8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ mfma_f32x4 run(v8 a, v8 b, mfma_f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma16<f16_t> {
    typedef f16_t v8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ mfma_f32x4 run(v8 a, v8 b, mfma_f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

// ---- 16-byte packets of a tensor's element type (HBM-bound elementwise kernels) ------------------------------
template <typename T> struct Vec;       // 16-byte packets
template <> struct Vec<float> { static constexpr int N = 4; };
template <> struct Vec<f16_t> { static constexpr int N = 8; };
template <> struct Vec<bf16_t> { static constexpr int N = 8; };

template <typename T, bool VEC> struct Pack {
    static constexpr int N = VEC ? Vec<T>::N : 1;
    float v[N];
    template <bool NT = false>
    __device__ __forceinline__ void load(const T* p) {
        if (VEC) {
            typedef uint32_t raw4 __attribute__((ext_vector_type(4)));
            // NT: streamed once, no reuse expected (the 128^3 volumes are larger than the 256 MB last-level cache)
            const raw4 raw = NT ? __builtin_nontemporal_load(reinterpret_cast<const raw4*>(p)) : *reinterpret_cast<const raw4*>(p);
            T tmp[N];
            memcpy(tmp, &raw, 16);
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = to_f32(tmp[i]);
        } else {
            v[0] = to_f32(p[0]);
        }
    }
    template <bool NT = false>
    __device__ __forceinline__ void store(T* p) const {
        if (VEC) {
            typedef uint32_t raw4 __attribute__((ext_vector_type(4)));
            T tmp[N];
#pragma unroll
            for (int i = 0; i < N; ++i) tmp[i] = from_f32<T>(v[i]);
            raw4 raw;
            memcpy(&raw, tmp, 16);
            if (NT) __builtin_nontemporal_store(raw, reinterpret_cast<raw4*>(p));
            else *reinterpret_cast<raw4*>(p) = raw;
        } else {
            p[0] = from_f32<T>(v[0]);
        }
    }
};

// ---- math (fast hardware forms; every use is a per-element or per-(element,state) hot op) -------
// raw v_exp_f32: results below 2^-126 flush to zero, which is benign for decay factors / sigmoid tails
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * kLog2e); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float sigmoidf(float x) { return fast_rcp(1.0f + fast_exp(-x)); }

// softplus with the reference's threshold (selective_scan_fwd_kernel.cuh:153-156: identity above 20)
__device__ __forceinline__ float softplus20(float x) {
    float e = fast_exp(x);
    float sp = (x < -15.0f) ? e : fast_log(1.0f + e);      // log1p(e) ~ e when e < 3e-7
    return x > 20.0f ? x : sp;
}

// ---- sequence-tensor view ---------------------------------------------------------------------
struct Seq {
    char* p;
    int64_t sb, st, sd;   // element strides
};
__host__ __device__ __forceinline__ Seq make_seq(const segm_seq& s) { return Seq{(char*)s.ptr, s.stride_b, s.stride_t, s.stride_d}; }

struct BC {
    char* p;
    int64_t sb, st, sn;   // one group only (group offset already applied)
};

template <typename T> __device__ __forceinline__ float ld(const Seq& s, int64_t off) {
    return to_f32(reinterpret_cast<const T*>(s.p)[off]);
}
template <typename T> __device__ __forceinline__ void st(const Seq& s, int64_t off, float v) {
    reinterpret_cast<T*>(s.p)[off] = from_f32<T>(v);
}

// ---- logical -> physical time ---------------------------------------------------------------------
// Time indices are 32-bit (the C ABI rejects L >= 2^31); element offsets are formed as
// base(int64) + uint32 t * uint32 stride, one v_mad_u64_u32.
//
// All three orders are one branch-free recurrence: logical tau = j*ns + k (0 <= k < ns) lives at
//      t = base + k*sA + j*(ns*sA + sW)
//   FORWARD      ns = 1, sA = 0,    sW = +1,     base = 0      -> t = tau
//   REVERSED     ns = 1, sA = 0,    sW = -1,     base = L-1    -> t = L-1-tau
//   INTERLEAVED  ns,     sA = L/ns, sW = -(L-1), base = 0      -> t = k*(L/ns) + j
// so a step is "t += sA; on k wrapping to 0: t += sW" with wave-uniform (SGPR) constants.
struct TimeMap {
    int32_t ns, sA, sW, base;
    int32_t L;
    uint32_t magic_lo;   // floor(2^32 / ns) + 1 (low 32 bits) : kk / ns for small kk without a divide
    uint32_t magic_hi;   // 1 when ns == 1
};

// floor(kk / ns) for 0 <= kk < 2^15
__device__ __forceinline__ int32_t small_div(const TimeMap& m, int32_t kk) {
    const uint64_t magic = ((uint64_t)m.magic_hi << 32) | m.magic_lo;
    return (int32_t)(((uint64_t)(uint32_t)kk * magic) >> 32);
}

// Walks logical time tau upward (next) or downward (prev) keeping the physical index t.
// tau may run past L (tail of the last chunk): t is then meaningless and callers mask with tau < L.
struct TimeIter {
    int32_t tau, t, k;
    __device__ __forceinline__ void seek(const TimeMap& m, int32_t tau0) {
        tau = tau0;
        const int32_t j = (int32_t)((uint32_t)tau0 / (uint32_t)m.ns);
        k = tau0 - j * m.ns;
        t = m.base + k * m.sA + j * (m.ns * m.sA + m.sW);
    }
    __device__ __forceinline__ void next(const TimeMap& m) {
        ++tau; ++k; t += m.sA;
        const bool w = k == m.ns;
        k = w ? 0 : k;
        t += w ? m.sW : 0;
    }
    __device__ __forceinline__ void prev(const TimeMap& m) {
        --tau; --k; t -= m.sA;
        const bool w = k < 0;
        k = w ? m.ns - 1 : k;
        t -= w ? m.sW : 0;
    }
    // move by n steps, |n| <= 64
    __device__ __forceinline__ void jump(const TimeMap& m, int n) {
        tau += n;
        const int32_t kk = k + n + 64 * m.ns;            // >= 0
        const int32_t q = small_div(m, kk);               // = floor((k+n)/ns) + 64
        k = kk - q * m.ns;
        t += n * m.sA + (q - 64) * m.sW;
    }
    // physical index of step tau + adv (0 <= adv <= 64) without moving
    __device__ __forceinline__ int32_t ahead(const TimeMap& m, int adv) const {
        const int32_t q = small_div(m, k + adv);
        return t + adv * m.sA + q * m.sW;
    }
};
// row offset in elements: uint32 * uint32 -> 64 bit
__device__ __forceinline__ int64_t row_off(int32_t t, int64_t stride) {
    return (int64_t)((uint64_t)(uint32_t)t * (uint64_t)(uint32_t)stride);
}

// ---- work-item geometry --------------------------------------------------------------------------
struct Geom {
    int32_t batch, dim, nstate;
    int32_t rw;        // channels per work item inside a wave: 64, 32 or 16
    int32_t g;         // work items per wave = 64 / rw
    int32_t ndt;       // d-tiles = ceil(dim / rw)
    int32_t chunk;     // logical steps per work item
    int32_t L;
    int32_t nchunks;   // ceil(L / chunk)
    int32_t ncg;       // chunk groups per (batch, d-tile) = ceil(nchunks / g)
    int64_t nwaves;    // batch * ndt * ncg
};

// What this lane works on.
struct Item {
    int32_t b, d, dt;      // batch, channel, d-tile
    int32_t gi, r;         // item index within the wave, channel index within the item
    int32_t chunk;         // chunk index
    bool wave_valid;       // the wave has at least something to do
    bool valid;            // this lane has a real (channel, chunk)
};

__device__ __forceinline__ Item locate(const Geom& gm, int64_t wave_id, int lane) {
    Item it;
    it.gi = lane / gm.rw;
    it.r = lane - it.gi * gm.rw;
    it.wave_valid = wave_id < gm.nwaves;
    uint32_t w = it.wave_valid ? (uint32_t)wave_id : 0u;   // the C ABI keeps nwaves < 2^31
    uint32_t cg = w % (uint32_t)gm.ncg;
    uint32_t rest = w / (uint32_t)gm.ncg;
    it.dt = (int32_t)(rest % (uint32_t)gm.ndt);
    it.b = (int32_t)(rest / (uint32_t)gm.ndt);
    it.d = it.dt * gm.rw + it.r;
    it.chunk = (int32_t)cg * gm.g + it.gi;
    it.valid = it.wave_valid && it.d < gm.dim && it.chunk < gm.nchunks;
    return it;
}

}  // namespace segm
