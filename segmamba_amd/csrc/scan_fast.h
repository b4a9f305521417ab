// Pieces shared by the "regular shape" scan kernels (scan_fwd_fast.hip, scan_bwd_fast.hip): wave-uniform time
// arithmetic, uniform-base + constant-lane-offset row streams, B / C staging, packed fp32 pairs.
#pragma once
#include <limits.h>

#include "scan_common.h"

namespace segm {

typedef float f2 __attribute__((ext_vector_type(2)));

#ifndef SEGM_PIN_F2
#define SEGM_PIN_F2(x) asm volatile("" : "+v"(x) : : "memory")
#endif

#ifndef SEGM_FAST_MIN_WAVES
#define SEGM_FAST_MIN_WAVES 3       // waves per SIMD the apply kernel is register-limited to (3072 waves at stage 0 = 3 per SIMD)
#endif

// ---- per-wave timeline (experiments only: -DSEGM_SCAN_TIMELINE, tools/gpu_scan_timeline.py) ------------------------------------
// Four s_memtime stamps per wave (entry, loop head, end of the first sub-tile, loop exit) plus the constant 100 MHz clock at entry and
// exit (comparable across CUs / XCDs) and the hardware id (SIMD / CU / SE / XCC), written to a side buffer
// [kernel slot][wave][8 x u64] that segm_debug_set_timeline() installs.  The product build compiles none of this.
#ifdef SEGM_SCAN_TIMELINE
extern __device__ unsigned long long* g_segm_timeline;
extern __device__ unsigned int g_segm_timeline_waves;      // waves per kernel slot
struct WaveTimeline {
    unsigned long long t[4], rt0;
    __device__ __forceinline__ void stamp(int i) {
        unsigned long long v;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : : "memory");
        t[i] = v;
    }
    __device__ __forceinline__ void begin() {
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0) : : "memory");
        stamp(0);
    }
    __device__ __forceinline__ void end(int slot, unsigned wave_id, int lane) {
        unsigned long long t4, rt1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t4) : : "memory");
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1) : : "memory");
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        if (lane == 0 && g_segm_timeline && wave_id < g_segm_timeline_waves) {
            unsigned long long* o = g_segm_timeline + ((size_t)slot * g_segm_timeline_waves + wave_id) * 8;
            o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3]; o[4] = t4; o[5] = rt0; o[6] = rt1;
            o[7] = ((unsigned long long)xcc << 32) | hw;
        }
    }
};
#define SEGM_TL_DECL() WaveTimeline tl_; tl_.begin()
#define SEGM_TL_STAMP(i) tl_.stamp(i)
#define SEGM_TL_END(slot, wid, lane) tl_.end(slot, wid, lane)
#else
#define SEGM_TL_DECL() ((void)0)
#define SEGM_TL_STAMP(i) ((void)0)
#define SEGM_TL_END(slot, wid, lane) ((void)0)
#endif

constexpr int kFS = 16;     // states
constexpr int kFT = 8;      // steps per sub-tile

// Wave-uniform description of where sub-tile s of every work item lives: physical row = T_item + U(s) + i * dT.
struct FastClock {
    int32_t dT;             // rows between consecutive logical steps inside a sub-tile
    int32_t ns;             // INTERLEAVED: slices (a sub-tile never wraps); otherwise INT_MAX
    int32_t sA;
    int32_t kk, jj, U;
    __device__ __forceinline__ void init(const TimeMap& tm) {
        const bool inter = tm.ns > 1;
        dT = inter ? tm.sA : tm.sA + tm.sW;                // FORWARD +1, REVERSED -1, INTERLEAVED L / ns
        ns = inter ? tm.ns : INT_MAX;
        sA = tm.sA;
        kk = 0; jj = 0; U = 0;
    }
    __device__ __forceinline__ int32_t next_U() const {     // U of the following sub-tile
        return (kk + kFT == ns) ? jj + 1 : U + kFT * dT;
    }
    __device__ __forceinline__ void advance() {
        U = next_U();
        kk += kFT;
        if (kk == ns) { kk = 0; ++jj; }
    }
};
// physical row of the first step of a work item (chunk start tau0; tau0 % ns == 0 for INTERLEAVED)
__device__ __forceinline__ int32_t fast_item_row(const TimeMap& tm, int32_t tau0) {
    if (tm.ns > 1) return (int32_t)((uint32_t)tau0 / (uint32_t)tm.ns);
    return tm.base + tau0 * (tm.sA + tm.sW);
}

// A per-lane stream of one sequence tensor: uniform base + constant lane offset.
struct FastRow {
    const char* base;       // p + b * stride_b                    (uniform)
    int64_t stb;            // stride_t in bytes                   (uniform)
    uint32_t loff;          // T_item * stride_t + d * stride_d    (bytes, per lane)
};
template <typename T> __device__ __forceinline__ FastRow fast_row(const Seq& s, int b_uniform, int32_t t_item, int d) {
    FastRow r;
    r.base = s.p + (int64_t)b_uniform * s.sb * (int64_t)sizeof(T);
    r.stb = s.st * (int64_t)sizeof(T);
    r.loff = (uint32_t)t_item * (uint32_t)r.stb + (uint32_t)d * (uint32_t)(s.sd * (int64_t)sizeof(T));
    return r;
}
// the kFT rows of the sub-tile at uniform row offset U
template <typename T>
__device__ __forceinline__ void fast_fetch(float (&dst)[kFT], const FastRow& r, int32_t U, int32_t dT) {
    const char* p = r.base + (int64_t)U * r.stb;
    const int64_t inc = (int64_t)dT * r.stb;
#pragma unroll
    for (int j = 0; j < kFT; ++j) dst[j] = to_f32(*reinterpret_cast<const T*>(p + (int64_t)j * inc + r.loff));
}

// B / C staging: lane r of a work item fetches elements e = r + i * RW of the kFT x 16 block of its sub-tile.
template <int RW> struct FastStage {
    static constexpr int EPL = kFT * kFS / RW;   // elements per lane: 2, 4 or 8
    const char* base;       // uniform
    int64_t stb;            // row stride in bytes (uniform)
    int64_t inc;            // byte distance between a lane's consecutive elements (uniform)
    uint32_t loff;          // per lane
    int32_t lds0, ldsinc;   // LDS index ([s][n] layout) of element 0 and the step between elements
};
template <typename T, int RW>
__device__ __forceinline__ FastStage<RW> fast_stage(const BC& m, int b_uniform, int32_t t_item, int32_t dT, int r) {
    FastStage<RW> st;
    st.base = m.p + (int64_t)b_uniform * m.sb * (int64_t)sizeof(T);
    st.stb = m.st * (int64_t)sizeof(T);
    const int64_t snb = m.sn * (int64_t)sizeof(T);
    if (m.st <= m.sn) {                                    // time fastest in memory: e -> (n = e / kFT, s = e % kFT)
        const int s = r % kFT, n = r / kFT;
        st.loff = (uint32_t)(t_item + s * dT) * (uint32_t)st.stb + (uint32_t)n * (uint32_t)snb;
        st.inc = (int64_t)(RW / kFT) * snb;
        st.lds0 = s * kFS + n;
        st.ldsinc = RW / kFT;
    } else {                                               // state fastest: e -> (s = e / 16, n = e % 16)
        const int s = r / kFS, n = r % kFS;
        st.loff = (uint32_t)(t_item + s * dT) * (uint32_t)st.stb + (uint32_t)n * (uint32_t)snb;
        st.inc = (int64_t)((RW >= kFS ? RW / kFS : 1) * dT) * st.stb;
        st.lds0 = s * kFS + n;
        st.ldsinc = (RW >= kFS ? RW / kFS : 1) * kFS;
    }
    return st;
}
template <typename T, int RW>
__device__ __forceinline__ void fast_stage_fetch(float (&v)[FastStage<RW>::EPL], const FastStage<RW>& st, int32_t U) {
    const char* p = st.base + (int64_t)U * st.stb;
#pragma unroll
    for (int i = 0; i < FastStage<RW>::EPL; ++i) v[i] = to_f32(*reinterpret_cast<const T*>(p + (int64_t)i * st.inc + st.loff));
}
template <int RW>
__device__ __forceinline__ void fast_stage_park(const float (&v)[FastStage<RW>::EPL], const FastStage<RW>& st, float* lds_item) {
#pragma unroll
    for (int i = 0; i < FastStage<RW>::EPL; ++i) lds_item[st.lds0 + i * st.ldsinc] = v[i];
}

// ------------------------------------------------------------------------------------------------------
// Buffer-addressed row streams (round 3).  The round-2 kernels formed a 64-bit per-lane address for every 2-byte access
// (v_lshl_add_u64 / v_mad_u64_u32: ~45 % of the VALU instructions around the recurrence were address arithmetic, in a
// kernel that is VALU-issue bound).  Here a tensor is a buffer resource whose base is a wave-uniform 64-bit pointer, a lane
// adds ONE constant 32-bit byte offset and the row of a step is a scalar byte offset:
//       address = base(batch, lowest row of the wave)  +  voff(lane)  +  soff(sub-tile, step)
// - no vector instruction takes part in addressing, and a per-batch span beyond 4 GiB (L = 2^24 rows of 96 fp32 channels)
// is addressable because only the rows ONE wave touches must fit 32 bits (FORWARD / REVERSED; an INTERLEAVED chunk spans the
// whole sequence).  All parts are non-negative: for a descending order (dT < 0) the uniform part counts down from
// `bias = chunk - 1` and the lane's anchor is the LOWEST row of its item.
// ------------------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0xffffffffu, 0x00020000);   // raw, no range check
}
// `ld` converts where it loads - fine in straight-line code, where the scheduler moves the conversion to the use.  A load whose
// value is consumed in a LATER basic block (a prefetch inside a conditional) must use `ld_raw` + `cvt_raw`: the conversion is a
// use, and a use in the loading block makes the compiler wait for the load there (s_waitcnt vmcnt right behind the loads).
template <typename T> struct BufIO;
template <> struct BufIO<float> {
    static __device__ __forceinline__ uint32_t ld_raw(rsrc_t r, uint32_t voff, uint32_t soff) {
        return __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
    }
    static __device__ __forceinline__ float cvt_raw(uint32_t v) { return __uint_as_float(v); }
    static __device__ __forceinline__ float ld(rsrc_t r, uint32_t voff, uint32_t soff) {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    }
    template <int AUX = 0>
    static __device__ __forceinline__ void st(rsrc_t r, uint32_t voff, uint32_t soff, float v) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, AUX);
    }
};
template <> struct BufIO<bf16_t> {
    static __device__ __forceinline__ uint32_t ld_raw(rsrc_t r, uint32_t voff, uint32_t soff) {
        return (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0);
    }
    static __device__ __forceinline__ float cvt_raw(uint32_t v) { return __uint_as_float(v << 16); }
    static __device__ __forceinline__ float ld(rsrc_t r, uint32_t voff, uint32_t soff) {
        return __uint_as_float((uint32_t)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0) << 16);
    }
    template <int AUX = 0>
    static __device__ __forceinline__ void st(rsrc_t r, uint32_t voff, uint32_t soff, float v) {
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, from_f32<bf16_t>(v)), r, voff, soff, AUX);
    }
};
template <> struct BufIO<f16_t> {
    static __device__ __forceinline__ uint32_t ld_raw(rsrc_t r, uint32_t voff, uint32_t soff) {
        return (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0);
    }
    static __device__ __forceinline__ float cvt_raw(uint32_t v) { return to_f32(__builtin_bit_cast(f16_t, (unsigned short)v)); }
    static __device__ __forceinline__ float ld(rsrc_t r, uint32_t voff, uint32_t soff) {
        return to_f32(__builtin_bit_cast(f16_t, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0)));
    }
    template <int AUX = 0>
    static __device__ __forceinline__ void st(rsrc_t r, uint32_t voff, uint32_t soff, float v) {
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, from_f32<f16_t>(v)), r, voff, soff, AUX);
    }
};

// what is uniform over a wave for all of its row streams, and the lane's anchor row
struct WaveRows {
    int32_t dT;            // rows between consecutive logical steps inside a sub-tile (+1, -1, or L / nslices)
    int32_t bias;          // dT < 0 ? chunk - 1 : 0    (added to every uniform row offset)
    int32_t row_lo;        // lowest physical row a lane of this wave touches          (uniform)
    int32_t lane_row;      // lowest row of this lane's item - row_lo                   (>= 0)
};
__device__ __forceinline__ WaveRows wave_rows(const TimeMap& tm, const Geom& gm, const Item& it) {
    WaveRows w;
    w.dT = tm.ns > 1 ? tm.sA : tm.sA + tm.sW;
    w.bias = w.dT < 0 ? gm.chunk - 1 : 0;
    const int32_t chunk0 = __builtin_amdgcn_readfirstlane(it.chunk - it.gi);      // the wave's items are chunks chunk0 .. chunk0 + g - 1
    const int32_t a_first = fast_item_row(tm, chunk0 * gm.chunk) - w.bias;
    const int32_t a_last = fast_item_row(tm, (chunk0 + gm.g - 1) * gm.chunk) - w.bias;
    w.row_lo = a_first < a_last ? a_first : a_last;
    w.lane_row = fast_item_row(tm, it.chunk * gm.chunk) - w.bias - w.row_lo;
    return w;
}
// one sequence tensor as a wave sees it
struct Stream {
    rsrc_t rs;
    uint32_t voff;         // lane_row * stride_t + d * stride_d, bytes
    int32_t stb;           // stride_t in bytes (uniform)
};
template <typename T> __device__ __forceinline__ Stream make_stream(const Seq& s, int b_uniform, const WaveRows& w, int d) {
    Stream r;
    r.stb = (int32_t)(s.st * (int64_t)sizeof(T));
    r.rs = make_rsrc(s.p + ((int64_t)b_uniform * s.sb + (int64_t)w.row_lo * s.st) * (int64_t)sizeof(T));
    r.voff = (uint32_t)w.lane_row * (uint32_t)r.stb + (uint32_t)d * (uint32_t)(s.sd * (int64_t)sizeof(T));
    return r;
}
// the kFT rows of the sub-tile whose first step sits `rows` rows (uniform, >= 0 together with j * dT) above the anchor
template <typename T>
__device__ __forceinline__ void stream_fetch(float (&dst)[kFT], const Stream& st, int32_t rows, int32_t dT) {
    // unsigned arithmetic: every partial term may wrap, the sum is the true offset in [0, 2^32)
    uint32_t so = (uint32_t)rows * (uint32_t)st.stb;      // running scalar offset: one s_add per row (not a multiply per row)
    const uint32_t inc = (uint32_t)(dT * st.stb);
#pragma unroll
    for (int j = 0; j < kFT; ++j) {
        dst[j] = BufIO<T>::ld(st.rs, st.voff, so);
        so += inc;
    }
}

// B / C staging through buffer loads: lane r of a work item fetches EPL elements of the kFT x 16 block of a sub-tile;
// element i sits  i * jrow  steps and  i * ninc  bytes after the lane's first one
template <int RW> struct StageStream {
    static constexpr int EPL = kFT * kFS / RW;
    rsrc_t rs;
    uint32_t voff;
    int32_t stb, jrow, ninc, jmax;
    int32_t lds0, ldsinc;
};
template <typename T, int RW>
__device__ __forceinline__ StageStream<RW> make_stage(const BC& m, int b_uniform, const WaveRows& w, int r) {
    StageStream<RW> st;
    st.stb = (int32_t)(m.st * (int64_t)sizeof(T));
    const int32_t snb = (int32_t)(m.sn * (int64_t)sizeof(T));
    st.rs = make_rsrc(m.p + ((int64_t)b_uniform * m.sb + (int64_t)w.row_lo * m.st) * (int64_t)sizeof(T));
    int j0, n0;
    if (m.st <= m.sn) {                                    // time fastest in memory: e -> (n = e / kFT, s = e % kFT)
        j0 = r % kFT; n0 = r / kFT;
        st.jrow = 0; st.ninc = (RW / kFT) * snb; st.jmax = kFT - 1;
        st.lds0 = j0 * kFS + n0; st.ldsinc = RW / kFT;
    } else {                                               // state fastest: e -> (s = e / 16, n = e % 16)
        j0 = r / kFS; n0 = r % kFS;
        st.jrow = RW >= kFS ? RW / kFS : 1; st.ninc = 0; st.jmax = st.jrow - 1;
        st.lds0 = j0 * kFS + n0; st.ldsinc = st.jrow * kFS;
    }
    const int32_t lane_steps = w.dT < 0 ? (st.jmax - j0) * (-w.dT) : j0 * w.dT;       // >= 0
    st.voff = (uint32_t)(w.lane_row + lane_steps) * (uint32_t)st.stb + (uint32_t)n0 * (uint32_t)snb;
    return st;
}
template <typename T, int RW>
__device__ __forceinline__ void stage_fetch_buf(float (&v)[StageStream<RW>::EPL], const StageStream<RW>& st, int32_t rows, int32_t dT) {
    // uniform rows of element i: rows + i * jrow * dT (+ jmax * dT for a descending order, where the lane part counts upwards)
    uint32_t so = (uint32_t)(rows + (dT < 0 ? st.jmax * dT : 0)) * (uint32_t)st.stb;
    const uint32_t inc = (uint32_t)(st.jrow * dT * st.stb + st.ninc);
#pragma unroll
    for (int i = 0; i < StageStream<RW>::EPL; ++i) {
        v[i] = BufIO<T>::ld(st.rs, st.voff, so);
        so += inc;
    }
}
// the same fetch without the conversion (see BufIO::ld_raw): raw element bits
template <typename T, int RW>
__device__ __forceinline__ void stage_fetch_raw(uint32_t (&v)[StageStream<RW>::EPL], const StageStream<RW>& st, int32_t rows, int32_t dT) {
    uint32_t so = (uint32_t)(rows + (dT < 0 ? st.jmax * dT : 0)) * (uint32_t)st.stb;
    const uint32_t inc = (uint32_t)(st.jrow * dT * st.stb + st.ninc);
#pragma unroll
    for (int i = 0; i < StageStream<RW>::EPL; ++i) {
        v[i] = BufIO<T>::ld_raw(st.rs, st.voff, so);
        so += inc;
    }
}
template <int RW>
__device__ __forceinline__ void stage_park_buf(const float (&v)[StageStream<RW>::EPL], const StageStream<RW>& st, float* lds_item) {
#pragma unroll
    for (int i = 0; i < StageStream<RW>::EPL; ++i) lds_item[st.lds0 + i * st.ldsinc] = v[i];
}

// U of sub-tile s computed directly (the backward walks the sub-tiles downwards)
__device__ __forceinline__ int32_t fast_U_of(const TimeMap& tm, int32_t s) {
    if (tm.ns > 1) {
        const uint32_t t8 = (uint32_t)s * (uint32_t)kFT;
        const uint32_t jj = t8 / (uint32_t)tm.ns, kk = t8 - jj * (uint32_t)tm.ns;
        return (int32_t)(kk * (uint32_t)tm.sA + jj);
    }
    return s * kFT * (tm.sA + tm.sW);
}

// ------------------------------------------------------------------------------------------------------
// reduce-scatter over the RW lanes of a work item: on return lane r holds, in v[0], the sum over the item's
// lanes of the value the lanes had at index (r mod V).  V = min(RW, 32) values per call.
// ------------------------------------------------------------------------------------------------------
// gfx950 form: no LDS traffic.  A stage with partner lane ^ m keeps v[i] on lanes with bit m clear and v[m + i] on
// lanes with it set, and adds the partner's copy of the same element.
//   m = 16  v_permlane16_swap_b32 exchanges the odd 16-lane rows of one register with the even rows of another: after
//           swapping (v[i], v[16 + i]) every lane holds its own and its partner's copy of the element it keeps.
//   m <= 8  partners are in the same row of 16: DPP operands (row_ror:8, row_shl/shr:4 with bank masks, quad_perm).
template <int CTRL> __device__ __forceinline__ float dpp_get(float x) {      // x of the lane CTRL selects (0 if none)
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), CTRL, 0xf, 0xf, true));
}
template <int RW, int V>
__device__ __forceinline__ void reduce_scatter(float (&v)[V], int r) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    if constexpr (V >= 32) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const u32x2_t sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[16 + i]), false, false);
            v[i] = __uint_as_float(sw.x) + __uint_as_float(sw.y);
        }
    }
    if constexpr (V >= 16) {
        const bool up = (r & 8) != 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float slo = v[i] + dpp_get<0x128>(v[i]);                   // row_ror:8 = lane ^ 8
            const float shi = v[8 + i] + dpp_get<0x128>(v[8 + i]);
            v[i] = up ? shi : slo;
        }
    }
    if constexpr (V >= 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t lo = __float_as_uint(v[i]), hi = __float_as_uint(v[4 + i]);
            // lanes with bit 2 clear (banks 0, 2) receive lo of lane + 4, the others hi of lane - 4
            uint32_t recv = __builtin_amdgcn_update_dpp(0u, lo, 0x104, 0xf, 0x5, false);       // row_shl:4
            recv = __builtin_amdgcn_update_dpp(recv, hi, 0x114, 0xf, 0xa, false);              // row_shr:4
            const uint32_t keep = __builtin_amdgcn_update_dpp(lo, hi, 0xe4, 0xf, 0xa, false);  // identity on banks 1, 3
            v[i] = __uint_as_float(keep) + __uint_as_float(recv);
        }
    }
    {
        const bool up2 = (r & 2) != 0, up1 = (r & 1) != 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float slo = v[i] + dpp_get<0x4e>(v[i]);                    // quad_perm:[2,3,0,1] = lane ^ 2
            const float shi = v[2 + i] + dpp_get<0x4e>(v[2 + i]);
            v[i] = up2 ? shi : slo;
        }
        const float slo = v[0] + dpp_get<0xb1>(v[0]);                        // quad_perm:[1,0,3,2] = lane ^ 1
        const float shi = v[1] + dpp_get<0xb1>(v[1]);
        v[0] = up1 ? shi : slo;
    }
    if (RW > V) v[0] += __shfl_xor(v[0], V);               // RW == 64: fold the two 32-lane halves
}

}  // namespace segm
