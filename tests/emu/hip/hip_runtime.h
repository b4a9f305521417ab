// TEST INFRASTRUCTURE ONLY - never part of the product.
//
// A tiny CPU emulation of the subset of the HIP programming model the kernels in
// segmamba_amd/csrc use, so their index arithmetic / barrier structure can be exercised in the
// build container (which has no GPU) before a gpurun call is spent.  The kernel sources are
// compiled UNCHANGED as plain C++ (host clang from ROCm's LLVM) with `-I tests/emu` placed ahead of
// the ROCm include path, so `#include <hip/hip_runtime.h>` resolves to this file.
//
// Model: every wave of a block is an OS thread and every lane a fiber (a hand-switched stack) of it; blocks run one
// after another.  A lane runs until its next synchronisation point, then the wave's scheduler resumes the next lane;
// when every live lane of the wave has arrived the wave goes on (wave-level sync) or first meets the other waves at a
// pthread barrier (block-level sync).  Kernels must be convergent at sync points, as on the hardware.
//   __shared__        -> function-local `static` (one copy, shared by the block's threads)
//   __syncthreads()   -> all lanes yield, then a pthread barrier over the block's waves
//   __shfl* / ballot  -> exchange through a per-wave buffer between two wave-level syncs (wave = 64 threads)
//   atomicAdd         -> compare-exchange loop
// Switching fibers costs tens of nanoseconds, so problem sizes are bounded by the arithmetic, not by futex traffic.
#pragma once
#include <pthread.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <tuple>
#include <utility>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
// `__shared__` arrays live in one named section so that the runtime can fill LDS with a NaN pattern before every block
// (HIPEMU_POISON_LDS=1): on the GPU LDS holds whatever the previous workgroup left, and 0 x NaN = NaN.
#define __shared__ static __attribute__((section("hipemu_lds")))

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };

extern thread_local uint3_emu threadIdx;
extern thread_local uint3_emu blockIdx;
extern thread_local dim3 blockDim;
extern thread_local dim3 gridDim;

typedef int hipError_t;
typedef void* hipStream_t;
static const hipError_t hipSuccess = 0;
static const hipError_t hipErrorInvalidValue = 1;
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }

namespace hipemu {
extern thread_local unsigned t_linear;      // linear thread id in block (of the lane that is running)
extern thread_local uint64_t* t_wave_buf;   // 64 exchange slots of this wave
void sync_block();                          // every lane of the block
void sync_wave();                           // every lane of the calling lane's wave
inline uint64_t* wave_slots() { return t_wave_buf; }

template <typename T> inline T exchange(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "exchange");
    uint64_t* s = wave_slots();
    uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
    s[t_linear % 64] = raw;
    sync_wave();
    uint64_t got = s[src_lane & 63];
    sync_wave();
    T out; memcpy(&out, &got, sizeof(T));
    return out;
}

struct LaunchArgsBase { virtual void run() = 0; virtual ~LaunchArgsBase() {} };
void launch(dim3 grid, dim3 block, LaunchArgsBase* body);

template <typename F, typename... A> struct LaunchArgs : LaunchArgsBase {
    F f; std::tuple<A...> args;
    LaunchArgs(F f_, A... a) : f(f_), args(a...) {}
    void run() override { std::apply(f, args); }
};
}  // namespace hipemu


template <typename K, typename... A>
static inline void hipLaunchKernelGGL_impl(K kernel, dim3 grid, dim3 block, A... args) {
    hipemu::LaunchArgs<K, A...> la(kernel, args...);
    hipemu::launch(grid, block, &la);
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL_impl(kernel, dim3(grid), dim3(block), __VA_ARGS__)

static inline void __syncthreads() { hipemu::sync_block(); }

template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    int lane = hipemu::t_linear % 64;
    return hipemu::exchange(v, lane ^ mask);
}
template <typename T> static inline T __shfl(T v, int src, int width = 64) {
    int lane = hipemu::t_linear % 64;
    int base = lane & ~(width - 1);
    return hipemu::exchange(v, base + (src & (width - 1)));
}
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int lane = hipemu::t_linear % 64;
    int src = lane + (int)d;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return hipemu::exchange(v, src);
}
static inline unsigned long long __ballot(int pred) {
    unsigned long long bit = pred ? 1ull : 0ull;
    unsigned long long r = 0;
    for (int i = 0; i < 64; ++i) r |= (hipemu::exchange(bit, i) << i);
    return r;
}

static inline float atomicAdd(float* addr, float val) {
    uint32_t* ia = reinterpret_cast<uint32_t*>(addr);
    uint32_t old = __atomic_load_n(ia, __ATOMIC_RELAXED);
    for (;;) {
        float f; memcpy(&f, &old, 4); f += val;
        uint32_t nw; memcpy(&nw, &f, 4);
        if (__atomic_compare_exchange_n(ia, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
            float o; memcpy(&o, &old, 4); return o;
        }
    }
}

// vector types / raw builtins used by the kernels
struct float4 { float x, y, z, w; } __attribute__((aligned(16)));
struct float2 { float x, y; } __attribute__((aligned(8)));
struct uint4 { unsigned x, y, z, w; } __attribute__((aligned(16)));
static inline float2 make_float2(float x, float y) { float2 r; r.x = x; r.y = y; return r; }
static inline float hipemu_exp2f(float x) { return exp2f(x); }
static inline float hipemu_rcpf(float x) { return 1.0f / x; }
static inline float hipemu_log2f(float x) { return log2f(x); }
#define __builtin_amdgcn_exp2f hipemu_exp2f
#define __builtin_amdgcn_rcpf hipemu_rcpf
#define __builtin_amdgcn_logf hipemu_log2f
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
#define __builtin_amdgcn_readfirstlane(x) (x)   /* only used on values that are uniform over the wave */

// ---- cross-lane data movement (scan_fast.h reduce_scatter: DPP operands and v_permlane16_swap_b32) ---------------------------
// __builtin_amdgcn_update_dpp(old, src, dpp_ctrl, row_mask, bank_mask, bound_ctrl): every lane reads `src` of the lane dpp_ctrl
// selects inside its row of 16; lanes whose row / bank (4 lanes) is masked off keep `old`; a source outside the row gives 0
// with bound_ctrl and `old` without.  Controls: quad_perm 0x00..0xff, row_shl:n 0x100+n (reads lane + n), row_shr:n 0x110+n
// (reads lane - n), row_ror:n 0x120+n (reads lane - n mod 16), row_mirror 0x140, row_half_mirror 0x141.
static inline uint32_t hipemu_update_dpp(uint32_t old, uint32_t src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const int lane = hipemu::t_linear % 64, row = lane >> 4, i = lane & 15;
    int s = i;
    bool valid = true;
    if (ctrl >= 0 && ctrl <= 0xff) s = (i & ~3) | ((ctrl >> (2 * (i & 3))) & 3);
    else if (ctrl >= 0x101 && ctrl <= 0x10f) { s = i + (ctrl - 0x100); valid = s <= 15; }
    else if (ctrl >= 0x111 && ctrl <= 0x11f) { s = i - (ctrl - 0x110); valid = s >= 0; }
    else if (ctrl >= 0x121 && ctrl <= 0x12f) s = (i - (ctrl - 0x120)) & 15;
    else if (ctrl == 0x140) s = 15 - i;
    else if (ctrl == 0x141) s = (i & 8) | (7 - (i & 7));
    else { fprintf(stderr, "hipemu: dpp_ctrl 0x%x not emulated\n", ctrl); abort(); }
    const uint32_t got = hipemu::exchange(src, valid ? (row * 16 + s) : lane);
    const bool enabled = ((row_mask >> row) & 1) && ((bank_mask >> (i >> 2)) & 1);
    if (!enabled) return old;
    return valid ? got : (bound_ctrl ? 0u : old);
}
#define __builtin_amdgcn_update_dpp hipemu_update_dpp
// v_add_f32_dpp vdst, vsrc0, vsrc1 <ctrl> row_mask bank_mask (no bound_ctrl): on the lanes of the enabled rows / banks
// vdst = vsrc0[the lane ctrl selects] + vsrc1[this lane] (a lane without a source is not written); every other lane keeps vdst.
// Models the bank-masked assembly of csrc/scan_bwd_w8.hip (the same operand / control / mask tuples are pasted into its asm text).
static inline float hipemu_v_add_f32_dpp(float vdst, float vsrc0, float vsrc1, int ctrl, int row_mask, int bank_mask) {
    uint32_t s0, d;
    memcpy(&s0, &vsrc0, 4);
    memcpy(&d, &vdst, 4);
    const uint32_t marker = 0x7fc0dead;                      // what a lane without write would see: detected below
    const uint32_t got = hipemu_update_dpp(marker, s0, ctrl, row_mask, bank_mask, false);
    const int lane = hipemu::t_linear % 64, row = lane >> 4, i = lane & 15;
    const bool enabled = ((row_mask >> row) & 1) && ((bank_mask >> (i >> 2)) & 1);
    if (!enabled || (got == marker && s0 != marker)) return vdst;
    float g;
    memcpy(&g, &got, 4);
    return g + vsrc1;
}
// v_permlane16_swap_b32 vdst, vsrc: the odd rows of 16 lanes of vdst trade places with the even rows of vsrc (lanes 16..31 of
// vdst <-> lanes 0..15 of vsrc, 48..63 <-> 32..47); returns {new vdst, new vsrc}
typedef unsigned hipemu_u32x2 __attribute__((ext_vector_type(2)));
static inline hipemu_u32x2 hipemu_permlane16_swap(uint32_t vdst, uint32_t vsrc, bool, bool) {
    const int lane = hipemu::t_linear % 64;
    const bool odd = (lane >> 4) & 1;
    const uint32_t src_of_partner = hipemu::exchange(vsrc, lane ^ 16);
    const uint32_t dst_of_partner = hipemu::exchange(vdst, lane ^ 16);
    hipemu_u32x2 r;
    r.x = odd ? src_of_partner : vdst;
    r.y = odd ? vsrc : dst_of_partner;
    return r;
}
#define __builtin_amdgcn_permlane16_swap hipemu_permlane16_swap
// ds_read_b64_tr_b16 (gfx950 LDS transpose read) as measured on the hardware (tools/experiments/tr16_probe.hip): every lane reads the
// four 16-bit values E at its own address; lane i of a 16-lane group then receives E[4 j + (i >> 2)][i & 3] for j = 0 .. 3
typedef short hipemu_v4s __attribute__((ext_vector_type(4)));
static inline hipemu_v4s hipemu_ds_read_tr16(const void* p) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, static_cast<const char*>(p) + 4, 4);
    const int lane = hipemu::t_linear % 64, grp = lane >> 4, i = lane & 15;
    hipemu_v4s r;
    for (int j = 0; j < 4; ++j) {
        const int src = grp * 16 + 4 * j + (i >> 2);
        const uint32_t slo = hipemu::exchange(lo, src), shi = hipemu::exchange(hi, src);
        short e[4];
        memcpy(e, &slo, 4);
        memcpy(e + 2, &shi, 4);
        r[j] = e[i & 3];
    }
    return r;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) hipemu_ds_read_tr16((const void*)(p))
// v_permlane32_swap_b32 vdst, vsrc: the upper 32 lanes of vdst trade places with the lower 32 lanes of vsrc
static inline hipemu_u32x2 hipemu_permlane32_swap(uint32_t vdst, uint32_t vsrc, bool, bool) {
    const int lane = hipemu::t_linear % 64;
    const bool upper = lane >= 32;
    const uint32_t src_of_partner = hipemu::exchange(vsrc, lane ^ 32);
    const uint32_t dst_of_partner = hipemu::exchange(vdst, lane ^ 32);
    hipemu_u32x2 r;
    r.x = upper ? src_of_partner : vdst;
    r.y = upper ? vsrc : dst_of_partner;
    return r;
}
#define __builtin_amdgcn_permlane32_swap hipemu_permlane32_swap

// ---- raw buffer resources (scan_fast.h Stream / StageStream): base pointer + lane byte offset + uniform byte offset ------------
// Range check as the hardware applies it to a raw buffer (stride 0): the LANE offset (voffset + immediate) is compared with
// num_records - the scalar offset takes no part in it; a load out of range returns zeros without touching memory, a store is
// dropped (conv3d_wgrad.hip relies on both: zero padding by offset, whole rows by num_records = 0).
struct hipemu_rsrc { char* base; uint32_t num_records; };
typedef hipemu_rsrc __amdgpu_buffer_rsrc_t;
static inline hipemu_rsrc hipemu_make_rsrc(void* p, short, int n, int) { hipemu_rsrc r; r.base = static_cast<char*>(p); r.num_records = (uint32_t)n; return r; }
#define __builtin_amdgcn_make_buffer_rsrc hipemu_make_rsrc
template <typename V> static inline V hipemu_buf_ld(hipemu_rsrc r, uint32_t voff, uint32_t soff) {
    V v;
    if ((uint64_t)voff + sizeof(V) > (uint64_t)r.num_records) { memset(&v, 0, sizeof(V)); return v; }
    memcpy(&v, r.base + (size_t)voff + (size_t)soff, sizeof(V)); return v;
}
template <typename V> static inline void hipemu_buf_st(V v, hipemu_rsrc r, uint32_t voff, uint32_t soff) {
    if ((uint64_t)voff + sizeof(V) > (uint64_t)r.num_records) return;
    memcpy(r.base + (size_t)voff + (size_t)soff, &v, sizeof(V));
}
#define __builtin_amdgcn_raw_buffer_load_b16(r, v, s, aux) hipemu_buf_ld<unsigned short>(r, v, s)
#define __builtin_amdgcn_raw_buffer_load_b32(r, v, s, aux) hipemu_buf_ld<uint32_t>(r, v, s)
typedef uint32_t hipemu_u32x4 __attribute__((ext_vector_type(4)));
#define __builtin_amdgcn_raw_buffer_load_b128(r, v, s, aux) hipemu_buf_ld<hipemu_u32x4>(r, v, s)
#define __builtin_amdgcn_raw_buffer_load_b64(r, v, s, aux) hipemu_buf_ld<hipemu_u32x2>(r, v, s)
#define __builtin_amdgcn_raw_buffer_store_b16(d, r, v, s, aux) hipemu_buf_st<unsigned short>(d, r, v, s)
#define __builtin_amdgcn_raw_buffer_store_b32(d, r, v, s, aux) hipemu_buf_st<uint32_t>(d, r, v, s)
#define __builtin_amdgcn_raw_buffer_store_b128(d, r, v, s, aux) hipemu_buf_st<hipemu_u32x4>(d, r, v, s)
#define __builtin_amdgcn_raw_buffer_store_b64(d, r, v, s, aux) hipemu_buf_st<hipemu_u32x2>(d, r, v, s)

// ---- MFMA / funnel-shift emulation (conv3d_wgrad.hip) -------------------------------------------------------------
static inline uint32_t hipemu_alignbyte(uint32_t hi, uint32_t lo, uint32_t n) {
    uint64_t v = ((uint64_t)hi << 32) | lo;
    return (uint32_t)(v >> (8 * (n & 3)));
}
#define __builtin_amdgcn_alignbyte hipemu_alignbyte
// v_perm_b32: result byte i = byte sel[i] of the pool {lo = bytes 0..3, hi = bytes 4..7} (selectors 0..7 only)
static inline uint32_t hipemu_perm(uint32_t hi, uint32_t lo, uint32_t sel) {
    const uint64_t pool = ((uint64_t)hi << 32) | lo;
    uint32_t out = 0;
    for (int i = 0; i < 4; ++i) out |= (uint32_t)((pool >> (8 * ((sel >> (8 * i)) & 7))) & 0xff) << (8 * i);
    return out;
}
#define __builtin_amdgcn_perm hipemu_perm
typedef __bf16 hipemu_bf16x8 __attribute__((ext_vector_type(8)));
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
namespace hipemu { extern std::vector<unsigned char>* g_wave_big; }   // 64 lanes x 64 bytes per wave
// D = A * B + C with A[i = l & 15][k = 8 (l >> 4) .. +7], B[k][j = l & 15], D[row = 4 (l >> 4) + r][col = l & 15]
static inline hipemu_f32x4 hipemu_mfma_16x16x32_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x4 c, int, int, int) {
    const unsigned lane = hipemu::t_linear % 64, wave = hipemu::t_linear / 64;
    unsigned char* buf = hipemu::g_wave_big->data() + (size_t)wave * 64 * 64;
    memcpy(buf + lane * 64, &a, 16);
    memcpy(buf + lane * 64 + 16, &b, 16);
    hipemu::sync_wave();
    hipemu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (lane >> 4) + r, col = lane & 15;
        float s = 0.f;
        for (int k = 0; k < 32; ++k) {
            __bf16 av, bv;
            memcpy(&av, buf + (row + 16 * (k / 8)) * 64 + 2 * (k % 8), 2);
            memcpy(&bv, buf + (col + 16 * (k / 8)) * 64 + 16 + 2 * (k % 8), 2);
            s += (float)av * (float)bv;
        }
        d[r] += s;
    }
    hipemu::sync_wave();
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 hipemu_mfma_16x16x32_bf16
typedef _Float16 hipemu_f16x8 __attribute__((ext_vector_type(8)));
static inline hipemu_f32x4 hipemu_mfma_16x16x32_f16(hipemu_f16x8 a, hipemu_f16x8 b, hipemu_f32x4 c, int, int, int) {
    const unsigned lane = hipemu::t_linear % 64, wave = hipemu::t_linear / 64;
    unsigned char* buf = hipemu::g_wave_big->data() + (size_t)wave * 64 * 64;
    memcpy(buf + lane * 64, &a, 16);
    memcpy(buf + lane * 64 + 16, &b, 16);
    hipemu::sync_wave();
    hipemu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (lane >> 4) + r, col = lane & 15;
        float s = 0.f;
        for (int k = 0; k < 32; ++k) {
            _Float16 av, bv;
            memcpy(&av, buf + (row + 16 * (k / 8)) * 64 + 2 * (k % 8), 2);
            memcpy(&bv, buf + (col + 16 * (k / 8)) * 64 + 16 + 2 * (k % 8), 2);
            s += (float)av * (float)bv;
        }
        d[r] += s;
    }
    hipemu::sync_wave();
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_f16 hipemu_mfma_16x16x32_f16
