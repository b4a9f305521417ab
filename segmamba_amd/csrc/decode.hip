// Single-token decode steps (C ABI: segm_causal_conv1d_update, segm_selective_state_update).
//
// Replace  causal_conv1d_cuda.causal_conv1d_update(x, conv_state, weight, bias?, silu)
//              reference causal-conv1d/csrc/causal_conv1d.cpp:270-330 (oracle: causal_conv1d_interface.py:84-104)
//          selective_state_update(state, x, dt, A, B, C, D?, z?, dt_bias?, dt_softplus)
//              reference mamba/mamba_ssm/ops/triton/selective_state_update.py:99-155 (oracle: :157-192), a Triton kernel
// - the two native ops behind `Mamba.step` (mamba_simple.py:356-401).  SegMamba itself never decodes; these
// complete the package API (§8f rank 4).  One lane per (batch, channel): the per-channel state row lives in registers for
// the step, everything is read and written once; launch-latency bound at decode sizes.
#include "segm_device.h"

namespace segm {

constexpr int kDecBlock = 256;

struct ConvUpdDev {
    const char* x;      int64_t x_sb, x_sd;
    char* state;        int64_t s_sb, s_sd, s_sw;
    char* out;          int64_t o_sb, o_sd;
    const float* weight;
    const float* bias;
    int32_t batch, dim, width, silu;
};

// state <- roll(state, -1); state[..., -1] = x; out = act(sum_w state * weight + bias)     (causal_conv1d_interface.py:98-104)
template <typename T>
__global__ void __launch_bounds__(kDecBlock) conv1d_update_kernel(ConvUpdDev P) {
    const int64_t i = (int64_t)blockIdx.x * kDecBlock + threadIdx.x;
    if (i >= (int64_t)P.batch * P.dim) return;
    const int b = (int)(i / P.dim), d = (int)(i - (int64_t)b * P.dim);
    T* st = reinterpret_cast<T*>(P.state) + b * P.s_sb + d * P.s_sd;
    const T xv = reinterpret_cast<const T*>(P.x)[b * P.x_sb + d * P.x_sd];
    float acc = P.bias ? P.bias[d] : 0.f;
    for (int w = 0; w < P.width; ++w) {
        const T v = w + 1 < P.width ? st[(w + 1) * P.s_sw] : xv;
        st[w * P.s_sw] = v;
        acc += to_f32(v) * P.weight[d * P.width + w];
    }
    if (P.silu) acc = acc * sigmoidf(acc);
    reinterpret_cast<T*>(P.out)[b * P.o_sb + d * P.o_sd] = from_f32<T>(acc);
}

struct StateUpdDev {
    char* state;        int64_t s_sb, s_sd, s_sn;
    const char* x;      int64_t x_sb, x_sd;
    const char* dt;     int64_t dt_sb, dt_sd;
    const char* z;      int64_t z_sb, z_sd;
    char* out;          int64_t o_sb, o_sd;
    const char* B;      int64_t B_sb, B_sn;
    const char* C;      int64_t C_sb, C_sn;
    const float* A;
    const float* D;
    const float* dt_bias;
    int32_t batch, dim, dstate, dt_softplus;
};

// dt' = softplus(dt + bias); state <- state exp(dt' A) + dt' B x; out = <state, C> + D x; out *= silu(z)
// (selective_state_update.py:184-192; the state is rounded to its own dtype before it enters the output, as there)
template <typename T, typename S>
__global__ void __launch_bounds__(kDecBlock) state_update_kernel(StateUpdDev P) {
    const int64_t i = (int64_t)blockIdx.x * kDecBlock + threadIdx.x;
    if (i >= (int64_t)P.batch * P.dim) return;
    const int b = (int)(i / P.dim), d = (int)(i - (int64_t)b * P.dim);
    S* st = reinterpret_cast<S*>(P.state) + b * P.s_sb + d * P.s_sd;
    const float x = to_f32(reinterpret_cast<const T*>(P.x)[b * P.x_sb + d * P.x_sd]);
    float dt = to_f32(reinterpret_cast<const T*>(P.dt)[b * P.dt_sb + d * P.dt_sd]);
    if (P.dt_bias) dt += P.dt_bias[d];
    if (P.dt_softplus) dt = softplus20(dt);
    const T* Bp = reinterpret_cast<const T*>(P.B) + b * P.B_sb;
    const T* Cp = reinterpret_cast<const T*>(P.C) + b * P.C_sb;
    float acc = 0.f;
    for (int n = 0; n < P.dstate; ++n) {
        const float a = fast_exp(dt * P.A[(int64_t)d * P.dstate + n]);
        const float h = to_f32(st[n * P.s_sn]) * a + dt * to_f32(Bp[n * P.B_sn]) * x;
        const S hs = from_f32<S>(h);
        st[n * P.s_sn] = hs;
        acc += to_f32(hs) * to_f32(Cp[n * P.C_sn]);
    }
    if (P.D) acc += x * P.D[d];
    if (P.z) {
        const float z = to_f32(reinterpret_cast<const T*>(P.z)[b * P.z_sb + d * P.z_sd]);
        acc *= z * sigmoidf(z);
    }
    reinterpret_cast<T*>(P.out)[b * P.o_sb + d * P.o_sd] = from_f32<T>(acc);
}

template <typename T>
static void launch_state_update(const StateUpdDev& P, int state_dtype, dim3 grid, hipStream_t st) {
    if (state_dtype == SEGM_F32) hipLaunchKernelGGL((state_update_kernel<T, float>), grid, dim3(kDecBlock), 0, st, P);
    else if (state_dtype == SEGM_F16) hipLaunchKernelGGL((state_update_kernel<T, f16_t>), grid, dim3(kDecBlock), 0, st, P);
    else hipLaunchKernelGGL((state_update_kernel<T, bf16_t>), grid, dim3(kDecBlock), 0, st, P);
}

static bool dtype_ok(int d) { return d == SEGM_F32 || d == SEGM_F16 || d == SEGM_BF16; }

}  // namespace segm

using namespace segm;

extern "C" int segm_causal_conv1d_update(const segm_conv1d_update_args* a) {
    if (!a) return SEGM_E_NULL;
    if (a->batch <= 0 || a->dim <= 0) return SEGM_E_SHAPE;
    if (a->width < 2 || a->width > 4) return SEGM_E_WIDTH;
    if (!dtype_ok(a->dtype)) return SEGM_E_DTYPE;
    if (!a->x || !a->conv_state || !a->out || !a->weight) return SEGM_E_NULL;
    ConvUpdDev P;
    P.x = (const char*)a->x; P.x_sb = a->x_stride_b; P.x_sd = a->x_stride_d;
    P.state = (char*)a->conv_state; P.s_sb = a->state_stride_b; P.s_sd = a->state_stride_d; P.s_sw = a->state_stride_w;
    P.out = (char*)a->out; P.o_sb = a->out_stride_b; P.o_sd = a->out_stride_d;
    P.weight = a->weight; P.bias = a->bias;
    P.batch = a->batch; P.dim = a->dim; P.width = a->width; P.silu = a->silu;
    const int64_t total = (int64_t)a->batch * a->dim;
    const dim3 grid((unsigned)((total + kDecBlock - 1) / kDecBlock));
    hipStream_t st = (hipStream_t)a->stream;
    if (a->dtype == SEGM_F32) hipLaunchKernelGGL((conv1d_update_kernel<float>), grid, dim3(kDecBlock), 0, st, P);
    else if (a->dtype == SEGM_F16) hipLaunchKernelGGL((conv1d_update_kernel<f16_t>), grid, dim3(kDecBlock), 0, st, P);
    else hipLaunchKernelGGL((conv1d_update_kernel<bf16_t>), grid, dim3(kDecBlock), 0, st, P);
    return (int)hipGetLastError();
}

extern "C" int segm_selective_state_update(const segm_state_update_args* a) {
    if (!a) return SEGM_E_NULL;
    if (a->batch <= 0 || a->dim <= 0) return SEGM_E_SHAPE;
    if (a->dstate < 1 || a->dstate > 256) return SEGM_E_DSTATE;         // the reference's limit (selective_scan.cpp:247)
    if (!dtype_ok(a->dtype) || !dtype_ok(a->state_dtype)) return SEGM_E_DTYPE;
    if (!a->state || !a->x || !a->dt || !a->out || !a->A || !a->B || !a->C) return SEGM_E_NULL;
    StateUpdDev P;
    P.state = (char*)a->state; P.s_sb = a->state_stride_b; P.s_sd = a->state_stride_d; P.s_sn = a->state_stride_n;
    P.x = (const char*)a->x; P.x_sb = a->x_stride_b; P.x_sd = a->x_stride_d;
    P.dt = (const char*)a->dt; P.dt_sb = a->dt_stride_b; P.dt_sd = a->dt_stride_d;
    P.z = (const char*)a->z; P.z_sb = a->z_stride_b; P.z_sd = a->z_stride_d;
    P.out = (char*)a->out; P.o_sb = a->out_stride_b; P.o_sd = a->out_stride_d;
    P.B = (const char*)a->B; P.B_sb = a->B_stride_b; P.B_sn = a->B_stride_n;
    P.C = (const char*)a->C; P.C_sb = a->C_stride_b; P.C_sn = a->C_stride_n;
    P.A = a->A; P.D = a->D; P.dt_bias = a->dt_bias;
    P.batch = a->batch; P.dim = a->dim; P.dstate = a->dstate; P.dt_softplus = a->dt_softplus;
    const int64_t total = (int64_t)a->batch * a->dim;
    const dim3 grid((unsigned)((total + kDecBlock - 1) / kDecBlock));
    hipStream_t st = (hipStream_t)a->stream;
    if (a->dtype == SEGM_F32) launch_state_update<float>(P, a->state_dtype, grid, st);
    else if (a->dtype == SEGM_F16) launch_state_update<f16_t>(P, a->state_dtype, grid, st);
    else launch_state_update<bf16_t>(P, a->state_dtype, grid, st);
    return (int)hipGetLastError();
}
