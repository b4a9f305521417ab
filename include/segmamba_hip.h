/*
 * segmamba_hip.h - C ABI of libsegmamba_hip.so: the MI355X (gfx950) kernels of SegMamba's hot path.
 *
 * Every entry point replaces one native binding of the reference (ge-xing/SegMamba); the reference
 * line it replaces is cited next to it.  Plain pointers, sizes and strides only - no torch types.
 *
 *   - all tensors are caller-allocated DEVICE memory; the library never allocates, frees or retains
 *     a pointer (SURVEY.md §8b "Ownership");
 *   - every call is asynchronous on `stream` (a hipStream_t; NULL = the null stream) and may be issued
 *     concurrently from several host threads / processes;
 *   - return value: 0 on success, <0 = SEGM_E_* argument error (nothing was launched),
 *     >0 = the hipError_t reported by a launch.
 *
 * Layout.  Sequence tensors are described by explicit element strides for the logical index
 * (batch, time, channel), so both the reference's channel-first (B, D, L) tensors
 * (stride_t == 1) and the channel-last (B, L, D) tensors the kernels are tuned for
 * (stride_d == 1: one lane per channel, coalesced 64-channel rows) are accepted as they are.
 * Views into larger buffers (e.g. the x / z halves of `xz`, or dx / dz halves of `dxz`,
 * reference selective_scan_interface.py:175,244-245) are expressed through the strides.
 *
 * Time order.  The tri-directional Mamba block (reference mamba_simple.py:215-264) runs the same
 * operator on the sequence as stored, reversed (`xz.flip(-1)`, :231) and slice-interleaved
 * (`stack(chunk(nslices))`, :245-247).  Instead of materialising those copies, every operator takes
 * `time_order`: logical step tau is read from / written to physical index
 *      FORWARD      t = tau
 *      REVERSED     t = L-1-tau
 *      INTERLEAVED  t = (tau % nslices) * (L / nslices) + tau / nslices      (needs L % nslices == 0)
 * so outputs land where the reference's `.flip(-1)` / inverse permutation (:261,264) would put them.
 */
#ifndef SEGMAMBA_HIP_H
#define SEGMAMBA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEGM_ABI_VERSION 10

enum segm_dtype { SEGM_F32 = 0, SEGM_F16 = 1, SEGM_BF16 = 2 };
enum segm_time_order { SEGM_TIME_FORWARD = 0, SEGM_TIME_REVERSED = 1, SEGM_TIME_INTERLEAVED = 2 };

enum segm_status {
    SEGM_OK = 0,
    SEGM_E_NULL = -1,        /* a required pointer is NULL                                            */
    SEGM_E_SHAPE = -2,       /* non-positive size, dim % n_groups != 0, L % nslices != 0, ...        */
    SEGM_E_DSTATE = -3,      /* dstate outside [1, 16] for the scan (reference limit is 256,
                                selective_scan.cpp:247; SegMamba uses 16), [1, 256] for the decode step   */
    SEGM_E_DTYPE = -4,       /* unknown dtype code                                                    */
    SEGM_E_WIDTH = -5,       /* conv width outside [2, 4]  (reference causal_conv1d.cpp:157)          */
    SEGM_E_WORKSPACE = -6,   /* workspace pointer NULL or too small                                   */
    SEGM_E_TIME_ORDER = -7   /* unknown time order                                                    */
};

/* logical (batch, time, channel) view; strides in ELEMENTS */
typedef struct segm_seq {
    void* ptr;
    int64_t stride_b, stride_t, stride_d;
} segm_seq;

/* logical (batch, group, time, state) view of the input-dependent B / C matrices; strides in ELEMENTS */
typedef struct segm_bc {
    void* ptr;
    int64_t stride_b, stride_g, stride_t, stride_n;
} segm_bc;

/* ------------------------------------------------------------------------------------------------
 * Selective scan, forward.
 * Replaces  selective_scan_cuda.fwd(u, delta, A, B, C, D?, z?, delta_bias?, delta_softplus)
 *           -> [out, x, (out_z)]      reference mamba/csrc/selective_scan/selective_scan.cpp:226-336
 *           (kernel selective_scan_fwd_kernel.cuh:67-303).
 *
 *   h_t = exp(delta_t * A) . h_{t-1} + delta_t * u_t * B_t ,  y_t = <C_t, h_t> + D * u_t ,
 *   out = y ,  out_z = y * silu(z) ,  delta = softplus(delta + delta_bias) if delta_softplus.
 *
 * Real A, input-dependent B and C (the only variant on the SegMamba path).
 * Instead of the reference's opaque `x` (per-2048-chunk scan state, selective_scan.cpp:304-313) the
 * state needed by the backward is an opaque checkpoint buffer `ckpt` of
 * segm_selective_scan_ckpt_bytes(); the reference's only other use of `x`
 * (`last_state = x[:, :, -1, 1::2]`, selective_scan_interface.py:40) is the explicit `last_state`.
 * ------------------------------------------------------------------------------------------------ */
typedef struct segm_scan_fwd_args {
    int32_t batch, dim, dstate, n_groups;
    int64_t seqlen;
    int32_t dtype;            /* segm_dtype of u, delta, z, B, C, out, out_z                         */
    int32_t delta_softplus;
    int32_t time_order;       /* segm_time_order                                                     */
    int32_t nslices;          /* INTERLEAVED only                                                    */
    int32_t chunk;            /* steps per work item; 0 = choose. Must match between fwd and bwd     */
    int32_t reserved;
    segm_seq u, delta;        /* required                                                            */
    segm_seq z;               /* ptr NULL = no gate                                                  */
    segm_seq out;             /* ptr NULL = do not store the un-gated y                              */
    segm_seq out_z;           /* required iff z.ptr                                                  */
    segm_bc B, C;             /* required                                                            */
    const float* A;           /* (dim, dstate) contiguous, fp32                                      */
    const float* D;           /* (dim) fp32 or NULL                                                  */
    const float* delta_bias;  /* (dim) fp32 or NULL                                                  */
    float* last_state;        /* (batch, dim, dstate) contiguous fp32 or NULL                        */
    float* ckpt;              /* segm_selective_scan_ckpt_bytes() or NULL (inference)                */
    void* workspace;          /* segm_selective_scan_fwd_workspace_bytes()                           */
    size_t workspace_bytes;
    void* stream;
    /* Optional (ABI 4): the causal depthwise conv1d + SiLU in front of the scan computed INSIDE the scan launches ("causal
     * depthwise conv1d fused into the same launch").  conv_width 0 = off.  With conv_width in [2, 4], `u` is the conv INPUT x and
     * every pass forms u_t = SiLU(conv_bias + sum_k conv_weight[d][k] x[t - (width-1-k)]) along this call's time order, rounded to
     * the element type exactly as segm_causal_conv1d_fwd stores it (results are bit-identical to conv1d followed by the scan).
     * Regular shapes with delta_softplus and a gate z only (SEGM_E_SHAPE otherwise).  Measured slower than the separate conv1d
     * launch on MI355X (the scan passes are bound by instructions issued): opt-in, see DESIGN.md section 0 row N1. */
    const float* conv_weight; /* (dim, conv_width) contiguous fp32                                  */
    const float* conv_bias;   /* (dim) fp32 or NULL                                                  */
    int32_t conv_width, reserved2;
    /* Optional (ABI 6): delta = dt_proj(x_dbl[:, :dt_rank]) formed INSIDE the forward passes (reference
     * selective_scan_interface.py:181-182 computes it with a GEMM launch and stores it).  dt_rank 0 = off.  With dt_rank in
     * [1, 8]: `dt_x` points at the dt columns of x_dbl (element type `dtype`, dt_rank consecutive columns per time row, strides in
     * elements), `dt_weight` is the (dim, dt_rank) projection weight as fp32, and `delta` becomes an OUTPUT: both passes form
     * delta_t = sum_r dt_weight[d][r] dt_x[t][r], rounded to `dtype` as the stored tensor is, and the apply pass writes it for the
     * backward.  Regular shapes with one B / C group only (SEGM_E_SHAPE otherwise).  Opt-in: see DESIGN.md section 0 row N1. */
    const void* dt_x;
    int64_t dt_stride_b, dt_stride_t;
    const float* dt_weight;
    int32_t dt_rank, reserved3;
} segm_scan_fwd_args;

int segm_selective_scan_fwd(const segm_scan_fwd_args* args);
/* `n` forward scans in one call: the three directions of a Mamba(bimamba_type="v3") layer (reference
 * mamba_simple.py:216-264 issues three selective_scan_cuda.fwd calls per layer, one per parameter set).  When the blocks share
 * batch / dim / dstate / seqlen / dtype / chunk / stream, have one B / C group and a regular shape, they run as ONE grid with a
 * direction axis (3 x the waves: the small stages of SegMamba cannot fill 1024 SIMDs with one direction); otherwise one after
 * the other, exactly as n calls of segm_selective_scan_fwd.  Every block owns its workspace / outputs. */
int segm_selective_scan_fwd_multi(const segm_scan_fwd_args* args, int32_t n);
size_t segm_selective_scan_fwd_workspace_bytes(int32_t batch, int32_t dim, int32_t dstate, int64_t seqlen,
                                               int32_t chunk);
size_t segm_selective_scan_ckpt_bytes(int32_t batch, int32_t dim, int32_t dstate, int64_t seqlen);
/* the chunk length `chunk = 0` resolves to (so callers can record it for the backward) */
int32_t segm_selective_scan_default_chunk(int32_t batch, int32_t dim, int64_t seqlen);
/* 1 when a scan of this geometry runs on the regular-shape kernels (dstate 16, channel count a multiple of 16 / 32 / 64, whole
 * chunks, a time order that is affine inside 8-step sub-tiles): the condition of segm_selective_scan_{fwd,bwd}_multi sharing one
 * grid and of the conv_weight option.  One B / C group is assumed; chunk 0 = the default chunk. */
int32_t segm_selective_scan_regular_shape(int32_t batch, int32_t dim, int32_t dstate, int64_t seqlen, int32_t chunk,
                                          int32_t time_order, int32_t nslices);

/* ------------------------------------------------------------------------------------------------
 * Selective scan, backward.
 * Replaces  selective_scan_cuda.bwd(u, delta, A, B, C, D?, z?, delta_bias?, dout, x?, out?, dz?,
 *                                   delta_softplus, recompute_out_z)
 *           -> [du, ddelta, dA, dB, dC, dD, ddelta_bias, (dz), (out_z)]
 *           reference selective_scan.cpp:338-492 (kernel selective_scan_bwd_kernel.cuh:75-489).
 *
 * du / ddelta / dz may alias slices of a larger buffer (dx / dz halves of dxz).  dA, dD,
 * ddelta_bias, dB, dC are fp32 and are OVERWRITTEN (the reference zero-fills then accumulates,
 * selective_scan.cpp:460-466; here the library clears what it accumulates into).
 * `out` (the un-gated y written by the forward) is required iff z.ptr; `ckpt` is the forward's.
 * ------------------------------------------------------------------------------------------------ */
typedef struct segm_scan_bwd_args {
    segm_scan_fwd_args f;     /* same meaning as in the forward; f.out = saved y (read), f.out_z ignored,
                                 f.last_state ignored, f.workspace ignored                             */
    segm_seq dout;            /* required; gradient w.r.t. out_z (or out when no z)                    */
    segm_seq du, ddelta;      /* required                                                             */
    segm_seq dz;              /* required iff f.z.ptr                                                  */
    segm_bc dB, dC;           /* logical (batch, group, time, state); fp32 unless dbc_native           */
    float* dA;                /* (dim, dstate) fp32                                                    */
    float* dD;                /* (dim) fp32 or NULL                                                    */
    float* ddelta_bias;       /* (dim) fp32 or NULL                                                    */
    void* workspace;          /* segm_selective_scan_bwd_workspace_bytes()                             */
    size_t workspace_bytes;
    int32_t dbc_native;       /* ABI 5.  0: dB / dC are fp32 (the reference's accumulation buffers,
                                 selective_scan.cpp:461-462).  1: dB / dC have f.dtype - what the reference
                                 returns after its final cast (:488) - and are written once, finished, e.g.
                                 straight into columns of the x_proj gradient operand.  Only the
                                 deterministic kernel does this (segm_selective_scan_bwd_deterministic()
                                 tells whether a launch takes it); SEGM_E_SHAPE otherwise             */
    int32_t reserved_b;
} segm_scan_bwd_args;

int segm_selective_scan_bwd(const segm_scan_bwd_args* args);
int segm_selective_scan_bwd_multi(const segm_scan_bwd_args* args, int32_t n);   /* see segm_selective_scan_fwd_multi */
size_t segm_selective_scan_bwd_workspace_bytes(int32_t batch, int32_t dim, int32_t dstate, int64_t seqlen,
                                               int32_t chunk);
/* 1 when segm_selective_scan_bwd(args) runs the kernels whose dB / dC are sums in a fixed order (per-d-tile fp32 slabs added in
 * tile order: no atomics, no zero-initialised buffer, dbc_native allowed), 0 when it takes the kernels that accumulate dB / dC
 * atomically over d-tiles - irregular shapes (dim not a multiple of 16, dstate != 16, ...) and views whose rows span more than
 * 4 GiB per batch element; those results are NOT bit-reproducible run to run (float atomics).  The answer is formed from the
 * forward tensors of args->f (shape, layout and the 32-bit span bound) - the same predicate the launch applies to all tensors: a
 * gradient tensor that breaks the span bound sends a dbc_native = 0 launch to the general kernels and makes a dbc_native = 1
 * launch return SEGM_E_SHAPE.  No launch, no side effect. */
int segm_selective_scan_bwd_deterministic(const segm_scan_bwd_args* args);

/* ------------------------------------------------------------------------------------------------
 * Causal depthwise conv1d (+ optional SiLU), forward / backward.
 * Replaces  causal_conv1d_cuda.causal_conv1d_fwd(x, weight, bias?, silu) -> out
 *           reference causal-conv1d/csrc/causal_conv1d.cpp:130-189 (kernel causal_conv1d_fwd.cu:39-130)
 *      and  causal_conv1d_cuda.causal_conv1d_bwd(x, weight, bias?, dout, dx?, silu)
 *           -> [dx, dweight, dbias]   reference causal_conv1d.cpp:191-268 (kernel causal_conv1d_bwd.cu:46-240).
 *
 *   o_t = bias + sum_w weight[d, w] * x_{t-(width-1-w)}   (zero left pad),  out = o * sigmoid(o) if silu.
 *
 * weight (dim, width) and bias (dim) are fp32 contiguous (the reference also accepts 16-bit
 * weights; cast on the host).  dweight / dbias are fp32 and OVERWRITTEN.
 * ------------------------------------------------------------------------------------------------ */
typedef struct segm_conv1d_args {
    int32_t batch, dim, width, silu;
    int64_t seqlen;
    int32_t dtype;            /* of x, out, dout, dx */
    int32_t time_order, nslices;
    int32_t reserved;
    segm_seq x;               /* required */
    segm_seq out;             /* forward: required.  backward: ignored */
    const float* weight;      /* required */
    const float* bias;        /* or NULL  */
    /* backward only */
    segm_seq dout, dx;
    float* dweight;           /* (dim, width) */
    float* dbias;             /* (dim) or NULL */
    void* workspace;          /* backward: segm_causal_conv1d_bwd_workspace_bytes() */
    size_t workspace_bytes;
    void* stream;
} segm_conv1d_args;

int segm_causal_conv1d_fwd(const segm_conv1d_args* args);
int segm_causal_conv1d_bwd(const segm_conv1d_args* args);
/* `n` launches in one call; consecutive blocks that share batch / dim / width / seqlen / dtype / stream (up to three: the three
 * directions of a Mamba v3 layer, reference mamba_simple.py:216-264) run as one grid with a direction axis */
int segm_causal_conv1d_fwd_multi(const segm_conv1d_args* args, int32_t n);
int segm_causal_conv1d_bwd_multi(const segm_conv1d_args* args, int32_t n);
size_t segm_causal_conv1d_bwd_workspace_bytes(int32_t batch, int32_t dim, int32_t width, int64_t seqlen);

/* ------------------------------------------------------------------------------------------------
 * Weight gradient of a 3x3x3 stride-1 pad-1 convolution (the stem / decoder convolutions of SegMamba).
 * The reference gets this from cuDNN through torch.nn.Conv3d (monai/networks/blocks/convolutions.py:143-151,
 * model_segmamba/segmamba.py:95-99); on MI355X MIOpen's im2col solver for the 48-channel 128^3 layers is the single
 * largest item of a training step, so this one operator has a hand-written MFMA kernel.
 *
 *   dW[co, ci, kz, ky, kx] = sum_{b,z,y,x} dY[b, co, z, y, x] * X[b, ci, z+kz-1, y+ky-1, x+kx-1]
 *
 * x, dy: bf16 or fp16, logical (batch, channel, depth, height, width), W contiguous, every other stride a multiple of 8
 * elements, 16-byte aligned bases (channel slices of NCDHW tensors qualify).  cout a multiple of 48, cin a multiple of
 * 48 or below 48 (a narrow first layer), width a multiple of 8.  dw: contiguous (cout, cin, 3, 3, 3), fp32 or bf16, OVERWRITTEN.
 * ------------------------------------------------------------------------------------------------ */
typedef struct segm_conv3d_wgrad_args {
    int32_t batch, cin, cout, depth, height, width;
    int32_t dtype;            /* of x and dy: SEGM_BF16 or SEGM_F16      */
    int32_t dw_dtype;         /* SEGM_BF16, SEGM_F16 or SEGM_F32         */
    const void* x;   int64_t x_stride_b, x_stride_c, x_stride_z, x_stride_y;
    const void* dy;  int64_t dy_stride_b, dy_stride_c, dy_stride_z, dy_stride_y;
    void* dw;
    void* workspace;          /* segm_conv3d_k3_wgrad_workspace_bytes() */
    size_t workspace_bytes;
    void* stream;
} segm_conv3d_wgrad_args;

int segm_conv3d_k3_wgrad(const segm_conv3d_wgrad_args* args);
size_t segm_conv3d_k3_wgrad_workspace_bytes(int32_t batch, int32_t cin, int32_t cout, int32_t depth, int32_t height,
                                            int32_t width);

/* ------------------------------------------------------------------------------------------------
 * 3x3x3 stride-1 pad-1 convolution, forward (and data gradient, given flipped / transposed weights).
 * Replaces torch.nn.Conv3d -> cuDNN for the 48-input-channel 3x3x3 layers (reference model_segmamba/segmamba.py:95-131,
 * monai/networks/blocks/dynunet_block.py:44-111; the data gradient is what autograd asks cuDNN for in their backward).
 *
 *   y[b, co, z, y, x] = bias[co] + sum_{ci, kz, ky, kx} w[co, ci, kz, ky, kx] * x[b, ci, z+kz-1, y+ky-1, x+kx-1]
 *
 * x, y, w_packed: bf16 or fp16 (one dtype).  x (batch, cin, depth, height, width); y (batch, cout, depth, height,
 * width); W contiguous, every other stride a
 * multiple of 8 elements, 16-byte aligned bases.  1 <= cin <= 48 (wider layers are sums over 48-channel blocks), cout a
 * multiple of 16, width a multiple of 8.
 * w_packed: (cout, 3, 3, 3, 48) contiguous, i.e. weight.permute(0, 2, 3, 4, 1) - the input channel fastest - zero
 * padded to 48 input channels.
 * bias: (cout) fp32 or NULL.
 * flags: SEGM_CONV_FWD_ACCUMULATE adds the result to what `y` already holds (the 48-channel blocks of a wider layer
 * accumulate in place); SEGM_CONV_FWD_CHAIN (cout % 48 == 0 only) selects the kernel whose K parts are pipelined
 * through LDS instead of reduced at every output row - same results up to the order of fp32 additions;
 * SEGM_CONV_FWD_PITCH48 (with CHAIN only) lays its LDS rows out without padding (a bank-conflict experiment);
 * SEGM_CONV_FWD_CHAIN32 (alone or with ACCUMULATE) is the chained kernel on 32-wide x blocks, two workgroups per CU.
 * ------------------------------------------------------------------------------------------------ */
enum segm_conv_fwd_flags { SEGM_CONV_FWD_ACCUMULATE = 1, SEGM_CONV_FWD_CHAIN = 2, SEGM_CONV_FWD_PITCH48 = 4,
                           SEGM_CONV_FWD_CHAIN32 = 8 };

typedef struct segm_conv3d_fwd_args {
    int32_t batch, cin, cout, depth, height, width;
    int32_t dtype, flags;
    const void* x;   int64_t x_stride_b, x_stride_c, x_stride_z, x_stride_y;
    void* y;         int64_t y_stride_b, y_stride_c, y_stride_z, y_stride_y;
    const void* w_packed;
    const float* bias;
    void* stream;
    /* ABI 8: statistics of the result for the InstanceNorm behind the convolution (reference dynunet_block.py:98-111: every 3x3x3
     * convolution of the stem / decoder is followed by one).  stats_partials (NULL = off): fp32 (batch, cout, stats_nparts, 4) that
     * receives {count, sum y, sum y^2, 0} of the values this launch stores (the fp32 sums before rounding; with
     * SEGM_CONV_FWD_ACCUMULATE: of the accumulated values), one partial per (depth, y part, x block [, x pair]) - hand it to
     * segm_instnorm_fwd's stats_partials.  stats_nparts must equal segm_conv3d_k3_fwd_stats_parts(); only the launches with
     * SEGM_CONV_FWD_CHAIN | SEGM_CONV_FWD_PITCH48 or SEGM_CONV_FWD_CHAIN32 have the epilogue (SEGM_E_SHAPE otherwise). */
    float* stats_partials;
    int32_t stats_nparts, reserved;
} segm_conv3d_fwd_args;

int segm_conv3d_k3_fwd(const segm_conv3d_fwd_args* args);
int32_t segm_conv3d_k3_fwd_stats_parts(int32_t depth, int32_t height, int32_t width, int32_t batch, int32_t cout, int32_t flags);

/* ------------------------------------------------------------------------------------------------
 * ABI 9 (round 6): the same convolution on CHANNEL-LAST volumes, 48 -> 48 channels (csrc/conv3d_cl.hip).
 * Replaces the same reference calls as segm_conv3d_k3_fwd (model_segmamba/segmamba.py:91-132,
 * monai/networks/blocks/dynunet_block.py:44-111) for activations stored (batch, depth, height, width, channel):
 *
 *   y[b, z, y, x, co] = bias[co] + sum_{ci, kz, ky, kx} w[co, ci, kz, ky, kx] * x[b, z+kz-1, y+ky-1, x+kx-1, ci]
 *
 * x, y: bf16 or fp16 (one dtype), channels contiguous, every other stride a multiple of 8 elements and >= 48, 16-byte aligned
 * bases, width a multiple of 16, channels == 48 (wider layers: sums over 48-channel blocks with SEGM_CONV_CL_ACCUMULATE).
 * w_image: the weights as MFMA operand fragments, (14, 9, 64, 8) elements of the activations' dtype: element i is
 * w.flatten()[index[i]] (0 where index[i] < 0) with index from segm_conv3d_k3_cl_pack_index (host function, no GPU work).
 * The data gradient is the same call on dy with the image of flip(w, (2, 3, 4)).transpose(0, 1).
 * bias: (48) fp32 or NULL.  flags: SEGM_CONV_CL_ACCUMULATE adds to what y holds; SEGM_CONV_CL_WAVES8 runs eight waves of two
 * voxel tiles per workgroup instead of four waves of four (a scheduling choice, same results).
 * ------------------------------------------------------------------------------------------------ */
enum segm_conv_cl_flags { SEGM_CONV_CL_ACCUMULATE = 1, SEGM_CONV_CL_WAVES8 = 2 };

typedef struct segm_conv3d_cl_args {
    int32_t batch, channels, depth, height, width;
    int32_t dtype, flags, reserved;
    const void* x;   int64_t x_stride_b, x_stride_z, x_stride_y, x_stride_x;
    void* y;         int64_t y_stride_b, y_stride_z, y_stride_y, y_stride_x;
    const void* w_image;
    const float* bias;
    void* stream;
} segm_conv3d_cl_args;

int segm_conv3d_k3_fwd_cl(const segm_conv3d_cl_args* args);
int segm_conv3d_k3_cl_pack_index(int32_t* out, int64_t n);

/* ------------------------------------------------------------------------------------------------
 * ABI 10: 3x3x3 stride-1 pad-1 convolution of WIDE layers on SMALL volumes (csrc/conv3d_cube.hip): NCDHW, cin % 32 == 0,
 * cout a multiple of 64 or 96, depth / height / width multiples of 8 - the 16^3 / 8^3 levels of SegMamba's encoder and decoder (192 ... 768
 * channels; reference model_segmamba/segmamba.py:91-132, monai/networks/blocks/dynunet_block.py:44-111, unetr_block.py:82-84;
 * torch.nn.Conv3d -> cuDNN there).  A workgroup owns 8 x 8 x 8 voxels x 64 / 96 / 128 output channels, stages the halo cube of
 * 32 input channels per round in LDS and runs all 27 taps from it; the contraction is split over workgroups and a second launch
 * adds the fp32 partial sums in a fixed order (+ bias, + the existing y with SEGM_CONV_CUBE_ACCUMULATE) and rounds once; with one
 * split (enough cubes to fill the device) the first launch finishes the values itself.
 *
 * x: (batch, cin, D, H, W), y: (batch, cout, D, H, W), element strides for b / c / z / y (x contiguous), every stride a multiple
 * of 8, 16-byte aligned bases.  w_image: cout * cin * 27 elements of x's dtype arranged by segm_conv3d_k3_cube_pack_index:
 * out[i] = flat index into the (cout_w, cin_w, 3, 3, 3) weight; flipped = 1 gives the image of the DATA GRADIENT (a convolution
 * of dy with cout = cin_w, cin = cout_w and mirrored taps) - the same launch computes it.
 * segm_conv3d_k3_cube_plan: the column tiles per wave (nt: 2 / 3 / 4 = 64 / 96 / 128 channels per workgroup) and the number of
 * splits the launch will use (values > 0 on entry are kept when valid), and the fp32 elements the workspace must hold.
 * ------------------------------------------------------------------------------------------------ */
enum segm_conv_cube_flags { SEGM_CONV_CUBE_ACCUMULATE = 1 };

typedef struct segm_conv3d_cube_args {
    int32_t batch, cin, cout, depth, height, width;
    int32_t dtype, flags;
    int32_t nt, splits;       /* 0 = the plan's choice */
    const void* x;   int64_t x_stride_b, x_stride_c, x_stride_z, x_stride_y;
    void* y;         int64_t y_stride_b, y_stride_c, y_stride_z, y_stride_y;
    const void* w_image;
    const float* bias;        /* (cout) fp32 or NULL */
    void* workspace;          /* fp32, segm_conv3d_k3_cube_plan's workspace_elems (0 with one split: may be NULL) */
    int64_t workspace_elems;
    void* stream;
    /* InstanceNorm partials of what the launch stores (segm_instnorm_fwd_args.stats_partials): fp32 (batch * cout, stats_nparts, 4)
     * {count, sum, sum of squares, -}, stats_nparts = segm_conv3d_k3_cube_stats_parts(depth, height, width, the plan's splits);
     * NULL = not wanted */
    float* stats_partials;
    int32_t stats_nparts, reserved;
} segm_conv3d_cube_args;

int segm_conv3d_k3_cube_fwd(const segm_conv3d_cube_args* args);
int segm_conv3d_k3_cube_plan(int32_t batch, int32_t cin, int32_t cout, int32_t depth, int32_t height, int32_t width,
                             int32_t* nt, int32_t* splits, int64_t* workspace_elems);
int segm_conv3d_k3_cube_pack_index(int32_t* out, int64_t n, int32_t cout_w, int32_t cin_w, int32_t flipped);
int32_t segm_conv3d_k3_cube_stats_parts(int32_t depth, int32_t height, int32_t width, int32_t splits);

/* The images of many weights in ONE launch (what a training step needs after every weight update): descriptor i says where the
 * (cout_w, cin_w, 3, 3, 3) weight starts in `src` (element offset; co_stride = elements between its output channels - a channel
 * slice of a wider weight keeps the wide stride; the (cin, 27) part contiguous), where its image starts in `out` (a multiple of
 * 8), whether it is the data-gradient image, and the first block of the grid that works on it; blocks per image =
 * (Cout / 16) * (Cin / 32) of the convolution the image is for.  descs is a DEVICE array, first_block ascending.  The result is
 * element for element what segm_conv3d_k3_cube_pack_index describes. */
typedef struct segm_cube_pack_desc {
    int64_t src_off, out_off;
    int32_t cout_w, cin_w, co_stride, flipped;
    int32_t first_block, reserved;
} segm_cube_pack_desc;

int segm_conv3d_k3_cube_pack_multi(const void* src, void* out, const segm_cube_pack_desc* descs, int32_t ndesc, int32_t nblocks,
                                   void* stream);

/* The weight gradient of those layers (segm_conv3d_wgrad_args as for segm_conv3d_k3_wgrad): cin % 32 == 0, cout % 64 == 0, depth /
 * height / width multiples of 8.  The contraction runs over voxels and NCDHW has x contiguous: both operands are staged in LDS in
 * their native row layout (dY cube and X halo cube of 64 x 32 channels), a wave owns a 16 x 16 (co, ci) tile for all 27 taps, the
 * kx = 0 / 2 operands are register shifts; cube ranges are split over workgroups and a second launch adds the partial sums in a
 * fixed order.  Replaces what autograd asks cuDNN for in the backward of those Conv3d layers. */
int segm_conv3d_k3_cube_wgrad(const segm_conv3d_wgrad_args* args);
size_t segm_conv3d_k3_cube_wgrad_workspace_bytes(int32_t batch, int32_t cin, int32_t cout, int32_t depth, int32_t height,
                                                 int32_t width);

/* ------------------------------------------------------------------------------------------------
 * InstanceNorm3d (+ residual) (+ activation), forward and backward.
 * Replaces the torch.nn.InstanceNorm3d -> [+ residual] -> ReLU / LeakyReLU chains of the stem and decoder
 * (reference model_segmamba/segmamba.py:96-130,147,169-187; monai/networks/blocks/dynunet_block.py:98-111), which the
 * reference runs as three to four separate ATen kernels per call: no affine parameters, no running statistics,
 * biased variance, y = act((x - mean) / sqrt(var + eps) + residual).
 *
 * x, residual, y, dy, dx, dresidual: (instances, spatial) contiguous, instances = batch * channels, one dtype.
 * mean, rstd: (instances) fp32, written by the forward and read by the backward.
 * act: 0 none, 1 ReLU, 2 LeakyReLU(slope).
 * Backward: `y` (the forward's output) is required iff act != 0 and a residual was added (the activation mask is then
 * not recomputable from x); pass NULL otherwise.  dresidual (NULL = not wanted) receives dy * act'(.).
 * ------------------------------------------------------------------------------------------------ */
typedef struct segm_instnorm_fwd_args {
    int32_t instances, dtype, act, reserved;
    int64_t spatial;
    float slope, eps;
    const void* x;
    const void* residual;     /* or NULL */
    void* y;
    float* mean;
    float* rstd;
    void* workspace;          /* segm_instnorm_workspace_bytes() */
    size_t workspace_bytes;
    void* stream;
    /* elements between consecutive (b, c) instances of x / residual / y; 0 = spatial (dense).  128^3 volumes are kept with a
     * padded channel stride (a 4 MiB stride puts all channels of a row into one L2 set / memory channel) */
    int64_t x_instance_stride, residual_instance_stride, y_instance_stride;
    /* ABI 8: statistics already summed by the producer of x (segm_conv3d_k3_fwd's stats_partials): fp32 (instances, stats_nparts, 4)
     * of {count, sum, sum of squares, -}; the statistics launch is skipped and the partials are merged (Chan's update, fixed order)
     * by the apply launch.  NULL / 0: the library makes its own pass over x. */
    const float* stats_partials;
    int32_t stats_nparts, reserved2;
} segm_instnorm_fwd_args;

typedef struct segm_instnorm_bwd_args {
    int32_t instances, dtype, act, reserved;
    int64_t spatial;
    float slope, reserved2;
    const void* x;
    const void* dy;
    const void* y;            /* see above; or NULL */
    const float* mean;
    const float* rstd;
    void* dx;
    void* dresidual;          /* or NULL */
    void* workspace;
    size_t workspace_bytes;
    void* stream;
    int64_t x_instance_stride, dy_instance_stride, y_instance_stride, dx_instance_stride, dresidual_instance_stride;   /* as above */
} segm_instnorm_bwd_args;

int segm_instnorm_fwd(const segm_instnorm_fwd_args* args);
int segm_instnorm_bwd(const segm_instnorm_bwd_args* args);
size_t segm_instnorm_workspace_bytes(int32_t instances, int64_t spatial);

/* ------------------------------------------------------------------------------------------------
 * Batched transpose (+ add): out[b, c, r] = in[b, r, c] (+ add[b, c, r]).
 * Replaces the transposing copies around a Mamba layer - `x.reshape(B, C, n).transpose(-1, -2)` feeding LayerNorm and
 * `out.transpose(-1, -2).reshape(B, C, *dims)` + skip (reference model_segmamba/segmamba.py:60-75) - which the
 * reference leaves to strided ATen copies.  in (batch, rows, cols), add / out (batch, cols, rows), contiguous.
 * ------------------------------------------------------------------------------------------------ */
typedef struct segm_transpose_args {
    int32_t batch, rows, cols, dtype;
    const void* in;
    const void* add;          /* or NULL */
    void* out;
    void* stream;
} segm_transpose_args;

int segm_transpose_add(const segm_transpose_args* args);

/* ------------------------------------------------------------------------------------------------
 * ABI 9: out = a + b + c, element-wise, one pass (fp32 sum, one rounding).  Replaces the two binary adds of
 * `out + out_b + out_s` in front of out_proj (reference mamba/mamba_ssm/modules/mamba_simple.py:160 / :264) and of the three
 * directions' `dxz` contributions in the backward pass of the v3 block (autograd's fan-in adds in the reference).
 * a, b, c, out: `count` elements of one dtype (fp32 / fp16 / bf16), dense, 16-byte aligned, count a multiple of 16 bytes' worth;
 * out may be a.
 * ------------------------------------------------------------------------------------------------ */
typedef struct segm_add3_args {
    int64_t count;
    int32_t dtype, reserved;
    const void* a;
    const void* b;
    const void* c;
    void* out;
    void* stream;
} segm_add3_args;

int segm_add3(const segm_add3_args* args);

/* ------------------------------------------------------------------------------------------------
 * ABI 10: out[i] = src[map(i)] for 16-bit elements, i < count (count % 8 == 0, out and map 16-byte aligned) - the one gather per
 * training step that refreshes every re-arranged 16-bit copy of a weight (fragment images, packed blocks; the reference casts and
 * re-lays-out nothing: torch.nn.Conv3d -> cuDNN reads the fp32 / autocast weight as it is).  mode 0: map = int32 source index per
 * element; mode 1: map = int32 (first index, step) per group of eight consecutive elements (an arithmetic progression in src).
 * ------------------------------------------------------------------------------------------------ */
typedef struct segm_gather16_args {
    int64_t count;
    int32_t mode, reserved;
    const void* src;
    const int32_t* map;
    void* out;
    void* stream;
} segm_gather16_args;

int segm_gather16(const segm_gather16_args* args);

/* ------------------------------------------------------------------------------------------------
 * Volume -> tokens with LayerNorm over the channels, forward and backward.
 * Replaces `x.reshape(B, C, n).transpose(-1, -2)` followed by `nn.LayerNorm(C)` at the entry of a Mamba layer
 * (reference model_segmamba/segmamba.py:60-66): a transposing copy, an fp32 LayerNorm and a cast in the reference.
 *
 *   forward   x (batch, channels, spatial) -> y (batch, spatial, channels) = (x - mean_c) * rstd_c * gamma + beta,
 *             mean / rstd (batch, spatial) fp32 (kept for the backward)
 *   backward  dy (batch, spatial, channels), x, mean, rstd, gamma -> dx (batch, channels, spatial), dgamma, dbeta (fp32,
 *             OVERWRITTEN)
 * x, y, dy, dx: one dtype, contiguous.  gamma, beta, dgamma, dbeta: fp32 (channels).  channels and spatial multiples of 8
 * (4 for fp32); channels <= 384 (192 for fp32).
 * ------------------------------------------------------------------------------------------------ */
typedef struct segm_layernorm_args {
    int32_t batch, channels, dtype, reserved;
    int64_t spatial;
    float eps, reserved2;
    const void* x;
    void* y;                  /* forward */
    const float* gamma;
    const float* beta;        /* forward */
    float* mean;
    float* rstd;
    const void* dy;           /* backward ... */
    void* dx;
    float* dgamma;
    float* dbeta;
    void* workspace;          /* segm_layernorm_tokens_workspace_bytes(); backward only */
    size_t workspace_bytes;
    void* stream;
} segm_layernorm_args;

int segm_layernorm_tokens_fwd(const segm_layernorm_args* args);
int segm_layernorm_tokens_bwd(const segm_layernorm_args* args);
size_t segm_layernorm_tokens_workspace_bytes(int32_t batch, int32_t channels, int64_t spatial);

/* ------------------------------------------------------------------------------------------------
 * Gradient clipping + SGD (momentum, Nesterov, weight decay) over a list of fp32 tensors, in two passes.
 * Replaces `torch.nn.utils.clip_grad_norm_(model.parameters(), 12)` followed by `optimizer.step()` of
 * `torch.optim.SGD(lr=1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)` (reference light_training/trainer.py:461-470,
 * 3_train.py:51-52):
 *
 *   norm = ||all gradients||_2 ;  c = min(1, max_norm / (norm + 1e-6))        (max_norm <= 0: c = 1)
 *   g' = c g + weight_decay p ;  m = momentum m + g' ;  p -= lr (nesterov ? g' + momentum m : m)
 *
 * params / grads / momenta / numel are HOST arrays of `ntensors` entries (device pointers; numel may be 0).  Momentum
 * buffers start at zero (torch's first step, buf = g', is then the same formula).  The gradients are read, not
 * rescaled in place.  workspace: segm_sgd_clip_step_workspace_bytes(); on return (stream order) its first two floats hold
 * {c, norm}.
 * ------------------------------------------------------------------------------------------------ */
typedef struct segm_sgd_args {
    int32_t ntensors, nesterov;
    float* const* params;
    const float* const* grads;
    float* const* momenta;
    const int64_t* numel;
    float lr, momentum, weight_decay, max_norm;
    void* workspace;
    size_t workspace_bytes;
    void* stream;
    /* ABI 8: fp16 autocast with a loss scale (GradScaler, reference light_training/trainer.py:65-67,461-466).  loss_scale: device
     * pointer to the scale S the gradients carry, or NULL (no scaling).  With it the step is unscale_ -> clip -> step without a pass
     * over the gradients: norm = |g| / S, update coefficient = clip / S; if |g| is inf / nan the parameters and momenta are left
     * untouched.  found_inf: device pointer that receives 1.0f (skipped) or 0.0f - the tensor torch._amp_update_scale_ takes - or NULL. */
    const float* loss_scale;
    float* found_inf;
} segm_sgd_args;

int segm_sgd_clip_step(const segm_sgd_args* args);
size_t segm_sgd_clip_step_workspace_bytes(int32_t ntensors, const int64_t* numel);

/* ------------------------------------------------------------------------------------------------
 * Cross entropy over the class axis of (batch, classes, spatial) logits, forward and gradient in one pass.
 * Replaces `nn.CrossEntropyLoss()(pred, label)` and its backward (reference 3_train.py:48,57-66).
 *
 *   loss_v = logsumexp_c(x[b, :, s]) - x[b, label, s] ;  dlogits[b, c, s] = softmax_c(x)[c] - [c == label]
 *   voxels whose label == ignore_index contribute neither loss nor gradient.  Any other label outside [0, classes) is a
 *   caller error (ATen raises a device-side assertion): it turns the loss sum and that voxel's gradient into NaN.
 *
 * logits, dlogits: (batch, classes, spatial) contiguous, one dtype (fp32 / fp16 / bf16; arithmetic in fp32); labels
 * (batch, spatial) int64; classes <= 16.  loss_partial / count_partial: fp32 arrays of segm_cross_entropy_partials()
 * entries (per-workgroup sums of the losses and of the number of counted voxels; the caller adds them and divides -
 * `mean` reduction - and scales dlogits by 1 / count).
 * ------------------------------------------------------------------------------------------------ */
typedef struct segm_cross_entropy_args {
    int32_t batch, classes, dtype, reserved;
    int64_t spatial, ignore_index;
    const void* logits;
    const int64_t* labels;
    void* dlogits;
    float* loss_partial;
    float* count_partial;
    void* stream;
} segm_cross_entropy_args;

int segm_cross_entropy(const segm_cross_entropy_args* args);
int32_t segm_cross_entropy_partials(int32_t batch, int64_t spatial);

/* ------------------------------------------------------------------------------------------------
 * Single-token decode steps (the native ops behind `Mamba.step`, reference mamba_simple.py:356-401).
 *
 * segm_causal_conv1d_update replaces causal_conv1d_cuda.causal_conv1d_update(x, conv_state, weight, bias?, silu)
 *   (reference causal-conv1d/csrc/causal_conv1d.cpp:270-330; Python causal_conv1d_interface.py:68-82):
 *   conv_state <- roll(conv_state, -1) with x in the last slot; out = act(sum_w conv_state * weight + bias).
 *   x, out (batch, dim), conv_state (batch, dim, width) of `dtype`, element strides; weight (dim, width), bias (dim) fp32.
 *
 * segm_selective_state_update replaces selective_state_update(state, x, dt, A, B, C, D?, z?, dt_bias?, dt_softplus)
 *   (reference mamba/mamba_ssm/ops/triton/selective_state_update.py:99-155, a Triton kernel):
 *   dt' = softplus(dt + dt_bias); state <- state exp(dt' A) + dt' B x; out = <state, C> + D x; out *= silu(z).
 *   state (batch, dim, dstate) of `state_dtype` (updated in place); x, dt, z, out (batch, dim) and B, C (batch, dstate) of
 *   `dtype`; A (dim, dstate), D, dt_bias (dim) fp32.  1 <= dstate <= 256.
 * ------------------------------------------------------------------------------------------------ */
typedef struct segm_conv1d_update_args {
    int32_t batch, dim, width, silu;
    int32_t dtype, reserved;
    const void* x;      int64_t x_stride_b, x_stride_d;
    void* conv_state;   int64_t state_stride_b, state_stride_d, state_stride_w;
    void* out;          int64_t out_stride_b, out_stride_d;
    const float* weight;
    const float* bias;        /* or NULL */
    void* stream;
} segm_conv1d_update_args;

int segm_causal_conv1d_update(const segm_conv1d_update_args* args);

typedef struct segm_state_update_args {
    int32_t batch, dim, dstate, dt_softplus;
    int32_t dtype, state_dtype;
    void* state;        int64_t state_stride_b, state_stride_d, state_stride_n;
    const void* x;      int64_t x_stride_b, x_stride_d;
    const void* dt;     int64_t dt_stride_b, dt_stride_d;
    const void* z;      int64_t z_stride_b, z_stride_d;      /* z NULL = no gate */
    void* out;          int64_t out_stride_b, out_stride_d;
    const void* B;      int64_t B_stride_b, B_stride_n;
    const void* C;      int64_t C_stride_b, C_stride_n;
    const float* A;
    const float* D;           /* or NULL */
    const float* dt_bias;     /* or NULL */
    void* stream;
} segm_state_update_args;

int segm_selective_state_update(const segm_state_update_args* args);

/* ------------------------------------------------------------------------------------------------
 * Row-streaming projection  y[m, :] = x[m, :] W^T + bias  for tall activations (rows >> k, n).
 * Replaces the `nn.Linear` calls of the Mamba block on (batch * length, channels) activations - in_proj, x_proj,
 * out_proj - and their data gradients (reference mamba_simple.py:204-208, 264; selective_scan_interface.py:185-205,
 * 247-275), which the reference hands to cuBLAS.  x (rows, k) and y (rows, n) with element row strides (views into wider
 * tensors are fine), w (n, k) contiguous - the nn.Linear layout - all of one 16-bit dtype; bias (n) fp32 or NULL.
 * k a multiple of 8, at most 2048 (up to 192: W stationary in registers; beyond, round 6: W streamed from L2 as well - the
 * projections of stages 2 / 3); n a multiple of 4; x rows 16-byte aligned, y rows 8-byte aligned.
 * accumulate != 0: y += x W^T (+ bias), e.g. `torch.addmm(dconv, dx_dbl, x_proj_weight)` of the reference's backward
 * (selective_scan_interface.py:276).
 * ------------------------------------------------------------------------------------------------ */
typedef struct segm_linear_args {
    int64_t rows;
    int32_t k, n;
    int32_t dtype, accumulate;
    const void* x;      int64_t x_stride_row;
    const void* w;
    const float* bias;
    void* y;            int64_t y_stride_row;
    void* stream;
} segm_linear_args;

int segm_linear_rows(const segm_linear_args* args);

/* ------------------------------------------------------------------------------------------------
 * Channel-first 1x1x1 convolution  y[b, co, s] = sum_ci w[co, ci] x[b, ci, s] + bias[co]  on (batch, channels, voxels)
 * activations whose voxels are contiguous (NCDHW, or channel slices of it).  Replaces the `Conv3d(kernel_size=1)` calls of
 * the conv stem that the reference hands to cuDNN: the residual branch `conv3` of MONAI's UnetResBlock
 * (monai/networks/blocks/dynunet_block.py:72-96), UnetOutBlock (:247-263), GSC.proj3 / proj4 and MlpChannel.fc1 / fc2
 * (model_segmamba/segmamba.py:78-131) - and, with the transposed weight, their data gradients.
 * x, w, y of one 16-bit dtype; w (cout, w_stride) row-major with w_stride >= cin a multiple of 8 and zero padding columns;
 * bias (cout) fp32 or NULL; cin, cout <= 96; spatial (voxels per channel) a multiple of 64; element strides of batch and
 * channel: multiples of 8 for x (16-byte aligned rows), of 4 for y.  accumulate != 0: y += ... (the second half of a
 * concatenated input, `conv3(cat(up, skip))` = conv3a(up) + conv3b(skip)).
 * ------------------------------------------------------------------------------------------------ */
typedef struct segm_pointwise_args {
    int32_t batch, cin, cout;
    int32_t dtype, accumulate;
    int64_t spatial;
    const void* x;      int64_t x_stride_b, x_stride_c;
    const void* w;      int32_t w_stride;
    const float* bias;
    void* y;            int64_t y_stride_b, y_stride_c;
    void* stream;
} segm_pointwise_args;

int segm_pointwise_cf(const segm_pointwise_args* args);

/* ------------------------------------------------------------------------------------------------
 * The encoder's stem convolution: Conv3d(kernel 7, stride 2, padding 3) on at most 4 input channels
 * (model_segmamba/segmamba.py:141; cuDNN in the reference), forward.
 * x4: the input channel-LAST with exactly 4 channels, (batch, din, hin, win, 4) contiguous (missing channels zero) - one
 * transposing copy of the small input, made by the caller; w_packed: (cout, 7, 7, 8, 4) contiguous = weight[co][ci][kz][ky][kx]
 * at [co][kz][ky][kx][ci], kx slot 7 and missing channels zero; bias (cout) fp32 or NULL; y (batch, cout, din/2, hin/2, win/2)
 * contiguous.  One 16-bit dtype; cout <= 48; din, hin even; win a multiple of 32.
 * ------------------------------------------------------------------------------------------------ */
typedef struct segm_stem_args {
    int32_t batch, cout;
    int32_t din, hin, win;
    int32_t dtype;
    const void* x4;
    const void* w_packed;
    const float* bias;
    void* y;
    void* stream;
    /* 0 / 0 = the stem proper (kernel 7, stride 2).  3 / 1: the first 3x3x3 stride-1 padding-1 convolution of the conv stem on
     * the same few input channels (UnetResBlock conv1 of `encoder1`, model_segmamba/segmamba.py:236-244): w_packed (cout, 3, 3, 8, 4),
     * y (batch, cout, din, hin, win), win a multiple of 16. */
    int32_t kernel_size, stride;
    int64_t y_channel_stride;   /* elements between channels of y (batch stride = cout x that); 0 = dense */
} segm_stem_args;

int segm_stem_conv_fwd(const segm_stem_args* args);

/* Weight gradient of the same convolution (the reference: cuDNN's backward-filter through autograd).
 * x4 as above; dy (batch, cout, din/2, hin/2, win/2) contiguous, same 16-bit dtype; dw_packed: fp32 (7, 7, cout16, 32) with
 * cout16 = cout rounded up to 16: [kz][ky][co][kx slot * 4 + ci] (slot 7 / missing channels / co >= cout are padding);
 * workspace: segm_stem_conv_wgrad_workspace_bytes() of per-slab partials, added in a fixed order (deterministic).
 * win in {64, 128, 256}. */
typedef struct segm_stem_wgrad_args {
    int32_t batch, cout;
    int32_t din, hin, win;
    int32_t dtype;
    const void* x4;
    const void* dy;
    float* dw_packed;
    void* workspace;    size_t workspace_bytes;
    void* stream;
    /* 0 / 0 = kernel 7, stride 2.  3 / 1: the 3x3x3 stride-1 layer (see segm_stem_args): dw_packed fp32 (3, 3, cout16, 16) =
     * [kz][ky][co][kx slot (4) * 4 + ci], workspace from segm_stem_conv_wgrad_workspace_bytes2; win in {32, 64, 128}. */
    int32_t kernel_size, stride;
    int64_t dy_channel_stride;  /* elements between channels of dy; 0 = dense */
} segm_stem_wgrad_args;

size_t segm_stem_conv_wgrad_workspace_bytes(int32_t batch, int32_t cout, int32_t din, int32_t hin);
size_t segm_stem_conv_wgrad_workspace_bytes2(int32_t batch, int32_t cout, int32_t din, int32_t hin, int32_t kernel_size, int32_t stride);
int segm_stem_conv_wgrad(const segm_stem_wgrad_args* args);

/* ------------------------------------------------------------------------------------------------
 * Weight gradients of the projections: out (m, n) fp32 = sum_k a[k][.] b[k][.] over a long k (tokens / voxels).
 * The reference gets them from autograd -> cuBLAS: dW of in_proj / out_proj / x_proj / dt_proj
 * (mamba/mamba_ssm/modules/mamba_simple.py:204-208,264; ops/selective_scan_interface.py:272-276) and of the 1x1x1
 * convolutions (monai/networks/blocks/dynunet_block.py:72-96,247-263).
 *   SEGM_WGEMM_TN  a (k, m), b (k, n) row-major, row strides a_stride_row / b_stride_row (elements; any alignment: rows that are
 *                  not 16-byte aligned take 2-byte loads); batch strides ignored; m, n <= 1024
 *   SEGM_WGEMM_NT  a (batch, m, k), b (batch, n, k), unit stride along k, k % 32 == 0, strides % 8 == 0, 16-byte aligned;
 *                  out = sum over the batch of a[i] b[i]^T; m, n <= 96
 * 16-bit operands, fp32 accumulation, per-wave partial sums in the workspace added in a fixed order (bitwise repeatable).
 * ------------------------------------------------------------------------------------------------ */
enum segm_wgemm_layout { SEGM_WGEMM_TN = 0, SEGM_WGEMM_NT = 1 };

typedef struct segm_wgrad_gemm_args {
    int32_t layout, dtype;
    int32_t m, n;
    int64_t k;
    int32_t batch, reserved;
    const void* a;  int64_t a_stride_row, a_stride_batch;
    const void* b;  int64_t b_stride_row, b_stride_batch;
    float* out;
    void* workspace;    size_t workspace_bytes;
    void* stream;
} segm_wgrad_gemm_args;

size_t segm_wgrad_gemm_workspace_bytes(int32_t layout, int32_t m, int32_t n, int64_t k, int32_t batch);

/* out (m, n) fp32 = wide^T skinny: wide (k, m) with unit column stride, m % 8 == 0, m <= 2048, rows 16-byte aligned (base pointer and
 * wide_stride_row % 8 == 0; otherwise SEGM_E_SHAPE); skinny (k, n <= 32); both 16-bit, row strides in elements, k = tokens.  The weight gradient of Mamba's dt_proj - `einsum("dB,Br->dr", ddelta, x_dbl[:, :R])`, reference
 * selective_scan_interface.py:272, R = 3 ... 24 - as the streaming reduction it is (the vendor GEMM spends a tile on three
 * columns).  Per-slab partial sums in the workspace, added in a fixed order.  out is OVERWRITTEN. */
typedef struct segm_skinny_tn_args {
    int64_t k;
    int32_t m, n;
    int32_t dtype, reserved;
    const void* wide;    int64_t wide_stride_row;
    const void* skinny;  int64_t skinny_stride_row;
    float* out;
    void* workspace;     size_t workspace_bytes;
    void* stream;
} segm_skinny_tn_args;
size_t segm_skinny_tn_workspace_bytes(int32_t m, int32_t n, int64_t k);
int segm_skinny_tn(const segm_skinny_tn_args* args);
int segm_wgrad_gemm(const segm_wgrad_gemm_args* args);
/* out[c] (fp32, OVERWRITTEN) = sum over batch and voxels of x[b][c][s]: the bias gradient `dy.sum((0, 2, 3, 4))` of the convolutions
 * that have a bias (reference segmamba.py:78-89, 95-131, 141, 254; cuDNN backward-bias / ATen there).  x (batch, channels, spatial)
 * with unit voxel stride, strides in elements (stride_channel >= spatial: padded volumes are fine); fp32 / fp16 / bf16.  Partials per
 * (b, segment, c) in the workspace, added in a fixed order. */
typedef struct segm_channel_sum_args {
    const void* x;
    int64_t stride_batch, stride_channel, spatial;
    int32_t batch, channels, dtype, reserved;
    float* out;
    void* workspace;     size_t workspace_bytes;
    void* stream;
} segm_channel_sum_args;
size_t segm_channel_sum_workspace_bytes(int32_t batch, int32_t channels, int64_t spatial);
int segm_channel_sum(const segm_channel_sum_args* args);


/* ------------------------------------------------------------------------------------------------ */
int segm_abi_version(void);
const char* segm_status_string(int status);

/* ------------------------------------------------------------------------------------------------
 * Depth-to-space / space-to-depth by 2 x 2 x 2 (ABI 7).
 * Replaces the permuting copy behind the GEMM of a ConvTranspose3d with kernel_size = stride = 2 (reference
 * monai/networks/blocks/unetr_block.py:52-60 via dynunet_block.get_conv_layer) and, in its backward, the inverse gather:
 *
 *   direction 0 (depth-to-space)   vol[b, c, 2z+i, 2y+j, 2x+k] = blk[b, c, i, j, k, z, y, x]
 *   direction 1 (space-to-depth)   blk[b, c, i, j, k, z, y, x] = vol[b, c, 2z+i, 2y+j, 2x+k]
 *
 * blk is contiguous (batch, channels, 2, 2, 2, depth, height, width); vol is (batch, channels, 2 depth, 2 height, 2 width) with
 * element strides vol_stride_b / _c / _z / _y and unit stride along x (padded volumes).  16-bit element types; width % 8 == 0;
 * 16-byte aligned rows.
 * ------------------------------------------------------------------------------------------------ */
typedef struct segm_d2s_args {
    int32_t batch, channels, depth, height, width;      /* of the LOW-resolution block tensor */
    int32_t dtype, direction, reserved;
    void* blk;
    void* vol;
    int64_t vol_stride_b, vol_stride_c, vol_stride_z, vol_stride_y;
    void* stream;
} segm_d2s_args;

int segm_depth_to_space2(const segm_d2s_args* args);

#ifdef __cplusplus
}
#endif
#endif /* SEGMAMBA_HIP_H */
