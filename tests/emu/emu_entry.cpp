// TEST INFRASTRUCTURE ONLY: compiles the kernel translation units as plain C++ against the CPU emulation
// of the HIP runtime (tests/emu/hip/hip_runtime.h) into libsegmamba_emu.so, exposing the same C ABI.
#include "../../segmamba_amd/csrc/capi.hip"
#include "../../segmamba_amd/csrc/scan_fwd.hip"
#include "../../segmamba_amd/csrc/scan_fwd_fast.hip"
#include "../../segmamba_amd/csrc/conv1d.hip"
#include "../../segmamba_amd/csrc/scan_bwd.hip"
#include "../../segmamba_amd/csrc/scan_bwd_fast.hip"
#include "../../segmamba_amd/csrc/conv3d_wgrad.hip"
#include "../../segmamba_amd/csrc/conv3d_fwd.hip"
#include "../../segmamba_amd/csrc/instnorm.hip"
#include "../../segmamba_amd/csrc/layout.hip"
#include "../../segmamba_amd/csrc/layernorm.hip"
