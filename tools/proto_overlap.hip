// GPU probe (not product code): do VALU work and the row streams overlap?  A copy of three 16-bit row streams into two
// (8-step sub-tiles, next sub-tile's loads in flight) with K v_exp_f32 + K v_fma_f32 per lane and step added, either
// independent of the loaded data or fed by it.  If the time follows max(copy, valu) the pipes overlap; if it follows the sum
// something serialises them.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/proto_overlap.hip -o build/proto_overlap
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ float bf(uint16_t w) { return __uint_as_float((uint32_t)w << 16); }
__device__ __forceinline__ uint16_t tobf(float f) { return (uint16_t)(__float_as_uint(f) >> 16); }

template <int K, bool DEP, bool MEM> __global__ void __launch_bounds__(256) kern(uint16_t* u, uint16_t* dl, uint16_t* z, uint16_t* oz, uint16_t* out,
                                                                        int st, int T) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * 4 + wave;
    const int64_t row0 = (int64_t)item * T;
    const uint16_t* up = u + row0 * st + lane; const uint16_t* dp = dl + row0 * st + lane; const uint16_t* zp = z + row0 * st + lane;
    uint16_t* op = oz + row0 * st + lane; uint16_t* yp = out + row0 * st + lane;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = lane * 1e-3f + i;
    uint16_t nu[8], nd[8], nz[8];
    auto fetch = [&](int t0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MEM) { nu[j] = up[(t0 + j) * st]; nd[j] = dp[(t0 + j) * st]; nz[j] = zp[(t0 + j) * st]; }
            else { nu[j] = (uint16_t)(t0 + j); nd[j] = (uint16_t)(t0 * 3 + j); nz[j] = (uint16_t)(t0 + 2 * j); }
        }
    };
    fetch(0);
    float sink = 0.f;
    for (int t0 = 0; t0 < T; t0 += 8) {
        uint16_t cu[8], cd[8], cz[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { cu[j] = nu[j]; cd[j] = nd[j]; cz[j] = nz[j]; }
        fetch(t0 + 8 < T ? t0 + 8 : t0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = bf(cu[j]), b = bf(cd[j]), c = bf(cz[j]);
            const float x = DEP ? a * 0.01f : 0.01f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float e;
                asm volatile("v_exp_f32 %0, %1" : "=v"(e) : "v"(acc[k & 15]));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(acc[k & 15]) : "v"(e), "v"(x), "v"(acc[(k + 1) & 15]));
            }
            const float y = a + b + (DEP ? acc[0] : 0.f), yz = c + a;
            if (MEM) { yp[(t0 + j) * st] = tobf(y); op[(t0 + j) * st] = tobf(yz); }
            else sink += y + yz;
        }
    }
    float r = sink;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc[i];
    if (r == 123.456f) out[0] = 1;
}

template <int K, bool DEP, bool MEM> static float run(uint16_t* u, uint16_t* dl, uint16_t* z, uint16_t* oz, uint16_t* out, int st, int T, int64_t rows) {
    const int nblk = (int)(rows / T / 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((kern<K, DEP, MEM>), dim3(nblk), dim3(256), 0, 0, u, dl, z, oz, out, st, T);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((kern<K, DEP, MEM>), dim3(nblk), dim3(256), 0, 0, u, dl, z, oz, out, st, T);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 20 * 1e3f;
}

int main() {
    const int64_t rows = 2 * 262144;
    const int st = 192;
    const size_t n = (size_t)rows * st;
    uint16_t *u, *dl, *z, *oz, *out;
    CHK(hipMalloc(&u, n * 2)); CHK(hipMalloc(&dl, n * 2)); CHK(hipMalloc(&z, n * 2)); CHK(hipMalloc(&oz, n * 2)); CHK(hipMalloc(&out, n * 2));
    CHK(hipMemset(u, 0x3c, n * 2)); CHK(hipMemset(dl, 0x3c, n * 2)); CHK(hipMemset(z, 0x3c, n * 2));
    printf("us per launch, 33.6 M lane-steps; K exp + K fma per lane and step\n");
    for (int T : {64, 128}) {
#define ROW(K) printf("T=%3d K=%2d  valu only %6.1f   copy + independent valu %6.1f   copy + dependent valu %6.1f\n", T, K, \
                      run<K, false, false>(u, dl, z, oz, out, st, T, rows), run<K, false, true>(u, dl, z, oz, out, st, T, rows), \
                      run<K, true, true>(u, dl, z, oz, out, st, T, rows)); fflush(stdout);
        ROW(0) ROW(4) ROW(8) ROW(16) ROW(32)
    }
    return 0;
}
