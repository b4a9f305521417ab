#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short v4s __attribute__((ext_vector_type(4)));
// probe: LDS holds lds[i] = i (16-bit).  Each lane passes byte address addr_tab[lane]; result[lane][0..3] is written out.
__global__ void probe(const int* addr_tab, short* out) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int a = addr_tab[threadIdx.x];
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)((__attribute__((address_space(3))) char*)lds + a));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = r[j];
}
int main() {
    int h[64]; short o[256];
    int *d; short* dout;
    hipMalloc(&d, sizeof(h)); hipMalloc(&dout, sizeof(o));
    for (int mode = 0; mode < 3; ++mode) {
        for (int l = 0; l < 64; ++l) {
            if (mode == 0) h[l] = 8 * l;                                         // lane l: elements 4 l .. 4 l + 3
            if (mode == 1) h[l] = 2 * ((l & 3) * 64 + 4 * ((l & 15) >> 2) + 256 * (l >> 4));   // [4][64] row-major block per group: row = l & 3, cols 4 (i >> 2)
            if (mode == 2) h[l] = 2 * (((l & 15) >> 2) * 64 + 4 * (l & 3) + 256 * (l >> 4));   // row = i >> 2, cols 4 (i & 3)
        }
        hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
        probe<<<1, 64>>>(d, dout);
        hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d addr elem %4d: %4d %4d %4d %4d\n", l, h[l] / 2, o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3]);
    }
    return 0;
}
