// ds_read_b64_tr_b16 semantics probe: every lane supplies the LDS address of 4 contiguous u16; LDS holds u16 = index.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short short4v __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int idx;
    if (mode == 0) idx = l * 4;                       // lane l -> elements 4l .. 4l+3
    else idx = (l & 15) * 64 + (l >> 4) * 4;           // lane (l&15) -> row l&15 of a [16][64] matrix, cols 4*(l>>4)
    short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(&lds[idx]));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
