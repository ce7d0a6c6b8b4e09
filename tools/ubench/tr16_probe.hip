// tr16_probe.hip — what does ds_read_b64_tr_b16 return? (gfx950; build: hipcc --offload-arch=gfx950 -O2 tr16_probe.hip -o tr16_probe)
// Test 1: lane l supplies the address of halves [4l, 4l+4) of an LDS array holding its own indices.
// Test 2: lane l supplies the address of halves [4*perm(l), ...) with perm = reverse inside each 16-lane group,
//         to see whether the DATA a lane receives depends on which lane supplied which address.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short *out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    int l = threadIdx.x;
    int src = mode == 0 ? l : ((l & ~15) | (15 - (l & 15)));
    if (mode == 2) src = l * 3;                    // arbitrary 8-B aligned, non-contiguous addresses
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4 *)(lds + src * 4));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short *d, h[256];
    hipMalloc(&d, sizeof(h));
    for (int mode = 0; mode < 3; ++mode) {
        k<<<1, 64>>>(d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
