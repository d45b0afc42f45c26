#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* gbl_vp;
__global__ void k(const uint4* __restrict__ src, uint4* __restrict__ dst, int n) {
    __shared__ __attribute__((aligned(16))) uint4 buf[512];
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    for (int k2 = 0; k2 < 2; ++k2) {
        const int i = tid + 256 * k2;
        const bool ok = (i % 5) != 0;                                   // some lanes masked: they zero their own cell
        uint4* cell0 = buf + 256 * k2 + wv * 64;                          // wave-uniform base
        if (ok) __builtin_amdgcn_global_load_lds((gbl_vp)(src + i), (lds_vp)cell0, 16, 0, 0);
        else cell0[lane] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    for (int k2 = 0; k2 < 2; ++k2) dst[tid + 256 * k2] = buf[tid + 256 * k2];
}
int main() {
    uint4 *s, *d; hipMalloc(&s, 512 * 16); hipMalloc(&d, 512 * 16);
    uint4 h[512]; for (int i = 0; i < 512; ++i) h[i] = make_uint4(i, i * 2, i * 3, i * 4);
    hipMemcpy(s, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, s, d, 512);
    uint4 o[512]; hipMemcpy(o, d, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 512; ++i) { uint4 e = (i % 5) ? h[i] : make_uint4(0, 0, 0, 0); if (o[i].x != e.x || o[i].w != e.w) { if (bad < 5) printf("mismatch %d: %u %u\n", i, o[i].x, e.x); ++bad; } }
    printf("glds test: %d mismatches\n", bad);
    return bad != 0;
}
