// Development: where do the 16 bytes of each lane of global_load_lds_dwordx4 land in LDS on gfx950?
// hipcc --offload-arch=gfx950 -O2 -o lds_dma_layout lds_dma_layout.hip && ./lds_dma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const unsigned* src, unsigned* out)
{
    __shared__ __attribute__((aligned(16))) unsigned buf[512];
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < 512; i += 64) buf[i] = 0xdeadbeefu;
    __syncthreads();
    typedef __attribute__((address_space(1))) const void gptr;
    typedef __attribute__((address_space(3))) void lptr;
    __builtin_amdgcn_global_load_lds((gptr*)(src + 4 * lane), (lptr*)buf, 16, 0, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = buf[i];
}
int main()
{
    unsigned h[256], *d, *o, r[512];
    for (int i = 0; i < 256; ++i) h[i] = i;          // lane l holds words 4l .. 4l+3
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    for (int i = 0; i < 24; ++i) printf("lds[%d] = %u\n", i, r[i]);
    int linear = 1; for (int i = 0; i < 256; ++i) linear &= (r[i] == (unsigned)i);
    printf("lane-linear 16-byte layout: %s\n", linear ? "yes" : "no");
    printf("lds[64] = %u lds[128] = %u lds[255] = %u lds[256] = %x\n", r[64], r[128], r[255], r[256]);
    return 0;
}
