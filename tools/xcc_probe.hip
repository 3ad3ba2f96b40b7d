// Which XCD does workgroup i of a grid land on?  (s_getreg XCC_ID; one 148 KiB-LDS block per CU, as the team kernel)
// build: hipcc --offload-arch=gfx950 -O2 -o build/xcc_probe tools/xcc_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out, unsigned* ctr)
{
    extern __shared__ float sm[];
    if (threadIdx.x == 0) {
        const unsigned x = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID, bits 3:0
        const unsigned cu = __builtin_amdgcn_s_getreg((4) | (8 << 6) | (3 << 11));    // HW_REG_HW_ID cu_id bits 11:8
        const unsigned arr = atomicAdd(ctr + (x & 7), 1u);
        out[blockIdx.x] = x | (arr << 8) | (cu << 16);
        sm[0] = 1.0f;
    }
    __syncthreads();
    // stay resident a little so that all blocks coexist
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) { }
}
int main()
{
    for (int grid : {256, 128, 64, 16}) {
        unsigned *d, *c;
        hipMalloc(&d, grid * 4); hipMalloc(&c, 64); hipMemset(c, 0, 64);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(c, 0, 64);
            hipLaunchKernelGGL(k, dim3(grid), dim3(1024), 148 * 1024, 0, d, c);
            hipDeviceSynchronize();
            std::vector<unsigned> h(grid); unsigned hc[8];
            hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost); hipMemcpy(hc, c, 32, hipMemcpyDeviceToHost);
            int mism = 0;
            for (int i = 0; i < grid; ++i) if ((h[i] & 0xff) != unsigned(i % 8)) ++mism;
            printf("grid %3d rep %d: blocks per XCD %u %u %u %u %u %u %u %u ; blockIdx %% 8 != XCC_ID for %d blocks\n", grid, rep, hc[0], hc[1], hc[2], hc[3], hc[4], hc[5], hc[6], hc[7], mism);
        }
        hipFree(d); hipFree(c);
    }
    return 0;
}
