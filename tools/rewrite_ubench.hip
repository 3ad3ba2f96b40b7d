// Can a CU re-read what it wrote a few microseconds ago from cache instead of HBM?  Each block (1024 threads, one per CU)
// owns NSIG "signals" of 88000 floats: phase A writes them (plain or streaming stores), block barrier, phase B reads
// them back, scales and rewrites them in place.  Compared with the same work as two kernels (A for all, then B for all).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kSig4 = 22000;                             // float4 per signal (2000 x 44 floats)

template <int MODE, bool NT>                             // MODE 0: A then B per signal (fused); 1: A only; 2: B only
__global__ __launch_bounds__(1024) void k(f4* buf, int nsig_total, float seed)
{
    for (int sig = blockIdx.x; sig < nsig_total; sig += gridDim.x) {
        f4* p = buf + static_cast<size_t>(sig) * kSig4;
        if (MODE != 2) {
            for (int i = threadIdx.x; i < kSig4; i += 1024) {
                f4 v = {seed + i, seed - i, seed * i, seed};
                if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
            }
        }
        if (MODE == 0) __syncthreads();
        if (MODE != 1) {
            for (int i = threadIdx.x; i < kSig4; i += 1024) {
                f4 v = p[i];
                v = (v - 1.5f) * 0.25f;
                p[i] = v;
            }
        }
        if (MODE == 0) __syncthreads();
    }
}
template <int MODE, bool NT> float run(f4* d, int nsig)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(1024), 0, 0, d, nsig, 1.0f);
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(1024), 0, 0, d, nsig, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20;
}
int main()
{
    const int nsig = 1024;
    f4* d; hipMalloc(&d, static_cast<size_t>(nsig) * kSig4 * sizeof(f4));
    printf("1024 signals x 352 KB = 360 MB; times per pass over all signals\n");
    printf("plain stores : A only %.3f ms | B only %.3f ms | fused A,barrier,B per signal %.3f ms\n", run<1, false>(d, nsig), run<2, false>(d, nsig), run<0, false>(d, nsig));
    printf("nt stores in A: A only %.3f ms | B only %.3f ms | fused A,barrier,B per signal %.3f ms\n", run<1, true>(d, nsig), run<2, true>(d, nsig), run<0, true>(d, nsig));
    return 0;
}
