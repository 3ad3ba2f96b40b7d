// Mailbox round trips UNDER LOAD: the ping-pong of mail_latency2.hip between two CUs of one XCD while ~240 other CUs stream non-temporal
// stores to HBM (what the team kernel's own output does): agent-scope store/load against plain store + nt load through the XCD's L2.
#include <hip/hip_runtime.h>
#include <cstdio>
using gu64 = __attribute__((address_space(1))) unsigned long long;
template <int MODE> __device__ __forceinline__ unsigned long long rd(gu64* p)
{
    unsigned long long v;
    if constexpr (MODE == 0) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int MODE> __device__ __forceinline__ void wr(gu64* p, unsigned long long v)
{
    if constexpr (MODE == 0) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
__global__ void hog(float4* buf, size_t n4, int iters, unsigned* stop)
{
    extern __shared__ float sm[];
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (int it = 0; it < iters; ++it) {
        for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride)
            { typedef float f4v __attribute__((ext_vector_type(4))); __builtin_nontemporal_store(f4v{1.f, 2.f, 3.f, (float)it}, reinterpret_cast<f4v*>(buf + i)); }
        if (*reinterpret_cast<volatile unsigned*>(stop)) break;
    }
    if (threadIdx.x == 0) sm[0] = 0.f;
}
template <int MODE>
__global__ void pingpong(unsigned long long* w, unsigned long long* out, int a, int b, int rounds)
{
    if (threadIdx.x != 0 || (blockIdx.x != a && blockIdx.x != b)) return;
    gu64* p = (gu64*)w;
    const bool first = blockIdx.x == a;
    const unsigned long long t0 = wall_clock64();
    unsigned long long fails = 0;
    for (unsigned long long i = 1; i <= (unsigned long long)rounds; ++i) {
        const unsigned long long mine = 2 * i - (first ? 1 : 0), want = first ? 2 * i : 2 * i - 1;
        if (first) wr<MODE>(p, mine);
        int spin = 0;
        for (; spin < 20000; ++spin) if (rd<MODE>(p) >= want) break;
        if (spin == 20000) { ++fails; break; }
        if (!first) wr<MODE>(p, mine);
    }
    out[blockIdx.x] = (wall_clock64() - t0) | (fails << 48);
}
int main()
{
    unsigned long long *w, *o; float4* big; unsigned* stop;
    const size_t n4 = (size_t)1 << 26;                                   // 1 GiB
    hipMalloc(&w, 4096); hipMalloc(&o, 256 * 8); hipMalloc(&big, n4 * 16); hipMalloc(&stop, 4);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipFuncSetAttribute(reinterpret_cast<const void*>(hog), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int load = 0; load < 2; ++load)
        for (int mode = 0; mode < 2; ++mode) {
            hipMemset(w, 0, 64); hipMemset(stop, 0, 4); hipDeviceSynchronize();
            if (load) hipLaunchKernelGGL(hog, dim3(240), dim3(1024), 90 * 1024, s1, big, n4, 50, stop);
            // blocks 0 and 8 of a 16-block grid: XCD 0 twice (blockIdx % 8)
            if (mode == 0) hipLaunchKernelGGL((pingpong<0>), dim3(16), dim3(64), 0, s2, w, o, 0, 8, 2000);
            else hipLaunchKernelGGL((pingpong<1>), dim3(16), dim3(64), 0, s2, w, o, 0, 8, 2000);
            hipStreamSynchronize(s2);
            unsigned one = 1; hipMemcpyAsync(stop, &one, 4, hipMemcpyHostToDevice, s2); hipDeviceSynchronize();
            unsigned long long h[16]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
            printf("%-10s %-34s %.2f us per round trip (two hops)%s\n", load ? "under load" : "idle", mode == 0 ? "agent-scope store / load" : "plain store / nt load (XCD's L2)",
                   (h[0] & 0xffffffffffffull) / 2000.0 / 100.0, (h[0] >> 48) ? "  [a wait ran out]" : ""); fflush(stdout);
        }
    return 0;
}
