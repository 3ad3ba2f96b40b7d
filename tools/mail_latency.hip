// Round-trip time of one look at a mailbox word: agent-scope atomic load (served from the memory side) against an atomic OR of 0 that
// executes in the XCD's L2 (workgroup scope), and a publish -> seen latency between two CUs of one XCD / of different XCDs.
// build: hipcc --offload-arch=gfx950 -O2 -o build/mail_latency tools/mail_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using gu64 = __attribute__((address_space(1))) unsigned long long;
// an atomic OR of 0 that RETURNS the word, executed in the XCD's L2 (no sc1: not agent scope; from inline assembly: the compiler turns
// an idempotent read-modify-write at workgroup scope into a plain load, which hits the CU's L1 for ever)
__device__ __forceinline__ unsigned long long l2_look(gu64* p)
{
    unsigned long long v, z = 0;
    asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(z) : "memory");
    return v;
}
__device__ __forceinline__ void l2_put(gu64* p, unsigned long long v)
{
    asm volatile("global_atomic_swap_x2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
__global__ void look(unsigned long long* w, unsigned long long* out, int mode)
{
    if (threadIdx.x != 0) return;
    gu64* p = (gu64*)(w + 64 * blockIdx.x);
    unsigned long long acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < 200; ++i) {
        unsigned long long v;
        if (mode == 0) v = __hip_atomic_load(p + (acc & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (mode == 1) v = l2_look(p + (acc & 1));
        else if (mode == 2) v = __hip_atomic_fetch_or(p + (acc & 1), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else v = __hip_atomic_load(p + (acc & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        acc += v;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x] = (t1 - t0) + (acc & 1);
}
// ping-pong between block a (writer first) and block b: 100 round trips
__global__ void pingpong(unsigned long long* w, unsigned long long* out, int a, int b, int mode)
{
    if (threadIdx.x != 0 || (blockIdx.x != a && blockIdx.x != b)) return;
    gu64* p = (gu64*)w;
    const bool first = blockIdx.x == a;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (unsigned long long i = 1; i <= 100; ++i) {
        const unsigned long long mine = 2 * i - (first ? 1 : 0), want = first ? 2 * i : 2 * i - 1;
        if (first) {
            if (mode == 0) __hip_atomic_store(p, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else l2_put(p, mine);
        }
        for (int spin = 0; spin < 20000; ++spin) {
            unsigned long long v;
            if (mode == 0) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v = l2_look(p);
            if (v >= want) break;
        }
        if (!first) {
            if (mode == 0) __hip_atomic_store(p, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else l2_put(p, mine);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x] = t1 - t0;
}
int main()
{
    unsigned long long *w, *o;
    hipMalloc(&w, 256 * 64 * 8); hipMalloc(&o, 256 * 8); hipMemset(w, 0, 256 * 64 * 8);
    const char* names[] = {"atomic load, agent scope", "atomic or 0, workgroup scope (L2)", "atomic or 0, agent scope", "atomic load, workgroup scope"};
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(look, dim3(256), dim3(64), 0, 0, w, o, mode);
        hipDeviceSynchronize();
        unsigned long long h[256]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
        printf("%-40s %.0f shader cycles per look (256 CUs looking at once)\n", names[mode], s / 256 / 200);
    }
    for (int mode = 0; mode < 2; ++mode)
        for (int pair = 0; pair < 3; ++pair) {
            const int a = 0, b = pair == 0 ? 8 : pair == 1 ? 1 : 129;      // same XCD (0 and 8), neighbouring XCDs, far
            hipMemset(w, 0, 64);
            hipLaunchKernelGGL(pingpong, dim3(256), dim3(64), 0, 0, w, o, a, b, mode);
            hipError_t e = hipDeviceSynchronize();
            unsigned long long h[256]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
            printf("ping-pong blocks %d <-> %d, %s: %.0f cycles per round trip (two hops) %s\n", a, b, mode == 0 ? "agent-scope store / load" : "L2 exchange / or (workgroup scope)", h[a] / 100.0, e == hipSuccess ? "" : hipGetErrorString(e));
        }
    return 0;
}
