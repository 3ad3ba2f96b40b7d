// Which load flavours see another CU's store through the XCD's L2 (same XCD) -- and how fast?  Writer: plain / sc0 / sc1 store; reader variants below.
#include <hip/hip_runtime.h>
#include <cstdio>
using gu64 = __attribute__((address_space(1))) unsigned long long;
template <int RD> __device__ __forceinline__ unsigned long long rd(gu64* p)
{
    unsigned long long v;
    if constexpr (RD == 0) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (RD == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (RD == 2) asm volatile("buffer_inv sc0\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (RD == 3) asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (RD == 4) asm volatile("global_load_dwordx2 %0, %1, off sc0 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (RD == 5) asm volatile("buffer_inv sc1\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int WR> __device__ __forceinline__ void wr(gu64* p, unsigned long long v)
{
    if constexpr (WR == 0) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if constexpr (WR == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    if constexpr (WR == 2) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
template <int RD, int WR>
__global__ void pingpong(unsigned long long* w, unsigned long long* out, int a, int b)
{
    if (threadIdx.x != 0 || (blockIdx.x != a && blockIdx.x != b)) return;
    gu64* p = (gu64*)w;
    const bool first = blockIdx.x == a;
    const unsigned long long t0 = wall_clock64();
    unsigned long long fails = 0;
    for (unsigned long long i = 1; i <= 100; ++i) {
        const unsigned long long mine = 2 * i - (first ? 1 : 0), want = first ? 2 * i : 2 * i - 1;
        if (first) wr<WR>(p, mine);
        int spin = 0;
        for (; spin < 3000; ++spin) if (rd<RD>(p) >= want) break;
        if (spin == 3000) { ++fails; for (int k = 0; k < 3000 && rd<0>(p) < want; ++k) { } }
        if (!first) wr<WR>(p, mine);
    }
    out[blockIdx.x] = (wall_clock64() - t0) | (fails << 48);
}
template <int RD, int WR> void run(unsigned long long* w, unsigned long long* o, const char* name)
{
    for (int pair = 0; pair < 2; ++pair) {
        const int a = 0, b = pair == 0 ? 8 : 1;
        hipMemset(w, 0, 64);
        hipLaunchKernelGGL((pingpong<RD, WR>), dim3(256), dim3(64), 0, 0, w, o, a, b);
        hipDeviceSynchronize();
        unsigned long long h[256]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-46s blocks %d <-> %d (%s): %.2f us per round trip, %llu of 100 never seen by this reader\n", name, a, b, pair == 0 ? "same XCD" : "other XCD",
               (h[a] & 0xffffffffffffull) / 100.0 / 100.0, h[a] >> 48); fflush(stdout);
    }
}
int main()
{
    unsigned long long *w, *o;
    hipMalloc(&w, 4096); hipMalloc(&o, 256 * 8);
    run<0, 0>(w, o, "store sc1 / load sc1 (agent)");
    run<1, 2>(w, o, "store plain / load sc0");
    run<1, 1>(w, o, "store sc0 / load sc0");
    run<2, 2>(w, o, "store plain / buffer_inv sc0 + load");
    run<3, 2>(w, o, "store plain / load nt");
    run<4, 2>(w, o, "store plain / load sc0 nt");
    run<5, 2>(w, o, "store plain / buffer_inv sc1 + load");
    run<0, 2>(w, o, "store plain / load sc1");
    return 0;
}
