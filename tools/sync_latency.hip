// Development: what the host pays to learn that a ~13 us kernel is done -- hipStreamSynchronize vs a hipStreamQuery spin vs a flag the kernel
// stores to pinned host memory itself.  hipcc --offload-arch=gfx950 -O2 tools/sync_latency.hip -o /tmp/sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void spin_kernel(unsigned ticks, volatile unsigned* flag, unsigned v)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (flag && threadIdx.x == 0 && blockIdx.x == 0) { __threadfence_system(); *flag = v; }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t st; hipStreamCreate(&st);
    unsigned* hflag; hipHostMalloc(reinterpret_cast<void**>(&hflag), 64, hipHostMallocMapped); *hflag = 0;
    unsigned* dflag; hipHostGetDevicePointer(reinterpret_cast<void**>(&dflag), hflag, 0);
    const unsigned ticks = 1000;   // 10 us at 100 MHz
    for (int mode = 0; mode < 4; ++mode) {
        std::vector<double> t;
        for (int i = 0; i < 2200; ++i) {
            const double t0 = now();
            if (mode == 3) {
                hipLaunchKernelGGL(spin_kernel, dim3(16), dim3(1024), 0, st, ticks, dflag, static_cast<unsigned>(i + 1));
                while (*reinterpret_cast<volatile unsigned*>(hflag) != static_cast<unsigned>(i + 1)) {}
            } else {
                hipLaunchKernelGGL(spin_kernel, dim3(16), dim3(1024), 0, st, ticks, static_cast<volatile unsigned*>(nullptr), 0u);
                if (mode == 0) hipStreamSynchronize(st);
                else if (mode == 1) { while (hipStreamQuery(st) == hipErrorNotReady) {} }
                else { hipDeviceSynchronize(); }
            }
            const double t1 = now();
            if (i >= 200) t.push_back(t1 - t0);
            if (mode == 3) hipStreamSynchronize(st);
        }
        std::sort(t.begin(), t.end());
        const char* names[] = {"hipStreamSynchronize", "hipStreamQuery spin", "hipDeviceSynchronize", "flag in pinned host memory stored by the kernel"};
        std::printf("%-50s launch + 10 us kernel + completion: median %.1f us, p10 %.1f, p90 %.1f\n", names[mode], t[t.size() / 2], t[t.size() / 10], t[t.size() * 9 / 10]);
    }
    return 0;
}
