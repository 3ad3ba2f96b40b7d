// HBM streaming microbenchmark: float4 copy, in-place scale, write-only (360 MB like the C2 feature batch)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void inplace4(float4* a, size_t n, float m, float s) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = a[i]; v.x=(v.x-m)*s; v.y=(v.y-m)*s; v.z=(v.z-m)*s; v.w=(v.w-m)*s; a[i]=v; }
}
__global__ void write4(float4* a, size_t n, float m) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = make_float4(m,m,m,m);
}
int main() {
  const size_t bytes = 1024ull * 2000 * 44 * 4, n4 = bytes / 16;
  float4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {2048, 8192, 32768}) {
    for (int kind = 0; kind < 3; ++kind) {
      float best = 1e9;
      for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) copy4<<<grid, 256>>>(a, b, n4); else if (kind == 1) inplace4<<<grid, 256>>>(a, n4, 0.1f, 1.0f); else write4<<<grid, 256>>>(a, n4, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
      }
      const double moved = (kind == 2 ? 1.0 : 2.0) * bytes;
      printf("grid %6d %-8s %.4f ms  %.2f TB/s (bytes moved %.0f MB)\n", grid, kind == 0 ? "copy" : kind == 1 ? "inplace" : "write", best, moved / best / 1e9, moved / 1e6);
    }
  }
  return 0;
}
