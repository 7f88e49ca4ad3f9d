// Micro-benchmark: VALU wave-instruction issue rate per SIMD vs waves per SIMD, for the
// instruction kinds the traversal loop is made of (fma / mul+add pairs / cmp+cndmask).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void kpk(float *out, int iters, float s) {
  f2 a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
  const f2 s2 = {s, s}, one = {1.0f, 1.0f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], s2, one);
  }
  float r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) a[i] = __builtin_fmaf(a[i], s, 1.0f);
      if (KIND == 1) a[i] = a[i] * s + 1.0f;                 // contract off -> mul, add
      if (KIND == 2) a[i] = (a[i] > s) ? a[i] - 1.0f : a[i] + s;  // cmp, sub, add, cndmask
      if (KIND == 3) a[i] = 1.0f / a[i] + s;               // IEEE division
    }
  }
  float r = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND>
void run(const char *name, int instr_per_elem, float *d, int cus) {
  for (int wps : {1, 2, 4, 8}) {
    int blocks = cus * wps;  // 256 threads = 4 waves = 1 per SIMD
    int iters = 20000 / (KIND == 3 ? 10 : 1);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr_per_simd = (double)iters * 16 * instr_per_elem * wps;
    printf("%-10s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, wps, ms,
           ms * 1e6 / winstr_per_simd, ms * 1e6 / winstr_per_simd * 2.4);
  }
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  int cus = p.multiProcessorCount;
  float *d; hipMalloc(&d, (size_t)cus * 8 * 256 * 4);
  for (int wps : {1, 2, 4, 8}) {
    int blocks = cus * wps, iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kpk, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kpk, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double w = (double)iters * 8 * wps;
    printf("pk_fma     waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", wps, ms, ms * 1e6 / w, ms * 1e6 / w * 2.4);
  }
  run<0>("fma", 1, d, cus);
  run<1>("mul+add", 2, d, cus);
  run<2>("cmp-select", 4, d, cus);
  run<3>("div(~11)", 11, d, cus);
  return 0;
}
