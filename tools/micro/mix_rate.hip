// Micro-benchmark: issue cost of the instruction mixes the traversal is made of, per SIMD, against
// waves per SIMD.  Each kernel runs ITERS x a straight-line block on register operands (no memory);
// the static instruction count of the block comes from the disassembly (tools/micro/README).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/mix_rate.hip -o gpurun_out/mix_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// KIND 0: box test as in wide_boxes<MODE 0> (sub, mul, max, min3, cross gap, doubt band)
// KIND 1: cmp + cndmask pairs (each select depends on its own compare)
// KIND 2: max3 / min3 chains
// KIND 3: compare-exchange network of wide_step (5 x cmp + 4 cndmask)
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s, float t) {
  float a[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) a[i] = threadIdx.x * 0.001f + i * s;
  float acc = 0;
  int iacc = 0;
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float ox = a[0] + c, oy = a[1], oz = a[2], rx = a[3], ry = a[4], rz = a[5];
        const float tx0 = (a[6] - ox) * rx, tx1 = (a[7] - ox) * rx;
        const float ty0 = (a[8] - oy) * ry, ty1 = (a[9] - oy) * ry;
        const float tz0 = (a[10] - oz) * rz, tz1 = (a[11] - oz) * rz;
        const float mx = fmaxf(ty0, tz0), my = fmaxf(tx0, tz0), mz = fmaxf(tx0, ty0);
        const float E = fmaxf(tx0, mx), X = fminf(fminf(tx1, ty1), tz1);
        const float g = fminf(fminf(tx1 - mx, ty1 - my), tz1 - mz);
        const bool h = g >= 0, doubt = fabsf(g) <= 4.76837158203125e-07f * (fabsf(E) + fabsf(X));
        acc += h ? E : X;
        iacc += doubt;
      }
      a[0] += acc * 1e-30f;
    } else if (KIND == 1) {
#pragma unroll
      for (int i = 0; i < 12; ++i) a[i] = (a[i] > t) ? a[(i + 1) % 12] : a[(i + 5) % 12];
    } else if (KIND == 2) {
#pragma unroll
      for (int i = 0; i < 12; ++i) a[i] = fmaxf(fmaxf(a[i], a[(i + 1) % 12]), t) + fminf(fminf(a[(i + 2) % 12], s), a[(i + 3) % 12]);
    } else {
      float k0 = a[0], k1 = a[1], k2 = a[2], k3 = a[3];
      int r0 = iacc, r1 = iacc + 1, r2 = iacc + 2, r3 = iacc + 3;
#define CS(ka, ra, kb, rb) { const bool sw = kb < ka; const float x0 = sw ? kb : ka, x1 = sw ? ka : kb; const int y0 = sw ? rb : ra, y1 = sw ? ra : rb; ka = x0; kb = x1; ra = y0; rb = y1; }
      CS(k0, r0, k1, r1) CS(k2, r2, k3, r3) CS(k0, r0, k2, r2) CS(k1, r1, k3, r3) CS(k1, r1, k2, r2)
#undef CS
      a[0] = k3 + s; a[1] = k0 - s; a[2] = k1 * t; a[3] = k2;
      iacc = r0 ^ r3;
    }
  }
  float r = acc + iacc;
#pragma unroll
  for (int i = 0; i < 12; ++i) r += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
void run(const char *name, float *d, int cus) {
  for (int wps : {1, 2, 4, 5, 8}) {
    int blocks = cus * wps, iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f, 0.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // cycles per loop iteration per SIMD and wave
    printf("%-12s waves/SIMD %d: %.3f ms -> %.1f cycles per iteration per wave, %.1f per iteration per SIMD-share\n", name, wps, ms,
           ms * 1e-3 * 2.4e9 / iters, ms * 1e-3 * 2.4e9 / iters / wps);
  }
}
int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  int cus = p.multiProcessorCount;
  float *d; (void)hipMalloc(&d, (size_t)cus * 8 * 256 * 4);
  run<0>("box4", d, cus);
  run<1>("cmp-sel12", d, cus);
  run<2>("max3min3x12", d, cus);
  run<3>("sort4", d, cus);
  return 0;
}
