// Accuracy of x ** p = v_exp_f32(p * v_log_f32(x)) (rt_kernels.inl pow_x_p) against double-precision pow, and of the
// device's powf beside it: max relative error over x in (0, 1] where x ** p >= floor, per exponent.
// build: hipcc --offload-arch=gfx950 -O2 -o pow_probe tools/micro/pow_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float *x, float p, float *fast, float *libm, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fast[i] = __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(x[i]));
  libm[i] = powf(x[i], p);
}
int main() {
  const int n = 1 << 22;
  std::vector<float> x(n), f(n), l(n);
  // dense near 1 (where large exponents still give visible values) and uniform over (0, 1]
  for (int i = 0; i < n / 2; ++i) x[i] = (float)(i + 1) / (float)(n / 2);
  for (int i = n / 2; i < n; ++i) x[i] = 1.0f - (float)(i - n / 2) * (0.05f / (float)(n / 2));
  float *dx, *df, *dl;
  hipMalloc(&dx, n * 4); hipMalloc(&df, n * 4); hipMalloc(&dl, n * 4);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  const float ps[] = {0.5f, 2, 3, 8, 10, 32, 50, 100, 128, 500, 1000, 4000, 10000, 100000};
  printf("%10s %14s %14s %14s %14s\n", "p", "fast>=1e-4", "fast>=1e-2", "powf>=1e-4", "fast abs max");
  for (float p : ps) {
    k<<<(n + 255) / 256, 256>>>(dx, p, df, dl, n);
    hipMemcpy(f.data(), df, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(l.data(), dl, n * 4, hipMemcpyDeviceToHost);
    double e4 = 0, e2 = 0, l4 = 0, ab = 0;
    for (int i = 0; i < n; ++i) {
      const double r = pow((double)x[i], (double)p);
      const double ef = fabs(f[i] - r), el = fabs(l[i] - r);
      if (ef > ab) ab = ef;
      if (r >= 1e-4) { if (ef / r > e4) e4 = ef / r; if (el / r > l4) l4 = el / r; }
      if (r >= 1e-2 && ef / r > e2) e2 = ef / r;
    }
    printf("%10g %14.3e %14.3e %14.3e %14.3e\n", p, e4, e2, l4, ab);
  }
  return 0;
}
