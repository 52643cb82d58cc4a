// Relative error of v_rcp_f64 / v_rsq_f64 alone and after one Newton step, against the correctly rounded quotient (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* a, double* r0, double* r1, double* q0, double* q1, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  double x = a[i];
  double r = __builtin_amdgcn_rcp(x); r0[i] = r;
  double e = __builtin_fma(-x, r, 1.0); r1[i] = __builtin_fma(r, e, r);
  double y = __builtin_amdgcn_rsq(x); q0[i] = y;
  double h = 0.5 * x; double e2 = __builtin_fma(-h * y, y, 0.5); q1[i] = __builtin_fma(y, e2, y);
}
int main() {
  const int n = 1 << 20; std::vector<double> a(n); std::mt19937_64 g(1); std::uniform_real_distribution<double> u(-30, 30);
  for (auto& v : a) v = std::exp2(u(g)) * (1.0 + (double)(g() >> 11) / 9007199254740992.0);
  double *da, *d0, *d1, *d2, *d3; hipMalloc(&da, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&d3, n * 8);
  hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(da, d0, d1, d2, d3, n); hipDeviceSynchronize();
  std::vector<double> r0(n), r1(n), q0(n), q1(n);
  hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(q0.data(), d2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(q1.data(), d3, n * 8, hipMemcpyDeviceToHost);
  double m0 = 0, m1 = 0, s0 = 0, s1 = 0;
  for (int i = 0; i < n; i++) {
    long double t = 1.0L / (long double)a[i], ts = 1.0L / sqrtl((long double)a[i]);
    m0 = fmax(m0, (double)fabsl(((long double)r0[i] - t) / t)); m1 = fmax(m1, (double)fabsl(((long double)r1[i] - t) / t));
    s0 = fmax(s0, (double)fabsl(((long double)q0[i] - ts) / ts)); s1 = fmax(s1, (double)fabsl(((long double)q1[i] - ts) / ts));
  }
  printf("v_rcp_f64 max relative error: %.3e (2^%.1f); after one Newton step: %.3e (2^%.1f)\n", m0, log2(m0), m1, log2(m1));
  printf("v_rsq_f64 max relative error: %.3e (2^%.1f); after one Newton step: %.3e (2^%.1f)\n", s0, log2(s0), s1, log2(s1));
  return 0;
}
