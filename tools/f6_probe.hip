// Probe of the gfx950 FP6 (e3m2 "bf6") conversions: rounding, saturation, scale semantics, element order (GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v32f __attribute__((ext_vector_type(32)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
__global__ void k(const float* x, float* y, float sc_in, float sc_out) {
  v16f a, b;
  for (int i = 0; i < 16; ++i) { a[i] = x[i]; b[i] = x[16 + i]; }
  v6u r = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(a, b, sc_in);
  v32f d = __builtin_amdgcn_cvt_scalef32_pk32_f32_bf6(r, sc_out);
  for (int i = 0; i < 32; ++i) y[i] = d[i];
  for (int i = 0; i < 6; ++i) y[32 + i] = __builtin_bit_cast(float, r[i]);
}
int main() {
  float hx[32], *dx, *dy, hy[38];
  const float vals[32] = {0.f, 0.03f, 0.0625f, 0.1f, 0.125f, 0.2f, 0.25f, 0.3f, 0.4f, 0.5f, 0.6f, 0.75f, 0.9f, 1.0f, 1.1f, 1.3f,
                          1.5f, 1.8f, 2.0f, 2.6f, 3.0f, 5.0f, 7.0f, 9.0f, 13.0f, 20.0f, 27.0f, 28.0f, 30.0f, 100.0f, -1.3f, -40.f};
  for (int i = 0; i < 32; ++i) hx[i] = vals[i];
  hipMalloc(&dx, 128); hipMalloc(&dy, 38 * 4); hipMemcpy(dx, hx, 128, hipMemcpyHostToDevice);
  const float scs[3][2] = {{1.f, 1.f}, {4.f, 1.f}, {0.25f, 1.f}};
  for (auto& s : scs) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dy, s[0], s[1]); hipMemcpy(hy, dy, 38 * 4, hipMemcpyDeviceToHost);
    printf("scale_in %g scale_out %g:\n", s[0], s[1]);
    for (int i = 0; i < 32; ++i) printf("  %g->%g", hx[i], hy[i]);
    printf("\n");
  }
  return 0;
}
