// Probe 2: what does ONE lane's scale byte multiply in v_mfma_scale_f32_16x16x128_f8f6f4?  (tools/, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const i32x8* a, const i32x8* b, const int* sa, const int* sb, float* c) {
  const int l = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
  for (int r = 0; r < 4; ++r) c[l * 4 + r] = acc[r];
}
int main() {
  unsigned char ha[64][32], hb[64][32]; int sa[64], sb[64]; float C[256];
  void *da, *db, *dsa, *dsb, *dc;
  hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dc, 1024);
  auto run = [&]() {
    hipMemcpy(da, ha, 2048, hipMemcpyHostToDevice); hipMemcpy(db, hb, 2048, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (const i32x8*)da, (const i32x8*)db, (const int*)dsa, (const int*)dsb, (float*)dc);
    hipMemcpy(C, dc, 1024, hipMemcpyDeviceToHost);
  };
  const int probes[] = {0, 5, 16, 21, 32, 48, 63};
  for (int p : probes) {
    // A = 1 everywhere, B nonzero only in the bytes of ONE lane group g (lanes 16g .. 16g+15): which g sees lane p's A scale?
    for (int g = 0; g < 4; ++g) {
      memset(ha, 0x38, sizeof ha);
      for (int l = 0; l < 64; ++l) memset(hb[l], (l >> 4) == g ? 0x38 : 0x00, 32);
      for (int l = 0; l < 64; ++l) { sa[l] = 127; sb[l] = 127; }
      sa[p] = 128;   // x2 on lane p of the A scale register (byte 0; opsel 0)
      run();
      printf("A-scale x2 on lane %2d, B live in lane group %d: column 0 of C by row:", p, g);
      for (int row = 0; row < 16; ++row) printf(" %g", C[((row >> 2) * 16 + 0) * 4 + (row & 3)]);
      printf("\n");
    }
  }
  // byte selection: put x2 in byte 1 of every lane's A scale, opsel still 0 -> no effect expected
  memset(ha, 0x38, sizeof ha); memset(hb, 0x38, sizeof hb);
  for (int l = 0; l < 64; ++l) { sa[l] = 127 | (128 << 8); sb[l] = 127; }
  run();
  printf("x2 in byte 1 of all A scales, opsel 0: C[0][0] = %g (128 = byte 0 is used)\n", C[0]);
  return 0;
}
