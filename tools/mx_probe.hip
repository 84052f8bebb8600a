// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3): which (lane, byte) holds A[i][k] / B[k][j], and what a lane's
// scale byte multiplies.  Prints which layout hypothesis reproduces a host matmul.  (tools/, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void k(const i32x8* a, const i32x8* b, const int* sa, const int* sb, float* c) {
  const int l = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
  for (int r = 0; r < 4; ++r) c[l * 4 + r] = acc[r];
}

static const float kVals[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, -1.f};
static const unsigned char kCodes[8] = {0x00, 0x30, 0x38, 0x3C, 0x40, 0x44, 0x48, 0xB8};   // e4m3fn: bias 7

int main() {
  float A[16][128], B[128][16];
  unsigned char Ac[16][128], Bc[128][16];
  srand(3);
  for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 128; ++kk) { int v = rand() & 7; A[i][kk] = kVals[v]; Ac[i][kk] = kCodes[v]; }
  for (int kk = 0; kk < 128; ++kk) for (int j = 0; j < 16; ++j) { int v = rand() & 7; B[kk][j] = kVals[v]; Bc[kk][j] = kCodes[v]; }
  for (int hyp = 0; hyp < 2; ++hyp) {
    for (int scaled = 0; scaled < 2; ++scaled) {
      unsigned char ha[64][32], hb[64][32]; int sa[64], sb[64];
      for (int l = 0; l < 64; ++l) {
        for (int j = 0; j < 32; ++j) {
          const int kk = hyp == 0 ? 32 * (l >> 4) + j : 16 * (l >> 4) + (j & 15) + 64 * (j >> 4);
          ha[l][j] = Ac[l & 15][kk]; hb[l][j] = Bc[kk][l & 15];
        }
        // scaled: the lanes of k-block 1 carry 2^+1 on A and the lanes of k-block 2 carry 2^-2 on B
        sa[l] = (scaled && (l >> 4) == 1) ? 128 : 127;
        sb[l] = (scaled && (l >> 4) == 2) ? 125 : 127;
      }
      void *da, *db, *dsa, *dsb, *dc;
      hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dc, 1024);
      hipMemcpy(da, ha, 2048, hipMemcpyHostToDevice); hipMemcpy(db, hb, 2048, hipMemcpyHostToDevice);
      hipMemcpy(dsa, sa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb, 256, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (const i32x8*)da, (const i32x8*)db, (const int*)dsa, (const int*)dsb, (float*)dc);
      float C[256]; hipMemcpy(C, dc, 1024, hipMemcpyDeviceToHost);
      double worst = 0;
      for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r, col = l & 15;
        double ref = 0;
        for (int kk = 0; kk < 128; ++kk) {
          const int blk = hyp == 0 ? kk >> 5 : ((kk & 63) >> 4);     // which lane group holds this k under the hypothesis
          double s = 1.0;
          if (scaled && blk == 1) s *= 2.0;
          if (scaled && blk == 2) s *= 0.25;
          ref += s * A[row][kk] * B[kk][col];
        }
        worst = fmax(worst, fabs(ref - C[l * 4 + r]));
      }
      printf("hypothesis %d (%s) scaled=%d: max |C - ref| = %g\n", hyp, hyp == 0 ? "k = 32*(lane>>4) + byte" : "k = 16*(lane>>4) + (byte&15) + 64*(byte>>4)", scaled, worst);
    }
  }
  return 0;
}
