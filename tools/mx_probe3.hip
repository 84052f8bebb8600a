// Probes for the f16 + fp8-cross-term scheme (tools/, not part of the library):
//  (1) v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 x e4m3): operand lane -> k map, constant E8M0 scales, C layout;
//  (2) ds_read_b64_tr_b8: which LDS bytes each lane receives;
//  (3) v_cvt_pk_fp8_f32 / v_cvt_scalef32_pk_fp8_f32: rounding and saturation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void k_mfma(const i32x8* a, const i32x8* b, int sa, int sb, float* c) {
  const int l = threadIdx.x;
  f32x16 acc = {};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa, 0, sb);
  for (int r = 0; r < 16; ++r) c[l * 16 + r] = acc[r];
}

__global__ void k_tr8(const unsigned char* img, int pitch, const int* lane_off, unsigned char* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = img[i];
  __syncthreads();
  const i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)(lds + lane_off[threadIdx.x]));
  memcpy(out + threadIdx.x * 8, &v, 8);
}

__global__ void k_cvt(const float* x, int n, unsigned char* out, unsigned char* out_scaled, float scale) {
  const int i = threadIdx.x + blockIdx.x * blockDim.x;
  if (2 * i + 1 >= n + 1) return;
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false);
  out[2 * i] = r & 0xFF; out[2 * i + 1] = (r >> 8) & 0xFF;
  typedef __attribute__((ext_vector_type(2))) short s16x2;
  s16x2 z = {0, 0};
  s16x2 q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, x[2 * i], x[2 * i + 1], scale, false);
  out_scaled[2 * i] = q[0] & 0xFF; out_scaled[2 * i + 1] = (q[0] >> 8) & 0xFF;
}

static float e4m3_to_f(unsigned char c) {
  const int s = c >> 7, e = (c >> 3) & 15, m = c & 7;
  float v;
  if (e == 0) v = ldexpf((float)m, -9);
  else if (e == 15 && m == 7) v = NAN;
  else v = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -v : v;
}
static unsigned char f_to_e4m3_rne_sat(float x) {   // reference: round to nearest even, saturate to 448
  unsigned char best = 0; float bd = INFINITY;
  const float ax = fminf(fabsf(x), 448.0f);
  for (int c = 0; c < 0x7F; ++c) {
    const float v = e4m3_to_f((unsigned char)c);
    const float d = fabsf(v - ax);
    if (d < bd || (d == bd && !(c & 1))) { bd = d; best = (unsigned char)c; }
  }
  return best | (x < 0 || (x == 0 && std::signbit(x)) ? 0x80 : 0);
}

int main() {
  // ---------------- (1) MFMA layout
  static float A[32][64], B[64][32];
  static unsigned char Ac[32][64], Bc[64][32];
  srand(5);
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) { unsigned char c = (rand() % 0x60) + 0x10; if (rand() & 1) c |= 0x80; Ac[i][k] = c; A[i][k] = e4m3_to_f(c); }
  for (int k = 0; k < 64; ++k) for (int j = 0; j < 32; ++j) { unsigned char c = (rand() % 0x60) + 0x10; if (rand() & 1) c |= 0x80; Bc[k][j] = c; B[k][j] = e4m3_to_f(c); }
  void *da, *db, *dc; hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dc, 4096);
  for (int hyp = 0; hyp < 3; ++hyp) {
    unsigned char ha[64][32], hb[64][32];
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
      int kk;
      if (hyp == 0) kk = 32 * (l >> 5) + j;                       // natural
      else if (hyp == 1) kk = 16 * (l >> 5) + (j & 15) + 32 * (j >> 4);   // alternative interleave
      else kk = (j * 2 + (l >> 5)) ;                                // arbitrary consistent permutation: lane half takes odd / even k
      ha[l][j] = Ac[l & 31][kk]; hb[l][j] = Bc[kk][l & 31];
    }
    hipMemcpy(da, ha, 2048, hipMemcpyHostToDevice); hipMemcpy(db, hb, 2048, hipMemcpyHostToDevice);
    for (int sc = 0; sc < 2; ++sc) {
      const int sa = sc ? 127 - 11 : 127, sb = sc ? 127 + 3 : 127;
      hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, (const i32x8*)da, (const i32x8*)db, sa, sb, (float*)dc);
      static float C[64 * 16]; hipMemcpy(C, dc, 4096, hipMemcpyDeviceToHost);
      double worst = 0, big = 0;
      for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
        const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        double ref = 0;
        for (int k = 0; k < 64; ++k) ref += (double)A[row][k] * B[k][col];
        ref *= sc ? ldexp(1.0, -8) : 1.0;
        worst = fmax(worst, fabs(ref - C[l * 16 + r])); big = fmax(big, fabs(ref));
      }
      printf("mfma 32x32x64 hyp %d scale(%d,%d): max |C - ref| = %g (max |ref| %g)\n", hyp, sa, sb, worst, big);
    }
  }
  // ---------------- (2) ds_read_b64_tr_b8: image byte value = its own offset (mod 251 to tell rows apart), several address patterns
  {
    unsigned char img[4096]; for (int i = 0; i < 4096; ++i) img[i] = (unsigned char)(i % 251);
    void *dimg, *doff, *dout; hipMalloc(&dimg, 4096); hipMalloc(&doff, 256); hipMalloc(&dout, 512);
    hipMemcpy(dimg, img, 4096, hipMemcpyHostToDevice);
    const int pitch = 64;
    for (int pat = 0; pat < 2; ++pat) {
      int off[64];
      for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, i = l & 15;
        // pattern 0: lane 2q+p of the group -> row q, bytes 8p..8p+7 of a block of 8 rows x 16 bytes; groups take columns 16 g
        // pattern 1: lane 8p+q -> row q, bytes 8p..
        const int q = pat == 0 ? i >> 1 : i & 7, p = pat == 0 ? i & 1 : i >> 3;
        off[l] = q * pitch + 16 * g + 8 * p;
      }
      hipMemcpy(doff, off, 256, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k_tr8, dim3(1), dim3(64), 0, 0, (const unsigned char*)dimg, pitch, (const int*)doff, (unsigned char*)dout);
      unsigned char out[512]; hipMemcpy(out, dout, 512, hipMemcpyDeviceToHost);
      printf("tr8 pattern %d (lane: bytes as LDS offsets = row*64+col):\n", pat);
      for (int l = 0; l < 64; l += (l < 18 ? 1 : 15)) {
        printf("  lane %2d:", l);
        for (int b = 0; b < 8; ++b) {
          // find offset with that value among the block (row<8, col<64)
          int found = -1;
          for (int r = 0; r < 8 && found < 0; ++r) for (int c = 0; c < 64; ++c) if ((r * pitch + c) % 251 == out[l * 8 + b]) { found = r * 100 + c; break; }
          printf(" r%dc%02d", found / 100, found % 100);
        }
        printf("\n");
      }
    }
  }
  // ---------------- (3) cvt
  {
    const int n = 4096; static float x[4096]; static unsigned char got[4096], gots[4096];
    for (int i = 0; i < n; ++i) { const float m = (rand() / (float)RAND_MAX) * 2 - 1; x[i] = ldexpf(m, (rand() % 24) - 12); }
    x[0] = 448; x[1] = 449; x[2] = 464; x[3] = 480; x[4] = 1000; x[5] = -1e6; x[6] = 0.0009765625f; x[7] = 0.001953125f; x[8] = 0.0029296875f; x[9] = 1e-9f; x[10] = -0.f; x[11]=INFINITY;
    void *dx, *dg, *dgs; hipMalloc(&dx, n * 4); hipMalloc(&dg, n); hipMalloc(&dgs, n);
    hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_cvt, dim3(n / 2 / 64), dim3(64), 0, 0, (const float*)dx, n, (unsigned char*)dg, (unsigned char*)dgs, 0.25f);
    hipMemcpy(got, dg, n, hipMemcpyDeviceToHost); hipMemcpy(gots, dgs, n, hipMemcpyDeviceToHost);
    int bad = 0, bads_div = 0, bads_mul = 0;
    for (int i = 0; i < n; ++i) {
      const unsigned char want = f_to_e4m3_rne_sat(x[i]);
      if (got[i] != want && !(i == 11)) { if (bad < 8) printf("cvt_pk_fp8_f32(%g) = 0x%02x (%g), RNE-sat reference 0x%02x (%g)\n", x[i], got[i], e4m3_to_f(got[i]), want, e4m3_to_f(want)); ++bad; }
      if (gots[i] != f_to_e4m3_rne_sat(x[i] / 0.25f)) ++bads_div;
      if (gots[i] != f_to_e4m3_rne_sat(x[i] * 0.25f)) ++bads_mul;
    }
    printf("cvt_pk_fp8_f32: %d of %d differ from RNE + saturate-to-448; specials: 448->%02x 449->%02x 464->%02x 480->%02x 1000->%02x -1e6->%02x inf->%02x 2^-10->%02x 2^-9->%02x 3*2^-10->%02x\n", bad, n, got[0], got[1], got[2], got[3], got[4], got[5], got[11], got[6], got[7], got[8]);
    printf("cvt_scalef32_pk_fp8_f32(scale 0.25): mismatches if it DIVIDES by scale: %d, if it MULTIPLIES: %d\n", bads_div, bads_mul);
  }
  return 0;
}
