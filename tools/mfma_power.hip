// Power / rate microbenchmark: which bf16 MFMA shape is cheaper per flop on gfx950?  (timing-only, tools/gpu_mfma_power.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;

template <int SHAPE, int FMT = 0>
__global__ __launch_bounds__(256, 2) void k(const bf16x8* in, float* out, int iters) {
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x + 256 * i]; b[i] = in[threadIdx.x + 256 * (4 + i)]; }
  if constexpr (SHAPE == 16) {
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else if constexpr (SHAPE == 8) {
    // the same 32 accumulator tiles, K = 128 per round: 4 bf16 products (hi x hi) + 2 block-scaled fp8 products (the
    // cross terms) instead of 12 bf16 products
    i32x8 qa[2], qb[2];
    for (int i = 0; i < 2; ++i) { qa[i] = ((const i32x8*)in)[512 + threadIdx.x + 256 * i]; qb[i] = ((const i32x8*)in)[512 + ((threadIdx.x + 64) & 255) + 256 * i]; }
    const int sa = 0x7f7f7f7f, sb = 0x7f7f7f7f;   // E8M0 1.0
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters / 4; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + r) & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(qa[i & 1], qb[(i >> 1) & 1], acc[i], FMT, FMT, 0, sa, 0, sb);
        acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(qa[(i + 1) & 1], qb[(i >> 2) & 1], acc[i], FMT, FMT, 0, sa, 0, sb);
      }
    }
    float s = 0;
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else if constexpr (SHAPE == 88) {
    // the 32x32 form of the same work (what gemm_f8_kernel issues): 8 accumulator tiles of 32 x 32, per K = 128 eight v_mfma_f32_32x32x16 (fp16 plane) and four
    // block-scaled v_mfma_scale_f32_32x32x64 (the two cross terms) -- the same 4096 matrix-pipe cycles per round as SHAPE 8's 16x16 instructions
    i32x8 qa[2], qb[2];
    for (int i = 0; i < 2; ++i) { qa[i] = ((const i32x8*)in)[512 + threadIdx.x + 256 * i]; qb[i] = ((const i32x8*)in)[512 + ((threadIdx.x + 64) & 255) + 256 * i]; }
    const int sa = 0x7f7f7f7f, sb = 0x7f7f7f7f;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    for (int it = 0; it < iters / 4; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + r) & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa[(i + r) & 1], qb[(i >> (r & 1)) & 1], acc[i], 0, 0, 0, sa, 0, sb);
      }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else if constexpr (SHAPE == 3) {
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters / 4; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i)
#pragma unroll
        for (int r = 0; r < 12; ++r) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + r) & 3], b[((i >> 2) + r) & 3], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + r) & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}

int main(int argc, char** argv) {
  const int shape = atoi(argv[1]); const double secs = atof(argv[2]);
  bf16x8* in; float* out;
  hipMalloc(&in, 2048 * 16); hipMalloc(&out, 512 * 256 * 4);
  unsigned short* h = (unsigned short*)malloc(2048 * 16);
  srand(1);
  for (int i = 0; i < 2048 * 8; ++i) h[i] = (unsigned short)(0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15));   // random mantissas, |x| in [0.5, 1)
  if (shape == 8 || shape == 88 || shape == 6 || shape == 4) for (int i = 1024 * 8; i < 2048 * 8; ++i) h[i] = (unsigned short)(rand() & 0x7777) | (unsigned short)(rand() & 0x8080);   // fp8 bytes, no NaN
  hipMemcpy(in, h, 2048 * 16, hipMemcpyHostToDevice);
  const int iters = 20000;   // x 32 MFMAs (16x16x32) or 16 (32x32x16): same flops per iteration
  auto t0 = std::chrono::steady_clock::now(); long n = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    for (int r = 0; r < 10; ++r) {
      if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(512), dim3(256), 0, 0, in, out, iters);
      else if (shape == 8) hipLaunchKernelGGL(k<8>, dim3(512), dim3(256), 0, 0, in, out, iters);
      else if (shape == 88) hipLaunchKernelGGL(k<88>, dim3(512), dim3(256), 0, 0, in, out, iters);
      else if (shape == 6) hipLaunchKernelGGL((k<8, 3>), dim3(512), dim3(256), 0, 0, in, out, iters);   // cross terms in bf6 (e3m2)
      else if (shape == 4) hipLaunchKernelGGL((k<8, 4>), dim3(512), dim3(256), 0, 0, in, out, iters);   // cross terms in fp4 (e2m1)
      else if (shape == 3) hipLaunchKernelGGL(k<3>, dim3(512), dim3(256), 0, 0, in, out, iters);
      else hipLaunchKernelGGL(k<32>, dim3(512), dim3(256), 0, 0, in, out, iters);
    }
    hipDeviceSynchronize(); n += 10;
  }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const double flops = (double)n * 512 * 4 * iters * 32 * 16384.0;
  if (shape == 8 || shape == 88 || shape == 3 || shape == 6 || shape == 4) printf("shape %d: %.3f us per (32 tiles x K=128) round\n", shape, dt / ((double)n * (iters / 4)) * 1e6);
  else printf("shape %d: %.1f TFLOP/s\n", shape, flops / dt / 1e12);
  return 0;
}
