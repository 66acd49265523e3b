// Micro-benchmark for an idea NOT (yet) in the product: the IMDCT / matrixing GEMMs on v_mfma_f32_16x16x32_f16 with
// operands split into fp16 hi + lo (three products: hi*hi, hi*lo, lo*hi) instead of v_mfma_f32_16x16x4_f32.
//   1. are fp16 denormal inputs kept by the MFMA?
//   2. error of the 3-product split against double, next to the f32 MFMA chain's
//   3. time: a wave's "granule" = 64 f32 MFMAs + N fillers   against   42 f16 MFMAs + N + split work, 4 waves per SIMD
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/f16split tools/ubench/mfma_f16split.cpp && /tmp/f16split
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 f16x2 __attribute__((ext_vector_type(2)));

// A: 16 x 32 (row i, k), B: 32 x 16 (k, col j); lane l holds A[l % 16][8 * (l / 16) .. + 8) and B[8 * (l / 16) .. + 8)[l % 16]
// D: lane l holds D[4 * (l / 16) + r][l % 16], r = 0..3
__global__ void k_acc(const float* A, const float* B, float* D32, float* D16, float scale) {
  const int l = threadIdx.x, i = l & 15, kb = l >> 4;
  // f32 chain: 8 MFMAs of K = 4: lane holds A[i][4 kk + kb], B[4 kk + kb][i]
  f32x4 c = {0, 0, 0, 0};
  for (int kk = 0; kk < 8; kk++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * 32 + 4 * kk + kb], B[(4 * kk + kb) * 16 + i], c, 0, 0, 0);
  for (int r = 0; r < 4; r++) D32[(4 * kb + r) * 16 + i] = c[r];
  f16x8 ah, al, bh, bl;
  for (int q = 0; q < 8; q++) {
    const float a = A[i * 32 + 8 * kb + q] * scale, b = B[(8 * kb + q) * 16 + i];
    const _Float16 h = (_Float16)a;
    ah[q] = h; al[q] = (_Float16)(a - (float)h);
    const _Float16 g = (_Float16)b;
    bh[q] = g; bl[q] = (_Float16)(b - (float)g);
  }
  f32x4 d = {0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, d, 0, 0, 0);
  d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, d, 0, 0, 0);
  d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, d, 0, 0, 0);
  for (int r = 0; r < 4; r++) D16[(4 * kb + r) * 16 + i] = d[r] / scale;
}

template <int MODE>
__global__ __launch_bounds__(1024) void k_time(float* out, int iters, int fill) {
  const int lane = threadIdx.x & 63;
  float a[20];
  for (int i = 0; i < 20; i++) a[i] = out[lane + i];
  float x[16];
  for (int i = 0; i < 16; i++) x[i] = out[lane + 32 + i];
  const float b = out[lane + 60], c = out[lane + 61];
  f32x4 m[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f16x8 bh, bl;
  for (int q = 0; q < 8; q++) { bh[q] = (_Float16)(lane + q); bl[q] = (_Float16)(0.001f * q); }
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int g = 0; g < 16; g++)
#pragma unroll
        for (int t = 0; t < 4; t++) m[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(g + t) % 20], b, m[t], 0, 0, 0);
    } else {
      // split 32 + 24 values (pairs: cvt_pkrtz, 2 cvt back, 2 sub, cvt_pkrtz), then 42 MFMAs
      f16x8 ah[4], al[4];
#pragma unroll
      for (int v = 0; v < 4; v++)
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
          const float p0 = a[(2 * v + q) % 20] + x[(v + q) & 15], p1 = a[(2 * v + q + 1) % 20];
          const f16x2 h = __builtin_amdgcn_cvt_pkrtz(p0, p1);
          const f16x2 lo = __builtin_amdgcn_cvt_pkrtz(p0 - (float)h[0], p1 - (float)h[1]);
          ah[v][q] = (_Float16)h[0]; ah[v][q + 1] = (_Float16)h[1]; al[v][q] = (_Float16)lo[0]; al[v][q + 1] = (_Float16)lo[1];
        }
#pragma unroll
      for (int g = 0; g < 14; g++) {
        m[g & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[g & 3], bh, m[g & 3], 0, 0, 0);
        m[(g + 1) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[g & 3], bl, m[(g + 1) & 3], 0, 0, 0);
        m[(g + 2) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[g & 3], bh, m[(g + 2) & 3], 0, 0, 0);
      }
      if (MODE == 2) {      // the split work of the 24 matrixing values too
#pragma unroll
        for (int q = 0; q < 12; q++) {
          const f16x2 h = __builtin_amdgcn_cvt_pkrtz(x[q], x[q + 1]);
          const f16x2 lo = __builtin_amdgcn_cvt_pkrtz(x[q] - (float)h[0], x[q + 1] - (float)h[1]);
          x[q] = (float)lo[0] + (float)lo[1] + (float)h[0];
        }
      }
    }
    for (int f = 0; f < fill; f++)
#pragma unroll
      for (int i = 0; i < 16; i++) x[i] = __builtin_fmaf(x[i], b, c);
  }
  float s = 0;
  for (int t = 0; t < 4; t++) s += m[t][0] + m[t][1] + m[t][2] + m[t][3];
  for (int i = 0; i < 16; i++) s += x[i];
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}

int main() {
  float hA[512], hB[512], h32[256], h16[256];
  float *dA, *dB, *d32, *d16;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&d32, 1024); hipMalloc(&d16, 1024);
  // 1. denormals: A = 2^-20 everywhere (an fp16 subnormal), B = 1: D = 32 * 2^-20 if they are kept
  for (int i = 0; i < 512; i++) { hA[i] = ldexpf(1.0f, -20); hB[i] = 1.0f; }
  hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_acc, dim3(1), dim3(64), 0, 0, dA, dB, d32, d16, 1.0f);
  hipMemcpy(h16, d16, 1024, hipMemcpyDeviceToHost);
  printf("denormal inputs (2^-20 x 32): f16 path gives %g (kept: %g)\n", h16[0], 32 * ldexp(1.0, -20));
  // 2. accuracy: A ~ spectra (|x| up to ~2, many small), B ~ cosines
  srand(7);
  for (int scale_log = 0; scale_log <= 8; scale_log += 8) {
    double worst32 = 0, worst16 = 0, rms32 = 0, rms16 = 0;
    for (int trial = 0; trial < 200; trial++) {
      for (int i = 0; i < 512; i++) {
        const double u = rand() / (double)RAND_MAX, v = rand() / (double)RAND_MAX;
        hA[i] = (float)((u - 0.5) * 4.0 * pow(10.0, -6.0 * v * (trial & 1)));      // odd trials: magnitudes over six decades
        hB[i] = (float)cos(rand() * 1e-3);
      }
      hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k_acc, dim3(1), dim3(64), 0, 0, dA, dB, d32, d16, ldexpf(1.0f, scale_log));
      hipMemcpy(h32, d32, 1024, hipMemcpyDeviceToHost); hipMemcpy(h16, d16, 1024, hipMemcpyDeviceToHost);
      for (int r = 0; r < 16; r++)
        for (int cI = 0; cI < 16; cI++) {
          double ref = 0;
          for (int k = 0; k < 32; k++) ref += (double)hA[r * 32 + k] * (double)hB[k * 16 + cI];
          const double e32 = fabs(h32[r * 16 + cI] - ref), e16 = fabs(h16[r * 16 + cI] - ref);
          if (e32 > worst32) worst32 = e32;
          if (e16 > worst16) worst16 = e16;
          rms32 += e32 * e32; rms16 += e16 * e16;
        }
    }
    printf("inputs x 2^%d: abs error vs double, outputs |y| ~ 5: f32 chain max %.3g rms %.3g | f16 3-product max %.3g rms %.3g\n", scale_log,
           worst32, sqrt(rms32 / (200 * 256)), worst16, sqrt(rms16 / (200 * 256)));
  }
  // 3. time
  float* dout; hipMalloc(&dout, 256 * 1024 * 4 + 4096); hipMemset(dout, 0, 256 * 1024 * 4 + 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int fill = 0; fill <= 96; fill += 96) {
    float ms[3];
    for (int mode = 0; mode < 3; mode++) {
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, 0);
        if (mode == 0) hipLaunchKernelGGL(k_time<0>, dim3(256), dim3(1024), 0, 0, dout, 50, fill);
        if (mode == 1) hipLaunchKernelGGL(k_time<1>, dim3(256), dim3(1024), 0, 0, dout, 50, fill);
        if (mode == 2) hipLaunchKernelGGL(k_time<2>, dim3(256), dim3(1024), 0, 0, dout, 50, fill);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[mode], e0, e1);
      }
    }
    printf("4 waves/SIMD, 50 iterations, %4d filler FMAs per iteration: 64 f32 MFMAs %.1f us | 42 f16 MFMAs + split of 32 values %.1f us | + split of 24 more %.1f us\n",
           fill * 16, ms[0] * 1e3, ms[1] * 1e3, ms[2] * 1e3);
  }
  return 0;
}
