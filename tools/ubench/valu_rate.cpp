// Micro-benchmark: VALU / MFMA issue rates on gfx950 as seen by ONE SIMD with 1, 2, 4 waves.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters, unsigned long long* ticks) {
  float a[16];
  for (int i = 0; i < 16; i++) a[i] = out[threadIdx.x + i];
  float b = out[threadIdx.x + 17], c = out[threadIdx.x + 18];
  f32x4 m0 = {0, 0, 0, 0}, m1 = {0, 0, 0, 0}, m2 = {0, 0, 0, 0}, m3 = {0, 0, 0, 0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {            // 16 independent FMA chains, 64 FMAs per iteration
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 16; i++) a[i] = __builtin_fmaf(a[i], b, c);
    } else if (MODE == 1) {     // 64 cndmask/mov-class ops
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 16; i++) a[i] = (a[i] > c) ? a[i] - b : a[i] + b;
    } else if (MODE == 2) {     // 16 MFMA 16x16x4 f32 on 4 accumulators
#pragma unroll
      for (int r = 0; r < 4; r++) {
        m0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b, m0, 0, 0, 0);
        m1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b, m1, 0, 0, 0);
        m2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b, m2, 0, 0, 0);
        m3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b, m3, 0, 0, 0);
      }
    } else if (MODE == 3) {     // MFMA interleaved with 4 FMAs each
#pragma unroll
      for (int r = 0; r < 4; r++) {
        m0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b, m0, 0, 0, 0);
        a[4] = __builtin_fmaf(a[4], b, c); a[5] = __builtin_fmaf(a[5], b, c); a[6] = __builtin_fmaf(a[6], b, c); a[7] = __builtin_fmaf(a[7], b, c);
        m1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b, m1, 0, 0, 0);
        a[8] = __builtin_fmaf(a[8], b, c); a[9] = __builtin_fmaf(a[9], b, c); a[10] = __builtin_fmaf(a[10], b, c); a[11] = __builtin_fmaf(a[11], b, c);
        m2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b, m2, 0, 0, 0);
        a[12] = __builtin_fmaf(a[12], b, c); a[13] = __builtin_fmaf(a[13], b, c); a[14] = __builtin_fmaf(a[14], b, c); a[15] = __builtin_fmaf(a[15], b, c);
        m3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b, m3, 0, 0, 0);
        a[4] = __builtin_fmaf(a[4], b, c); a[5] = __builtin_fmaf(a[5], b, c); a[6] = __builtin_fmaf(a[6], b, c); a[7] = __builtin_fmaf(a[7], b, c);
      }
    } else if (MODE == 4) {     // f64 mul
      double d[8];
#pragma unroll
      for (int i = 0; i < 8; i++) d[i] = a[i];
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) d[i] = d[i] * (double)b;
#pragma unroll
      for (int i = 0; i < 8; i++) a[i] = (float)d[i];
    } else if (MODE == 5) {     // packed fma: 32 v_pk_fma = 64 FMAs
      typedef float v2 __attribute__((ext_vector_type(2)));
      v2 p[8];
#pragma unroll
      for (int i = 0; i < 8; i++) p[i] = (v2){a[2 * i], a[2 * i + 1]};
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) p[i] = __builtin_elementwise_fma(p[i], (v2){b, b}, (v2){c, c});
#pragma unroll
      for (int i = 0; i < 8; i++) { a[2 * i] = p[i].x; a[2 * i + 1] = p[i].y; }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = m0[0] + m1[1] + m2[2] + m3[3];
  for (int i = 0; i < 16; i++) s += a[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, int ops_per_iter) {
  float* d; unsigned long long* t;
  hipMalloc(&d, 1 << 24); hipMemset(d, 0, 1 << 24); hipMalloc(&t, 8 * 65536);
  const int iters = 2000;
  for (int waves_per_simd = 1; waves_per_simd <= 8; waves_per_simd *= 2) {
    int blocks = 256 * 4 * waves_per_simd;     // 64-thread blocks: fills every SIMD with that many waves
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, t);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, t); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[64]; hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
    double tick = (double)h[0] / iters / ops_per_iter;
    double rate = (double)blocks * iters * ops_per_iter / (ms * 1e-3) / (1024.0) / 1e9;   // wave-instr per ns per SIMD
    printf("%-28s waves/SIMD %d: %.2f ticks per instr per wave; %.3f wave-instr/ns/SIMD => %.2f cycles/instr @2.4GHz per SIMD\n", name, waves_per_simd, tick, rate, 2.4 / rate);
  }
  hipFree(d); hipFree(t);
}
int main() {
  run<0>("v_fma_f32 (16 chains)", 64);
  run<1>("cmp+sub/add+cndmask", 64 * 4);
  run<2>("mfma 16x16x4 f32", 16);
  run<3>("mfma + 4 fma each", 16);
  run<4>("v_mul_f64", 64);
  run<5>("v_pk_fma_f32", 32);
  return 0;
}
