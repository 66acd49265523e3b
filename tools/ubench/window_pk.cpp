// Micro-benchmark: does packed f32 arithmetic pay for the two fattest phases of k_decode_g?
//   window  -- the 18 x 16-tap sums of a lane (decode_core.h ph_window_own / _hist): 288 v_fma_f32, or 144 v_pk_fma_f32 on
//              pairs of time slots (t, t + 1) (the odd-age history kept shifted by one so that its pairs are register-aligned)
//   ms      -- (float)((double)x * sqrt(1/2)) of P:1923-1926: v_cvt_f64_f32 + v_mul_f64 + v_cvt_f32_f64 per value
//   rates   -- v_pk_mul_f32, v_pk_add_f32, v_perm_b32, v_med3 + v_cvt
// 1, 2, 4 waves per SIMD, NO MFMA anywhere near (the guide's "packed f32 is an anti-lever" note is about MFMA-adjacent slots).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o window_pk window_pk.cpp && ./window_pk
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(64) void k(float* io, int iters, unsigned long long* ticks) {
  const int lane = threadIdx.x;
  float E[34], O[34], we[8], wo[8];
  for (int i = 0; i < 34; i++) { E[i] = io[lane + 64 * i]; O[i] = io[lane + 64 * (34 + i)]; }
  for (int k2 = 0; k2 < 8; k2++) { we[k2] = io[lane + 64 * (70 + k2)]; wo[k2] = io[lane + 64 * (80 + k2)]; }
  float sum[18];
  for (int t = 0; t < 18; t++) sum[t] = 0.0f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {               // scalar: sum[t] = chain over k of we[k] E[15 + t - 2k], wo[k] O[15 + t - 2k - 1]
#pragma unroll
      for (int t = 0; t < 18; t++) {
        float acc = sum[t] * 0.5f;
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) {
          acc = __builtin_fmaf(we[k2], E[15 + t - 2 * k2], acc);
          acc = __builtin_fmaf(wo[k2], O[15 + t - 2 * k2 - 1], acc);
        }
        sum[t] = acc;
      }
    } else if (MODE == 1) {        // packed: pairs (t, t + 1), t even; E pairs start even, O is held shifted: Os[s] = O[s - 1]
#pragma unroll
      for (int t = 0; t < 18; t += 2) {
        v2 acc = (v2){sum[t], sum[t + 1]} * (v2){0.5f, 0.5f};
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) {
          // E index 15 + t - 2k is ODD for even t: so E too is held shifted by one (Es[s] = E[s - 1]): index 16 + t - 2k even
          acc = __builtin_elementwise_fma((v2){we[k2], we[k2]}, (v2){E[16 + t - 2 * k2], E[17 + t - 2 * k2]}, acc);
          acc = __builtin_elementwise_fma((v2){wo[k2], wo[k2]}, (v2){O[14 + t - 2 * k2], O[15 + t - 2 * k2]}, acc);
        }
        sum[t] = acc.x; sum[t + 1] = acc.y;
      }
    } else if (MODE == 2) {        // MS scaling in binary64, 36 values
#pragma unroll
      for (int t = 0; t < 18; t++) {
        sum[t] = (float)((double)(sum[t] + E[t]) * 0.70710678118654752440);
        E[t] = (float)((double)(E[t] - O[t]) * 0.70710678118654752440);
      }
    } else if (MODE == 3) {        // 36 v_pk_mul_f32 = 72 products
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int t = 0; t < 18; t += 2) {
          v2 a = (v2){sum[t], sum[t + 1]} * (v2){E[t], E[t + 1]};
          sum[t] = a.x; sum[t + 1] = a.y;
        }
    } else if (MODE == 4) {        // the same 72 products as v_mul_f32
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int t = 0; t < 18; t++) sum[t] = sum[t] * E[t];
    } else if (MODE == 5) {        // conversion tail: med3 + cvt + compare/select per value (18), as pcm_convert18 has it
#pragma unroll
      for (int t = 0; t < 18; t++) {
        const int n = (int)__builtin_amdgcn_fmed3f(sum[t] * 32767.0f, -32767.0f, 32767.0f);
        sum[t] = __int_as_float((sum[t] <= 65538.0f) ? n : -32767) + E[t];
      }
    } else if (MODE == 6) {        // MS scaling with binary32 error-free products (p, e, t, s): 4 ops per value, packed: 2
#pragma unroll
      for (int t = 0; t < 18; t += 2) {
        const v2 chi = {0x1.6a09e6p-1f, 0x1.6a09e6p-1f}, clo = {0x1.9fcef4p-27f, 0x1.9fcef4p-27f};
        v2 x = (v2){sum[t], sum[t + 1]} + (v2){E[t], E[t + 1]};
        v2 p = x * chi;
        v2 e = __builtin_elementwise_fma(x, chi, -p);
        v2 tt = __builtin_elementwise_fma(x, clo, e);
        v2 s = p + tt;
        v2 d = (p - s) + tt;
        sum[t] = s.x + d.x * 0x1p-30f; sum[t + 1] = s.y + d.y * 0x1p-30f;
        v2 y = (v2){E[t], E[t + 1]} - (v2){O[t], O[t + 1]};
        p = y * chi; e = __builtin_elementwise_fma(y, chi, -p); tt = __builtin_elementwise_fma(y, clo, e); s = p + tt; d = (p - s) + tt;
        E[t] = s.x + d.x * 0x1p-30f; E[t + 1] = s.y + d.y * 0x1p-30f;
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.0f;
  for (int t = 0; t < 18; t++) s += sum[t] + E[t];
  io[blockIdx.x * 64 + lane] = s;
  if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, int ops_per_iter) {
  float* d; unsigned long long* t;
  hipMalloc(&d, 1 << 24); hipMemset(d, 0, 1 << 24); hipMalloc(&t, 8 * 65536);
  const int iters = 1000;
  for (int w = 1; w <= 4; w *= 2) {
    const int blocks = 256 * 4 * w;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, t);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, t); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_iter_ns = ms * 1e6 / iters / w;      // ns of one SIMD per iteration of one wave
    printf("%-44s waves/SIMD %d: %8.1f ns per pass per wave-slot = %6.2f SIMD-cycles @2.4GHz per counted op (%d ops)\n",
           name, w, per_iter_ns, per_iter_ns * 2.4 / ops_per_iter, ops_per_iter);
  }
  hipFree(d); hipFree(t);
}
int main() {
  run<0>("window, 288 v_fma_f32", 288);
  run<1>("window, 144 v_pk_fma_f32", 144);
  run<2>("MS scale in f64 (36 x cvt+mul+cvt)", 36);
  run<6>("MS scale in f32 EFT, packed (36 values)", 36);
  run<3>("v_pk_mul_f32 (36 = 72 products)", 36);
  run<4>("v_mul_f32 (72)", 72);
  run<5>("mul + med3 + cvt + cmp + cndmask (18 values)", 18);
  return 0;
}
