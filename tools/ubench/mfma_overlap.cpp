// Micro-benchmark: does an f32 MFMA (v_mfma_f32_16x16x4_f32) of one wave overlap with VALU / LDS work of ANOTHER wave
// on the same SIMD, and with independent VALU work of the SAME wave?  Same questions for the bf16 MFMA for contrast.
// One 512-thread workgroup per CU = 2 waves per SIMD (waves w and w + 4 share a SIMD; HW_ID is printed to check).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_overlap mfma_overlap.cpp && ./mfma_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

enum Role { R_IDLE = 0, R_MFMA32, R_FMA, R_INT, R_LDS, R_MFMABF, R_MIX4, R_MIX8, R_MIX12, R_MIXI8, R_CVT64 };

template <int role>
__device__ __noinline__ void body(int iters, float* out, float* lds, unsigned long long* ticks, int slot) {
  const int lane = threadIdx.x & 63;
  float a[16];
  for (int i = 0; i < 16; i++) a[i] = out[lane + i];
  float b = out[lane + 17], c = out[lane + 18];
  int ia[16];
  for (int i = 0; i < 16; i++) ia[i] = (int)a[i] + i;
  f32x4 m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0;
  bf16x8 ba, bb;
  for (int i = 0; i < 8; i++) { ba[i] = (short)(lane + i); bb[i] = (short)(lane * 3 + i); }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    switch (role) {   // compile-time
      case R_MFMA32:   // 16 MFMA
#pragma unroll
        for (int r = 0; r < 4; r++) {
          m0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b, m0, 0, 0, 0);
          m1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b, m1, 0, 0, 0);
          m2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b, m2, 0, 0, 0);
          m3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b, m3, 0, 0, 0);
        }
        break;
      case R_MFMABF:   // 16 bf16 MFMA 16x16x32
#pragma unroll
        for (int r = 0; r < 4; r++) {
          m0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, m0, 0, 0, 0);
          m1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, m1, 0, 0, 0);
          m2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, m2, 0, 0, 0);
          m3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, m3, 0, 0, 0);
        }
        break;
      case R_FMA:      // 192 FMA
#pragma unroll
        for (int r = 0; r < 12; r++)
#pragma unroll
          for (int i = 0; i < 16; i++) a[i] = __builtin_fmaf(a[i], b, c);
        break;
      case R_INT:      // 192 integer ops
#pragma unroll
        for (int r = 0; r < 12; r++)
#pragma unroll
          for (int i = 0; i < 16; i++) ia[i] = (ia[i] ^ (ia[(i + 1) & 15] >> 3)) + it;
        break;
      case R_LDS:      // 64 ds_read_b32
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int i = 0; i < 16; i++) a[i] += lds[(lane + 64 * i + 33 * r + (ia[0] & 1)) & 4095];
        break;
      case R_MIX4: case R_MIX8: case R_MIX12: {   // 16 x (1 MFMA + k independent FMAs) in one wave
        const int k = role == R_MIX4 ? 4 : role == R_MIX8 ? 8 : 12;
#pragma unroll
        for (int r = 0; r < 16; r++) {
          f32x4& m = (r & 3) == 0 ? m0 : (r & 3) == 1 ? m1 : (r & 3) == 2 ? m2 : m3;
          m = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, m, 0, 0, 0);
          if (k == 4) {
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = __builtin_fmaf(a[i], b, c);
          } else if (k == 8) {
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = __builtin_fmaf(a[i], b, c);
          } else {
#pragma unroll
            for (int i = 0; i < 12; i++) a[i] = __builtin_fmaf(a[i], b, c);
          }
        }
        break;
      }
      case R_MIXI8:    // 16 x (1 MFMA + 8 independent integer ops)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          f32x4& m = (r & 3) == 0 ? m0 : (r & 3) == 1 ? m1 : (r & 3) == 2 ? m2 : m3;
          m = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, m, 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 8; i++) ia[i] = (ia[i] ^ (ia[(i + 1) & 15] >> 3)) + it;
        }
        break;
      case R_CVT64: {  // 32 x (cvt f64, mul f64, cvt i32)
#pragma unroll
        for (int i = 0; i < 16; i++) {
          ia[i] += (int)((double)a[i] * 32767.0);
          ia[i] ^= (int)((double)a[(i + 1) & 15] * 32765.0);
        }
        break;
      }
      default: break;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = m0[0] + m1[1] + m2[2] + m3[3];
  for (int i = 0; i < 16; i++) s += a[i] + (float)ia[i];
  out[(blockIdx.x * blockDim.x + threadIdx.x) & 65535] = s;
  if (lane == 0) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    ticks[(size_t)blockIdx.x * 16 + slot] = t1 - t0;
    ticks[(size_t)blockIdx.x * 16 + 8 + slot] = hwid;
  }
}

__global__ __launch_bounds__(512) void k(int role_lo, int role_hi, int iters, float* out, unsigned long long* ticks) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = (float)i;
  const int w = threadIdx.x >> 6;
  const int role = __builtin_amdgcn_readfirstlane(w < 4 ? role_lo : role_hi);
  switch (role) {
#define C(R) case R: body<R>(iters, out, lds, ticks, w); break;
    C(R_IDLE) C(R_MFMA32) C(R_FMA) C(R_INT) C(R_LDS) C(R_MFMABF) C(R_MIX4) C(R_MIX8) C(R_MIX12) C(R_MIXI8) C(R_CVT64)
#undef C
  }
}

static const char* names[] = {"idle", "mfma_f32 x16", "fma x192", "int x192", "ds_read x64", "mfma_bf16 x16",
                              "16x(mfma+4fma)", "16x(mfma+8fma)", "16x(mfma+12fma)", "16x(mfma+8int)", "cvt64 x32"};

int main() {
  float* d; unsigned long long* t;
  hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20); hipMalloc(&t, 256 * 16 * 8);
  const int iters = 500;
  const int pairs[][2] = {
      {R_MFMA32, R_IDLE}, {R_FMA, R_IDLE}, {R_INT, R_IDLE}, {R_LDS, R_IDLE}, {R_MFMABF, R_IDLE}, {R_CVT64, R_IDLE},
      {R_MIX4, R_IDLE}, {R_MIX8, R_IDLE}, {R_MIX12, R_IDLE}, {R_MIXI8, R_IDLE},
      {R_MFMA32, R_MFMA32}, {R_FMA, R_FMA}, {R_INT, R_INT}, {R_LDS, R_LDS},
      {R_MFMA32, R_FMA}, {R_MFMA32, R_INT}, {R_MFMA32, R_LDS}, {R_MFMA32, R_CVT64},
      {R_MFMABF, R_FMA}, {R_MFMABF, R_INT}, {R_FMA, R_INT}, {R_FMA, R_LDS}, {R_MIX8, R_MIX8}, {R_MIX8, R_FMA},
  };
  for (auto& p : pairs) {
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, p[0], p[1], iters, d, t);
      hipDeviceSynchronize();
    }
    unsigned long long h[16];
    hipMemcpy(h, t + 16 * 7, sizeof h, hipMemcpyDeviceToHost);   // workgroup 7
    printf("%-16s | %-16s : ticks/iter lo %.1f %.1f %.1f %.1f  hi %.1f %.1f %.1f %.1f   simd", names[p[0]], names[p[1]],
           h[0] / (double)iters, h[1] / (double)iters, h[2] / (double)iters, h[3] / (double)iters, h[4] / (double)iters,
           h[5] / (double)iters, h[6] / (double)iters, h[7] / (double)iters);
    for (int w = 0; w < 8; w++) printf(" %llu", (h[8 + w] >> 4) & 3);
    printf("\n");
  }
  return 0;
}
