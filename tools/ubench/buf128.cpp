// raw_buffer_store_b128 / raw_buffer_load_b128 with aux = sc1 through make_buffer_rsrc: does lane l's float4 at voffset 16 (k 64 + l)
// land where a plain float index says it does?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* blk, float* out) {
  const int lane = threadIdx.x;
  auto rs = __builtin_amdgcn_make_buffer_rsrc(blk, 0, 9216, 0x00020000);
  for (int kk = 0; kk < 9; kk++) {
    u32x4 v = {__builtin_bit_cast(unsigned, (float)(1000 * kk + 4 * lane)), __builtin_bit_cast(unsigned, (float)(1000 * kk + 4 * lane + 1)),
               __builtin_bit_cast(unsigned, (float)(1000 * kk + 4 * lane + 2)), __builtin_bit_cast(unsigned, (float)(1000 * kk + 4 * lane + 3))};
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (kk * 64 + lane) * 16, 0, 16);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int kk = 0; kk < 9; kk++) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (kk * 64 + (63 - lane)) * 16, 0, 16);
    out[(kk * 64 + lane) * 4 + 0] = __builtin_bit_cast(float, v.x); out[(kk * 64 + lane) * 4 + 1] = __builtin_bit_cast(float, v.y);
    out[(kk * 64 + lane) * 4 + 2] = __builtin_bit_cast(float, v.z); out[(kk * 64 + lane) * 4 + 3] = __builtin_bit_cast(float, v.w);
  }
}
int main() {
  float *d, *o; hipMalloc(&d, 9216); hipMalloc(&o, 9216); hipMemset(d, 0, 9216);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  std::vector<float> h(2304), g(2304);
  hipMemcpy(h.data(), d, 9216, hipMemcpyDeviceToHost); hipMemcpy(g.data(), o, 9216, hipMemcpyDeviceToHost);
  int bad = 0, bad2 = 0;
  for (int kk = 0; kk < 9; kk++) for (int l = 0; l < 64; l++) for (int c = 0; c < 4; c++) {
    bad += h[(kk * 64 + l) * 4 + c] != (float)(1000 * kk + 4 * l + c);
    bad2 += g[(kk * 64 + l) * 4 + c] != (float)(1000 * kk + 4 * (63 - l) + c);
  }
  printf("stored wrong: %d, loaded wrong: %d; first values %g %g %g %g | %g\n", bad, bad2, h[0], h[1], h[2], h[3], h[256]);
  return 0;
}
