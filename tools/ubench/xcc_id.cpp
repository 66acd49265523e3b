// Which XCD does workgroup b run on?  HW_REG_XCC_ID (s_getreg) against b % 8, for a grid of 256 and of 1024 workgroups.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/xcc_id tools/ubench/xcc_id.cpp && /tmp/xcc_id
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out) {
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20);          // 4 bits of register 20
    out[blockIdx.x * 4 + 1] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20);         // all 32
    out[blockIdx.x * 4 + 2] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID (reference)
  }
}
int main() {
  for (int n : {256, 1024}) {
    unsigned* d; hipMalloc(&d, n * 16);
    hipLaunchKernelGGL(k, dim3(n), dim3(1024), 0, 0, d);
    std::vector<unsigned> h(n * 4);
    hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
    int match = 0; unsigned seen = 0;
    for (int b = 0; b < n; b++) { match += (h[b * 4] & 7u) == (unsigned)(b & 7); seen |= 1u << (h[b * 4] & 15u); }
    printf("grid %d: XCC_ID & 7 == blockIdx %% 8 for %d of %d workgroups; values seen (mask) 0x%x; first eight: ", n, match, n, seen);
    for (int b = 0; b < 8; b++) printf("%u(0x%x) ", h[b * 4], h[b * 4 + 1]);
    printf("\n");
    hipFree(d);
  }
  return 0;
}
