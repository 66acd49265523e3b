// standalone check that the system HIP runtime (/opt/rocm) sees the GPU outside python/torch
#include <hip/hip_runtime.h>
#include <stdio.h>
int main() {
  int n = -1;
  hipError_t e = hipGetDeviceCount(&n);
  printf("hipGetDeviceCount -> %s, n=%d\n", hipGetErrorString(e), n);
  if (e == hipSuccess && n > 0) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device 0: %s, %d CUs, clock %d kHz, LDS/block %zu\n", p.name, p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock);
  }
  return 0;
}
