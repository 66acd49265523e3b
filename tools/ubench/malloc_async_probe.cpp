// Does memory from hipMallocAsync ever come back with a kernel's writes missing?  The sequence of an LSF batch of the streaming API
// (engine.hip launch_decode until round 6): allocate on a non-blocking stream, one kernel writes, the next reads, free on the
// stream, synchronise; then the same again at once.  A fresh process per trial (the failures were only ever seen in a process's
// first seconds).  Prints the number of 16-byte words the second kernel found unwritten.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/malloc_async_probe.cpp -o /tmp/map && for i in $(seq 50); do /tmp/map; done | sort | uniq -c
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_write(uint4* p, int n, unsigned tag) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = make_uint4(tag, (unsigned)i, tag ^ 0x5a5a5a5au, ~(unsigned)i); }
__global__ void k_check(const uint4* p, int n, unsigned tag, unsigned* bad) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { const uint4 v = p[i]; if (v.x != tag || v.y != (unsigned)i) atomicAdd(bad, 1u); } }
int main(int argc, char** argv) {
  const size_t big = argc > 2 ? (size_t)atol(argv[2]) : (size_t)8 * 4608;   // bytes of the first buffer (16-byte words counted)
  const char mode = argc > 1 ? argv[1][0] : 'a';     // a: hipMallocAsync | m: hipMalloc | s: hipMallocAsync + a stream synchronise before the first kernel | z: + hipMemsetAsync first
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned* bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
  unsigned total = 0;
  for (int batch = 0; batch < 4; batch++) {
    const int n = 8 * 4608 / 16 + 8 * 512 / 16;
    uint4 *a, *b;
    if (mode == 'm' || mode == 'M') { hipMalloc((void**)&a, big); hipMalloc((void**)&b, 8 * 512); }
    else { hipMallocAsync((void**)&a, big, s); hipMallocAsync((void**)&b, 8 * 512, s); }
    if (mode == 'M') { hipMemset(a, 0, big); hipMemset(b, 0, 8 * 512); }       // M: hipMalloc + a synchronous hipMemset before use
    if (mode == 's') hipStreamSynchronize(s);
    if (mode == 'z') { hipMemsetAsync(a, 0, big, s); hipMemsetAsync(b, 0, 8 * 512, s); }
    hipLaunchKernelGGL(k_write, dim3(64), dim3(256), 0, s, a, (int)(big / 16), 0x1000u + batch);
    hipLaunchKernelGGL(k_write, dim3(8), dim3(256), 0, s, b, 8 * 512 / 16, 0x2000u + batch);
    hipLaunchKernelGGL(k_check, dim3(64), dim3(64), 0, s, a, (int)(big / 16), 0x1000u + batch, bad);
    hipLaunchKernelGGL(k_check, dim3(8), dim3(64), 0, s, b, 8 * 512 / 16, 0x2000u + batch, bad);
    if (mode == 'm' || mode == 'M') { hipStreamSynchronize(s); hipFree(a); hipFree(b); } else { hipFreeAsync(a, s); hipFreeAsync(b, s); }
    hipStreamSynchronize(s);
    unsigned h; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost); total = h; (void)n;
  }
  printf("mode %c unwritten words: %u\n", mode, total);
  return 0;
}
