// What a launch of the granule kernel's SHAPE costs with nothing in it (256 workgroups x 1024 threads x 157 KB of LDS, and the 8-wave
// shape), back-to-back on one stream, HIP events around 200 launches: the floor under k_decode_g's 8.7 us "everything off" time.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/launch_floor.cpp -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ unsigned char smem[];
__global__ void __launch_bounds__(1024) k_empty(int* out) { if (out && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = smem[0]; }
// touches its LDS block, reads a granule's records (2 x (128 + 1152) B) and writes its PCM (2304 B) per wave: C2's algorithmic bytes
__global__ void __launch_bounds__(1024) k_touch(const uint4* in, uint4* out) {
  const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  uint4 a = in[(size_t)w * 160 + lane], b = in[(size_t)w * 160 + 64 + lane], c = lane < 32 ? in[(size_t)w * 160 + 128 + lane] : uint4{0, 0, 0, 0};
  reinterpret_cast<uint4*>(smem)[threadIdx.x] = a;
  __syncthreads();
  uint4 s = reinterpret_cast<uint4*>(smem)[threadIdx.x ^ 64];
  s.x ^= b.x ^ c.y;
  out[(size_t)w * 144 + lane] = s; out[(size_t)w * 144 + 64 + lane] = b;
  if (lane < 16) out[(size_t)w * 144 + 128 + lane] = c;
}
template <typename F> static float time_us(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 20; i++) f();
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int i = 0; i < 200; i++) f();
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / 200;
}
int main() {
  uint4 *in, *out;
  hipMalloc(&in, (size_t)4096 * 160 * 16); hipMalloc(&out, (size_t)4096 * 144 * 16);
  hipMemset(in, 1, (size_t)4096 * 160 * 16);
  hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k_touch, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const struct { int wgs, thr, lds; } shapes[] = {{256, 1024, 157 * 1024}, {512, 512, 80 * 1024}, {256, 1024, 16 * 1024}, {4096, 64, 0}, {1, 64, 0}};
  for (auto s : shapes) {
    float e = time_us([&] { hipLaunchKernelGGL(k_empty, dim3(s.wgs), dim3(s.thr), s.lds, 0, (int*)nullptr); });
    float t = s.thr * s.wgs == 262144 ? time_us([&] { hipLaunchKernelGGL(k_touch, dim3(s.wgs), dim3(s.thr), s.lds > 16384 ? s.lds : 16384, 0, in, out); }) : 0.f;
    printf("%4d workgroups x %4d threads, %3d KB LDS: empty %.2f us per launch, records in + PCM out only %.2f us\n", s.wgs, s.thr, s.lds / 1024, e, t);
  }
  return 0;
}
