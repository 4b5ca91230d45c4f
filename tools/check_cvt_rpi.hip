// Is v_cvt_rpi_i32_f32(x) == v_cvt_flr_i32_f32(x + 0.5f) for every float32 x?  (map_quantize rounds half up as
// floor(x + 0.5) with the addition rounded to float32, utils.py / maps.py:537-544; one instruction instead of two
// only if the hardware's "round to nearest, ties to +inf" agrees for ALL inputs.)  Exhaustive over the 2^32 patterns.
//   hipcc --offload-arch=gfx950 -O2 tools/check_cvt_rpi.hip -o tools/tmp/check_cvt_rpi && tools/tmp/check_cvt_rpi
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(unsigned long long* bad, unsigned* first) {
  const unsigned long long i0 = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 256ull;
  unsigned long long n = 0;
  for (unsigned long long i = i0; i < i0 + 256ull; ++i) {
    const float x = __uint_as_float((unsigned)i);
    float y = x + 0.5f;
    asm volatile("" : "+v"(y));
    int a, b;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(a) : "v"(y));
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(b) : "v"(x));
    if (a != b) { ++n; atomicMin(first + (x < 0.0f ? 1 : 0), (unsigned)i & 0x7fffffffu); }
  }
  if (n) atomicAdd(bad, n);
}
int main() {
  unsigned long long* bad; unsigned* first;
  hipMalloc(&bad, 8); hipMalloc(&first, 8);
  hipMemset(bad, 0, 8); hipMemset(first, 0xff, 8);
  k<<<65536, 256>>>(bad, first);
  unsigned long long h = 0; unsigned f[2];
  hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(f, first, 8, hipMemcpyDeviceToHost);
  union { unsigned u; float x; } p = {f[0]}, q = {f[1] | 0x80000000u};
  printf("patterns that differ: %llu  (smallest |x|: %g = 0x%08x positive, %g negative)\n", h, p.x, f[0], q.x);
  return 0;
}
