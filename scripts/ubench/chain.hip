// Microbenchmark: cost per iteration of the serial "x += l * x[lane i of my half]" chain in several encodings.
// One wave; s_memtime around 4096 iterations.  Build: hipcc --offload-arch=gfx950 -O3 chain.hip -o chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
__device__ __forceinline__ float bcast(float v, int i) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i)); }
template <int V> __global__ void k(float* out, unsigned long long* cyc, const float* lin) {
  const int lane = threadIdx.x; const bool upper = lane >= 32;
  float x = out[lane], l = lin[lane];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll 8
  for (int it = 0; it < N; it++) {
    const int i = it & 31;
    if (V == 0) { x = fmaf(l, x, 1e-3f); }                                                  // dependent FMA only
    if (V == 1) { x = fmaf(l, bcast(x, i), x); }                                             // readlane + fma
    if (V == 2) { const float a = bcast(x, i), b = bcast(x, i + 32); x = fmaf(l, upper ? b : a, x); }   // select
    if (V == 3) { const float a = bcast(x, i), b = bcast(x, i + 32);
      asm volatile("s_nop 0\n\ts_mov_b64 exec, %4\n\tv_fmac_f32_e32 %0, %2, %1\n\ts_mov_b64 exec, %5\n\tv_fmac_f32_e32 %0, %3, %1\n\ts_mov_b64 exec, -1"
                   : "+v"(x) : "v"(l), "s"(a), "s"(b), "s"(0x00000000ffffffffull), "s"(0xffffffff00000000ull)); }
    if (V == 4) { const int src = ((lane & 32) + i) << 2; x = fmaf(l, __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(x))), x); }
    if (V == 5) { __shared__ float sh[64]; sh[lane] = x; __builtin_amdgcn_wave_barrier(); x = fmaf(l, sh[(lane & 32) + i], x); __builtin_amdgcn_wave_barrier(); }
    if (V == 6) { const float a = bcast(x, i), b = bcast(x, i + 32); const float sel = upper ? b : a;   // + independent work
      x = fmaf(l, sel, x); }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[lane] = x; if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
  float *out, *lin; unsigned long long* cyc;
  hipMalloc(&out, 256); hipMalloc(&lin, 256); hipMalloc(&cyc, 8);
  float h[64]; for (int i = 0; i < 64; i++) h[i] = 1e-3f * i; hipMemcpy(out, h, 256, hipMemcpyHostToDevice);
  for (int i = 0; i < 64; i++) h[i] = 1e-4f; hipMemcpy(lin, h, 256, hipMemcpyHostToDevice);
  const char* names[] = {"fma chain", "readlane+fma", "2 readlane + select + fma", "2 readlane + exec-masked fmac x2", "ds_bpermute + fma", "LDS write/read + fma", "(same as 2)"};
#define RUN(V) { hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, out, cyc, lin); hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, out, cyc, lin); \
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-40s %.1f memtime ticks/iter\n", names[V], (double)c / N); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
  return 0;
}
