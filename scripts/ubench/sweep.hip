// Microbenchmark of the triangular sweep of the two-env kernel (per-dof form, LDS reads batched by 8).
#include <hip/hip_runtime.h>
#include <cstdio>
#define RS 20
__device__ __forceinline__ float bcast(float v, int i) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i)); }
__device__ __forceinline__ float bcast2(float v, int i, bool upper) { const float a = bcast(v, i), b = bcast(v, i + 32); return upper ? b : a; }
__device__ __forceinline__ float mask_select(const float v, const unsigned long long m) { float r; asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(m)); return r; }
template <int V> __global__ void k(float* out, unsigned long long* cyc, int iters, int ND, const unsigned long long* masks) {
  __shared__ __align__(16) float lds[2 * 32 * RS];
  const int lane = threadIdx.x, sl = lane & 31; const bool upper = lane >= 32;
  float* HR = lds + (upper ? 32 * RS : 0);
  for (int i = sl; i < 32 * RS; i += 32) HR[i] = 1e-3f * (i % 13);
  const int ddepth = sl % 17, dsub = 1 + (sl * 7) % 9;
  float x = 1.f + lane;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    int i = ND - 1;
    if (V == 0) {
      for (; i >= 7; i -= 8) {
        float l[8];
#pragma unroll
        for (int u = 0; u < 8; u++) l[u] = HR[(i - u) * RS + ddepth];
#pragma unroll
        for (int u = 0; u < 8; u++) { const bool anc = sl < i - u && i - u < sl + dsub; x = fmaf(anc ? -l[u] : 0.f, bcast2(x, i - u, upper), x); }
      }
    } else if (V == 1) {                       // masks from a constant table
      typedef const unsigned long long __attribute__((address_space(4)))* cm;
      const cm MK = (cm)masks;
      for (; i >= 7; i -= 8) {
        float l[8];
#pragma unroll
        for (int u = 0; u < 8; u++) l[u] = HR[(i - u) * RS + ddepth];
#pragma unroll
        for (int u = 0; u < 8; u++) x = fmaf(-mask_select(l[u], MK[i - u]), bcast2(x, i - u, upper), x);
      }
    } else if (V == 2) {                       // no predicate at all (lower bound)
      for (; i >= 7; i -= 8) {
        float l[8];
#pragma unroll
        for (int u = 0; u < 8; u++) l[u] = HR[(i - u) * RS + ddepth];
#pragma unroll
        for (int u = 0; u < 8; u++) x = fmaf(-l[u], bcast2(x, i - u, upper), x);
      }
    } else if (V == 3) {                       // single readlane (one env per wave)
      for (; i >= 7; i -= 8) {
        float l[8];
#pragma unroll
        for (int u = 0; u < 8; u++) l[u] = HR[(i - u) * RS + ddepth];
#pragma unroll
        for (int u = 0; u < 8; u++) { const bool anc = sl < i - u && i - u < sl + dsub; x = fmaf(anc ? -l[u] : 0.f, bcast(x, i - u), x); }
      }
    }
    x *= 0.5f;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[lane] = x; if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc; unsigned long long* mk; unsigned long long h[32];
  (void)hipMalloc(&out, 256); (void)hipMalloc(&cyc, 8); (void)hipMalloc(&mk, sizeof h);
  for (int i = 0; i < 32; i++) h[i] = 0x0000ffff0000ffffull >> (i % 7);
  (void)hipMemcpy(mk, h, sizeof h, hipMemcpyHostToDevice);
  const char* names[] = {"compare predicates, select broadcast", "scalar-cache masks, select broadcast", "no predicate", "one env per wave (single readlane)"};
  const int iters = 500, ND = 32;
#define RUN(V) { hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, out, cyc, iters, ND, mk); hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, out, cyc, iters, ND, mk); \
  unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-44s %.1f ticks/dof\n", names[V], (double)c / iters / 32); }
  RUN(0) RUN(1) RUN(2) RUN(3)
  return 0;
}
