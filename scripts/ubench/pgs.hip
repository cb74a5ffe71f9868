// Microbenchmark of the PGS row step of the constraint kernel (fmj_cons_rows.inc, PGS_SWEEPS): lane = row, the lane's row of
// A in registers; per row: v_max (candidate update of every lane), v_cndmask (lane e keeps the residual its update was made
// from), v_readlane (row e's update), v_fmac (every residual).  Variants: 0 = as shipped (one env per wave), 1 = two envs per
// wave (rows of env A in lanes 0..31, of env B in lanes 32..63: two readlanes, a select, one fmac for both), 2 = one env, no
// capture (lower bound of the chain).  Run with 1 and with 2 waves per SIMD: build with hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#define NA 32
__device__ __forceinline__ float bcast(float v, int i) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i)); }
template <int V, int NR> __global__ void k(float* out, unsigned long long* cyc, int sweeps) {
  const int lane = threadIdx.x & 63; const bool upper = lane >= 32;
  float areg[NA];
#pragma unroll
  for (int i = 0; i < NA; i++) areg[i] = (i == (lane & 31) ? -1.f : 1e-3f * ((lane * 7 + i * 3) % 11 - 5));
  float res = 0.01f * lane - 0.3f, nf = -0.1f * (lane % 3);
  float acc = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < sweeps; it++) {
    float capr = 0.f;
#pragma unroll
    for (int e = 0; e < NR; e++) {
      float cand; asm("v_max_f32_e32 %0, %1, %2" : "=v"(cand) : "v"(nf), "v"(res));
      if (V == 0) {
        unsigned long long bit_; asm volatile("s_lshl_b64 %0, 1, %1" : "=s"(bit_) : "n"(e) : "scc");
        asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(capr) : "v"(res), "s"(bit_));
        res = fmaf(areg[e], bcast(cand, e), res);
      } else if (V == 1) {
        unsigned long long bit_; { unsigned lo_; asm volatile("s_lshl_b32 %0, 1, %1" : "=s"(lo_) : "n"(e) : "scc"); bit_ = ((unsigned long long)lo_ << 32) | lo_; }
        asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(capr) : "v"(res), "s"(bit_));
        const float a = bcast(cand, e), b = bcast(cand, e + 32);
        res = fmaf(areg[e], upper ? b : a, res);
      } else if (V == 3) {            // capture in the shadow of the readlane -> fmac wait
        unsigned long long bit_; asm volatile("s_lshl_b64 %0, 1, %1" : "=s"(bit_) : "n"(e) : "scc");
        const float s_ = bcast(cand, e);
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(capr) : "v"(res), "s"(bit_));
        res = fmaf(areg[e], s_, res);
      } else if (V == 4) {            // two envs per wave, capture in the shadow
        unsigned long long bit_; { unsigned lo_; asm volatile("s_lshl_b32 %0, 1, %1" : "=s"(lo_) : "n"(e) : "scc"); bit_ = ((unsigned long long)lo_ << 32) | lo_; }
        const float a = bcast(cand, e), b = bcast(cand, e + 32);
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(capr) : "v"(res), "s"(bit_));
        res = fmaf(areg[e], upper ? b : a, res);
      } else if (V == 5) {            // one env, capture of cand after the fmac (off the chain entirely)
        unsigned long long bit_; asm volatile("s_lshl_b64 %0, 1, %1" : "=s"(bit_) : "n"(e) : "scc");
        res = fmaf(areg[e], bcast(cand, e), res);
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(capr) : "v"(cand), "s"(bit_));
      } else {
        res = fmaf(areg[e], bcast(cand, e), res);
      }
    }
    float capc; asm("v_max_f32_e32 %0, %1, %2" : "=v"(capc) : "v"(nf), "v"(capr));
    nf -= capc; acc += capc * capr;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x + blockIdx.x * blockDim.x] = res + nf + acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 4 * 65536); (void)hipMalloc(&cyc, 8);
  const int sweeps = 2000;
#define RUN(V, NR, THREADS, label) { hipLaunchKernelGGL((k<V, NR>), dim3(1), dim3(THREADS), 0, 0, out, cyc, sweeps); hipLaunchKernelGGL((k<V, NR>), dim3(1), dim3(THREADS), 0, 0, out, cyc, sweeps); \
  unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-58s %2d rows, %d wave(s)/SIMD: %6.1f cycles per row step (%5.1f per env-row)\n", label, NR, THREADS / 256 ? THREADS / 256 : 1, (double)c / sweeps / NR, (double)c / sweeps / NR / ((V == 1 || V == 4) ? 2 : 1)); }
  RUN(0, 24, 64, "one env per wave (shipped)") RUN(0, 24, 512, "one env per wave (shipped)") RUN(0, 24, 1024, "one env per wave (shipped)")
  RUN(2, 24, 64, "one env per wave, no capture") RUN(2, 24, 512, "one env per wave, no capture")
  RUN(1, 24, 64, "two envs per wave") RUN(1, 24, 512, "two envs per wave")
  RUN(3, 24, 64, "one env, capture after the readlane") RUN(5, 24, 64, "one env, capture of cand after the fmac") RUN(4, 24, 64, "two envs, capture after the readlanes")
  return 0;
}
