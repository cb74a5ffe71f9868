// Microbenchmark of the PAIR-mode PGS turn of the two-env constraint kernel (fmj_cons2_rows.inc): rows of env A in lanes 0..31, of
// env B in lanes 32..63, turn e serves row e of both.  How does a lane get ITS half's update?
//   0  as shipped in round 4: the lane's row of A split into two zero-padded multiplicands (64 registers), two readlanes, two fmacs
//   1  32 registers, two readlanes, the two fmacs under exec masks of the halves (three exec writes per turn)
//   2  32 registers, ONE ds_swizzle (bit-mask mode: and 0, or e = every lane reads lane e of its own group of 32), one fmac
//   3  32 registers, two readlanes, v_mov + v_cndmask select, one fmac
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/pgs2.hip -o scripts/ubench/pgs2
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float bcast(float v, int i) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i)); }
template <int V, int NR> __global__ void k(float* out, unsigned long long* cyc, int sweeps) {
  const int lane = threadIdx.x & 63; const bool upper = lane >= 32;
  float areg[64];
#pragma unroll
  for (int i = 0; i < 32; i++) {
    const float a = (i == (lane & 31) ? -1.f : 1e-3f * ((lane * 7 + i * 3) % 11 - 5));
    if (V == 0 || V == 4) { areg[i] = upper ? 0.f : a; areg[32 + i] = upper ? a : 0.f; } else { areg[i] = a; areg[32 + i] = 0.f; }
  }
  float res = 0.01f * (lane & 31) - 0.3f + 0.001f * upper, nf = -0.1f * (lane % 3);
  float acc = 0.f;
  const unsigned long long pairbase = 0x0000000100000001ull, lomask = 0x00000000ffffffffull;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < sweeps; it++) {
    float capr = 0.f;
#pragma unroll
    for (int e = 0; e < NR; e++) {
      if (V == 4) {      // hand-scheduled turn: 7 issue slots, the waits (VALU write -> v_readlane: 1; v_readlane's SGPR -> VALU read: 2) filled by its own instructions
        float cand; unsigned long long bit_; float sa, sb;
        asm volatile("v_max_f32_e32 %[cand], %[nf], %[res]\n\t"
                     "s_lshl_b64 %[bit], %[pb], %[e]\n\t"
                     "v_readlane_b32 %[sa], %[cand], %[e]\n\t"
                     "v_readlane_b32 %[sb], %[cand], %[e32]\n\t"
                     "v_cndmask_b32_e64 %[capr], %[capr], %[res], %[bit]\n\t"
                     "v_fmac_f32_e32 %[res], %[sa], %[a0]\n\t"
                     "v_fmac_f32_e32 %[res], %[sb], %[a1]"
                     : [cand] "=&v"(cand), [bit] "=&s"(bit_), [sa] "=&s"(sa), [sb] "=&s"(sb), [capr] "+v"(capr), [res] "+v"(res)
                     : [nf] "v"(nf), [pb] "s"(pairbase), [e] "n"(e), [e32] "n"(e + 32), [a0] "v"(areg[e]), [a1] "v"(areg[32 + e]) : "scc");
        continue;
      }
      float cand; asm("v_max_f32_e32 %0, %1, %2" : "=v"(cand) : "v"(nf), "v"(res));
      unsigned long long bit_; asm volatile("s_lshl_b64 %0, %1, %2" : "=s"(bit_) : "s"(pairbase), "n"(e) : "scc");
      if (V == 0) {
        const float ua = bcast(cand, e), ub = bcast(cand, e + 32);
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(capr) : "v"(res), "s"(bit_));
        res = fmaf(areg[e], ua, res); res = fmaf(areg[32 + e], ub, res);
      } else if (V == 1) {
        const float ua = bcast(cand, e), ub = bcast(cand, e + 32);
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(capr) : "v"(res), "s"(bit_));
        asm volatile("s_mov_b64 exec, %4\n\tv_fmac_f32_e32 %0, %2, %1\n\ts_not_b64 exec, exec\n\tv_fmac_f32_e32 %0, %3, %1\n\ts_mov_b64 exec, -1"
                     : "+v"(res) : "v"(areg[e]), "s"(ua), "s"(ub), "s"(lomask) : "scc");
      } else if (V == 2) {
        float u; asm volatile("ds_swizzle_b32 %0, %1 offset:%2" : "=v"(u) : "v"(cand), "n"((e & 31) << 5));
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(capr) : "v"(res), "s"(bit_));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(u));
        res = fmaf(areg[e], u, res);
      } else if (V == 3) {
        const float ua = bcast(cand, e), ub = bcast(cand, e + 32);
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(capr) : "v"(res), "s"(bit_));
        res = fmaf(areg[e], upper ? ub : ua, res);
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
  float h[5][64];
#define RUN(V, NR, THREADS, label) { hipLaunchKernelGGL((k<V, NR>), dim3(1), dim3(THREADS), 0, 0, out, cyc, sweeps); hipLaunchKernelGGL((k<V, NR>), dim3(1), dim3(THREADS), 0, 0, out, cyc, sweeps); \
  unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(h[V], out, 256, hipMemcpyDeviceToHost); \
  printf("%-64s %2d rows, %d wave(s)/SIMD: %6.1f ticks per turn (%5.1f per env-row)\n", label, NR, THREADS / 256 ? THREADS / 256 : 1, (double)c / sweeps / NR, (double)c / sweeps / NR / 2); }
  RUN(0, 24, 64, "0 zero-padded multiplicands, 64 registers (round 4)") RUN(0, 24, 512, "0 zero-padded multiplicands, 64 registers (round 4)")
  RUN(1, 24, 64, "1 exec-masked fmacs, 32 registers") RUN(1, 24, 512, "1 exec-masked fmacs, 32 registers")
  RUN(2, 24, 64, "2 ds_swizzle broadcast, 32 registers") RUN(2, 24, 512, "2 ds_swizzle broadcast, 32 registers")
  RUN(3, 24, 64, "3 select, 32 registers") RUN(3, 24, 512, "3 select, 32 registers")
  RUN(4, 24, 64, "4 as 0, the turn hand-scheduled in one asm block (7 slots)") RUN(4, 24, 512, "4 as 0, the turn hand-scheduled in one asm block (7 slots)")
  int bad = 0;
  for (int v = 1; v < 5; v++) for (int i = 0; i < 64; i++) bad += h[v][i] != h[0][i];
  printf("results differing from variant 0: %d of 256\n", bad);
  return 0;
}
