// Microbenchmark of one elimination round of the two-env factorisation (fmj_dual.inc, L phase): members publish
// 20-float rows + 1/D, every lane reads the three member rows and applies them.  One wave, ticks per round.
#include <hip/hip_runtime.h>
#include <cstdio>
#define RS 20
#define NR 14
typedef float f2_t __attribute__((ext_vector_type(2)));
struct Round { int p0, p1, p2, depth; unsigned long long anc[3]; unsigned long long desc[3]; };
typedef const Round __attribute__((address_space(4)))* cround_p;
__device__ __forceinline__ float mask_select(const float v, const unsigned long long m) { float r; asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(m)); return r; }
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
template <int V> __global__ void k(float* out, unsigned long long* cyc, const Round* rounds, int iters) {
  __shared__ __align__(16) float lds[2 * (32 * RS + 32)];
  const int lane = threadIdx.x, sl = lane & 31; const bool upper = lane >= 32;
  float* HR = lds + (upper ? 32 * RS + 32 : 0); float* XV = HR + 32 * RS;
  for (int i = sl; i < 32 * RS + 32; i += 32) HR[i] = 1e-3f * (i % 17);
  f2_t r[RS / 2];
  for (int d = 0; d < RS / 2; d++) r[d] = f2_t{1.f + lane, 2.f};
  float diag = 3.f + lane; const int ddepth = sl % 17;
  const cround_p RND = (cround_p)rounds;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    int p0 = RND[0].p0, p1 = RND[0].p1, p2 = RND[0].p2;
    unsigned long long a0 = RND[0].anc[0], a1 = RND[0].anc[1], a2 = RND[0].anc[2];
#pragma unroll 1
    for (int rd = 0; rd < NR; rd++) {
      int np0, np1, np2; unsigned long long na0, na1, na2;
      if (V == 0) { const int rn = rd + 1 < NR ? rd + 1 : rd; np0 = RND[rn].p0; np1 = RND[rn].p1; np2 = RND[rn].p2; na0 = RND[rn].anc[0]; na1 = RND[rn].anc[1]; na2 = RND[rn].anc[2]; }
      else { np0 = (p0 + 3) & 31; np1 = (p1 + 3) & 31; np2 = (p2 + 3) & 31; na0 = a0 * 3; na1 = a1 * 5; na2 = a2 * 7; }   // no memory
      if (V == 11) {}
      else if (V >= 7 && V <= 9) {
        const float dinv = __builtin_amdgcn_rcpf(diag);
#pragma unroll
        for (int d = 0; d < RS; d += 4) *(float4*)(HR + sl * RS + d) = make_float4(r[d / 2].x, r[d / 2].y, r[d / 2 + 1].x, r[d / 2 + 1].y);
        XV[sl] = dinv;
      }
      else if (sl == p0 || sl == p1 || sl == p2) {
        const float dinv = __builtin_amdgcn_rcpf(diag);
#pragma unroll
        for (int d = 0; d < RS; d += 4) *(float4*)(HR + sl * RS + d) = make_float4(r[d / 2].x, r[d / 2].y, r[d / 2 + 1].x, r[d / 2 + 1].y);
        XV[sl] = dinv;
      }
      WSYNC();
#define APPLY(p_, am_) do { const float tk_ = HR[(p_) * RS + ddepth]; const float dki_ = XV[p_]; float4 rk_[RS / 4]; \
        _Pragma("unroll") for (int g = 0; g < RS / 4; g++) rk_[g] = *(const float4*)(HR + (p_) * RS + 4 * g); \
        const float t_ = mask_select(tk_ * dki_, am_); const f2_t nt_ = f2_t{-t_, -t_}; \
        _Pragma("unroll") for (int g = 0; g < RS / 4; g++) { r[2 * g] = __builtin_elementwise_fma(nt_, f2_t{rk_[g].x, rk_[g].y}, r[2 * g]); \
          r[2 * g + 1] = __builtin_elementwise_fma(nt_, f2_t{rk_[g].z, rk_[g].w}, r[2 * g + 1]); } diag = fmaf(-t_, tk_, diag); } while (0)
#define APPLYN(NG_, p_, am_) do { const float tk_ = HR[(p_) * RS + ddepth]; const float dki_ = XV[p_]; float4 rk_[NG_]; \
        _Pragma("unroll") for (int g = 0; g < NG_; g++) rk_[g] = *(const float4*)(HR + (p_) * RS + 4 * g); \
        const float t_ = mask_select(tk_ * dki_, am_); const f2_t nt_ = f2_t{-t_, -t_}; \
        _Pragma("unroll") for (int g = 0; g < NG_; g++) { r[2 * g] = __builtin_elementwise_fma(nt_, f2_t{rk_[g].x, rk_[g].y}, r[2 * g]); \
          r[2 * g + 1] = __builtin_elementwise_fma(nt_, f2_t{rk_[g].z, rk_[g].w}, r[2 * g + 1]); } diag = fmaf(-t_, tk_, diag); } while (0)
#define ROUNDN(NG_) do { APPLYN(NG_, p0, a0); if (p1 >= 0) { APPLYN(NG_, p1, a1); APPLYN(NG_, p2, a2); } } while (0)
      if (V == 3) { const int ng = (RND[rd].depth + 3) >> 2; if (ng <= 1) ROUNDN(1); else if (ng == 2) ROUNDN(2); else if (ng == 3) ROUNDN(3); else if (ng == 4) ROUNDN(4); else ROUNDN(5); }
      else if (V == 4) { ROUNDN(3); }
      else if (V == 5) { /* scalar fmas instead of pk */ 
        const float tk_ = HR[p0 * RS + ddepth]; const float dki_ = XV[p0]; float4 rk_[5];
        _Pragma("unroll") for (int g = 0; g < 5; g++) rk_[g] = *(const float4*)(HR + p0 * RS + 4 * g);
        const float t_ = -mask_select(tk_ * dki_, a0);
        _Pragma("unroll") for (int g = 0; g < 5; g++) { r[2*g].x = fmaf(t_, rk_[g].x, r[2*g].x); r[2*g].y = fmaf(t_, rk_[g].y, r[2*g].y); r[2*g+1].x = fmaf(t_, rk_[g].z, r[2*g+1].x); r[2*g+1].y = fmaf(t_, rk_[g].w, r[2*g+1].y); }
        diag = fmaf(t_, tk_, diag); }
      else if (V == 6) { APPLY(p0, a0); }
      else if (V == 7 || V == 11) { APPLY(p0, a0); APPLY(p1 < 0 ? p0 : p1, a1); APPLY(p2, a2); }
      else if (V == 8) {
        const int q1 = p1 < 0 ? p0 : p1;
        const float tk0 = HR[p0 * RS + ddepth], tk1 = HR[q1 * RS + ddepth], tk2 = HR[p2 * RS + ddepth];
        const float dk0 = XV[p0], dk1 = XV[q1], dk2 = XV[p2];
        float4 r0[5], r1[5], r2[5];
        _Pragma("unroll") for (int g = 0; g < 5; g++) { r0[g] = *(const float4*)(HR + p0 * RS + 4 * g); r1[g] = *(const float4*)(HR + q1 * RS + 4 * g); r2[g] = *(const float4*)(HR + p2 * RS + 4 * g); }
        const float t0 = mask_select(tk0 * dk0, a0), t1 = mask_select(tk1 * dk1, a1), t2 = mask_select(tk2 * dk2, a2);
        const f2_t n0 = f2_t{-t0, -t0}, n1 = f2_t{-t1, -t1}, n2 = f2_t{-t2, -t2};
        _Pragma("unroll") for (int g = 0; g < 5; g++) {
          r[2*g] = __builtin_elementwise_fma(n0, f2_t{r0[g].x, r0[g].y}, r[2*g]); r[2*g+1] = __builtin_elementwise_fma(n0, f2_t{r0[g].z, r0[g].w}, r[2*g+1]);
          r[2*g] = __builtin_elementwise_fma(n1, f2_t{r1[g].x, r1[g].y}, r[2*g]); r[2*g+1] = __builtin_elementwise_fma(n1, f2_t{r1[g].z, r1[g].w}, r[2*g+1]);
          r[2*g] = __builtin_elementwise_fma(n2, f2_t{r2[g].x, r2[g].y}, r[2*g]); r[2*g+1] = __builtin_elementwise_fma(n2, f2_t{r2[g].z, r2[g].w}, r[2*g+1]); }
        diag = fmaf(-t0, tk0, fmaf(-t1, tk1, fmaf(-t2, tk2, diag))); }
      else if (V == 9) { APPLYN(3, p0, a0); APPLYN(3, p1 < 0 ? p0 : p1, a1); APPLYN(3, p2, a2); }
      else if (V == 10) { diag += HR[p0 * RS + ddepth] * 1e-9f; }
      else {
      APPLY(p0, a0);
      if (V == 2 || p1 >= 0) { APPLY(p1 < 0 ? p0 : p1, a1); APPLY(p2, a2); }
      }
      WSYNC();
      p0 = np0; p1 = np1; p2 = np2; a0 = na0; a1 = na1; a2 = na2;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = diag; for (int d = 0; d < RS / 2; d++) acc += r[d].x + r[d].y;
  out[lane] = acc; if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc; Round* rd; Round h[NR];
  (void)hipMalloc(&out, 256); (void)hipMalloc(&cyc, 8); (void)hipMalloc(&rd, sizeof h);
  for (int r = 0; r < NR; r++) { h[r].p0 = (3 * r) % 30; h[r].p1 = (r >= 4 && r < 12) ? (3 * r + 1) % 30 : -1; h[r].p2 = (r >= 4 && r < 12) ? (3 * r + 2) % 30 : h[r].p0;
    h[r].depth = 16 - r; for (int c = 0; c < 3; c++) { h[r].anc[c] = 0x000000ff000000ffull << (r % 8); h[r].desc[c] = 0; } }
  (void)hipMemcpy(rd, h, sizeof h, hipMemcpyHostToDevice);
  const char* names[] = {"records via s_load (as shipped)", "records computed in SALU (no SMEM)", "no SMEM, always three members", "groups by depth (switch)", "3 of 5 groups always", "single member, scalar FMAs", "single member, pk FMAs", "all lanes publish, 3 members", "all publish, 3 members, loads first", "all publish, 3 members, 3 groups", "publish + one read only", "apply only (no publish)"};
  const int iters = 200;
#define RUN(V) { hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, out, cyc, rd, iters); hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, out, cyc, rd, iters); \
  unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-40s %.1f ticks/round\n", names[V], (double)c / iters / NR); }
  RUN(2) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11)
  return 0;
}
