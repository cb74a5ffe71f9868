// Microbenchmark of the PAIR-mode PGS turn of the two-env constraint kernel (fmj_cons2_rows.inc): rows of env A in lanes 0..31, of
// env B in lanes 32..63, turn e serves row e of both.  How does a lane get ITS half's update?
//   0  as shipped in round 4: the lane's row of A split into two zero-padded multiplicands (64 registers), two readlanes, two fmacs
//   1  32 registers, two readlanes, the two fmacs under exec masks of the halves (three exec writes per turn)
//   2  32 registers, ONE ds_swizzle (bit-mask mode: and 0, or e = every lane reads lane e of its own group of 32), one fmac
//   3  32 registers, two readlanes, v_mov + v_cndmask select, one fmac
//   5  TWO rows per lane (rows l and l + 16 of the lane's env in lanes l and l + 16 alike): the update comes as a DPP row_newbcast
//      operand of the two fmacs - no readlane, no scalar: v_max, v_cndmask, two v_fmac_dpp
//   6  as 5, v_max and the capture under a two-lane exec mask written by the scalar unit (cand keeps each lane's own update)
//   10 three vector instructions per turn: v_max_f32_dpp under the bank mask latches the update itself (register e & 3 of the set), the two fmacs read it
//      through DPP; four turns are one asm statement.  The force comes from the latched update; no residual is captured (res + nf are compared)
//   8, 9  calibration: a chain of dependent v_fmac / eight independent chains (ticks per VALU instruction)
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
  float acc = 0.f, res1 = 0.f, nf1 = 0.f;
  const unsigned long long pairbase = 0x0000000100000001ull, lomask = 0x00000000ffffffffull, pairbase16 = 0x0001000100010001ull;
  if ((V >= 5 && V <= 7) || V == 10) {      // rows l and l + 16 into lanes l and l + 16 alike: v_permlane16_swap (gfx950) trades the odd rows of 16 lanes of its first
                               // operand for the even ones of its second
#pragma unroll
    for (int i = 0; i < 32; i++) { float y = areg[i]; asm volatile("v_permlane16_swap_b32_e32 %0, %1" : "+v"(areg[i]), "+v"(y)); areg[32 + i] = y; }
    res1 = res; asm volatile("v_permlane16_swap_b32_e32 %0, %1" : "+v"(res), "+v"(res1));
    nf1 = nf; asm volatile("v_permlane16_swap_b32_e32 %0, %1" : "+v"(nf), "+v"(nf1));
  }
  if (V == 8 || V == 9) {
    float c[8];
#pragma unroll
    for (int i = 0; i < 8; i++) c[i] = res + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < sweeps; it++) {
#pragma unroll
      for (int e = 0; e < 64; e++) {
        if (V == 8) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(c[0]) : "v"(nf), "v"(areg[e & 31]));
        else asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(c[e & 7]) : "v"(nf), "v"(areg[e & 31]));
      }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) r += c[i];
    out[threadIdx.x + blockIdx.x * blockDim.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    return;
  }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < sweeps; it++) {
    float capr = 0.f;
#pragma unroll
    for (int e = 0; e < NR; e++) {
      if ((V >= 5 && V <= 7) || V == 10) break;
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
    if (V == 10) {
      static_assert(V != 10 || NR == 24, "variant 10 is written for 24 rows");
      float dq[8];
#pragma unroll
      for (int i = 0; i < 8; i++) dq[i] = 0.f;
      // S = the set the bank's turns read, O = the other; l_i = the previous turn's column of O, a_i = the turn's column of S
#define T10(i, lp_, dp_) \
        "v_max_f32_dpp %[d" #i "], %[nf], %[S] quad_perm:[0,1,2,3] row_mask:0xf bank_mask:%[bm]\n\t" \
        "v_fmac_f32_dpp %[O], %[" dp_ "], %[l" #i "] row_newbcast:%[" lp_ "] row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 0\n\t" \
        "v_fmac_f32_dpp %[S], %[d" #i "], %[a" #i "] row_newbcast:%[e" #i "] row_mask:0xf bank_mask:0xf\n\t"
#define T10X(i, lp_, dp_) \
        "v_max_f32_dpp %[d" #i "], %[nf], %[S] quad_perm:[0,1,2,3] row_mask:0xf bank_mask:%[bm]\n\t" \
        "v_fmac_f32_dpp %[O], %[" dp_ "], %[l" #i "] row_newbcast:%[" lp_ "] row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 1\n\t" \
        "v_fmac_f32_dpp %[O], %[d" #i "], %[a" #i "] row_newbcast:%[e" #i "] row_mask:0xf bank_mask:0xf\n\t"
#define T10OPS(E0_, s_, XL_, dprev_) \
        : [S] "+v"(s_ ? res1 : res), [O] "+v"(s_ ? res : res1), [d0] "+v"(dq[4 * s_]), [d1] "+v"(dq[4 * s_ + 1]), [d2] "+v"(dq[4 * s_ + 2]), [d3] "+v"(dq[4 * s_ + 3]) \
        : [nf] "v"(s_ ? nf1 : nf), [dp] "v"(dprev_), [bm] "n"(1 << (((E0_) & 15) >> 2)), \
          [a0] "v"(areg[32 * s_ + (E0_)]), [a1] "v"(areg[32 * s_ + (E0_) + 1]), [a2] "v"(areg[32 * s_ + (E0_) + 2]), [a3] "v"(areg[32 * (XL_ ? 1 - s_ : s_) + (E0_) + 3]), \
          [l0] "v"(areg[32 * (1 - s_) + ((E0_) > 0 ? (E0_) - 1 : 0)]), [l1] "v"(areg[32 * (1 - s_) + (E0_)]), [l2] "v"(areg[32 * (1 - s_) + (E0_) + 1]), [l3] "v"(areg[32 * (1 - s_) + (E0_) + 2]), \
          [em] "n"(((E0_) + 15) & 15), [e0] "n"((E0_) & 15), [e1] "n"(((E0_) + 1) & 15), [e2] "n"(((E0_) + 2) & 15), [e3] "n"(((E0_) + 3) & 15)
      asm volatile("s_nop 1\n\tv_max_f32_dpp %[d0], %[nf], %[S] quad_perm:[0,1,2,3] row_mask:0xf bank_mask:%[bm]\n\ts_nop 1\n\tv_fmac_f32_dpp %[S], %[d0], %[a0] row_newbcast:%[e0] row_mask:0xf bank_mask:0xf\n\t"
                   T10(1, "e0", "d0") T10(2, "e1", "d1") T10(3, "e2", "d2") T10OPS(0, 0, 0, dq[0]));
      asm volatile(T10(0, "em", "dp") T10(1, "e0", "d0") T10(2, "e1", "d1") T10(3, "e2", "d2") T10OPS(4, 0, 0, dq[3]));
      asm volatile(T10(0, "em", "dp") T10(1, "e0", "d0") T10(2, "e1", "d1") T10(3, "e2", "d2") T10OPS(8, 0, 0, dq[3]));
      asm volatile(T10(0, "em", "dp") T10(1, "e0", "d0") T10(2, "e1", "d1") T10X(3, "e2", "d2") T10OPS(12, 0, 1, dq[3]));
      asm volatile(T10(0, "em", "dp") T10(1, "e0", "d0") T10(2, "e1", "d1") T10(3, "e2", "d2") T10OPS(16, 1, 0, dq[3]));
      asm volatile(T10(0, "em", "dp") T10(1, "e0", "d0") T10(2, "e1", "d1") T10X(3, "e2", "d2") T10OPS(20, 1, 1, dq[7]));
      asm volatile("s_nop 1\n\tv_fmac_f32_dpp %[rp], %[cp], %[ap] row_newbcast:7 row_mask:0xf bank_mask:0xf" : [rp] "+v"(res1) : [cp] "v"(dq[7]), [ap] "v"(areg[32 + 23]));
      const int q = lane & 3;
      float d0 = q == 0 ? dq[0] : q == 1 ? dq[1] : q == 2 ? dq[2] : dq[3];
      float d1 = q == 0 ? dq[4] : q == 1 ? dq[5] : q == 2 ? dq[6] : dq[7];
      if ((lane & 15) >= NR) d0 = 0.f;
      if ((lane & 15) + 16 >= NR) d1 = 0.f;
      nf -= d0; nf1 -= d1;
      continue;
    }
    if (V >= 5 && V <= 7) {
      // lanes l and l + 16 of a half both hold rows l (res, nf, areg[0..31]) and l + 16 (res1, nf1, areg[32..63]) - set up below
      float capr1 = 0.f;
      float cq[8], cnd[2] = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 8; i++) cq[i] = 0.f;
#pragma unroll
      for (int e = 0; e < NR; e++) {
        float cand; unsigned long long bit_;
        if (V == 6) {
          // the capture as a DPP move whose bank mask (an immediate) lets only the four lanes of lane e's bank write: register e & 3 of the row set
          // is the lane's own capture for lane & 3 == e & 3, junk elsewhere; picked after the sweep.  No scalar instruction in the turn
#define T6(RES_, NF_, CQ_, BM_) asm volatile("v_max_f32_e32 %[cand], %[nf], %[resx]\n\t" \
                         "v_mov_b32_dpp %[cq], %[resx] quad_perm:[0,1,2,3] row_mask:0xf bank_mask:" #BM_ "\n\t" \
                         "s_nop 0\n\t" \
                         "v_fmac_f32_dpp %[res], %[cand], %[a0] row_newbcast:%[el] row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f32_dpp %[res1], %[cand], %[a1] row_newbcast:%[el] row_mask:0xf bank_mask:0xf" \
                         : [cand] "=&v"(cand), [cq] "+v"(CQ_), [res] "+v"(res), [res1] "+v"(res1) \
                         : [nf] "v"(NF_), [resx] "v"(RES_), [el] "n"(e & 15), [a0] "v"(areg[e]), [a1] "v"(areg[32 + e]));
          const int b = (e & 15) >> 2;
          if (e < 16) { if (b == 0) T6(res, nf, cq[e & 3], 0x1) else if (b == 1) T6(res, nf, cq[e & 3], 0x2) else if (b == 2) T6(res, nf, cq[e & 3], 0x4) else T6(res, nf, cq[e & 3], 0x8) }
          else { if (b == 0) T6(res1, nf1, cq[4 + (e & 3)], 0x1) else if (b == 1) T6(res1, nf1, cq[4 + (e & 3)], 0x2) else if (b == 2) T6(res1, nf1, cq[4 + (e & 3)], 0x4) else T6(res1, nf1, cq[4 + (e & 3)], 0x8) }
        }
        if (V == 7) {
          // 4 issue slots per turn, no scalar, no s_nop: the capture is a bank-masked DPP move (as 6), and the fmac of the row set the NEXT turn
          // does not read is issued one turn late, between the next turn's v_max and its capture - so every DPP read comes two
          // instructions after the write of its register (v_max -> own fmac: the late fmac and the capture; fmac -> capture of that set: v_max and the late fmac)
          const int s_ = e < 16 ? 0 : 1, sn_ = (e + 1 < NR && e + 1 >= 16) ? 1 : 0;      // the set this turn reads; the set the next turn reads (sweep end: set 0)
          const int sp_ = 1 - s_;                                                       // the late fmac of turn e - 1 writes the set this turn does not read
#define T7(BM_) { if (e == 0) asm volatile("v_max_f32_e32 %[cc], %[nf], %[rs]\n\t" \
                         "v_mov_b32_dpp %[cq], %[rs] quad_perm:[0,1,2,3] row_mask:0xf bank_mask:" #BM_ "\n\t" \
                         "s_nop 0\n\t" \
                         "v_fmac_f32_dpp %[rn], %[cc], %[an] row_newbcast:%[el] row_mask:0xf bank_mask:0xf" \
                         : [cc] "=&v"(cnd[e & 1]), [cq] "+v"(cq[4 * s_ + (e & 3)]), [rn] "+v"(sn_ ? res1 : res) \
                         : [nf] "v"(s_ ? nf1 : nf), [rs] "v"(s_ ? res1 : res), [el] "n"(e & 15), [an] "v"(areg[32 * sn_ + e])); \
            else if (sn_ == sp_) asm volatile("v_max_f32_e32 %[cc], %[nf], %[rs]\n\t" \
                         "v_fmac_f32_dpp %[rn], %[cp], %[ap] row_newbcast:%[elp] row_mask:0xf bank_mask:0xf\n\t" \
                         "v_mov_b32_dpp %[cq], %[rs] quad_perm:[0,1,2,3] row_mask:0xf bank_mask:" #BM_ "\n\t" \
                         "v_fmac_f32_dpp %[rn], %[cc], %[an] row_newbcast:%[el] row_mask:0xf bank_mask:0xf" \
                         : [cc] "=&v"(cnd[e & 1]), [cq] "+v"(cq[4 * s_ + (e & 3)]), [rn] "+v"(sn_ ? res1 : res) \
                         : [nf] "v"(s_ ? nf1 : nf), [rs] "v"(s_ ? res1 : res), [el] "n"(e & 15), [an] "v"(areg[32 * sn_ + e]), \
                           [cp] "v"(cnd[(e + 1) & 1]), [elp] "n"((e + 15) & 15), [ap] "v"(areg[32 * sp_ + (e > 0 ? e - 1 : 0)])); \
            else asm volatile("v_max_f32_e32 %[cc], %[nf], %[rs]\n\t" \
                         "v_fmac_f32_dpp %[rp], %[cp], %[ap] row_newbcast:%[elp] row_mask:0xf bank_mask:0xf\n\t" \
                         "v_mov_b32_dpp %[cq], %[rs] quad_perm:[0,1,2,3] row_mask:0xf bank_mask:" #BM_ "\n\t" \
                         "v_fmac_f32_dpp %[rn], %[cc], %[an] row_newbcast:%[el] row_mask:0xf bank_mask:0xf" \
                         : [cc] "=&v"(cnd[e & 1]), [cq] "+v"(cq[4 * s_ + (e & 3)]), [rn] "+v"(sn_ ? res1 : res), [rp] "+v"(sp_ ? res1 : res) \
                         : [nf] "v"(s_ ? nf1 : nf), [rs] "v"(s_ ? res1 : res), [el] "n"(e & 15), [an] "v"(areg[32 * sn_ + e]), \
                           [cp] "v"(cnd[(e + 1) & 1]), [elp] "n"((e + 15) & 15), [ap] "v"(areg[32 * sp_ + (e > 0 ? e - 1 : 0)])); }
          const int b = (e & 15) >> 2;
          if (b == 0) T7(0x1) else if (b == 1) T7(0x2) else if (b == 2) T7(0x4) else T7(0x8)
          if (e == NR - 1)      // the sweep's last late fmac: the set the (absent) next turn would not read
            asm volatile("s_nop 0\n\tv_fmac_f32_dpp %[rp], %[cp], %[ap] row_newbcast:%[elp] row_mask:0xf bank_mask:0xf"
                         : [rp] "+v"(res1) : [cp] "v"(cnd[e & 1]), [elp] "n"(e & 15), [ap] "v"(areg[32 + e]));
        }
        if (V == 5) {
          if (e < 16)
            asm volatile("v_max_f32_e32 %[cand], %[nf], %[res]\n\t"
                         "s_lshl_b64 %[bit], %[pb], %[e]\n\t"
                         "v_cndmask_b32_e64 %[capr], %[capr], %[res], %[bit]\n\t"
                         "v_fmac_f32_dpp %[res], %[cand], %[a0] row_newbcast:%[el] row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f32_dpp %[res1], %[cand], %[a1] row_newbcast:%[el] row_mask:0xf bank_mask:0xf"
                         : [cand] "=&v"(cand), [bit] "=&s"(bit_), [capr] "+v"(capr), [res] "+v"(res), [res1] "+v"(res1)
                         : [nf] "v"(nf), [pb] "s"(pairbase16), [e] "n"(e), [el] "n"(e & 15), [a0] "v"(areg[e]), [a1] "v"(areg[32 + e]) : "scc");
          else
            asm volatile("v_max_f32_e32 %[cand], %[nf], %[res1]\n\t"
                         "s_lshl_b64 %[bit], %[pb], %[e]\n\t"
                         "v_cndmask_b32_e64 %[capr], %[capr], %[res1], %[bit]\n\t"
                         "v_fmac_f32_dpp %[res], %[cand], %[a0] row_newbcast:%[el] row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f32_dpp %[res1], %[cand], %[a1] row_newbcast:%[el] row_mask:0xf bank_mask:0xf"
                         : [cand] "=&v"(cand), [bit] "=&s"(bit_), [capr] "+v"(capr1), [res] "+v"(res), [res1] "+v"(res1)
                         : [nf] "v"(nf1), [pb] "s"(pairbase16), [e] "n"(e & 15), [el] "n"(e & 15), [a0] "v"(areg[e]), [a1] "v"(areg[32 + e]) : "scc");
        }
      }
      if (V == 6 || V == 7) {
        const int q = lane & 3;
        capr = q == 0 ? cq[0] : q == 1 ? cq[1] : q == 2 ? cq[2] : cq[3];
        capr1 = q == 0 ? cq[4] : q == 1 ? cq[5] : q == 2 ? cq[6] : cq[7];
        if ((lane & 15) >= NR) capr = 0.f;
        if ((lane & 15) + 16 >= NR) capr1 = 0.f;
      }
      float capc; asm("v_max_f32_e32 %0, %1, %2" : "=v"(capc) : "v"(nf), "v"(capr));
      float capc1; asm("v_max_f32_e32 %0, %1, %2" : "=v"(capc1) : "v"(nf1), "v"(capr1));
      if (it == 0 && blockIdx.x == 0 && threadIdx.x < 64) out[8192 + lane] = (lane & 16) ? capr1 : capr;
      nf -= capc; nf1 -= capc1; acc += ((lane & 16) ? capc1 * capr1 : capc * capr);
      continue;
    }
    float capc; asm("v_max_f32_e32 %0, %1, %2" : "=v"(capc) : "v"(nf), "v"(capr));
    nf -= capc; acc += capc * capr;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (((V >= 5 && V <= 7) || V == 10) && (lane & 16)) { res = res1; nf = nf1; }
  out[threadIdx.x + blockIdx.x * blockDim.x] = res + nf + acc;
  if (blockIdx.x == 0 && threadIdx.x < 64) { out[4096 + 3 * threadIdx.x] = res; out[4096 + 3 * threadIdx.x + 1] = nf; out[4096 + 3 * threadIdx.x + 2] = acc; }
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 4 * 65536); (void)hipMalloc(&cyc, 16);
  const int sweeps = 2000;
  float h[11][64]; static float h3[11][192]; static float hc[11][64];
#define RUN(V, NR, THREADS, label) { hipLaunchKernelGGL((k<V, NR>), dim3(1), dim3(THREADS), 0, 0, out, cyc, sweeps); hipLaunchKernelGGL((k<V, NR>), dim3(1), dim3(THREADS), 0, 0, out, cyc, sweeps); \
  unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); \
  hipLaunchKernelGGL((k<V, NR>), dim3(1), dim3(THREADS), 0, 0, out, cyc + 1, 3);      /* the results compared: three sweeps, far from the fixed point every variant reaches */ \
  (void)hipMemcpy(h[V], out, 256, hipMemcpyDeviceToHost); (void)hipMemcpy(h3[V], out + 4096, 768, hipMemcpyDeviceToHost); (void)hipMemcpy(hc[V], out + 8192, 256, hipMemcpyDeviceToHost); \
  printf("%-64s %2d rows, %d wave(s)/SIMD: %6.1f ticks per turn (%5.1f per env-row)\n", label, NR, THREADS / 256 ? THREADS / 256 : 1, (double)c / sweeps / NR, (double)c / sweeps / NR / 2); }
  RUN(0, 24, 64, "0 zero-padded multiplicands, 64 registers (round 4)") RUN(0, 24, 512, "0 zero-padded multiplicands, 64 registers (round 4)")
  RUN(1, 24, 64, "1 exec-masked fmacs, 32 registers") RUN(1, 24, 512, "1 exec-masked fmacs, 32 registers")
  RUN(2, 24, 64, "2 ds_swizzle broadcast, 32 registers") RUN(2, 24, 512, "2 ds_swizzle broadcast, 32 registers")
  RUN(3, 24, 64, "3 select, 32 registers") RUN(3, 24, 512, "3 select, 32 registers")
  RUN(4, 24, 64, "4 as 0, the turn hand-scheduled in one asm block (7 slots)") RUN(4, 24, 512, "4 as 0, the turn hand-scheduled in one asm block (7 slots)")
  RUN(5, 24, 64, "5 two rows per lane, DPP row_newbcast operands (4 VALU)") RUN(5, 24, 512, "5 two rows per lane, DPP row_newbcast operands (4 VALU)")
  RUN(6, 24, 64, "6 as 5, capture = DPP move under a bank mask, s_nop (no scalar)") RUN(6, 24, 512, "6 as 5, capture = DPP move under a bank mask, s_nop (no scalar)")
  RUN(7, 24, 64, "7 as 6, the second row set's fmac one turn late (4 slots)") RUN(7, 24, 512, "7 as 6, the second row set's fmac one turn late (4 slots)")
  RUN(10, 24, 64, "10 three vector instructions: the bank-masked v_max_dpp latches the update") RUN(10, 24, 512, "10 three vector instructions: the bank-masked v_max_dpp latches the update")
  RUN(8, 64, 64, "8 calibration: dependent v_fmac chain (ticks per instruction)") RUN(8, 64, 512, "8 calibration: dependent v_fmac chain")
  RUN(9, 64, 64, "9 calibration: eight independent v_fmac chains") RUN(9, 64, 512, "9 calibration: eight independent v_fmac chains")
  int bad = 0;
  for (int v = 1; v < 8; v++) for (int i = 0; i < 64; i++) { bad += h[v][i] != h[0][i]; if (h[v][i] != h[0][i] && v == 7) printf("variant %d lane %d: %.9g against %.9g  res %.9g/%.9g nf %.9g/%.9g acc %.9g/%.9g\n", v, i, h[v][i], h[0][i], h3[v][3*i], h3[5][3*i], h3[v][3*i+1], h3[5][3*i+1], h3[v][3*i+2], h3[5][3*i+2]); }
  for (int i = 0; i < 64; i++) if (hc[7][i] != hc[5][i]) printf("capture of sweep 0, lane %d: %.9g (7) %.9g (6) against %.9g (5)\n", i, hc[7][i], hc[6][i], hc[5][i]);
  { int b10 = 0; for (int i = 0; i < 64; i++) b10 += (h3[10][3 * i] != h3[0][3 * i]) || (h3[10][3 * i + 1] != h3[0][3 * i + 1]); printf("variant 10: lanes whose residual or force differ from variant 0: %d of 64\n", b10); }
  printf("results differing from variant 0: %d of 448\n", bad);
  return 0;
}
