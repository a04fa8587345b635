// Internal interface of the LAIR weights / resampling / imputation stage (cs_lair.hip: proposal sets of up to
// 64 per example, one wavefront per group of examples; cs_lair_large.hip: larger sets, three launches).
#pragma once
#include "cs_mlp.h"

struct CsDiv { uint32_t mul, d; };
struct CsResampleArgs {
  cs_lair_cfg cfg;
  cs_lair_bufs b;
  CsPhilox rng;
  int Kp, S, G, c;       // resampling groups: G groups, c draws each
  int mixG, mixc;        // mixture groups (dmis: 1,K'; within groups: G,c; smis: K',1)
  int Kimp;              // chains that are imputed (<= K: historical proposals are not)
  float* x_copy;         // optional [B,Kimp,D]: the new imputations are also written here (a run that keeps them)
  int NS;                // output slots of the resampling = G*c (K', or glr_pad_to when padded)
  int lds_per_wave;      // floats
  int E;                 // examples per wavefront
  int P, logP;           // lanes per example: the power of two >= S (lane = e*P + s)
  int q_stride;          // floats between the staged proposal parameters of consecutive examples
  int z_stride;          // floats between staged latents of consecutive proposals
  CsDiv dZd, dZq, dS, dKp, dNS, dDQ, dKimp, dMC, dmixG, dc;  // divisors of the index arithmetic (Zq = Zd/4)
};

// Division of a small index by a launch constant: floor(t / d) = hi32(t * ceil(2^32 / d)), exact
// while t * d < 2^32 (the host checks the ranges).  An integer division by a run-time value costs
// ~40 instructions, and the staging / imputation loops of this kernel did three per element.
__device__ __forceinline__ int cs_div(int t, const CsDiv& dv) {
  return dv.d == 1u ? t : (int)__umulhi((unsigned)t, dv.mul);
}
static inline CsDiv cs_mkdiv(int d) {
  CsDiv v;
  v.d = (uint32_t)d;
  v.mul = d > 1 ? (uint32_t)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)d) : 0u;
  return v;
}

// The stage for proposal sets of more than 64 per example (cs_lair_large.hip); `a` has Kp, S, G, c, mixG, mixc,
// Kimp, NS and the configuration / buffers filled in.
int cs_lair_resample_large(CsResampleArgs& a, hipStream_t stream);

#ifdef __HIPCC__
// Sampling of the missing values from the resampled generator parameters and imputation
// (vae_imputation.py:682-685, 724; clipping test_step_vae_sampling.py:1942-1950): one work item = 4 consecutive
// dimensions 4*dq.. of chain n of example be (one Philox block = 4 normals), `src` the flattened '(i k)' index of
// the proposal resampled into the chain.  Only the first Kimp chains are imputed (the historical ones beyond them
// only propose); the new values go to X_out[B,Kimp,D] when given, else in place.
__device__ __forceinline__ void cs_lair_impute_item(const CsResampleArgs& a, int64_t be, int n, int dq, int src) {
  const int B = a.cfg.B, K = a.cfg.K, D = a.cfg.D, Kp = a.Kp, Kimp = a.Kimp;
  const bool bern = a.cfg.gen_dist == CS_DIST_BERN;
  const bool stt = cs_dist_is_studentt(a.cfg.gen_dist);
  const bool other = a.cfg.gen_dist > CS_DIST_BERN;  // Student-t, logit-normal
  const int np = cs_gen_nparams(a.cfg.gen_dist);
  const bool clip = a.b.clip_min || a.b.clip_max;
  const int i = cs_div(src, a.dKp), kp = src - i * Kp;
  const int64_t rho = ((int64_t)i * B + be) * Kp + kp;
  const float* xr = a.b.X + (be * K + n) * D;
  float* xw = a.b.X_out ? a.b.X_out + (be * Kimp + n) * D : a.b.X + (be * K + n) * D;
  const uint8_t* mr = a.b.M + be * D;
  // every input of the item is requested before anything is computed or stored: X is updated in
  // place, so a load placed after a store could not be moved ahead of it, and the item would pay
  // one memory round trip per dimension
  const float* gr = a.b.gen + rho * np * D;
  typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte access at 4-byte alignment
  const bool full = 4 * dq + 4 <= D;
  f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0, xo = g0, dr = g0;
  f32x4 g2 = {1.f, 1.f, 1.f, 1.f}, dc = g2;  // Student-t: degrees of freedom, chi-square draws
  float mk[4];
  if (full) {
    g0 = *(const f32x4_u*)(gr + 4 * dq);
    if (!bern) g1 = *(const f32x4_u*)(gr + D + 4 * dq);
    if (stt) g2 = *(const f32x4_u*)(gr + 2 * D + 4 * dq);
    xo = *(const f32x4_u*)(xr + 4 * dq);
    if (a.b.eps_x) dr = *(const f32x4_u*)(a.b.eps_x + (be * Kimp + n) * D + 4 * dq);
    if (stt && a.b.chi2_x) dc = *(const f32x4_u*)(a.b.chi2_x + (be * Kimp + n) * D + 4 * dq);
    uint32_t mb;
    __builtin_memcpy(&mb, mr + 4 * dq, 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) mk[r] = (float)((mb >> (8 * r)) & 0xffu);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int d = 4 * dq + r;
      const bool in = d < D;
      mk[r] = in ? (float)mr[d] : 1.0f;
      if (in) {
        g0[r] = gr[d];
        if (!bern) g1[r] = gr[D + d];
        if (stt) g2[r] = gr[2 * D + d];
        xo[r] = xr[d];
        if (a.b.eps_x) dr[r] = a.b.eps_x[(be * Kimp + n) * D + d];
        if (stt && a.b.chi2_x) dc[r] = a.b.chi2_x[(be * Kimp + n) * D + d];
      }
    }
  }
  if (!a.b.eps_x) {
    uint32_t x[4];
    cs_philox_block(a.rng, CS_STREAM_X, (a.rng.row_offset + be) * Kimp + n, (uint32_t)dq, x);
    if (bern) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dr[r] = cs_u01f(x[r]);
    } else {
      dr = cs_normal4(x);
    }
  }
  if (stt && !a.b.chi2_x) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      dc[r] = cs_chi2_sample(a.rng, CS_STREAM_X, (a.rng.row_offset + be) * Kimp + n, (uint32_t)(4 * dq + r), g2[r]);
  }
  f32x4 xn;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int d = 4 * dq + r;
    float xm;
    if (bern) xm = (dr[r] < cs_sigmoid(g0[r])) ? 1.0f : 0.0f;
    else if (other) xm = cs_gen_sample(a.cfg.gen_dist, g0[r], g1[r], g2[r], dr[r], dc[r]);
    else xm = g0[r] + g1[r] * dr[r];
    xn[r] = xo[r] * mk[r] + xm * (1.0f - mk[r]);
    if (d < D) {
      if (a.b.clip_min) xn[r] = (xn[r] != xn[r]) ? xn[r] : fmaxf(xn[r], a.b.clip_min[d]);
      if (a.b.clip_max) xn[r] = (xn[r] != xn[r]) ? xn[r] : fminf(xn[r], a.b.clip_max[d]);
      if (clip && !(fabsf(xn[r]) < INFINITY)) xn[r] = xo[r];
    }
  }
  float* xc = a.x_copy ? a.x_copy + (be * Kimp + n) * D : nullptr;
  if (full) {
    *(f32x4_u*)(xw + 4 * dq) = xn;
    if (xc) *(f32x4_u*)(xc + 4 * dq) = xn;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * dq + r < D) {
        xw[4 * dq + r] = xn[r];
        if (xc) xc[4 * dq + r] = xn[r];
      }
  }
}
#endif
