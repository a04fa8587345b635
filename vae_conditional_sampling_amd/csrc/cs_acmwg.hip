// AC-MWG step (TestVAEAdaptiveCollapsedMetropolisWithinGibbs.sampler,
// irwg/sampling/test_step_vae_sampling.py:1664-1768) and its imputation history
// (ImputationHistoryQueue_with_restrictedavailability, irwg/sampling/utils.py:83-237).
//
// One step = 4 launches on one stream:
//   prepare   per chain: history index draw -> ring slot -> row offset for the encoder's
//             gather; prior-mixture Bernoulli; prior proposal                (:1666-1668,1687-1691)
//   encoder   fused MLP on the gathered historical imputations -> q           (:1671,1683)
//   decoder   Z_new = loc + scale*eps (or the prior proposal), fused MLP, generator
//             parameterisation, masked log-likelihood of the current X         (:1684,1718-1725)
//   accept    proposal log-densities (+ mixture with the prior), acceptance ratio, u < alpha,
//             availability update, multiply-blend of the chain state, x_m sampling, in-place
//             imputation, ring write, new log-likelihood                       (:1693-1755)
// The ring [T,B,K,D] and the availability counters [B,K] live in HBM; the host only tracks the
// tail slot and the "full" flag, which evolve deterministically.
//
// The same kernels serve the plain Metropolis-within-Gibbs sampler
// (TestVAEMetropolisWithinGibbs.sampler, :1219-1282) through cfg.flags: the proposal conditions on
// the current imputation (no history) and the log-likelihoods sum over all dimensions.
#include "cs_mlp.h"

#include <cstring>

struct CsAcmwgArgs {
  cs_acmwg_cfg cfg;
  cs_acmwg_bufs b;
  CsPhilox rng;
  float* x_copy;   // optional [B,K,D]: the accept kernel writes the new imputations here too (a run that keeps them)
};

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- prepare: one thread per (chain, quad of latent dimensions) ----
// Quad 0 of a chain draws the history slot and writes the chain's flags; every quad of a chain that proposes from the
// prior draws its own 4 normals (one counter-RNG block; with one thread per chain a block was evaluated once per
// element, 32 in a row for a chain that took the prior, and set the launch's time at any batch size).
__global__ void __launch_bounds__(256) cs_acmwg_prepare_kernel(const CsAcmwgArgs a, int ZQ) {
  const int64_t gt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int B = a.cfg.B, K = a.cfg.K, D = a.cfg.D, Zd = a.cfg.Zd;
  const int64_t t = gt / ZQ;
  const int q = (int)(gt - t * ZQ);
  if (t >= (int64_t)B * K) return;
  const int64_t grow = a.rng.row_offset * K + t;  // global chain row
  if (q == 0 && !(a.cfg.flags & CS_MWG_PROPOSE_FROM_CURRENT)) {
    // uniform index among the available history entries (utils.py:219-237)
    int64_t bound = a.b.avail[t];
    if (bound < 1) bound = 1;
    int64_t r;
    if (a.b.r_hist) {
      r = a.b.r_hist[t];
    } else {
      uint32_t x[4];
      cs_philox_block(a.rng, CS_STREAM_HIST, grow, 0, x);
      r = (int64_t)(((uint64_t)x[0] * (uint64_t)bound) >> 32);
    }
    int64_t slot = r;
    if (a.cfg.hist_full) slot = (r + a.cfg.hist_tail + 1) % a.cfg.hist_len;
    a.b.slots[t] = slot;
    a.b.in_off[t] = (slot * B * K + t) * D;
  }
  // prior-mixture mask and prior proposal (:1687-1691)
  uint8_t up = 0;
  if (a.cfg.prior_mixture_probability > 0.0) {
    const float u = a.b.u_prior ? a.b.u_prior[t] : cs_uniform_elem(a.rng, CS_STREAM_UPRIOR, grow, 0);
    up = u < (float)a.cfg.prior_mixture_probability;
    if (up) {
      const int d0 = 4 * q;
      if (a.b.z_prior) {
        for (int d = d0; d < d0 + 4 && d < Zd; ++d) a.b.z_alt[t * Zd + d] = a.b.z_prior[t * Zd + d];
      } else {
        uint32_t x[4];
        cs_philox_block(a.rng, CS_STREAM_ZPRIOR, grow, (uint32_t)q, x);
        const f32x4 n = cs_normal4(x);
        for (int d = d0; d < d0 + 4 && d < Zd; ++d) a.b.z_alt[t * Zd + d] = n[d - d0];
      }
    }
  }
  if (q == 0) a.b.use_prior[t] = up;
}

// ---- accept: LP lanes per chain (LP a power of two, 64 / LP chains per wavefront) ----
// A lane owns quads of 4 consecutive dimensions (quad l, l + LP, ... of the chain's latent / data
// row): one counter-RNG block yields the 4 normals (or uniforms) of a quad, so no draw is computed
// twice, and sums over a chain are log2(LP) butterfly steps.
template <int LP>
__global__ void __launch_bounds__(256) cs_acmwg_accept_kernel(const CsAcmwgArgs a) {
  const int lane = threadIdx.x & 63, l = lane & (LP - 1);
  const int64_t t = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (64 / LP) + (lane / LP);
  const int B = a.cfg.B, K = a.cfg.K, D = a.cfg.D, Zd = a.cfg.Zd;
  const bool live = t < (int64_t)B * K;
  const int64_t tc = live ? t : 0;  // dead lanes compute on chain 0 and store nothing
  const int64_t b = tc / K;
  const int64_t grow = a.rng.row_offset * K + tc;
  const double p = a.cfg.prior_mixture_probability;
  const bool bern = a.cfg.gen_dist == CS_DIST_BERN;
  const bool other = a.cfg.gen_dist > CS_DIST_BERN;          // Student-t, logit-normal
  const bool stt = cs_dist_is_studentt(a.cfg.gen_dist);
  const int P = cs_gen_nparams(a.cfg.gen_dist) * D;
  const bool use_hist = !(a.cfg.flags & CS_MWG_PROPOSE_FROM_CURRENT);
  const bool joint_old = (a.cfg.flags & CS_MWG_JOINT_TARGET_OLD) != 0;
  auto csum = [&](float v) {
#pragma unroll
    for (int o = LP >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
  };

  // proposal / prior log-densities of Z_new and Z_old (:1693-1715)
  float zn_q = 0.f, zo_q = 0.f, zn_p = 0.f, zo_p = 0.f;
  const float* q = a.b.q + tc * 2 * Zd;
  const float* Zn = a.b.Z_new + tc * Zd;
  float* Zo = a.b.Z_old + tc * Zd;
  typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte access at 4-byte alignment
  for (int d0 = 4 * l; d0 < Zd; d0 += 4 * LP) {
    f32x4 vloc = {0.f, 0.f, 0.f, 0.f}, vsc = {1.f, 1.f, 1.f, 1.f}, vzn = vloc, vzo = vloc;
    if (d0 + 4 <= Zd) {
      vloc = *(const f32x4_u*)(q + d0); vsc = *(const f32x4_u*)(q + Zd + d0);
      vzn = *(const f32x4_u*)(Zn + d0); vzo = *(const f32x4_u*)(Zo + d0);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (d0 + r < Zd) { vloc[r] = q[d0 + r]; vsc[r] = q[Zd + d0 + r]; vzn[r] = Zn[d0 + r]; vzo[r] = Zo[d0 + r]; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int d = d0 + r;
      if (d < Zd) {
        const float loc = vloc[r], sc = vsc[r];
        const float zn = vzn[r], zo = vzo[r];
        zn_q += cs_normal_logpdf(zn, loc, sc);
        zo_q += cs_normal_logpdf(zo, loc, sc);
        zn_p += cs_stdnormal_logpdf(zn);
        zo_p += cs_stdnormal_logpdf(zo);
      }
    }
  }
  zn_q = csum(zn_q); zo_q = csum(zo_q); zn_p = csum(zn_p); zo_p = csum(zo_p);
  const float zn_prior = zn_p;
  if (p > 0.0) {
    const float lp = logf((float)p), lq = logf((float)(1.0 - p));
    {
      const float x0 = zn_q + lq, x1 = zn_p + lp;
      float m = fmaxf(x0, x1);
      if (!(fabsf(m) < INFINITY)) m = 0.f;
      zn_q = logf(expf(x0 - m) + expf(x1 - m)) + m;
    }
    {
      const float x0 = zo_q + lq, x1 = zo_p + lp;
      float m = fmaxf(x0, x1);
      if (!(fabsf(m) < INFINITY)) m = 0.f;
      zo_q = logf(expf(x0 - m) + expf(x1 - m)) + m;
    }
  }
  const float zold_prior = a.b.z_old_prior_lp[tc];
  const float log_accept = (a.b.ll_new[tc] + zn_prior) - (a.b.x_old_lp[tc] + zold_prior) + (zo_q - zn_q);
  const float u = a.b.u_accept ? a.b.u_accept[tc] : cs_uniform_elem(a.rng, CS_STREAM_UACC, grow, 0);
  const bool acc = u < expf(log_accept);
  const float fa = acc ? 1.0f : 0.0f, fn = acc ? 0.0f : 1.0f;

  if (live && l == 0) {
    a.b.accepted[t] = acc;
    a.b.log_accept[t] = log_accept;
    a.b.z_old_prior_lp[t] = zold_prior * fn + zn_prior * fa;
    if (use_hist) {
      // availability (utils.py:94-104): accepted chains see the whole history, then the enqueue
      // of a full ring ages every chain by one
      int64_t av = a.b.avail[t];
      const int64_t len = a.cfg.hist_full ? a.cfg.hist_len : a.cfg.hist_tail + 1;
      if (acc) av = len;
      if (a.cfg.hist_full) av = av > 0 ? av - 1 : 0;
      a.b.avail[t] = av;
    }
  }
  // multiply-blend of the chain state (:1739-1743)
  if (live) {
    for (int d0 = 4 * l; d0 < Zd; d0 += 4 * LP) {
      if (d0 + 4 <= Zd) {
        const f32x4 zo = *(const f32x4_u*)(Zo + d0), zn = *(const f32x4_u*)(Zn + d0);
        *(f32x4_u*)(Zo + d0) = zo * fn + zn * fa;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (d0 + r < Zd) Zo[d0 + r] = Zo[d0 + r] * fn + Zn[d0 + r] * fa;
      }
    }
  }
  // x_m ~ p(x_m | z), imputation, ring write, new masked log-likelihood (:1744-1755)
  float* ring_row = nullptr;
  uint8_t* ring_row8 = nullptr;  // binary ring: holds x != 0
  if (use_hist) {
    const int new_tail = (a.cfg.hist_tail + 1) % a.cfg.hist_len;
    const int64_t e = ((int64_t)new_tail * B * K + tc) * D;
    if (a.cfg.ring_u8) ring_row8 = (uint8_t*)a.b.ring + e;
    else ring_row = a.b.ring + e;
  }
  float* go = a.b.gen_old + tc * P;
  const float* gn = a.b.gen_new + tc * P;
  float* Xr = a.b.X + tc * D;
  float* Xc = a.x_copy ? a.x_copy + tc * D : nullptr;
  const uint8_t* Mr = a.b.M + b * D;
  float ll = 0.f;
  for (int d0 = 4 * l; d0 < D; d0 += 4 * LP) {
    // inputs of the quad first (X and the generator parameters are updated in place)
    const bool full = d0 + 4 <= D;
    f32x4 xo = {0.f, 0.f, 0.f, 0.f}, g0o = xo, g0n = xo, g1o = {1.f, 1.f, 1.f, 1.f}, g1n = g1o, dr = xo;
    f32x4 g2o = g1o, g2n = g1o, dc = g1o;  // Student-t: degrees of freedom (old, new), chi-square draws
    float m[4];
    if (full) {
      xo = *(const f32x4_u*)(Xr + d0);
      g0o = *(const f32x4_u*)(go + d0); g0n = *(const f32x4_u*)(gn + d0);
      if (!bern) { g1o = *(const f32x4_u*)(go + D + d0); g1n = *(const f32x4_u*)(gn + D + d0); }
      if (stt) { g2o = *(const f32x4_u*)(go + 2 * D + d0); g2n = *(const f32x4_u*)(gn + 2 * D + d0); }
      if (a.b.eps_x) dr = *(const f32x4_u*)(a.b.eps_x + tc * D + d0);
      if (stt && a.b.chi2_x) dc = *(const f32x4_u*)(a.b.chi2_x + tc * D + d0);
      uint32_t mb;
      __builtin_memcpy(&mb, Mr + d0, 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) m[r] = (float)((mb >> (8 * r)) & 0xffu);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = d0 + r;
        const bool in = d < D;
        m[r] = in ? (float)Mr[d] : 1.0f;
        if (in) {
          xo[r] = Xr[d]; g0o[r] = go[d]; g0n[r] = gn[d];
          if (!bern) { g1o[r] = go[D + d]; g1n[r] = gn[D + d]; }
          if (stt) { g2o[r] = go[2 * D + d]; g2n[r] = gn[2 * D + d]; }
          if (a.b.eps_x) dr[r] = a.b.eps_x[tc * D + d];
          if (stt && a.b.chi2_x) dc[r] = a.b.chi2_x[tc * D + d];
        }
      }
    }
    if (!a.b.eps_x) {
      uint32_t x[4];
      cs_philox_block(a.rng, CS_STREAM_X, grow, (uint32_t)(d0 >> 2), x);
      if (bern) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dr[r] = cs_u01f(x[r]);
      } else {
        dr = cs_normal4(x);
      }
    }
    f32x4 b0 = xo, b1 = xo, b2 = xo, xn4 = xo;  // blended generator parameters, new imputation
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int d = d0 + r;
      if (d < D) {
        float xn, lpx;
        if (bern) {
          const float lg = g0o[r] * fn + g0n[r] * fa;
          const float xm = (dr[r] < cs_sigmoid(lg)) ? 1.0f : 0.0f;
          xn = xo[r] * m[r] + xm * (1.0f - m[r]);
          lpx = cs_bernoulli_logpmf(xn, lg);
          b0[r] = lg;
        } else if (other) {
          const float mean = g0o[r] * fn + g0n[r] * fa;
          const float sc = g1o[r] * fn + g1n[r] * fa;
          const float df = g2o[r] * fn + g2n[r] * fa;
          float c = dc[r];
          if (stt && !a.b.chi2_x) c = cs_chi2_sample(a.rng, CS_STREAM_X, grow, (uint32_t)d, df);
          const float xm = cs_gen_sample(a.cfg.gen_dist, mean, sc, df, dr[r], c);
          xn = xo[r] * m[r] + xm * (1.0f - m[r]);
          lpx = cs_gen_logpdf(a.cfg.gen_dist, xn, mean, sc, df);
          b0[r] = mean;
          b1[r] = sc;
          b2[r] = df;
        } else {
          const float mean = g0o[r] * fn + g0n[r] * fa;
          const float sc = g1o[r] * fn + g1n[r] * fa;
          const float xm = mean + sc * dr[r];
          xn = xo[r] * m[r] + xm * (1.0f - m[r]);
          lpx = cs_normal_logpdf(xn, mean, sc);
          b0[r] = mean;
          b1[r] = sc;
        }
        xn4[r] = xn;
        ll += joint_old ? lpx : lpx * m[r];
      }
    }
    if (live) {
      if (full) {
        *(f32x4_u*)(go + d0) = b0;
        if (!bern) *(f32x4_u*)(go + D + d0) = b1;
        if (stt) *(f32x4_u*)(go + 2 * D + d0) = b2;
        *(f32x4_u*)(Xr + d0) = xn4;
        if (Xc) *(f32x4_u*)(Xc + d0) = xn4;
        if (ring_row) *(f32x4_u*)(ring_row + d0) = xn4;
        if (ring_row8) {
#pragma unroll
          for (int r = 0; r < 4; ++r) ring_row8[d0 + r] = xn4[r] != 0.0f;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (d0 + r < D) {
            go[d0 + r] = b0[r];
            if (!bern) go[D + d0 + r] = b1[r];
            if (stt) go[2 * D + d0 + r] = b2[r];
            Xr[d0 + r] = xn4[r];
            if (Xc) Xc[d0 + r] = xn4[r];
            if (ring_row) ring_row[d0 + r] = xn4[r];
            if (ring_row8) ring_row8[d0 + r] = xn4[r] != 0.0f;
          }
      }
    }
  }
  ll = csum(ll);
  if (live && l == 0) a.b.x_old_lp[t] = ll;
}

// ---- small helpers ----
__global__ void cs_rowsum_stdnormal_kernel(const float* Z, int64_t rows, int Zd, float* out) {
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (t >= rows) return;
  float s = 0.f;
  for (int d = lane; d < Zd; d += 64) s += cs_stdnormal_logpdf(Z[t * Zd + d]);
  s = wsum(s);
  if (lane == 0) out[t] = s;
}

// sum_d w_d log p(X[t,d] | gen[t]) per chain t = (b,k); w = M[b,:] or 1.  One wavefront per chain.
__global__ void __launch_bounds__(256) cs_gen_loglik_kernel(const float* gen, int gen_dist, const float* X,
                                                            const uint8_t* M, int64_t chains, int K, int D, float* out) {
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (t >= chains) return;
  const int64_t b = t / K;
  const bool bern = gen_dist == CS_DIST_BERN;
  const float* g = gen + t * cs_gen_nparams(gen_dist) * D;
  float ll = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float x = X[t * D + d];
    float lp;
    if (bern) lp = cs_bernoulli_logpmf(x, g[d]);
    else if (gen_dist > CS_DIST_BERN) lp = cs_gen_logpdf(gen_dist, x, g[d], g[D + d], cs_dist_is_studentt(gen_dist) ? g[2 * D + d] : 0.0f);
    else lp = cs_normal_logpdf(x, g[d], g[D + d]);
    ll += M ? lp * (float)M[b * D + d] : lp;
  }
  ll = wsum(ll);
  if (lane == 0) out[t] = ll;
}

template <typename T>
__global__ void cs_history_enqueue_kernel(T* ring_slot, const float* X, int64_t n, int64_t* avail, int64_t chains,
                                          int was_full) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) ring_slot[t] = sizeof(T) == 1 ? (T)(X[t] != 0.0f) : (T)X[t];
  if (was_full && t < chains) {
    const int64_t av = avail[t];
    avail[t] = av > 0 ? av - 1 : 0;
  }
}

__global__ void cs_history_available_kernel(int64_t* avail, const uint8_t* mask, int64_t chains, int64_t len) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < chains && (mask == nullptr || mask[t])) avail[t] = len;
}

// ----------------------------------------------------------------------------------------
static int check_acmwg(const cs_acmwg_cfg* c, const cs_acmwg_bufs* b) {
  CS_REQUIRE(c && b, "null argument");
  CS_REQUIRE(c->B >= 0 && c->K >= 1 && c->D >= 1 && c->Zd >= 1, "bad sizes");
  CS_REQUIRE(c->latent_dist == CS_DIST_NORMAL || c->latent_dist == CS_DIST_NORMAL_WITH_EPS,
             "latent distribution must be normal / normal_with_eps");
  CS_REQUIRE((c->gen_dist >= CS_DIST_NORMAL && c->gen_dist <= CS_DIST_BERN) || cs_dist_is_studentt(c->gen_dist) ||
                 c->gen_dist == CS_DIST_LOGITNORMAL_WITH_EPS,
             "bad generator distribution");
  CS_REQUIRE(c->prior_mixture_probability >= 0.0 && c->prior_mixture_probability < 1.0,
             "prior_mixture_probability must be in [0,1)");
  return CS_OK;
}

static void fill_mlp_common(CsMlpArgs& a, const cs_acmwg_cfg* c) {
  a.K = c->K; a.Kp = c->K; a.Zd = c->Zd; a.B = c->B; a.I = 1; a.D = c->D;
}

// Z = loc + scale*eps from the encoder on X, generator parameters, masked log-likelihood and
// prior log-density: the chain state after the warm-up (:1639-1660).
extern "C" int cs_acmwg_init_state(const cs_mlp_t* enc, const cs_mlp_t* dec, const cs_acmwg_cfg* c,
                                   const cs_acmwg_bufs* b, const cs_rng* rng, void* stream) {
  if (int rc = check_acmwg(c, b)) return rc;
  CS_REQUIRE(enc && dec && b->X && b->M && b->q && b->Z_old && b->gen_old && b->z_old_prior_lp && b->x_old_lp,
             "null argument");
  if (c->B == 0) return CS_OK;
  CsMlpArgs e;
  std::memset(&e, 0, sizeof(e));
  e.net = enc->dev; e.mode = CS_MLP_ENC; e.rows = (int64_t)c->B * c->K;
  e.in = b->X; e.in_stride = c->D; e.q = b->q;
  fill_mlp_common(e, c);
  e.latent_dist = c->latent_dist;
  if (int rc = cs_mlp_launch(e, (hipStream_t)stream)) return rc;
  CsMlpArgs d;
  std::memset(&d, 0, sizeof(d));
  d.net = dec->dev; d.mode = CS_MLP_DEC; d.rows = e.rows;
  fill_mlp_common(d, c);
  d.q = b->q; d.gen_dist = c->gen_dist; d.eps_z = b->eps_z; d.rng = cs_make_philox(rng);
  d.Z = b->Z_old; d.gen = b->gen_old; d.gen_ll = b->x_old_lp; d.X = b->X;
  d.M = (c->flags & CS_MWG_JOINT_TARGET_OLD) ? nullptr : b->M;
  if (int rc = cs_mlp_launch(d, (hipStream_t)stream)) return rc;
  const int64_t rows = e.rows;
  hipLaunchKernelGGL(cs_rowsum_stdnormal_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     b->Z_old, rows, c->Zd, b->z_old_prior_lp);
  CS_HIP(hipGetLastError());
  return CS_OK;
}

static int acmwg_step(const cs_mlp_t* enc, const cs_mlp_t* dec, const cs_acmwg_cfg* c, const cs_acmwg_bufs* b,
                      const cs_rng* rng, float* x_copy, void* stream);
extern "C" int cs_acmwg_step(const cs_mlp_t* enc, const cs_mlp_t* dec, const cs_acmwg_cfg* c, const cs_acmwg_bufs* b,
                             const cs_rng* rng, void* stream) {
  return acmwg_step(enc, dec, c, b, rng, nullptr, stream);
}
static int acmwg_step(const cs_mlp_t* enc, const cs_mlp_t* dec, const cs_acmwg_cfg* c, const cs_acmwg_bufs* b,
                      const cs_rng* rng, float* x_copy, void* stream) {
  if (int rc = check_acmwg(c, b)) return rc;
  CS_REQUIRE(enc && dec && b->X && b->M && b->Z_old && b->gen_old && b->z_old_prior_lp && b->x_old_lp && b->q &&
                 b->Z_new && b->gen_new && b->ll_new && b->accepted && b->log_accept,
             "null argument");
  const bool use_hist = !(c->flags & CS_MWG_PROPOSE_FROM_CURRENT);
  const bool mix = c->prior_mixture_probability > 0.0;
  if (use_hist) {
    CS_REQUIRE(b->ring && b->avail && b->slots && b->in_off, "null history argument");
    CS_REQUIRE(c->hist_len >= 1 && c->hist_tail >= 0 && c->hist_tail < c->hist_len, "bad history state");
  }
  CS_REQUIRE(!(use_hist || mix) || (b->use_prior && b->z_alt), "null argument (use_prior / z_alt)");
  CS_REQUIRE(enc->dev.in_dim == c->D && enc->dev.out_dim == 2 * c->Zd && enc->dev.split == 2,
             "encoder shape does not match the configuration");
  CS_REQUIRE(dec->dev.in_dim == c->Zd && dec->dev.out_dim == cs_gen_nparams(c->gen_dist) * c->D, "decoder shape mismatch");
  if (c->B == 0) return CS_OK;
  hipStream_t st = (hipStream_t)stream;
  CsAcmwgArgs a;
  a.cfg = *c; a.b = *b; a.rng = cs_make_philox(rng);
  a.x_copy = x_copy;
  const int64_t chains = (int64_t)c->B * c->K;
  if (use_hist || mix) {
    const int ZQ = (c->Zd + 3) / 4;
    hipLaunchKernelGGL(cs_acmwg_prepare_kernel, dim3((unsigned)((chains * ZQ + 255) / 256)), dim3(256), 0, st, a, ZQ);
    CS_HIP(hipGetLastError());
  }
  CsMlpArgs e;
  std::memset(&e, 0, sizeof(e));
  e.net = enc->dev; e.mode = CS_MLP_ENC; e.rows = chains;
  if (use_hist) {
    e.in = b->ring; e.in_off = b->in_off;
    e.in_u8 = c->ring_u8 ? 1 : 0;
  } else {
    e.in = b->X; e.in_stride = c->D;
  }
  e.q = b->q;
  fill_mlp_common(e, c);
  e.latent_dist = c->latent_dist;
  if (int rc = cs_mlp_launch(e, st)) return rc;
  if (c->proposal_temperature > 0.0f || c->proposal_anneal != CS_ANNEAL_NONE)  // :1672-1680, :1222-1230
    if (int rc = cs_proposal_transform(b->q, b->M, c->B, c->K, c->K, c->Zd, c->D, c->latent_dist, CS_PRIOR_STD_NORMAL,
                                       c->proposal_temperature, c->proposal_anneal, stream))
      return rc;
  CsMlpArgs d;
  std::memset(&d, 0, sizeof(d));
  d.net = dec->dev; d.mode = CS_MLP_DEC; d.rows = chains;
  fill_mlp_common(d, c);
  d.q = b->q; d.gen_dist = c->gen_dist; d.eps_z = b->eps_z; d.rng = a.rng;
  if (use_hist || mix) {
    d.z_override = b->use_prior; d.z_alt = b->z_alt;
  }
  d.Z = b->Z_new; d.gen = b->gen_new; d.gen_ll = b->ll_new; d.X = b->X;
  d.M = (c->flags & CS_MWG_JOINT_TARGET_NEW) ? nullptr : b->M;
  if (int rc = cs_mlp_launch(d, st)) return rc;
  {
    // lanes per chain: the power of two that covers the quads of the longer of a latent and a data row
    const int nq = ((c->D > c->Zd ? c->D : c->Zd) + 3) / 4;
#define CS_ACCEPT(LP_)                                                                                              \
  hipLaunchKernelGGL(cs_acmwg_accept_kernel<LP_>, dim3((unsigned)((chains + 4 * (64 / LP_) - 1) / (4 * (64 / LP_)))), \
                     dim3(256), 0, st, a)
    if (nq <= 4) CS_ACCEPT(4);
    else if (nq <= 8) CS_ACCEPT(8);
    else if (nq <= 16) CS_ACCEPT(16);
    else if (nq <= 32) CS_ACCEPT(32);
    else CS_ACCEPT(64);
#undef CS_ACCEPT
  }
  CS_HIP(hipGetLastError());
  return CS_OK;
}

// T steps in one call: the loop of the MWG samplers for callers that need nothing on the host between steps.  The
// history scalars advance as the reference's queue does (utils.py:41-44).
extern "C" int cs_acmwg_run(const cs_mlp_t* enc, const cs_mlp_t* dec, const cs_acmwg_cfg* c0, const cs_acmwg_bufs* b0,
                            const cs_rng* rng, int32_t T, const cs_acmwg_run_out* out, const cs_run_observe* obs,
                            void* stream) {
  CS_REQUIRE(c0 && b0 && T >= 0, "bad argument");
  if (int rc = check_acmwg(c0, b0)) return rc;
  CsObserver observer;
  if (int rc = observer.begin(enc, obs, c0->B, c0->K, c0->D, c0->Zd, c0->latent_dist)) return rc;
  const int64_t chains = (int64_t)c0->B * c0->K;
  const int64_t n_z = chains * c0->Zd, n_x = chains * c0->D;
  const bool use_hist = !(c0->flags & CS_MWG_PROPOSE_FROM_CURRENT);
  cs_acmwg_cfg c = *c0;
  cs_rng r = rng ? *rng : cs_rng{0, 0, 0};
  for (int32_t t = 0; t < T; ++t) {
    cs_acmwg_bufs b = *b0;
    if (b.r_hist) b.r_hist += t * chains;
    if (b.eps_z) b.eps_z += t * n_z;
    if (b.z_prior) b.z_prior += t * n_z;
    if (b.u_prior) b.u_prior += t * chains;
    if (b.u_accept) b.u_accept += t * chains;
    if (b.eps_x) b.eps_x += t * n_x;
    if (b.chi2_x) b.chi2_x += t * n_x;
    if (out) {
      if (out->accepted_store) b.accepted = out->accepted_store + t * chains;
      if (out->log_accept_store) b.log_accept = out->log_accept_store + t * chains;
      if (out->slots_store) b.slots = out->slots_store + t * chains;
    }
    cs_rng rt = r;
    rt.iteration = r.iteration + (uint32_t)t;
    // (the accept kernel writes the kept copy itself: no copy launch in the chain)
    float* kept = obs && obs->X_store ? obs->X_store + t * n_x : nullptr;
    int rc = acmwg_step(enc, dec, &c, &b, &rt, kept, stream);
    if (use_hist) {
      c.hist_tail = (c.hist_tail + 1) % c.hist_len;
      if (c.hist_tail == c.hist_len - 1) c.hist_full = 1;
    }
    if (rc == CS_OK && out && out->Z_store && chains &&
        hipMemcpyAsync(out->Z_store + t * n_z, b.Z_old, (size_t)n_z * sizeof(float), hipMemcpyDeviceToDevice,
                       (hipStream_t)stream) != hipSuccess)
      rc = cs_fail(CS_EHIP, "copy of the chain latents into Z_store failed");
    if (rc == CS_OK) rc = observer.iteration(t, b.X, b.M, (hipStream_t)stream, kept != nullptr);
    if (rc != CS_OK) {
      (void)observer.finish((hipStream_t)stream);
      return rc;
    }
  }
  return observer.finish((hipStream_t)stream);
}

// LAIR's historical proposals: Xc[b] = (X[b, 0..K), ring[slots[h], b, chain] for h < H).
struct CsHistConcatArgs {
  const float* X; const float* ring; const uint8_t* ring8; float* Xc; const int64_t* chain_of_example; CsPhilox rng;
  int64_t B; int K, D, H; int use_chain_of_proposal;
  int k_lo, k_cnt, h0;   // this launch writes chains [k_lo, k_lo + k_cnt) of Xc; slots / chain_of_proposal hold proposals h0..
  int slots[64]; int chain_of_proposal[64];
};

__global__ void __launch_bounds__(256) cs_history_concat_kernel(const CsHistConcatArgs a) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int KH = a.K + a.H;
  if (t >= a.B * a.k_cnt * a.D) return;
  const int d = (int)(t % a.D);
  const int64_t bk = t / a.D;
  const int k = a.k_lo + (int)(bk % a.k_cnt);
  const int64_t b = bk / a.k_cnt;
  float v;
  if (k < a.K) {
    v = a.X[(b * a.K + k) * a.D + d];
  } else {
    const int h = k - a.K - a.h0;
    int chain;
    if (a.use_chain_of_proposal) {
      chain = a.chain_of_proposal[h];
    } else if (a.chain_of_example) {
      chain = (int)a.chain_of_example[b];
    } else {
      uint32_t x[4];
      cs_philox_block(a.rng, CS_STREAM_HIST, a.rng.row_offset + b, 0, x);
      chain = (int)(((uint64_t)x[0] * (uint64_t)a.K) >> 32);
    }
    const int64_t e = (((int64_t)a.slots[h] * a.B + b) * a.K + chain) * a.D + d;
    v = a.ring8 ? (float)a.ring8[e] : a.ring[e];
  }
  a.Xc[(b * KH + k) * a.D + d] = v;
}

static int history_concat(const float* X, const float* ring, const uint8_t* ring8, int64_t B, int32_t K, int32_t D, int32_t H,
                          const int32_t* slots, const int32_t* chain_of_proposal, const int64_t* chain_of_example,
                          const cs_rng* rng, float* Xc, void* stream) {
  CS_REQUIRE(X && (ring || ring8) && Xc && slots && B >= 0 && K >= 1 && D >= 1, "bad argument");
  CS_REQUIRE(H >= 1, "number of historical proposals must be positive");
  for (int h = 0; h < H; ++h)
    CS_REQUIRE(slots[h] >= 0 && (!chain_of_proposal || (chain_of_proposal[h] >= 0 && chain_of_proposal[h] < K)),
               "bad slot / chain index");
  if (B == 0) return CS_OK;
  CsHistConcatArgs a;
  a.X = X; a.ring = ring; a.ring8 = ring8; a.Xc = Xc; a.chain_of_example = chain_of_example; a.rng = cs_make_philox(rng);
  a.B = B; a.K = K; a.D = D; a.H = H; a.use_chain_of_proposal = chain_of_proposal != nullptr;
  // the slot / chain tables travel in the kernel arguments, 64 historical proposals per launch; the first launch
  // also copies the K current chains
  for (int h0 = 0; h0 < H; h0 += 64) {
    const int hc = (H - h0) < 64 ? (H - h0) : 64;
    a.h0 = h0;
    a.k_lo = h0 == 0 ? 0 : K + h0;
    a.k_cnt = h0 == 0 ? K + hc : hc;
    for (int h = 0; h < 64; ++h) {
      a.slots[h] = h < hc ? slots[h0 + h] : 0;
      a.chain_of_proposal[h] = (h < hc && chain_of_proposal) ? chain_of_proposal[h0 + h] : 0;
    }
    const int64_t n = B * a.k_cnt * D;
    hipLaunchKernelGGL(cs_history_concat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    CS_HIP(hipGetLastError());
  }
  return CS_OK;
}

extern "C" int cs_history_concat(const float* X, const float* ring, int64_t B, int32_t K, int32_t D, int32_t H,
                                 const int32_t* slots, const int32_t* chain_of_proposal, const int64_t* chain_of_example,
                                 const cs_rng* rng, float* Xc, void* stream) {
  return history_concat(X, ring, nullptr, B, K, D, H, slots, chain_of_proposal, chain_of_example, rng, Xc, stream);
}

extern "C" int cs_history_concat_u8(const float* X, const uint8_t* ring, int64_t B, int32_t K, int32_t D, int32_t H,
                                    const int32_t* slots, const int32_t* chain_of_proposal, const int64_t* chain_of_example,
                                    const cs_rng* rng, float* Xc, void* stream) {
  return history_concat(X, nullptr, ring, B, K, D, H, slots, chain_of_proposal, chain_of_example, rng, Xc, stream);
}

template <typename T>
static int history_enqueue(T* ring, int64_t* avail, int32_t hist_len, int32_t new_tail, int32_t was_full, const float* X,
                           int64_t B, int32_t K, int32_t D, void* stream) {
  CS_REQUIRE(ring && (avail || !was_full) && X && hist_len >= 1 && new_tail >= 0 && new_tail < hist_len, "bad argument");
  const int64_t chains = B * K, n = chains * D;
  if (n == 0) return CS_OK;
  hipLaunchKernelGGL(cs_history_enqueue_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     ring + (int64_t)new_tail * n, X, n, avail, chains, was_full);
  CS_HIP(hipGetLastError());
  return CS_OK;
}

extern "C" int cs_history_enqueue(float* ring, int64_t* avail, int32_t hist_len, int32_t new_tail, int32_t was_full,
                                  const float* X, int64_t B, int32_t K, int32_t D, void* stream) {
  return history_enqueue<float>(ring, avail, hist_len, new_tail, was_full, X, B, K, D, stream);
}

extern "C" int cs_history_enqueue_u8(uint8_t* ring, int64_t* avail, int32_t hist_len, int32_t new_tail, int32_t was_full,
                                     const float* X, int64_t B, int32_t K, int32_t D, void* stream) {
  return history_enqueue<uint8_t>(ring, avail, hist_len, new_tail, was_full, X, B, K, D, stream);
}

extern "C" int cs_history_set_available(int64_t* avail, const uint8_t* mask, int64_t chains, int64_t len, void* stream) {
  CS_REQUIRE(avail && chains >= 0, "bad argument");
  if (chains == 0) return CS_OK;
  hipLaunchKernelGGL(cs_history_available_kernel, dim3((unsigned)((chains + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, avail, mask, chains, len);
  CS_HIP(hipGetLastError());
  return CS_OK;
}

extern "C" int cs_gen_loglik_rows(const float* gen, int32_t gen_dist, const float* X, const uint8_t* M, int64_t B,
                                  int32_t K, int32_t D, float* out, void* stream) {
  CS_REQUIRE(gen && X && out && B >= 0 && K >= 1 && D >= 1, "bad argument");
  CS_REQUIRE((gen_dist >= CS_DIST_NORMAL && gen_dist <= CS_DIST_BERN) || cs_dist_is_studentt(gen_dist) ||
                 gen_dist == CS_DIST_LOGITNORMAL_WITH_EPS,
             "bad generator distribution");
  const int64_t chains = B * K;
  if (chains == 0) return CS_OK;
  hipLaunchKernelGGL(cs_gen_loglik_kernel, dim3((unsigned)((chains + 3) / 4)), dim3(256), 0, (hipStream_t)stream, gen,
                     gen_dist, X, M, chains, K, D, out);
  CS_HIP(hipGetLastError());
  return CS_OK;
}

// sum_d log N(Z[row,d]; 0, 1): prior log-density of the chain state (:1655)
extern "C" int cs_stdnormal_logprob_rows(const float* Z, int64_t rows, int32_t Zd, float* out, void* stream) {
  CS_REQUIRE(Z && out && rows >= 0 && Zd >= 1, "bad argument");
  if (rows == 0) return CS_OK;
  hipLaunchKernelGGL(cs_rowsum_stdnormal_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     Z, rows, Zd, out);
  CS_HIP(hipGetLastError());
  return CS_OK;
}
