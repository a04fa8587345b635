// LAIR iteration stages (importance_resampling_gibbs_iteration,
// irwg/sampling/vae_imputation.py:484-726) on top of the fused MLP kernel:
//   cs_lair_encode           encoder + proposal parameterisation           (:551, vae.py:214-219)
//   cs_lair_decode           Z sampling + decoder + masked log-likelihood  (:596-603, weights.py:46-47)
//   cs_lair_resample_impute  DM-MIS weights, resampling, x_m sampling, in-place imputation,
//                            ESS / perplexity   (weights.py:50-85; vae_imputation.py:284-308,
//                            335-378, 380-458, 677-726, 804-834)
#include "cs_lair.h"

#include <cstdlib>
#include <cstring>
#include <mutex>

// ----------------------------------------------------------------------------------------
// One wavefront per example.  Lane s < S = I*K' owns proposal s = i*K' + k'.
// Proposals are partitioned in G groups of c members per importance sample (gr: G = 1,
// c = K'; glr: member k' belongs to group k' mod G, '(c k)' layout of the reference); the
// deterministic-mixture proposal density and the resampling are both "within group" when the
// corresponding option asks for it.
// ----------------------------------------------------------------------------------------

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// torch.logsumexp over the lanes with `in` set (others ignored): max, masked to 0 when not
// finite, log(sum(exp(x - max))) + max.
__device__ __forceinline__ float wave_logsumexp(float x, bool in) {
  float m = wave_max(in ? x : -INFINITY);
  if (!(fabsf(m) < INFINITY)) m = 0.0f;
  const float s = wave_sum(in ? expf(x - m) : 0.0f);
  return logf(s) + m;
}

// E = floor(64/S) examples share a wavefront: lane = e*S + s.  Reductions over the proposals of
// one example are loops of wave shuffles over its S lanes (S is small: 5..25 in the
// configurations), so every lane does useful work.
// VEC: Zd is a multiple of 4 and q / Z are 16-byte aligned, so staging and the density loops move
// 4 latent dimensions per instruction.  The staged rows are padded (q_stride / 4 odd, z_stride =
// Zd + 4, or odd strides without VEC) so that the examples / proposals of a wavefront fall into
// different LDS banks.
template <bool VEC>
__global__ void __launch_bounds__(64) cs_lair_resample_kernel(const CsResampleArgs a) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  const int B = a.cfg.B, K = a.cfg.K, D = a.cfg.D, Zd = a.cfg.Zd, I = a.cfg.I;
  const int Kp = a.Kp, S = a.S, G = a.G, c = a.c, E = a.E, NS = a.NS;
  const int QS = a.q_stride, ZS = a.z_stride;
  const int64_t b0 = (int64_t)blockIdx.x * E;            // first example of this wave
  const int ne = (int)((B - b0) < E ? (B - b0) : E);     // examples present
  const bool beta_q = a.cfg.latent_dist == CS_DIST_BETA;
  float* l_loc = lds;                      // [E][QS]  (Kp*Zd used)
  float* l_2var = l_loc + E * QS;          // [E][QS]  1/(2*scale^2)
  float* l_z = l_2var + E * QS;            // [E][S][ZS]
  int* l_idx = (int*)(l_z + E * S * ZS);   // [E][2*NS] resampled global indices (two draws)
  float* l_c = (float*)(l_idx + E * 2 * NS);  // [E][S][mixc] component log-densities
  float* l_L = l_c + E * S * a.mixc;       // [E][Kp]  normal posteriors: a component's proposal-independent terms
  float* l_logs = l_L + E * Kp;            // [E][QS]  Beta posteriors only: log-normaliser per dimension
  // -DCS_RS_PROFILE (debug builds only, see tools/rs_phases.py): the wavefront's clock at the phase
  // boundaries; the phase durations then overwrite the first entries of ess / tess / perplexity
#ifdef CS_RS_PROFILE
  unsigned long long T[9]; int tn = 0;
#define CS_TICK() T[tn++] = __builtin_amdgcn_s_memtime()
#else
#define CS_TICK()
#endif
  CS_TICK();

  {
    const float* qb = a.b.q + b0 * Kp * 2 * Zd;
    // (loc, sc) -> staged triple.  Beta: (alpha, beta) -> alpha-1, beta-1 and the log-normaliser of
    // torch's Dirichlet.log_prob, lgamma(alpha+beta) - (lgamma(alpha) + lgamma(beta))
    auto stage1 = [&](float loc, float sc, float& o0, float& o1, float& o2) {
      if (beta_q) {
        o0 = loc - 1.0f;
        o1 = sc - 1.0f;
        o2 = lgammaf(loc + sc) - (lgammaf(loc) + lgammaf(sc));
      } else {
        o0 = loc;
        o1 = __builtin_amdgcn_rcpf(2.0f * (sc * sc));
        o2 = 0.0f;   // (log(scale) is summed per component below, not staged)
      }
    };
    if (VEC) {
      // one loop for both copies, so that the loads of q and Z share their round trips to memory
      const int Zq = Zd >> 2;
      const int nq = ne * Kp * Zq, nz = ne * S * Zq;
#pragma unroll 2
      for (int t = lane; t < nz; t += 64) {   // nz >= nq (S >= Kp)
        const int es = cs_div(t, a.dZq), dq = t - es * Zq;   // es = e*S + s; for q: ej = e*Kp + j
        const int e = cs_div(es, a.dS), s = es - e * S;
        const int i = cs_div(s, a.dKp), kp = s - i * Kp;
        const f32x4 zz = *(const f32x4*)(a.b.Z + (((int64_t)i * B + b0 + e) * Kp + kp) * Zd + 4 * dq);
        const bool hq = t < nq;
        const int eq = cs_div(es, a.dKp), j = es - eq * Kp;
        f32x4 loc = {1.0f, 1.0f, 1.0f, 1.0f}, sc = loc;
        if (hq) {
          loc = *(const f32x4*)(qb + (int64_t)es * 2 * Zd + 4 * dq);
          sc = *(const f32x4*)(qb + (int64_t)es * 2 * Zd + Zd + 4 * dq);
        }
        *(f32x4*)(l_z + es * ZS + 4 * dq) = zz;
        if (hq) {
          float o0[4], o1[4], o2[4];
          stage1(loc.x, sc.x, o0[0], o1[0], o2[0]);
          stage1(loc.y, sc.y, o0[1], o1[1], o2[1]);
          stage1(loc.z, sc.z, o0[2], o1[2], o2[2]);
          stage1(loc.w, sc.w, o0[3], o1[3], o2[3]);
          const int o = eq * QS + j * Zd + 4 * dq;
          *(f32x4*)(l_loc + o) = f32x4{o0[0], o0[1], o0[2], o0[3]};
          *(f32x4*)(l_2var + o) = f32x4{o1[0], o1[1], o1[2], o1[3]};
          if (beta_q) *(f32x4*)(l_logs + o) = f32x4{o2[0], o2[1], o2[2], o2[3]};
        }
      }
    } else {
      for (int t = lane; t < ne * Kp * Zd; t += 64) {
        const int ej = cs_div(t, a.dZd), d = t - ej * Zd;  // ej = e*Kp + j
        const int e = cs_div(ej, a.dKp), j = ej - e * Kp;
        const int o = e * QS + j * Zd + d;
        float o2;
        stage1(qb[(int64_t)ej * 2 * Zd + d], qb[(int64_t)ej * 2 * Zd + Zd + d], l_loc[o], l_2var[o], o2);
        if (beta_q) l_logs[o] = o2;
      }
      for (int t = lane; t < ne * S * Zd; t += 64) {
        const int es = cs_div(t, a.dZd), d = t - es * Zd;
        const int e = cs_div(es, a.dS), s = es - e * S;
        const int i = cs_div(s, a.dKp), kp = s - i * Kp;
        l_z[es * ZS + d] = a.b.Z[(((int64_t)i * B + b0 + e) * Kp + kp) * Zd + d];
      }
    }
  }
  // Normal posteriors: the terms of a component's log-density that do not depend on the proposal,
  // L[e][j] = sum_d log(scale_jd) + Zd * log(sqrt(2 pi)), summed once per component (in order of d)
  // instead of once per proposal.  Lane e*Kp + j takes its component's scales from q again (the staging loop
  // has just pulled those lines in): staging the logarithms would cost a third array of the posteriors' size in
  // LDS, which is what limits the number of resident wavefronts.
  if (!beta_q && lane < ne * Kp) {
    const float* sr = a.b.q + (b0 * Kp + lane) * 2 * Zd + Zd;
    float Lj = 0.0f;
    if (VEC) {
      for (int d = 0; d < Zd; d += 4) {
        const f32x4 sc = *(const f32x4*)(sr + d);
        Lj += cs_logf(sc.x);
        Lj += cs_logf(sc.y);
        Lj += cs_logf(sc.z);
        Lj += cs_logf(sc.w);
      }
    } else {
      for (int d = 0; d < Zd; ++d) Lj += cs_logf(sr[d]);
    }
    l_L[lane] = Lj + (float)Zd * CS_LOG_SQRT_2PI;
  }
  __syncthreads();
  CS_TICK();

  // An example owns P lanes, P the power of two >= S, so that the reductions over its proposals are
  // log2(P) butterfly steps instead of S broadcasts.
  const int P = a.P;
  const int e_l = lane >> a.logP;           // example slot of this lane
  const int s = lane & (P - 1);             // proposal
  const bool act = e_l < ne && s < S;
  const int base = e_l * P;                 // first lane of this example
  const int64_t b = b0 + e_l;
  const int i_s = cs_div(s, a.dKp), kp_s = s - i_s * Kp;
  // Resampling group of this proposal and its position inside the group.  gr / glr / lr: group
  // kp mod G, members in '(i c)' order (lr: G = K', c = 1, so the group of chain k holds its I
  // importance samples).  ir: the flattened '(i k)' axis is cut into K' runs of I consecutive
  // entries (the reference's reshape(B, K, -1), vae_imputation.py:463-464).
  // With glr_pad_to the groups are laid out for a larger proposal count (G = pad_to / c) whose
  // extra members do not exist: groups then hold fewer than I*c members (:339-341, 395-405).
  const bool ir = a.cfg.scheme == CS_RS_IR;
  const int g_s = ir ? s / I : kp_s % G;
  // bit jj: proposal jj of this example is in this lane's group; member order = lane order, and
  // the lane index of a member is its flattened '(i k)' proposal index
  unsigned long long gmask = 0;
  for (int jj = 0; jj < S; ++jj) {
    const int gj = ir ? jj / I : (jj % Kp) % G;
    if (gj == g_s) gmask |= 1ull << jj;
  }
  const int gsize = __popcll(gmask);
  const int li_s = __popcll(gmask & ((1ull << s) - 1ull));  // position inside the group
  auto nth_member = [&](int n) {                            // flattened index of the group's n-th member
    unsigned long long m = gmask;
    for (int q = 0; q < n; ++q) m &= m - 1ull;
    return (int)__ffsll((long long)m) - 1;
  };

  // reductions over the lanes of this example (optionally only its resampling group); the group
  // form walks the members, the whole-example form (also a single group) is a butterfly
  const bool one_group = G == 1 && !ir;
  auto seg_max = [&](float x, bool grouped) {
    if (!grouped || one_group) {
      float m = s < S ? x : -INFINITY;
      for (int o = P >> 1; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      return m;
    }
    float m = -INFINITY;
    for (int jj = 0; jj < S; ++jj) {
      const float v = __shfl(x, base + jj);
      if ((gmask >> jj) & 1ull) m = fmaxf(m, v);
    }
    return m;
  };
  auto seg_sum = [&](float x, bool grouped) {
    if (!grouped || one_group) {
      float t = s < S ? x : 0.0f;
      for (int o = P >> 1; o > 0; o >>= 1) t += __shfl_xor(t, o);
      return t;
    }
    float t = 0.0f;
    for (int jj = 0; jj < S; ++jj) {
      const float v = __shfl(x, base + jj);
      if ((gmask >> jj) & 1ull) t += v;
    }
    return t;
  };
  // torch.logsumexp: max (0 when not finite), log(sum(exp(x - max))) + max
  auto seg_lse = [&](float x, bool grouped) {
    float m = seg_max(x, grouped);
    if (!(fabsf(m) < INFINITY)) m = 0.0f;
    return logf(seg_sum(expf(x - m), grouped)) + m;
  };

  // requested here, consumed after the densities
  const int64_t rho_s = ((int64_t)i_s * B + b) * Kp + kp_s;
  const float gll = act ? a.b.gen_ll[rho_s] : 0.0f;

  // ---- component log-densities: one work item per (example, proposal, component slot), spread
  // over all 64 lanes (the proposals of the wave's examples alone would leave most lanes idle
  // during the most expensive loop).  Components of proposal kp: j = kp mod mixG + mixG*cc, cc < mixc
  // (smis: mixG = K', mixc = 1, so j = kp); j >= K' is a member that the padded layout lacks.
  {
    const int MC = a.mixc;
    for (int t = lane; t < ne * S * MC; t += 64) {
      const int es = cs_div(t, a.dMC), cc = t - es * MC;
      const int e = cs_div(es, a.dS), s2 = es - e * S;
      const int kp2 = s2 - cs_div(s2, a.dKp) * Kp;
      const int j = kp2 - cs_div(kp2, a.dmixG) * a.mixG + a.mixG * cc;
      float cj = -INFINITY;
      if (j < Kp) {
        // log-density of component j at the proposal, summed over the latent dimensions in order
        const float* z = l_z + es * ZS;
        const float* qloc = l_loc + e * QS + j * Zd;
        const float* q2v = l_2var + e * QS + j * Zd;
        const float* qls = l_logs + e * QS + j * Zd;
        cj = 0.0f;
        if (beta_q) {
          for (int d = 0; d < Zd; ++d)
            cj += (cs_xlogy(qloc[d], z[d]) + cs_xlogy(q2v[d], 1.0f - z[d])) + qls[d];
        } else if (VEC) {
          for (int d = 0; d < Zd; d += 4) {
            const f32x4 zz = *(const f32x4*)(z + d);
            const f32x4 lo = *(const f32x4*)(qloc + d);
            const f32x4 iv = *(const f32x4*)(q2v + d);
            float df;
            df = zz.x - lo.x; cj -= (df * df) * iv.x;
            df = zz.y - lo.y; cj -= (df * df) * iv.y;
            df = zz.z - lo.z; cj -= (df * df) * iv.z;
            df = zz.w - lo.w; cj -= (df * df) * iv.w;
          }
          cj -= l_L[e * Kp + j];
        } else {
          for (int d = 0; d < Zd; ++d) {
            const float df = z[d] - qloc[d];
            cj -= (df * df) * q2v[d];
          }
          cj -= l_L[e * Kp + j];
        }
      }
      l_c[t] = cj;
    }
  }
  __syncthreads();

  float lw = 0.0f;
  if (act) {
    const float* z = l_z + (e_l * S + s) * ZS;
    float pl = 0.0f;
    if (a.cfg.prior == CS_PRIOR_UNIFORM) {
      // Uniform(0, 1 + 1e-8).log_prob (the upper bound is 1 in float32): log(1[0 <= z < 1]) - log(1)
      for (int d = 0; d < Zd; ++d) pl += (z[d] >= 0.0f && z[d] < 1.0f) ? 0.0f : -INFINITY;
    } else if (VEC) {
      for (int d = 0; d < Zd; d += 4) {
        const f32x4 zz = *(const f32x4*)(z + d);
        pl += cs_stdnormal_logpdf(zz.x);
        pl += cs_stdnormal_logpdf(zz.y);
        pl += cs_stdnormal_logpdf(zz.z);
        pl += cs_stdnormal_logpdf(zz.w);
      }
    } else {
      for (int d = 0; d < Zd; ++d) pl += cs_stdnormal_logpdf(z[d]);
    }
    // deterministic-mixture density of the proposal: logsumexp over its components - log(count)
    const float* cv = l_c + (e_l * S + s) * a.mixc;
    float ql;
    if (a.cfg.weighting == CS_W_SMIS) {
      ql = cv[0];
    } else {
      const int gm = kp_s - cs_div(kp_s, a.dmixG) * a.mixG;
      float m = -INFINITY, sum = 0.0f;
      int ncomp = 0;
      for (int cc = 0; cc < a.mixc && gm + a.mixG * cc < Kp; ++cc) {
        m = fmaxf(m, cv[cc]);
        ++ncomp;
      }
      if (!(fabsf(m) < INFINITY)) m = 0.0f;
      for (int cc = 0; cc < ncomp; ++cc) sum += expf(cv[cc] - m);
      ql = logf(sum) + m - logf((float)ncomp);
    }
    lw = gll + pl - ql;
    a.b.log_w[rho_s] = lw;
  }
  const float tlw = (a.cfg.weight_alpha == 1.0f) ? lw : lw * a.cfg.weight_alpha;
  CS_TICK();

  // ---- ESS / perplexity over all S proposals (vae_imputation.py:708-719) ----
  {
    const float lnw = lw - seg_lse(lw, false);
    const float ess = expf(-seg_lse(2.0f * lnw, false));
    const float perp = expf(-seg_sum(lnw * expf(lnw), false));
    float tess = ess;
    if (a.cfg.weight_alpha != 1.0f) {
      const float tl = tlw - seg_lse(tlw, false);
      tess = expf(-seg_lse(2.0f * tl, false));
    }
    if (act && s == 0) {
      a.b.ess[b] = ess;
      a.b.tess[b] = tess;
      a.b.perplexity[b] = perp;
    }
  }

  CS_TICK();
  // ---- resampling within groups ----
  const float nl = tlw - seg_lse(tlw, true);  // normalised log-weight (Categorical.logits)
  // ir returns the (transformed) log-weights themselves, in '(i k)' order (vae_imputation.py:479-480)
  // (padded glr returns them in '(i k)' order, :371-374)
  const bool flat_nlw = ir || a.cfg.glr_pad_to > 0;
  if (act) a.b.norm_log_w[b * S + (flat_nlw ? s : g_s * (I * c) + li_s)] = ir ? tlw : nl;
  float p;
  if (a.cfg.method == CS_RM_MULTINOMIAL) {
    // Categorical.probs = softmax(normalised logits)
    const float mx = seg_max(nl, true);
    const float ex = expf(nl - mx);
    p = ex / seg_sum(ex, true);
  } else {
    // systematic: exp(normalised log-weights), renormalised by their sum (:813-817)
    const float ex = expf(nl);
    p = ex / seg_sum(ex, true);
  }
  // sequential float32 running sum inside each group, in member order
  float run = 0.0f, cum = 0.0f;
  for (int jj = 0; jj < S; ++jj) {
    const float pj = __shfl(p, base + jj);
    if ((gmask >> jj) & 1ull) run += pj;
    if (s == jj) cum = run;
  }
  const bool last_in_group = li_s == gsize - 1;
  // one draw per group (lr, ir): torch.multinomial(p, 1) = argmax(p / Exp(1)), first maximum
  const bool race = a.cfg.method == CS_RM_MULTINOMIAL && (a.cfg.scheme == CS_RS_LR || ir);
  const unsigned long long ex_mask = (S == 64 ? ~0ull : ((1ull << S) - 1ull)) << base;

  const int ndraw_sets = (a.b.latents_resampled != nullptr) ? 2 : 1;
  CS_TICK();
  int* my_idx = l_idx + e_l * 2 * NS;
  for (int set = 0; set < ndraw_sets; ++set) {
    if (race) {
      const float* ein = set == 0 ? a.b.e_res : a.b.e_res2;
      float v = -INFINITY;
      if (act) {
        const int el = g_s * I + li_s;  // element of the [K', I] block of exponentials of example b
        float ex;
        if (ein) {
          ex = ein[b * S + el];
        } else {
          uint32_t x[4];
          cs_philox_block(a.rng, set == 0 ? CS_STREAM_URES : CS_STREAM_URES2, a.rng.row_offset + b, (uint32_t)(el >> 2), x);
          ex = -logf(1.0f - cs_u01f(x[el & 3]));
        }
        v = p / ex;
      }
      float best = -INFINITY;
      int best_li = 0, cnt = 0;
      for (int jj = 0; jj < S; ++jj) {
        const float vj = __shfl(v, base + jj);
        if ((gmask >> jj) & 1ull) {
          if (vj > best) { best = vj; best_li = cnt; }
          ++cnt;
        }
      }
      if (act && li_s == 0) my_idx[set * NS + g_s] = nth_member(best_li);
    } else if (a.cfg.method == CS_RM_MULTINOMIAL) {
      // torch.multinomial: cum / total, last forced to 1, left bisection with float64 uniforms
      float cn = cum / run;
      if (last_in_group) cn = 1.0f;
      const double* uin = set == 0 ? a.b.u_res : a.b.u_res2;
      // Draw d = g*c + t of the example is resolved by lane d of the example (S draws at a time):
      // the lane fetches / generates its uniform, then the members' CDF values are broadcast one
      // by one and every lane counts those of ITS draw's group that lie below its uniform.
      const int ND = G * c;
      for (int d0 = 0; d0 < ND; d0 += S) {
        const int d = d0 + s;
        const bool has = act && d < ND;
        double u_mine = 0.0;
        if (has) {
          if (uin) u_mine = uin[b * ND + d];
          else u_mine = cs_uniform64_elem(a.rng, set == 0 ? CS_STREAM_URES : CS_STREAM_URES2, a.rng.row_offset + b, (uint32_t)d);
        }
        const int gd = cs_div(d, a.dc), td = d - gd * c;
        int li = 0, cnt = 0;
        unsigned long long gm = 0;   // members of the draw's group
        for (int jj = 0; jj < S; ++jj) {
          const float cj = __shfl(cn, base + jj);
          const int gj = one_group ? 0 : __shfl(g_s, base + jj);
          if (gj == gd) {
            gm |= 1ull << jj;
            ++cnt;
            li += ((double)cj < u_mine) ? 1 : 0;
          }
        }
        if (has) {
          if (li > cnt - 1) li = cnt - 1;
          for (int q = 0; q < li; ++q) gm &= gm - 1ull;   // li-th member of the group
          // output slot t*G + g ('c b k d -> b (c k) d')
          my_idx[set * NS + td * G + gd] = (int)__ffsll((long long)gm) - 1;
        }
      }
    } else {
      const float* uin = set == 0 ? a.b.u_sys : a.b.u_sys2;
      // n = clamp(floor(cumsum(p)*m + U), 0, m); index[t] = #{members : n <= t}
      float U = 0.0f;
      if (act) {
        if (uin) U = uin[b * G + g_s];
        else U = cs_uniform_elem(a.rng, set == 0 ? CS_STREAM_URES : CS_STREAM_URES2, a.rng.row_offset + b, (uint32_t)g_s);
      }
      float nf = floorf(cum * (float)c + U);
      nf = fminf(fmaxf(nf, 0.0f), (float)c);
      const int n = (int)nf;
      for (int g = 0; g < G; ++g) {
        for (int t = 0; t < c; ++t) {
          const unsigned long long mask = __ballot(act && g_s == g && n <= t);
          int li = __popcll(mask & ex_mask);
          if (act && g_s == g && li_s == 0) {
            if (li > gsize - 1) li = gsize - 1;  // the reference would index out of bounds here
            my_idx[set * NS + t * G + g] = nth_member(li);
          }
        }
      }
    }
  }
  __syncthreads();
  CS_TICK();
  for (int t = lane; t < ne * NS; t += 64) {
    const int e = cs_div(t, a.dNS), n = t - e * NS;
    a.b.idx[(b0 + e) * NS + n] = l_idx[e * 2 * NS + n];
    if (ndraw_sets == 2 && a.b.idx2) a.b.idx2[(b0 + e) * NS + n] = l_idx[e * 2 * NS + NS + n];
  }

  // ---- resampled latents (second, independent draw; vae_imputation.py:649-650) ----
  if (ndraw_sets == 2) {
    for (int t = lane; t < ne * NS * Zd; t += 64) {
      const int en = cs_div(t, a.dZd), d = t - en * Zd;
      const int e = cs_div(en, a.dNS), n = en - e * NS;
      a.b.latents_resampled[b0 * NS * Zd + t] = l_z[(e * S + l_idx[e * 2 * NS + NS + n]) * ZS + d];
    }
  }

  // ---- sample the missing values from the resampled generator parameters and impute ----
  // work item = 4 consecutive dimensions of one chain (cs_lair_impute_item)
  CS_TICK();
  const int DQ = (D + 3) >> 2;
  const int Kimp = a.Kimp;
  for (int t = lane; t < ne * Kimp * DQ; t += 64) {
    const int en = cs_div(t, a.dDQ), dq = t - en * DQ;
    const int e = cs_div(en, a.dKimp), n = en - e * Kimp;
    cs_lair_impute_item(a, b0 + e, n, dq, l_idx[e * 2 * NS + n]);
  }
#ifdef CS_RS_PROFILE
  CS_TICK();
  __syncthreads();
  // E >= 3 examples per wave: 7 phase durations, 3 per array; E == 2: the first 6, 2 per array
  if (lane == 0 && ne >= 2 && ne == E) {
    const int per = E >= 3 ? 3 : 2;
    for (int i = 0; i + 1 < tn && i < 3 * per; ++i) {
      float* dst = i < per ? a.b.ess : (i < 2 * per ? a.b.tess : a.b.perplexity);
      dst[b0 + i % per] = (float)(T[i + 1] - T[i]);
    }
  }
#endif
}

// Test hook: 1 sends every proposal-set size through the path for more than 64 proposals per example.
static int g_force_large = 0;
extern "C" int cs_debug_force_large_resample(int on) {
  const int prev = g_force_large;
  g_force_large = on ? 1 : 0;
  return prev;
}

// ----------------------------------------------------------------------------------------
static int check_cfg(const cs_lair_cfg* c, const cs_lair_bufs* b) {
  CS_REQUIRE(c && b, "null argument");
  CS_REQUIRE(c->B >= 0 && c->K >= 1 && c->R >= 0 && c->I >= 1 && c->D >= 1 && c->Zd >= 1, "bad sizes");
  CS_REQUIRE(c->latent_dist == CS_DIST_NORMAL || c->latent_dist == CS_DIST_NORMAL_WITH_EPS ||
                 c->latent_dist == CS_DIST_BETA,
             "latent distribution must be normal / normal_with_eps / beta");
  CS_REQUIRE((c->gen_dist >= CS_DIST_NORMAL && c->gen_dist <= CS_DIST_BERN) || cs_dist_is_studentt(c->gen_dist) ||
                 c->gen_dist == CS_DIST_LOGITNORMAL_WITH_EPS,
             "bad generator distribution");
  CS_REQUIRE(c->prior == CS_PRIOR_STD_NORMAL || c->prior == CS_PRIOR_UNIFORM, "bad prior");
  return CS_OK;
}

// ----------------------------------------------------------------------------------------
// Proposal tempering / annealing to the prior (vae.py:258-401), in place on q.  One wavefront per
// example: the missing fraction is a wave reduction over the mask row, then the K*Zd parameter
// pairs of the example are transformed with the reference's float32 operation order, including
// its round trip raw = softplus^-1(scale'), scale'' = softplus(raw) (+- 1e-8).
// ----------------------------------------------------------------------------------------
struct CsPropTransformArgs {
  float* q; const uint8_t* M; int64_t B; int K, Kp, Zd, D, latent_dist; float temperature; int anneal;
};

__device__ __forceinline__ float cs_softplus_inverse_ref(float x) { return x + logf(-expm1f(-x)); }  // softplus_inverse.py:3-7
__device__ __forceinline__ float cs_softplus_ref(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

__global__ void __launch_bounds__(256) cs_proposal_transform_kernel(const CsPropTransformArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= a.B) return;
  float mf = 0.0f;
  if (a.anneal != CS_ANNEAL_NONE) {
    int obs = 0;
    for (int d = lane; d < a.D; d += 64) obs += a.M[b * a.D + d] ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) obs += __shfl_xor(obs, o);
    mf = (float)(a.D - obs) / (float)a.D;
  }
  const float eps = a.latent_dist == CS_DIST_NORMAL_WITH_EPS ? 1e-8f : 0.0f;
  const float kc = a.anneal == CS_ANNEAL_LINEAR_025 ? 0.25f : 1.0f;
  const bool beta = a.latent_dist == CS_DIST_BETA;
  // one positive parameter through tempering and annealing
  auto pos = [&](float s) {
    if (a.temperature > 0.0f) s = cs_softplus_ref(cs_softplus_inverse_ref(s * a.temperature - eps)) + eps;
    if (a.anneal != CS_ANNEAL_NONE) {
      const float k = 1.0f / s;
      const float km = (1.0f - mf) + (a.anneal == CS_ANNEAL_LINEAR_025 ? (k * kc) * mf : k * mf);
      s = cs_softplus_ref(cs_softplus_inverse_ref(s * km - eps)) + eps;
    }
    return s;
  };
  for (int t = lane; t < a.K * a.Zd; t += 64) {
    const int k = t / a.Zd, z = t - k * a.Zd;
    float* qr = a.q + (b * a.Kp + k) * 2 * a.Zd;
    if (beta) {
      qr[z] = pos(qr[z]);
    } else if (a.anneal != CS_ANNEAL_NONE) {
      qr[z] = qr[z] * ((1.0f - mf) + 0.0f * mf);
    }
    qr[a.Zd + z] = pos(qr[a.Zd + z]);
  }
}

extern "C" int cs_proposal_transform(float* q, const uint8_t* M, int64_t B, int32_t K, int32_t Kp, int32_t Zd, int32_t D,
                                     int32_t latent_dist, int32_t prior, float temperature, int32_t anneal, void* stream) {
  CS_REQUIRE(q && B >= 0 && K >= 1 && Kp >= K && Zd >= 1 && D >= 1, "bad argument");
  CS_REQUIRE(anneal >= CS_ANNEAL_NONE && anneal <= CS_ANNEAL_LINEAR_025, "bad anneal type");
  CS_REQUIRE(latent_dist == CS_DIST_NORMAL || latent_dist == CS_DIST_NORMAL_WITH_EPS || latent_dist == CS_DIST_BETA,
             "latent distribution must be normal / normal_with_eps / beta");
  if (anneal != CS_ANNEAL_NONE) {
    CS_REQUIRE(M, "annealing needs the mask");
    // the reference defines the annealing for (normal posterior, normal prior) and (Beta, uniform) only
    if ((latent_dist == CS_DIST_BETA) != (prior == CS_PRIOR_UNIFORM))
      return cs_fail(CS_EUNSUPPORTED, "annealing is not defined for this posterior / prior pair");
  }
  if (B == 0 || (!(temperature > 0.0f) && anneal == CS_ANNEAL_NONE)) return CS_OK;
  CsPropTransformArgs a;
  a.q = q; a.M = M; a.B = B; a.K = K; a.Kp = Kp; a.Zd = Zd; a.D = D; a.latent_dist = latent_dist;
  a.temperature = temperature; a.anneal = anneal;
  hipLaunchKernelGGL(cs_proposal_transform_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  CS_HIP(hipGetLastError());
  return CS_OK;
}

// q[b,k] = the prior as a proposal: N(0, 1), or Beta(1, 1) for the uniform prior
// (standard_importance_resampling, vae_imputation.py:920-930).
__global__ void __launch_bounds__(256) cs_prior_proposals_kernel(float* q, int64_t n, int Zd, int beta) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < n) q[t] = (beta || ((t / Zd) & 1)) ? 1.0f : 0.0f;
}

extern "C" int cs_prior_proposals(float* q, int64_t B, int32_t Kp, int32_t Zd, int32_t latent_dist, void* stream) {
  CS_REQUIRE(q && B >= 0 && Kp >= 1 && Zd >= 1, "bad argument");
  CS_REQUIRE(latent_dist == CS_DIST_NORMAL || latent_dist == CS_DIST_NORMAL_WITH_EPS || latent_dist == CS_DIST_BETA,
             "latent distribution must be normal / normal_with_eps / beta");
  const int64_t n = B * Kp * 2 * Zd;
  if (n == 0) return CS_OK;
  hipLaunchKernelGGL(cs_prior_proposals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q,
                     n, Zd, latent_dist == CS_DIST_BETA ? 1 : 0);
  CS_HIP(hipGetLastError());
  return CS_OK;
}

extern "C" int cs_lair_encode(const cs_mlp_t* enc, const cs_lair_cfg* c, const cs_lair_bufs* b, void* stream) {
  if (int rc = check_cfg(c, b)) return rc;
  if (c->B == 0) return CS_OK;  // empty batch: nothing to do (its tensors have no storage)
  CS_REQUIRE(enc && b->X && b->q, "null argument");
  CS_REQUIRE(enc->dev.in_dim == c->D && enc->dev.out_dim == 2 * c->Zd && enc->dev.split == 2,
             "encoder shape does not match the configuration");
  CsMlpArgs a;
  std::memset(&a, 0, sizeof(a));
  a.net = enc->dev;
  a.mode = CS_MLP_ENC;
  a.rows = (int64_t)c->B * c->K;
  a.in = b->X;
  a.in_stride = c->D;
  a.q = b->q;
  a.K = c->K; a.Kp = c->K + c->R; a.Zd = c->Zd; a.B = c->B; a.I = c->I; a.D = c->D;
  a.latent_dist = c->latent_dist;
  if (int rc = cs_mlp_launch(a, (hipStream_t)stream)) return rc;
  if (c->proposal_temperature > 0.0f || c->proposal_anneal != CS_ANNEAL_NONE)
    return cs_proposal_transform(b->q, b->M, c->B, c->K, c->K + c->R, c->Zd, c->D, c->latent_dist, c->prior,
                                 c->proposal_temperature, c->proposal_anneal, stream);
  return CS_OK;
}

extern "C" int cs_lair_decode(const cs_mlp_t* dec, const cs_lair_cfg* c, const cs_lair_bufs* b, const cs_rng* rng,
                              void* stream) {
  if (int rc = check_cfg(c, b)) return rc;
  if (c->B == 0) return CS_OK;  // empty batch: nothing to do (its tensors have no storage)
  CS_REQUIRE(dec && b->X && b->M && b->q && b->Z && b->gen && b->gen_ll, "null argument");
  const int np = cs_gen_nparams(c->gen_dist);
  CS_REQUIRE(dec->dev.in_dim == c->Zd && dec->dev.out_dim == np * c->D && dec->dev.split == np,
             "decoder shape does not match the configuration");
  CsMlpArgs a;
  std::memset(&a, 0, sizeof(a));
  a.net = dec->dev;
  a.mode = CS_MLP_DEC;
  a.K = c->K; a.Kp = c->K + c->R; a.Zd = c->Zd; a.B = c->B; a.I = c->I; a.D = c->D;
  a.rows = (int64_t)c->I * c->B * a.Kp;
  a.q = b->q;
  a.gen_dist = c->gen_dist;
  a.latent_dist = c->latent_dist;
  a.eps_z = b->eps_z;
  a.rng = cs_make_philox(rng);
  a.Z = b->Z;
  a.gen = b->gen;
  a.gen_ll = b->gen_ll;
  a.X = b->X;
  a.M = b->M;
  return cs_mlp_launch(a, (hipStream_t)stream);
}

static int lair_resample_impute(const cs_lair_cfg* c, const cs_lair_bufs* b, const cs_rng* rng, float* x_copy, void* stream);
extern "C" int cs_lair_resample_impute(const cs_lair_cfg* c, const cs_lair_bufs* b, const cs_rng* rng, void* stream) {
  return lair_resample_impute(c, b, rng, nullptr, stream);
}
// x_copy: optional [B,K_impute,D], receives the new imputations beside X (a run that keeps every iteration's)
static int lair_resample_impute(const cs_lair_cfg* c, const cs_lair_bufs* b, const cs_rng* rng, float* x_copy, void* stream) {
  if (int rc = check_cfg(c, b)) return rc;
  if (c->B == 0) return CS_OK;  // empty batch: nothing to do (its tensors have no storage)
  CS_REQUIRE(b->X && b->M && b->q && b->Z && b->gen && b->gen_ll && b->log_w && b->norm_log_w && b->idx &&
                 b->ess && b->tess && b->perplexity,
             "null argument");
  CsResampleArgs a;
  std::memset(&a, 0, sizeof(a));
  a.cfg = *c;
  a.b = *b;
  a.rng = cs_make_philox(rng);
  a.x_copy = x_copy;
  a.Kp = c->K + c->R;
  a.S = c->I * a.Kp;
  a.Kimp = c->K_impute > 0 ? c->K_impute : c->K;
  CS_REQUIRE(a.Kimp <= c->K, "K_impute must be <= K");
  CS_REQUIRE(c->glr_pad_to == 0 || c->glr_pad_to >= a.Kp, "glr_pad_to must be >= K+R");
  // proposal count that the group layout of glr / dmis_within_groups is built for
  const int Kg = c->glr_pad_to > 0 && c->scheme == CS_RS_GLR ? c->glr_pad_to : a.Kp;
  if (c->scheme == CS_RS_GR) {
    a.G = 1; a.c = a.Kp;
  } else if (c->scheme == CS_RS_GLR) {
    CS_REQUIRE(c->cluster_size > 1 && Kg % c->cluster_size == 0, "glr needs cluster_size > 1 dividing K+R (or glr_pad_to)");
    a.c = c->cluster_size; a.G = Kg / a.c;
  } else if (c->scheme == CS_RS_LR || c->scheme == CS_RS_IR) {
    a.c = 1; a.G = a.Kp;  // one draw per chain among I candidates
  } else {
    return cs_fail(CS_EUNSUPPORTED, "unknown resampling scheme");
  }
  if (c->weighting == CS_W_DMIS) {
    a.mixG = 1; a.mixc = a.Kp;
  } else if (c->weighting == CS_W_DMIS_WITHIN_GROUPS) {
    CS_REQUIRE(c->cluster_size > 1 && Kg % c->cluster_size == 0, "grouped weights need cluster_size > 1 dividing K+R");
    a.mixc = c->cluster_size; a.mixG = Kg / a.mixc;
  } else if (c->weighting == CS_W_SMIS) {
    a.mixG = a.Kp; a.mixc = 1;
  } else {
    return cs_fail(CS_EUNSUPPORTED, "weighting scheme not implemented");
  }
  CS_REQUIRE(c->method == CS_RM_MULTINOMIAL || c->method == CS_RM_SYSTEMATIC, "bad resampling method");
  if (c->B == 0) return CS_OK;
  a.NS = a.G * a.c;
  // more than 64 proposals per example: the three-launch path
  if (a.S > 64 || g_force_large) return cs_lair_resample_large(a, (hipStream_t)stream);
  // 16-byte staging / density loops when the rows allow it
  const bool vec = (c->Zd % 4 == 0) && (((uintptr_t)b->q | (uintptr_t)b->Z) & 15u) == 0;
  // staged rows padded so that the examples (q) and proposals (Z) of a wavefront start in different banks
  const int qn = a.Kp * c->Zd;
  a.q_stride = vec ? (((qn >> 2) | 1) << 2) : (qn | 1);
  a.z_stride = vec ? c->Zd + 4 : c->Zd + 1;
  // floats of LDS: loc and 1/(2 var) rows (a third array of that size for Beta posteriors), latents, indices,
  // component log-densities, per-component constants
  const int per_example = (c->latent_dist == CS_DIST_BETA ? 3 : 2) * a.q_stride + a.S * a.z_stride + 2 * a.NS +
                          a.S * a.mixc + a.Kp;
  a.P = 1; a.logP = 0;
  while (a.P < a.S) { a.P <<= 1; ++a.logP; }
  a.E = 64 / a.P;
  // more examples per wavefront fill more lanes, but the LDS working set limits how many wavefronts are resident
  // and the kernel lives on hiding one wavefront's memory round trips behind the others: up to 10 KB per wavefront
  // measured best (MiniBooNE: 4 examples of 5 proposals, 8.8 KB, 18 resident wavefronts per CU, 136 us per 65 536
  // examples against 151 with 5 examples and 159 with 3; HEPMASS / GAS: 2 examples of 17 proposals, 9.9 KB)
  static const int lds_kb = getenv("CSVAE_RS_LDS_KB") ? atoi(getenv("CSVAE_RS_LDS_KB")) : 10;  // (measurement switch)
  while (a.E > 1 && (size_t)a.E * per_example * 4 > (size_t)lds_kb * 1024) --a.E;
  // small batches: fewer examples per wavefront, so that the launch spreads over the chip and the loops of a
  // wavefront (staging, imputation: a memory round trip per pass) are one pass long
  {
    const int64_t e_cap = (c->B + 2047) / 2048;
    if (a.E > e_cap) a.E = (int)e_cap;
  }
  CS_REQUIRE((size_t)a.E * per_example * 4 <= 64 * 1024, "proposal set too large for the LDS working set");
  a.lds_per_wave = a.E * per_example;
  const int DQ = (c->D + 3) >> 2;
  // ranges of the multiply-high divisions: index * divisor < 2^32
  CS_REQUIRE((uint64_t)64 * c->Zd * c->Zd < (1ull << 32) && (uint64_t)64 * DQ * DQ < (1ull << 32), "D or Zd too large");
  a.dZd = cs_mkdiv(c->Zd); a.dZq = cs_mkdiv(vec ? c->Zd >> 2 : 1); a.dS = cs_mkdiv(a.S); a.dKp = cs_mkdiv(a.Kp);
  a.dNS = cs_mkdiv(a.NS); a.dDQ = cs_mkdiv(DQ); a.dKimp = cs_mkdiv(a.Kimp);
  a.dMC = cs_mkdiv(a.mixc); a.dmixG = cs_mkdiv(a.mixG); a.dc = cs_mkdiv(a.c);
  const unsigned grid = (unsigned)((c->B + a.E - 1) / a.E);
  if (vec)
    hipLaunchKernelGGL(cs_lair_resample_kernel<true>, dim3(grid), dim3(64), (size_t)a.lds_per_wave * 4, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(cs_lair_resample_kernel<false>, dim3(grid), dim3(64), (size_t)a.lds_per_wave * 4, (hipStream_t)stream, a);
  CS_HIP(hipGetLastError());
  return CS_OK;
}

static int lair_step(const cs_mlp_t* enc, const cs_mlp_t* dec, const cs_lair_cfg* c, const cs_lair_bufs* b, const cs_rng* rng,
                     float* x_copy, void* stream) {
  if (int rc = cs_lair_encode(enc, c, b, stream)) return rc;
  if (int rc = cs_lair_decode(dec, c, b, rng, stream)) return rc;
  return lair_resample_impute(c, b, rng, x_copy, stream);
}
extern "C" int cs_lair_step(const cs_mlp_t* enc, const cs_mlp_t* dec, const cs_lair_cfg* c, const cs_lair_bufs* b,
                            const cs_rng* rng, void* stream) {
  return lair_step(enc, dec, c, b, rng, nullptr, stream);
}

// ----------------------------------------------------------------------------------------
// Generator parameterisation of given latents, and sampling + imputation from generator
// parameters: the x_m | z step of pseudo_gibbs_iteration (vae_imputation.py:28-45), of the
// initial 'vae_samples' imputation (test_step_vae_sampling.py:279-287 with vae.py:166-178)
// and of the final decode of LAIR (test_step_vae_sampling.py:2019-2026).
// ----------------------------------------------------------------------------------------
extern "C" int cs_decode_params(const cs_mlp_t* dec, const float* Z, int64_t rows, int32_t gen_dist, float* gen,
                                void* stream) {
  CS_REQUIRE(dec && Z && gen && rows >= 0, "bad argument");
  CS_REQUIRE(dec->dev.split == cs_gen_nparams(gen_dist), "decoder output split does not match the distribution");
  CsMlpArgs a;
  std::memset(&a, 0, sizeof(a));
  a.net = dec->dev;
  a.mode = CS_MLP_DEC;
  a.rows = rows;
  a.in = Z;
  a.in_stride = dec->dev.in_dim;
  a.K = 1; a.Kp = 1; a.B = 1; a.I = 1;
  a.Zd = dec->dev.in_dim;
  a.D = dec->dev.out_dim / dec->dev.split;
  a.gen_dist = gen_dist;
  a.gen = gen;
  return cs_mlp_launch(a, (hipStream_t)stream);
}

struct CsSampleImputeArgs {
  const float* gen; int gen_dist; const float* eps; const float* chi2; CsPhilox rng; uint32_t stream_id;
  float* X; const uint8_t* M; int64_t rows; int K, D; const float* clip_min; const float* clip_max;
  int flags;
};

__global__ void __launch_bounds__(256) cs_sample_impute_kernel(const CsSampleImputeArgs a) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.rows * a.D) return;
  const int64_t row = t / a.D;
  const int d = (int)(t - row * a.D);
  const int64_t b = row / a.K;
  float xm;
  if (a.gen_dist == CS_DIST_BERN) {
    const float u = a.eps ? a.eps[t] : cs_uniform_elem(a.rng, a.stream_id, a.rng.row_offset * a.K + row, d);
    xm = (u < cs_sigmoid(a.gen[t])) ? 1.0f : 0.0f;
  } else if (a.gen_dist > CS_DIST_BERN) {  // Student-t, logit-normal
    const int np = cs_gen_nparams(a.gen_dist);
    const float* g = a.gen + row * np * a.D + d;
    const float e = a.eps ? a.eps[t] : cs_normal_elem(a.rng, a.stream_id, a.rng.row_offset * a.K + row, d);
    float df = 1.0f, c = 1.0f;
    if (np == 3) {
      df = g[2 * a.D];
      c = a.chi2 ? a.chi2[t] : cs_chi2_sample(a.rng, a.stream_id, a.rng.row_offset * a.K + row, d, df);
    }
    xm = cs_gen_sample(a.gen_dist, g[0], g[a.D], df, e, c);
  } else {
    const float e = a.eps ? a.eps[t] : cs_normal_elem(a.rng, a.stream_id, a.rng.row_offset * a.K + row, d);
    xm = a.gen[row * 2 * a.D + d] + a.gen[row * 2 * a.D + a.D + d] * e;
  }
  // pseudo-Gibbs safety (vae_imputation.py:34-41): clip the sample, keep the old value where
  // the sample is not finite; then blend
  if (a.clip_min) xm = (xm != xm) ? xm : fmaxf(xm, a.clip_min[d]);
  if (a.clip_max) xm = (xm != xm) ? xm : fminf(xm, a.clip_max[d]);
  float xn = xm;
  if (a.M) {
    const float m = (float)a.M[b * a.D + d];
    float xo = a.X[t];
    if ((a.flags & 1) && m == 0.0f) xo = 0.0f;
    if ((a.flags & 2) && !(fabsf(xm) < INFINITY)) xm = xo;
    xn = xo * m + xm * (1.0f - m);
  }
  a.X[t] = xn;
}

extern "C" int cs_sample_impute(const float* gen, int32_t gen_dist, const float* eps, const cs_rng* rng,
                                uint32_t stream_id, float* X, const uint8_t* M, int64_t B, int32_t K, int32_t D,
                                const float* clip_min, const float* clip_max, int32_t flags, void* stream) {
  return cs_sample_impute2(gen, gen_dist, eps, nullptr, rng, stream_id, X, M, B, K, D, clip_min, clip_max, flags, stream);
}

extern "C" int cs_sample_impute2(const float* gen, int32_t gen_dist, const float* eps, const float* chi2, const cs_rng* rng,
                                 uint32_t stream_id, float* X, const uint8_t* M, int64_t B, int32_t K, int32_t D,
                                 const float* clip_min, const float* clip_max, int32_t flags, void* stream) {
  CS_REQUIRE(gen && X && B >= 0 && K >= 1 && D >= 1, "bad argument");
  CS_REQUIRE((gen_dist >= CS_DIST_NORMAL && gen_dist <= CS_DIST_BERN) || cs_dist_is_studentt(gen_dist) ||
                 gen_dist == CS_DIST_LOGITNORMAL_WITH_EPS,
             "bad generator distribution");
  if (B == 0) return CS_OK;
  CsSampleImputeArgs a;
  a.gen = gen; a.gen_dist = gen_dist; a.eps = eps; a.chi2 = chi2; a.rng = cs_make_philox(rng); a.stream_id = stream_id;
  a.X = X; a.M = M; a.rows = B * K; a.K = K; a.D = D; a.clip_min = clip_min; a.clip_max = clip_max;
  a.flags = flags;
  const int64_t n = a.rows * D;
  hipLaunchKernelGGL(cs_sample_impute_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  CS_HIP(hipGetLastError());
  return CS_OK;
}

// ----------------------------------------------------------------------------------------
// T LAIR iterations in one call: the loop `for t in range(num_iterations)` of
// TestVAELatentAdaptiveImportanceResampling.sampler (test_step_vae_sampling.py:1887-1962) for callers that need
// no host-side work between the iterations.  Iteration t runs exactly cs_lair_step with rng.iteration + t; its
// proposals and log-weights go straight into the t-th block of the stores when those are given (the proposals
// kept for the final resampling, :1938-1940), its statistics into the t-th row of the statistics stores.
// Injected draws, when given, hold T consecutive blocks of the per-iteration shapes.
// ----------------------------------------------------------------------------------------
extern "C" int cs_lair_run(const cs_mlp_t* enc, const cs_mlp_t* dec, const cs_lair_cfg* c, const cs_lair_bufs* b0,
                           const cs_rng* rng, int32_t T, const cs_lair_run_out* out, void* stream) {
  return cs_lair_run_observed(enc, dec, c, b0, rng, T, out, nullptr, stream);
}

extern "C" int cs_lair_run_observed(const cs_mlp_t* enc, const cs_mlp_t* dec, const cs_lair_cfg* c, const cs_lair_bufs* b0,
                                    const cs_rng* rng, int32_t T, const cs_lair_run_out* out, const cs_run_observe* obs,
                                    void* stream) {
  CS_REQUIRE(c && b0 && T >= 0, "bad argument");
  if (int rc = check_cfg(c, b0)) return rc;
  if (obs) CS_REQUIRE(!b0->X_out && (c->K_impute == 0 || c->K_impute == c->K), "an observed run imputes all chains in place");
  CsObserver observer;
  if (int rc = observer.begin(enc, obs, c->B, c->K, c->D, c->Zd, c->latent_dist)) return rc;
  const int64_t B = c->B, Kp = c->K + c->R, I = c->I, Zd = c->Zd;
  const int64_t Kimp = c->K_impute > 0 ? c->K_impute : c->K;
  const int64_t Kg = c->glr_pad_to > 0 && c->scheme == CS_RS_GLR ? c->glr_pad_to : Kp;
  // per-iteration element counts of the injected draws (layouts of cs_lair_bufs)
  const int64_t n_eps_z = I * B * Kp * Zd, n_eps_x = B * Kimp * c->D;
  const int64_t n_u_res = B * Kg;                                                      // multinomial gr / glr
  const int64_t n_u_sys = c->scheme == CS_RS_GR ? B : (c->scheme == CS_RS_GLR ? B * (Kg / (c->cluster_size > 0 ? c->cluster_size : 1)) : B * Kp);
  const int64_t n_e_res = B * Kp * I;                                                  // lr / ir, multinomial
  cs_rng r = rng ? *rng : cs_rng{0, 0, 0};
  for (int32_t t = 0; t < T; ++t) {
    cs_lair_bufs b = *b0;
    if (b.eps_z) b.eps_z += t * n_eps_z;
    if (b.eps_x) b.eps_x += t * n_eps_x;
    if (b.chi2_x) b.chi2_x += t * n_eps_x;
    if (b.u_res) b.u_res += t * n_u_res;
    if (b.u_res2) b.u_res2 += t * n_u_res;
    if (b.u_sys) b.u_sys += t * n_u_sys;
    if (b.u_sys2) b.u_sys2 += t * n_u_sys;
    if (b.e_res) b.e_res += t * n_e_res;
    if (b.e_res2) b.e_res2 += t * n_e_res;
    if (out) {
      if (out->Z_store) b.Z = out->Z_store + t * I * B * Kp * Zd;
      if (out->log_w_store) b.log_w = out->log_w_store + t * I * B * Kp;
      if (out->ess_store) b.ess = out->ess_store + t * B;
      if (out->tess_store) b.tess = out->tess_store + t * B;
      if (out->perplexity_store) b.perplexity = out->perplexity_store + t * B;
    }
    cs_rng rt = r;
    rt.iteration = r.iteration + (uint32_t)t;
    // (the imputation kernel writes the kept copy itself: no copy launch in the chain)
    float* kept = obs && obs->X_store ? obs->X_store + (int64_t)t * B * c->K * c->D : nullptr;
    int rc = lair_step(enc, dec, c, &b, &rt, kept, stream);
    if (rc == CS_OK) rc = observer.iteration(t, b.X, b.M, (hipStream_t)stream, kept != nullptr);
    if (rc != CS_OK) {
      (void)observer.finish((hipStream_t)stream);
      return rc;
    }
  }
  return observer.finish((hipStream_t)stream);
}

// ----------------------------------------------------------------------------------------
// LAIR final resampling (test_step_vae_sampling.py:1998-2009): per example a Categorical over
// all S = T*I*K' stored proposals, n draws, gather of the chosen latents.
// One wavefront per example.  The CDF is the float32 running sum in index order divided by the
// total with the last entry forced to 1 (torch.multinomial), kept in a caller workspace; each
// draw is a left bisection with its float64 uniform.
// ----------------------------------------------------------------------------------------
struct CsFinalArgs {
  const float* logw;   // [TI,B,Kp]
  const float* Zp;     // [TI,B,Kp,Zd]
  const double* u;     // [B,n] or null
  CsPhilox rng;
  float* cdf;          // [B,S] workspace
  int64_t* idx;        // [B,n]
  float* Zout;         // [B,n,Zd]
  int64_t B, TI, n;
  int Kp, Zd;
};

__global__ void __launch_bounds__(256) cs_final_resample_kernel(const CsFinalArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const int64_t S = a.TI * a.Kp;
  auto lw_at = [&](int64_t s) { const int64_t ti = s / a.Kp; return a.logw[(ti * a.B + b) * a.Kp + (s - ti * a.Kp)]; };
  // Categorical(logits): l - logsumexp(l), then softmax
  float m = -INFINITY;
  for (int64_t s = lane; s < S; s += 64) m = fmaxf(m, lw_at(s));
  m = wave_max(m);
  if (!(fabsf(m) < INFINITY)) m = 0.f;
  float sum = 0.f;
  for (int64_t s = lane; s < S; s += 64) sum += expf(lw_at(s) - m);
  const float lse = logf(wave_sum(sum)) + m;
  float m2 = -INFINITY;
  for (int64_t s = lane; s < S; s += 64) m2 = fmaxf(m2, lw_at(s) - lse);
  m2 = wave_max(m2);
  float sum2 = 0.f;
  for (int64_t s = lane; s < S; s += 64) sum2 += expf((lw_at(s) - lse) - m2);
  sum2 = wave_sum(sum2);
  // sequential running sum, 64 entries per round
  float* cdf = a.cdf + b * S;
  float run = 0.f;
  for (int64_t s0 = 0; s0 < S; s0 += 64) {
    const int64_t s = s0 + lane;
    const float p = s < S ? expf((lw_at(s) - lse) - m2) / sum2 : 0.f;
    float cum = 0.f;
    const int cnt = (int)((S - s0) < 64 ? (S - s0) : 64);
    for (int jj = 0; jj < cnt; ++jj) {
      run += __shfl(p, jj);
      if (lane == jj) cum = run;
    }
    if (s < S) cdf[s] = cum;
  }
  const float total = run;
  for (int64_t s = lane; s < S; s += 64) cdf[s] = (s == S - 1) ? 1.0f : cdf[s] / total;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (int64_t t = lane; t < a.n; t += 64) {
    const double u = a.u ? a.u[b * a.n + t]
                         : cs_uniform64_elem(a.rng, CS_STREAM_UFINAL, a.rng.row_offset + b, (uint32_t)t);
    int64_t lo = 0, hi = S;
    while (hi - lo > 0) {
      const int64_t mid = lo + (hi - lo) / 2;
      if ((double)cdf[mid] < u) lo = mid + 1; else hi = mid;
    }
    if (lo > S - 1) lo = S - 1;
    a.idx[b * a.n + t] = lo;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (int64_t e = lane; e < a.n * a.Zd; e += 64) {
    const int64_t t = e / a.Zd;
    const int d = (int)(e - t * a.Zd);
    const int64_t s = a.idx[b * a.n + t];
    const int64_t ti = s / a.Kp;
    a.Zout[(b * a.n + t) * a.Zd + d] = a.Zp[((ti * a.B + b) * a.Kp + (s - ti * a.Kp)) * a.Zd + d];
  }
}

extern "C" int cs_final_resample(const float* logw, const float* Zprops, int64_t TI, int64_t B, int32_t Kp, int32_t Zd,
                                 const double* u, const cs_rng* rng, int64_t n, float* cdf_ws, int64_t* idx,
                                 float* Zout, void* stream) {
  CS_REQUIRE(logw && Zprops && cdf_ws && idx && Zout, "null argument");
  CS_REQUIRE(TI >= 1 && B >= 0 && Kp >= 1 && Zd >= 1 && n >= 1, "bad sizes");
  if (B == 0) return CS_OK;
  CsFinalArgs a;
  a.logw = logw; a.Zp = Zprops; a.u = u; a.rng = cs_make_philox(rng); a.cdf = cdf_ws; a.idx = idx; a.Zout = Zout;
  a.B = B; a.TI = TI; a.n = n; a.Kp = Kp; a.Zd = Zd;
  hipLaunchKernelGGL(cs_final_resample_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  CS_HIP(hipGetLastError());
  return CS_OK;
}

// ----------------------------------------------------------------------------------------
// cs_final_resample for callers that may not allocate (code under HIP graph capture, serving loops):
// the CDF workspace is a fixed array in the code object's device memory instead of a caller buffer.
// Every GPU holds its own copy of the array at its own address; the address is looked up once per
// device (first call there) and host code only ever uses that looked-up pointer.
// ----------------------------------------------------------------------------------------
#define CS_FINAL_STATIC_FLOATS (8ll << 20)  // 32 MiB per GPU: B * TI * K' <= 8 388 608
static __device__ __align__(256) float cs_final_static_cdf[CS_FINAL_STATIC_FLOATS];

static std::mutex g_final_static_mutex;
static float* g_final_static_ptr[64];  // per device ordinal; nullptr = not looked up yet

extern "C" int cs_final_resample_static(const float* logw, const float* Zprops, int64_t TI, int64_t B, int32_t Kp,
                                        int32_t Zd, const double* u, const cs_rng* rng, int64_t n, int64_t* idx,
                                        float* Zout, void* stream) {
  CS_REQUIRE(logw && Zprops && idx && Zout, "null argument");
  CS_REQUIRE(TI >= 1 && B >= 0 && Kp >= 1 && Zd >= 1 && n >= 1, "bad sizes");
  CS_REQUIRE(TI <= CS_FINAL_STATIC_FLOATS && B <= CS_FINAL_STATIC_FLOATS &&
                 B * TI <= CS_FINAL_STATIC_FLOATS && B * TI * Kp <= CS_FINAL_STATIC_FLOATS,
             "cs_final_resample_static: B * TI * K' exceeds the static workspace of 8 388 608 floats; use "
             "cs_final_resample with a caller workspace");
  if (B == 0) return CS_OK;
  int dev = -1;
  CS_HIP(hipGetDevice(&dev));
  CS_REQUIRE(dev >= 0 && dev < 64, "device ordinal out of range");
  float* cdf;
  {
    std::lock_guard<std::mutex> lock(g_final_static_mutex);
    if (!g_final_static_ptr[dev]) {
      void* p = nullptr;
      CS_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(cs_final_static_cdf)));
      CS_REQUIRE(p != nullptr, "hipGetSymbolAddress returned a null address");
      g_final_static_ptr[dev] = (float*)p;
    }
    cdf = g_final_static_ptr[dev];
  }
  // every entry of an example's CDF row is written before it is read: nothing has to start at zero
  CsFinalArgs a;
  a.logw = logw; a.Zp = Zprops; a.u = u; a.rng = cs_make_philox(rng); a.cdf = cdf; a.idx = idx; a.Zout = Zout;
  a.B = B; a.TI = TI; a.n = n; a.Kp = Kp; a.Zd = Zd;
  hipLaunchKernelGGL(cs_final_resample_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  CS_HIP(hipGetLastError());
  return CS_OK;
}
