// LAIR weights / resampling / imputation for proposal sets of more than 64 per example
// (importance_resampling_gibbs_iteration with many chains: K' = K + R in the hundreds or thousands,
// irwg/sampling/vae_imputation.py:605-726; weights irwg/utils/weights.py:37-85 and
// vae_imputation.py:380-458; resampling :284-378, 460-482, 804-834).
//
// The one-wavefront kernel of cs_lair.hip keeps an example's whole proposal set in the lanes of a wavefront.
// Here the stage is three launches:
//   cs_lair_logw_large_kernel      log-weights.  The deterministic mixture costs K' component densities per
//                                  proposal (K'^2 * Zd terms per example), so this launch is spread over
//                                  (example, tile of 128 proposals) workgroups; a thread owns one proposal and
//                                  walks the components, which the workgroup stages tile by tile in LDS.
//   cs_lair_resample_large_kernel  one workgroup per example: ESS / perplexity, normalisation within the
//                                  resampling groups, the float32 running-sum CDF of every group (sequential in
//                                  member order, as torch.multinomial builds it), and the draws.
//   cs_lair_impute_large_kernel    sampling of the missing values from the resampled proposals' generator
//                                  parameters and imputation, spread over (example, chain, 4 dimensions).
// Arithmetic follows the small kernel term by term (component densities summed over the latent dimensions in
// order, torch.logsumexp as max / sum / log, sequential CDF); reductions over many proposals are necessarily
// summed in a different order than torch's, which the float32 parity tolerance covers.
#include "cs_lair.h"

#define LG_TP 128  // proposals per workgroup of the log-weight launch
#define LG_TC 32   // mixture components staged per round
#define LG_RT 256  // threads of the resampling workgroup

namespace {

// ---- geometry of the resampling groups ---------------------------------------------------------------------
// gr / glr / lr: proposal (i, kp) is in group kp mod G; the members of a group in flattened '(i k)' order are
// (i, g + r*G), r < cg(g), i < I, so the member at position li = i*cg + r.  ir: the flattened axis is cut in
// runs of I (vae_imputation.py:463-464).  A padded layout (glr_pad_to) has G*c > K' slots: groups then hold
// fewer members than I*c, possibly none.
struct CsGroups {
  int ir, I, Kp, G, q, rem;
  __device__ __forceinline__ int cg(int g) const { return q + (g < rem ? 1 : 0); }
  __device__ __forceinline__ int size(int g) const { return ir ? I : I * cg(g); }
  __device__ __forceinline__ int offset(int g) const { return ir ? g * I : I * (g * q + (g < rem ? g : rem)); }
  __device__ __forceinline__ void locate(int s, int& g, int& li) const {
    if (ir) { g = s / I; li = s - g * I; return; }
    const int i = s / Kp, kp = s - i * Kp;
    g = kp % G;
    li = i * cg(g) + kp / G;
  }
  __device__ __forceinline__ int member(int g, int n) const {
    if (ir) return g * I + n;
    const int w = cg(g);
    const int i = n / w, r = n - i * w;
    return i * Kp + g + r * G;
  }
};

__device__ __forceinline__ CsGroups cs_groups(const CsResampleArgs& a) {
  CsGroups gr;
  gr.ir = a.cfg.scheme == CS_RS_IR;
  gr.I = a.cfg.I; gr.Kp = a.Kp; gr.G = a.G;
  gr.q = a.Kp / a.G; gr.rem = a.Kp % a.G;
  return gr;
}

// torch.logsumexp on a running (max, sum of exp) pair: one new term
__device__ __forceinline__ void lse_add(float c, float& m, float& sum) {
  if (c > m) {
    sum *= expf(m - c);   // m = -inf at the start: 0 * exp(-inf) = 0
    m = c;
  }
  sum += (c == m) ? 1.0f : expf(c - m);   // c == m also covers infinite terms (inf - inf)
}
__device__ __forceinline__ float lse_finish(float m, float sum) { return logf(sum) + m; }

// test hook: 1 keeps the scalar log-weight kernel where the 16-byte form would run
int g_logw_scalar = 0;

}  // namespace

extern "C" int cs_debug_force_scalar_logw(int on) {
  g_logw_scalar = on ? 1 : 0;
  return CS_OK;
}

// ----------------------------------------------------------------------------------------------------------
// log w[i,b,k'] = log p(x_o | z) + log p(z) - log q_mix(z)            (weights.py:46-57, 69-85)
// UNIFORM: every proposal of the example has the same components (dmis: all K'), staged in LDS by the
// workgroup; otherwise each thread reads the few components of its own mixture group (dmis_within_groups,
// vae_imputation.py:407-447; smis: the proposal's own posterior, weights.py:5-35) from q directly.
// ----------------------------------------------------------------------------------------------------------
template <bool BETA, bool UNIFORM>
__global__ void __launch_bounds__(LG_TP) cs_lair_logw_large_kernel(const CsResampleArgs a, int tiles) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int B = a.cfg.B, Zd = a.cfg.Zd, Kp = a.Kp, S = a.S;
  const int64_t b = blockIdx.x / tiles;
  const int s = (int)(blockIdx.x - b * tiles) * LG_TP + tid;
  const bool act = s < S;
  const int ZS = Zd | 1;                       // odd row stride: the threads' rows start in different banks
  float* l_z = lds;                            // [TP][ZS]  z          (Beta: log z)
  float* l_z1 = l_z + LG_TP * ZS;              // [TP][ZS]  Beta only: log(1 - z)
  float* l_c0 = l_z1 + (BETA ? LG_TP * ZS : 0);  // [TC][Zd]  loc        | alpha - 1
  float* l_c1 = l_c0 + LG_TC * Zd;             // [TC][Zd]  1/(2 s^2)  | beta - 1
  float* l_c2 = l_c1 + LG_TC * Zd;             // [TC][Zd]  log s      | log-normaliser of the dimension
  float* l_L = l_c2 + LG_TC * Zd;              // [TC]      Normal: sum_d log s + Zd log sqrt(2 pi)
  const int i_s = act ? s / Kp : 0, kp_s = act ? s - i_s * Kp : 0;
  const int64_t rho = ((int64_t)i_s * B + b) * Kp + kp_s;
  float* zr = l_z + tid * ZS;
  float* zr1 = l_z1 + tid * ZS;

  // the thread's own latent: prior log-density (summed over the dimensions in order) and its LDS row
  float pl = 0.0f;
  if (act) {
    const float* zg = a.b.Z + rho * Zd;
    for (int d = 0; d < Zd; ++d) {
      const float z = zg[d];
      if (a.cfg.prior == CS_PRIOR_UNIFORM) pl += (z >= 0.0f && z < 1.0f) ? 0.0f : -INFINITY;  // Uniform(0, 1 + 1e-8)
      else pl += cs_stdnormal_logpdf(z);
      if (BETA) { zr[d] = logf(z); zr1[d] = logf(1.0f - z); }
      else zr[d] = z;
    }
  }
  // (loc, scale) -> the three staged values of one component dimension.  Beta: (alpha, beta) -> alpha - 1,
  // beta - 1 and the log-normaliser of torch's Dirichlet.log_prob
  auto stage1 = [&](float loc, float sc, float& o0, float& o1, float& o2) {
    if (BETA) {
      o0 = loc - 1.0f;
      o1 = sc - 1.0f;
      o2 = lgammaf(loc + sc) - (lgammaf(loc) + lgammaf(sc));
    } else {
      o0 = loc;
      o1 = __builtin_amdgcn_rcpf(2.0f * (sc * sc));
      o2 = cs_logf(sc);
    }
  };
  float m = -INFINITY, sum = 0.0f, single = 0.0f;
  int ncomp = 0;
  if (UNIFORM) {
    for (int j0 = 0; j0 < Kp; j0 += LG_TC) {
      const int tc = (Kp - j0) < LG_TC ? (Kp - j0) : LG_TC;
      __syncthreads();   // the previous tile is consumed
      const float* qb = a.b.q + (b * Kp + j0) * 2 * Zd;
      for (int t = tid; t < tc * Zd; t += LG_TP) {
        const int jj = cs_div(t, a.dZd), d = t - jj * Zd;
        stage1(qb[(int64_t)jj * 2 * Zd + d], qb[(int64_t)jj * 2 * Zd + Zd + d], l_c0[t], l_c1[t], l_c2[t]);
      }
      __syncthreads();
      if (!BETA) {
        if (tid < tc) {
          float Lj = 0.0f;
          for (int d = 0; d < Zd; ++d) Lj += l_c2[tid * Zd + d];
          l_L[tid] = Lj + (float)Zd * CS_LOG_SQRT_2PI;
        }
        __syncthreads();
      }
      if (act) {
        for (int jj = 0; jj < tc; ++jj) {
          const float* c0 = l_c0 + jj * Zd;
          const float* c1 = l_c1 + jj * Zd;
          float cj = 0.0f;
          if (BETA) {
            const float* c2 = l_c2 + jj * Zd;
            for (int d = 0; d < Zd; ++d) {
              const float t0 = c0[d] == 0.0f ? 0.0f : c0[d] * zr[d];     // torch.xlogy
              const float t1 = c1[d] == 0.0f ? 0.0f : c1[d] * zr1[d];
              cj += (t0 + t1) + c2[d];
            }
          } else {
            for (int d = 0; d < Zd; ++d) {
              const float df = zr[d] - c0[d];
              cj = __builtin_fmaf(-(df * df), c1[d], cj);   // spelled out: the 16-byte form below must round the same way
            }
            cj -= l_L[jj];
          }
          lse_add(cj, m, sum);
        }
      }
    }
    ncomp = Kp;
  } else if (act) {
    // components of proposal kp: j = kp mod mixG + mixG*cc, cc < mixc; j >= K' is a member that a padded
    // layout lacks (smis: mixG = K', mixc = 1, so j = kp)
    const int gm = kp_s % a.mixG;
    for (int cc = 0; cc < a.mixc; ++cc) {
      const int j = gm + a.mixG * cc;
      if (j >= Kp) break;
      const float* qr = a.b.q + (b * Kp + j) * 2 * Zd;
      float cj = 0.0f, Lj = 0.0f;
      for (int d = 0; d < Zd; ++d) {
        float o0, o1, o2;
        stage1(qr[d], qr[Zd + d], o0, o1, o2);
        if (BETA) {
          const float t0 = o0 == 0.0f ? 0.0f : o0 * zr[d];
          const float t1 = o1 == 0.0f ? 0.0f : o1 * zr1[d];
          cj += (t0 + t1) + o2;
        } else {
          const float df = zr[d] - o0;
          cj -= (df * df) * o1;
          Lj += o2;
        }
      }
      if (!BETA) cj -= Lj + (float)Zd * CS_LOG_SQRT_2PI;
      single = cj;
      lse_add(cj, m, sum);
      ++ncomp;
    }
  }
  if (act) {
    const float ql = (a.cfg.weighting == CS_W_SMIS) ? single : lse_finish(m, sum) - logf((float)ncomp);
    a.b.log_w[rho] = a.b.gen_ll[rho] + pl - ql;
  }
}

// ----------------------------------------------------------------------------------------------------------
// One workgroup per example.  `arr` holds one float per proposal in group-contiguous order: the log-weights,
// then in place the normalised log-weights, the probabilities, and the running sums.
// ----------------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float blk_max(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float blk_sum(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
// torch.logsumexp of f(arr[lo..hi)) by the whole workgroup
template <class F>
__device__ __forceinline__ float blk_lse(const float* arr, int lo, int hi, float* red, F f) {
  float m = -INFINITY;
  for (int j = lo + (int)threadIdx.x; j < hi; j += LG_RT) m = fmaxf(m, f(arr[j]));
  m = blk_max(m, red);
  if (!(fabsf(m) < INFINITY)) m = 0.0f;
  float t = 0.0f;
  for (int j = lo + (int)threadIdx.x; j < hi; j += LG_RT) t += expf(f(arr[j]) - m);
  return logf(blk_sum(t, red)) + m;
}

}  // namespace

__global__ void __launch_bounds__(LG_RT) cs_lair_resample_large_kernel(const CsResampleArgs a) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int B = a.cfg.B, Zd = a.cfg.Zd, I = a.cfg.I, Kp = a.Kp, S = a.S, G = a.G, c = a.c, NS = a.NS;
  const int64_t b = blockIdx.x;
  float* arr = lds;              // [S]
  float* red = lds + ((S + 3) & ~3);  // [4]
  const CsGroups gr = cs_groups(a);
  const bool ir = gr.ir;
  const float alpha = a.cfg.weight_alpha;

  for (int s = tid; s < S; s += LG_RT) {
    const int i = s / Kp, kp = s - i * Kp;
    int g, li;
    gr.locate(s, g, li);
    arr[gr.offset(g) + li] = a.b.log_w[((int64_t)i * B + b) * Kp + kp];
  }
  __syncthreads();

  // ---- ESS / perplexity over all S proposals (vae_imputation.py:708-719) ----
  {
    const float lse_all = blk_lse(arr, 0, S, red, [](float x) { return x; });
    const float ess = expf(-blk_lse(arr, 0, S, red, [=](float x) { return 2.0f * (x - lse_all); }));
    float t = 0.0f;
    for (int j = tid; j < S; j += LG_RT) {
      const float lnw = arr[j] - lse_all;
      t += lnw * expf(lnw);
    }
    const float perp = expf(-blk_sum(t, red));
    float tess = ess;
    if (alpha != 1.0f) {
      const float lse_t = blk_lse(arr, 0, S, red, [=](float x) { return x * alpha; });
      tess = expf(-blk_lse(arr, 0, S, red, [=](float x) { return 2.0f * (x * alpha - lse_t); }));
    }
    if (tid == 0) {
      a.b.ess[b] = ess;
      a.b.tess[b] = tess;
      a.b.perplexity[b] = perp;
    }
  }
  __syncthreads();
  if (alpha != 1.0f) {
    for (int j = tid; j < S; j += LG_RT) arr[j] *= alpha;   // transformed log-weights (vae_imputation.py:643-644)
    __syncthreads();
  }

  // ---- normalisation and probabilities within the resampling groups ----
  // ir returns the (transformed) log-weights themselves, in '(i k)' order (vae_imputation.py:479-480); a padded
  // glr returns the normalised ones in '(i k)' order (:371-374); otherwise '(g i c)'
  const bool flat_nlw = ir || a.cfg.glr_pad_to > 0;
  const bool multinomial = a.cfg.method == CS_RM_MULTINOMIAL;
  const bool race = multinomial && (a.cfg.scheme == CS_RS_LR || ir);   // one draw per group: argmax(p / Exp(1))
  float* nlw_out = a.b.norm_log_w + b * S;
  auto nlw_slot = [&](int g, int li) { return flat_nlw ? gr.member(g, li) : g * (I * c) + li; };
  if (G < 64) {
    // few groups: the workgroup handles them one after the other
    for (int g = 0; g < G; ++g) {
      const int lo = gr.offset(g), hi = lo + gr.size(g);
      if (hi == lo) continue;
      const float lse = blk_lse(arr, lo, hi, red, [](float x) { return x; });
      float mx = -INFINITY;
      for (int j = lo + tid; j < hi; j += LG_RT) {
        const float x = arr[j], nl = x - lse;
        nlw_out[nlw_slot(g, j - lo)] = ir ? x : nl;
        arr[j] = nl;
        mx = fmaxf(mx, nl);
      }
      // Categorical.probs = softmax(normalised logits); systematic: exp(normalised log-weights) renormalised
      // by their sum (:813-817)
      mx = multinomial ? blk_max(mx, red) : 0.0f;
      float t = 0.0f;
      for (int j = lo + tid; j < hi; j += LG_RT) t += expf(arr[j] - mx);
      const float tot = blk_sum(t, red);
      for (int j = lo + tid; j < hi; j += LG_RT) arr[j] = expf(arr[j] - mx) / tot;
    }
    __syncthreads();
    // sequential float32 running sum inside each group, in member order
    if (!race && tid < G) {
      const int lo = gr.offset(tid), hi = lo + gr.size(tid);
      float run = 0.0f;
      for (int j = lo; j < hi; ++j) { run += arr[j]; arr[j] = run; }
    }
  } else {
    // many groups: a thread per group, everything in member order
    for (int g = tid; g < G; g += LG_RT) {
      const int lo = gr.offset(g), hi = lo + gr.size(g);
      if (hi == lo) continue;
      float m = -INFINITY;
      for (int j = lo; j < hi; ++j) m = fmaxf(m, arr[j]);
      if (!(fabsf(m) < INFINITY)) m = 0.0f;
      float t = 0.0f;
      for (int j = lo; j < hi; ++j) t += expf(arr[j] - m);
      const float lse = logf(t) + m;
      float mx = -INFINITY;
      for (int j = lo; j < hi; ++j) {
        const float x = arr[j], nl = x - lse;
        nlw_out[nlw_slot(g, j - lo)] = ir ? x : nl;
        arr[j] = nl;
        mx = fmaxf(mx, nl);
      }
      if (!multinomial) mx = 0.0f;
      t = 0.0f;
      for (int j = lo; j < hi; ++j) t += expf(arr[j] - mx);
      float run = 0.0f;
      for (int j = lo; j < hi; ++j) {
        const float p = expf(arr[j] - mx) / t;
        run += p;
        arr[j] = race ? p : run;
      }
    }
  }
  __syncthreads();

  // ---- draws ----
  const int ndraw_sets = (a.b.latents_resampled != nullptr) ? 2 : 1;
  const int64_t row = a.rng.row_offset + b;
  for (int set = 0; set < ndraw_sets; ++set) {
    const uint32_t stream = set == 0 ? CS_STREAM_URES : CS_STREAM_URES2;
    int64_t* out = set == 0 ? a.b.idx + b * NS : (a.b.idx2 ? a.b.idx2 + b * NS : nullptr);
    // slot <- flattened '(i k)' index src; the second set also gathers the latents (the second, independent
    // resampling of vae_imputation.py:649-650)
    auto emit = [&](int slot, int src) {
      if (out) out[slot] = src;
      if (set == 1 && src >= 0) {
        const int i = src / Kp, kp = src - i * Kp;
        const float* zs = a.b.Z + (((int64_t)i * B + b) * Kp + kp) * Zd;
        float* zo = a.b.latents_resampled + (b * NS + slot) * Zd;
        for (int d = 0; d < Zd; ++d) zo[d] = zs[d];
      }
    };
    if (race) {
      // torch.multinomial(p, 1) = argmax(p / e), e ~ Exp(1) per candidate, first maximum
      const float* ein = set == 0 ? a.b.e_res : a.b.e_res2;
      for (int g = tid; g < G; g += LG_RT) {
        const int lo = gr.offset(g), n = gr.size(g);
        float best = -INFINITY;
        int best_li = 0;
        for (int li = 0; li < n; ++li) {
          const int el = g * I + li;   // element of the [K', I] block of exponentials of the example
          float ex;
          if (ein) {
            ex = ein[b * S + el];
          } else {
            uint32_t x[4];
            cs_philox_block(a.rng, stream, row, (uint32_t)(el >> 2), x);
            ex = -logf(1.0f - cs_u01f(x[el & 3]));
          }
          const float v = arr[lo + li] / ex;
          if (v > best) { best = v; best_li = li; }
        }
        emit(g, n ? gr.member(g, best_li) : -1);
      }
    } else if (multinomial) {
      // torch.multinomial: running sum / total with the last entry forced to 1, left bisection with float64 uniforms.
      // Draw d = g*c + t of the example lands in output slot t*G + g ('c b k d -> b (c k) d')
      const double* uin = set == 0 ? a.b.u_res : a.b.u_res2;
      const int ND = G * c;
      for (int d = tid; d < ND; d += LG_RT) {
        const int g = d / c, t = d - g * c;
        const int lo = gr.offset(g), n = gr.size(g);
        const double u = uin ? uin[b * ND + d] : cs_uniform64_elem(a.rng, stream, row, (uint32_t)d);
        int src = -1;
        if (n > 0) {
          const float total = arr[lo + n - 1];
          int left = 0, right = n;
          while (right - left > 0) {
            const int mid = left + (right - left) / 2;
            const float cn = (mid == n - 1) ? 1.0f : arr[lo + mid] / total;
            if ((double)cn < u) left = mid + 1; else right = mid;
          }
          if (left > n - 1) left = n - 1;
          src = gr.member(g, left);
        }
        emit(t * G + g, src);
      }
    } else {
      // systematic: n_j = clamp(floor(cumsum(p)_j * c + U), 0, c); index[t] = #{members : n_j <= t}  (:818-834)
      const float* uin = set == 0 ? a.b.u_sys : a.b.u_sys2;
      const int ND = G * c;
      for (int d = tid; d < ND; d += LG_RT) {
        const int g = d / c, t = d - g * c;
        const int lo = gr.offset(g), n = gr.size(g);
        const float U = uin ? uin[b * G + g] : cs_uniform_elem(a.rng, stream, row, (uint32_t)g);
        int src = -1;
        if (n > 0) {
          int left = 0, right = n;   // first member with n_j > t (n_j is non-decreasing)
          while (right - left > 0) {
            const int mid = left + (right - left) / 2;
            float nf = floorf(arr[lo + mid] * (float)c + U);
            nf = fminf(fmaxf(nf, 0.0f), (float)c);
            if ((int)nf <= t) left = mid + 1; else right = mid;
          }
          if (left > n - 1) left = n - 1;   // the reference would index out of bounds here
          src = gr.member(g, left);
        }
        emit(t * G + g, src);
      }
    }
  }
}

__global__ void __launch_bounds__(256) cs_lair_impute_large_kernel(const CsResampleArgs a, int DQ, int64_t items) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= items) return;
  const int64_t en = t / DQ;
  const int dq = (int)(t - en * DQ);
  const int64_t be = en / a.Kimp;
  const int n = (int)(en - be * a.Kimp);
  const int src = (int)a.b.idx[be * a.NS + n];
  if (src < 0) return;   // slot of an empty group of a padded layout
  cs_lair_impute_item(a, be, n, dq, src);
}

// The common case of the launch above -- Normal posteriors, every proposal under all K' components (dmis), Zd a
// multiple of 4, 16-byte aligned q and Z -- with 16-byte accesses: a thread keeps its latent in registers (ZQ = Zd / 4
// quads; ZQ = 0: any Zd, the latent in its LDS row), the staged components are read a quad of dimensions at a time
// (one address per wavefront: a broadcast), two components run side by side so that their dependent sums overlap.
// The same terms in the same order as the kernel above: the results are bit-identical.
template <int ZQ>
__global__ void __launch_bounds__(LG_TP) cs_lair_logw_large_vec_kernel(const CsResampleArgs a, int tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int B = a.cfg.B, Zd = a.cfg.Zd, Kp = a.Kp, S = a.S;
  const int Zq = Zd >> 2;
  const int64_t b = blockIdx.x / tiles;
  const int s = (int)(blockIdx.x - b * tiles) * LG_TP + tid;
  const bool act = s < S;
  const int ZS = Zd + 4;                       // row stride of the latents in LDS (ZQ = 0): 16-byte aligned rows whose
                                               // 16-byte slots fall into different banks for 16 consecutive threads
  float* l_z = lds;                            // [TP][ZS]  (ZQ = 0 only)
  float* l_c0 = l_z + (ZQ == 0 ? LG_TP * ZS : 0);  // [TC][Zd]  loc
  float* l_c1 = l_c0 + LG_TC * Zd;             // [TC][Zd]  1/(2 s^2)
  float* l_c2 = l_c1 + LG_TC * Zd;             // [TC][Zd]  log s
  float* l_L = l_c2 + LG_TC * Zd;              // [TC]      sum_d log s + Zd log sqrt(2 pi)
  const int i_s = act ? s / Kp : 0, kp_s = act ? s - i_s * Kp : 0;
  const int64_t rho = ((int64_t)i_s * B + b) * Kp + kp_s;
  f32x4 zreg[ZQ > 0 ? ZQ : 1];
  f32x4* zr = (f32x4*)(l_z + tid * ZS);
  float pl = 0.0f;
  if (act) {
    const f32x4* zg = (const f32x4*)(a.b.Z + rho * Zd);
    for (int dq = 0; dq < (ZQ > 0 ? ZQ : Zq); ++dq) {
      const f32x4 z = zg[dq];
      if (a.cfg.prior == CS_PRIOR_UNIFORM) {
#pragma unroll
        for (int r = 0; r < 4; ++r) pl += (z[r] >= 0.0f && z[r] < 1.0f) ? 0.0f : -INFINITY;
      } else {
        pl += cs_stdnormal_logpdf(z.x); pl += cs_stdnormal_logpdf(z.y);
        pl += cs_stdnormal_logpdf(z.z); pl += cs_stdnormal_logpdf(z.w);
      }
      if (ZQ > 0) zreg[dq] = z;
      else zr[dq] = z;
    }
  }
  float m = -INFINITY, sum = 0.0f;
  // log-density of component jj of the staged tile at the thread's latent, summed over the dimensions in order
  auto dens2 = [&](int jj, bool two, float& ca, float& cb) {
    const f32x4* c0a = (const f32x4*)(l_c0 + jj * Zd);
    const f32x4* c1a = (const f32x4*)(l_c1 + jj * Zd);
    const f32x4* c0b = c0a + (two ? Zq : 0);
    const f32x4* c1b = c1a + (two ? Zq : 0);
    ca = 0.0f; cb = 0.0f;
#pragma unroll
    for (int dq = 0; dq < (ZQ > 0 ? ZQ : Zq); ++dq) {
      const f32x4 z = ZQ > 0 ? zreg[dq] : zr[dq];
      const f32x4 ma = c0a[dq], wa = c1a[dq], mb = c0b[dq], wb = c1b[dq];
      float df;
      df = z.x - ma.x; ca = __builtin_fmaf(-(df * df), wa.x, ca);
      df = z.x - mb.x; cb = __builtin_fmaf(-(df * df), wb.x, cb);
      df = z.y - ma.y; ca = __builtin_fmaf(-(df * df), wa.y, ca);
      df = z.y - mb.y; cb = __builtin_fmaf(-(df * df), wb.y, cb);
      df = z.z - ma.z; ca = __builtin_fmaf(-(df * df), wa.z, ca);
      df = z.z - mb.z; cb = __builtin_fmaf(-(df * df), wb.z, cb);
      df = z.w - ma.w; ca = __builtin_fmaf(-(df * df), wa.w, ca);
      df = z.w - mb.w; cb = __builtin_fmaf(-(df * df), wb.w, cb);
    }
    ca -= l_L[jj];
    cb -= l_L[jj + (two ? 1 : 0)];
  };
  for (int j0 = 0; j0 < Kp; j0 += LG_TC) {
    const int tc = (Kp - j0) < LG_TC ? (Kp - j0) : LG_TC;
    __syncthreads();   // the previous tile is consumed
    const float* qb = a.b.q + (b * Kp + j0) * 2 * Zd;
    for (int t = tid; t < tc * Zq; t += LG_TP) {
      const int jj = cs_div(t, a.dZq), dq = t - jj * Zq;
      const f32x4 loc = *(const f32x4*)(qb + (int64_t)jj * 2 * Zd + 4 * dq);
      const f32x4 sc = *(const f32x4*)(qb + (int64_t)jj * 2 * Zd + Zd + 4 * dq);
      f32x4 o1, o2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o1[r] = __builtin_amdgcn_rcpf(2.0f * (sc[r] * sc[r]));
        o2[r] = cs_logf(sc[r]);
      }
      *(f32x4*)(l_c0 + jj * Zd + 4 * dq) = loc;
      *(f32x4*)(l_c1 + jj * Zd + 4 * dq) = o1;
      *(f32x4*)(l_c2 + jj * Zd + 4 * dq) = o2;
    }
    __syncthreads();
    if (tid < tc) {
      float Lj = 0.0f;
      for (int d = 0; d < Zd; ++d) Lj += l_c2[tid * Zd + d];
      l_L[tid] = Lj + (float)Zd * CS_LOG_SQRT_2PI;
    }
    __syncthreads();
    if (act) {
      int jj = 0;
      for (; jj + 1 < tc; jj += 2) {
        float ca, cb;
        dens2(jj, true, ca, cb);
        lse_add(ca, m, sum);
        lse_add(cb, m, sum);
      }
      if (jj < tc) {
        float ca, cb;
        dens2(jj, false, ca, cb);
        lse_add(ca, m, sum);
      }
    }
  }
  if (act) a.b.log_w[rho] = a.b.gen_ll[rho] + pl - (lse_finish(m, sum) - logf((float)Kp));
}

// ----------------------------------------------------------------------------------------------------------
int cs_lair_resample_large(CsResampleArgs& a, hipStream_t stream) {
  const cs_lair_cfg& c = a.cfg;
  const int64_t B = c.B;
  const int Zd = c.Zd, S = a.S;
  CS_REQUIRE((uint64_t)S * (uint64_t)a.Kp < (1ull << 32) && (uint64_t)(LG_TC + 1) * Zd * Zd < (1ull << 32),
             "proposal set too large");
  a.dKp = cs_mkdiv(a.Kp);
  a.dZd = cs_mkdiv(Zd);
  const bool beta = c.latent_dist == CS_DIST_BETA;
  const bool uniform = c.weighting == CS_W_DMIS;
  // log-weights
  {
    const int ZS = Zd | 1;
    const size_t lds = ((size_t)LG_TP * ZS * (beta ? 2 : 1) + (size_t)3 * LG_TC * Zd + LG_TC) * 4;
    CS_REQUIRE(lds <= 64 * 1024, "latent dimension too large for proposal sets of more than 64 per example");
    const int tiles = (S + LG_TP - 1) / LG_TP;
    CS_REQUIRE(B * tiles < (1ll << 31), "batch too large");
    const dim3 grid((unsigned)(B * tiles));
    const bool vec = !beta && uniform && (Zd & 3) == 0 && (((uintptr_t)a.b.q | (uintptr_t)a.b.Z) & 15) == 0 && !g_logw_scalar;
    if (vec) {
      a.dZq = cs_mkdiv(Zd >> 2);
      const size_t ldsv = ((size_t)(Zd == 16 || Zd == 32 ? 0 : LG_TP * (Zd + 4)) + (size_t)3 * LG_TC * Zd + LG_TC) * 4;
      CS_REQUIRE(ldsv <= 64 * 1024, "latent dimension too large for proposal sets of more than 64 per example");
      if (Zd == 16) hipLaunchKernelGGL((cs_lair_logw_large_vec_kernel<4>), grid, dim3(LG_TP), ldsv, stream, a, tiles);
      else if (Zd == 32) hipLaunchKernelGGL((cs_lair_logw_large_vec_kernel<8>), grid, dim3(LG_TP), ldsv, stream, a, tiles);
      else hipLaunchKernelGGL((cs_lair_logw_large_vec_kernel<0>), grid, dim3(LG_TP), ldsv, stream, a, tiles);
    } else if (beta) {
      if (uniform) hipLaunchKernelGGL((cs_lair_logw_large_kernel<true, true>), grid, dim3(LG_TP), lds, stream, a, tiles);
      else hipLaunchKernelGGL((cs_lair_logw_large_kernel<true, false>), grid, dim3(LG_TP), lds, stream, a, tiles);
    } else {
      if (uniform) hipLaunchKernelGGL((cs_lair_logw_large_kernel<false, true>), grid, dim3(LG_TP), lds, stream, a, tiles);
      else hipLaunchKernelGGL((cs_lair_logw_large_kernel<false, false>), grid, dim3(LG_TP), lds, stream, a, tiles);
    }
    CS_HIP(hipGetLastError());
  }
  // resampling
  {
    const size_t lds = ((size_t)((S + 3) & ~3) + 4) * 4;
    CS_REQUIRE(lds <= 160 * 1024, "more than 40 000 proposals per example");
    if (lds > 64 * 1024)
      CS_HIP(hipFuncSetAttribute((const void*)cs_lair_resample_large_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
    hipLaunchKernelGGL(cs_lair_resample_large_kernel, dim3((unsigned)B), dim3(LG_RT), lds, stream, a);
    CS_HIP(hipGetLastError());
  }
  // imputation
  {
    const int DQ = (c.D + 3) >> 2;
    const int64_t items = B * a.Kimp * DQ;
    CS_REQUIRE((items + 255) / 256 < (1ll << 31), "batch too large");
    hipLaunchKernelGGL(cs_lair_impute_large_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, a, DQ,
                       items);
    CS_HIP(hipGetLastError());
  }
  return CS_OK;
}
