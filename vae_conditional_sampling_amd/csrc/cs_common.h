// Shared device/host helpers for the conditional-sampling HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/csvae.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef __HIPCC__
// One 16-byte-per-lane global->LDS copy with the source addressed as scalar base + 32-bit lane
// offset and the LDS destination base in M0.  Written as asm because the compiler's selection of the
// builtin re-associates (base + constant) + lane offset into a 64-bit vector add per copy, and on this
// library's matrix kernels every vector instruction displaces exact-f32 matrix work; here the per-copy
// address arithmetic is scalar.  Every LDS-DMA copy of a kernel that uses it must go through this
// function, which sets M0 itself.
__device__ __forceinline__ void cs_dma16(const void* sbase, unsigned lane_off, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0"
               :
               : "s"(sbase), "v"(lane_off), "s"(lds_base)
               : "memory", "m0");  // M0 is written: the compiler must not keep a value of its own in it across the copy
}
__device__ __forceinline__ unsigned cs_scalar_u32(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
// Wait states between the last matrix instruction of a kernel's main loop and the first instruction that
// reads its accumulators.  The compiler pads its own code to the documented minimum counted from the ISSUE of
// the matrix instruction (12 states for an 8-pass instruction).  Behind a run of back-to-back matrix
// instructions issue runs ahead of execution, and that minimum proved too short: the disassembly of the build
// that failed shows the output layer's last v_mfma followed, 12 states later, by v_accvgpr_read of the tile's
// last register, and that register now and then lacked the final product (last bits of two output columns,
// a few launches in a hundred).  64 further states follow the final matrix instruction of every matrix kernel;
// the scheduling barriers on both sides keep every matrix instruction above and every reader below them
// (an asm statement alone orders memory accesses, not register-only instructions).
// tools/hazard_audit.py checks the distance in the shipped code objects (tests/test_hazard_audit.py).
__device__ __forceinline__ void cs_mfma_drain() {
#ifndef CS_NO_MFMA_DRAIN   // tools/hazard_audit.py can be pointed at a build without it: what the compiler emits alone
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#endif
}
#endif


#define CS_LOG_SQRT_2PI 0.91893853320467274178f

// ---- error reporting ---------------------------------------------------------------
void cs_set_error(const std::string& msg);
int cs_fail(int code, const std::string& msg);
#define CS_HIP(call)                                                                      \
  do {                                                                                    \
    hipError_t e__ = (call);                                                              \
    if (e__ != hipSuccess)                                                                \
      return cs_fail(CS_EHIP, std::string(#call) + ": " + hipGetErrorString(e__));        \
  } while (0)
#define CS_REQUIRE(cond, msg)                                                             \
  do {                                                                                    \
    if (!(cond)) return cs_fail(CS_EINVAL, std::string(msg) + " [" #cond "]");            \
  } while (0)

// ---- Philox4x32-10 counter RNG -----------------------------------------------------
// Draws are a pure function of (seed, stream, iteration, global row, element), so results do
// not depend on how examples are batched or sharded over GPUs.
struct CsPhilox {
  uint32_t seed_lo, seed_hi;
  uint32_t iter;       // sampler iteration
  int64_t row_offset;  // added to the local example index (rank shard offset)
};

enum CsStream : uint32_t {
  CS_STREAM_Z = 1,        // latent proposal noise, one row per (i, example, proposal)
  CS_STREAM_X = 2,        // imputation noise, one row per (example, chain)
  CS_STREAM_URES = 3,     // resampling uniforms, one row per example
  CS_STREAM_URES2 = 4,    // second resampling (resampled latents)
  CS_STREAM_ZPRIOR = 5,   // AC-MWG prior proposal
  CS_STREAM_UPRIOR = 6,   // AC-MWG prior-mixture Bernoulli
  CS_STREAM_UACC = 7,     // AC-MWG accept uniform
  CS_STREAM_HIST = 8,     // AC-MWG history index
  CS_STREAM_INIT_Z = 9,   // initial imputation latent
  CS_STREAM_INIT_X = 10,  // initial imputation noise
  CS_STREAM_UFINAL = 11,  // final resampling uniforms
};

__host__ __device__ inline void cs_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Block `blk` of the draw sequence of (stream, row): 4 x uint32.
__host__ __device__ inline void cs_philox_block(const CsPhilox& p, uint32_t stream, int64_t row,
                                                uint32_t blk, uint32_t out[4]) {
  uint64_t r = (uint64_t)(row);
  cs_philox4x32(blk, (uint32_t)r, (uint32_t)(r >> 32) ^ (stream << 24), p.iter, p.seed_lo,
                p.seed_hi, out);
}

// uint32 -> float uniform in (0,1): 24 mantissa bits, never 0 or 1.
__host__ __device__ inline float cs_u01f(uint32_t x) {
  return (float)(x >> 8) * 5.9604644775390625e-08f + 2.98023223876953125e-08f;
}
// 2 x uint32 -> double uniform in [0,1) with 53 bits.
__host__ __device__ inline double cs_u01d(uint32_t hi, uint32_t lo) {
  uint64_t v = (((uint64_t)hi << 32) | lo) >> 11;
  return (double)v * 1.1102230246251565e-16;
}

struct CsPhilox cs_make_philox(const cs_rng* r);

#ifdef __HIPCC__
// 4 standard normals from one Philox block (two Box-Muller pairs).  Hardware transcendental
// instructions (v_log_f32, v_sqrt_f32, v_sin_f32 / v_cos_f32, which take the angle in
// revolutions) are accurate to ~1e-6 absolute here, which is immaterial for random draws; every
// consumer (kernels and the cs_rng_normal dump used by the tests) goes through this function.
__device__ inline f32x4 cs_normal4(const uint32_t x[4]) {
  const float r0 = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(cs_u01f(x[0])));  // sqrt(-2 ln u)
  const float r1 = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(cs_u01f(x[2])));
  const float t0 = cs_u01f(x[1]), t1 = cs_u01f(x[3]);
  f32x4 n;
  n.x = r0 * __builtin_amdgcn_cosf(t0); n.y = r0 * __builtin_amdgcn_sinf(t0);
  n.z = r1 * __builtin_amdgcn_cosf(t1); n.w = r1 * __builtin_amdgcn_sinf(t1);
  return n;
}
// Element `e` of the standard-normal sequence of (stream,row).
__device__ inline float cs_normal_elem(const CsPhilox& p, uint32_t stream, int64_t row, uint32_t e) {
  uint32_t x[4];
  cs_philox_block(p, stream, row, e >> 2, x);
  f32x4 n = cs_normal4(x);
  uint32_t c = e & 3u;
  return c == 0 ? n.x : (c == 1 ? n.y : (c == 2 ? n.z : n.w));
}
__device__ inline float cs_uniform_elem(const CsPhilox& p, uint32_t stream, int64_t row, uint32_t e) {
  uint32_t x[4];
  cs_philox_block(p, stream, row, e >> 2, x);
  return cs_u01f(x[e & 3u]);
}
__device__ inline double cs_uniform64_elem(const CsPhilox& p, uint32_t stream, int64_t row, uint32_t e) {
  uint32_t x[4];
  cs_philox_block(p, stream, row, e >> 1, x);
  return (e & 1u) ? cs_u01d(x[2], x[3]) : cs_u01d(x[0], x[1]);
}

// Gamma(shape, 1) by Marsaglia-Tsang squeeze-free rejection; one Philox block per attempt
// (normal from words 0-1, acceptance uniform from word 2, small-shape boost uniform from word 3).
// Attempts are bounded: after 32 rejections (probability < 1e-40) the mode-like value d is taken.
__device__ inline float cs_gamma_sample(const CsPhilox& p, uint32_t stream, int64_t row, uint32_t blk0, float shape) {
  const float a = shape < 1.0f ? shape + 1.0f : shape;
  const float d = a - 1.0f / 3.0f, c = 1.0f / sqrtf(9.0f * d);
  float g = d, boost = 1.0f;
  for (uint32_t t = 0; t < 32; ++t) {
    uint32_t x[4];
    cs_philox_block(p, stream, row, blk0 + t, x);
    const f32x4 n = cs_normal4(x);
    if (t == 0 && shape < 1.0f) boost = powf(cs_u01f(x[3]), 1.0f / shape);
    const float v1 = 1.0f + c * n.x;
    if (v1 <= 0.0f) continue;
    const float v = v1 * v1 * v1;
    if (logf(cs_u01f(x[2])) < 0.5f * n.x * n.x + d - d * v + d * logf(v)) {
      g = d * v;
      break;
    }
  }
  return g * boost;
}
// Beta(alpha, beta) = Ga / (Ga + Gb), clamped to the open unit interval as torch's Dirichlet
// sampler does ([FLT_MIN, 1 - 2^-24]).  Element e of (stream,row) owns blocks [64e, 64e+64).
__device__ inline float cs_beta_sample(const CsPhilox& p, uint32_t stream, int64_t row, uint32_t e, float alpha, float beta) {
  const float ga = cs_gamma_sample(p, stream, row, 64u * e, alpha);
  const float gb = cs_gamma_sample(p, stream, row, 64u * e + 32u, beta);
  const float z = ga / (ga + gb);
  return fminf(fmaxf(z, 1.17549435e-38f), 0.99999994f);
}
// torch.xlogy: 0 where x == 0, else x * log(y)
__device__ inline float cs_xlogy(float x, float y) { return x == 0.0f ? 0.0f : x * logf(y); }

// exp / log / reciprocal through the hardware transcendental instructions (v_exp_f32 = 2^x,
// v_log_f32 = log2, v_rcp_f32; each accurate to 1 ulp).  The library versions cost 50-150
// instructions apiece, which the epilogues of the fused kernels pay per output element; these
// stay within a few ulp (the product x*log2(e) is formed in two parts so that large |x| does not
// lose relative accuracy).
__device__ inline float cs_expf(float x) {
  const float t = x * 1.44269502162933349609375f;
  const float r = __builtin_fmaf(x, 1.44269502162933349609375f, -t) + x * 1.92596299112661746e-8f;
  const float e = __builtin_amdgcn_exp2f(t);
  return __builtin_fmaf(e, r * 0.693147182464599609375f, e);
}
__device__ inline float cs_logf(float x) { return __builtin_amdgcn_logf(x) * 0.693147182464599609375f; }
// log(1 + u) for u in [0, 1]: log(w) * u / (w - 1) with w = fl(1 + u) cancels the rounding of w.
__device__ inline float cs_log1pf01(float u) {
  const float w = 1.0f + u;
  const float d = w - 1.0f;
  return d == 0.0f ? u : cs_logf(w) * (u * __builtin_amdgcn_rcpf(d));
}
// torch.nn.functional.softplus (beta 1, threshold 20): log(1 + e^x) = max(x, 0) + log1p(e^-|x|);
// above the threshold the correction is below half an ulp of x, so x itself comes out.
__device__ inline float cs_softplus(float x) { return fmaxf(x, 0.0f) + cs_log1pf01(cs_expf(-fabsf(x))); }
// torch.distributions.Normal.log_prob: -(x-loc)^2 / (2 scale^2) - log(scale) - log(sqrt(2 pi)).
__device__ inline float cs_normal_logpdf(float x, float loc, float scale) {
  float d = x - loc;
  return -(d * d) * __builtin_amdgcn_rcpf(2.0f * (scale * scale)) - cs_logf(scale) - CS_LOG_SQRT_2PI;
}
__device__ inline float cs_stdnormal_logpdf(float z) { return -(z * z) / 2.0f - CS_LOG_SQRT_2PI; }
// -BCEWithLogits(l, x) = x*l - softplus(l), in torch's stable form.
__device__ inline float cs_bernoulli_logpmf(float x, float l) {
  return -((1.0f - x) * l + fmaxf(-l, 0.0f) + cs_log1pf01(cs_expf(-fabsf(l))));
}
__device__ inline float cs_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + cs_expf(-x)); }

// ---- generator families of the LAIR path beyond normal / Bernoulli (irwg/models/vae.py:220-229, 244-247) ----
__host__ __device__ inline bool cs_dist_is_studentt(int d) { return d == CS_DIST_STUDENTT || d == CS_DIST_STUDENTT_EPS22; }
// parameter vectors of D entries per row in `gen`
__host__ __device__ inline int cs_gen_nparams(int d) { return d == CS_DIST_BERN ? 1 : (cs_dist_is_studentt(d) ? 3 : 2); }
// torch.distributions.StudentT.log_prob
__device__ inline float cs_studentt_logpdf(float x, float df, float loc, float scale) {
  const float y = (x - loc) / scale;
  const float Z = logf(scale) + 0.5f * logf(df) + 0.5f * 1.1447298858494002f + lgammaf(0.5f * df) - lgammaf(0.5f * (df + 1.0f));
  return -0.5f * (df + 1.0f) * log1pf(y * y / df) - Z;
}
// LogitNormal = SigmoidTransform of Normal(loc, scale) (irwg/utils/logit_normal.py): log_prob(v) = Normal.log_prob(u) +
// softplus(-u) + softplus(u) with u = logit(clamp(v, FLT_MIN, 1 - FLT_EPSILON))
__device__ inline float cs_logitnormal_logpdf(float v, float loc, float scale) {
  const float y = fminf(fmaxf(v, 1.17549435e-38f), 1.0f - 1.1920929e-7f);
  const float u = logf(y) - log1pf(-y);
  return cs_normal_logpdf(u, loc, scale) + cs_softplus(-u) + cs_softplus(u);
}
__device__ inline float cs_clipped_sigmoid(float x) {
  return fminf(fmaxf(1.0f / (1.0f + expf(-x)), 1.17549435e-38f), 1.0f - 1.1920929e-7f);
}
// log p(x | parameters) of one dimension; p2 only for Student-t (df)
__device__ inline float cs_gen_logpdf(int dist, float x, float p0, float p1, float p2) {
  if (dist == CS_DIST_BERN) return cs_bernoulli_logpmf(x, p0);
  if (cs_dist_is_studentt(dist)) return cs_studentt_logpdf(x, p2, p0, p1);
  if (dist == CS_DIST_LOGITNORMAL_WITH_EPS) return cs_logitnormal_logpdf(x, p0, p1);
  return cs_normal_logpdf(x, p0, p1);
}
// x ~ p(. | parameters) from a standard normal n (Bernoulli: a uniform) and, for Student-t, a chi-square(df) draw c
__device__ inline float cs_gen_sample(int dist, float p0, float p1, float p2, float n, float c) {
  if (dist == CS_DIST_BERN) return (n < cs_sigmoid(p0)) ? 1.0f : 0.0f;
  if (cs_dist_is_studentt(dist)) return p0 + p1 * (n * rsqrtf(c / p2));
  if (dist == CS_DIST_LOGITNORMAL_WITH_EPS) return cs_clipped_sigmoid(p0 + p1 * n);
  return p0 + p1 * n;
}
// chi-square(df) from the counter RNG: 2 * Gamma(df / 2), drawn from the companion stream (stream | 16) of the
// normal draws' stream; element e of (stream,row) owns its blocks [32e, 32e + 32)
__device__ inline float cs_chi2_sample(const CsPhilox& p, uint32_t stream, int64_t row, uint32_t e, float df) {
  return 2.0f * cs_gamma_sample(p, stream | 16u, row, 32u * e, 0.5f * df);
}
#endif
