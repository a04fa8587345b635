// Per-iteration imputation diagnostics of the sampling harness, computed on the device so that no
// imputation leaves HBM during the sampler loop.
//
// Replaces imputation_rmse_metric (irwg/sampling/imputation_metrics.py:85-102),
// imputation_mae_metric (:133-143) and the closed-form part of imputation_latent_wass (:200-213),
// as called once per iteration from test_step (irwg/sampling/test_step_vae_sampling.py:431-445).
#include "cs_common.h"
#include "cs_mlp.h"

#include <cstdlib>
#include <cstring>
#include <mutex>

namespace {

// One wavefront per (example, chain): squared / absolute error over the missing entries divided by
// their count (a count of 0 is replaced by 1, :92-93).
__global__ void __launch_bounds__(256) cs_imputation_error_kernel(const float* X, const float* Xt, const uint8_t* M,
                                                                  int64_t rows, int K, int D, float* rmse, float* mae) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t b = row / K;
  const float* x = X + row * D;
  const float* t = Xt + b * D;
  const uint8_t* m = M + b * D;
  float se = 0.0f, ae = 0.0f, cnt = 0.0f;
  for (int d = lane; d < D; d += 64) {
    const float miss = m[d] ? 0.0f : 1.0f;
    const float df = x[d] - t[d];
    se += df * df * miss;  // (X_imp - X_true)**2 * ~M: a non-finite difference times 0 stays NaN, as in torch
    ae += fabsf(df) * miss;
    cnt += miss;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    se += __shfl_xor(se, o);
    ae += __shfl_xor(ae, o);
    cnt += __shfl_xor(cnt, o);
  }
  if (lane == 0) {
    if (cnt == 0.0f) cnt = 1.0f;
    if (rmse) rmse[row] = sqrtf(se / cnt);
    if (mae) mae[row] = ae / cnt;
  }
}

// Narrow data (D <= 64, the tabular sets): a wavefront per row would leave most lanes idle and every row's
// load a partial line.  A block of TR threads takes TR consecutive rows: their TR*D values are one contiguous
// run, fetched with whole-line loads into LDS (row stride DS odd, so the per-row walk below hits TR different
// banks), as are the true values and the masks of the examples the rows belong to; then one thread per row.
// The sum over d runs in order of d.
template <int TR>
__global__ void __launch_bounds__(TR) cs_imputation_error_tile_kernel(const float* X, const float* Xt, const uint8_t* M,
                                                                      int64_t rows, int K, int D, int DS, int nb_max,
                                                                      float* rmse, float* mae) {
  extern __shared__ float lds[];
  float* l_x = lds;                  // [TR][DS]
  float* l_t = l_x + TR * DS;        // [nb_max][DS] true values
  float* l_m = l_t + nb_max * DS;    // [nb_max][DS] 1 = missing
  const int tid = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * TR;
  const int nr = (int)((rows - row0) < TR ? (rows - row0) : TR);
  const int64_t b0 = row0 / K;
  const int nb = (int)((row0 + nr - 1) / K - b0) + 1;
  const float* xs = X + row0 * D;
  for (int i = tid; i < nr * D; i += TR) {
    const int r = i / D, d = i - r * D;
    l_x[r * DS + d] = xs[i];
  }
  const float* ts = Xt + b0 * D;
  const uint8_t* ms = M + b0 * D;
  for (int i = tid; i < nb * D; i += TR) {
    const int r = i / D, d = i - r * D;
    l_t[r * DS + d] = ts[i];
    l_m[r * DS + d] = ms[i] ? 0.0f : 1.0f;
  }
  __syncthreads();
  if (tid >= nr) return;
  const int64_t row = row0 + tid;
  const int bl = (int)(row / K - b0);
  const float* x = l_x + tid * DS;
  const float* t = l_t + bl * DS;
  const float* m = l_m + bl * DS;
  float se = 0.0f, ae = 0.0f, cnt = 0.0f;
  for (int d = 0; d < D; ++d) {
    const float miss = m[d];
    const float df = x[d] - t[d];
    se += df * df * miss;
    ae += fabsf(df) * miss;
    cnt += miss;
  }
  if (cnt == 0.0f) cnt = 1.0f;
  if (rmse) rmse[row] = sqrtf(se / cnt);
  if (mae) mae[row] = ae / cnt;
}

// W[b,k] = sum_z (mu_i - mu_t)^2 + sum_z (var_i + var_t - 2 sqrt(var_i var_t)) between the Gaussian
// posteriors of the imputed and the true datapoint; q rows are (loc | scale).
__global__ void __launch_bounds__(256) cs_latent_wass_kernel(const float* q_imp, const float* q_true, int64_t rows, int K,
                                                             int Zd, float* out) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const int64_t b = row / K;
  const float* qi = q_imp + row * 2 * Zd;
  const float* qt = q_true + b * 2 * Zd;
  float w1 = 0.0f, w2 = 0.0f;
  for (int z = 0; z < Zd; ++z) {
    const float dm = qi[z] - qt[z];
    w1 += dm * dm;
    const float vi = qi[Zd + z] * qi[Zd + z], vt = qt[Zd + z] * qt[Zd + z];
    w2 += vi + vt - 2.0f * sqrtf(vi * vt);
  }
  out[row] = w1 + w2;
}

}  // namespace

extern "C" int cs_imputation_errors(const float* X, const float* X_true, const uint8_t* M, int64_t B, int32_t K, int32_t D,
                                    float* rmse, float* mae, void* stream) {
  CS_REQUIRE(X && X_true && M && (rmse || mae), "null argument");
  CS_REQUIRE(B >= 0 && K >= 1 && D >= 1, "bad sizes");
  const int64_t rows = B * K;
  if (rows == 0) return CS_OK;
  if (D <= 64) {
    constexpr int TR = 128;
    const int DS = D | 1;
    const int nb_max = (TR + K - 2) / K + 1;   // examples that TR consecutive rows can belong to
    const size_t lds = (size_t)(TR + 2 * nb_max) * DS * sizeof(float);
    hipLaunchKernelGGL(cs_imputation_error_tile_kernel<TR>, dim3((unsigned)((rows + TR - 1) / TR)), dim3(TR), lds,
                       (hipStream_t)stream, X, X_true, M, rows, K, D, DS, nb_max, rmse, mae);
  } else {
    hipLaunchKernelGGL(cs_imputation_error_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, X,
                       X_true, M, rows, K, D, rmse, mae);
  }
  CS_HIP(hipGetLastError());
  return CS_OK;
}

extern "C" int cs_latent_wasserstein(const float* q_imp, const float* q_true, int64_t B, int32_t K, int32_t Zd, float* out,
                                     void* stream) {
  CS_REQUIRE(q_imp && q_true && out, "null argument");
  CS_REQUIRE(B >= 0 && K >= 1 && Zd >= 1, "bad sizes");
  const int64_t rows = B * K;
  if (rows == 0) return CS_OK;
  hipLaunchKernelGGL(cs_latent_wass_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q_imp,
                     q_true, rows, K, Zd, out);
  CS_HIP(hipGetLastError());
  return CS_OK;
}

// ----------------------------------------------------------------------------------------
// The per-iteration work of the harness inside a run of several iterations (cs_run_observe): keep the
// imputations, compute the diagnostics (test_step_vae_sampling.py:386-391, 431-445).
// ----------------------------------------------------------------------------------------
namespace {
struct SideStream { hipStream_t stream = nullptr; hipEvent_t copied = nullptr, done = nullptr; };
SideStream g_side[64];
std::mutex g_side_mutex;
const bool g_observe_inline = getenv("CSVAE_OBSERVE_INLINE") != nullptr;  // measurements without the second stream
}  // namespace

int CsObserver::begin(const cs_mlp_t* enc_, const cs_run_observe* o_, int64_t B_, int32_t K_, int32_t D_, int32_t Zd_,
                      int32_t latent_dist_) {
  o = o_; enc = enc_; B = B_; K = K_; D = D_; Zd = Zd_; latent_dist = latent_dist_;
  side = nullptr;
  if (!o) return CS_OK;
  CS_REQUIRE(!(o->rmse_store || o->mae_store) || o->X_true, "rmse / mae stores need X_true");
  if (o->wass_store) {
    CS_REQUIRE(o->q_true && o->q_work && enc, "the latent distance needs the encoder, q_true and q_work");
    if (latent_dist == CS_DIST_BETA) return cs_fail(CS_EUNSUPPORTED, "latent Wasserstein distance: Gaussian posteriors only");
  }
  const bool diagnostics = o->rmse_store || o->mae_store || o->wass_store;
  if (!diagnostics || !o->X_store || (o->wass_store && enc->prog) || g_observe_inline || B == 0) return CS_OK;
  int dev = 0;
  CS_HIP(hipGetDevice(&dev));
  CS_REQUIRE(dev >= 0 && dev < 64, "device index out of range");
  std::lock_guard<std::mutex> lock(g_side_mutex);
  SideStream& s = g_side[dev];
  if (!s.stream) {
    CS_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    CS_HIP(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
    CS_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  }
  side = s.stream; copied = s.copied; done = s.done;
  return CS_OK;
}

int CsObserver::iteration(int32_t t, const float* X, const uint8_t* M, hipStream_t stream, bool already_kept) {
  if (!o || B == 0) return CS_OK;
  const int64_t chains = B * K;
  const float* Xd = X;            // what the diagnostics read
  hipStream_t ds = stream;        // and where
  if (o->X_store) {
    float* kept = o->X_store + (int64_t)t * chains * D;
    if (!already_kept)
      CS_HIP(hipMemcpyAsync(kept, X, (size_t)chains * D * sizeof(float), hipMemcpyDeviceToDevice, stream));
    if (side) {
      // (the events are the device's, shared by all callers: a wait refers to the record made just before it only
      // if no other thread records in between)
      std::lock_guard<std::mutex> lock(g_side_mutex);
      CS_HIP(hipEventRecord(copied, stream));
      CS_HIP(hipStreamWaitEvent(side, copied, 0));
      Xd = kept;
      ds = side;
    }
  }
  if (o->rmse_store || o->mae_store)
    if (int rc = cs_imputation_errors(Xd, o->X_true, M, B, K, D, o->rmse_store ? o->rmse_store + (int64_t)t * chains : nullptr,
                                      o->mae_store ? o->mae_store + (int64_t)t * chains : nullptr, ds))
      return rc;
  if (o->wass_store) {
    // VAE.compute_var_distr of the imputed data (imputation_metrics.py:203-207): the encoder on X as it is
    cs_lair_cfg c;
    std::memset(&c, 0, sizeof(c));
    c.B = (int32_t)B; c.K = K; c.R = 0; c.I = 1; c.D = D; c.Zd = Zd; c.latent_dist = latent_dist;
    c.gen_dist = CS_DIST_NORMAL; c.weight_alpha = 1.0f; c.prior = CS_PRIOR_STD_NORMAL;
    cs_lair_bufs b;
    std::memset(&b, 0, sizeof(b));
    b.X = const_cast<float*>(Xd); b.q = o->q_work;
    if (int rc = cs_lair_encode(enc, &c, &b, ds)) return rc;
    if (int rc = cs_latent_wasserstein(o->q_work, o->q_true, B, K, Zd, o->wass_store + (int64_t)t * chains, ds)) return rc;
  }
  return CS_OK;
}

int CsObserver::finish(hipStream_t stream) {
  if (!side) return CS_OK;
  std::lock_guard<std::mutex> lock(g_side_mutex);
  CS_HIP(hipEventRecord(done, side));
  CS_HIP(hipStreamWaitEvent(stream, done, 0));
  return CS_OK;
}
