/* Plain-C restatement of the integer / index work on the conditional-sampling path.
 *
 * TEST INFRASTRUCTURE ONLY: this is the checker for the HIP kernels' bit-exact parts; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 *   history_*            ImputationHistoryQueue_with_restrictedavailability
 *                        (irwg/sampling/utils.py:83-237): availability counters, tail, full.
 *   multinomial_left     torch.multinomial(probs, n, replacement=True) on CPU as used by
 *                        Categorical.sample (irwg/sampling/vae_imputation.py:292-293,349-350;
 *                        test_step_vae_sampling.py:2001-2005): float32 running sum, divide by the
 *                        total, last entry = 1, left bisection with float64 uniforms.
 *   systematic_index     _systematic_sample (irwg/sampling/vae_imputation.py:826-830) from given
 *                        probabilities and the row's uniform.
 *   philox4x32_10 ...    the counter RNG of the HIP library (no counterpart in the reference,
 *                        which uses torch's generators): key/counter layout of csrc/cs_common.h.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* ---------------- history queue ---------------- */
typedef struct {
  int64_t T, B, K;
  int64_t tail; /* -1 = empty */
  int32_t full;
} cs_hist;

void history_init(cs_hist* h, int64_t T, int64_t B, int64_t K, int64_t* avail) {
  h->T = T; h->B = B; h->K = K; h->tail = -1; h->full = 0;
  memset(avail, 0, sizeof(int64_t) * (size_t)(B * K));
}
int64_t history_len(const cs_hist* h) { return h->full ? h->T : h->tail + 1; }
/* utils.py:94-99 */
void history_update_available(const cs_hist* h, int64_t* avail, const uint8_t* mask) {
  const int64_t len = history_len(h);
  for (int64_t i = 0; i < h->B * h->K; ++i)
    if (mask[i]) avail[i] = len;
}
/* utils.py:101-113 (the data write itself is a plain copy and is left to the caller) */
int64_t history_enqueue(cs_hist* h, int64_t* avail) {
  if (h->full)
    for (int64_t i = 0; i < h->B * h->K; ++i) avail[i] = avail[i] > 0 ? avail[i] - 1 : 0;
  h->tail = (h->tail + 1) % h->T;
  if (h->tail == h->T - 1) h->full = 1;
  return h->tail;
}
/* utils.py:222-232: draw bound and slot of a draw r */
int64_t history_draw_bound(const int64_t* avail, int64_t i) { return avail[i] < 1 ? 1 : avail[i]; }
int64_t history_slot(const cs_hist* h, int64_t r) { return h->full ? (r + h->tail + 1) % h->T : r; }

/* ---------------- multinomial with replacement ---------------- */
void multinomial_left(const float* probs, int64_t S, const double* u, int64_t n, int64_t* idx, float* cum) {
  float s = 0.0f;
  for (int64_t j = 0; j < S; ++j) { s += probs[j]; cum[j] = s; }
  for (int64_t j = 0; j < S; ++j) cum[j] = cum[j] / s;
  cum[S - 1] = 1.0f;
  for (int64_t t = 0; t < n; ++t) {
    int64_t lo = 0, hi = S;
    while (hi - lo > 0) {
      const int64_t mid = lo + (hi - lo) / 2;
      if ((double)cum[mid] < u[t]) lo = mid + 1; else hi = mid;
    }
    idx[t] = lo;
  }
}

/* ---------------- systematic resampling ---------------- */
void systematic_index(const float* probs, int64_t S, float U, int64_t m, int64_t* idx) {
  /* n_j = clamp(floor(cumsum(p)_j * m + U), 0, m); index[t] = #{j : n_j <= t} */
  float s = 0.0f;
  for (int64_t t = 0; t < m; ++t) idx[t] = 0;
  for (int64_t j = 0; j < S; ++j) {
    s += probs[j];
    float nf = floorf(s * (float)m + U);
    if (nf < 0.0f) nf = 0.0f;
    if (nf > (float)m) nf = (float)m;
    const int64_t nj = (int64_t)nf;
    for (int64_t t = nj; t < m; ++t) idx[t] += 1;
  }
}

/* ---------------- Philox4x32-10 ---------------- */
void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* block `blk` of (seed, stream, iteration, row) as the library keys it */
void cs_block(uint64_t seed, uint32_t stream, uint32_t iter, int64_t row, uint32_t blk, uint32_t out[4]) {
  const uint64_t r = (uint64_t)row;
  philox4x32_10(blk, (uint32_t)r, (uint32_t)(r >> 32) ^ (stream << 24), iter, (uint32_t)seed, (uint32_t)(seed >> 32), out);
}
float cs_u01f(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f + 2.98023223876953125e-08f; }
double cs_u01d(uint32_t hi, uint32_t lo) { return (double)((((uint64_t)hi << 32) | lo) >> 11) * 1.1102230246251565e-16; }
void cs_uniform_rows(uint64_t seed, uint32_t stream, uint32_t iter, int64_t row0, int64_t rows, int32_t cols, float* out) {
  uint32_t x[4];
  for (int64_t r = 0; r < rows; ++r)
    for (int32_t e = 0; e < cols; ++e) {
      cs_block(seed, stream, iter, row0 + r, (uint32_t)e >> 2, x);
      out[r * cols + e] = cs_u01f(x[e & 3]);
    }
}
void cs_uniform64_rows(uint64_t seed, uint32_t stream, uint32_t iter, int64_t row0, int64_t rows, int32_t cols, double* out) {
  uint32_t x[4];
  for (int64_t r = 0; r < rows; ++r)
    for (int32_t e = 0; e < cols; ++e) {
      cs_block(seed, stream, iter, row0 + r, (uint32_t)e >> 1, x);
      out[r * cols + e] = (e & 1) ? cs_u01d(x[2], x[3]) : cs_u01d(x[0], x[1]);
    }
}
/* history index of the counter RNG: floor(x * bound / 2^32) */
int64_t cs_hist_draw(uint64_t seed, uint32_t stream, uint32_t iter, int64_t row, int64_t bound) {
  uint32_t x[4];
  cs_block(seed, stream, iter, row, 0, x);
  return (int64_t)(((uint64_t)x[0] * (uint64_t)bound) >> 32);
}
