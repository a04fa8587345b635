// Order statistics of the rows of a matrix by radix selection: the per-iteration quantiles of the point-wise
// diagnostics over a whole test set (test_epoch_end, irwg/sampling/test_step_vae_sampling.py:824-858:
// np.quantile over the (n k) values of every iteration -- 2.8 million values per iteration on the HEPMASS test set).
//
// A float's bits, with the sign handled, order as the float does.  Three passes over the data fix 11, 11 and 10 bits
// of every wanted order statistic: a pass counts, for each row and each distinct prefix fixed so far, how many of
// the row's values with that prefix have each value of the next digit; a small kernel then walks the counts to the
// digit that holds the wanted rank.  3 reads of the matrix and nothing written but counts: HBM-bound, 12 bytes per
// value, against the 8 passes + 8 writes of a sort.  A value equal to the result is returned (-0.0 and 0.0 are
// told apart by the order, equal as floats); NaNs order above +inf (negative-sign NaNs below -inf): callers that
// must give NaN for a row holding one test for it separately.
#include "cs_common.h"

namespace {

constexpr int SEL_MAXQ = 16;     // order statistics per row
constexpr int SEL_BINS = 2048;   // 11 bits per pass (the last pass: 10)
constexpr int SEL_GROUP = 4;     // histograms a workgroup keeps in LDS at a time

__device__ __forceinline__ uint32_t sel_key(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float sel_value(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

struct SelState {          // per row
  uint32_t prefix[SEL_MAXQ];   // bits fixed so far of target j (high bits)
  int64_t rank[SEL_MAXQ];      // rank of target j among the values sharing its prefix
  int32_t slot[SEL_MAXQ];      // index of the first target with the same prefix (the histogram they share)
  int32_t nslots;              // distinct prefixes: targets j with slot[j] == j
};

// shift / width of the digit of a pass
__device__ __forceinline__ int sel_shift(int pass) { return pass == 0 ? 21 : pass == 1 ? 10 : 0; }
__device__ __forceinline__ int sel_bits(int pass) { return pass == 2 ? 10 : 11; }

__global__ void __launch_bounds__(256) cs_select_init_kernel(SelState* st, const int64_t* ranks, int nq, int64_t rows) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  SelState& s = st[row];
  for (int j = 0; j < nq; ++j) { s.prefix[j] = 0; s.rank[j] = ranks[j]; s.slot[j] = 0; }
  s.nslots = 1;
}

// counts[row][slot j][digit] += ...   for the slice of the row this block owns
__global__ void __launch_bounds__(256) cs_select_count_kernel(const float* x, int64_t m, const SelState* st, int nq, int pass,
                                                              uint32_t* counts) {
  extern __shared__ uint32_t lh[];   // [SEL_GROUP][SEL_BINS]
  __shared__ uint32_t s_pf[SEL_MAXQ];
  __shared__ int s_own[SEL_MAXQ];
  __shared__ int s_ns;
  const int64_t row = blockIdx.y;
  const SelState& s = st[row];
  const int shift = sel_shift(pass), nb = 1 << sel_bits(pass);
  if (threadIdx.x == 0) {   // the distinct prefixes of this row
    int ns = 0;
    for (int j = 0; j < nq; ++j)
      if (s.slot[j] == j) { s_pf[ns] = s.prefix[j]; s_own[ns] = j; ++ns; }
    s_ns = ns;
  }
  __syncthreads();
  const int ns = s_ns;
  const int64_t per = (m + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < m ? lo + per : m;
  const float* xr = x + row * m;
  const int pshift = shift + sel_bits(pass);   // bits below the prefix (pass 0: no prefix)
  uint32_t* cr = counts + row * (int64_t)nq * SEL_BINS;
  // SEL_GROUP histograms at a time (32 KB of LDS: several workgroups per CU to hide the loads behind); a row with more
  // distinct prefixes reads its slice once per group
  for (int g0 = 0; g0 < ns; g0 += SEL_GROUP) {
    const int gn = ns - g0 < SEL_GROUP ? ns - g0 : SEL_GROUP;
    for (int i = threadIdx.x; i < gn * SEL_BINS; i += 256) lh[i] = 0;
    __syncthreads();
    auto count = [&](float v) {
      const uint32_t k = sel_key(v);
      const uint32_t kp = pass == 0 ? 0u : (k >> pshift);
      const uint32_t dg = (k >> shift) & (uint32_t)(nb - 1);
      for (int t = 0; t < gn; ++t)
        if (kp == s_pf[g0 + t]) { atomicAdd(&lh[t * SEL_BINS + dg], 1u); break; }
    };
    int64_t i = lo + threadIdx.x;
    for (; i + 768 < hi; i += 1024) {   // four loads in flight per thread
      const float v0 = xr[i], v1 = xr[i + 256], v2 = xr[i + 512], v3 = xr[i + 768];
      count(v0); count(v1); count(v2); count(v3);
    }
    for (; i < hi; i += 256) count(xr[i]);
    __syncthreads();
    for (int t = threadIdx.x; t < gn * SEL_BINS; t += 256) {
      const uint32_t c = lh[t];
      if (c) atomicAdd(&cr[s_own[g0 + t / SEL_BINS] * SEL_BINS + (t % SEL_BINS)], c);
    }
    __syncthreads();
  }
}

// one wavefront per row: walk each target's counts to its digit, fix it, regroup the targets by prefix, clear the counts
__global__ void __launch_bounds__(64) cs_select_step_kernel(SelState* st, int nq, int pass, uint32_t* counts, float* out) {
  const int64_t row = blockIdx.x;
  const int lane = threadIdx.x;
  SelState& s = st[row];
  const int nb = 1 << sel_bits(pass), bits = sel_bits(pass);
  uint32_t* cr = counts + row * (int64_t)nq * SEL_BINS;
  for (int j = 0; j < nq; ++j) {
    const uint32_t* h = cr + s.slot[j] * SEL_BINS;
    // lanes take consecutive runs of nb / 64 digits; exclusive prefix over the lanes, then a walk inside one run
    const int run = nb / 64;
    uint64_t mine = 0;
    for (int d = 0; d < run; ++d) mine += h[lane * run + d];
    uint64_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
      const uint64_t v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    const uint64_t before = incl - mine;
    const uint64_t r = (uint64_t)s.rank[j];
    const bool has = r >= before && r < incl;
    const unsigned long long who = __ballot(has);
    const int src = who ? (int)__ffsll((long long)who) - 1 : 63;   // (a rank beyond the row: the last run)
    uint32_t digit = 0;
    uint64_t rem = 0;
    if (lane == src) {
      uint64_t acc = before;
      int d = 0;
      for (; d < run - 1; ++d) {
        const uint64_t c = h[lane * run + d];
        if (r < acc + c) break;
        acc += c;
      }
      digit = (uint32_t)(lane * run + d);
      rem = r - acc;
    }
    digit = __shfl(digit, src);
    rem = __shfl(rem, src);
    if (lane == 0) {
      s.prefix[j] = (s.prefix[j] << bits) | digit;
      s.rank[j] = (int64_t)rem;
    }
  }
  __syncthreads();
  if (lane == 0) {
    int ns = 0;
    for (int j = 0; j < nq; ++j) {
      int f = j;
      for (int i = 0; i < j; ++i)
        if (s.prefix[i] == s.prefix[j]) { f = i; break; }
      s.slot[j] = f;
      if (f == j) ++ns;
    }
    s.nslots = ns;
    if (pass == 2)
      for (int j = 0; j < nq; ++j) out[row * nq + j] = sel_value(s.prefix[j]);
  }
  for (int i = lane; i < nq * SEL_BINS; i += 64) cr[i] = 0;
}

}  // namespace

// out[row, j] = the ranks[j]-th smallest (0-based) of x[row, 0..m), for nq <= 16 ranks (host array) common to all rows.
// workspace: cs_row_order_statistics_workspace(rows, nq) bytes of device memory.
extern "C" int64_t cs_row_order_statistics_workspace(int64_t rows, int32_t nq) {
  if (rows < 0 || nq < 1 || nq > SEL_MAXQ) return -1;
  return rows * ((int64_t)sizeof(SelState) + (int64_t)nq * SEL_BINS * 4) + SEL_MAXQ * 8 + 256;
}

extern "C" int cs_row_order_statistics(const float* x, int64_t rows, int64_t m, const int64_t* ranks, int32_t nq, float* out,
                                       void* workspace, int64_t workspace_bytes, void* stream) {
  CS_REQUIRE(x && ranks && out && workspace, "null argument");
  CS_REQUIRE(rows >= 0 && m >= 1 && nq >= 1 && nq <= SEL_MAXQ, "bad sizes (1 .. 16 ranks per row)");
  CS_REQUIRE(rows <= 65535, "at most 65535 rows per call");
  CS_REQUIRE(workspace_bytes >= cs_row_order_statistics_workspace(rows, nq), "workspace too small");
  for (int j = 0; j < nq; ++j) CS_REQUIRE(ranks[j] >= 0 && ranks[j] < m, "rank outside the row");
  if (rows == 0) return CS_OK;
  hipStream_t st = (hipStream_t)stream;
  char* w = (char*)workspace;
  int64_t* d_ranks = (int64_t*)w;
  w += SEL_MAXQ * 8;
  w = (char*)(((uintptr_t)w + 255) & ~(uintptr_t)255);
  SelState* state = (SelState*)w;
  uint32_t* counts = (uint32_t*)(w + rows * (int64_t)sizeof(SelState));
  CS_HIP(hipMemcpyAsync(d_ranks, ranks, (size_t)nq * 8, hipMemcpyHostToDevice, st));
  CS_HIP(hipMemsetAsync(counts, 0, (size_t)rows * nq * SEL_BINS * 4, st));
  hipLaunchKernelGGL(cs_select_init_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, state, d_ranks, nq, rows);
  // blocks per row: enough to fill the chip, each with a slice long enough to pay for its histogram
  int64_t bpr = (m + (1 << 16) - 1) >> 16;
  const int64_t want = (2048 + rows - 1) / rows;
  if (bpr > want && want >= 1) bpr = bpr < 4 * want ? bpr : 4 * want;
  if (bpr < 1) bpr = 1;
  if (bpr > 65535) bpr = 65535;
  const size_t lds = (size_t)SEL_GROUP * SEL_BINS * 4;
  for (int pass = 0; pass < 3; ++pass) {
    hipLaunchKernelGGL(cs_select_count_kernel, dim3((unsigned)bpr, (unsigned)rows), dim3(256), lds, st, x, m, state, nq, pass,
                       counts);
    hipLaunchKernelGGL(cs_select_step_kernel, dim3((unsigned)rows), dim3(64), 0, st, state, nq, pass, counts, out);
  }
  CS_HIP(hipGetLastError());
  return CS_OK;
}
