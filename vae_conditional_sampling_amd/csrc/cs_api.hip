// Error reporting, version and RNG dump entry points of libcsvae.so.
#include "cs_common.h"

static thread_local std::string g_cs_error;

void cs_set_error(const std::string& msg) { g_cs_error = msg; }
int cs_fail(int code, const std::string& msg) {
  g_cs_error = msg;
  return code;
}

extern "C" int cs_version(void) { return 1; }
extern "C" const char* cs_last_error(void) { return g_cs_error.c_str(); }

CsPhilox cs_make_philox(const cs_rng* r) {
  CsPhilox p;
  p.seed_lo = (uint32_t)(r ? r->seed : 0);
  p.seed_hi = (uint32_t)((r ? r->seed : 0) >> 32);
  p.iter = r ? r->iteration : 0;
  p.row_offset = r ? r->example_offset : 0;
  return p;
}

// kind 0: normal f32, 1: uniform f32, 2: uniform f64
template <int KIND, typename T>
__global__ void cs_rng_dump_kernel(CsPhilox p, uint32_t stream_id, int64_t rows, int cols, T* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int64_t r0 = i / cols;
  const uint32_t e = (uint32_t)(i - r0 * cols);
  const int64_t r = p.row_offset + r0;
  if (KIND == 0) out[i] = (T)cs_normal_elem(p, stream_id, r, e);
  else if (KIND == 1) out[i] = (T)cs_uniform_elem(p, stream_id, r, e);
  else out[i] = (T)cs_uniform64_elem(p, stream_id, r, e);
}

template <int KIND, typename T>
static int rng_dump(const cs_rng* rng, uint32_t stream_id, int64_t rows, int32_t cols, T* out, void* stream) {
  CS_REQUIRE(out && rows >= 0 && cols >= 0, "bad argument");
  const int64_t n = rows * cols;
  if (n == 0) return CS_OK;
  CsPhilox p = cs_make_philox(rng);
  hipLaunchKernelGGL((cs_rng_dump_kernel<KIND, T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, p, stream_id, rows, cols, out);
  CS_HIP(hipGetLastError());
  return CS_OK;
}

extern "C" int cs_rng_normal(const cs_rng* rng, uint32_t stream_id, int64_t rows, int32_t cols, float* out, void* stream) {
  return rng_dump<0, float>(rng, stream_id, rows, cols, out, stream);
}
extern "C" int cs_rng_uniform(const cs_rng* rng, uint32_t stream_id, int64_t rows, int32_t cols, float* out, void* stream) {
  return rng_dump<1, float>(rng, stream_id, rows, cols, out, stream);
}
extern "C" int cs_rng_uniform64(const cs_rng* rng, uint32_t stream_id, int64_t rows, int32_t cols, double* out, void* stream) {
  return rng_dump<2, double>(rng, stream_id, rows, cols, out, stream);
}
