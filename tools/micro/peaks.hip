// What this GPU sustains, measured: dense bf16 and exact-f32 matrix instruction streams over the whole
// chip, and an HBM stream (read + write).  Prints one JSON object; the repository keeps the result of a
// run on the MI355X box as roofline.json beside the data-sheet peaks bench.py prices against.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/peaks tools/micro/peaks.hip && tools/micro/peaks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } \
  } while (0)

// 4 independent accumulator tiles per wave, WAVES waves per SIMD
template <int KIND>
__global__ void __launch_bounds__(256) mfma_stream(float* out, int iters, float seed) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = seed * (i + r);
  if (KIND == 0) {  // v_mfma_f32_32x32x16_bf16: 32*32*16*2 = 32768 FLOP, 8 passes (32 cycles)
    bf16x8 a, b;
#pragma unroll
    for (int r = 0; r < 8; ++r) { a[r] = (__bf16)(seed + r); b[r] = (__bf16)(seed * r); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  } else {  // v_mfma_f32_32x32x2_f32: 4096 FLOP, 16 passes (64 cycles)
    float a = seed + threadIdx.x, b = seed * 2 + threadIdx.x;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[threadIdx.x] = s;  // keeps the accumulators live
}

__global__ void __launch_bounds__(256) stream_copy(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

__global__ void __launch_bounds__(256) stream_read(const f32x4* __restrict__ src, float* out, size_t n) {
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += src[i];
  if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = s.x;
}

template <typename F>
static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipDeviceSynchronize());
  double best = 1e30;
  for (int r = 0; r < reps; ++r) {
    CHECK(hipEventRecord(e0));
    launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  float* out; CHECK(hipMalloc(&out, 4096));
  printf("{\"device\": \"%s\", \"compute_units\": %d, \"clock_mhz\": %d", p.gcnArchName, cus, p.clockRate / 1000);
  for (int waves = 1; waves <= 2; ++waves) {
    const int blocks = cus * waves, iters = 4000;
    {
      const double ms = time_ms([&] { hipLaunchKernelGGL(mfma_stream<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); }, 5);
      const double flop = (double)blocks * 4 * iters * 32 * 32768.0;
      printf(", \"bf16_mfma_32x32x16_tflops_%dwave\": %.1f", waves, flop / (ms * 1e-3) / 1e12);
    }
    {
      const double ms = time_ms([&] { hipLaunchKernelGGL(mfma_stream<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); }, 5);
      const double flop = (double)blocks * 4 * iters * 32 * 4096.0;
      printf(", \"f32_mfma_32x32x2_tflops_%dwave\": %.1f", waves, flop / (ms * 1e-3) / 1e12);
    }
  }
  const size_t bytes = (size_t)4 << 30, n = bytes / 16;
  f32x4 *a, *b;
  CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes));
  CHECK(hipMemset(a, 1, bytes)); CHECK(hipMemset(b, 0, bytes));
  const double msc = time_ms([&] { hipLaunchKernelGGL(stream_copy, dim3(cus * 16), dim3(256), 0, 0, a, b, n); }, 5);
  const double msr = time_ms([&] { hipLaunchKernelGGL(stream_read, dim3(cus * 16), dim3(256), 0, 0, a, out, n); }, 5);
  printf(", \"hbm_copy_GBps\": %.0f, \"hbm_read_GBps\": %.0f}\n", 2.0 * bytes / (msc * 1e-3) / 1e9, bytes / (msr * 1e-3) / 1e9);
  return 0;
}
