// What the matrix pipes sustain on DATA, not on constant registers: v_mfma_f32_16x16x32_bf16 streams (the
// instruction of the 16-row fused MLP kernel; two waves per SIMD on all CUs, 8 independent accumulator tiles per
// wave, operands in registers, no memory traffic at all) with (a) one constant operand pair and (b) 8 + 8 operand
// registers of random bf16 values, every value with its own mantissa, used in rotation -- the bit patterns the low-order
// terms of a 3-way bf16 split have.  Each mode runs back to back for about 1.5 s and reports the average rate of the
// last second, so that the clock the chip settles at under the load is the one measured.  Prints one JSON object.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_data tools/micro/mfma_data.hip && tools/micro/mfma_data
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } \
  } while (0)

__device__ __forceinline__ unsigned mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// two bf16 values in [1, 2) or (-2, -1] with random sign and mantissa
__device__ __forceinline__ unsigned rand_pair(unsigned h) {
  return (h & 0x807f807fU) | 0x3f803f80U;
}
// the same for fp16 (sign, exponent 15, 10 random mantissa bits)
__device__ __forceinline__ unsigned rand_pair_f16(unsigned h) {
  return (h & 0x83ff83ffU) | 0x3c003c00U;
}

// F16: v_mfma_f32_16x16x32_f16 instead of the bf16 form
template <bool RANDOM, bool F16>
__global__ void __launch_bounds__(256) mfma_stream(float* out, int iters, unsigned seed) {
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 a[8], b[8];
  const unsigned id = (blockIdx.x * 256 + threadIdx.x) * 64u + seed;
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      a[k][r] = RANDOM ? (F16 ? rand_pair_f16(mix(id + 8 * k + r)) : rand_pair(mix(id + 8 * k + r))) : (F16 ? 0x3c003c00U : 0x3f803f80U);
      b[k][r] = RANDOM ? (F16 ? rand_pair_f16(mix(id + 32 + 8 * k + r)) : rand_pair(mix(id + 32 + 8 * k + r))) : (F16 ? 0x3c003c00U : 0x3f803f80U);
    }
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        acc[i] = F16 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[(k + i) & 7]),
                                                              __builtin_bit_cast(f16x8, b[k]), acc[i], 0, 0, 0)
                     : __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[(k + i) & 7]),
                                                               __builtin_bit_cast(bf16x8, b[k]), acc[i], 0, 0, 0);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[threadIdx.x] = s;  // keeps the accumulators live
}

template <bool RANDOM, bool F16>
static double sustained_tflops(float* out, int blocks) {
  const int iters = 2000;  // 2000 * 64 instructions * 16 cycles: about 1 ms per launch
  const double flop = (double)blocks * 4 * iters * 64 * 16384.0;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t0 = now();
  while (now() - t0 < 0.5) {  // settle
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((mfma_stream<RANDOM, F16>), dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
    CHECK(hipDeviceSynchronize());
  }
  int n = 0;
  t0 = now();
  while (now() - t0 < 1.0) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((mfma_stream<RANDOM, F16>), dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
    CHECK(hipDeviceSynchronize());
    n += 20;
  }
  return flop * n / (now() - t0) / 1e12;
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  float* out; CHECK(hipMalloc(&out, 4096));
  const double c = sustained_tflops<false, false>(out, 2 * cus);
  const double r = sustained_tflops<true, false>(out, 2 * cus);
  const double c2 = sustained_tflops<false, false>(out, 2 * cus);
  const double hc = sustained_tflops<false, true>(out, 2 * cus);
  const double hr = sustained_tflops<true, true>(out, 2 * cus);
  printf("{\"device\": \"%s\", \"compute_units\": %d, \"instruction\": \"v_mfma_f32_16x16x32_bf16\", \"waves_per_simd\": 2, "
         "\"constant_operands_tflops\": %.1f, \"random_operands_tflops\": %.1f, \"constant_operands_again_tflops\": %.1f, "
         "\"f16_constant_operands_tflops\": %.1f, \"f16_random_operands_tflops\": %.1f}\n",
         p.gcnArchName, cus, c, r, c2, hc, hr);
  return 0;
}
