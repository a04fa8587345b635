// Micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 under the operand patterns of the fused
// MLP kernel.  One wave per SIMD (256 threads per block), `blocks` blocks.
#include <cstring>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VARIANT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k(float* out, long long* ticks, int iters, float seed, const float4* gsrc) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  __shared__ f32x4 lds[16 * 64];
  __shared__ f32x4 lds2[8 * 256];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16 * 64; i += 256) lds[i] = f32x4{seed, seed * i, 1.f, 2.f};
  __syncthreads();
  f32x16 acc[8];
  f32x16 bsrc[2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = seed * (i + r);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) bsrc[i][r] = seed * (3 * i + r);
  float a0 = seed + threadIdx.x, a1 = seed * 2 + threadIdx.x;
  if (VARIANT == 8 || VARIANT >= 11) {
    acc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, acc[7], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) { float x = acc[7][r]; asm volatile("" : "+a"(x)); acc[7][r] = x; }
  }
  f32x4 stg = f32x4{0.f, 0.f, 0.f, 0.f};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (VARIANT == 0) {  // 4 independent accumulators, VGPR operands
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a0, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a0, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a1, acc[3], 0, 0, 0);
      }
    } else if (VARIANT == 1) {  // 2 alternating accumulators, VGPR operands
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a0, acc[1], 0, 0, 0);
      }
    } else if (VARIANT == 2) {  // 2 alternating accumulators, B operand = element of another accumulator tile
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bsrc[0][u & 15], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bsrc[0][u & 15], acc[1], 0, 0, 0);
      }
    } else if (VARIANT == 3) {  // one accumulator, back-to-back dependent
#pragma unroll
      for (int u = 0; u < 64; ++u) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, acc[0], 0, 0, 0);
    } else if (VARIANT == 5) {  // B operand pinned in the accumulator file (AGPR)
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        float b = bsrc[0][u & 15];
        asm volatile("" : "+a"(b));
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1], 0, 0, 0);
      }
    } else if (VARIANT == 6) {  // A operands streamed from LDS (ds_read_b128 per 8 MFMAs), B in VGPR
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const f32x4 w0 = lds[(2 * g) * 64 + (threadIdx.x & 63)], w1 = lds[(2 * g + 1) * 64 + (threadIdx.x & 63)];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.x, a0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.x, a0, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.y, a1, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.y, a1, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.z, a0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.z, a0, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.w, a1, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.w, a1, acc[1], 0, 0, 0);
      }
    } else if (VARIANT == 7) {  // both: A from LDS, B in AGPR, 8 different accumulators like the MLP kernel
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const f32x4 w0 = lds[(2 * g) * 64 + (threadIdx.x & 63)], w1 = lds[(2 * g + 1) * 64 + (threadIdx.x & 63)];
        float b0 = bsrc[0][(4 * (g >> 2) + 0) & 15], b1 = bsrc[0][(4 * (g >> 2) + 1) & 15], b2 = bsrc[0][(4 * (g >> 2) + 2) & 15], b3 = bsrc[0][(4 * (g >> 2) + 3) & 15];
        asm volatile("" : "+a"(b0), "+a"(b1), "+a"(b2), "+a"(b3));
        const int nt = (2 * g) & 7;
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.x, b0, acc[nt], 0, 0, 0);
        acc[nt + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.x, b0, acc[nt + 1], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.y, b1, acc[nt], 0, 0, 0);
        acc[nt + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.y, b1, acc[nt + 1], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.z, b2, acc[nt], 0, 0, 0);
        acc[nt + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.z, b2, acc[nt + 1], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.w, b3, acc[nt], 0, 0, 0);
        acc[nt + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.w, b3, acc[nt + 1], 0, 0, 0);
      }
    } else if (VARIANT == 8) {  // B operand read straight from another accumulator tile (AGPR), no copies
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, acc[7][u & 15], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, acc[7][u & 15], acc[1], 0, 0, 0);
      }
    } else if (VARIANT == 9 || VARIANT == 10) {  // MFMA stream with one 1-KB global->LDS copy per 16 (9) / 4 (10) MFMAs
      constexpr int PER = VARIANT == 9 ? 16 : 4;
#pragma unroll
      for (int u = 0; u < 64; ++u) {
        acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, acc[u & 3], 0, 0, 0);
        if (u % PER == 1) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + ((it * 64 + u) & 1023) * 256 + threadIdx.x),
                                           (__attribute__((address_space(3))) void*)(lds + (u & 15) * 64), 16, 0, 0);
        }
      }
    } else if (VARIANT >= 11 && VARIANT <= 15) {
      // the MLP kernel's stage loop: 16 groups of 8 MFMAs, weight fragments double-buffered from LDS,
      // B operand = register of another accumulator tile; 12: + one LDS-DMA copy per 2 groups;
      // 13: + s_waitcnt vmcnt(0) and a workgroup barrier per stage; 14: barrier only
      if (VARIANT == 13) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      if (VARIANT >= 13) __syncthreads();
      f32x4 w[2][2];
      w[0][0] = lds[0 * 64 + lane]; w[0][1] = lds[1 * 64 + lane];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int c = g & 1, n = c ^ 1, nt = (2 * g) & 3, q = g >> 1;
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][0].x, acc[7][4 * q + 0], acc[nt], 0, 0, 0);
        acc[nt + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][1].x, acc[7][4 * q + 0], acc[nt + 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        w[n][0] = lds[((2 * (g + 1)) & 15) * 64 + lane];
        w[n][1] = lds[((2 * (g + 1) + 1) & 15) * 64 + lane];
        if (VARIANT == 15 && (g & 1)) {  // register-staged copy: load now, write the previous one
          lds2[((g - 1) & 7) * 256 + threadIdx.x] = stg;
          stg = ((const f32x4*)gsrc)[((it * 8 + g) & 1023) * 256 + threadIdx.x];
        }
        if ((VARIANT == 12 || VARIANT == 13) && (g & 1))
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + ((it * 8 + g) & 1023) * 256 + threadIdx.x),
                                           (__attribute__((address_space(3))) void*)(lds2 + g * 256 + (threadIdx.x & ~63)), 16, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][0].y, acc[7][4 * q + 1], acc[nt], 0, 0, 0);
        acc[nt + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][1].y, acc[7][4 * q + 1], acc[nt + 1], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][0].z, acc[7][4 * q + 2], acc[nt], 0, 0, 0);
        acc[nt + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][1].z, acc[7][4 * q + 2], acc[nt + 1], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][0].w, acc[7][4 * q + 3], acc[nt], 0, 0, 0);
        acc[nt + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][1].w, acc[7][4 * q + 3], acc[nt + 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (VARIANT == 16) {
      // no LDS: weight fragments loaded straight from global memory (L2-resident stream), 4 groups ahead
      const f32x4* gs = (const f32x4*)gsrc;
      f32x4 w[5][2];
#pragma unroll
      for (int p = 0; p < 4; ++p) { w[p][0] = gs[(2 * p) * 64 + lane]; w[p][1] = gs[(2 * p + 1) * 64 + lane]; }
#pragma unroll
      for (int g = 0; g < 10; ++g) {
        const int c = g % 5, n = (g + 4) % 5, nt = (2 * g) & 3, q = (g >> 1) & 3;
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][0].x, acc[7][4 * q + 0], acc[nt], 0, 0, 0);
        acc[nt + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][1].x, acc[7][4 * q + 0], acc[nt + 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        w[n][0] = gs[(((it * 10 + g + 4) * 2) & 2047) * 64 + lane];
        w[n][1] = gs[(((it * 10 + g + 4) * 2 + 1) & 2047) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][0].y, acc[7][4 * q + 1], acc[nt], 0, 0, 0);
        acc[nt + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][1].y, acc[7][4 * q + 1], acc[nt + 1], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][0].z, acc[7][4 * q + 2], acc[nt], 0, 0, 0);
        acc[nt + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][1].z, acc[7][4 * q + 2], acc[nt + 1], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][0].w, acc[7][4 * q + 3], acc[nt], 0, 0, 0);
        acc[nt + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][1].w, acc[7][4 * q + 3], acc[nt + 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (VARIANT == 4) {  // 8 accumulators round robin
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, acc[i], 0, 0, 0);
    }
  }
  long long t1 = clock64();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s + bsrc[1][3] + lds2[threadIdx.x].x;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int V>
void run(int blocks, int iters) {
  float* out; long long* ticks; float4* gsrc;
  hipMalloc(&gsrc, 1024 * 256 * 16 + 4096); hipMemset(gsrc, 0, 1024 * 256 * 16);
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&ticks, blocks * 4 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, ticks, 10, 1.0f, gsrc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, ticks, iters, 1.0f, gsrc);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[4]; hipMemcpy(h, ticks, 32, hipMemcpyDeviceToHost);
  double n = (V == 16 ? 80.0 : 64.0) * iters;
  printf("variant %d blocks %d: %.1f ticks/MFMA, %.2f ns/MFMA (wall), %.1f TFLOP/s chip-equivalent\n", V, blocks,
         h[0] / n, ms * 1e6 / n, blocks * 4 * n * 4096 / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(ticks); hipFree(gsrc);
}

// Sustained clock: a pure MFMA stream on all CUs for about a second; prints the device tick rate
// (ticks per wall ns) and the delivered rate of each ~35 ms launch.
void sustain() {
  const int blocks = 256, iters = 20000, reps = 30;
  float* out; long long* ticks; float4* gsrc;
  hipMalloc(&gsrc, 1024 * 256 * 16 + 4096); hipMemset(gsrc, 0, 1024 * 256 * 16);
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&ticks, blocks * 4 * 8);
  hipEvent_t e[reps + 1];
  for (auto& x : e) hipEventCreate(&x);
  hipEventRecord(e[0]);
  for (int r = 0; r < reps; ++r) {
    hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, ticks + (size_t)0, iters, 1.0f, gsrc);
    hipEventRecord(e[r + 1]);
  }
  hipDeviceSynchronize();
  long long h[4]; hipMemcpy(h, ticks, 32, hipMemcpyDeviceToHost);
  for (int r = 0; r < reps; ++r) {
    float ms; hipEventElapsedTime(&ms, e[r], e[r + 1]);
    const double n = 64.0 * iters;
    printf("sustain rep %2d: %.3f ms, %.1f TFLOP/s%s\n", r, ms, blocks * 4 * n * 4096 / (ms * 1e-3) / 1e12,
           r == reps - 1 ? "" : "");
    if (r == reps - 1) printf("last launch: %.1f ticks/MFMA, tick rate %.3f GHz\n", h[0] / n, h[0] / (ms * 1e6));
  }
}

int main(int argc, char** argv) {
  int iters = 2000;
  if (argc > 1 && !strcmp(argv[1], "sustain")) { sustain(); return 0; }
  for (int blocks : {1, 256}) {
    run<0>(blocks, iters); run<1>(blocks, iters); run<2>(blocks, iters); run<3>(blocks, iters); run<4>(blocks, iters); run<5>(blocks, iters); run<6>(blocks, iters); run<7>(blocks, iters); run<8>(blocks, iters); run<9>(blocks, iters); run<10>(blocks, iters); run<11>(blocks, iters); run<12>(blocks, iters); run<13>(blocks, iters); run<14>(blocks, iters); run<15>(blocks, iters); run<16>(blocks, iters * 64 / 80);
  }
  return 0;
}
