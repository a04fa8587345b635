// Micro-benchmark: the inner loop of a hidden layer (256 -> 256) computed as 6 bf16 matrix
// products on 3-way bf16 splits of both f32 operands (a = a0 + a1 + a2 exactly, products
// a_i * b_j for i + j <= 2, f32 accumulation), the f32-equivalent alternative to the exact-f32
// matrix instruction at 6/16 of its cost.
//
// One workgroup = 4 waves (one per SIMD), each wave owns 32 batch rows and all 256 output
// features: 8 accumulator tiles of v_mfma_f32_32x32x16_bf16.  Per k-step (16 input features):
// 48 matrix instructions, 24 ds_read_b128 of pre-split weights, the 3-way split of 8 activation
// values per lane.  Variants:
//   0  matrix instructions only
//   1  + weight fragments from LDS
//   2  + activation split in the vector ALU
//   3  + the weights streamed global -> LDS by LDS-DMA (2 k-steps per ring slot, one barrier per slot)
//   4  as 3, two output tiles at a time so that consecutive matrix instructions use different accumulators
//   5  as 3 on v_mfma_f32_16x16x32_bf16 (16 output tiles of 16 features x two 16-row halves per 32-wide k-step)
//   6  as 3 with the weights staged through registers instead of LDS-DMA: global_load_dwordx4 four at a time,
//      ds_write_b128 two output tiles later, spread over the slot's 16 tile groups
//   7  as 3 with the LDS-DMA copies spread over the slot's tile groups (one per group) instead of issued together
// Prints f32-equivalent TFLOP/s (2*rows*in*out per layer) for the whole chip.
//   hipcc -O3 --offload-arch=gfx950 -o split_rate split_rate.hip && ./split_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ float lo_f(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_f(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// 8 f32 values -> three bf16x8 fragments
__device__ __forceinline__ void split8(const float* v, u32x4& s0, u32x4& s1, u32x4& s2) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float x = v[2 * p], y = v[2 * p + 1];
    const unsigned p0 = cvt_pk(x, y);
    const float rx = x - lo_f(p0), ry = y - hi_f(p0);
    const unsigned p1 = cvt_pk(rx, ry);
    const float sx = rx - lo_f(p1), sy = ry - hi_f(p1);
    const unsigned p2 = cvt_pk(sx, sy);
    s0[p] = p0; s1[p] = p1; s2[p] = p2;
  }
}
__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

#define MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a), as_bf(b), c, 0, 0, 0)
#define MFMA16(a, b, c) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(a), as_bf(b), c, 0, 0, 0)

constexpr int SLOT_BYTES = 2 * 8 * 3 * 1024;  // 2 k-steps x 8 tiles x 3 splits x 1 KB

template <int V>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k(float* out, int layers, const u32x4* gsrc, unsigned gmask) {
  extern __shared__ u32x4 lds[];  // 2 slots
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2 * SLOT_BYTES / 16; i += 256) lds[i] = u32x4{0x3f803f80u, 0x3c003c00u, (unsigned)i, 0x3f803f80u};
  __syncthreads();
  f32x16 act[8], acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { act[t][r] = 0.001f * (lane + r + t); acc[t][r] = 0.0f; }
  u32x4 b0 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b1 = b0, b2 = b0;
  u32x4 a0 = b0, a1 = b0, a2 = b0, pb0 = b0, pb1 = b0, pb2 = b0;
  f32x4 acc4[16][2];
#pragma unroll
  for (int t = 0; t < 16; ++t) { acc4[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc4[t][1] = acc4[t][0]; }
  size_t gpos = (size_t)blockIdx.x * 977;
  const u32x4* stg_src = gsrc;
  u32x4 stg[4];
  for (int L = 0; L < layers; ++L) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int slot = (ks >> 1) & 1;
      if ((V == 6 || V == 7) && (ks & 1) == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stg_src = gsrc + ((gpos * (SLOT_BYTES / 16)) & gmask) + threadIdx.x;
        ++gpos;
      } else if (V >= 3 && V <= 5 && (ks & 1) == 0) {
        // the slot that the previous pair of k-steps used is free after this barrier; fill it for the
        // pair after this one
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const u32x4* src = gsrc + ((gpos * (SLOT_BYTES / 16)) & gmask);
        ++gpos;
#pragma unroll
        for (int c = 0; c < SLOT_BYTES / 4096; ++c)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c * 256 + threadIdx.x),
                                           (__attribute__((address_space(3))) void*)(lds + (slot ^ 1) * (SLOT_BYTES / 16) + c * 256 + wave * 64),
                                           16, 0, 0);
      }
      if (V >= 2) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = act[ks >> 1][8 * (ks & 1) + j];
        split8(v, b0, b1, b2);
      }
      const u32x4* wl = lds + slot * (SLOT_BYTES / 16) + (ks & 1) * (8 * 3 * 64) + lane;
      if (V == 5) {
        // the same work on v_mfma_f32_16x16x32_bf16: this k-step pair = one 32-wide k-step, 16 output tiles
        // of 16 features, two 16-row halves of the wave's 32 rows (same LDS bytes, same count x half the cycles)
        if (ks & 1) {
          const u32x4* wl2 = lds + slot * (SLOT_BYTES / 16) + lane;
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            const u32x4 c0 = wl2[(t * 3 + 0) * 64], c1 = wl2[(t * 3 + 1) * 64], c2 = wl2[(t * 3 + 2) * 64];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              const u32x4 e0 = hh ? b0 : pb0, e1 = hh ? b1 : pb1, e2 = hh ? b2 : pb2;
              MFMA16(c0, e0, acc4[t][hh]);
              MFMA16(c0, e1, acc4[t][hh]);
              MFMA16(c1, e0, acc4[t][hh]);
              MFMA16(c0, e2, acc4[t][hh]);
              MFMA16(c1, e1, acc4[t][hh]);
              MFMA16(c2, e0, acc4[t][hh]);
            }
          }
        } else {
          pb0 = b0; pb1 = b1; pb2 = b2;
        }
      } else if (V == 4) {  // two tiles at a time, consecutive matrix instructions on different accumulators
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
          const u32x4 c0 = wl[(t * 3 + 0) * 64], c1 = wl[(t * 3 + 1) * 64], c2 = wl[(t * 3 + 2) * 64];
          const u32x4 d0 = wl[(t * 3 + 3) * 64], d1 = wl[(t * 3 + 4) * 64], d2 = wl[(t * 3 + 5) * 64];
          MFMA(c0, b0, acc[t]); MFMA(d0, b0, acc[t + 1]);
          MFMA(c0, b1, acc[t]); MFMA(d0, b1, acc[t + 1]);
          MFMA(c1, b0, acc[t]); MFMA(d1, b0, acc[t + 1]);
          MFMA(c0, b2, acc[t]); MFMA(d0, b2, acc[t + 1]);
          MFMA(c1, b1, acc[t]); MFMA(d1, b1, acc[t + 1]);
          MFMA(c2, b0, acc[t]); MFMA(d2, b0, acc[t + 1]);
        }
      } else {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (V >= 1) { a0 = wl[(t * 3 + 0) * 64]; a1 = wl[(t * 3 + 1) * 64]; a2 = wl[(t * 3 + 2) * 64]; }
        if (V == 6) {
          // group G = 8*(ks&1) + t of the slot: batches of 4 copies loaded at G = 0, 4, 8 and written at G = 3, 7, 11
          const int G = 8 * (ks & 1) + t;
          u32x4* dstb = lds + (slot ^ 1) * (SLOT_BYTES / 16) + threadIdx.x;
          if (G == 0 || G == 4 || G == 8) {
            const int c0 = G;
#pragma unroll
            for (int c = 0; c < 4; ++c) stg[c] = __builtin_nontemporal_load(stg_src + (c0 + c) * 256);
          }
          if (G == 3 || G == 7 || G == 11) {
            const int c0 = G - 3;
#pragma unroll
            for (int c = 0; c < 4; ++c) dstb[(c0 + c) * 256] = stg[c];
          }
        }
        if (V == 7) {
          const int G = 8 * (ks & 1) + t;
          if (G < 12)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(stg_src + G * 256),
                                             (__attribute__((address_space(3))) void*)(lds + (slot ^ 1) * (SLOT_BYTES / 16) + G * 256 + wave * 64),
                                             16, 0, 0);
        }
        MFMA(a0, b0, acc[t]);
        MFMA(a0, b1, acc[t]);
        MFMA(a1, b0, acc[t]);
        MFMA(a0, b2, acc[t]);
        MFMA(a1, b1, acc[t]);
        MFMA(a2, b0, acc[t]);
      }
      }
    }
    // next layer: the accumulators become the activations (bias / ReLU would go here)
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = V == 5 ? acc4[2 * t + (r >> 3)][(r >> 2) & 1][r & 3] : acc[t][r];
        act[t][r] = fmaxf(v, 0.0f) * 1e-3f;
        acc[t][r] = 0.25f;
        if (V == 5) acc4[2 * t + (r >> 3)][(r >> 2) & 1][r & 3] = 0.25f;
      }
  }
  float s = 0;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += act[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V>
void run(int blocks, int layers, size_t gbytes) {
  float* out; u32x4* gsrc;
  hipMalloc(&gsrc, gbytes + SLOT_BYTES);
  {  // pseudo-random bf16 bit patterns of moderate exponent: the clock the chip holds depends on the data
    const size_t nw = (gbytes + SLOT_BYTES) / 4;
    unsigned* h = (unsigned*)malloc(nw * 4);
    unsigned x = 12345u;
    for (size_t i = 0; i < nw; ++i) {
      x = x * 1664525u + 1013904223u;
      h[i] = 0x3c003c00u | (x & 0x807f807fu) | ((x >> 9) & 0x03000300u);
    }
    hipMemcpy(gsrc, h, nw * 4, hipMemcpyHostToDevice);
    free(h);
  }
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT_BYTES);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const unsigned gmask = (unsigned)(gbytes / 16 - 1);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 2 * SLOT_BYTES, 0, out, 2, gsrc, gmask);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 2 * SLOT_BYTES, 0, out, layers, gsrc, gmask);
  hipEventRecord(e1); hipDeviceSynchronize();
  if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); exit(1); }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 128 * 256 * 256 * 2 * layers;  // f32-equivalent
  printf("variant %d blocks %d layers %d weights-window %zu MB: %.3f ms, %.1f f32-equivalent TFLOP/s, %.2f TB/s streamed\n", V, blocks,
         layers, gbytes >> 20, ms, flops / (ms * 1e-3) / 1e12, V >= 3 ? (double)blocks * layers * 8 * SLOT_BYTES / (ms * 1e-3) / 1e12 : 0.0);
  hipFree(out); hipFree(gsrc);
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 2048, layers = argc > 2 ? atoi(argv[2]) : 24;
  run<0>(blocks, layers, 4 << 20);
  run<1>(blocks, layers, 4 << 20);
  run<2>(blocks, layers, 4 << 20);
  run<3>(blocks, layers, 2 << 20);
  run<3>(blocks, layers, 4 << 20);
  run<3>(blocks, layers, 32 << 20);
  run<4>(blocks, layers, 4 << 20);
  run<3>(blocks, layers, 4 << 20);
  run<5>(blocks, layers, 4 << 20);
  run<3>(blocks, layers, 4 << 20);
  run<5>(blocks, layers, 4 << 20);
  run<6>(blocks, layers, 4 << 20);
  run<7>(blocks, layers, 4 << 20);
  run<3>(blocks, layers, 4 << 20);
  run<6>(blocks, layers, 4 << 20);
  run<7>(blocks, layers, 4 << 20);
  return 0;
}
