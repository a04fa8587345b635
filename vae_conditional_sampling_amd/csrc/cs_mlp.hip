// Fused whole-network residual-MLP forward for gfx950 (MI355X), exact-f32 MFMA.
//
// Replaces ResidualFCNetwork.forward (irwg/models/neural_nets.py:163-169, blocks :87-103):
//   h = relu(W0 x + b0);  per block  h = relu(h + W2 relu(W1 h + b1) + b2);  y = Wf h + bf
// for the encoder (model.var_latent_network) and decoder (model.generator_network) of the
// MLP VAEs, with the distribution parameterisation of VAE.create_distribution
// (irwg/models/vae.py:203-256) and the masked log-likelihood of weights.py:46-47 fused into
// the epilogue, and the proposal sampling Z = loc + scale*eps (vae_imputation.py:596-597)
// fused into the decoder prologue.
//
// Design (one wavefront = 32 rows, the whole network in registers):
//   * v_mfma_f32_32x32x2_f32 computes Y^T = W * X^T: the A operand is a weight fragment
//     (lane l: W[n = l&31][k = l>>5]), the B operand an activation fragment
//     (lane l: act[k = l>>5][row = l&31]), and the 32x32 result tile has the batch row on the
//     lane and 16 output features in the lane's registers
//     (feature = (r&3) + 8*(r>>2) + 4*(l>>5)).  Register r of an output tile is therefore
//     directly the B operand of k-step r of the next layer, provided the weight fragment of
//     that k-step holds column k = 32*kt + (r&3) + 8*(r>>2) + 4*(l>>5): activations never
//     leave the register file between layers (no LDS, no HBM).
//   * the host packs every layer's weights into the order the lanes consume them
//     ("operand stream": for each k-tile of 32, for q in 0..3, for each n-tile, for each
//     lane: 4 consecutive k = 32*kt + 8*q + 4*(l>>5) + 0..3), so one ds_read_b128 feeds 4 MFMAs
//     and the LDS image is lane-linear (conflict-free, fillable by global_load_lds).
//   * a workgroup (1..4 wavefronts, one per SIMD) streams the operand stream through a
//     two-stage LDS ring with global_load_lds_dwordx4; a stage is one k-tile of one layer
//     (32 KB for hidden width 256), i.e. 128 MFMAs = 8192 cycles per wavefront.
//   * accumulators start from the bias (residual layer: from h + b2).
#include "cs_mlp.h"

#include <cstdlib>
#include <cstring>
#include <vector>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
typedef __attribute__((address_space(3))) float lds_float;

// ----------------------------------------------------------------------------------------
// One stage = one k-tile (32 input features) of one layer: 4*NTN weight fragments of 1 KB in the
// LDS slot `sb`, 16*NTN MFMAs per wavefront.
//   * fragments are read two at a time (8 MFMAs' worth) into one of two register sets, behind the
//     first two MFMAs of the preceding group, so the LDS latency is covered by matrix work and
//     no register is rewritten while an MFMA that reads it is in flight;
//   * the global->LDS copies of the NEXT stage (RPS x global_load_lds_dwordx4 per thread, 1 KB
//     per wave-instruction) are spread over the groups: issued back to back at the stage start
//     they cost ~64 cycles of matrix-pipe idle each, interleaved ~20;
//   * sched_barrier pins that order.
//   * `side(g, NG)` is a hook for work placed inside the MFMA stream (unused in production).
template <int NTN, int RPS, int NTHR, class Side>
__device__ __forceinline__ void stage_mma(f32x16 (&acc)[NTN], const f32x16& bop, const lds_f32x4* sb, int lane,
                                          const f32x4* pf_src, lds_f32x4* pf_dst, Side side) {
  constexpr int NG = 4 * NTN / 2;  // groups per stage
  static_assert(NTN % 2 == 0, "n-tiles are processed in pairs");
  f32x4 w[2][2];
  w[0][0] = sb[0 * 64 + lane];
  w[0][1] = sb[1 * 64 + lane];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int q = (2 * g) / NTN, nt = (2 * g) % NTN;
    const int c = g & 1, n = c ^ 1;
    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][0].x, bop[4 * q + 0], acc[nt], 0, 0, 0);
    acc[nt + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][1].x, bop[4 * q + 0], acc[nt + 1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (g + 1 < NG) {
      w[n][0] = sb[(2 * (g + 1)) * 64 + lane];
      w[n][1] = sb[(2 * (g + 1) + 1) * 64 + lane];
    }
#pragma unroll
    for (int r = g * RPS / NG; r < (g + 1) * RPS / NG; ++r)
      __builtin_amdgcn_global_load_lds(GLB_PTR(pf_src + r * NTHR),
                                       (__attribute__((address_space(3))) void*)(pf_dst + r * NTHR), 16, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][0].y, bop[4 * q + 1], acc[nt], 0, 0, 0);
    acc[nt + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][1].y, bop[4 * q + 1], acc[nt + 1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    side(g, NG);
    __builtin_amdgcn_sched_barrier(0);
    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][0].z, bop[4 * q + 2], acc[nt], 0, 0, 0);
    acc[nt + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][1].z, bop[4 * q + 2], acc[nt + 1], 0, 0, 0);
    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][0].w, bop[4 * q + 3], acc[nt], 0, 0, 0);
    acc[nt + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][1].w, bop[4 * q + 3], acc[nt + 1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NTN>
__device__ __forceinline__ void load_bias(f32x16 (&acc)[NTN], const lds_float* lb, int h) {
#pragma unroll
  for (int nt = 0; nt < NTN; ++nt) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b = *(const lds_f32x4*)(lb + 32 * nt + 8 * g + 4 * h);
      acc[nt][4 * g + 0] = b.x; acc[nt][4 * g + 1] = b.y;
      acc[nt][4 * g + 2] = b.z; acc[nt][4 * g + 3] = b.w;
    }
  }
}
// The activations live in the accumulator registers (AGPRs) from the MFMA that produces them to
// the MFMAs that consume them as B operands.  VALU instructions cannot address AGPRs, so the
// elementwise steps go through one scratch VGPR per element, in place, and the "+a" constraint
// keeps the value's home in the accumulator file (otherwise the compiler keeps a second, VGPR
// copy of every activation and runs out of registers).
__device__ __forceinline__ float relu_acc(float x) {
  float t;
  asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_max_f32 %1, 0, %1\n\tv_accvgpr_write_b32 %0, %1" : "+a"(x), "=&v"(t));
  return x;
}
__device__ __forceinline__ float add_acc(float x, float b) {
  float t;
  asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_add_f32 %1, %1, %2\n\tv_accvgpr_write_b32 %0, %1"
               : "+a"(x), "=&v"(t)
               : "v"(b));
  return x;
}
template <int NTN>
__device__ __forceinline__ void add_bias(f32x16 (&acc)[NTN], const lds_float* lb, int h) {
#pragma unroll
  for (int nt = 0; nt < NTN; ++nt) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b = *(const lds_f32x4*)(lb + 32 * nt + 8 * g + 4 * h);
      acc[nt][4 * g + 0] = add_acc(acc[nt][4 * g + 0], b.x);
      acc[nt][4 * g + 1] = add_acc(acc[nt][4 * g + 1], b.y);
      acc[nt][4 * g + 2] = add_acc(acc[nt][4 * g + 2], b.z);
      acc[nt][4 * g + 3] = add_acc(acc[nt][4 * g + 3], b.w);
    }
  }
  asm volatile("s_nop 7" ::: "memory");  // accvgpr_write -> MFMA operand wait states
}
template <int NTN>
__device__ __forceinline__ void relu_tiles(f32x16 (&acc)[NTN]) {
#pragma unroll
  for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = relu_acc(acc[nt][r]);
  asm volatile("s_nop 7" ::: "memory");
}

// Slice g of ng of a 16-register tile: elements [16g/ng, 16(g+1)/ng).
__device__ __forceinline__ void relu_slice(f32x16& t, int g, int ng) {
#pragma unroll
  for (int r = 16 * g / ng; r < 16 * (g + 1) / ng; ++r) t[r] = relu_acc(t[r]);
}
// lb_tile: bias of this tile's 32 features; register r of lane half h holds feature 8(r>>2)+4h+(r&3)
__device__ __forceinline__ void addb_slice(f32x16& t, const lds_float* lb_tile, int h, int g, int ng) {
#pragma unroll
  for (int r = 16 * g / ng; r < 16 * (g + 1) / ng; ++r) t[r] = add_acc(t[r], lb_tile[8 * (r >> 2) + 4 * h + (r & 3)]);
}
__device__ __forceinline__ void setb_slice(f32x16& t, const lds_float* lb_tile, int h, int g, int ng) {
#pragma unroll
  for (int r = 16 * g / ng; r < 16 * (g + 1) / ng; ++r) t[r] = lb_tile[8 * (r >> 2) + 4 * h + (r & 3)];
}
// wait states between an MFMA's result write and a VALU read of it inside an asm statement
#define CS_MFMA_RESULT_NOPS() asm volatile("s_nop 15\n\ts_nop 7" ::: "memory")

// ----------------------------------------------------------------------------------------
template <int NT, int NTO, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(1, 1)))
cs_mlp_kernel(const CsMlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE_F4 = 4 * NT * 64;   // float4 per hidden-layer stage
  constexpr int OSTAGE_F4 = 4 * NTO * 64; // float4 per output-layer stage
  constexpr int SLOT_F4 = STAGE_F4 > OSTAGE_F4 ? STAGE_F4 : OSTAGE_F4;  // ring slot
  constexpr int nthr = 64 * WAVES;
  constexpr int RPS = SLOT_F4 / nthr;  // global->LDS copies per thread per stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int nl = 2 * a.net.nb + 2;
  const int bst = a.net.bias_stride;
  lds_float* lbias = (lds_float*)smem;
  lds_f32x4* ring = (lds_f32x4*)(smem + (((size_t)nl * bst * 4 + 1023) & ~(size_t)1023));

  const int64_t row = ((int64_t)blockIdx.x * (nthr >> 6) + wave) * 32 + j;
  const bool rvalid = row < a.rows;
  int dbg_i = 0;
#define CS_STAMP()                                                                         \
  do {                                                                                     \
    if (a.dbg && lane == 0) a.dbg[(row >> 5) * 16 + (dbg_i < 15 ? dbg_i : 15)] = clock64(); \
    ++dbg_i;                                                                               \
  } while (0)
  CS_STAMP();

  // ---- operand-stream prefetch machinery ----
  const f32x4* gp = a.net.wstream;
  int par = 0;
  const lds_f32x4* sb = ring;
  const f32x4* pf_src = gp + tid;           // this thread's source of the next slot
  lds_f32x4* pf_dst = ring + (wave << 6);   // this wave's destination in the next slot
#pragma unroll
  for (int r = 0; r < RPS; ++r)             // stage 0
    __builtin_amdgcn_global_load_lds(GLB_PTR(pf_src + r * nthr), (__attribute__((address_space(3))) void*)(pf_dst + r * nthr),
                                     16, 0, 0);
  for (int i = tid; i < nl * bst; i += nthr) lbias[i] = a.net.bias[i];

  // Enter the next stage: its slot has landed (own copies waited for, then the barrier covers
  // the other waves' copies and their reads of the slot about to be overwritten); point the
  // prefetch at the stage after it.
#define CS_NEXT_STAGE(CUR_F4)                                   \
  do {                                                          \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            \
    __syncthreads();                                            \
    gp += (CUR_F4);                                             \
    pf_src = gp + tid;                                          \
    pf_dst = ring + (par ^ 1) * SLOT_F4 + (wave << 6);          \
    sb = ring + par * SLOT_F4;                                  \
    par ^= 1;                                                   \
  } while (0)

  // ---- prologue: B-operand fragments of the network input ----
  // xin[kt][4q+r] = in[row][32kt + 8q + 4h + r]
  f32x16 xin[4];
  int64_t b_idx = 0, kp_idx = 0;
  if (a.mode == CS_MLP_DEC) {
    const int64_t bk = row / a.Kp;
    kp_idx = row - bk * a.Kp;
    b_idx = bk % a.B;
  } else if (a.mode == CS_MLP_ENC) {
    b_idx = row / a.K;
    kp_idx = row - b_idx * a.K;
  }
  if (a.mode == CS_MLP_DEC && a.in == nullptr) {
    const int Zd = a.Zd;
    const float* qr = a.q + (b_idx * a.Kp + kp_idx) * 2 * Zd;
    const bool ovr = rvalid && a.z_override != nullptr && a.z_override[row] != 0;
    const int64_t i_idx = rvalid ? row / ((int64_t)a.Kp * a.B) : 0;
    const int64_t prow = (a.rng.row_offset + b_idx) * ((int64_t)a.I * a.Kp) + i_idx * a.Kp + kp_idx;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = 32 * kt + 8 * q + 4 * h;
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (kt < a.net.nkt0 && rvalid && c0 < Zd) {
          f32x4 e;
          if (a.eps_z == nullptr && !ovr) {
            uint32_t x[4];
            cs_philox_block(a.rng, CS_STREAM_Z, prow, (uint32_t)(c0 >> 2), x);
            e = cs_normal4(x);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = c0 + r;
            if (c < Zd) {
              if (ovr) {
                z[r] = a.z_alt[row * Zd + c];
              } else {
                const float ev = a.eps_z ? a.eps_z[row * Zd + c] : e[r];
                z[r] = qr[c] + qr[Zd + c] * ev;
              }
              a.Z[row * Zd + c] = z[r];
            }
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) xin[kt][4 * q + r] = z[r];
      }
    }
  } else {
    const float* xr = a.in + (a.in_off ? (rvalid ? a.in_off[row] : 0) : row * a.in_stride);
    const int in_dim = a.net.in_dim;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 32 * kt + 8 * q + 4 * h + r;
          xin[kt][4 * q + r] = (kt < a.net.nkt0 && rvalid && c < in_dim) ? xr[c] : 0.0f;
        }
      }
    }
  }
  __syncthreads();  // biases visible
  CS_STAMP();

  // ---- first layer ----
  // Elementwise work (bias, ReLU) runs between the layers, not inside the MFMA stream: on
  // gfx950 a VALU instruction issued by the wave that owns the in-flight MFMA delays that wave's
  // next MFMA by more than its own issue time (measured: interleaving was 5 % slower).
  f32x16 A[NT], T[NT];
  auto no_side = [](int, int) {};
  load_bias<NT>(A, lbias, h);
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    if (kt < a.net.nkt0) {
      CS_NEXT_STAGE(STAGE_F4);
      stage_mma<NT, RPS, nthr>(A, xin[kt], sb, lane, pf_src, pf_dst, no_side);
    }
  }
  CS_MFMA_RESULT_NOPS();
  relu_tiles<NT>(A);
  CS_STAMP();

  // ---- residual blocks ----
  for (int blk = 0; blk < a.net.nb; ++blk) {
    load_bias<NT>(T, lbias + (1 + 2 * blk) * bst, h);
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      CS_NEXT_STAGE(STAGE_F4);
      stage_mma<NT, RPS, nthr>(T, A[kt], sb, lane, pf_src, pf_dst, no_side);
    }
    CS_MFMA_RESULT_NOPS();
    relu_tiles<NT>(T);
    add_bias<NT>(A, lbias + (2 + 2 * blk) * bst, h);
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      CS_NEXT_STAGE(STAGE_F4);
      stage_mma<NT, RPS, nthr>(A, T[kt], sb, lane, pf_src, pf_dst, no_side);
    }
    CS_MFMA_RESULT_NOPS();
    relu_tiles<NT>(A);
    CS_STAMP();
  }

  // ---- output layer ----
  f32x16 O[NTO];
  load_bias<NTO>(O, lbias + (nl - 1) * bst, h);
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
    CS_NEXT_STAGE(OSTAGE_F4);
    stage_mma<NTO, RPS, nthr>(O, A[kt], sb, lane, pf_src, pf_dst, no_side);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the trailing prefetch
#undef CS_NEXT_STAGE
  CS_STAMP();

  // ---- epilogues ----
  const int tph = a.net.tph;
  if (a.mode == CS_MLP_RAW) {
    if (rvalid) {
      float* yr = a.out + row * a.net.out_dim;
#pragma unroll
      for (int t = 0; t < NTO; ++t) {
        const int half = t / tph, tt = t - half * tph;
        if (half < a.net.split) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int w = 32 * tt + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (w < a.net.out_half) yr[half * a.net.out_half + w] = O[t][r];
          }
        }
      }
    }
    CS_STAMP();
    return;
  }
  // paired (first-half, second-half) epilogues: tile t pairs with tile t + tph
  if (a.mode == CS_MLP_ENC) {
    if (rvalid) {
      const int Zd = a.Zd;
      float* qr = a.q + (b_idx * a.Kp + kp_idx) * 2 * Zd;
#pragma unroll
      for (int t = 0; t < NTO / 2; ++t) {
        if (t < tph) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int z = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (z < Zd) {
              float sc = cs_softplus(O[t + NTO / 2][r]);
              if (a.latent_eps) sc += 1e-8f;
              qr[z] = O[t][r];
              qr[Zd + z] = sc;
            }
          }
        }
      }
      // prior-replenishing proposals (vae_imputation.py:563-577): loc 0 and, after
      // softplus(+1e-8), scale exactly 1; written once per example by the lanes of chain 0
      if (kp_idx == 0 && a.Kp > a.K) {
        float* qp = a.q + (b_idx * a.Kp + a.K) * 2 * Zd;
        for (int t = h; t < (a.Kp - a.K) * 2 * Zd; t += 2) qp[t] = ((t / Zd) & 1) ? 1.0f : 0.0f;
      }
    }
    CS_STAMP();
    return;
  }
  // CS_MLP_DEC
  {
    const int D = a.D;
    float ll = 0.0f;
    const bool want_ll = a.gen_ll != nullptr;
    if (rvalid) {
      const int64_t xk = kp_idx < a.K ? kp_idx : 0;
      const float* xr = want_ll ? a.X + (b_idx * a.K + xk) * D : nullptr;
      const uint8_t* mr = want_ll ? a.M + b_idx * D : nullptr;
      if (a.gen_dist == CS_DIST_BERN) {
        float* gr = a.gen + row * D;
#pragma unroll
        for (int t = 0; t < NTO; ++t) {
          if (t < tph) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int d = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
              if (d < D) {
                const float l = O[t][r];
                gr[d] = l;
                if (want_ll) ll += cs_bernoulli_logpmf(xr[d], l) * (float)mr[d];
              }
            }
          }
        }
      } else {
        float* gr = a.gen + row * 2 * D;
        const float eps = a.gen_dist == CS_DIST_NORMAL_WITH_EPS ? 1e-8f : 0.0f;
#pragma unroll
        for (int t = 0; t < NTO / 2; ++t) {
          if (t < tph) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int d = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
              if (d < D) {
                const float mean = O[t][r];
                const float sc = cs_softplus(O[t + NTO / 2][r]) + eps;
                gr[d] = mean;
                gr[D + d] = sc;
                if (want_ll) ll += cs_normal_logpdf(xr[d], mean, sc) * (float)mr[d];
              }
            }
          }
        }
      }
    }
    ll += __shfl_xor(ll, 32);
    if (want_ll && rvalid && h == 0) a.gen_ll[row] = ll;
    CS_STAMP();
  }
}

// ----------------------------------------------------------------------------------------
template <int NT, int NTO, int WAVES>
static int launch_w(const CsMlpArgs& a, hipStream_t stream) {
  const int64_t tiles = (a.rows + 31) / 32;
  const int64_t grid = (tiles + WAVES - 1) / WAVES;
  const int nl = 2 * a.net.nb + 2;
  const size_t bias_bytes = (((size_t)nl * a.net.bias_stride * 4 + 1023) & ~(size_t)1023);
  const size_t lds = bias_bytes + 2 * (size_t)(4 * (NT > NTO ? NT : NTO) * 64) * 16;
  CS_REQUIRE(lds <= 160 * 1024, "network does not fit the LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    CS_HIP(hipFuncSetAttribute((const void*)cs_mlp_kernel<NT, NTO, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((cs_mlp_kernel<NT, NTO, WAVES>), dim3((unsigned)grid), dim3(64 * WAVES), lds, stream, a);
  CS_HIP(hipGetLastError());
  return CS_OK;
}

template <int NT, int NTO>
static int launch_t(const CsMlpArgs& a, hipStream_t stream) {
  // small batches: one wavefront per workgroup so that more CUs take part
  const int64_t tiles = (a.rows + 31) / 32;
  return tiles >= 4 * 256 ? launch_w<NT, NTO, 4>(a, stream) : launch_w<NT, NTO, 1>(a, stream);
}

static long long* g_trace = nullptr;
// Profiling hook: when set, every fused-MLP launch writes 16 clock64() stamps per wavefront
// (row tile) into buf: start, after prologue, after the first layer, after each block, after
// the output layer, end.  buf must hold 16 * ceil(rows/32) entries.  NULL disables it.
extern "C" int cs_debug_set_trace(long long* buf) {
  g_trace = buf;
  return CS_OK;
}

int cs_mlp_launch(const CsMlpArgs& a_in, hipStream_t stream) {
  CsMlpArgs a = a_in;
  a.dbg = g_trace;
  if (a.rows <= 0) return CS_OK;
  const int nt = a.net.nt, nto = a.net.nto;
  CS_REQUIRE(a.rows < ((int64_t)1 << 31) * 32, "too many rows");
  if (a.net.split == 2) CS_REQUIRE(nto == 2 * a.net.tph, "bad tile split");
  if (a.mode == CS_MLP_ENC) CS_REQUIRE(a.net.split == 2, "encoder output must be (loc | scale)");
  if (a.mode == CS_MLP_DEC && a.gen_dist != CS_DIST_BERN) CS_REQUIRE(a.net.split == 2, "generator output must be (loc | scale)");
#define CS_CASE(NT_, NTO_) \
  if (nt == NT_ && nto == NTO_) return launch_t<NT_, NTO_>(a, stream);
  CS_CASE(2, 2) CS_CASE(2, 4) CS_CASE(4, 2) CS_CASE(4, 4) CS_CASE(8, 2) CS_CASE(8, 4)
#undef CS_CASE
  return cs_fail(CS_EUNSUPPORTED, "no kernel instance for hidden=" + std::to_string(nt * 32) +
                                      " output tiles=" + std::to_string(nto));
}

// ----------------------------------------------------------------------------------------
// host: packing
// ----------------------------------------------------------------------------------------
namespace {
// Append one layer (row-major W[n_real][k_real]) to the operand stream.
// tile_row(t, i) -> source row of output tile t, lane row i (or -1 for padding).
template <class RowFn>
void pack_layer(std::vector<float>& s, const float* W, int k_real, int kt_count, int ntiles, RowFn tile_row) {
  for (int kt = 0; kt < kt_count; ++kt)
    for (int q = 0; q < 4; ++q)
      for (int nt = 0; nt < ntiles; ++nt)
        for (int lane = 0; lane < 64; ++lane) {
          const int h = lane >> 5, i = lane & 31;
          const int n = tile_row(nt, i);
          for (int r = 0; r < 4; ++r) {
            const int k = 32 * kt + 8 * q + 4 * h + r;
            s.push_back((n >= 0 && k < k_real) ? W[(size_t)n * k_real + k] : 0.0f);
          }
        }
}
}  // namespace

extern "C" int cs_mlp_create(const cs_mlp_desc* d, const float* W0, const float* b0, const float* Wblk,
                             const float* bblk, const float* Wf, const float* bf, cs_mlp_t** out) {
  CS_REQUIRE(d && W0 && b0 && Wf && bf && out, "null argument");
  CS_REQUIRE(d->num_blocks == 0 || (Wblk && bblk), "null block weights");
  const int H = d->hidden, nb = d->num_blocks;
  CS_REQUIRE(H == 64 || H == 128 || H == 256, "hidden width must be 64, 128 or 256");
  CS_REQUIRE(d->in_dim >= 1 && d->in_dim <= 128, "in_dim must be in [1,128]");
  CS_REQUIRE(d->out_split == 1 || d->out_split == 2, "out_split must be 1 or 2");
  CS_REQUIRE(d->out_dim >= 1 && d->out_dim % d->out_split == 0, "out_dim not divisible by out_split");
  CS_REQUIRE(nb >= 0 && nb <= 16, "num_blocks must be in [0,16]");
  const int out_half = d->out_dim / d->out_split;
  const int tph = (out_half + 31) / 32;
  int nto = d->out_split * tph;
  CS_REQUIRE(nto <= 4, "output too wide (each half must be <= 64, or <= 128 unsplit)");
  if (nto == 1 || nto == 3) nto += 1;  // kernel instances exist for 2 and 4 output tiles
  const int NT = H / 32;
  const int nkt0 = (d->in_dim + 31) / 32;
  const int bst = 32 * (NT > nto ? NT : nto);

  std::vector<float> s;
  s.reserve((size_t)(nkt0 + 2 * nb * NT) * 4 * NT * 64 * 4 + (size_t)NT * 4 * nto * 64 * 4 + 4 * NT * 64 * 4);
  auto hid_row = [&](int t, int i) { return 32 * t + i; };
  pack_layer(s, W0, d->in_dim, nkt0, NT, hid_row);
  for (int b = 0; b < nb; ++b)
    for (int l = 0; l < 2; ++l) pack_layer(s, Wblk + ((size_t)b * 2 + l) * H * H, H, NT, NT, hid_row);
  auto out_row = [&](int t, int i) {
    const int half = t / tph, w = 32 * (t - half * tph) + i;
    return (half < d->out_split && w < out_half) ? half * out_half + w : -1;
  };
  pack_layer(s, Wf, H, NT, nto, out_row);
  s.resize(s.size() + (size_t)4 * (NT > nto ? NT : nto) * 64 * 4, 0.0f);  // the last prefetch reads one slot past the end

  const int nl = 2 * nb + 2;
  std::vector<float> bias((size_t)nl * bst, 0.0f);
  for (int i = 0; i < H; ++i) bias[i] = b0[i];
  for (int b = 0; b < nb; ++b)
    for (int l = 0; l < 2; ++l)
      for (int i = 0; i < H; ++i) bias[(size_t)(1 + 2 * b + l) * bst + i] = bblk[((size_t)b * 2 + l) * H + i];
  for (int t = 0; t < nto; ++t)
    for (int i = 0; i < 32; ++i) {
      const int n = out_row(t, i);
      if (n >= 0) bias[(size_t)(nl - 1) * bst + 32 * t + i] = bf[n];
    }

  cs_mlp* m = new cs_mlp();
  m->desc = *d;
  m->d_w = nullptr;
  m->d_b = nullptr;
  hipError_t e = hipGetDevice(&m->device);
  if (e == hipSuccess) e = hipMalloc(&m->d_w, s.size() * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&m->d_b, bias.size() * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(m->d_w, s.data(), s.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(m->d_b, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (m->d_w) (void)hipFree(m->d_w);
    if (m->d_b) (void)hipFree(m->d_b);
    delete m;
    return cs_fail(CS_EHIP, std::string("cs_mlp_create: ") + hipGetErrorString(e));
  }
  m->dev.wstream = (const f32x4*)m->d_w;
  m->dev.bias = (const float*)m->d_b;
  m->dev.nb = nb;
  m->dev.nkt0 = nkt0;
  m->dev.in_dim = d->in_dim;
  m->dev.out_dim = d->out_dim;
  m->dev.out_half = out_half;
  m->dev.tph = tph;
  m->dev.nt = NT;
  m->dev.nto = nto;
  m->dev.split = d->out_split;
  m->dev.bias_stride = bst;
  *out = m;
  return CS_OK;
}

extern "C" int cs_mlp_destroy(cs_mlp_t* m) {
  if (!m) return CS_OK;
  (void)hipFree(m->d_w);
  (void)hipFree(m->d_b);
  delete m;
  return CS_OK;
}

extern "C" int cs_mlp_forward(const cs_mlp_t* net, const float* x, int64_t rows, float* y, void* stream) {
  CS_REQUIRE(net && x && y, "null argument");
  CS_REQUIRE(rows >= 0, "negative row count");
  CsMlpArgs a;
  std::memset(&a, 0, sizeof(a));
  a.net = net->dev;
  a.mode = CS_MLP_RAW;
  a.rows = rows;
  a.in = x;
  a.in_stride = net->dev.in_dim;
  a.out = y;
  return cs_mlp_launch(a, (hipStream_t)stream);
}
