// Layer-program networks: convolutional ResNet encoder / decoder and wide MLPs.
//
// Replaces ResNetEncoder.forward (irwg/models/vae_resnet.py:128-149; EncoderBlock :60-78),
// ResNetDecoder.forward (:259-277; DecoderBlock :166-184; Interpolate :14-22) in eval mode
// (BatchNorm folded by the caller, Dropout2d = identity), and ResidualFCNetwork.forward
// (irwg/models/neural_nets.py:163-169) for widths the fused kernel does not cover.
//
// Execution model: one kernel per layer over NHWC activations held in a per-handle workspace;
// rows (images) are processed in chunks so that the workspace stays within a fixed budget.
//   * convolutions / linear layers with cin, cout multiples of 32: implicit GEMM on the matrix cores.  Arithmetic by
//     the handle's / process's setting: exact f32 (v_mfma_f32_32x32x2_f32: cs_conv_mfma_kernel, cs_conv3x3_halo_kernel),
//     the 3-term bf16 split, or (default) the 2-term fp16 split of the convolutions with one power-of-two operand scale
//     per image (cs_conv3x3_halo_split_kernel, cs_conv3x3_halo_resident_kernel, cs_conv_mfma_split_kernel; TERMS);
//   * everything else (first convolution from 1 channel, last convolution to 1-2 channels, the
//     small linear layers): cs_conv_direct_kernel, one thread per output element;
//   * the distribution parameterisation of VAE.create_distribution (irwg/models/vae.py:203-256),
//     the masked log-likelihood (irwg/utils/weights.py:46-47) and Z = loc + scale*eps
//     (vae_imputation.py:596-597) run as separate elementwise kernels around the layers, with the
//     same buffer semantics as the prologue / epilogues of the fused MLP kernel.
#include "cs_net.h"

#include <cstring>
#include <mutex>

namespace {

// ----------------------------------------------------------------------------------------
// implicit-GEMM convolution on the matrix cores
// ----------------------------------------------------------------------------------------
struct CsConvArgs {
  const float* in;    // [n, hin, win, cin]
  const float* w;     // packed [tap][cin_p/32][cout_p][32] (mfma; _p = padded to 32, zero filled) or
                      // torch [cout][cin][k][k] (direct)
  const uint16_t* w3; // split-bf16 packed [tap][cin/32][cout_p][term 3][32 k] (see cs_conv3x3_halo_split_kernel) or null
  const float* bias;  // [cout] or null
  const float* res;   // [n, hout, wout, cout] or null
  float* out;
  int cin, cout, hin, win, hout, wout, ksize, stride, pad_h, pad_w, us, relu, in_nchw, out_nchw;
  int cout_p;    // cout rounded up to 32 (mfma)
  int64_t npix;  // n * hout * wout
  // Output placement.  osc = 1: pixel (i, j) of the hout x wout grid is output pixel (i, j).  osc = 2 (phase of a
  // nearest-neighbour-upsampled convolution): it is pixel (2i + ooh, 2j + oow) of a 2*hout x 2*wout output; with
  // rep it is written to all four pixels (2i + {0,1}, 2j + {0,1}), each with its own residual.
  int osc, ooh, oow, rep;
  // all four phases of an upsampled convolution in one launch of the halo kernel (grid z = phase 2*ooh + oow):
  // distance in bf16 entries between the packed weights of consecutive phases
  int64_t w3_phase_stride;
  // fp16-split arithmetic of the halo kernel (null / unused otherwise): weights as two fp16 terms of w 2^wexp, packed
  // [tap][cin/32][cout_p/32] bricks of [k-step 2][term 2][lane 64] x 8 fp16; wexp per phase of a four-phase launch,
  // else wexp[0]; in_amax[n] = bits of the largest |in| of image n of this launch (the power of two that brings it
  // into [2^14, 2^15) scales the image's operands); out_amax[n] receives the same for the output (atomic max,
  // zeroed by the caller), or null.  amax_lds = byte offset of the block's table of output maxima in LDS.
  const uint16_t* w3h;
  int wexp[4];
  int64_t w3h_phase_stride;
  const unsigned* in_amax;
  unsigned* out_amax;
  int amax_lds;
};

// Joins `bits` (the bits of a non-negative float) to a per-image maximum in global memory.  (Measured and not kept: a
// load of the word in front of the atomic, to skip atomics that cannot raise it -- thousands of wavefronts reading
// one word serialise on it as much as their atomics do.  Callers keep the number of joins per image small instead.)
__device__ __forceinline__ void conv_amax_join(unsigned* p, unsigned bits) { atomicMax(p, bits); }

__device__ __forceinline__ int64_t conv_out_pix(const CsConvArgs& a, int64_t gp, int ooh, int oow) {
  if (a.osc == 1) return gp;
  const int hw = a.hout * a.wout;
  const int64_t n = gp / hw;
  const int rem = (int)(gp - n * hw);
  const int i = rem / a.wout, j = rem - i * a.wout;
  return (n * (2 * a.hout) + 2 * i + ooh) * (2 * a.wout) + 2 * j + oow;
}

// Epilogue of the matrix-core convolutions: the lane holds pixel `gp`; register quad q of tile ct
// holds channels co0 + 32ct + 8q + 4h + 0..3.  Bias, residual and ReLU are applied here.
template <int CT>
__device__ __forceinline__ void conv_store(const CsConvArgs& a, const f32x16 (&acc)[CT], int64_t gp, int co0, int h) {
  // out_amax (NHWC outputs of 32-multiple channel counts only): the largest magnitude the lane stores, merged over
  // the wave's lanes of the same image (consecutive lanes hold consecutive pixels) before one atomic per image
  float amx = 0.0f;
  const bool want_amax = a.out_amax != nullptr && (a.cout & 31) == 0 && !a.out_nchw;
  if (gp < a.npix && (a.cout & 31) == 0 && !a.out_nchw) {
    for (int ph = 0; ph < (a.rep ? 4 : 1); ++ph) {
      const int64_t op = a.rep ? conv_out_pix(a, gp, ph >> 1, ph & 1) : conv_out_pix(a, gp, a.ooh, a.oow);
      float* orow = a.out + op * a.cout + co0;
      const float* rrow = a.res ? a.res + op * a.cout + co0 : nullptr;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = ct * 32 + 8 * q + 4 * h;
          f32x4 v = {acc[ct][4 * q], acc[ct][4 * q + 1], acc[ct][4 * q + 2], acc[ct][4 * q + 3]};
          if (a.bias) v += *(const f32x4*)(a.bias + co0 + c);
          if (rrow) v += *(const f32x4*)(rrow + c);
          if (a.relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
          }
          *(f32x4*)(orow + c) = v;
          if (want_amax) amx = fmaxf(fmaxf(amx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
    }
  } else if (gp < a.npix) {
    // scalar path: channel counts that are not multiples of 32, channel-major outputs (osc = 1 only)
    const int hw = a.hout * a.wout;
    const int64_t n = gp / hw;
    const int rem = (int)(gp - n * hw);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + ct * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
        if (co < a.cout) {
          float v = acc[ct][r];
          if (a.bias) v += a.bias[co];
          if (a.res) v += a.res[gp * a.cout + co];
          if (a.relu) v = fmaxf(v, 0.0f);
          if (a.out_nchw) a.out[(n * a.cout + co) * hw + rem] = v;
          else a.out[gp * a.cout + co] = v;
        }
      }
  }
  if (want_amax) {  // every lane of the wave arrives here
    const int lane = (int)(threadIdx.x & 63);
    const int img = gp < a.npix ? (int)(gp / (a.hout * a.wout)) : -1;
    amx = fmaxf(amx, __shfl_xor(amx, 32));  // the two halves of a pixel's channels
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      const float om = __shfl_down(amx, o);
      const int oi = __shfl_down(img, o);
      if ((lane & 31) + o < 32 && oi == img) amx = fmaxf(amx, om);
    }
    const int pi = __shfl_up(img, 1);
    if (lane < 32 && img >= 0 && ((lane & 31) == 0 || pi != img) && amx > 0.0f) conv_amax_join(a.out_amax + img, __float_as_uint(amx));
  }
}

// Coalesced form of conv_store for NHWC outputs whose channel count is a multiple of 32: the wavefront's PT x CT
// accumulator tiles (32*PT pixels x 32*CT channels) are transposed through a wave-private LDS region `tw` of
// 32*PT rows x (128*CT + 16) bytes, so that every store (and residual load) instruction of the wavefront covers
// 1 KiB of whole pixel rows instead of 32 separate 32-byte pieces.  The caller has made sure that nobody reads the
// region's previous contents any more.  row_gp = global pixel of the wavefront's first row.  PLACED: the rows are
// phase (ooh, oow) of an upsampled output (conv_out_pix), else output pixel = pixel.
typedef __attribute__((address_space(3))) unsigned char conv_lds_byte;
typedef __attribute__((address_space(3))) f32x4 conv_lds_f32x4;
// s_amax (or null): LDS table of the block's per-image output maxima (bits of |value|, zeroed by the caller); pos0 =
// position of the wavefront's first row counted from the first pixel of the block's first image, hw = pixels per image.
template <int CT, int PT, bool PLACED>
__device__ __forceinline__ void conv_store_rows(const CsConvArgs& a, const f32x16 (&acc)[PT][CT], conv_lds_byte* tw,
                                                int64_t row_gp, int co0, int lane, int ooh, int oow,
                                                __attribute__((address_space(3))) unsigned* s_amax = nullptr, int pos0 = 0,
                                                int hw = 1) {
  constexpr int RL = CT * 128, RS = RL + 16, LPR = CT * 8, RPI = 64 / LPR;
  const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int p = 0; p < PT; ++p)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(conv_lds_f32x4*)(tw + (p * 32 + l31) * RS + (ct * 32 + 8 * q + 4 * h) * 4) =
            f32x4{acc[p][ct][4 * q], acc[p][ct][4 * q + 1], acc[p][ct][4 * q + 2], acc[p][ct][4 * q + 3]};
  const int lr = lane / LPR, lc = (lane % LPR) * 4;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) bias4 = *(const f32x4*)(a.bias + co0 + lc);
  f32x4 v[32 * PT / RPI];
#pragma unroll
  for (int i = 0; i < 32 * PT / RPI; ++i) {
    const int row = i * RPI + lr;
    const int64_t gp = row_gp + row;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.res && gp < a.npix) v[i] = *(const f32x4*)(a.res + (PLACED ? conv_out_pix(a, gp, ooh, oow) : gp) * a.cout + co0 + lc);
  }
  // running maximum of the lane's rows of one image; handed to the table when the rows move on to the next image
  int img = 0, next_img = 0;
  float mx = 0.0f;
  if (s_amax) {
    img = (pos0 + lr) / hw;
    next_img = (img + 1) * hw;
  }
#pragma unroll
  for (int i = 0; i < 32 * PT / RPI; ++i) {
    const int row = i * RPI + lr;
    const int64_t gp = row_gp + row;
    f32x4 o = *(const conv_lds_f32x4*)(tw + row * RS + lc * 4) + bias4 + v[i];
    if (a.relu) {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], 0.0f);
    }
    if (gp < a.npix) *(f32x4*)(a.out + (PLACED ? conv_out_pix(a, gp, ooh, oow) : gp) * a.cout + co0 + lc) = o;
    if (s_amax && gp < a.npix) {
      while (pos0 + row >= next_img) {
        if (mx > 0.0f) atomicMax((unsigned*)(s_amax + img), __float_as_uint(mx));
        mx = 0.0f;
        ++img;
        next_img += hw;
      }
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
    }
  }
  if (s_amax && mx > 0.0f) atomicMax((unsigned*)(s_amax + img), __float_as_uint(mx));
}

// Block = 4 wavefronts = 128 output pixels x (32*CT) output channels; a wavefront owns 32 pixels.
// D[cout][pixel] = sum_k W[cout][k] * act[pixel][k]: the A operand is the weight tile, the B
// operand the activation tile, so a lane ends up with 4 consecutive output channels of one pixel
// per register quad (16-byte stores into NHWC).  K runs over (tap, 32-channel chunk); per chunk
// both tiles sit in LDS as [row][32 k] (row stride 36 floats), and each lane reads 4 consecutive
// k with one ds_read_b128: within a group of 8 k, MFMA step j uses k = 4*(lane>>5) + j on both
// operands, which covers the 8 values once.  The next chunk is fetched into registers while the
// current one is multiplied.  Channel counts that are not multiples of 32 are zero padded in the
// packed weights; the activation loads and the stores of such layers take a guarded scalar path
// (the decoder's first linear layer, the last convolution to 1-2 channels in either memory order).
template <int CT, int PT>
__global__ void __launch_bounds__(256) cs_conv_mfma_kernel(const CsConvArgs a) {
  constexpr int RS = 36;
  constexpr int NPIX = 128 * PT;  // output pixels per block; a wavefront owns 32 * PT of them
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  float* s_pix = s_dyn;              // [NPIX][RS]
  float* s_w = s_dyn + NPIX * RS;    // [CT*32][RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int c4 = tid & 7;
  const int ncc = (a.cin + 31) >> 5;
  const bool cin_vec = (a.cin & 31) == 0;
  const int nchunks = a.ksize * a.ksize * ncc;
  const int hup = a.hin << a.us, wup = a.win << a.us;

  // the 4*PT activation rows this thread stages
  int ih0[4 * PT], iw0[4 * PT];
  int64_t base[4 * PT];
  bool pv[4 * PT];
#pragma unroll
  for (int p = 0; p < 4 * PT; ++p) {
    const int64_t gp = (int64_t)blockIdx.x * NPIX + (tid >> 3) + 32 * p;
    pv[p] = gp < a.npix;
    const int64_t g = pv[p] ? gp : 0;
    const int hw = a.hout * a.wout;
    const int64_t n = g / hw;
    const int rem = (int)(g - n * hw);
    const int oh = rem / a.wout, ow = rem - oh * a.wout;
    ih0[p] = oh * a.stride - a.pad_h;
    iw0[p] = ow * a.stride - a.pad_w;
    base[p] = n * a.hin * a.win;
  }
  const float* wblk = a.w + (int64_t)blockIdx.y * (CT * 32) * 32;

  f32x16 acc[PT][CT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pt][ct][r] = 0.0f;
  f32x4 rp[4 * PT], rw[CT];
  auto fetch = [&](int t) {
    const int tap = t / ncc, cc = t - tap * ncc;
    const int kh = tap / a.ksize, kw = tap - kh * a.ksize;
#pragma unroll
    for (int p = 0; p < 4 * PT; ++p) {
      const int uh = ih0[p] + kh, uw = iw0[p] + kw;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pv[p] && uh >= 0 && uh < hup && uw >= 0 && uw < wup) {
        const int c0 = cc * 32 + c4 * 4;
        const float* src = a.in + ((base[p] + (int64_t)(uh >> a.us) * a.win + (uw >> a.us)) * a.cin + c0);
        if (cin_vec) {
          v = *(const f32x4*)src;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (c0 + r < a.cin) v[r] = src[r];
        }
      }
      rp[p] = v;
    }
    const float* wt = wblk + (int64_t)t * a.cout_p * 32;
#pragma unroll
    for (int i = 0; i < CT; ++i) rw[i] = *(const f32x4*)(wt + (tid + 256 * i) * 4);
  };
  auto stash = [&]() {
#pragma unroll
    for (int p = 0; p < 4 * PT; ++p) *(f32x4*)(s_pix + ((tid >> 3) + 32 * p) * RS + c4 * 4) = rp[p];
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const int idx = tid + 256 * i;
      *(f32x4*)(s_w + (idx >> 3) * RS + (idx & 7) * 4) = rw[i];
    }
  };

  // One LDS stage; the next chunk waits in registers while this one is multiplied.  (Two LDS
  // stages with one barrier per chunk were measured slower: fewer blocks fit a CU.)
  fetch(0);
  const float* sp = s_pix + (wave * 32 * PT + l31) * RS + 4 * h;
  const float* sw = s_w + l31 * RS + 4 * h;
  for (int t = 0; t < nchunks; ++t) {
    __syncthreads();  // previous chunk's reads are done
    stash();
    __syncthreads();
    if (t + 1 < nchunks) fetch(t + 1);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 b[PT], w[CT];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) b[pt] = *(const f32x4*)(sp + pt * 32 * RS + 8 * g);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) w[ct] = *(const f32x4*)(sw + ct * 32 * RS + 8 * g);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
            acc[pt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ct][j], b[pt][j], acc[pt][ct], 0, 0, 0);
    }
  }

  cs_mfma_drain();
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
    conv_store<CT>(a, acc[pt], (int64_t)blockIdx.x * NPIX + wave * 32 * PT + pt * 32 + l31, blockIdx.y * (CT * 32), h);
}

// 3x3, stride 1, pad 1 convolutions that keep the spatial size (most layers of the ResNets): output
// pixels and input pixels share one linear index, so the 9 taps of a block's 128 output pixels
// read one contiguous window of 128 + 2(W+1) input pixels.  That window ("halo tile") is staged in
// LDS ONCE per 32-channel chunk and reused by all 9 taps (per-lane row offset kh*W + kw, zeroed
// where the tap leaves the image), which cuts the activation traffic of the generic kernel ~6x;
// the 9 weight tiles of the chunk stream through a two-stage LDS ring (one barrier per tap).
template <int CT>
__global__ void __launch_bounds__(256) cs_conv3x3_halo_kernel(const CsConvArgs a, int NR, int NP) {
  constexpr int RS = 36, W_F = CT * 32 * RS;
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  float* s_w = s_dyn;              // [2][CT*32][RS]
  float* s_halo = s_dyn + 2 * W_F; // [NR][RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31, c4 = tid & 7;
  const int W = a.win, H = a.hin;
  const int ncc = a.cin >> 5;
  const int64_t p0 = (int64_t)blockIdx.x * 128;
  const float* wblk = a.w + (int64_t)blockIdx.y * (CT * 32) * 32;

  // which of the 9 taps stay inside the image for this lane's output pixel
  const int64_t gp = p0 + wave * 32 + l31;
  unsigned tapmask = 0;
  {
    const int hw = H * W;
    const int64_t g = gp < a.npix ? gp : 0;
    const int rem = (int)(g % hw);
    const int oh = rem / W, ow = rem - oh * W;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ih = oh + t / 3 - 1, iw = ow + t % 3 - 1;
      if (gp < a.npix && ih >= 0 && ih < H && iw >= 0 && iw < W) tapmask |= 1u << t;
    }
  }

  f32x16 acc[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.0f;

  constexpr int NPMAX = 8;  // halo rows * 8 float4 / 256 threads, W + 1 <= 64
  f32x4 rh[NPMAX], rw[CT];
  auto fetch_halo = [&](int cc) {
#pragma unroll
    for (int p = 0; p < NPMAX; ++p) {
      const int row = (tid >> 3) + 32 * p;
      const int64_t q = p0 - (W + 1) + row;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p < NP && row < NR && q >= 0 && q < a.npix) v = *(const f32x4*)(a.in + q * a.cin + cc * 32 + c4 * 4);
      rh[p] = v;
    }
  };
  auto stash_halo = [&]() {
#pragma unroll
    for (int p = 0; p < NPMAX; ++p) {
      const int row = (tid >> 3) + 32 * p;
      if (p < NP && row < NR) *(f32x4*)(s_halo + row * RS + c4 * 4) = rh[p];
    }
  };
  auto fetch_w = [&](int t) {
    const float* wt = wblk + (int64_t)t * a.cout_p * 32;
#pragma unroll
    for (int i = 0; i < CT; ++i) rw[i] = *(const f32x4*)(wt + (tid + 256 * i) * 4);
  };
  auto stash_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const int idx = tid + 256 * i;
      *(f32x4*)(s_w + buf * W_F + (idx >> 3) * RS + (idx & 7) * 4) = rw[i];
    }
  };

  fetch_halo(0);
  fetch_w(0);  // tap 0 of chunk 0
  int step = 0;
  for (int cc = 0; cc < ncc; ++cc) {
    __syncthreads();  // every read of the previous halo tile is done
    stash_halo();
    for (int tap = 0; tap < 9; ++tap, ++step) {
      const int buf = step & 1;
      stash_w(buf);  // this stage was last read two steps ago, behind the previous step's barrier
      __syncthreads();
      if (tap < 8) {
        fetch_w((tap + 1) * ncc + cc);
      } else if (cc + 1 < ncc) {
        fetch_w(cc + 1);
        fetch_halo(cc + 1);
      }
      const float* sp = s_halo + (wave * 32 + l31 + (tap / 3) * W + (tap % 3)) * RS + 4 * h;
      const float* sw = s_w + buf * W_F + l31 * RS + 4 * h;
      const bool inside = (tapmask >> tap) & 1u;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 b = *(const f32x4*)(sp + 8 * g);
        if (!inside) b = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 w[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) w[ct] = *(const f32x4*)(sw + ct * 32 * RS + 8 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ct][j], b[j], acc[ct], 0, 0, 0);
      }
    }
  }
  cs_mfma_drain();
  conv_store<CT>(a, acc, gp, blockIdx.y * (CT * 32), h);
}

// Split-bf16 variant of cs_conv3x3_halo_kernel: every f32 operand is split exactly into three bf16
// terms and a product is accumulated in f32 from the six leading term products on
// v_mfma_f32_32x32x16_bf16 (float32-equivalent results at 6/16 of the matrix time; the arithmetic
// of cs_mlp_split_kernel.inc).
//
// A block of NW wavefronts covers 32*PT*NW consecutive output pixels x 32*CT output channels; a
// wavefront owns PT pixel tiles x CT channel tiles (PT*CT accumulators of 16 registers).
//   * Weights: split on the host into bricks [channel tile][k-step 2][term 3][lane 64] x 16 bytes, the
//     order the lanes consume them, so a tile of a (tap, 32-channel chunk) step is W_B contiguous bytes and
//     a fragment read is base + constant.  The tiles stream through a three-slot LDS ring by LDS-DMA
//     (no registers, no LDS stores): while slot p is read, the copies into slot p+1 are issued in the first
//     groups of the step; every wave waits for its own copies in front of the step's single barrier, which
//     sits in the MIDDLE of the step (the previous contents of slot p+1 were last read in step p-2, which
//     every wave had left before it passed the barrier of step p-1).  Step boundaries carry no barrier.
//   * Activations: the window of 32*PT*NW + 2(W+1) input pixels that the 9 taps of the block's outputs read is
//     split into its three terms as it is staged (once per 32-channel chunk) into rows of [term][32 k] bf16 =
//     192 bytes; a lane's 8 consecutive k of k-step s are the 16-byte slot 2s + (lane >> 5), stored at slot
//     ^ ((row >> 2) & 3): any 16 rows that are distinct modulo 16 fall into 16 different bank groups.  A tap
//     whose input pixel lies outside the image reads a row of zeros instead (one select on the address).
//   * Fragments are read one group (6*PT matrix instructions) ahead of their use, across step boundaries too;
//     scheduling barriers pin that order.
typedef unsigned u32x4n __attribute__((ext_vector_type(4)));
typedef unsigned u32x2n __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8n __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2n __attribute__((ext_vector_type(2)));
typedef float f32x2n __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef __attribute__((address_space(3))) u32x4n lds_u32x4n;
__device__ __forceinline__ unsigned conv_cvt_pk(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2n){lo, hi}, bf16x2n));
}
#define CS_CONV_MFMA32(A_, B_, C_) \
  C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8n, A_), __builtin_bit_cast(bf16x8n, B_), C_, 0, 0, 0)
// -DCS_CONV_STAMPS (debug builds only, see tools/conv_stamps.py): the 100 MHz clock of wave 0 of every workgroup at the
// phase boundaries of the halo kernel (entry, window and first weight tile staged, last matrix instruction issued,
// stores issued), plus the shader clock over the loop
#ifdef CS_CONV_STAMPS
__device__ unsigned long long cs_conv_stamps[8 * 8192];
#define CS_CONV_STAMP(i_, v_)                                                                     \
  if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 8192) cs_conv_stamps[blockIdx.x * 8 + (i_)] = (v_)
#else
#define CS_CONV_STAMP(i_, v_)
#endif
constexpr int CS_HALO_NPMAX = 10;  // staging passes: (32*PT*NW + 2*64) rows * 8 threads / (64*NW) threads
// ---- fp16 split (TERMS = 2) ----
// Every f32 operand as two fp16 terms, three products a0*b0 + a0*b1 + a1*b0 on v_mfma_f32_32x32x16_f16 (half the matrix
// work of the three-term bf16 split; the neglected term is below 2^-22 |a||b|).  fp16 has 5 exponent bits, so the
// operands are normalised by exact powers of two: the weights of a layer by the one that brings the largest into
// [2^14, 2^15) (host), the activations of an IMAGE by the one that does so for its largest magnitude (in_amax: all
// taps of an output pixel must share a scale, and a per-image one leaves every image's result independent of the
// other images of the launch); the accumulators come out scaled by both and are scaled back, exactly, before bias,
// residual and ReLU.  An operand is represented to 2^-22 of itself or 2^-39 of its image's largest, whichever is
// coarser.
typedef _Float16 f16x8n __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2n __attribute__((ext_vector_type(2)));
#define CS_CONV_MFMA32H(A_, B_, C_) \
  C_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8n, A_), __builtin_bit_cast(f16x8n, B_), C_, 0, 0, 0)
__device__ __forceinline__ unsigned conv_mix_pk_f16(float x, float y, float g) {  // (fp16(x g), fp16(y g)), g a power of two
  unsigned h;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]\n\tv_fma_mixhi_f16 %0, %3, %2, 0 op_sel_hi:[0,0,0]"
      : "=&v"(h)
      : "v"(x), "v"(g), "v"(y));
  return h;
}
__device__ __forceinline__ float conv_mix_rem_lo(float x, float g, unsigned h) {  // x g - (low half of h), exact
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(g), "v"(h));
  return r;
}
__device__ __forceinline__ float conv_mix_rem_hi(float x, float g, unsigned h) {
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(g), "v"(h));
  return r;
}
__device__ __forceinline__ unsigned conv_cvt_pk_f16(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2n){lo, hi}, f16x2n));
}
// 2^-e for the image maximum whose bits are `bits`: m 2^-e in [2^14, 2^15); an image of zeros (or a maximum outside
// 2^-100 .. 2^100, clamped) keeps a finite scale
__device__ __forceinline__ float conv_image_scale(unsigned bits) {
  int be = (int)((bits & 0x7fffffffu) >> 23);
  if ((bits & 0x7fffffffu) == 0u) return 1.0f;
  be = be < 27 ? 27 : (be > 227 ? 227 : be);
  return __uint_as_float((unsigned)(127 - (be - 127 - 14)) << 23);
}
template <int CT, int PT, int NW, bool PHASE, int TERMS = 3>
__global__ void __launch_bounds__(64 * NW, 2) cs_conv3x3_halo_split_kernel(const CsConvArgs a, int NR, int NP) {
  // PHASE: phase (pa, pb) = (blockIdx.z >> 1, blockIdx.z & 1) of a nearest-neighbour-upsampled 3x3 convolution, a 2x2
  // convolution of the input at its own resolution (see CsUpsampleMode): tap (dh, dw) reads the input pixel that tap
  // (pa + dh, pb + dw) of a same-size 3x3 convolution reads, so the window, the tap masks and the fragment addresses
  // are those of the 3x3 case; the output pixel is (2i + pa, 2j + pb) of the upsampled grid.
  constexpr int NT = PHASE ? 4 : 9;
  CS_CONV_STAMP(0, __builtin_amdgcn_s_memrealtime());
  const int pa = PHASE ? (int)(blockIdx.z >> 1) : 0, pb = PHASE ? (int)(blockIdx.z & 1) : 0;
  constexpr int NTHR = 64 * NW, RPP = NTHR / 8;  // threads; window rows staged per pass
  constexpr int PX = 32 * PT;                    // pixels per wavefront
  // window rows: [term][32 k] of 2-byte entries.  TERMS = 3: 192-byte rows, 16-byte slot (k-step s, half h) of a term at
  // (2s + h) ^ ((row >> 2) & 3), term t 64 t bytes on.  TERMS = 2: 128-byte rows of eight slots, slot
  // (4t + 2s + h) ^ ((row >> 1) & 7): 16 consecutive rows (and the lane groups of a 16-byte read) cover the 16
  // positions of the 256-byte bank period once.
  constexpr int ROWB = 64 * TERMS, W_B = CT * 2048 * TERMS;
  constexpr int NPC = (W_B / 16 + NTHR - 1) / NTHR;  // 16-byte copies of a weight tile per thread
  constexpr int G = 2 * CT, GB = G / 2;              // groups (k-step, channel tile) of a step; barrier group
  // weight ring: three slots and copies one step ahead, or (fp16 split: a step is half as long, shorter than a copy's
  // round trip to L2) four slots and copies two steps ahead, waited for with a counted vmcnt
  constexpr int NSLOT = TERMS == 2 ? 4 : 3, LEAD = NSLOT - 2;
  extern __shared__ __attribute__((aligned(1024))) unsigned char s_raw[];
  lds_byte* s_w = (lds_byte*)s_raw;      // [NSLOT][W_B]
  lds_byte* s_halo = s_w + NSLOT * W_B;      // [NR][ROWB], then 256 bytes of zeros, then (TERMS = 2) the rows' scales
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = (int)cs_scalar_u32((unsigned)tid >> 6);
  const int h = lane >> 5, l31 = lane & 31, c4 = tid & 7;
  const int W = a.win, H = a.hin;
  const int ncc = a.cin >> 5;
  const int64_t p0 = (int64_t)blockIdx.x * (PX * NW);
  const unsigned char* wblk =
      (TERMS == 2 ? (const unsigned char*)(a.w3h + (PHASE ? (int64_t)blockIdx.z * a.w3h_phase_stride : 0))
                  : (const unsigned char*)(a.w3 + (PHASE ? (int64_t)blockIdx.z * a.w3_phase_stride : 0))) +
      (int64_t)blockIdx.y * W_B;
  const unsigned zero_off = (unsigned)NR * ROWB;
  __attribute__((address_space(3))) float* s_g = (__attribute__((address_space(3))) float*)(s_halo + zero_off + 256);  // [NR]
  __attribute__((address_space(3))) unsigned* s_amax = (__attribute__((address_space(3))) unsigned*)((lds_byte*)s_raw + a.amax_lds);
  if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);  // evens out issue arbitration between the two waves of a SIMD

  unsigned tapmask[PT];
  int row0[PT];
  f32x16 acc[PT][CT];

  f32x4 rh[CS_HALO_NPMAX];
  auto fetch_halo = [&](int cc) {
#pragma unroll
    for (int p = 0; p < CS_HALO_NPMAX; ++p) {
      const int row = (tid >> 3) + RPP * p;
      const int64_t q = p0 - (W + 1) + row;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p < NP && row < NR && q >= 0 && q < a.npix) v = *(const f32x4*)(a.in + q * a.cin + cc * 32 + c4 * 4);
      rh[p] = v;
    }
  };
  auto stash_halo = [&]() {
#pragma unroll
    for (int p = 0; p < CS_HALO_NPMAX; ++p) {
      const int row = (tid >> 3) + RPP * p;
      if (TERMS == 2) {
        if (p < NP && row < NR) {
          // 4 consecutive k of one row -> 2 terms x 4 fp16 of the values times the row's power of two
          const f32x4 v = rh[p];
          const float g = s_g[row];
          const unsigned a0 = conv_mix_pk_f16(v.x, v.y, g), b0 = conv_mix_pk_f16(v.z, v.w, g);
          const float r0 = conv_mix_rem_lo(v.x, g, a0), r1 = conv_mix_rem_hi(v.y, g, a0);
          const float r2 = conv_mix_rem_lo(v.z, g, b0), r3 = conv_mix_rem_hi(v.w, g, b0);
          const unsigned a1 = conv_cvt_pk_f16(r0, r1), b1 = conv_cvt_pk_f16(r2, r3);
          const unsigned o = (unsigned)row * ROWB + ((unsigned)((c4 >> 1) ^ ((row >> 1) & 7)) << 4) + ((unsigned)(c4 & 1) << 3);
          *(__attribute__((address_space(3))) u32x2n*)(s_halo + o) = u32x2n{a0, b0};
          *(__attribute__((address_space(3))) u32x2n*)(s_halo + (o ^ 64u)) = u32x2n{a1, b1};
        }
      } else if (p < NP && row < NR) {
        // 4 consecutive k of one row -> 3 terms x 4 bf16
        const f32x4 v = rh[p];
        const unsigned a0 = conv_cvt_pk(v.x, v.y), b0 = conv_cvt_pk(v.z, v.w);
        const float r0 = v.x - __uint_as_float(a0 << 16), r1 = v.y - __uint_as_float(a0 & 0xffff0000u);
        const float r2 = v.z - __uint_as_float(b0 << 16), r3 = v.w - __uint_as_float(b0 & 0xffff0000u);
        const unsigned a1 = conv_cvt_pk(r0, r1), b1 = conv_cvt_pk(r2, r3);
        const unsigned a2 = conv_cvt_pk(r0 - __uint_as_float(a1 << 16), r1 - __uint_as_float(a1 & 0xffff0000u));
        const unsigned b2 = conv_cvt_pk(r2 - __uint_as_float(b1 << 16), r3 - __uint_as_float(b1 & 0xffff0000u));
        lds_byte* dst = s_halo + row * ROWB + (((c4 >> 1) ^ ((row >> 2) & 3)) << 4) + ((c4 & 1) << 3);
        *(__attribute__((address_space(3))) u32x2n*)(dst) = u32x2n{a0, b0};
        *(__attribute__((address_space(3))) u32x2n*)(dst + 64) = u32x2n{a1, b1};
        *(__attribute__((address_space(3))) u32x2n*)(dst + 128) = u32x2n{a2, b2};
      }
    }
  };
  // LDS-DMA of weight tile t into ring slot `slot`: thread copy r moves 16 bytes at r*NTHR*16 + tid*16
  const unsigned dma_off = (unsigned)tid * 16u;
  const unsigned ring_lds = cs_scalar_u32((unsigned)(size_t)s_w) + (unsigned)wave * 1024u;
  auto dma_copy = [&](int t, int slot, int r) {
    if (r * NTHR * 16 + wave * 1024 < W_B) {
      const unsigned long long src = (unsigned long long)(wblk + (int64_t)t * a.cout_p * ROWB + (int64_t)r * NTHR * 16);
      const void* ssrc =
          (const void*)(((unsigned long long)cs_scalar_u32((unsigned)(src >> 32)) << 32) | cs_scalar_u32((unsigned)src));
      cs_dma16(ssrc, dma_off, ring_lds + (unsigned)(slot * W_B + r * NTHR * 16));
    }
  };
  // activation fragment address of pixel tile p for tap `tap`, k-step 0 (k-step 1: ^ 32), term 0 (b_term)
  auto b_addr = [&](int tap, int p) -> unsigned {
    const int kh = PHASE ? pa + (tap >> 1) : tap / 3, kw = PHASE ? pb + (tap & 1) : tap % 3;  // position in the 3x3 window
    const int rowB = row0[p] + kh * W + kw;
    const unsigned o = (unsigned)rowB * ROWB + (unsigned)((h ^ (TERMS == 2 ? (rowB >> 1) & 7 : (rowB >> 2) & 3)) << 4);
    return ((tapmask[p] >> (kh * 3 + kw)) & 1u) ? o : zero_off + (unsigned)(h << 4);
  };
  auto b_term = [&](unsigned o, int tm) -> unsigned { return TERMS == 2 ? o ^ ((unsigned)tm << 6) : o + (unsigned)tm * 64u; };

  // the window and the first weight tile are requested first: the tap masks are computed under their latency
  fetch_halo(0);
#pragma unroll
  for (int r = 0; r < NPC; ++r) dma_copy(0 * ncc + 0, 0, r);
  if (LEAD == 2) {
#pragma unroll
    for (int r = 0; r < NPC; ++r) dma_copy(1 * ncc + 0, 1, r);  // step 1 = tap 1 of chunk 0
  }
  if (tid < 16) *(lds_u32x4n*)(s_halo + zero_off + tid * 16) = u32x4n{0u, 0u, 0u, 0u};
  const int hw = H * W;
  // position of the block's first pixel inside its image: one 64-bit division per workgroup (uniform), 32-bit
  // arithmetic per lane (a block spans fewer than 2^10 pixels)
  const int rem0 = (int)cs_scalar_u32((unsigned)(p0 % hw));
  const int64_t img0 = (p0 - rem0) / hw;  // image of the block's first pixel (uniform)
  if (TERMS == 2) {
    // the power of two of every window row's image; rows outside the launch hold zeros under any scale.  Window row r
    // is pixel rem0 - (W + 1) + r counted from the first pixel of image img0 (W + 1 <= 2 hw: at most two images back).
    for (int r = tid; r < NR; r += NTHR) {
      const int64_t q = p0 - (W + 1) + r;
      const int rel = (int)((unsigned)(rem0 - (W + 1) + r + 2 * hw) / (unsigned)hw) - 2;
      s_g[r] = (q >= 0 && q < a.npix) ? conv_image_scale(a.in_amax[img0 + rel]) : 1.0f;
    }
    if (a.out_amax)
      for (int r = tid; r < (PX * NW) / hw + 2; r += NTHR) s_amax[r] = 0u;
  }
#pragma unroll
  for (int p = 0; p < PT; ++p) {
    row0[p] = wave * PX + p * 32 + l31;
    const int64_t gp = p0 + row0[p];
    const int rem = (int)((unsigned)(rem0 + row0[p]) % (unsigned)hw);
    const int oh = rem / W, ow = rem - oh * W;
    tapmask[p] = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ih = oh + t / 3 - 1, iw = ow + t % 3 - 1;
      if (gp < a.npix && ih >= 0 && ih < H && iw >= 0 && iw < W) tapmask[p] |= 1u << t;
    }
  }

#pragma unroll
  for (int p = 0; p < PT; ++p)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][ct][r] = 0.0f;
  if (TERMS == 2) __syncthreads();  // the rows' scales
  stash_halo();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  CS_CONV_STAMP(1, __builtin_amdgcn_s_memrealtime());
  CS_CONV_STAMP(4, __builtin_amdgcn_s_memtime());

  u32x4n wr[2][TERMS];
  u32x4n br[2][PT][TERMS];
  unsigned bo[PT], bn[PT];
  const lds_byte* wl = s_w + lane * 16;
  for (int cc = 0; cc < ncc; ++cc) {
    // ring slot of the chunk's first step: 9 taps return to slot 0 with every chunk, 4 taps do not
    const int sb = (NT % NSLOT) ? (int)cs_scalar_u32((unsigned)((cc * NT) % NSLOT)) : 0;
    // first fragments of the chunk: group 0 of tap 0
#pragma unroll
    for (int tm = 0; tm < TERMS; ++tm) wr[0][tm] = *(const lds_u32x4n*)(wl + sb * W_B + tm * 1024);
#pragma unroll
    for (int p = 0; p < PT; ++p) {
      bo[p] = b_addr(0, p);
#pragma unroll
      for (int tm = 0; tm < TERMS; ++tm) br[0][p][tm] = *(const lds_u32x4n*)(s_halo + b_term(bo[p], tm));
    }
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
      const int slot = sb + tap % NSLOT >= NSLOT ? sb + tap % NSLOT - NSLOT : sb + tap % NSLOT;
      const int nslot = slot == NSLOT - 1 ? 0 : slot + 1;
      const int cslot = LEAD == 2 ? (nslot == NSLOT - 1 ? 0 : nslot + 1) : nslot;  // slot the step's copies fill
      // the step LEAD steps on, if there is one, and its weight tile
      const bool more = tap + LEAD < NT || cc + 1 < ncc;
      const int tnext = tap + LEAD < NT ? (tap + LEAD) * ncc + cc : (tap + LEAD - NT) * ncc + cc + 1;
      if (tap < NT - 1) {
#pragma unroll
        for (int p = 0; p < PT; ++p) bn[p] = b_addr(tap + 1, p);
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int s = g / CT, ct = g % CT;
        if (g == GB) {
          // the copies of the NEXT step have landed: all of the wave's (LEAD = 1), all but the ones just issued
          if (LEAD == 2 && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPC) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          // the next chunk's window: requested behind the barrier of tap 5, so that it has a whole step before
          // the next wait for the DMA copies also waits for it
          if (tap == (PHASE ? 2 : 5) && cc + 1 < ncc) fetch_halo(cc + 1);
        }
        // fragments of the next group (the first group of the next tap behind the last one)
        if (g + 1 < G) {
          const int s2 = (g + 1) / CT, ct2 = (g + 1) % CT;
#pragma unroll
          for (int tm = 0; tm < TERMS; ++tm)
            wr[(g + 1) & 1][tm] = *(const lds_u32x4n*)(wl + slot * W_B + ((ct2 * 2 + s2) * TERMS + tm) * 1024);
        } else if (tap < NT - 1) {
#pragma unroll
          for (int tm = 0; tm < TERMS; ++tm) wr[0][tm] = *(const lds_u32x4n*)(wl + nslot * W_B + tm * 1024);
        }
        if (ct == CT - 1) {
          if (s == 0) {
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
              for (int tm = 0; tm < TERMS; ++tm) br[1][p][tm] = *(const lds_u32x4n*)(s_halo + b_term(bo[p] ^ 32u, tm));
          } else if (tap < NT - 1) {
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
              for (int tm = 0; tm < TERMS; ++tm) br[0][p][tm] = *(const lds_u32x4n*)(s_halo + b_term(bn[p], tm));
          }
        }
        const u32x4n(&w)[TERMS] = wr[g & 1];
        if constexpr (TERMS == 2) {
#pragma unroll
          for (int p = 0; p < PT; ++p) {
            CS_CONV_MFMA32H(w[0], br[s][p][0], acc[p][ct]);
            CS_CONV_MFMA32H(w[0], br[s][p][1], acc[p][ct]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (more && g < GB) {
#pragma unroll
            for (int r = 0; r < NPC; ++r)
              if ((r < GB ? r : GB - 1) == g) dma_copy(tnext, cslot, r);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int p = 0; p < PT; ++p) CS_CONV_MFMA32H(w[1], br[s][p][0], acc[p][ct]);
          __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
          for (int p = 0; p < PT; ++p) {
            CS_CONV_MFMA32(w[0], br[s][p][0], acc[p][ct]);
            CS_CONV_MFMA32(w[0], br[s][p][1], acc[p][ct]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (more && g < GB) {
#pragma unroll
            for (int r = 0; r < NPC; ++r)
              if ((r < GB ? r : GB - 1) == g) dma_copy(tnext, cslot, r);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int p = 0; p < PT; ++p) {
            CS_CONV_MFMA32(w[1], br[s][p][0], acc[p][ct]);
            CS_CONV_MFMA32(w[0], br[s][p][TERMS - 1], acc[p][ct]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int p = 0; p < PT; ++p) {
            CS_CONV_MFMA32(w[1], br[s][p][1], acc[p][ct]);
            CS_CONV_MFMA32(w[TERMS - 1], br[s][p][0], acc[p][ct]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int p = 0; p < PT; ++p) bo[p] = bn[p];
    }
    if (cc + 1 < ncc) {
      __syncthreads();  // every read of this chunk's window is done
      stash_halo();
      __syncthreads();
    }
  }
  cs_mfma_drain();
  CS_CONV_STAMP(2, __builtin_amdgcn_s_memrealtime());
  CS_CONV_STAMP(5, __builtin_amdgcn_s_memtime());
  if constexpr (TERMS == 2) {
    // back to true values: the lane's pixel of tile p is window row row0[p] + W + 1
    const int wexp = PHASE ? a.wexp[blockIdx.z] : a.wexp[0];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
      // the image's scale is 2^-e: scale back by 2^(e - wexp) in one exact step (the two powers of two multiplied
      // into one float could leave its range)
      const int e = 127 - (int)(__float_as_uint(s_g[row0[p] + W + 1]) >> 23);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][ct][r] = __builtin_ldexpf(acc[p][ct][r], e - wexp);
    }
  }
  if ((a.cout & 31) == 0 && !a.out_nchw) {
    __syncthreads();  // the ring and the window are dead: the LDS becomes per-wave transposition space
    const bool amax = TERMS == 2 && a.out_amax != nullptr;
    conv_store_rows<CT, PT, PHASE>(a, acc, (conv_lds_byte*)s_raw + wave * (PX * (CT * 128 + 16)), p0 + wave * PX,
                                   blockIdx.y * (CT * 32), lane, pa, pb, amax ? s_amax : nullptr, rem0 + wave * PX, hw);
    if (amax) {
      // the block's table of per-image maxima joins the launch's (image of the block's first pixel: (p0 - rem0) / hw)
      __syncthreads();
      for (int r = tid; r < (PX * NW) / hw + 2; r += NTHR) {
        const unsigned v = s_amax[r];
        if (v != 0u) conv_amax_join(a.out_amax + img0 + r, v);
      }
    }
  } else {
#pragma unroll
    for (int p = 0; p < PT; ++p) conv_store<CT>(a, acc[p], p0 + row0[p], blockIdx.y * (CT * 32), h);
  }
  CS_CONV_STAMP(3, __builtin_amdgcn_s_memrealtime());
}
// 32 -> 32 channel same-size 3x3 convolutions (the 14x14 stages of the MNIST-GMM nets): 30 algorithmic GFLOP against
// 0.4-0.6 GB of activations per layer at 8192 images, bound by HBM, and a block of the kernel above is all prologue and
// epilogue (one 32-channel chunk).  Here the whole split weight set (9 taps x 6 KiB) stays in LDS and a persistent
// workgroup per CU walks over items of 32*NW pixels: the next item's window is requested before the current item's
// matrix instructions and staged behind them, the residual rows are requested before them too, the stores are never
// waited for; two barriers per item, none inside it.
// TERMS = 2: the fp16 split of cs_conv3x3_halo_split_kernel (two fp16 terms, three products, 128-byte window rows with
// its slot swizzle, weights from a.w3h, operands of image n scaled by the power of two from in_amax[n], fetched with
// the window rows).
template <int NW, int TERMS = 3>
__global__ void __launch_bounds__(64 * NW, 2) cs_conv3x3_halo_resident_kernel(const CsConvArgs a, int NR, int NP, int nitems) {
  constexpr int NTHR = 64 * NW, RPP = NTHR / 8;
  constexpr int ROWB = 64 * TERMS, W_B = 2048 * TERMS, NPMAX = 6;
  constexpr int RS = 128 + 16;  // transposition row: 32 channels + padding
  extern __shared__ __attribute__((aligned(1024))) unsigned char s_raw[];
  lds_byte* s_w = (lds_byte*)s_raw;                     // [9][W_B]
  lds_byte* s_halo = s_w + 9 * W_B;                     // [NR][ROWB], then 256 bytes of zeros
  lds_byte* s_t = s_halo + (((unsigned)NR * ROWB + 256u + 1023u) & ~1023u);  // [NW][32][RS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = (int)cs_scalar_u32((unsigned)tid >> 6);
  const int h = lane >> 5, l31 = lane & 31, c4 = tid & 7;
  const int W = a.win, H = a.hin;
  const unsigned zero_off = (unsigned)NR * ROWB;
  if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);

  f32x4 rh[NPMAX];
  unsigned rg[NPMAX];  // TERMS = 2: bits of the row's image maximum
  auto fetch_halo = [&](int item) {
    const int64_t q0 = (int64_t)item * (32 * NW) - (W + 1);
#pragma unroll
    for (int p = 0; p < NPMAX; ++p) {
      const int row = (tid >> 3) + RPP * p;
      const int64_t q = q0 + row;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      unsigned gb = 0u;
      if (p < NP && row < NR && q >= 0 && q < a.npix) {
        v = *(const f32x4*)(a.in + q * 32 + c4 * 4);
        if (TERMS == 2) gb = a.in_amax[(unsigned)q / (unsigned)(H * W)];
      }
      rh[p] = v;
      rg[p] = gb;
    }
  };
  auto stash_halo = [&]() {
#pragma unroll
    for (int p = 0; p < NPMAX; ++p) {
      const int row = (tid >> 3) + RPP * p;
      if (TERMS == 2) {
        if (p < NP && row < NR) {
          const f32x4 v = rh[p];
          const float g = conv_image_scale(rg[p]);
          const unsigned a0 = conv_mix_pk_f16(v.x, v.y, g), b0 = conv_mix_pk_f16(v.z, v.w, g);
          const float r0 = conv_mix_rem_lo(v.x, g, a0), r1 = conv_mix_rem_hi(v.y, g, a0);
          const float r2 = conv_mix_rem_lo(v.z, g, b0), r3 = conv_mix_rem_hi(v.w, g, b0);
          const unsigned a1 = conv_cvt_pk_f16(r0, r1), b1 = conv_cvt_pk_f16(r2, r3);
          const unsigned o = (unsigned)row * ROWB + ((unsigned)((c4 >> 1) ^ ((row >> 1) & 7)) << 4) + ((unsigned)(c4 & 1) << 3);
          *(__attribute__((address_space(3))) u32x2n*)(s_halo + o) = u32x2n{a0, b0};
          *(__attribute__((address_space(3))) u32x2n*)(s_halo + (o ^ 64u)) = u32x2n{a1, b1};
        }
      } else if (p < NP && row < NR) {
        const f32x4 v = rh[p];
        const unsigned a0 = conv_cvt_pk(v.x, v.y), b0 = conv_cvt_pk(v.z, v.w);
        const float r0 = v.x - __uint_as_float(a0 << 16), r1 = v.y - __uint_as_float(a0 & 0xffff0000u);
        const float r2 = v.z - __uint_as_float(b0 << 16), r3 = v.w - __uint_as_float(b0 & 0xffff0000u);
        const unsigned a1 = conv_cvt_pk(r0, r1), b1 = conv_cvt_pk(r2, r3);
        const unsigned a2 = conv_cvt_pk(r0 - __uint_as_float(a1 << 16), r1 - __uint_as_float(a1 & 0xffff0000u));
        const unsigned b2 = conv_cvt_pk(r2 - __uint_as_float(b1 << 16), r3 - __uint_as_float(b1 & 0xffff0000u));
        lds_byte* dst = s_halo + row * ROWB + (((c4 >> 1) ^ ((row >> 2) & 3)) << 4) + ((c4 & 1) << 3);
        *(__attribute__((address_space(3))) u32x2n*)(dst) = u32x2n{a0, b0};
        *(__attribute__((address_space(3))) u32x2n*)(dst + 64) = u32x2n{a1, b1};
        *(__attribute__((address_space(3))) u32x2n*)(dst + 128) = u32x2n{a2, b2};
      }
    }
  };

  // all weight tiles: tile of tap t = [t][cin/32 = 1][cout_p/32 = 1] brick; 9 * W_B / 1024 copies of 1 KiB over the waves
  {
    const unsigned ring_lds = cs_scalar_u32((unsigned)(size_t)s_w);
    for (int piece = wave; piece < 9 * W_B / 1024; piece += NW) {
      const unsigned long long src =
          (unsigned long long)((TERMS == 2 ? (const unsigned char*)a.w3h : (const unsigned char*)a.w3) + (int64_t)piece * 1024);
      const void* ssrc =
          (const void*)(((unsigned long long)cs_scalar_u32((unsigned)(src >> 32)) << 32) | cs_scalar_u32((unsigned)src));
      cs_dma16(ssrc, (unsigned)lane * 16u, ring_lds + (unsigned)piece * 1024u);
    }
  }
  if (tid < 16) *(lds_u32x4n*)(s_halo + zero_off + tid * 16) = u32x4n{0u, 0u, 0u, 0u};
  if ((int)blockIdx.x < nitems) fetch_halo(blockIdx.x);
  stash_halo();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const lds_byte* wl = s_w + lane * 16;
  // lane's part of the transposed tile: row lr (of 8 per pass), channels lc .. lc + 3
  const int lr = lane >> 3, lc = (lane & 7) * 4;
  lds_byte* tw = s_t + wave * (32 * RS);
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) bias4 = *(const f32x4*)(a.bias + lc);

  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int64_t p0 = (int64_t)item * (32 * NW);
    const int row0 = wave * 32 + l31;
    const int64_t gp = p0 + row0;
    unsigned tapmask = 0;
    {
      const int hw = H * W;
      const unsigned g = gp < a.npix ? (unsigned)gp : 0u;
      const int rem = (int)(g % (unsigned)hw);
      const int oh = rem / W, ow = rem - oh * W;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int ih = oh + t / 3 - 1, iw = ow + t % 3 - 1;
        if (gp < a.npix && ih >= 0 && ih < H && iw >= 0 && iw < W) tapmask |= 1u << t;
      }
    }
    auto b_addr = [&](int tap) -> unsigned {
      const int rowB = row0 + (tap / 3) * W + (tap % 3);
      const unsigned o = (unsigned)rowB * ROWB + (unsigned)((h ^ (TERMS == 2 ? (rowB >> 1) & 7 : (rowB >> 2) & 3)) << 4);
      return ((tapmask >> tap) & 1u) ? o : zero_off + (unsigned)(h << 4);
    };
    // requests that the matrix instructions below cover: the next item's window, this item's residual rows
    const int next = item + gridDim.x;
    // TERMS = 2: the maximum of the lane's pixel's image, for the way back to true values
    unsigned my_amax = 0u;
    if (TERMS == 2 && gp < a.npix) my_amax = a.in_amax[(unsigned)gp / (unsigned)(H * W)];
    if (next < nitems) fetch_halo(next);
    f32x4 rs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t g2 = p0 + wave * 32 + i * 8 + lr;
      rs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.res && g2 < a.npix) rs[i] = *(const f32x4*)(a.res + g2 * a.cout + lc);
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    u32x4n wr[2][TERMS], br[2][TERMS];
    unsigned bo = b_addr(0), bn = 0;
    auto b_term = [&](unsigned o, int tm) -> unsigned { return TERMS == 2 ? o ^ ((unsigned)tm << 6) : o + (unsigned)tm * 64u; };
#pragma unroll
    for (int tm = 0; tm < TERMS; ++tm) {
      wr[0][tm] = *(const lds_u32x4n*)(wl + tm * 1024);
      br[0][tm] = *(const lds_u32x4n*)(s_halo + b_term(bo, tm));
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap < 8) bn = b_addr(tap + 1);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        // fragments of the next group (the first group of the next tap behind the second one)
        if (s == 0) {
#pragma unroll
          for (int tm = 0; tm < TERMS; ++tm) {
            wr[1][tm] = *(const lds_u32x4n*)(wl + tap * W_B + (TERMS + tm) * 1024);
            br[1][tm] = *(const lds_u32x4n*)(s_halo + b_term(bo ^ 32u, tm));
          }
        } else if (tap < 8) {
#pragma unroll
          for (int tm = 0; tm < TERMS; ++tm) {
            wr[0][tm] = *(const lds_u32x4n*)(wl + (tap + 1) * W_B + tm * 1024);
            br[0][tm] = *(const lds_u32x4n*)(s_halo + b_term(bn, tm));
          }
        }
        const u32x4n(&w)[TERMS] = wr[s];
        const u32x4n(&b)[TERMS] = br[s];
        if constexpr (TERMS == 2) {
          CS_CONV_MFMA32H(w[0], b[0], acc);
          CS_CONV_MFMA32H(w[0], b[1], acc);
          __builtin_amdgcn_sched_barrier(0);
          CS_CONV_MFMA32H(w[1], b[0], acc);
          __builtin_amdgcn_sched_barrier(0);
        } else {
          CS_CONV_MFMA32(w[0], b[0], acc);
          CS_CONV_MFMA32(w[0], b[1], acc);
          __builtin_amdgcn_sched_barrier(0);
          CS_CONV_MFMA32(w[1], b[0], acc);
          CS_CONV_MFMA32(w[0], b[TERMS - 1], acc);
          __builtin_amdgcn_sched_barrier(0);
          CS_CONV_MFMA32(w[1], b[1], acc);
          CS_CONV_MFMA32(w[TERMS - 1], b[0], acc);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      bo = bn;
    }
    cs_mfma_drain();
    if constexpr (TERMS == 2) {  // back to true values, in one exact step
      const int e = 127 - (int)(__float_as_uint(conv_image_scale(my_amax)) >> 23) - a.wexp[0];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = __builtin_ldexpf(acc[r], e);
    }
    // epilogue: transpose through the wave's LDS region so that a store covers 8 whole pixel rows
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *(conv_lds_f32x4*)(tw + l31 * RS + (8 * q + 4 * h) * 4) = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
    // out_amax (images of at least 32 pixels: the wave's 32 rows lie in at most two images): the rows in front of the
    // image boundary `bnd` belong to image img_a, the others to the next one
    const unsigned first = (unsigned)(p0 + wave * 32);
    const int img_a = (int)(first / (unsigned)(H * W));
    const int bnd = (img_a + 1) * (H * W) - (int)first;
    float m_a = 0.0f, m_b = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i * 8 + lr;
      const int64_t g2 = p0 + wave * 32 + row;
      f32x4 o = *(const conv_lds_f32x4*)(tw + row * RS + lc * 4) + bias4 + rs[i];
      if (a.relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], 0.0f);
      }
      if (g2 < a.npix) *(f32x4*)(a.out + g2 * a.cout + lc) = o;
      if (a.out_amax && g2 < a.npix) {
        const float m = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
        if (row < bnd) m_a = fmaxf(m_a, m);
        else m_b = fmaxf(m_b, m);
      }
    }
    if (a.out_amax) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        m_a = fmaxf(m_a, __shfl_xor(m_a, o));
        m_b = fmaxf(m_b, __shfl_xor(m_b, o));
      }
      if (lane == 0 && m_a > 0.0f) conv_amax_join(a.out_amax + img_a, __float_as_uint(m_a));
      if (lane == 1 && m_b > 0.0f) conv_amax_join(a.out_amax + img_a + 1, __float_as_uint(m_b));
    }
    if (next < nitems) {
      // barriers without a wait for the stores above (only LDS traffic has to be settled)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every read of this item's window is done
      stash_halo();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
}

#undef CS_CONV_MFMA32

// Split-bf16 variant of cs_conv_mfma_kernel (any kernel size / stride / upsampling, linear layers):
// the activation tile of a chunk is split into its three bf16 terms as it is staged, the weight tile
// comes pre-split in the brick order of cs_conv3x3_halo_split_kernel; the activation rows have its row layout and
// slot swizzle.
// TERMS = 2: the fp16 split of the halo kernel (two fp16 terms, three products, operands of image n scaled by the power
// of two from in_amax[n]: a source pixel belongs to its output pixel's image; 128-byte rows with that kernel's slot
// swizzle; weights from a.w3h with exponent a.wexp[0]).
template <int CT, int TERMS = 3>
__global__ void __launch_bounds__(256) cs_conv_mfma_split_kernel(const CsConvArgs a) {
  constexpr int ROWB = 64 * TERMS, W_B = CT * 32 * ROWB;
  constexpr int NPC = (W_B / 16 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  unsigned char* s_pix = s_raw;             // [128][ROWB]
  unsigned char* s_w = s_raw + 128 * ROWB;  // [CT*32][ROWB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int c4 = tid & 7;
  const int ncc = (a.cin + 31) >> 5;
  const bool cin_vec = (a.cin & 31) == 0;
  const int nchunks = a.ksize * a.ksize * ncc;
  const int hup = a.hin << a.us, wup = a.win << a.us;

  int ih0[4], iw0[4];
  int64_t base[4];
  bool pv[4];
  float gsc[4];  // TERMS = 2: the power of two of the row's image
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int64_t gp = (int64_t)blockIdx.x * 128 + (tid >> 3) + 32 * p;
    pv[p] = gp < a.npix;
    const int64_t g = pv[p] ? gp : 0;
    const int hw = a.hout * a.wout;
    const int64_t n = g / hw;
    const int rem = (int)(g - n * hw);
    const int oh = rem / a.wout, ow = rem - oh * a.wout;
    ih0[p] = oh * a.stride - a.pad_h;
    iw0[p] = ow * a.stride - a.pad_w;
    base[p] = n * a.hin * a.win;
    gsc[p] = (TERMS == 2 && pv[p]) ? conv_image_scale(a.in_amax[n]) : 1.0f;
  }
  const unsigned char* wblk = (TERMS == 2 ? (const unsigned char*)a.w3h : (const unsigned char*)a.w3) + (int64_t)blockIdx.y * W_B;

  f32x16 acc[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.0f;
  f32x4 rp[4];
  u32x4n rw[NPC];
  // chunks run tap by tap, the 32-channel chunks of a tap consecutively: the four source rows of this thread are
  // located once per tap (f_tap, f_cc: the next chunk to fetch), then advance by 32 channels
  int f_tap = 0, f_cc = 0;
  const float* srcp[4];
  bool inb[4];
  auto fetch = [&](int t) {
    if (f_cc == 0) {
      const int kh = f_tap / a.ksize, kw = f_tap - kh * a.ksize;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int uh = ih0[p] + kh, uw = iw0[p] + kw;
        inb[p] = pv[p] && uh >= 0 && uh < hup && uw >= 0 && uw < wup;
        srcp[p] = a.in + ((base[p] + (int64_t)(uh >> a.us) * a.win + (uw >> a.us)) * a.cin + c4 * 4);
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (inb[p]) {
        const float* src = srcp[p] + f_cc * 32;
        if (cin_vec) {
          v = *(const f32x4*)src;
        } else {
          const int c0 = f_cc * 32 + c4 * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (c0 + r < a.cin) v[r] = src[r];
        }
      }
      rp[p] = v;
    }
    if (++f_cc == ncc) {
      f_cc = 0;
      ++f_tap;
    }
    const unsigned char* wt = wblk + (int64_t)t * a.cout_p * ROWB;
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int idx = tid + 256 * i;
      if (idx < W_B / 16) rw[i] = *(const u32x4n*)(wt + idx * 16);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = (tid >> 3) + 32 * p;
      const f32x4 v = rp[p];
      if constexpr (TERMS == 2) {
        const float g = gsc[p];
        const unsigned a0 = conv_mix_pk_f16(v.x, v.y, g), b0 = conv_mix_pk_f16(v.z, v.w, g);
        const float r0 = conv_mix_rem_lo(v.x, g, a0), r1 = conv_mix_rem_hi(v.y, g, a0);
        const float r2 = conv_mix_rem_lo(v.z, g, b0), r3 = conv_mix_rem_hi(v.w, g, b0);
        const unsigned a1 = conv_cvt_pk_f16(r0, r1), b1 = conv_cvt_pk_f16(r2, r3);
        const unsigned o = (unsigned)row * ROWB + ((unsigned)((c4 >> 1) ^ ((row >> 1) & 7)) << 4) + ((unsigned)(c4 & 1) << 3);
        *(u32x2n*)(s_pix + o) = u32x2n{a0, b0};
        *(u32x2n*)(s_pix + (o ^ 64u)) = u32x2n{a1, b1};
        continue;
      }
      const unsigned a0 = conv_cvt_pk(v.x, v.y), b0 = conv_cvt_pk(v.z, v.w);
      const float r0 = v.x - __uint_as_float(a0 << 16), r1 = v.y - __uint_as_float(a0 & 0xffff0000u);
      const float r2 = v.z - __uint_as_float(b0 << 16), r3 = v.w - __uint_as_float(b0 & 0xffff0000u);
      const unsigned a1 = conv_cvt_pk(r0, r1), b1 = conv_cvt_pk(r2, r3);
      const unsigned a2 = conv_cvt_pk(r0 - __uint_as_float(a1 << 16), r1 - __uint_as_float(a1 & 0xffff0000u));
      const unsigned b2 = conv_cvt_pk(r2 - __uint_as_float(b1 << 16), r3 - __uint_as_float(b1 & 0xffff0000u));
      unsigned char* dst = s_pix + row * ROWB + (((c4 >> 1) ^ ((row >> 2) & 3)) << 4) + ((c4 & 1) << 3);
      *(u32x2n*)(dst) = u32x2n{a0, b0};
      *(u32x2n*)(dst + 64) = u32x2n{a1, b1};
      *(u32x2n*)(dst + 128) = u32x2n{a2, b2};
    }
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int idx = tid + 256 * i;
      if (idx < W_B / 16) *(u32x4n*)(s_w + idx * 16) = rw[i];
    }
  };

  fetch(0);
  const int rowB = wave * 32 + l31;
  const unsigned char* bp = s_pix + rowB * ROWB;
  const int swz = TERMS == 2 ? (l31 >> 1) & 7 : (l31 >> 2) & 3;  // rowB and the weight rows differ from l31 by multiples of 32
  const unsigned char* ap = s_w + lane * 16;  // weight bricks: [channel tile][k-step][term][lane] x 16 bytes
  for (int t = 0; t < nchunks; ++t) {
    __syncthreads();  // previous chunk's reads are done
    stash();
    __syncthreads();
    if (t + 1 < nchunks) fetch(t + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int off = ((2 * s + h) ^ swz) << 4;
      u32x4n b[TERMS];
#pragma unroll
      for (int tm = 0; tm < TERMS; ++tm) b[tm] = *(const u32x4n*)(bp + (TERMS == 2 ? off ^ (tm << 6) : tm * 64 + off));
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        u32x4n w[TERMS];
#pragma unroll
        for (int tm = 0; tm < TERMS; ++tm) w[tm] = *(const u32x4n*)(ap + ((ct * 2 + s) * TERMS + tm) * 1024);
        if constexpr (TERMS == 2) {
          CS_CONV_MFMA32H(w[0], b[0], acc[ct]);
          CS_CONV_MFMA32H(w[0], b[1], acc[ct]);
          CS_CONV_MFMA32H(w[1], b[0], acc[ct]);
          continue;
        }
#define CS_CONV_MFMA(A_, B_) \
  acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8n, A_), __builtin_bit_cast(bf16x8n, B_), acc[ct], 0, 0, 0)
        CS_CONV_MFMA(w[0], b[0]);
        CS_CONV_MFMA(w[0], b[1]);
        CS_CONV_MFMA(w[1], b[0]);
        CS_CONV_MFMA(w[0], b[TERMS - 1]);
        CS_CONV_MFMA(w[1], b[1]);
        CS_CONV_MFMA(w[TERMS - 1], b[0]);
#undef CS_CONV_MFMA
      }
    }
  }
  cs_mfma_drain();
  if constexpr (TERMS == 2) {
    // back to true values, in one exact step: 2^(e - wexp), 2^-e the scale of the lane's pixel's image
    const int64_t gp = (int64_t)blockIdx.x * 128 + wave * 32 + l31;
    const float g = gp < a.npix ? conv_image_scale(a.in_amax[gp / (a.hout * a.wout)]) : 1.0f;
    const int e = 127 - (int)(__float_as_uint(g) >> 23) - a.wexp[0];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] = __builtin_ldexpf(acc[ct][r], e);
  }
  conv_store<CT>(a, acc, (int64_t)blockIdx.x * 128 + wave * 32 + l31, blockIdx.y * (CT * 32), h);
}

// V output channels per thread (channel fastest): any channel counts, either memory order.  V = 4
// (cout % 4 == 0, NHWC output) stores 16 bytes per thread: the layers that land here are bound by
// their output writes.
template <int V>
__global__ void __launch_bounds__(256) cs_conv_direct_kernel(const CsConvArgs a) {
  // small weight sets (the first convolution: 64 x 1 x 3 x 3) are staged in LDS as [tap][ci][cout],
  // so a thread's V output channels come from one 16-byte LDS read per (tap, ci)
  constexpr int LDS_W = 4096;
  __shared__ __attribute__((aligned(16))) float s_w[LDS_W];
  const int kk = a.ksize * a.ksize;
  const bool staged = V == 4 && a.cout * a.cin * kk <= LDS_W;
  if (staged) {
    for (int i = threadIdx.x; i < a.cout * a.cin * kk; i += 256) {
      const int co = i / (a.cin * kk), r = i - co * (a.cin * kk);  // r = ci*kk + tap
      const int ci = r / kk, tap = r - ci * kk;
      s_w[(tap * a.cin + ci) * a.cout + co] = a.w[i];
    }
    __syncthreads();
  }
  const int cv = a.cout / V;
  // grid-stride: the staged weights serve many outputs
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < a.npix * cv; idx += (int64_t)gridDim.x * 256) {
  const int hw = a.hout * a.wout;
  int co;
  int64_t gp, n;
  int rem;
  if (a.npix * cv < ((int64_t)1 << 31)) {  // 32-bit divisions: a 64-bit one costs about as much as the whole tap loop
    const unsigned i32 = (unsigned)idx, g32 = i32 / (unsigned)cv, n32 = g32 / (unsigned)hw;
    co = (int)(i32 - g32 * (unsigned)cv) * V;
    gp = g32;
    n = n32;
    rem = (int)(g32 - n32 * (unsigned)hw);
  } else {
    co = (int)(idx % cv) * V;
    gp = idx / cv;
    n = gp / hw;
    rem = (int)(gp - n * hw);
  }
  const int oh = rem / a.wout, ow = rem - oh * a.wout;
  const int hup = a.hin << a.us, wup = a.win << a.us;
  const int64_t wstride = (int64_t)a.cin * kk;  // between output channels
  float s[V];
#pragma unroll
  for (int v = 0; v < V; ++v) s[v] = 0.0f;
  for (int kh = 0; kh < a.ksize; ++kh) {
    const int uh = oh * a.stride - a.pad_h + kh;
    if (uh < 0 || uh >= hup) continue;
    for (int kw = 0; kw < a.ksize; ++kw) {
      const int uw = ow * a.stride - a.pad_w + kw;
      if (uw < 0 || uw >= wup) continue;
      const int sh = uh >> a.us, sw = uw >> a.us;
      const int tap = kh * a.ksize + kw;
      const float* wp = a.w + (int64_t)co * wstride + tap;
      const float* ip = a.in_nchw ? a.in + (n * a.cin * a.hin + sh) * a.win + sw : a.in + ((n * a.hin + sh) * a.win + sw) * a.cin;
      const int64_t istride = a.in_nchw ? (int64_t)a.hin * a.win : 1;
      for (int ci = 0; ci < a.cin; ++ci) {
        const float x = ip[ci * istride];
        if (V == 4 && staged) {
          const f32x4 w4 = *(const f32x4*)(s_w + (tap * a.cin + ci) * a.cout + co);
#pragma unroll
          for (int v = 0; v < V; ++v) s[v] = fmaf(x, w4[v & 3], s[v]);
        } else {
#pragma unroll
          for (int v = 0; v < V; ++v) s[v] = fmaf(x, wp[v * wstride + (int64_t)ci * kk], s[v]);
        }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < V; ++v) {
    if (a.bias) s[v] += a.bias[co + v];
    if (a.res) s[v] += a.res[gp * a.cout + co + v];
    if (a.relu) s[v] = fmaxf(s[v], 0.0f);
  }
  if (V == 4) {
    const f32x4 o = {s[0], s[V > 1 ? 1 : 0], s[V > 2 ? 2 : 0], s[V > 3 ? 3 : 0]};
    *(f32x4*)(a.out + gp * a.cout + co) = o;
  } else if (a.out_nchw) {
    a.out[(n * a.cout + co) * hw + rem] = s[0];
  } else {
    a.out[gp * a.cout + co] = s[0];
  }
  }
}

// Convolutions with at most 4 output channels over NHWC inputs of 32k channels (the decoders' last layer): bound by
// reading the input once, and the matrix kernels would pad the outputs to 32.  Plain f32 FMAs (exact float32
// arithmetic in either mode).  cin / 4 lanes share a pixel, each holds 4 consecutive input channels (a 16-byte
// load; a wavefront's loads of a tap are contiguous), the partial sums are reduced across the pixel's lanes.
// Weights staged in LDS as [tap][cin][COUT].
template <int COUT, int KS>
__global__ void __launch_bounds__(256) cs_conv_fewout_kernel(const CsConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_wf[];
  constexpr int kk = KS * KS;
  for (int i = threadIdx.x; i < a.cout * a.cin * kk; i += 256) {
    const int co = i / (a.cin * kk), r = i - co * (a.cin * kk);  // r = ci*kk + tap
    const int ci = r / kk, tap = r - ci * kk;
    s_wf[(tap * a.cin + ci) * COUT + co] = a.w[i];
  }
  __syncthreads();
  const int lpp = a.cin >> 2;                 // lanes per pixel: 8, 16, 32 or 64
  const int ppb = 256 / lpp;                  // pixels per block and pass
  const int sub = threadIdx.x & (lpp - 1);
  const int hw = a.hout * a.wout;
  const float* wp0 = s_wf + sub * 4 * COUT;
  for (int64_t gp = (int64_t)blockIdx.x * ppb + threadIdx.x / lpp; gp < a.npix; gp += (int64_t)gridDim.x * ppb) {
    const unsigned g32 = (unsigned)gp, n32 = g32 / (unsigned)hw;
    const int rem = (int)(g32 - n32 * (unsigned)hw);
    const int oh = rem / a.wout, ow = rem - oh * a.wout;
    const int ih0 = oh * a.stride - a.pad_h, iw0 = ow * a.stride - a.pad_w;
    // the lane's 4 channels of input pixel (ih0, iw0); taps are constant offsets from it
    const float* ip0 = a.in + (((int64_t)n32 * a.hin + ih0) * a.win + iw0) * a.cin + sub * 4;
    float s[COUT];
#pragma unroll
    for (int v = 0; v < COUT; ++v) s[v] = 0.0f;
#pragma unroll
    for (int kh = 0; kh < KS; ++kh)
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) {
        const int ih = ih0 + kh, iw = iw0 + kw;
        if (ih >= 0 && ih < a.hin && iw >= 0 && iw < a.win) {
          const f32x4 x = *(const f32x4*)(ip0 + (kh * a.win + kw) * a.cin);
          const float* wp = wp0 + (kh * KS + kw) * a.cin * COUT;
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int v = 0; v < COUT; ++v) s[v] = fmaf(x[c], wp[c * COUT + v], s[v]);
        }
      }
    // the pixel's lanes are consecutive lanes of one wavefront (whole pixels per wavefront: no lane is idle here
    // unless its whole pixel is past the end)
#pragma unroll
    for (int v = 0; v < COUT; ++v)
      for (int o = lpp >> 1; o > 0; o >>= 1) s[v] += __shfl_xor(s[v], o);
    if (sub == 0) {
      const int64_t n = n32;
#pragma unroll
      for (int v = 0; v < COUT; ++v)
        if (v < a.cout) {
          float r = s[v];
          if (a.bias) r += a.bias[v];
          if (a.res) r += a.res[gp * a.cout + v];
          if (a.relu) r = fmaxf(r, 0.0f);
          if (a.out_nchw) a.out[(n * a.cout + v) * hw + rem] = r;
          else a.out[gp * a.cout + v] = r;
        }
    }
  }
}

// 3x3, stride 1, padding 1 convolutions from CIN <= 3 channels to 16 .. 256 NHWC channels (the encoders' first layer):
// bound by writing the output.  One block per image: the image sits in LDS inside a border of zeros (no bounds
// checks), cout / 4 lanes share a pixel, each keeps the 9 * CIN * 4 weights of its 4 output channels in registers and
// stores 16 bytes per pixel (the stores of a wavefront are contiguous).  Plain f32 FMAs in the order of
// cs_conv_direct_kernel (tap, then input channel): the same results bit for bit.
template <int CIN>
__global__ void __launch_bounds__(256) cs_conv3x3_fewin_kernel(const CsConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_img[];  // [CIN][H + 2][W + 2]
  const int H = a.hin, W = a.win, WP = W + 2, plane = (H + 2) * WP, hw = H * W;
  const int64_t n = blockIdx.x;
  for (int i = threadIdx.x; i < CIN * plane; i += 256) {
    const int ci = i / plane, r = i - ci * plane;
    const int y = r / WP - 1, x = r - (y + 1) * WP - 1;
    float v = 0.0f;
    if (y >= 0 && y < H && x >= 0 && x < W)
      v = a.in_nchw ? a.in[(n * CIN + ci) * hw + y * W + x] : a.in[(n * hw + y * W + x) * CIN + ci];
    s_img[i] = v;
  }
  const int tpp = a.cout >> 2;  // lanes per pixel: a power of two
  const int co = (threadIdx.x & (tpp - 1)) * 4;
  float w[9 * CIN][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
      for (int v = 0; v < 4; ++v) w[t * CIN + ci][v] = a.w[((int64_t)(co + v) * CIN + ci) * 9 + t];
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) bias4 = *(const f32x4*)(a.bias + co);
  __syncthreads();
  const int ppp = 256 / tpp;  // pixels per pass
  const float inv_w = 1.0f / (float)W;
  float amx = 0.0f;  // out_amax: the largest magnitude this thread stores (the block is one image)
  for (int px = threadIdx.x / tpp; px < hw; px += ppp) {
    const int oh = (int)(((float)px + 0.5f) * inv_w), ow = px - oh * W;  // exact: (px + 0.5) / W is never within 1/(2W) of an integer
    const float* ip = s_img + oh * WP + ow;  // top-left corner of the pixel's 3x3 window in the bordered image
    float sacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) {
        const float x = ip[ci * plane + (t / 3) * WP + (t % 3)];
#pragma unroll
        for (int v = 0; v < 4; ++v) sacc[v] = fmaf(x, w[t * CIN + ci][v], sacc[v]);
      }
    const int64_t gp = n * hw + px;
    f32x4 o = f32x4{sacc[0], sacc[1], sacc[2], sacc[3]} + bias4;
    if (a.res) o += *(const f32x4*)(a.res + gp * a.cout + co);
    if (a.relu) {
#pragma unroll
      for (int v = 0; v < 4; ++v) o[v] = fmaxf(o[v], 0.0f);
    }
    *(f32x4*)(a.out + gp * a.cout + co) = o;
    amx = fmaxf(fmaxf(amx, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
  }
  if (a.out_amax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o));
    if ((threadIdx.x & 63) == 0 && amx > 0.0f) conv_amax_join(a.out_amax + n, __float_as_uint(amx));
  }
}

// 3x3, stride 1, padding 1 form of cs_conv_fewout_kernel: a block owns `sr` output rows of one image and stages the
// sr + 2 input rows it reads in LDS inside a border of zeros, so that the input crosses the memory system once
// (1 + 2 / sr times) instead of once per kernel row.  Same lane layout (cin / 4 lanes per pixel, 4 consecutive input
// channels each, weights in registers), same order of additions: the same results bit for bit.
template <int COUT>
__global__ void __launch_bounds__(256) cs_conv3x3_fewout_strip_kernel(const CsConvArgs a, int sr, int nstrips, int total) {
  extern __shared__ __attribute__((aligned(16))) float s_strip[];  // [sr + 2][W + 2][cin]
  typedef __attribute__((address_space(3))) unsigned char strip_lds_byte;
  const int H = a.hin, W = a.win, WP = W + 2, hw = H * W, cin = a.cin;
  const int c4n = cin >> 2;  // 16-byte pieces per pixel = lanes per pixel: 8, 16, 32 or 64
  const int lane = threadIdx.x & 63;
  const int wave = (int)cs_scalar_u32(threadIdx.x >> 6);
  const int sub = threadIdx.x & (c4n - 1);
  const int RB = W * c4n, opr = (RB + 63) >> 6;
  const unsigned lds0 = cs_scalar_u32((unsigned)(size_t)(strip_lds_byte*)s_strip);
  float w[9][4][COUT];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int v = 0; v < COUT; ++v) w[t][c][v] = v < a.cout ? a.w[((int64_t)v * cin + sub * 4 + c) * 9 + t] : 0.0f;
  // the two border pixels of every row stay zero for all strips of the block
  for (int i = threadIdx.x; i < (sr + 2) * 2 * c4n; i += 256) {
    const int yy = i / (2 * c4n), r = i - yy * 2 * c4n;
    const int px = r >= c4n ? W + 1 : 0, c4 = r >= c4n ? r - c4n : r;
    *(f32x4*)(s_strip + ((yy * WP + px) * c4n + c4) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int ppp = 256 / c4n;
  const float inv_w = 1.0f / (float)W;
  // the block walks over strips (weights loaded once); several blocks per CU overlap each other's staging
  for (int strip = blockIdx.x; strip < total; strip += gridDim.x) {
    const int64_t n = strip / nstrips;
    const int r0 = (int)(strip - n * nstrips) * sr;  // first output row of the strip
    const int nr = min(sr, H - r0);
    // Staging by LDS-DMA (no registers, every copy of the strip in flight at once): an image row is RB consecutive
    // 16-byte pieces in memory and in LDS; a copy instruction moves up to 64 of them, wavefront `wave` issues copies
    // wave, wave + 4, ...  Rows above / below the image are written as zeros.
    for (int op = wave; op < (nr + 2) * opr; op += 4) {
      const int yy = op / opr, j = op - yy * opr;
      const int y = r0 - 1 + yy;
      const int piece = j * 64 + lane;
      const unsigned dst = (unsigned)((yy * WP + 1) * c4n + j * 64) * 16u;
      if (piece < RB) {
        if (y >= 0 && y < H) {
          const unsigned long long src = (unsigned long long)(a.in + ((n * H + y) * W) * cin + j * 256);
          const void* ssrc =
              (const void*)(((unsigned long long)cs_scalar_u32((unsigned)(src >> 32)) << 32) | cs_scalar_u32((unsigned)src));
          cs_dma16(ssrc, (unsigned)lane * 16u, lds0 + cs_scalar_u32(dst));
        } else {
          *(f32x4*)((unsigned char*)s_strip + dst + lane * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int px = threadIdx.x / c4n; px < nr * W; px += ppp) {
      const int oy = (int)(((float)px + 0.5f) * inv_w), ow = px - oy * W;  // exact: (px + 0.5) / W is never within 1/(2W) of an integer
      const float* ip = s_strip + ((oy * WP + ow) * cin + sub * 4);
      float sacc[COUT];
#pragma unroll
      for (int v = 0; v < COUT; ++v) sacc[v] = 0.0f;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        // taps outside the image are skipped by cs_conv_fewout_kernel; here they multiply zeros, which leaves the sums as they are
        const f32x4 x = *(const f32x4*)(ip + ((t / 3) * WP + (t % 3)) * cin);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int v = 0; v < COUT; ++v) sacc[v] = fmaf(x[c], w[t][c][v], sacc[v]);
      }
#pragma unroll
      for (int v = 0; v < COUT; ++v)
        for (int o = c4n >> 1; o > 0; o >>= 1) sacc[v] += __shfl_xor(sacc[v], o);
      if (sub == 0) {
        const int rem = (r0 + oy) * W + ow;
        const int64_t gp = n * hw + rem;
#pragma unroll
        for (int v = 0; v < COUT; ++v)
          if (v < a.cout) {
            float r = sacc[v];
            if (a.bias) r += a.bias[v];
            if (a.res) r += a.res[gp * a.cout + v];
            if (a.relu) r = fmaxf(r, 0.0f);
            if (a.out_nchw) a.out[(n * a.cout + v) * hw + rem] = r;
            else a.out[gp * a.cout + v] = r;
          }
      }
    }
    __syncthreads();  // the strip is dead: the next one may be copied over it
  }
}

// AdaptiveAvgPool2d((1,1)) over NHWC: one thread per (row, channel).
// nn.LayerNorm(features) over the last dimension (eps 1e-5, biased variance) with its affine
// parameters, optional ReLU: the normalisation inside the reference's residual blocks
// (irwg/models/neural_nets.py:95-100).  One wavefront per row.
__global__ void __launch_bounds__(256) cs_layernorm_kernel(const float* in, float* out, int64_t rows, int n,
                                                           const float* gamma, const float* beta, int relu) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* x = in + r * n;
  float s = 0.0f;
  for (int i = lane; i < n; i += 64) s += x[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)n;
  float v = 0.0f;
  for (int i = lane; i < n; i += 64) {
    const float d = x[i] - mean;
    v += d * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const float rstd = 1.0f / sqrtf(v / (float)n + 1e-5f);
  for (int i = lane; i < n; i += 64) {
    float y = (x[i] - mean) * rstd * gamma[i] + beta[i];
    if (relu) y = fmaxf(y, 0.0f);
    out[r * n + i] = y;
  }
}

__global__ void __launch_bounds__(256) cs_avgpool_kernel(const float* in, float* out, int64_t n, int hw, int c) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * c) return;
  const int64_t r = idx / c;
  const int ch = (int)(idx - r * c);
  const float* p = in + r * hw * c + ch;
  float s = 0.0f;
  for (int i = 0; i < hw; ++i) s += p[(int64_t)i * c];
  out[idx] = s / (float)hw;
}

// F.interpolate(mode='nearest', size=...) over NHWC: src = min(floor(dst * (in/out)), in-1) with
// the scale in float32, as torch computes it.
// grid (slices, image groups): a block walks a slice of ONE image's output, so the per-image maximum of the output
// (out_amax, or null; bits of |value|, zeroed by the caller) costs one join per wavefront and slice.  (One join per
// wavefront of a flat one-element-per-thread grid, 10^5 blocks on the words of 2048 images, took four times the
// kernel's own time.)
__global__ void __launch_bounds__(256) cs_resize_kernel(const float* in, float* out, int64_t n, int hin, int win,
                                                        int hout, int wout, int c, unsigned* out_amax) {
  const int per = hout * wout * c;
  const float sh = (float)hin / (float)hout, sw = (float)win / (float)wout;
  for (int64_t r = blockIdx.y; r < n; r += gridDim.y) {
    float m = 0.0f;
    for (int e = (int)blockIdx.x * 256 + (int)threadIdx.x; e < per; e += (int)gridDim.x * 256) {
      const int ch = e % c;
      const int t = e / c;
      const int ow = t % wout, oh = t / wout;
      int ih = (int)floorf((float)oh * sh), iw = (int)floorf((float)ow * sw);
      ih = ih < hin - 1 ? ih : hin - 1;
      iw = iw < win - 1 ? iw : win - 1;
      const float v = in[((r * hin + ih) * win + iw) * c + ch];
      out[r * per + e] = v;
      m = fmaxf(m, fabsf(v));
    }
    if (out_amax) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      if ((threadIdx.x & 63) == 0 && m > 0.0f) conv_amax_join(out_amax + r, __float_as_uint(m));
    }
  }
}

// ----------------------------------------------------------------------------------------
// stage-in / finish kernels (rows are global row indices row0 + local)
// ----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) cs_gather_rows_kernel(const float* in, int64_t in_stride, const int64_t* in_off,
                                                             int64_t row0, int64_t rows, int dim, float* out, int in_u8) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * dim) return;
  const int64_t r = idx / dim;
  const int c = (int)(idx - r * dim);
  const int64_t row = row0 + r;
  const int64_t e = (in_off ? in_off[row] : row * in_stride) + c;
  out[idx] = in_u8 ? (float)((const uint8_t*)in)[e] : in[e];
}

// Z = loc + scale * eps (or the override row); one thread per 4 latent dimensions.  Beta
// posteriors (toy configuration, vae.py:239-243): Z ~ Beta(alpha, beta) from the counter RNG, or
// the injected sample itself (eps_z then holds the Beta draws).
__global__ void __launch_bounds__(256) cs_dec_stage_kernel(const CsMlpArgs a, int64_t row0, int64_t rows) {
  const int Zd = a.Zd, nq = (Zd + 3) >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * nq) return;
  int64_t row, bk, kp, b, i;
  int c0;
  if ((row0 + rows) * nq < ((int64_t)1 << 31)) {  // 32-bit index arithmetic (64-bit divisions cost ~100 instructions each)
    const uint32_t r32 = (uint32_t)row0 + (uint32_t)idx / (uint32_t)nq;
    c0 = (int)((uint32_t)idx % (uint32_t)nq) * 4;
    const uint32_t bk32 = r32 / (uint32_t)a.Kp;
    row = r32; bk = bk32; kp = r32 - bk32 * (uint32_t)a.Kp; b = bk32 % (uint32_t)a.B; i = bk32 / (uint32_t)a.B;
  } else {
    row = row0 + idx / nq;
    c0 = (int)(idx % nq) * 4;
    bk = row / a.Kp;
    kp = row - bk * a.Kp;
    b = bk % a.B;
    i = row / ((int64_t)a.Kp * a.B);
  }
  const float* qr = a.q + (b * a.Kp + kp) * 2 * Zd;
  const bool ovr = a.z_override != nullptr && a.z_override[row] != 0;
  f32x4 e = {0.f, 0.f, 0.f, 0.f};
  const bool beta = a.latent_dist == CS_DIST_BETA;
  const int64_t prow = (a.rng.row_offset + b) * ((int64_t)a.I * a.Kp) + i * a.Kp + kp;
  if (a.eps_z == nullptr && !ovr && !beta) {
    uint32_t x[4];
    cs_philox_block(a.rng, CS_STREAM_Z, prow, (uint32_t)(c0 >> 2), x);
    e = cs_normal4(x);
  }
  if (!beta && c0 + 4 <= Zd) {  // whole quad inside the row: 16-byte accesses (rows start at any multiple of 4 bytes)
    typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
    f32x4 z;
    if (ovr) {
      z = *(const f32x4_u*)(a.z_alt + row * Zd + c0);
    } else {
      if (a.eps_z) e = *(const f32x4_u*)(a.eps_z + row * Zd + c0);
      const f32x4 lo = *(const f32x4_u*)(qr + c0), sc = *(const f32x4_u*)(qr + Zd + c0);
#pragma unroll
      for (int r = 0; r < 4; ++r) z[r] = lo[r] + sc[r] * e[r];
    }
    *(f32x4_u*)(a.Z + row * Zd + c0) = z;
    return;
  }
  for (int r = 0; r < 4; ++r) {
    const int c = c0 + r;
    if (c >= Zd) break;
    float z;
    if (ovr) {
      z = a.z_alt[row * Zd + c];
    } else if (beta) {
      z = a.eps_z ? a.eps_z[row * Zd + c] : cs_beta_sample(a.rng, CS_STREAM_Z, prow, (uint32_t)c, qr[c], qr[Zd + c]);
    } else {
      const float ev = a.eps_z ? a.eps_z[row * Zd + c] : e[r];
      z = qr[c] + qr[Zd + c] * ev;
    }
    a.Z[row * Zd + c] = z;
  }
}

// raw [rows, 2 Zd] -> q (loc, softplus(raw) (+1e-8)); replenishing rows (loc 0, scale 1).
__global__ void __launch_bounds__(256) cs_enc_finish_kernel(const CsMlpArgs a, const float* raw, int64_t row0,
                                                            int64_t rows) {
  const int Zd = a.Zd;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * Zd) return;
  const int64_t r = idx / Zd;
  const int z = (int)(idx - r * Zd);
  const int64_t row = row0 + r;
  const int64_t b = row / a.K;
  const int64_t k = row - b * a.K;
  float* qr = a.q + (b * a.Kp + k) * 2 * Zd;
  const bool beta = a.latent_dist == CS_DIST_BETA;
  float sc = cs_softplus(raw[r * 2 * Zd + Zd + z]);
  if (a.latent_dist == CS_DIST_NORMAL_WITH_EPS) sc += 1e-8f;
  qr[z] = beta ? cs_softplus(raw[r * 2 * Zd + z]) : raw[r * 2 * Zd + z];
  qr[Zd + z] = sc;
  if (k == 0 && a.Kp > a.K) {  // replenishing proposals: N(0, 1), or Beta(1, 1) = the uniform prior
    float* qp = a.q + (b * a.Kp + a.K) * 2 * Zd;
    for (int t = z; t < (a.Kp - a.K) * 2 * Zd; t += Zd) qp[t] = (beta || ((t / Zd) & 1)) ? 1.0f : 0.0f;
  }
}

// raw [rows, P] -> generator parameters and the masked log-likelihood; one wavefront per row.
__global__ void __launch_bounds__(64) cs_dec_finish_kernel(const CsMlpArgs a, const float* raw, int64_t row0) {
  const int lane = threadIdx.x, D = a.D;
  const int64_t r = blockIdx.x, row = row0 + r;
  const int64_t bk = row / a.Kp;
  const int64_t kp = row - bk * a.Kp;
  const int64_t b = bk % a.B;
  const bool want_ll = a.gen_ll != nullptr;
  const int64_t xk = kp < a.K ? kp : 0;
  const float* xr = want_ll ? a.X + (b * a.K + xk) * D : nullptr;
  const uint8_t* mr = want_ll && a.M ? a.M + b * D : nullptr;  // no mask: every dimension counts
  float ll = 0.0f;
  if (a.gen_dist == CS_DIST_BERN) {
    const float* pr = raw + r * D;
    float* gr = a.gen + row * D;
    for (int d = lane; d < D; d += 64) {
      const float l = pr[d];
      gr[d] = l;
      if (want_ll) ll += cs_bernoulli_logpmf(xr[d], l) * (mr ? (float)mr[d] : 1.0f);
    }
  } else if (cs_dist_is_studentt(a.gen_dist)) {
    // raw (df_raw | loc | scale_raw) -> gen [loc | scale | df] (vae.py:220-229)
    const float* pr = raw + r * 3 * D;
    float* gr = a.gen + row * 3 * D;
    const float eps = a.gen_dist == CS_DIST_STUDENTT_EPS22 ? 0.25f : 0.0f;
    for (int d = lane; d < D; d += 64) {
      const float df = cs_softplus(pr[d]);
      const float loc = pr[D + d];
      const float sc = cs_softplus(pr[2 * D + d]) + eps;
      gr[d] = loc;
      gr[D + d] = sc;
      gr[2 * D + d] = df;
      if (want_ll) ll += cs_studentt_logpdf(xr[d], df, loc, sc) * (mr ? (float)mr[d] : 1.0f);
    }
  } else {
    const float* pr = raw + r * 2 * D;
    float* gr = a.gen + row * 2 * D;
    const float eps = (a.gen_dist == CS_DIST_NORMAL_WITH_EPS || a.gen_dist == CS_DIST_LOGITNORMAL_WITH_EPS) ? 1e-8f : 0.0f;
    for (int d = lane; d < D; d += 64) {
      const float mean = pr[d];
      const float sc = cs_softplus(pr[D + d]) + eps;
      gr[d] = mean;
      gr[D + d] = sc;
      if (want_ll) ll += cs_gen_logpdf(a.gen_dist, xr[d], mean, sc, 0.0f) * (mr ? (float)mr[d] : 1.0f);
    }
  }
  if (want_ll) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ll += __shfl_xor(ll, o);
    if (lane == 0) a.gen_ll[row] = ll;
  }
}

// out[n] = bits of max |x| over image n (elems floats per image, nimg images): grid (slices, image groups); `out`
// zeroed by the caller.
__global__ void __launch_bounds__(256) cs_image_amax_kernel(const float* x, int64_t elems, int64_t nimg, unsigned* out) {
  for (int64_t img = blockIdx.y; img < nimg; img += gridDim.y) {
    const float* xi = x + img * elems;
    float m = 0.0f;
    const int64_t n4 = elems >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
      const f32x4 v = *(const f32x4*)(xi + 4 * i);
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0)
      for (int64_t i = 4 * n4 + threadIdx.x; i < elems; i += 256) m = fmaxf(m, fabsf(xi[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) conv_amax_join(out + img, __float_as_uint(m));
  }
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }
bool g_conv_tail_split = getenv("CSVAE_NO_TAIL_SPLIT") == nullptr;  // measurements without the second launch
// launches that do not fill one round of CUs take the small blocks when a block has at least this many (tap, 32-channel
// chunk) steps (0: never)
int g_conv_small_launch_steps = getenv("CSVAE_SMALL_LAUNCH_STEPS") ? atoi(getenv("CSVAE_SMALL_LAUNCH_STEPS")) : 18;
// measurements with every per-image maximum taken by the reduction pass instead of the producing layers' epilogues
bool g_conv_fused_amax = getenv("CSVAE_NO_FUSED_AMAX") == nullptr;
// fp16 split, same-size 64-channel layers: 256-pixel blocks of 4 waves, two to a CU (74 KB of LDS each), so that one
// block's prologue and epilogue run under the other's matrix instructions (Omniglot encoder 6.05 -> 5.89 ms, decoder
// 7.51 -> 7.24 on one box; the same shape measured slower for the upsampling form, and 128-pixel 4-wave blocks at two per
// CU for the 128+-channel layers 4 % slower: twice the weight stream per CU).  CSVAE_HALO_224=0: 512-pixel 8-wave blocks.
bool g_resident_two = getenv("CSVAE_RESIDENT_ONE") == nullptr;  // measurements with one 8-wave workgroup per CU
bool g_halo_224 = getenv("CSVAE_HALO_224") == nullptr || atoi(getenv("CSVAE_HALO_224")) != 0;
// LDS bytes of cs_conv3x3_halo_split_kernel<CT, PT, NW>: three weight slots, the window, a row of zeros
// (or, if larger, the epilogue's per-wave transposition regions)
// terms = 2 (fp16 split): smaller tiles and rows, the rows' scales behind the window; hw > 0: plus the block's table of
// per-image output maxima at the end (its offset: halo3_lds with hw = 0)
inline size_t halo3_lds(int ct, int pt, int nw, int win, int terms = 3, int hw = 0) {
  const int nslot = terms == 2 ? 4 : 3;
  const size_t nr = (size_t)(32 * pt * nw + 2 * (win + 1));
  const size_t loop = (size_t)nslot * ct * 2048 * terms + nr * 64 * terms + 256 + (terms == 2 ? nr * 4 : 0);
  const size_t store = (size_t)nw * 32 * pt * (ct * 128 + 16);
  const size_t main = ((loop > store ? loop : store) + 15) & ~(size_t)15;
  return main + (hw > 0 ? 4 * ((size_t)(32 * pt * nw) / hw + 2) : 0);
}

// exact 3-way bf16 split of a finite float (round to nearest even at every step): v = t0 + t1 + t2
inline void bf16_split3(float v, uint16_t t[3]) {
  float r = v;
  for (int i = 0; i < 3; ++i) {
    uint32_t u;
    std::memcpy(&u, &r, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    t[i] = (uint16_t)(u >> 16);
    const uint32_t b = (uint32_t)t[i] << 16;
    float f;
    std::memcpy(&f, &b, 4);
    r -= f;
  }
}

// One matrix-kernel convolution described by `a` (a.w / a.w3 set; a.w3 non-null selects the split-bf16 kernels).
int launch_conv_mfma(const CsConvArgs& a, hipStream_t stream) {
  const bool same3 = a.ksize == 3 && a.stride == 1 && a.pad_h == 1 && a.pad_w == 1 && a.us == 0 && a.osc == 1 &&
                     a.hin == a.hout && a.win == a.wout && a.cin % 32 == 0 && a.win + 1 <= 64;
  // all four phases of an upsampled 3x3 convolution (launch_conv sets w3_phase_stride): the halo kernel's 2x2 form
  const bool phases4 = a.w3 && a.w3_phase_stride > 0;
  if (a.w3) {
    // split-bf16 arithmetic: the halo kernel for same-size 3x3 convolutions, else the generic one
    const unsigned gx = (unsigned)((a.npix + 127) / 128);
    int dev = 0;
    CS_HIP(hipGetDevice(&dev));
    CS_REQUIRE(dev >= 0 && dev < 64, "device index out of range");
#define CS_HALO3_LAUNCH_(CT_, PT_, NW_, PH_, TM_)                                                                        \
  do {                                                                                                                   \
    const int NR = 32 * PT_ * NW_ + 2 * (a.win + 1), NP = (NR * 8 + 64 * NW_ - 1) / (64 * NW_);                          \
    const size_t lds = halo3_lds(CT_, PT_, NW_, a.win, TM_, TM_ == 2 ? a.hout * a.wout : 0);                             \
    CS_REQUIRE(NP <= CS_HALO_NPMAX && lds <= 160 * 1024, "halo window does not fit");                                     \
    static bool attr_set[64] = {};                                                                                       \
    if (!attr_set[dev]) {                                                                                                \
      CS_HIP(hipFuncSetAttribute((const void*)cs_conv3x3_halo_split_kernel<CT_, PT_, NW_, PH_, TM_>,                     \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                               \
      attr_set[dev] = true;                                                                                              \
    }                                                                                                                    \
    CsConvArgs al = a;                                                                                                   \
    al.amax_lds = (int)halo3_lds(CT_, PT_, NW_, a.win, TM_, 0);                                                          \
    hipLaunchKernelGGL((cs_conv3x3_halo_split_kernel<CT_, PT_, NW_, PH_, TM_>),                                          \
                       dim3((unsigned)((a.npix + 32 * PT_ * NW_ - 1) / (32 * PT_ * NW_)), a.cout_p / (32 * CT_),         \
                            PH_ ? 4 : 1),                                                                                \
                       dim3(64 * NW_), lds, stream, al, NR, NP);                                                         \
  } while (0)
#define CS_HALO3_LAUNCH(CT_, PT_, NW_)                                  \
  do {                                                                  \
    if (half && phases4) CS_HALO3_LAUNCH_(CT_, PT_, NW_, true, 2);      \
    else if (half) CS_HALO3_LAUNCH_(CT_, PT_, NW_, false, 2);           \
    else if (phases4) CS_HALO3_LAUNCH_(CT_, PT_, NW_, true, 3);         \
    else CS_HALO3_LAUNCH_(CT_, PT_, NW_, false, 3);                     \
  } while (0)
#define CS_CONV3_LAUNCH(KERNEL_, LDS_, ...)                                                                             \
  do {                                                                                                                  \
    static bool attr_set[64] = {};                                                                                      \
    if (!attr_set[dev]) {                                                                                               \
      CS_HIP(hipFuncSetAttribute((const void*)KERNEL_, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));        \
      attr_set[dev] = true;                                                                                             \
    }                                                                                                                   \
    hipLaunchKernelGGL(KERNEL_, dim3(gx, a.cout_p / (32 * CT_)), dim3(256), LDS_, stream, a, ##__VA_ARGS__);            \
  } while (0)
    if (same3 && a.cin == 32 && a.cout == 32 && !a.out_nchw && a.npix < ((int64_t)1 << 31)) {
      // weights resident in LDS, persistent workgroups (cs_conv3x3_halo_resident_kernel).  fp16 split: 4-wave
      // workgroups of 128-pixel items, two to a CU (74 KB of LDS each), whose phases -- matrix instructions, stores,
      // staging of the next window -- interleave; bf16 split (96 KB at that shape): one 8-wave workgroup per CU.
      const bool half = a.w3h != nullptr && a.in_amax != nullptr;  // fp16 split
      static int ncu[64] = {};
      if (!ncu[dev]) CS_HIP(hipDeviceGetAttribute(&ncu[dev], hipDeviceAttributeMultiprocessorCount, dev));
#define CS_RESIDENT_LAUNCH(NW_, TM_, PER_CU_)                                                                                    \
  do {                                                                                                                           \
    const int NR = 32 * NW_ + 2 * (a.win + 1), NP = (NR * 8 + 64 * NW_ - 1) / (64 * NW_);                                        \
    const size_t lds = (size_t)9 * 2048 * TM_ + (((size_t)NR * 64 * TM_ + 256 + 1023) & ~(size_t)1023) + (size_t)NW_ * 32 * 144; \
    CS_REQUIRE(NP <= 6 && lds <= 160 * 1024, "halo window does not fit");                                                        \
    static bool attr_set[64] = {};                                                                                               \
    if (!attr_set[dev]) {                                                                                                        \
      CS_HIP(hipFuncSetAttribute((const void*)cs_conv3x3_halo_resident_kernel<NW_, TM_>,                                         \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                                       \
      attr_set[dev] = true;                                                                                                      \
    }                                                                                                                            \
    const int nitems = (int)((a.npix + 32 * NW_ - 1) / (32 * NW_));                                                              \
    const int grid = nitems < PER_CU_ * ncu[dev] ? nitems : PER_CU_ * ncu[dev];                                                  \
    hipLaunchKernelGGL((cs_conv3x3_halo_resident_kernel<NW_, TM_>), dim3((unsigned)grid), dim3(64 * NW_), lds, stream, a, NR,    \
                       NP, nitems);                                                                                              \
  } while (0)
      if (half && g_resident_two && a.win + 1 <= 32) CS_RESIDENT_LAUNCH(4, 2, 2);   // wider images: the window needs 8 waves to stage
      else if (half) CS_RESIDENT_LAUNCH(8, 2, 1);
      else CS_RESIDENT_LAUNCH(8, 3, 1);
#undef CS_RESIDENT_LAUNCH
    } else if (same3 || phases4) {
      // fp16 split where the caller supplied its weights and the input's per-image maxima
      const bool half = a.w3h != nullptr && a.in_amax != nullptr;
      // one 8-wave block per CU (two waves per SIMD); 64-channel layers give a wavefront two pixel tiles so
      // that a weight fragment feeds 12 matrix instructions
      const int big = a.cout_p % 128 == 0 ? 0 : a.cout_p % 64 == 0 ? (halo3_lds(2, 2, 8, a.win) <= 160 * 1024 ? 1 : 2) : 3;
      // same-size 64-channel layers under the fp16 split: 4-wave blocks, two to a CU (images up to 31 wide: 4 waves stage
      // a window of at most 320 rows in CS_HALO_NPMAX passes)
      const bool two_per_cu = half && g_halo_224 && !phases4 && a.win + 1 <= 32;
      auto halo = [&](const CsConvArgs& a, bool tail) -> int {  // `a` of the macros: this launch's part of the images
        if (tail) CS_HALO3_LAUNCH(1, 1, 4);
        else if (big == 0) CS_HALO3_LAUNCH(4, 1, 8);
        else if (big == 1 && two_per_cu) CS_HALO3_LAUNCH_(2, 2, 4, false, 2);
        else if (big == 1) CS_HALO3_LAUNCH(2, 2, 8);
        else if (big == 2) CS_HALO3_LAUNCH(2, 1, 8);
        else CS_HALO3_LAUNCH(1, 2, 4);
        CS_HIP(hipGetLastError());
        return CS_OK;
      };
      // One block occupies a CU, and all blocks of a launch take the same time: B blocks cost ceil(B / CUs) rounds.
      // Where the last round would be less than an eighth full, the images are divided: the first ones fill whole rounds,
      // the rest runs as a second launch of small blocks (128 pixels x 32 channels, 4 waves, three to a CU) that
      // spread over all CUs and whose matrix-instruction chains are an eighth as long.
      static int ncu[64] = {};
      if (!ncu[dev]) CS_HIP(hipDeviceGetAttribute(&ncu[dev], hipDeviceAttributeMultiprocessorCount, dev));
      const int pxb = big == 1 && !two_per_cu ? 512 : 256, cob = big == 0 ? 128 : big == 3 ? 32 : 64;
      const int64_t nyz = (int64_t)(a.cout_p / cob) * (phases4 ? 4 : 1);
      const int64_t nblocks = (a.npix + pxb - 1) / pxb * nyz;
      const int64_t rounds = nblocks / ncu[dev], last = nblocks - rounds * ncu[dev];
      const int hw = a.hout * a.wout;
      const int64_t n1 = rounds * ncu[dev] / nyz * pxb / hw;  // images whose blocks fill `rounds` rounds
      // (measured: a thinly filled last round costs about half a full one, the second launch 35 us + 60 us per
      // quarter round of work moved into it: it pays for blocks of at least 54 (tap, 32-channel chunk) steps, 256
      // channels at 3x3, and thin remainders)
      if (big != 3 && rounds >= 1 && last > 0 && 8 * last < ncu[dev] && n1 >= 1 && n1 * hw < a.npix &&
          (a.cin >> 5) * (phases4 ? 4 : 9) >= 54 && g_conv_tail_split) {
        CsConvArgs a1 = a, a2 = a;
        a1.npix = n1 * hw;
        a2.npix = a.npix - n1 * hw;
        a2.in = a.in + n1 * hw * a.cin;
        const int64_t opix = n1 * hw * (a.osc == 2 ? 4 : 1) * a.cout;
        a2.out = a.out + opix;
        if (a.res) a2.res = a.res + opix;
        if (a.in_amax) a2.in_amax = a.in_amax + n1;
        if (a.out_amax) a2.out_amax = a.out_amax + n1;
        const int rc = halo(a1, false);
        if (rc != CS_OK) return rc;
        return halo(a2, true);
      }
      // A launch of a few images (the reference's image configurations sample 10 - 20 examples at a time) leaves most
      // CUs idle while each block streams the weights of 64 - 128 output channels through its own loop: the small blocks
      // (32 output channels, 128 pixels, three to a CU) put the same work on up to eight times as many CUs.
      const int64_t nsmall = (a.npix + 127) / 128 * (a.cout_p / 32) * (phases4 ? 4 : 1);
      if (big != 3 && nblocks * 2 <= ncu[dev] && nsmall <= 3 * (int64_t)ncu[dev] && g_conv_small_launch_steps > 0 &&
          (a.cin >> 5) * (phases4 ? 4 : 9) >= g_conv_small_launch_steps)
        return halo(a, true);
      return halo(a, false);
    } else {
      // generic kernel.  Channel tiles per block: 4 or 2 where the channel count allows, fewer while the launch would
      // leave half the CUs without a block (a few rows: more blocks, each with a shorter pass over the weights)
      static int ncu[64] = {};
      if (!ncu[dev]) CS_HIP(hipDeviceGetAttribute(&ncu[dev], hipDeviceAttributeMultiprocessorCount, dev));
      int ct = a.cout_p % 128 == 0 ? 4 : a.cout_p % 64 == 0 ? 2 : 1;
      while (ct > 1 && g_conv_small_launch_steps > 0 && (int64_t)gx * (a.cout_p / (32 * ct)) * 2 <= ncu[dev]) ct >>= 1;
      const bool half = a.w3h != nullptr && a.in_amax != nullptr;  // fp16 split
#define CS_GENERIC_LAUNCH(CTV_)                                                                            \
  do {                                                                                                     \
    constexpr int CT_ = CTV_;                                                                              \
    if (half) CS_CONV3_LAUNCH((cs_conv_mfma_split_kernel<CT_, 2>), (size_t)(128 + CT_ * 32) * 128);        \
    else CS_CONV3_LAUNCH((cs_conv_mfma_split_kernel<CT_>), (size_t)(128 + CT_ * 32) * 192);                \
  } while (0)
      if (ct == 4) CS_GENERIC_LAUNCH(4);
      else if (ct == 2) CS_GENERIC_LAUNCH(2);
      else CS_GENERIC_LAUNCH(1);
#undef CS_GENERIC_LAUNCH
    }
#undef CS_CONV3_LAUNCH
#undef CS_HALO3_LAUNCH
#undef CS_HALO3_LAUNCH_
    CS_HIP(hipGetLastError());
    return CS_OK;
  }
  const int NR = 128 + 2 * (a.win + 1), NP = (NR * 8 + 255) / 256;
#define CS_CONV_LAUNCH(CT_)                                                                               \
  do {                                                                                                    \
    const unsigned gx = (unsigned)((a.npix + 127) / 128);                                                 \
    if (same3 && (size_t)(2 * CT_ * 32 + NR) * 36 * sizeof(float) <= 64 * 1024) {                         \
      const size_t lds = (size_t)(2 * CT_ * 32 + NR) * 36 * sizeof(float);                                \
      hipLaunchKernelGGL((cs_conv3x3_halo_kernel<CT_>), dim3(gx, a.cout_p / (32 * CT_)), dim3(256), lds,  \
                         stream, a, NR, NP);                                                              \
    } else {                                                                                              \
      const size_t lds = (size_t)(128 + CT_ * 32) * 36 * sizeof(float);                                   \
      hipLaunchKernelGGL((cs_conv_mfma_kernel<CT_, 1>), dim3(gx, a.cout_p / (32 * CT_)), dim3(256), lds,  \
                         stream, a);                                                                      \
    }                                                                                                     \
  } while (0)
  // Measured on the Omniglot / MNIST-GMM nets: 128-pixel x 128-channel blocks (two blocks per CU)
  // beat 256-pixel blocks, 64-channel blocks and a two-stage LDS ring by 2-7 %.
  if (a.cout_p % 128 == 0) CS_CONV_LAUNCH(4);
  else if (a.cout_p % 64 == 0) CS_CONV_LAUNCH(2);
  else CS_CONV_LAUNCH(1);
#undef CS_CONV_LAUNCH
  CS_HIP(hipGetLastError());
  return CS_OK;
}

// Convolutions to at most 4 channels from 32, 64, 128 or 256 NHWC channels: cs_conv_fewout_kernel.
bool fewout_eligible(const cs_layer_desc& d) {
  return d.op == CS_OP_CONV && d.cout <= 4 && (d.cin == 32 || d.cin == 64 || d.cin == 128 || d.cin == 256) && !d.in_nchw &&
         d.upsample == 1 && (d.ksize == 3 || d.ksize == 1) && (int64_t)d.ksize * d.ksize * d.cin * 4 * 4 <= 48 * 1024;
}

// How launch_conv runs layer L under the fp16-split arithmetic: 0 = not at all (another arithmetic is selected, a
// weight is not finite, or the layer runs on a kernel without that arithmetic: the few-channel and direct kernels,
// linear layers), 1 = on the halo kernel (same-size and four-phase upsampled 3x3 convolutions; its epilogue leaves
// the per-image maxima of the output), 2 = on the generic kernel, 3 = on the weight-resident 32 -> 32 channel kernel.
int layer_half_kind(const CsProgram& p, const CsLayer& L) {
  const cs_layer_desc& d = L.d;
  if ((p.arith >= 0 ? p.arith : cs_mlp_arith()) != CS_ARITH_F16X2 || L.wp3_off < 0 || L.wp3h_off < 0) return 0;
  if (d.op != CS_OP_CONV || d.in_nchw) return 0;
  const bool halo_ok = d.cin % 32 == 0 && d.cout % 32 == 0 && !d.out_nchw && d.win + 1 <= 64;
  if (L.up_mode == CS_UP_NONE) {
    const bool same3 = d.ksize == 3 && d.stride == 1 && d.pad == 1 && d.upsample == 1 && d.hin == d.hout && d.win == d.wout;
    if (same3 && d.cin == 32 && d.cout == 32 && !d.out_nchw) return d.win + 1 <= 64 ? 3 : 2;  // weight-resident kernel
    return same3 && halo_ok ? 1 : 2;
  }
  if (L.up_mode == CS_UP_REPLICATE) return 2;
  for (int ph = 0; ph < 4; ++ph)
    if (L.wph3h_off[ph] < 0) return 0;
  const int64_t st = L.wph3_off[1] - L.wph3_off[0], sth = L.wph3h_off[1] - L.wph3h_off[0];
  const bool spaced = st > 0 && L.wph3_off[2] - L.wph3_off[0] == 2 * st && L.wph3_off[3] - L.wph3_off[0] == 3 * st && sth > 0 &&
                      L.wph3h_off[2] - L.wph3h_off[0] == 2 * sth && L.wph3h_off[3] - L.wph3h_off[0] == 3 * sth;
  return (halo_ok && spaced && d.cout % 64 == 0 &&
          halo3_lds(d.cout % 128 == 0 ? 4 : 2, d.cout % 128 == 0 ? 1 : 2, 8, d.win) <= 160 * 1024)
             ? 1
             : 2;
}

// in_amax / out_amax: per-image maxima of the source / for the output (fp16 split, see CsConvArgs), or null
int launch_conv(const CsProgram& p, const CsLayer& L, const float* in, const float* res, float* out, int64_t rows,
                hipStream_t stream, const unsigned* in_amax = nullptr, unsigned* out_amax = nullptr,
                bool* wrote_amax = nullptr) {
  const cs_layer_desc& d = L.d;
  if (wrote_amax) *wrote_amax = false;  // set where the kernel chosen below fills out_amax
  CS_REQUIRE(rows * d.hout * d.wout < ((int64_t)1 << 31), "too many output pixels in one launch");
  CsConvArgs a;
  a.in = in; a.res = res; a.out = out;
  a.bias = d.b_off >= 0 ? p.d_blob + d.b_off : nullptr;
  a.cin = d.cin; a.cout = d.cout; a.hin = d.hin; a.win = d.win; a.hout = d.hout; a.wout = d.wout;
  a.ksize = d.ksize; a.stride = d.stride; a.pad_h = a.pad_w = d.pad; a.us = d.upsample == 2 ? 1 : 0; a.relu = d.relu;
  a.in_nchw = d.in_nchw; a.out_nchw = d.out_nchw;
  a.npix = rows * d.hout * d.wout;
  a.cout_p = (d.cout + 31) & ~31;
  a.osc = 1; a.ooh = a.oow = 0; a.rep = 0;
  a.w3_phase_stride = 0;
  a.w3 = nullptr;
  a.w3h = nullptr;
  a.w3h_phase_stride = 0;
  a.wexp[0] = a.wexp[1] = a.wexp[2] = a.wexp[3] = 0;
  a.in_amax = nullptr;
  a.out_amax = nullptr;
  a.amax_lds = 0;
  const int half_kind = in_amax != nullptr ? layer_half_kind(p, L) : 0;
  const bool half = half_kind == 1;
  const bool same3 = d.ksize == 3 && d.stride == 1 && d.pad == 1 && d.upsample == 1 && d.hin == d.hout && d.win == d.wout;
  if (fewout_eligible(d) && same3 && (size_t)3 * (d.win + 2) * d.cin * 4 <= 48 * 1024) {
    // strips of `sr` rows whose sr + 2 input rows fit 48 KiB of LDS (three blocks per CU)
    a.w = p.d_blob + d.w_off;
    const size_t rowb = (size_t)(d.win + 2) * d.cin * 4;
    int sr = (int)(48 * 1024 / rowb) - 2;
    if (sr > d.hin) sr = d.hin;
    const int nstrips = (d.hin + sr - 1) / sr;
    sr = (d.hin + nstrips - 1) / nstrips;  // even strips
    const size_t lds = (size_t)(sr + 2) * rowb;
    CS_REQUIRE(rows * nstrips < ((int64_t)1 << 31), "too many strips in one launch");
    const int total = (int)(rows * nstrips);
    const unsigned gx = (unsigned)(total < 3 * 256 ? total : 3 * 256);
    if (d.cout == 1) hipLaunchKernelGGL((cs_conv3x3_fewout_strip_kernel<1>), dim3(gx), dim3(256), lds, stream, a, sr, nstrips, total);
    else if (d.cout == 2) hipLaunchKernelGGL((cs_conv3x3_fewout_strip_kernel<2>), dim3(gx), dim3(256), lds, stream, a, sr, nstrips, total);
    else hipLaunchKernelGGL((cs_conv3x3_fewout_strip_kernel<4>), dim3(gx), dim3(256), lds, stream, a, sr, nstrips, total);
    CS_HIP(hipGetLastError());
    return CS_OK;
  }
  if (fewout_eligible(d)) {
    a.w = p.d_blob + d.w_off;
    const int ppb = 256 / (d.cin / 4);
    const size_t lds = (size_t)d.ksize * d.ksize * d.cin * (d.cout <= 2 ? 2 : 4) * sizeof(float);
    const int64_t nb = (a.npix + ppb - 1) / ppb;
    const unsigned gx = (unsigned)(nb < 4096 ? nb : 4096);  // grid-stride: the weights are staged once per block
    if (d.ksize == 3) {
      if (d.cout <= 2) hipLaunchKernelGGL((cs_conv_fewout_kernel<2, 3>), dim3(gx), dim3(256), lds, stream, a);
      else hipLaunchKernelGGL((cs_conv_fewout_kernel<4, 3>), dim3(gx), dim3(256), lds, stream, a);
    } else {
      if (d.cout <= 2) hipLaunchKernelGGL((cs_conv_fewout_kernel<2, 1>), dim3(gx), dim3(256), lds, stream, a);
      else hipLaunchKernelGGL((cs_conv_fewout_kernel<4, 1>), dim3(gx), dim3(256), lds, stream, a);
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
  }
  if (L.wp_off < 0 && same3 && d.cin <= 3 && !d.out_nchw && d.cout >= 16 && d.cout <= 256 && (d.cout & (d.cout - 1)) == 0 &&
      (size_t)d.cin * (d.hin + 2) * (d.win + 2) * 4 <= 48 * 1024) {
    a.w = p.d_blob + d.w_off;
    if (g_conv_fused_amax && out_amax) {  // a block of this kernel is one image
      a.out_amax = out_amax;
      if (wrote_amax) *wrote_amax = true;
    }
    const size_t lds = (size_t)d.cin * (d.hin + 2) * (d.win + 2) * 4;
    if (d.cin == 1) hipLaunchKernelGGL((cs_conv3x3_fewin_kernel<1>), dim3((unsigned)rows), dim3(256), lds, stream, a);
    else if (d.cin == 2) hipLaunchKernelGGL((cs_conv3x3_fewin_kernel<2>), dim3((unsigned)rows), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL((cs_conv3x3_fewin_kernel<3>), dim3((unsigned)rows), dim3(256), lds, stream, a);
    CS_HIP(hipGetLastError());
    return CS_OK;
  }
  if (L.wp_off < 0) {
    a.w = p.d_blob + d.w_off;
    const unsigned cap = 8192;
    if (d.cout % 4 == 0 && !d.out_nchw)
      hipLaunchKernelGGL(cs_conv_direct_kernel<4>, dim3(std::min(blocks_for(a.npix * (d.cout / 4)), cap)), dim3(256), 0, stream, a);
    else
      hipLaunchKernelGGL(cs_conv_direct_kernel<1>, dim3(std::min(blocks_for(a.npix * d.cout), cap)), dim3(256), 0, stream, a);
    CS_HIP(hipGetLastError());
    return CS_OK;
  }
  const bool split = L.wp3_off >= 0 && (p.arith >= 0 ? p.arith : cs_mlp_arith()) != CS_ARITH_F32;  // plain bf16 is a fused-MLP arithmetic: layer programs keep the split
  if (L.up_mode == CS_UP_NONE) {
    a.w = p.d_blob + L.wp_off;
    if (split) a.w3 = p.d_blob3 + L.wp3_off;
    if (split && half_kind != 0) {
      a.w3h = p.d_blob3h + L.wp3h_off;
      a.wexp[0] = L.wexp;
      a.in_amax = in_amax;
    }
    // kernels whose epilogue leaves the output's per-image maxima: the fp16-split halo kernel, the generic split
    // kernel (NHWC outputs of 32-multiple channel counts), the weight-resident kernel (images of 32 pixels and up)
    const bool halo3 = same3 && d.cin % 32 == 0 && d.win + 1 <= 64;
    const bool resident = halo3 && d.cin == 32 && d.cout == 32 && !d.out_nchw;
    if (split && out_amax && d.cout % 32 == 0 && !d.out_nchw &&
        (half || (g_conv_fused_amax && (resident ? d.hout * d.wout >= 32 : !halo3)))) {
      a.out_amax = out_amax;
      if (wrote_amax) *wrote_amax = true;
    }
    return launch_conv_mfma(a, stream);
  }
  // upsampled convolutions run on the input's own grid (see CsUpsampleMode)
  a.us = 0;
  a.hout = d.hin; a.wout = d.win;
  a.npix = rows * d.hin * d.win;
  a.osc = 2;
  if (L.up_mode == CS_UP_REPLICATE) {
    a.rep = 1;
    a.w = p.d_blob + L.wp_off;
    if (split) a.w3 = p.d_blob3 + L.wp3_off;
    if (split && half_kind == 2) {
      a.w3h = p.d_blob3h + L.wp3h_off;
      a.wexp[0] = L.wexp;
      a.in_amax = in_amax;
    }
    if (split && out_amax && g_conv_fused_amax && d.cout % 32 == 0 && !d.out_nchw) {  // generic kernel's store epilogue
      a.out_amax = out_amax;
      if (wrote_amax) *wrote_amax = true;
    }
    return launch_conv_mfma(a, stream);
  }
  a.ksize = 2;
  {
    // split arithmetic: one launch of the halo kernel for all four phases where its window fits
    const int64_t st = L.wph3_off[1] - L.wph3_off[0];
    const bool spaced = st > 0 && L.wph3_off[2] - L.wph3_off[0] == 2 * st && L.wph3_off[3] - L.wph3_off[0] == 3 * st;
    // (64 output channels and up: measured on 64 -> 32 channels at 7x7 -> 14x14, 8192 images, the halo kernel's
    // 256-pixel x 32-channel blocks take 527 us, the four launches of the generic kernel 353)
    if (split && spaced && d.cin % 32 == 0 && d.cout % 64 == 0 && !d.out_nchw && d.win + 1 <= 64 &&
        halo3_lds(d.cout % 128 == 0 ? 4 : 2, d.cout % 128 == 0 ? 1 : 2, 8, d.win) <= 160 * 1024) {
      a.w = p.d_blob + L.wph_off[0];
      a.w3 = p.d_blob3 + L.wph3_off[0];
      a.w3_phase_stride = st;
      if (half) {
        a.w3h = p.d_blob3h + L.wph3h_off[0];
        a.w3h_phase_stride = L.wph3h_off[1] - L.wph3h_off[0];
        for (int ph = 0; ph < 4; ++ph) a.wexp[ph] = L.wphexp[ph];
        a.in_amax = in_amax;
        a.out_amax = out_amax;
        if (wrote_amax) *wrote_amax = out_amax != nullptr;
      }
      return launch_conv_mfma(a, stream);
    }
  }
  if (split && out_amax && g_conv_fused_amax && d.cout % 32 == 0 && !d.out_nchw) {  // the four launches join one table
    a.out_amax = out_amax;
    if (wrote_amax) *wrote_amax = true;
  }
  for (int ph = 0; ph < 4; ++ph) {
    a.ooh = ph >> 1; a.oow = ph & 1;
    a.pad_h = 1 - a.ooh; a.pad_w = 1 - a.oow;
    a.w = p.d_blob + L.wph_off[ph];
    a.w3 = split ? p.d_blob3 + L.wph3_off[ph] : nullptr;
    if (split && half_kind == 2) {
      a.w3h = p.d_blob3h + L.wph3h_off[ph];
      a.wexp[0] = L.wphexp[ph];
      a.in_amax = in_amax;
    }
    const int rc = launch_conv_mfma(a, stream);
    if (rc != CS_OK) return rc;
  }
  return CS_OK;
}

// The matrix-core kernel takes every convolution / linear layer whose contraction is worth padding
// to 32 channels: all but layers reading very few channels (the first convolution from a 1-channel
// image is bound by its output writes, not by arithmetic) or a channel-major multi-channel input.
bool mfma_eligible(const cs_layer_desc& d) {
  return (d.op == CS_OP_CONV || d.op == CS_OP_LINEAR) && d.cin >= 16 && !(d.in_nchw && d.cin > 1);
}

int64_t g_workspace_budget_bytes = (int64_t)6 << 30;

}  // namespace

int cs_dec_stage_launch(const CsMlpArgs& a, int64_t row0, int64_t rows, hipStream_t stream) {
  const int64_t n = rows * ((a.Zd + 3) / 4);
  hipLaunchKernelGGL(cs_dec_stage_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, a, row0, rows);
  CS_HIP(hipGetLastError());
  return CS_OK;
}

extern "C" int cs_debug_set_small_launch_steps(int steps) {
  const int prev = g_conv_small_launch_steps;
  g_conv_small_launch_steps = steps;
  return prev;
}

extern "C" int cs_debug_set_workspace_budget(int64_t bytes) {
  CS_REQUIRE(bytes >= 0, "negative budget");
  g_workspace_budget_bytes = bytes ? bytes : ((int64_t)6 << 30);
  return CS_OK;
}

void cs_prog_destroy(CsProgram* p) {
  if (!p) return;
  if (p->d_blob) (void)hipFree(p->d_blob);
  if (p->d_blob3) (void)hipFree(p->d_blob3);
  if (p->d_blob3h) (void)hipFree(p->d_blob3h);
  if (p->amax_ws) (void)hipFree(p->amax_ws);
  if (p->ws) (void)hipFree(p->ws);
  for (int k = 0; k < 3; ++k) {
    if (p->side_stream[k]) (void)hipStreamDestroy(p->side_stream[k]);
    if (p->side_join[k]) (void)hipEventDestroy(p->side_join[k]);
  }
  if (p->side_fork) (void)hipEventDestroy(p->side_fork);
  delete p;
}

extern "C" int cs_net_create(const cs_layer_desc* layers, int32_t n_layers, const float* blob, int64_t blob_floats,
                             int32_t in_dim, int32_t out_dim, int32_t out_split, cs_mlp_t** out) {
  CS_REQUIRE(layers && blob && out, "null argument");
  CS_REQUIRE(n_layers >= 1 && n_layers <= 4096, "bad layer count");
  CS_REQUIRE(in_dim >= 1 && out_dim >= 1, "bad network dimensions");
  CS_REQUIRE(out_split >= 1 && out_split <= 3, "out_split must be 1, 2 or 3");
  CS_REQUIRE(out_dim % out_split == 0, "out_dim not divisible by out_split");
  CsProgram* p = new CsProgram();
  p->in_dim = in_dim;
  p->out_dim = out_dim;
  p->buf_elems[CS_NET_BUF_IN] = in_dim;
  p->buf_elems[CS_NET_BUF_OUT] = out_dim;
  std::vector<float> packed;
  std::vector<uint16_t> packed3, packed3h;
  int last_writer[CS_NET_NBUF];
  for (int b = 0; b < CS_NET_NBUF; ++b) last_writer[b] = -1;
  auto fail = [&](const char* msg) {
    delete p;
    return cs_fail(CS_EINVAL, std::string("cs_net_create: ") + msg);
  };
  for (int i = 0; i < n_layers; ++i) {
    cs_layer_desc d = layers[i];
    if (d.op == CS_OP_LINEAR) {
      d.hin = d.win = d.hout = d.wout = 1; d.ksize = 1; d.stride = 1; d.pad = 0; d.upsample = 1;
      d.in_nchw = d.out_nchw = 0;
    }
    if (d.op == CS_OP_LAYERNORM) {
      d.hin = d.win = d.hout = d.wout = 1; d.in_nchw = d.out_nchw = 0;
    }
    if (d.op < CS_OP_CONV || d.op > CS_OP_LAYERNORM) return fail("unknown op");
    if (d.src < 0 || d.src >= CS_NET_NBUF || d.dst < 1 || d.dst >= CS_NET_NBUF || d.src == d.dst) return fail("bad buffer id");
    if (d.res >= CS_NET_NBUF || d.res == d.dst) return fail("bad residual buffer id");
    if (d.cin < 1 || d.cout < 1 || d.hin < 1 || d.win < 1 || d.hout < 1 || d.wout < 1) return fail("bad layer shape");
    const int64_t in_elems = (int64_t)d.cin * d.hin * d.win, out_elems = (int64_t)d.cout * d.hout * d.wout;
    if (d.op == CS_OP_CONV || d.op == CS_OP_LINEAR) {
      if (d.ksize != 1 && d.ksize != 3) return fail("kernel size must be 1 or 3");
      if (d.upsample != 1 && d.upsample != 2) return fail("upsample must be 1 or 2");
      if (d.stride < 1) return fail("bad stride");
      const int hup = d.hin * d.upsample, wup = d.win * d.upsample;
      if ((hup + 2 * d.pad - d.ksize) / d.stride + 1 != d.hout || (wup + 2 * d.pad - d.ksize) / d.stride + 1 != d.wout)
        return fail("convolution output size does not match");
      const int64_t wn = (int64_t)d.cout * d.cin * d.ksize * d.ksize;
      if (d.w_off < 0 || d.w_off + wn > blob_floats) return fail("weight offset out of range");
      if (d.b_off >= 0 && d.b_off + d.cout > blob_floats) return fail("bias offset out of range");
      if (d.res >= 0 && d.out_nchw) return fail("residual with a channel-major output");
      if (d.b_off >= 0 && d.b_off % 4 != 0) return fail("bias offsets must be multiples of 4 floats");
    } else if (d.op == CS_OP_LAYERNORM) {
      if (d.cin != d.cout) return fail("channel counts must match");
      if (d.res >= 0) return fail("no residual input on a layer norm");
      if (d.w_off < 0 || d.w_off + d.cin > blob_floats || d.b_off < 0 || d.b_off + d.cin > blob_floats)
        return fail("layer-norm weight / bias offset out of range");
    } else {
      if (d.cin != d.cout) return fail("channel counts must match");
      if (d.op == CS_OP_AVGPOOL && (d.hout != 1 || d.wout != 1)) return fail("average pool is global");
      if (d.res >= 0 || d.relu) return fail("residual / ReLU only on convolutions");
    }
    if (d.src == CS_NET_BUF_IN ? in_elems != in_dim : in_elems > p->buf_elems[d.src]) return fail("source buffer not produced at this size");
    if (d.res >= 0 && out_elems > p->buf_elems[d.res]) return fail("residual buffer not produced at this size");
    if (d.dst == CS_NET_BUF_OUT) {
      if (out_elems != out_dim) return fail("output layer size does not match out_dim");
    } else if (out_elems > p->buf_elems[d.dst]) {
      p->buf_elems[d.dst] = out_elems;
    }
    CsLayer L;
    L.d = d;
    L.wp_off = -1;
    L.wp3_off = -1;
    L.src_layer = d.src == CS_NET_BUF_IN ? -1 : last_writer[d.src];
    last_writer[d.dst] = i;
    if (mfma_eligible(d) && !fewout_eligible(d)) {
      // Packs weights Wsrc [cout][cin][kk] for the matrix kernels: f32 as [tap][cin_p/32][cout_p][32], zero padded, and
      // the split-bf16 image [tap][cin_p/32][cout_p/32] bricks of [k-step 2][term 3][lane 64] x 8 bf16, the order in
      // which the lanes of a wavefront consume them as the A operand of v_mfma_f32_32x32x16_bf16 (lane l: channel
      // l & 31 of the tile, k = 16*step + 8*(l >> 5) .. + 7).
      const int ncc = (d.cin + 31) / 32, cout_p = (d.cout + 31) & ~31;
      // ... and the fp16-split image, the same bricks with two fp16 terms of w 2^kexp, kexp the power of two that
      // brings the largest weight of the pack into [2^14, 2^15) (*off3h = -1 when a weight is not finite)
      auto pack = [&](const float* Wsrc, int kk, int64_t* off, int64_t* off3, int64_t* off3h, int* kexp_out) {
        *off = blob_floats + (int64_t)packed.size();
        for (int tap = 0; tap < kk; ++tap)
          for (int cc = 0; cc < ncc; ++cc)
            for (int co = 0; co < cout_p; ++co)
              for (int c = 0; c < 32; ++c) {
                const int ci = cc * 32 + c;
                packed.push_back(co < d.cout && ci < d.cin ? Wsrc[((int64_t)co * d.cin + ci) * kk + tap] : 0.0f);
              }
        *off3 = (int64_t)packed3.size();
        for (int tap = 0; tap < kk; ++tap)
          for (int cc = 0; cc < ncc; ++cc)
            for (int tile = 0; tile < cout_p / 32; ++tile) {
              std::vector<uint16_t> brick(2 * 3 * 64 * 8, 0);
              for (int l = 0; l < 64; ++l) {
                const int co = tile * 32 + (l & 31);
                for (int st = 0; st < 2; ++st)
                  for (int j = 0; j < 8; ++j) {
                    const int ci = cc * 32 + st * 16 + (l >> 5) * 8 + j;
                    const float v = co < d.cout && ci < d.cin ? Wsrc[((int64_t)co * d.cin + ci) * kk + tap] : 0.0f;
                    uint16_t t3[3];
                    bf16_split3(v, t3);
                    for (int tm = 0; tm < 3; ++tm) brick[((st * 3 + tm) * 64 + l) * 8 + j] = t3[tm];
                  }
              }
              packed3.insert(packed3.end(), brick.begin(), brick.end());
            }
        bool finite = true;
        float mx = 0.0f;
        for (int64_t q = 0; q < (int64_t)d.cout * d.cin * kk; ++q) {
          if (!std::isfinite(Wsrc[q])) finite = false;
          mx = std::fmax(mx, std::fabs(Wsrc[q]));
        }
        int ex = 0;
        if (mx > 0.0f && std::isfinite(mx)) (void)std::frexp(mx, &ex);  // mx = m 2^ex, m in [0.5, 1)
        int kexp = mx > 0.0f ? 15 - ex : 0;
        kexp = kexp < -100 ? -100 : (kexp > 100 ? 100 : kexp);
        *kexp_out = kexp;
        *off3h = -1;
        if (!finite) return;
        *off3h = (int64_t)packed3h.size();
        auto f16_bits = [](float f) {
          const _Float16 hv = (_Float16)f;  // round to nearest even
          uint16_t u;
          std::memcpy(&u, &hv, 2);
          return u;
        };
        for (int tap = 0; tap < kk; ++tap)
          for (int cc = 0; cc < ncc; ++cc)
            for (int tile = 0; tile < cout_p / 32; ++tile) {
              std::vector<uint16_t> brick(2 * 2 * 64 * 8, 0);
              for (int l = 0; l < 64; ++l) {
                const int co = tile * 32 + (l & 31);
                for (int st = 0; st < 2; ++st)
                  for (int j = 0; j < 8; ++j) {
                    const int ci = cc * 32 + st * 16 + (l >> 5) * 8 + j;
                    const float v = co < d.cout && ci < d.cin ? std::ldexp(Wsrc[((int64_t)co * d.cin + ci) * kk + tap], kexp) : 0.0f;
                    const uint16_t t0 = f16_bits(v);
                    _Float16 h0;
                    std::memcpy(&h0, &t0, 2);
                    brick[((st * 2 + 0) * 64 + l) * 8 + j] = t0;
                    brick[((st * 2 + 1) * 64 + l) * 8 + j] = f16_bits(v - (float)h0);
                  }
              }
              packed3h.insert(packed3h.end(), brick.begin(), brick.end());
            }
      };
      const float* W = blob + d.w_off;
      const bool up2 = d.op == CS_OP_CONV && d.upsample == 2 && d.stride == 1 && !d.out_nchw && d.cout % 32 == 0;
      if (up2 && d.ksize == 3 && d.pad == 1) {
        // phase (a, b): 2x2 taps (dh, dw) over input rows i - 1 + a + dh; kernel rows kh with (2i + a + kh - 1) >> 1
        // equal to that row are summed (float64, rounded once), likewise the columns
        L.up_mode = CS_UP_PHASES;
        std::vector<float> Wph((size_t)d.cout * d.cin * 4);
        for (int ph = 0; ph < 4; ++ph) {
          const int pa = ph >> 1, pb = ph & 1;
          for (int64_t cc = 0; cc < (int64_t)d.cout * d.cin; ++cc)
            for (int dh = 0; dh < 2; ++dh)
              for (int dw = 0; dw < 2; ++dw) {
                double sum = 0.0;
                for (int kh = 0; kh < 3; ++kh)
                  for (int kw = 0; kw < 3; ++kw)
                    if (((pa + kh + 1) >> 1) - 1 + (1 - pa) == dh && ((pb + kw + 1) >> 1) - 1 + (1 - pb) == dw)
                      sum += (double)W[cc * 9 + kh * 3 + kw];
                Wph[cc * 4 + dh * 2 + dw] = (float)sum;
              }
          pack(Wph.data(), 4, &L.wph_off[ph], &L.wph3_off[ph], &L.wph3h_off[ph], &L.wphexp[ph]);
        }
        L.wp_off = L.wph_off[0];
        L.wp3_off = L.wph3_off[0];
        L.wp3h_off = (L.wph3h_off[0] < 0 || L.wph3h_off[1] < 0 || L.wph3h_off[2] < 0 || L.wph3h_off[3] < 0) ? -1 : L.wph3h_off[0];
      } else {
        if (up2 && d.ksize == 1 && d.pad == 0) L.up_mode = CS_UP_REPLICATE;
        pack(W, d.ksize * d.ksize, &L.wp_off, &L.wp3_off, &L.wp3h_off, &L.wexp);
      }
    }
    p->layers.push_back(L);
  }
  if (p->layers.back().d.dst != CS_NET_BUF_OUT) return fail("the last layer must write the output buffer");

  hipError_t e = hipGetDevice(&p->device);
  if (e == hipSuccess) e = hipMalloc(&p->d_blob, (blob_floats + packed.size() + 4) * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(p->d_blob, blob, blob_floats * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess && !packed.empty())
    e = hipMemcpy(p->d_blob + blob_floats, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess && !packed3.empty()) e = hipMalloc(&p->d_blob3, packed3.size() * sizeof(uint16_t));
  if (e == hipSuccess && !packed3.empty())
    e = hipMemcpy(p->d_blob3, packed3.data(), packed3.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
  if (e == hipSuccess && !packed3h.empty()) e = hipMalloc(&p->d_blob3h, packed3h.size() * sizeof(uint16_t));
  if (e == hipSuccess && !packed3h.empty())
    e = hipMemcpy(p->d_blob3h, packed3h.data(), packed3h.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    cs_prog_destroy(p);
    return cs_fail(CS_EHIP, std::string("cs_net_create: ") + hipGetErrorString(e));
  }
  cs_mlp* m = new cs_mlp();
  std::memset(&m->desc, 0, sizeof(m->desc));
  std::memset(&m->dev, 0, sizeof(m->dev));
  m->desc.in_dim = in_dim; m->desc.out_dim = out_dim; m->desc.out_split = out_split;
  m->d_w = nullptr; m->d_b = nullptr; m->d_w3 = nullptr; m->d_b3 = nullptr;
  m->device = p->device;
  m->prog = p;
  m->dev.in_dim = in_dim; m->dev.out_dim = out_dim; m->dev.out_half = out_dim / out_split; m->dev.split = out_split;
  m->dev.prog = p;
  m->dev.arith = -1;
  *out = m;
  return CS_OK;
}

// rows of the workspace that one pass of a launch of `rows` rows needs, and floats per row
static int64_t prog_chunk_rows(const CsProgram* p, int64_t rows, int64_t* per_row_out) {
  int64_t per_row = 0;
  for (int b = 0; b < CS_NET_NBUF; ++b) per_row += (p->buf_elems[b] + 3) & ~(int64_t)3;  // every buffer 16-byte aligned
  int64_t chunk = g_workspace_budget_bytes / (per_row * 4);
  if (chunk < 1) chunk = 1;
  if (chunk > rows) chunk = rows;
  chunk = (chunk + 3) & ~(int64_t)3;
  if (per_row_out) *per_row_out = per_row;
  return chunk;
}
static int prog_grow_workspace(CsProgram* p, int64_t chunk, int64_t per_row) {
  if (chunk <= p->ws_rows) return CS_OK;
  if (p->ws) CS_HIP(hipFree(p->ws));  // waits for the work that still uses it
  if (p->amax_ws) CS_HIP(hipFree(p->amax_ws));
  p->ws = nullptr;
  p->amax_ws = nullptr;
  p->ws_rows = 0;
  CS_HIP(hipMalloc(&p->ws, (size_t)chunk * per_row * sizeof(float)));
  CS_HIP(hipMalloc(&p->amax_ws, (p->layers.size() + 1) * (size_t)chunk * sizeof(unsigned)));
  p->ws_rows = chunk;
  return CS_OK;
}

extern "C" int cs_net_reserve(cs_mlp_t* net, int64_t rows) {
  CS_REQUIRE(net && rows >= 0, "bad argument");
  if (!net->prog || rows == 0) return CS_OK;  // the fused kernels have no workspace
  int dev = 0;
  CS_HIP(hipGetDevice(&dev));
  CS_REQUIRE(dev == net->prog->device, "network was created on another device");
  int64_t per_row = 0;
  const int64_t chunk = prog_chunk_rows(net->prog, rows, &per_row);
  return prog_grow_workspace(net->prog, chunk, per_row);
}

// A pass of at least this many rows (and enough work, see cs_prog_launch) runs as two halves on two streams (0: never).
int g_prog_two_streams = getenv("CSVAE_PROG_TWO_STREAMS") ? atoi(getenv("CSVAE_PROG_TWO_STREAMS")) : 8;
// streams and events of a pass that runs in parts (per network handle: calls on one handle may not overlap anyway,
// its workspace is one); false when they cannot be created
static bool prog_side(CsProgram* p, int nside) {
  if (!p->side_fork && hipEventCreateWithFlags(&p->side_fork, hipEventDisableTiming) != hipSuccess) return false;
  for (int k = 0; k < nside; ++k) {
    if (p->side_stream[k]) continue;
    if (hipStreamCreateWithFlags(&p->side_stream[k], hipStreamNonBlocking) != hipSuccess) { p->side_stream[k] = nullptr; return false; }
    if (hipEventCreateWithFlags(&p->side_join[k], hipEventDisableTiming) != hipSuccess) return false;
  }
  return true;
}
int g_prog_parts = getenv("CSVAE_PROG_PARTS") ? atoi(getenv("CSVAE_PROG_PARTS")) : 2;   // (measurement switch, 2..4)

extern "C" int cs_debug_set_prog_two_streams(int min_rows) {
  const int prev = g_prog_two_streams;
  g_prog_two_streams = min_rows;
  return prev;
}

int cs_prog_launch(const CsMlpArgs& a, hipStream_t stream) {
  CsProgram* p = (CsProgram*)a.net.prog;
  int dev = 0;
  CS_HIP(hipGetDevice(&dev));
  CS_REQUIRE(dev == p->device, "network was created on another device");
  if (a.mode == CS_MLP_DEC) {
    const int np = cs_gen_nparams(a.gen_dist);
    CS_REQUIRE(a.net.out_dim == np * a.D && (np == 1 || a.net.split == np),
               "generator output must be D logits, (mean | raw scale) or (raw df | loc | raw scale)");
  }
  if (a.mode == CS_MLP_ENC) CS_REQUIRE(a.net.out_dim == 2 * a.Zd, "encoder output must be (loc | raw scale)");

  // workspace: [staged input | scratch 1..4 | raw output] per row, for a chunk of rows.  It only grows here when
  // the caller has not reserved it (cs_net_reserve): growing allocates, and frees after waiting for the device.
  int64_t per_row = 0, prefix[CS_NET_NBUF];
  {
    int64_t off = 0;
    for (int b = 0; b < CS_NET_NBUF; ++b) {
      prefix[b] = off;
      off += (p->buf_elems[b] + 3) & ~(int64_t)3;
    }
  }
  const int64_t chunk = prog_chunk_rows(p, a.rows, &per_row);
  if (int rc = prog_grow_workspace(p, chunk, per_row)) return rc;
  // fp16 split: slot i of the table of per-image maxima belongs to the output of layer i (written by the halo
  // kernel's epilogue), the last slot to the network input; a source whose producer did not write its slot gets a
  // reduction pass of its own
  const size_t nlay = p->layers.size();
  bool any_half = false;
  std::vector<char> wants(nlay, 0);  // a later fp16-split layer reads this layer's output
  for (const CsLayer& L : p->layers)
    if (layer_half_kind(*p, L) != 0) {
      any_half = true;
      if (L.src_layer >= 0) wants[(size_t)L.src_layer] = 1;
    }
  for (int64_t row0 = 0; row0 < a.rows; row0 += chunk) {
    const int64_t rows = a.rows - row0 < chunk ? a.rows - row0 : chunk;
    // Rows are independent, and every launch of a layer pays a ramp and a drain of about two workgroup lifetimes
    // (60 - 80 us for the halo kernels, whatever the number of images) during which most CUs wait.  A pass of enough
    // rows therefore runs as two halves, the second on a stream of the library's own: while one half's layer
    // drains, the other half's workgroups take the CUs.  Each half has its own part of every workspace buffer and
    // of the maxima table; the caller's stream waits for the second half before the call returns.
    int nh = 1;
    // (it pays from about 40 GFLOP per pass: below that the layers' launches themselves are the time, and a split
    // doubles them -- MNIST-GMM nets at 256 images 4 % slower, Omniglot nets at 64 images 4 - 10 % faster)
    double pass_flops = 0.0;
    for (const CsLayer& L : p->layers)
      if (L.d.op == CS_OP_CONV || L.d.op == CS_OP_LINEAR)
        pass_flops += 2.0 * L.d.ksize * L.d.ksize * L.d.cin * L.d.cout * L.d.hout * L.d.wout * (double)rows;
    // (a negative setting divides every pass of at least that many rows: tests)
    if ((g_prog_two_streams > 0 && rows >= g_prog_two_streams && nlay >= 4 && pass_flops >= 4.0e10) ||
        (g_prog_two_streams < 0 && rows >= -g_prog_two_streams)) {
      nh = g_prog_parts < 2 ? 2 : g_prog_parts > 4 ? 4 : g_prog_parts;
      if (!prog_side(p, nh - 1)) nh = 1;
    }
    const bool side = nh > 1;
    struct Half {
      int64_t row_begin, rows, ws_off;
      hipStream_t stream;
      float* buf[CS_NET_NBUF];
      const float* input;
      float* raw;
      std::vector<char> have;
    } h[4];
    const int64_t part = side ? (((rows + nh - 1) / nh + 3) & ~(int64_t)3) : rows;
    for (int k = 0; k < nh; ++k) {
      h[k].ws_off = k * part;
      h[k].row_begin = row0 + h[k].ws_off;
      h[k].rows = std::min<int64_t>(part, rows - h[k].ws_off);
      h[k].stream = k ? p->side_stream[k - 1] : stream;
      for (int b = 0; b < CS_NET_NBUF; ++b)
        h[k].buf[b] = p->ws + prefix[b] * chunk + h[k].ws_off * ((p->buf_elems[b] + 3) & ~(int64_t)3);
      h[k].have.assign(nlay + 1, 0);
    }
    while (nh > 1 && h[nh - 1].rows <= 0) --nh;
    if (any_half) CS_HIP(hipMemsetAsync(p->amax_ws, 0, (nlay + 1) * (size_t)p->ws_rows * sizeof(unsigned), stream));
    if (side) {
      CS_HIP(hipEventRecord(p->side_fork, stream));
      for (int k = 1; k < nh; ++k) CS_HIP(hipStreamWaitEvent(h[k].stream, p->side_fork, 0));
    }
    int rc_all = CS_OK;
    for (int k = 0; k < nh && rc_all == CS_OK; ++k) {
      Half& H = h[k];
      if (a.mode == CS_MLP_DEC && a.in == nullptr) {
        rc_all = cs_dec_stage_launch(a, H.row_begin, H.rows, H.stream);
        H.input = a.Z + H.row_begin * a.Zd;
      } else if (a.in_off == nullptr && a.in_stride == p->in_dim && !a.in_u8) {
        H.input = a.in + H.row_begin * p->in_dim;
      } else {
        hipLaunchKernelGGL(cs_gather_rows_kernel, dim3(blocks_for(H.rows * p->in_dim)), dim3(256), 0, H.stream, a.in,
                           a.in_stride, a.in_off, H.row_begin, H.rows, p->in_dim, H.buf[CS_NET_BUF_IN], a.in_u8);
        H.input = H.buf[CS_NET_BUF_IN];
      }
      H.raw = a.mode == CS_MLP_RAW ? a.out + H.row_begin * p->out_dim : H.buf[CS_NET_BUF_OUT];
    }
    for (size_t li = 0; li < nlay && rc_all == CS_OK; ++li) {
      const CsLayer& L = p->layers[li];
      const cs_layer_desc& d = L.d;
      for (int k = 0; k < nh && rc_all == CS_OK; ++k) {
        Half& H = h[k];
        const int64_t hrows = H.rows;
        hipStream_t hs = H.stream;
        const float* src = d.src == CS_NET_BUF_IN ? H.input : H.buf[d.src];
        const float* res = d.res < 0 ? nullptr : (d.res == CS_NET_BUF_IN ? H.input : H.buf[d.res]);
        float* dst = d.dst == CS_NET_BUF_OUT ? H.raw : H.buf[d.dst];
        if (d.op == CS_OP_CONV || d.op == CS_OP_LINEAR) {
          const unsigned* in_amax = nullptr;
          unsigned* out_amax = nullptr;
          const int kind = any_half ? layer_half_kind(*p, L) : 0;
          if (kind != 0) {
            const size_t slot = L.src_layer < 0 ? nlay : (size_t)L.src_layer;
            unsigned* tab = p->amax_ws + slot * (size_t)p->ws_rows + H.ws_off;
            if (!H.have[slot]) {
              const int64_t elems = (int64_t)d.cin * d.hin * d.win;
              const unsigned gx = (unsigned)std::min<int64_t>((elems / 4 + 255) / 256 > 0 ? (elems / 4 + 255) / 256 : 1, 16);
              hipLaunchKernelGGL(cs_image_amax_kernel, dim3(gx, (unsigned)std::min<int64_t>(hrows, 32768)), dim3(256), 0, hs,
                                 src, elems, hrows, tab);
              H.have[slot] = 1;
            }
            in_amax = tab;
          }
          if (any_half && wants[li]) out_amax = p->amax_ws + li * (size_t)p->ws_rows + H.ws_off;
          bool wrote = false;
          rc_all = launch_conv(*p, L, src, res, dst, hrows, hs, in_amax, out_amax, &wrote);
          if (wrote) H.have[li] = 1;
        } else if (d.op == CS_OP_LAYERNORM) {
          hipLaunchKernelGGL(cs_layernorm_kernel, dim3((unsigned)((hrows + 3) / 4)), dim3(256), 0, hs, src, dst, hrows,
                             d.cin, p->d_blob + d.w_off, p->d_blob + d.b_off, d.relu);
        } else if (d.op == CS_OP_AVGPOOL) {
          hipLaunchKernelGGL(cs_avgpool_kernel, dim3(blocks_for(hrows * d.cin)), dim3(256), 0, hs, src, dst, hrows,
                             d.hin * d.win, d.cin);
        } else {
          unsigned* oam = (any_half && wants[li] && g_conv_fused_amax) ? p->amax_ws + li * (size_t)p->ws_rows + H.ws_off : nullptr;
          const int per = d.hout * d.wout * d.cin;
          const unsigned gx = (unsigned)std::max(1, std::min(8, (per + 2047) / 2048));  // 8+ elements per thread
          hipLaunchKernelGGL(cs_resize_kernel, dim3(gx, (unsigned)std::min<int64_t>(hrows, 32768)), dim3(256), 0, hs,
                             src, dst, hrows, d.hin, d.win, d.hout, d.wout, d.cin, oam);
          if (oam) H.have[li] = 1;
        }
      }
    }
    for (int k = 0; k < nh && rc_all == CS_OK; ++k) {
      Half& H = h[k];
      if (a.mode == CS_MLP_ENC)
        hipLaunchKernelGGL(cs_enc_finish_kernel, dim3(blocks_for(H.rows * a.Zd)), dim3(256), 0, H.stream, a, H.raw,
                           H.row_begin, H.rows);
      else if (a.mode == CS_MLP_DEC)
        hipLaunchKernelGGL(cs_dec_finish_kernel, dim3((unsigned)H.rows), dim3(64), 0, H.stream, a, H.raw, H.row_begin);
    }
    for (int k = 1; k < nh; ++k) {  // also on an error: nothing of the other streams may outlive the call unobserved
      (void)hipEventRecord(p->side_join[k - 1], h[k].stream);
      (void)hipStreamWaitEvent(stream, p->side_join[k - 1], 0);
    }
    if (rc_all != CS_OK) return rc_all;
    CS_HIP(hipGetLastError());
  }
  return CS_OK;
}

#ifdef CS_CONV_STAMPS
extern "C" int cs_debug_conv_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cs_conv_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif
