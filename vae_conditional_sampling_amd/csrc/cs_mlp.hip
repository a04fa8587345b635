// Host side of the fused residual-MLP kernel: operand-stream packing, handle management and
// launch dispatch.  The device code is in cs_mlp_kernel.inc (instantiated per hidden width in
// cs_mlp_h64/h128/h256.hip).
//
// Replaces ResidualFCNetwork.forward (irwg/models/neural_nets.py:163-169, blocks :87-103):
//   h = relu(W0 x + b0);  per block  h = relu(h + W2 relu(W1 h + b1) + b2);  y = Wf h + bf
// for the encoder (model.var_latent_network) and decoder (model.generator_network) of the
// MLP VAEs, with the distribution parameterisation of VAE.create_distribution
// (irwg/models/vae.py:203-256) and the masked log-likelihood of weights.py:46-47 fused into
// the epilogue, and the proposal sampling Z = loc + scale*eps (vae_imputation.py:596-597)
// fused into the decoder prologue.
#include "cs_mlp.h"
#include "cs_net.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

int cs_mlp_launch_h64(const CsMlpArgs& a, hipStream_t stream);
int cs_mlp_launch_h128(const CsMlpArgs& a, hipStream_t stream);
int cs_mlp_launch_h256(const CsMlpArgs& a, hipStream_t stream);
int cs_mlp_split_launch_h64(const CsMlpArgs& a, hipStream_t stream);
int cs_mlp_split_launch_h128(const CsMlpArgs& a, hipStream_t stream);
int cs_mlp_split_launch_h256(const CsMlpArgs& a, hipStream_t stream);
int cs_mlp_split16_launch_h64(const CsMlpArgs& a, hipStream_t stream, int waves, int terms);
int cs_mlp_split16_launch_h128(const CsMlpArgs& a, hipStream_t stream, int waves, int terms);
int cs_mlp_split16_launch_h256(const CsMlpArgs& a, hipStream_t stream, int waves, int terms);
bool cs_mlp_narrow_supported(const CsMlpArgs& a);
int cs_mlp_narrow_launch(const CsMlpArgs& a, hipStream_t stream);

// Row tile of the split-bf16 arithmetic: 32 (cs_mlp_split_kernel.inc, one wave per SIMD) or 16
// (cs_mlp_split16_kernel.inc, two waves per SIMD / half the dependent chain per row tile).
// CSVAE_SPLIT_TILE = 32 | 16 | auto (default): auto takes the 16-row kernel whenever its three-slot ring fits the
// LDS (measured faster at every batch size, MiniBooNE shape, one LAIR step: 0.189 vs 0.243 ms at 512 examples --
// the time of a small batch is the dependent chain of one row tile through the network, half as long with 16
// rows -- and 2.37 vs 2.52 ms at 65 536: tools/sweep_tile.py), in 4-wave workgroups of 64 rows up to 20 480 rows
// (320 workgroups) and 8-wave workgroups, two waves per SIMD, above.  CSVAE_SPLIT16_WAVES = 4 | 8 overrides the
// workgroup width.
static int g_split_tile = -1, g_split16_waves = -1;
static void cs_split_cfg() {
  if (g_split_tile >= 0) return;
  const char* e = std::getenv("CSVAE_SPLIT_TILE");
  g_split_tile = (e && std::strcmp(e, "16") == 0) ? 16 : (e && std::strcmp(e, "32") == 0) ? 32 : 0;
  const char* w = std::getenv("CSVAE_SPLIT16_WAVES");
  g_split16_waves = (w && std::strcmp(w, "4") == 0) ? 4 : (w && std::strcmp(w, "8") == 0) ? 8 : 0;
}

// Arithmetic of the fused-MLP matrix products, all but the last float32-equivalent (measured as accurate against a
// float64 evaluation as the exact-f32 kernel):
//   the 2-way fp16 split (3 fp16 matrix instructions per product, rows and weights normalised by exact powers of two;
//   the 16-row kernel with two terms per operand) -- the default;
//   the 3-way bf16 split (6 bf16 matrix instructions per product; float32's exponent range in every term): what the
//   default falls back to for a network with a non-finite weight, and the arithmetic of the layer programs;
//   exact f32 matrix instructions;
//   and -- opt-in only, never a default -- plain bf16 (operands rounded to bf16, one matrix instruction per product,
//   f32 accumulation; errors of the size of bf16 rounding, 2^-9 relative per operand).
// -1 = not decided yet: the environment variable CSVAE_MLP_ARITH (f16x2 | bf16x3 | f32 | bf16) is read at first use.
static int g_arith = -1;
int cs_mlp_arith() {
  if (g_arith < 0) {
    const char* e = std::getenv("CSVAE_MLP_ARITH");
    g_arith = (e && std::strcmp(e, "f32") == 0) ? CS_ARITH_F32 : (e && std::strcmp(e, "bf16") == 0) ? CS_ARITH_BF16
            : (e && std::strcmp(e, "bf16x3") == 0) ? CS_ARITH_BF16X3 : CS_ARITH_F16X2;
  }
  return g_arith;
}
extern "C" int cs_set_mlp_arith(int arith) {
  CS_REQUIRE(arith == CS_ARITH_F32 || arith == CS_ARITH_BF16X3 || arith == CS_ARITH_BF16 || arith == CS_ARITH_F16X2,
             "unknown arithmetic");
  g_arith = arith;
  return CS_OK;
}
extern "C" int cs_get_mlp_arith(void) { return cs_mlp_arith(); }
extern "C" int cs_mlp_set_arith(cs_mlp_t* net, int arith) {
  CS_REQUIRE(net, "null argument");
  CS_REQUIRE(arith == CS_ARITH_DEFAULT || arith == CS_ARITH_F32 || arith == CS_ARITH_BF16X3 || arith == CS_ARITH_BF16 ||
                 arith == CS_ARITH_F16X2,
             "unknown arithmetic");
  net->dev.arith = arith;
  if (net->prog) net->prog->arith = arith;
  return CS_OK;
}
extern "C" int cs_mlp_get_arith(const cs_mlp_t* net) {
  return net && net->dev.arith >= 0 ? net->dev.arith : cs_mlp_arith();
}

// -1: the small-launch kernel for launches of up to CS_NARROW_MAX_TILES row tiles; 0: never; 1: whenever an instance
// exists (tests).  CSVAE_NARROW gives the initial value.
#define CS_NARROW_MAX_TILES 512
static int g_narrow = [] {
  const char* e = std::getenv("CSVAE_NARROW");
  return (e && std::strcmp(e, "0") == 0) ? 0 : (e && std::strcmp(e, "1") == 0) ? 1 : -1;
}();
extern "C" int cs_debug_set_narrow(int mode) {
  const int prev = g_narrow;
  g_narrow = mode < 0 ? -1 : (mode > 0 ? 1 : 0);
  return prev;
}

static long long* g_trace = nullptr;
// Profiling hook: when set, every fused-MLP launch writes 16 int64 per wavefront (16-row tile)
// into buf: clock64() stamps at start, after prologue, after the first layer, after each block,
// after the output layer, end (slots 0..14) and where the wave ran (slot 15).  buf must hold
// 16 * ceil(rows/16) entries.  NULL disables it.
extern "C" int cs_debug_set_trace(long long* buf) {
  g_trace = buf;
  return CS_OK;
}

// LDS footprint of the 16-row kernel: bias table + three ring slots (cs_mlp_split16_launch_w)
static bool cs_split16_fits(const CsMlpDev& n, int waves, int terms) {
  const size_t nthr = 64 * (size_t)waves, mx = (size_t)(n.nt > n.nto ? n.nt : n.nto);
  const size_t slot_f4 = (mx * terms * 64 + nthr - 1) / nthr * nthr;
  const size_t bias = (((size_t)(2 * n.nb + 2) * n.bias_stride * 4) + 1023) & ~(size_t)1023;
  return bias + 3 * slot_f4 * 16 <= 160 * 1024;
}

int cs_mlp_launch(const CsMlpArgs& a_in, hipStream_t stream) {
  CsMlpArgs a = a_in;
  a.dbg = g_trace;
  if (a.rows <= 0) return CS_OK;
  if (a.net.prog) return cs_prog_launch(a, stream);
  CS_REQUIRE(!a.in_u8, "uint8 input rows (binary history ring) need a layer-program network");
  if (a.mode == CS_MLP_DEC)
    CS_REQUIRE(a.gen_dist >= CS_DIST_NORMAL && a.gen_dist <= CS_DIST_BERN,
               "Student-t / logit-normal generators need a layer-program decoder");
  cs_split_cfg();
  int arith = a.net.arith >= 0 ? a.net.arith : cs_mlp_arith();
  const int waves16 = g_split16_waves ? g_split16_waves : (a.rows <= 20480 ? 4 : 8);
  // The process-wide fp16 split covers every network it can: one with a non-finite weight (no fp16 stream) or beyond
  // the kernel's LDS budget runs in the bf16 split, which has float32's range.  A handle that asks for the fp16
  // split itself (cs_mlp_set_arith) is refused instead.
  if (arith == CS_ARITH_F16X2 && a.net.arith < 0 && (a.net.wstream16h == nullptr || !cs_split16_fits(a.net, waves16, 2)))
    arith = CS_ARITH_BF16X3;
  const bool tile16 = arith == CS_ARITH_BF16 || arith == CS_ARITH_F16X2 ||
                      (arith == CS_ARITH_BF16X3 && g_split_tile != 32 && cs_split16_fits(a.net, waves16, 3));
  if (arith == CS_ARITH_BF16) CS_REQUIRE(cs_split16_fits(a.net, waves16, 1), "network does not fit the LDS budget");
  if (arith == CS_ARITH_F16X2) {
    CS_REQUIRE(cs_split16_fits(a.net, waves16, 2), "network does not fit the LDS budget");
    CS_REQUIRE(a.net.wstream16h != nullptr, "a weight of this network is not finite: no fp16-split arithmetic");
  }
  const int terms16 = arith == CS_ARITH_BF16 ? 1 : (arith == CS_ARITH_F16X2 ? 2 : 3);
  if (a.mode == CS_MLP_DEC && a.in == nullptr &&
      (a.latent_dist == CS_DIST_BETA || (arith == CS_ARITH_BF16X3 && !tile16))) {
    // Beta proposals are drawn by a kernel of their own; the fused kernel then reads Z as its input.
    // The split-bf16 kernel runs one wavefront per SIMD, so nothing would cover the counter-RNG work of
    // its prologue: there too the proposals are drawn by the full-occupancy kernel first.
    if (int rc = cs_dec_stage_launch(a, 0, a.rows, stream)) return rc;
    a.in = a.Z;
    a.in_stride = a.Zd;
  }
  CS_REQUIRE(a.rows < ((int64_t)1 << 31) * 16, "too many rows");
  if (a.net.split == 2) CS_REQUIRE(a.net.nto == 2 * a.net.tph, "bad tile split");
  if (a.mode == CS_MLP_ENC) CS_REQUIRE(a.net.split == 2, "encoder output must be (loc | scale)");
  if (a.mode == CS_MLP_DEC && a.gen_dist != CS_DIST_BERN)
    CS_REQUIRE(a.net.split == 2, "generator output must be (loc | scale)");
  // Small launches of the fp16 split (the reference configurations' own batch sizes): one row tile per workgroup,
  // the layer's output features split over its wavefronts (cs_mlp_narrow_kernel.inc).  Up to 512 row tiles: two
  // workgroups per CU; beyond that the kernels that keep a row tile on one wavefront use the chip better.
  if (arith == CS_ARITH_F16X2 && g_narrow != 0 && !a.dbg && cs_mlp_narrow_supported(a) &&
      (g_narrow == 1 || a.rows <= (int64_t)CS_NARROW_MAX_TILES * 16))
    return cs_mlp_narrow_launch(a, stream);
  if (a.mode == CS_MLP_ENC) CS_REQUIRE(a.net.split == 2, "encoder output must be (loc | scale)");
  if (a.mode == CS_MLP_DEC && a.gen_dist != CS_DIST_BERN)
    CS_REQUIRE(a.net.split == 2, "generator output must be (loc | scale)");
  if (tile16) {
    switch (a.net.nt) {
      case 4: return cs_mlp_split16_launch_h64(a, stream, waves16, terms16);
      case 8: return cs_mlp_split16_launch_h128(a, stream, waves16, terms16);
      case 16: return cs_mlp_split16_launch_h256(a, stream, waves16, terms16);
    }
    return cs_fail(CS_EUNSUPPORTED, "no kernel instance for hidden=" + std::to_string(a.net.nt * 16));
  }
  if (arith == CS_ARITH_BF16X3) {
    switch (a.net.nt) {
      case 4: return cs_mlp_split_launch_h64(a, stream);
      case 8: return cs_mlp_split_launch_h128(a, stream);
      case 16: return cs_mlp_split_launch_h256(a, stream);
    }
  }
  switch (a.net.nt) {
    case 4: return cs_mlp_launch_h64(a, stream);
    case 8: return cs_mlp_launch_h128(a, stream);
    case 16: return cs_mlp_launch_h256(a, stream);
  }
  return cs_fail(CS_EUNSUPPORTED, "no kernel instance for hidden=" + std::to_string(a.net.nt * 16));
}

// ----------------------------------------------------------------------------------------
// host: packing
// ----------------------------------------------------------------------------------------
namespace {
// Append one layer (row-major W[n_real][k_real]) to the operand stream: per 32-wide k-tile, per
// 16-wide half, per 16-row output tile, per lane (i = lane&15, g = lane>>4): the 4 consecutive
// columns k = 32kt + 16ft2 + 4g + 0..3 of row tile_row(t, i) (-1 = padding).
template <class RowFn>
void pack_layer(std::vector<float>& s, const float* W, int k_real, int kt_count, int ntiles, RowFn tile_row) {
  for (int kt = 0; kt < kt_count; ++kt)
    for (int ft2 = 0; ft2 < 2; ++ft2)
      for (int nt = 0; nt < ntiles; ++nt)
        for (int lane = 0; lane < 64; ++lane) {
          const int g = lane >> 4, i = lane & 15;
          const int n = tile_row(nt, i);
          for (int r = 0; r < 4; ++r) {
            const int k = 32 * kt + 16 * ft2 + 4 * g + r;
            s.push_back((n >= 0 && k < k_real) ? W[(size_t)n * k_real + k] : 0.0f);
          }
        }
}

// round-to-nearest-even bf16 of a finite float, as the 16 high bits
uint16_t bf16_rne(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
float bf16_val(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
// Split-bf16 operand stream of one layer: per 32-wide k-tile, per k-step of 16, per 32-row output
// tile, per term p of the 3-way split, per lane (m = lane&31, h = lane>>5), the 8 columns
// k = 32kt + 16ks + 8(j>>2) + 4h + (j&3) of row tile_row(t, m) (-1 = padding).
template <class RowFn>
void pack_layer3(std::vector<uint16_t>& s, const float* W, int k_real, int kt_count, int ntiles, RowFn tile_row) {
  for (int kt = 0; kt < kt_count; ++kt)
    for (int ks = 0; ks < 2; ++ks)
      for (int t = 0; t < ntiles; ++t)
        for (int p = 0; p < 3; ++p)
          for (int lane = 0; lane < 64; ++lane) {
            const int m = lane & 31, h = lane >> 5;
            const int n = tile_row(t, m);
            for (int j = 0; j < 8; ++j) {
              const int k = 32 * kt + 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
              const float v = (n >= 0 && k < k_real) ? W[(size_t)n * k_real + k] : 0.0f;
              const uint16_t t0 = bf16_rne(v);
              const float r1 = v - bf16_val(t0);
              const uint16_t t1 = bf16_rne(r1);
              const uint16_t t2 = bf16_rne(r1 - bf16_val(t1));
              s.push_back(p == 0 ? t0 : (p == 1 ? t1 : t2));
            }
          }
}
// 16-row-tile split-bf16 operand stream of one layer: per k-step of 32 features, per 16-row output tile, per
// term p of the 3-way split, per lane (m = lane&15, g = lane>>4), the 8 columns
// k = 32s + 16(j>>2) + 4g + (j&3) of row tile_row(t, m) (-1 = padding): k-slot 8g + j of the matrix instruction
// carries the feature that a lane of group g holds in register j&3 of tile 2s + (j>>2) of the previous layer.
// fp16 of a float, round to nearest even (the compiler's conversion), as its 16 bits
uint16_t f16_rne(float f) {
  const _Float16 h = (_Float16)f;
  uint16_t u;
  std::memcpy(&u, &h, 2);
  return u;
}
float f16_val(uint16_t u) {
  _Float16 h;
  std::memcpy(&h, &u, 2);
  return (float)h;
}
// The same order with two fp16 terms per weight (w0 = fp16(w'), w1 = fp16(w' - w0)), w' = w 2^kexp: the stream of
// the fp16-split arithmetic.  kexp (returned through *kexp_out) is chosen per layer so that the largest weight lands in
// [2^14, 2^15): the high terms use fp16's range and the low terms are normal numbers down to weights 2^-17 of the
// largest; the kernel carries the exponent in its row scaling.  Returns false when a weight is not finite.
template <class RowFn>
bool pack_layer16h(std::vector<uint16_t>& s, const float* W, int n_real, int k_real, int ks_count, int ntiles, RowFn tile_row,
                   signed char* kexp_out) {
  bool ok = true;
  float mx = 0.0f;
  for (size_t i = 0; i < (size_t)n_real * k_real; ++i) {
    if (!std::isfinite(W[i])) ok = false;
    mx = std::fmax(mx, std::fabs(W[i]));
  }
  int ex = 0;
  if (mx > 0.0f && std::isfinite(mx)) (void)std::frexp(mx, &ex);  // mx = m 2^ex, m in [0.5, 1)
  int kexp = mx > 0.0f ? 15 - ex : 0;                              // mx 2^kexp in [2^14, 2^15)
  kexp = kexp < -100 ? -100 : (kexp > 100 ? 100 : kexp);
  *kexp_out = (signed char)kexp;
  for (int ks = 0; ks < ks_count; ++ks)
    for (int t = 0; t < ntiles; ++t)
      for (int p = 0; p < 2; ++p)
        for (int lane = 0; lane < 64; ++lane) {
          const int m = lane & 15, g = lane >> 4;
          const int n = tile_row(t, m);
          for (int j = 0; j < 8; ++j) {
            const int k = 32 * ks + 16 * (j >> 2) + 4 * g + (j & 3);
            const float v = (n >= 0 && k < k_real) ? std::ldexp(W[(size_t)n * k_real + k], kexp) : 0.0f;
            const uint16_t t0 = f16_rne(v);
            s.push_back(p == 0 ? t0 : f16_rne(v - f16_val(t0)));
          }
        }
  return ok;
}
// terms = 1: the plain-bf16 stream, one fragment (the rounded weights) per tile.
template <class RowFn>
void pack_layer16(std::vector<uint16_t>& s, const float* W, int k_real, int ks_count, int ntiles, RowFn tile_row,
                  int terms = 3) {
  for (int ks = 0; ks < ks_count; ++ks)
    for (int t = 0; t < ntiles; ++t)
      for (int p = 0; p < terms; ++p)
        for (int lane = 0; lane < 64; ++lane) {
          const int m = lane & 15, g = lane >> 4;
          const int n = tile_row(t, m);
          for (int j = 0; j < 8; ++j) {
            const int k = 32 * ks + 16 * (j >> 2) + 4 * g + (j & 3);
            const float v = (n >= 0 && k < k_real) ? W[(size_t)n * k_real + k] : 0.0f;
            const uint16_t t0 = bf16_rne(v);
            const float r1 = v - bf16_val(t0);
            const uint16_t t1 = bf16_rne(r1);
            const uint16_t t2 = bf16_rne(r1 - bf16_val(t1));
            s.push_back(p == 0 ? t0 : (p == 1 ? t1 : t2));
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
  const int tph = (out_half + 15) / 16;   // 16-feature output tiles per half
  int nto = d->out_split * tph;
  CS_REQUIRE(nto <= 8, "output too wide (each half must be <= 64, or <= 128 unsplit)");
  if (nto & 1) nto += 1;                   // tiles are processed in pairs
  const int NT = H / 16;
  const int NKT = H / 32;
  const int nkt0 = (d->in_dim + 31) / 32;
  const int bst = 16 * (NT > nto ? NT : nto);

  std::vector<float> s;
  auto hid_row = [&](int t, int i) { return 16 * t + i; };
  pack_layer(s, W0, d->in_dim, nkt0, NT, hid_row);
  for (int b = 0; b < nb; ++b)
    for (int l = 0; l < 2; ++l) pack_layer(s, Wblk + ((size_t)b * 2 + l) * H * H, H, NKT, NT, hid_row);
  auto out_row = [&](int t, int i) {
    const int half = t / tph, w = 16 * (t - half * tph) + i;
    return (half < d->out_split && w < out_half) ? half * out_half + w : -1;
  };
  pack_layer(s, Wf, H, NKT, nto, out_row);
  s.resize(s.size() + (size_t)2 * (NT > nto ? NT : nto) * 64 * 4, 0.0f);  // the last prefetch reads one slot past the end

  const int nl = 2 * nb + 2;
  std::vector<float> bias((size_t)nl * bst, 0.0f);
  for (int i = 0; i < H; ++i) bias[i] = b0[i];
  for (int b = 0; b < nb; ++b)
    for (int l = 0; l < 2; ++l)
      for (int i = 0; i < H; ++i) bias[(size_t)(1 + 2 * b + l) * bst + i] = bblk[((size_t)b * 2 + l) * H + i];
  for (int t = 0; t < nto; ++t)
    for (int i = 0; i < 16; ++i) {
      const int n = out_row(t, i);
      if (n >= 0) bias[(size_t)(nl - 1) * bst + 16 * t + i] = bf[n];
    }

  // split-bf16 variant: 32-row tiles; a (loc | scale) output tile holds 16 loc rows then the 16 scale rows
  // of the same feature indices
  const int NT3 = H / 32;
  const int nto3 = d->out_split == 2 ? (out_half + 15) / 16 : (d->out_dim + 31) / 32;
  const int bst3 = 32 * (NT3 > nto3 ? NT3 : nto3);
  std::vector<uint16_t> s3;
  auto hid_row3 = [&](int t, int mrow) { return 32 * t + mrow; };
  auto out_row3 = [&](int t, int mrow) {
    if (d->out_split == 2) {
      const int w = 16 * t + (mrow & 15);
      return w < out_half ? (mrow >> 4) * out_half + w : -1;
    }
    const int n = 32 * t + mrow;
    return n < d->out_dim ? n : -1;
  };
  pack_layer3(s3, W0, d->in_dim, nkt0, NT3, hid_row3);
  for (int b = 0; b < nb; ++b)
    for (int l = 0; l < 2; ++l) pack_layer3(s3, Wblk + ((size_t)b * 2 + l) * H * H, H, NKT, NT3, hid_row3);
  pack_layer3(s3, Wf, H, NKT, nto3, out_row3);
  s3.resize(s3.size() + ((size_t)2 * (NT3 > nto3 ? NT3 : nto3) * 3 * 64 + 256) * 8, 0);  // the last prefetch reads one (rounded) slot past the end
  std::vector<float> bias3((((size_t)nl * bst3 + 1023) / 1024) * 1024, 0.0f);  // whole 4 KB copy rounds
  for (int i = 0; i < H; ++i) bias3[i] = b0[i];
  for (int b = 0; b < nb; ++b)
    for (int l = 0; l < 2; ++l)
      for (int i = 0; i < H; ++i) bias3[(size_t)(1 + 2 * b + l) * bst3 + i] = bblk[((size_t)b * 2 + l) * H + i];
  for (int t = 0; t < nto3; ++t)
    for (int i = 0; i < 32; ++i) {
      const int n = out_row3(t, i);
      if (n >= 0) bias3[(size_t)(nl - 1) * bst3 + 32 * t + i] = bf[n];
    }

  // 16-row-tile split variant: the tile geometry (hid_row / out_row, nto, bias table) of the exact-f32 kernel
  std::vector<uint16_t> s16;
  pack_layer16(s16, W0, d->in_dim, nkt0, NT, hid_row);
  for (int b = 0; b < nb; ++b)
    for (int l = 0; l < 2; ++l) pack_layer16(s16, Wblk + ((size_t)b * 2 + l) * H * H, H, NKT, NT, hid_row);
  pack_layer16(s16, Wf, H, NKT, nto, out_row);
  s16.resize(s16.size() + ((size_t)2 * (NT > nto ? NT : nto) * 3 * 64 + 1024) * 8, 0);  // trailing prefetch: a rounded slot
  std::vector<uint16_t> s16b;  // plain-bf16 arithmetic: one term
  pack_layer16(s16b, W0, d->in_dim, nkt0, NT, hid_row, 1);
  for (int b = 0; b < nb; ++b)
    for (int l = 0; l < 2; ++l) pack_layer16(s16b, Wblk + ((size_t)b * 2 + l) * H * H, H, NKT, NT, hid_row, 1);
  pack_layer16(s16b, Wf, H, NKT, nto, out_row, 1);
  s16b.resize(s16b.size() + ((size_t)2 * (NT > nto ? NT : nto) * 64 + 1024) * 8, 0);
  std::vector<uint16_t> s16h;  // fp16-split arithmetic: two fp16 terms (left out when a weight exceeds fp16's range)
  signed char wexp[36] = {};
  bool h_ok = pack_layer16h(s16h, W0, H, d->in_dim, nkt0, NT, hid_row, &wexp[0]);
  for (int b = 0; b < nb; ++b)
    for (int l = 0; l < 2; ++l)
      h_ok = pack_layer16h(s16h, Wblk + ((size_t)b * 2 + l) * H * H, H, H, NKT, NT, hid_row, &wexp[1 + 2 * b + l]) && h_ok;
  h_ok = pack_layer16h(s16h, Wf, d->out_dim, H, NKT, nto, out_row, &wexp[2 * nb + 1]) && h_ok;
  s16h.resize(s16h.size() + ((size_t)3 * (NT > nto ? NT : nto) * 2 * 64 + 1024) * 8, 0);  // the kernel copies two periods ahead

  cs_mlp* m = new cs_mlp();
  m->desc = *d;
  m->d_w16 = nullptr;
  m->d_w16b = nullptr;
  m->d_w16h = nullptr;
  m->d_w = nullptr;
  m->d_b = nullptr;
  m->d_w3 = nullptr;
  m->d_b3 = nullptr;
  m->prog = nullptr;
  hipError_t e = hipGetDevice(&m->device);
  if (e == hipSuccess) e = hipMalloc(&m->d_w, s.size() * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&m->d_b, bias.size() * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(m->d_w, s.data(), s.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(m->d_b, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&m->d_w3, s3.size() * sizeof(uint16_t));
  if (e == hipSuccess) e = hipMalloc(&m->d_b3, bias3.size() * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(m->d_w3, s3.data(), s3.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(m->d_b3, bias3.data(), bias3.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&m->d_w16, s16.size() * sizeof(uint16_t));
  if (e == hipSuccess) e = hipMemcpy(m->d_w16, s16.data(), s16.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&m->d_w16b, s16b.size() * sizeof(uint16_t));
  if (e == hipSuccess) e = hipMemcpy(m->d_w16b, s16b.data(), s16b.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
  if (e == hipSuccess && h_ok) e = hipMalloc(&m->d_w16h, s16h.size() * sizeof(uint16_t));
  if (e == hipSuccess && h_ok) e = hipMemcpy(m->d_w16h, s16h.data(), s16h.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (m->d_w16) (void)hipFree(m->d_w16);
    if (m->d_w16b) (void)hipFree(m->d_w16b);
    if (m->d_w16h) (void)hipFree(m->d_w16h);
    if (m->d_w) (void)hipFree(m->d_w);
    if (m->d_b) (void)hipFree(m->d_b);
    if (m->d_w3) (void)hipFree(m->d_w3);
    if (m->d_b3) (void)hipFree(m->d_b3);
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
  m->dev.prog = nullptr;
  m->dev.wstream3 = (const f32x4*)m->d_w3;
  m->dev.bias3 = (const float*)m->d_b3;
  m->dev.nto3 = nto3;
  m->dev.bias_stride3 = bst3;
  m->dev.wstream16 = (const f32x4*)m->d_w16;
  m->dev.wstream16b = (const f32x4*)m->d_w16b;
  m->dev.wstream16h = (const f32x4*)m->d_w16h;
  std::memcpy(m->dev.wexp16h, wexp, sizeof(wexp));
  m->dev.arith = -1;
  *out = m;
  return CS_OK;
}

extern "C" int cs_mlp_destroy(cs_mlp_t* m) {
  if (!m) return CS_OK;
  cs_prog_destroy(m->prog);
  if (m->d_w) (void)hipFree(m->d_w);
  if (m->d_b) (void)hipFree(m->d_b);
  if (m->d_w3) (void)hipFree(m->d_w3);
  if (m->d_b3) (void)hipFree(m->d_b3);
  if (m->d_w16) (void)hipFree(m->d_w16);
  if (m->d_w16b) (void)hipFree(m->d_w16b);
  if (m->d_w16h) (void)hipFree(m->d_w16h);
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
