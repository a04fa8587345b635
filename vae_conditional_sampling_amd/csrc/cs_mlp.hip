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

#include <cstdlib>
#include <cstring>
#include <vector>

int cs_mlp_launch_h64(const CsMlpArgs& a, hipStream_t stream);
int cs_mlp_launch_h128(const CsMlpArgs& a, hipStream_t stream);
int cs_mlp_launch_h256(const CsMlpArgs& a, hipStream_t stream);

static long long* g_trace = nullptr;
// Profiling hook: when set, every fused-MLP launch writes 16 int64 per wavefront (16-row tile)
// into buf: clock64() stamps at start, after prologue, after the first layer, after each block,
// after the output layer, end (slots 0..14) and where the wave ran (slot 15).  buf must hold
// 16 * ceil(rows/16) entries.  NULL disables it.
extern "C" int cs_debug_set_trace(long long* buf) {
  g_trace = buf;
  return CS_OK;
}

int cs_mlp_launch(const CsMlpArgs& a_in, hipStream_t stream) {
  CsMlpArgs a = a_in;
  a.dbg = g_trace;
  if (a.rows <= 0) return CS_OK;
  if (a.net.prog) return cs_prog_launch(a, stream);
  if (a.mode == CS_MLP_DEC && a.in == nullptr && a.latent_dist == CS_DIST_BETA) {
    // Beta proposals are drawn by a kernel of their own; the fused kernel then reads Z as its input
    if (int rc = cs_dec_stage_launch(a, 0, a.rows, stream)) return rc;
    a.in = a.Z;
    a.in_stride = a.Zd;
  }
  CS_REQUIRE(a.rows < ((int64_t)1 << 31) * 16, "too many rows");
  if (a.net.split == 2) CS_REQUIRE(a.net.nto == 2 * a.net.tph, "bad tile split");
  if (a.mode == CS_MLP_ENC) CS_REQUIRE(a.net.split == 2, "encoder output must be (loc | scale)");
  if (a.mode == CS_MLP_DEC && a.gen_dist != CS_DIST_BERN)
    CS_REQUIRE(a.net.split == 2, "generator output must be (loc | scale)");
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

  cs_mlp* m = new cs_mlp();
  m->desc = *d;
  m->d_w = nullptr;
  m->d_b = nullptr;
  m->prog = nullptr;
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
  m->dev.prog = nullptr;
  *out = m;
  return CS_OK;
}

extern "C" int cs_mlp_destroy(cs_mlp_t* m) {
  if (!m) return CS_OK;
  cs_prog_destroy(m->prog);
  if (m->d_w) (void)hipFree(m->d_w);
  if (m->d_b) (void)hipFree(m->d_b);
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
