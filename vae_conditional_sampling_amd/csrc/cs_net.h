// Layer-program networks (convolutional ResNets, wide MLPs) -- internal interface.
//
// A network that does not fit the register-resident fused MLP kernel is a list of layers
// executed one kernel per layer over NHWC activations in a per-handle workspace.
#pragma once
#include <vector>

#include "cs_mlp.h"

// How a nearest-neighbour-upsampled (x2) convolution runs on the matrix kernels.  Output pixel (2i + a, 2j + b) of
// a 3x3 convolution over the upsampled image reads input rows {i - 1 + a, i + a} and columns {j - 1 + b, j + b} only,
// two of its three kernel rows (columns) falling on the same input row (column): the layer is four 2x2 convolutions
// of the input at its own resolution, one per output phase (a, b), with the coinciding taps' weights summed
// (16 instead of 36 tap products per 2x2 outputs).  A 1x1 convolution is computed once per input pixel and
// written to the four output pixels.
enum CsUpsampleMode { CS_UP_NONE = 0, CS_UP_REPLICATE = 1, CS_UP_PHASES = 4 };

struct CsLayer {
  cs_layer_desc d;
  int64_t wp_off;  // offset (floats) of the MFMA-packed weights in the device blob, -1: direct kernel
  int64_t wp3_off; // offset (bf16 entries) of the split-bf16 packed weights in d_blob3, -1: none
  int up_mode = CS_UP_NONE;
  int64_t wph_off[4] = {-1, -1, -1, -1};   // CS_UP_PHASES: packed weights of phase 2a + b (as wp_off / wp3_off)
  int64_t wph3_off[4] = {-1, -1, -1, -1};
  // fp16-split packed weights (two fp16 terms of w 2^wexp) in d_blob3h, -1: none (a weight is not finite)
  int64_t wp3h_off = -1;
  int64_t wph3h_off[4] = {-1, -1, -1, -1};
  int wexp = 0, wphexp[4] = {0, 0, 0, 0};
  int src_layer = -1;  // index of the layer that wrote this layer's source buffer, -1: the network input
};

struct CsProgram {
  std::vector<CsLayer> layers;
  float* d_blob = nullptr;   // caller's blob followed by the packed weights
  uint16_t* d_blob3 = nullptr;  // split-bf16 packed weights of the same-size 3x3 convolutions
  uint16_t* d_blob3h = nullptr; // fp16-split packed weights
  // per-image maxima of the activations (fp16 split): [layers + 1][ws_rows] words, slot i = output of layer i, the
  // last slot = the network input; zeroed at the start of every pass
  unsigned* amax_ws = nullptr;
  // second stream of a pass that runs as two halves (cs_prog_launch), created on first use
  hipStream_t side_stream[3] = {nullptr, nullptr, nullptr};
  hipEvent_t side_fork = nullptr, side_join[3] = {nullptr, nullptr, nullptr};
  int in_dim = 0, out_dim = 0;
  int64_t buf_elems[CS_NET_NBUF] = {0};  // floats per network row of every buffer
  // workspace (scratch buffers 1..4, staged input, raw output), grown on demand
  float* ws = nullptr;
  int64_t ws_rows = 0;
  int device = 0;
  int arith = -1;  // CS_ARITH_* of this network (cs_mlp_set_arith), or -1: the process default
};

// Runs the program in the role described by `a` (RAW / ENC / DEC, same meaning as for the
// fused kernel).
int cs_prog_launch(const CsMlpArgs& a, hipStream_t stream);
void cs_prog_destroy(CsProgram* p);
// Proposal sampling Z = sample(q) for rows [row0, row0+rows) of a DEC launch, as a kernel of its
// own (the layer programs, and posterior families the fused kernel's prologue does not sample).
int cs_dec_stage_launch(const CsMlpArgs& a, int64_t row0, int64_t rows, hipStream_t stream);
