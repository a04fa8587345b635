// Fused whole-network residual-MLP kernel (exact-f32 MFMA) -- internal interface.
#pragma once
#include "cs_common.h"

// Device-side description of a packed network.
struct CsMlpDev {
  const f32x4* wstream;  // packed MFMA A-operand stream (see cs_mlp.hip)
  const float* bias;     // [2*nb+2][bias_stride], final layer in tile order
  int nb;                // residual blocks
  int nkt0;              // first-layer k-tiles = ceil(in_dim/32)
  int in_dim, out_dim;
  int out_half;          // out_dim / out_split
  int tph;               // 16-feature output tiles per half = ceil(out_half/16)
  int nt;                // hidden/16
  int nto;               // output tiles = out_split*tph
  int split;
  int bias_stride;
  const void* prog;      // host pointer to a CsProgram when the network is a layer program, else null
  // split-bf16 variant (cs_mlp_split_kernel.inc): weights as three bf16 terms in 32-row tile order
  const f32x4* wstream3; // packed bf16x8 fragments (16 bytes each)
  const float* bias3;    // [2*nb+2][bias_stride3], final layer in 32-row tile order
  int nto3;              // output tiles of 32
  int bias_stride3;
  // 16-row-tile split-bf16 variant (cs_mlp_split16_kernel.inc): the same three bf16 terms in 16-row tile order;
  // bias table and output tile geometry are those of the exact-f32 kernel (bias, nto, tph)
  const f32x4* wstream16;
  const f32x4* wstream16b;  // the same order with one term, the weights rounded to bf16 (CS_ARITH_BF16)
  const f32x4* wstream16h;  // the same order with two fp16 terms (CS_ARITH_F16X2); null when a weight is not finite
  signed char wexp16h[36];  // per layer (first, block layers, output): the power of two its fp16 stream is scaled by
  int arith;             // CS_ARITH_* of this network, or -1: the process default (cs_set_mlp_arith)
};

struct CsProgram;
struct cs_mlp {
  cs_mlp_desc desc;
  CsMlpDev dev;
  void* d_w;
  void* d_b;
  void* d_w3;
  void* d_b3;
  void* d_w16;
  void* d_w16b;
  void* d_w16h;
  int device;
  CsProgram* prog;  // layer-program networks (cs_net_create); null for the fused MLP
};

enum CsMlpMode : int { CS_MLP_RAW = 0, CS_MLP_ENC = 1, CS_MLP_DEC = 2 };

struct CsMlpArgs {
  CsMlpDev net;
  int mode;
  int64_t rows;
  // RAW / ENC input: row r at in + r*in_stride, or in + in_off[r] when in_off != nullptr
  const float* in;
  int64_t in_stride;
  const int64_t* in_off;
  int in_u8;  // 1: `in` points to uint8_t rows (0 / 1 values, element offsets in in_off); layer programs only
  // RAW output [rows,out_dim]
  float* out;
  // ENC: rows are (b,k), k < K; q[(b*Kp+k)*2*Zd + {0,Zd} + z]
  // DEC: rows are (i,b,k'), k' < Kp
  float* q;
  int K, Kp, Zd, B, I, D;
  int latent_dist; // CS_DIST_* of the latent posterior (ENC epilogue, DEC proposal sampling)
  int gen_dist;    // CS_DIST_*
  // DEC prologue
  const float* eps_z;         // [rows,Zd] or nullptr -> Philox
  const uint8_t* z_override;  // optional [rows]: 1 -> take Z from z_alt instead
  const float* z_alt;         // [rows,Zd]
  CsPhilox rng;
  float* Z;                   // [rows,Zd] out
  // DEC epilogue
  float* gen;        // [rows,2,D] (normal) or [rows,D] (bern)
  float* gen_ll;     // [rows]
  const float* X;    // [B,K,D]
  const uint8_t* M;  // [B,D]
  long long* dbg;    // optional per-wave timestamps (tools/bench_mlp.py --trace)
};

int cs_mlp_launch(const CsMlpArgs& a, hipStream_t stream);

// The harness's per-iteration work of a run (cs_run_observe), defined with the diagnostics kernels.  begin() checks
// the stores and decides where the diagnostics run: on a stream of the library's own beside the run's (they read the
// kept copy X_store[t], which nothing overwrites, so iteration t + 1 need not wait for them) when the imputations are
// kept and the encoder is a fused-MLP handle (a layer program's workspace cannot serve two streams), else in the
// run's stream.  finish() makes the run's stream wait for what is still in flight.
struct CsObserver {
  const cs_run_observe* o = nullptr;
  const cs_mlp_t* enc = nullptr;
  int64_t B = 0;
  int32_t K = 0, D = 0, Zd = 0, latent_dist = 0;
  hipStream_t side = nullptr;     // null: everything in the run's stream
  hipEvent_t copied = nullptr, done = nullptr;
  int begin(const cs_mlp_t* enc, const cs_run_observe* o, int64_t B, int32_t K, int32_t D, int32_t Zd, int32_t latent_dist);
  // kept: X_store[t] already holds X (written by the iteration's own kernel)
  int iteration(int32_t t, const float* X, const uint8_t* M, hipStream_t stream, bool kept = false);
  int finish(hipStream_t stream);
};
// CS_ARITH_* in force for fused-MLP launches (cs_set_mlp_arith, or CSVAE_MLP_ARITH at first use)
int cs_mlp_arith();
