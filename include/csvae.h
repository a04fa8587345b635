/* csvae.h -- C ABI of the MI355X conditional-sampling library (libcsvae.so).
 *
 * The library implements, as hand-written HIP kernels for gfx950, the per-iteration work of
 * the two VAE conditional samplers of vsimkus/vae-conditional-sampling:
 *   LAIR   (irwg/sampling/vae_imputation.py:484-726, class at
 *           irwg/sampling/test_step_vae_sampling.py:1770) and
 *   AC-MWG (irwg/sampling/test_step_vae_sampling.py:1457-1768).
 * The reference has no native layer; these entry points are what its Python functions bind
 * to in this build (each declaration cites the reference code it replaces).
 *
 * Conventions
 *   - every function returns 0 (CS_OK) or a negative CS_E* code; cs_last_error() returns a
 *     thread-local message for the last failure on the calling thread;
 *   - all tensor arguments are caller-allocated DEVICE pointers, contiguous row-major,
 *     float32 unless stated (masks: uint8, 1 = observed; indices: int64; multinomial
 *     uniforms: float64); the library only owns the handles it creates;
 *   - work is enqueued asynchronously on the given hipStream_t (passed as void*; NULL = the
 *     default stream) of the current device; no entry point synchronises the device;
 *   - every stochastic entry point takes optional injected draws; a NULL draw pointer means
 *     "generate with the built-in counter RNG (Philox4x32-10)" keyed by
 *     (seed, stream, iteration, example_offset + example index, element), so results are
 *     independent of batching and of how examples are sharded over GPUs.
 */
#ifndef CSVAE_H
#define CSVAE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CS_OK 0
#define CS_EINVAL (-1)       /* bad argument / unsupported shape */
#define CS_EHIP (-2)         /* a HIP runtime call failed */
#define CS_EUNSUPPORTED (-3) /* option of the reference that this library does not implement */

/* distribution families (irwg/models/vae.py:19-28, create_distribution :203-256) */
#define CS_DIST_NORMAL 0          /* scale = softplus(raw)            */
#define CS_DIST_NORMAL_WITH_EPS 1 /* scale = softplus(raw) + 1e-8     */
#define CS_DIST_BERN 2            /* logits                           */
#define CS_DIST_BETA 3            /* latent only: (alpha, beta) = softplus(raw halves); vae.py:239-243 */
/* Generator (likelihood) families of the LAIR path beyond those (vae.py:220-229, 244-247): layer-program decoders
 * only.
 *   Student-t: the decoder outputs (df_raw | loc | scale_raw), D each; `gen` holds [loc | scale | df] with
 *   df = softplus(df_raw), scale = softplus(scale_raw) (+ 0.25 for STUDENTT_EPS22).  A sample is
 *   loc + scale * n * rsqrt(c / df), n standard normal (eps_x), c chi-square(df) (chi2_x) as torch's StudentT.rsample.
 *   Logit-normal: (loc | scale_raw), scale = softplus + 1e-8; a sample is sigmoid(loc + scale * n) clamped to
 *   [FLT_MIN, 1 - FLT_EPSILON], the density torch's SigmoidTransform of a Normal gives. */
#define CS_DIST_STUDENTT 4
#define CS_DIST_STUDENTT_EPS22 5
#define CS_DIST_LOGITNORMAL_WITH_EPS 6
/* prior over the latents (VAE.get_prior, vae.py:158-164) */
#define CS_PRIOR_STD_NORMAL 0
#define CS_PRIOR_UNIFORM 1        /* Uniform(0, 1 + 1e-8): use_uniform_prior (toy configuration) */

/* importance weighting (vae_imputation.py:615-640) */
#define CS_W_DMIS 0
#define CS_W_DMIS_WITHIN_GROUPS 1
#define CS_W_SMIS 2
/* resampling scheme (vae_imputation.py:646-674) and method (:291-298) */
#define CS_RS_GR 0
#define CS_RS_GLR 1
#define CS_RS_LR 2 /* lr_resampling, vae_imputation.py:310-333 */
#define CS_RS_IR 3 /* ir_resampling, vae_imputation.py:460-482 */
#define CS_RM_MULTINOMIAL 0
#define CS_RM_SYSTEMATIC 1

/* annealing of the variational proposal towards the prior by the example's fraction of missing
 * dimensions (VAE.anneal_var_distribution_to_prior_based_on_missingness, vae.py:308-401) */
#define CS_ANNEAL_NONE 0
#define CS_ANNEAL_LINEAR 1      /* 'linear'      */
#define CS_ANNEAL_LINEAR_025 2  /* 'linear_0.25' */

int cs_version(void);
const char* cs_last_error(void);

/* ---------------------------------------------------------------------------------------
 * Residual MLP (irwg/models/neural_nets.py:106-169, ResidualFCNetwork; eval mode, ReLU,
 * no normalisation layers).  Weights are given as HOST pointers in the state-dict layout
 * (nn.Linear: weight[out,in] row-major, bias[out]); the library repacks them into its MFMA
 * operand stream and uploads them to the current device.
 *   Wblk: [num_blocks][2][hidden][hidden], bblk: [num_blocks][2][hidden]
 * out_split = 2 when the output is a concatenation (first half | second half) of two
 * parameter groups (loc | raw scale; vae.py:190-195), 1 otherwise.
 * Limits: hidden in {64,128,256}; in_dim <= 128; out_dim/out_split <= 64.
 */
typedef struct cs_mlp cs_mlp_t;
typedef struct {
  int32_t in_dim, out_dim, num_blocks, hidden, out_split;
} cs_mlp_desc;

int cs_mlp_create(const cs_mlp_desc* desc, const float* W0, const float* b0, const float* Wblk,
                  const float* bblk, const float* Wf, const float* bf, cs_mlp_t** out);
int cs_mlp_destroy(cs_mlp_t* net);
/* y[rows,out_dim] = net(x[rows,in_dim]); replaces ResidualFCNetwork.forward
 * (neural_nets.py:163-169), i.e. model.var_latent_network(X) / model.generator_network(Z). */
int cs_mlp_forward(const cs_mlp_t* net, const float* x, int64_t rows, float* y, void* stream);

/* ---------------------------------------------------------------------------------------
 * Layer-program networks: the convolutional ResNet encoder / decoder
 * (ResNetEncoder.forward irwg/models/vae_resnet.py:128-149, EncoderBlock :60-78,
 * ResNetDecoder.forward :259-277, DecoderBlock :166-184, Interpolate :14-22) and MLPs wider
 * than the fused kernel covers (toy configuration, hidden 512).  The caller describes the
 * network as a list of layers over numbered activation buffers; BatchNorm (eval mode) is
 * folded into the convolution / linear weights and a bias by the caller, Dropout is the identity.
 * Activations are NHWC ([row, y, x, channel]); weights are given in torch's layouts
 * (convolution [cout][cin][kh][kw], linear [out][in]) and repacked by the library.
 * The handle is a cs_mlp_t: every entry point that takes an encoder / decoder accepts it.
 */
enum { CS_OP_CONV = 1, CS_OP_LINEAR = 2, CS_OP_AVGPOOL = 3, CS_OP_RESIZE = 4,
       CS_OP_LAYERNORM = 5 /* nn.LayerNorm(cin), eps 1e-5, weight at w_off, bias at b_off, optional ReLU:
                              residual blocks with use_layer_norm (irwg/models/neural_nets.py:95-100) */ };
/* buffer ids: 0 = network input [rows,in_dim], 1..4 = scratch, 5 = network output [rows,out_dim] */
enum { CS_NET_NBUF = 6, CS_NET_BUF_IN = 0, CS_NET_BUF_OUT = 5 };
typedef struct {
  int32_t op;             /* CS_OP_* */
  int32_t src, dst, res;  /* buffer ids; res = -1: no residual input (added before the ReLU) */
  int32_t relu;           /* 1: ReLU on the result */
  int32_t cin, cout;      /* channels (LINEAR: in / out features; AVGPOOL, RESIZE: cin == cout) */
  int32_t hin, win, hout, wout; /* spatial sizes of src and dst (LINEAR: all 1) */
  int32_t ksize, stride, pad;   /* CONV: square kernel 1 or 3 */
  int32_t upsample;       /* CONV: nearest-neighbour factor (1 or 2) applied to src first
                             (Interpolate(scale_factor=2) + conv, vae_resnet.py:32-43) */
  int32_t in_nchw, out_nchw; /* 1: src / dst is channel-major [row, channel, y, x] (image-shaped
                             network inputs and outputs, flattened as the reference's view does) */
  int64_t w_off, b_off;   /* offsets in floats into the weight blob; b_off = -1: no bias */
} cs_layer_desc;
int cs_net_create(const cs_layer_desc* layers, int32_t n_layers, const float* blob, int64_t blob_floats,
                  int32_t in_dim, int32_t out_dim, int32_t out_split, cs_mlp_t** out);
/* Activation workspace of a layer-program network for launches of up to `rows` network rows (rows are processed in
 * passes bounded by a 6 GiB workspace).  Without it the first launch that needs a larger workspace allocates it
 * (and frees the smaller one after waiting for the device) inside the call; after cs_net_reserve(net, rows) no
 * launch of <= rows rows allocates.  A no-op for fused-MLP handles, which have no workspace. */
int cs_net_reserve(cs_mlp_t* net, int64_t rows);

/* counter-RNG key */
typedef struct {
  uint64_t seed;
  uint32_t iteration;
  int64_t example_offset; /* global index of local example 0 (rank shard offset) */
} cs_rng;

/* ---------------------------------------------------------------------------------------
 * LAIR iteration = importance_resampling_gibbs_iteration (vae_imputation.py:484-726) for
 * weighting in {dmis, dmis_within_groups, smis}, scheme in {gr, glr, lr, ir}, method in
 * {multinomial, systematic}, with R prior-replenishing proposals.  K' = K + R.
 */
typedef struct {
  int32_t B, K, R, I, D, Zd;
  int32_t latent_dist, gen_dist; /* CS_DIST_* */
  int32_t weighting, scheme, method;
  int32_t cluster_size; /* glr / dmis_within_groups */
  float weight_alpha;   /* resampling uses alpha*log_w (polynomial transform,
                           test_step_vae_sampling.py:2059-2061); 1 = none */
  int32_t prior;        /* CS_PRIOR_* */
  float proposal_temperature; /* var_proposal_temperature (0 = off): see cs_proposal_transform */
  int32_t proposal_anneal;    /* CS_ANNEAL_* (var_proposal_anneal_type_to_prior) */
  /* Historical proposals (vae_imputation.py:504-548, 677-680): X then holds K chains of which only
   * the first K_impute are imputed (the others are imputations taken from the history, appended by
   * cs_history_concat, that only propose); 0 = all K.  glr_pad_to > 0: global-local resampling and
   * the grouped weights lay their groups out for glr_pad_to proposals of which only K' exist
   * (the iterations before the history is used, :546-548). */
  int32_t K_impute;
  int32_t glr_pad_to;
} cs_lair_cfg;

typedef struct {
  float* X;         /* [B,K,D]  in: current imputations; out: new imputations (in place) */
  const uint8_t* M; /* [B,D]    1 = observed */
  /* injected draws, any may be NULL (-> Philox) */
  const float* eps_z;   /* [I,B,K',Zd] standard normal */
  const double* u_res;  /* multinomial: gr [B,K'], glr [B,G,c] */
  const double* u_res2; /* second resampling for latents_resampled */
  const float* u_sys;   /* systematic: gr [B], glr [B,G] */
  const float* u_sys2;
  const float* eps_x; /* [B,K,D] standard normal (Bernoulli generator: uniforms) */
  /* outputs / workspace (required unless marked optional) */
  float* q;            /* [B,K',2,Zd] proposal loc and scale (after softplus) */
  float* Z;            /* [I,B,K',Zd] proposals */
  float* gen;          /* [I,B,K',2,D] generator mean and scale; Bernoulli: [I,B,K',D] logits; Student-t: [I,B,K',3,D]
                          loc, scale, df */
  float* gen_ll;       /* [I,B,K'] masked log-likelihood of X under each proposal */
  float* log_w;        /* [I,B,K'] */
  float* norm_log_w;   /* [B,I*K'] (gr '(i k)'; glr '(g i c)'; lr '(k i)'; ir: the log-weights, '(i k)') */
  int64_t* idx;        /* [B,NS] source proposal (flattened i*K'+k') of output slot; NS = K', or
                          glr_pad_to when the padded group layout is in use */
  float* ess;          /* [B] */
  float* tess;         /* [B] ESS of the transformed weights */
  float* perplexity;   /* [B] */
  float* latents_resampled; /* optional [B,NS,Zd] (return_resampled_latents) */
  int64_t* idx2;            /* optional [B,NS] indices of that second resampling */
  const float* clip_min;    /* optional [D] (test_step_vae_sampling.py:1942-1950) */
  const float* clip_max;
  /* lr / ir with the multinomial method draw ONE index per chain: torch.multinomial(p, 1) is
   * argmax(p / e) with e ~ Exp(1) per candidate; injected e[B,K',I] (NULL -> counter RNG) */
  const float* e_res;
  const float* e_res2;
  float* X_out; /* optional [B,K_impute,D]: the new imputations go here instead of in place */
  const float* chi2_x; /* Student-t generators: injected chi-square(df) draws [B,K,D] beside eps_x (NULL -> counter RNG) */
} cs_lair_bufs;

/* Historical proposals of LAIR (ImputationHistoryQueue.sample_history /
 * sample_history_with_chain_idx_for_each_sample, irwg/sampling/utils.py:28-68, and the concat of
 * vae_imputation.py:543-545): Xc[b, :K] = X[b], Xc[b, K+h] = ring[slots[h], b, chain], where the
 * ring is [hist_len,B,K,D] in device memory and slots[H] (HOST array, the stratified window draws
 * shifted by the tail, computed by the caller) is common to the batch.  The chain is
 * chain_of_proposal[h] (HOST array; global-local resampling) or, when that is NULL,
 * chain_of_example[b] (device, int64) or, when that is NULL too, a counter-RNG draw per example. */
int cs_history_concat(const float* X, const float* ring, int64_t B, int32_t K, int32_t D, int32_t H,
                      const int32_t* slots, const int32_t* chain_of_proposal, const int64_t* chain_of_example,
                      const cs_rng* rng, float* Xc, void* stream);

/* Tempering and annealing of proposal parameters, in place on q[B,K',2,Zd] rows k < K:
 * VAE.distribution_param_tempering (vae.py:258-306: scale * temperature; Beta: concentrations *
 * temperature) followed by VAE.anneal_var_distribution_to_prior_based_on_missingness
 * (vae.py:308-401; mf = fraction of missing dimensions of the example, k_m = (1 - mf) + k*mf with
 * k = 1/scale (x 0.25 for 'linear_0.25'); scale * k_m, loc * (1 - mf); Beta with the uniform prior:
 * concentrations * k_m), each including the reference's round trip through
 * softplus^-1 / softplus (+- 1e-8 for normal_with_eps).  temperature <= 0: no tempering.
 * M[B,D] is only read when anneal != CS_ANNEAL_NONE. */
int cs_proposal_transform(float* q, const uint8_t* M, int64_t B, int32_t K, int32_t Kp, int32_t Zd, int32_t D,
                          int32_t latent_dist, int32_t prior, float temperature, int32_t anneal, void* stream);

/* q[B,K',2,Zd] = the prior as the proposal of every chain (loc 0, scale 1; Beta(1,1) for the
 * uniform prior): standard_importance_resampling (vae_imputation.py:920-930) is
 * cs_prior_proposals + cs_lair_decode + cs_lair_resample_impute. */
int cs_prior_proposals(float* q, int64_t B, int32_t Kp, int32_t Zd, int32_t latent_dist, void* stream);

/* encoder forward + distribution parameterisation: q[b,k,:] for k < K
 * (vae_imputation.py:551 + vae.py:214-219), then cs_proposal_transform when the configuration asks
 * for it (vae_imputation.py:553-561); rows k >= K of q are set to the prior
 * replenishing proposal (loc 0, scale 1; vae_imputation.py:563-577). */
int cs_lair_encode(const cs_mlp_t* enc, const cs_lair_cfg* cfg, const cs_lair_bufs* b, void* stream);
/* Z = loc + scale*eps, decoder forward, generator parameterisation, masked log-likelihood
 * (vae_imputation.py:596-603, weights.py:46-47). */
int cs_lair_decode(const cs_mlp_t* dec, const cs_lair_cfg* cfg, const cs_lair_bufs* b, const cs_rng* rng,
                   void* stream);
/* importance weights, resampling, sampling of the missing values, in-place imputation, ESS and
 * perplexity (weights.py:50-57,69-85; vae_imputation.py:284-308,335-378,380-458,677-726). */
int cs_lair_resample_impute(const cs_lair_cfg* cfg, const cs_lair_bufs* b, const cs_rng* rng, void* stream);
/* the three stages above, back to back on one stream */
int cs_lair_step(const cs_mlp_t* enc, const cs_mlp_t* dec, const cs_lair_cfg* cfg, const cs_lair_bufs* b,
                 const cs_rng* rng, void* stream);

/* T LAIR iterations in one call: the loop of TestVAELatentAdaptiveImportanceResampling.sampler
 * (test_step_vae_sampling.py:1887-1962) for callers that need nothing on the host between iterations.
 * Iteration t is exactly cs_lair_step with rng->iteration + t.  With `out`, iteration t writes its proposals
 * and log-weights into block t of Z_store[T,I,B,K',Zd] / log_w_store[T,I,B,K'] (the proposals the reference
 * collects for the final resampling, :1938-1940) and its statistics into row t of ess_store / tess_store /
 * perplexity_store [T,B]; a NULL store leaves the corresponding buffer of `b` in use (overwritten every
 * iteration).  Injected draws in `b`, when not NULL, hold T consecutive blocks of their per-iteration shapes. */
typedef struct {
  float* Z_store;
  float* log_w_store;
  float* ess_store;
  float* tess_store;
  float* perplexity_store;
} cs_lair_run_out;
int cs_lair_run(const cs_mlp_t* enc, const cs_mlp_t* dec, const cs_lair_cfg* cfg, const cs_lair_bufs* b,
                const cs_rng* rng, int32_t T, const cs_lair_run_out* out, void* stream);

/* What the harness of test_step does with the chains after every iteration (test_step_vae_sampling.py:386-391
 * keeps the imputations, :431-445 computes the diagnostics), done by the run itself so that the harness need not
 * come back to the host between iterations.  After iteration t of a run:
 *   X_store[t]    = the imputations X[B,K,D]                                   (NULL: not kept)
 *   rmse_store[t] / mae_store[t] = cs_imputation_errors(X, X_true, M)          (NULL: not computed)
 *   wass_store[t] = cs_latent_wasserstein(encoder posterior of X, q_true)      (NULL: not computed; Gaussian
 *                   posteriors only; q_work[B,K,2,Zd] is the workspace of the encoder pass, q_true[B,2,Zd] the
 *                   posterior of the true data, encoded once by the caller)
 * each store holding T blocks of the per-iteration shape [B,K].  When X_store is given (and the encoder is a fused-MLP
 * handle, or wass_store is NULL) the diagnostics read the kept copies on a stream of the library's own beside
 * `stream`, which waits for them before the call returns: the caller sees ordinary stream order. */
typedef struct {
  float* X_store;
  const float* X_true; /* [B,D]; with rmse_store / mae_store */
  float* rmse_store;
  float* mae_store;
  const float* q_true;
  float* q_work;
  float* wass_store;
} cs_run_observe;
/* cs_lair_run with the harness's per-iteration work (obs may be NULL: cs_lair_run).  Needs in-place imputation
 * of all K chains (b->X_out NULL, cfg->K_impute 0 or K). */
int cs_lair_run_observed(const cs_mlp_t* enc, const cs_mlp_t* dec, const cs_lair_cfg* cfg, const cs_lair_bufs* b,
                         const cs_rng* rng, int32_t T, const cs_lair_run_out* out, const cs_run_observe* obs,
                         void* stream);

/* ---------------------------------------------------------------------------------------
 * AC-MWG (test_step_vae_sampling.py:1512-1768).  The imputation history
 * (ImputationHistoryQueue_with_restrictedavailability, irwg/sampling/utils.py:83-237) is a
 * caller-allocated ring[hist_len,B,K,D] plus availability counters avail[B,K] (int64) in
 * device memory; the caller tracks the two host scalars of the reference queue: the slot of
 * the newest entry (`tail_idx`) and whether the ring has wrapped (`full`).
 */
typedef struct {
  int32_t B, K, D, Zd;
  int32_t latent_dist, gen_dist;
  int32_t hist_len;  /* ring capacity (max_history_length) */
  int32_t hist_tail; /* slot of the newest entry before this call */
  int32_t hist_full; /* 1 once the ring has wrapped */
  int32_t flags;     /* CS_MWG_* bits below; 0 = AC-MWG as published */
  double prior_mixture_probability;
  float proposal_temperature; /* var_proposal_temperature (0 = off), applied to the step's proposal */
  int32_t proposal_anneal;    /* CS_ANNEAL_* */
  int32_t ring_u8;            /* 1: bufs.ring points to uint8_t [hist_len,B,K,D] holding x != 0 (the reference's
                               * torch.bool queue for binary data, test_step_vae_sampling.py:1537-1541: a quarter of the
                               * float ring's memory and traffic); the encoder must then be a layer program */
} cs_acmwg_cfg;

/* Variants of the Metropolis-within-Gibbs step that share the AC-MWG kernels.
 * CS_MWG_PROPOSE_FROM_CURRENT: the proposal conditions on the chain's current imputation instead
 *   of a historical one (TestVAEMetropolisWithinGibbs.sampler, test_step_vae_sampling.py:1221);
 *   no history is read or written: ring, avail, r_hist, slots and in_off may be NULL.
 * CS_MWG_JOINT_TARGET_NEW: the proposal's log-likelihood sums over all D dimensions of the current
 *   imputation, not only the observed ones (:1245; ablation_use_mwg_target :1722-1723).
 * CS_MWG_JOINT_TARGET_OLD: the same for the log-likelihood kept for the next step (:1217,1269;
 *   the AC-MWG ablation uses it for the state after the warm-up only, :1657-1658, and keeps the
 *   masked sum in its steps, :1755). */
#define CS_MWG_PROPOSE_FROM_CURRENT 1
#define CS_MWG_JOINT_TARGET_NEW 2
#define CS_MWG_JOINT_TARGET_OLD 4

typedef struct {
  float* X;         /* [B,K,D] imputations, updated in place */
  const uint8_t* M; /* [B,D] */
  /* chain state, updated in place */
  float* Z_old;          /* [B,K,Zd] */
  float* gen_old;        /* [B,K,2,D] generator mean|scale (Bernoulli: [B,K,D] logits; Student-t: [B,K,3,D] loc|scale|df) of Z_old */
  float* z_old_prior_lp; /* [B,K] log p(Z_old) */
  float* x_old_lp;       /* [B,K] masked log-likelihood of X under gen_old */
  /* history */
  float* ring;    /* [hist_len,B,K,D] */
  int64_t* avail; /* [B,K] */
  /* injected draws, any may be NULL (-> counter RNG) */
  const int64_t* r_hist; /* [B,K] integer in [0, max(avail,1)) (utils.py:225) */
  const float* eps_z;    /* [B,K,Zd] */
  const float* z_prior;  /* [B,K,Zd] prior proposal */
  const float* u_prior;  /* [B,K] */
  const float* u_accept; /* [B,K] */
  const float* eps_x;    /* [B,K,D] */
  /* workspace / outputs */
  int64_t* slots;     /* [B,K] ring slot of the sampled historical imputation */
  int64_t* in_off;    /* [B,K] workspace */
  uint8_t* use_prior; /* [B,K] */
  float* z_alt;       /* [B,K,Zd] workspace */
  float* q;           /* [B,K,2,Zd] proposal loc|scale */
  float* Z_new;       /* [B,K,Zd] */
  float* gen_new;     /* like gen_old */
  float* ll_new;      /* [B,K] */
  uint8_t* accepted;  /* [B,K] */
  float* log_accept;  /* [B,K] */
  const float* chi2_x; /* Student-t generators: injected chi-square(df) draws [B,K,D] beside eps_x (NULL -> counter RNG) */
} cs_acmwg_bufs;

/* chain state after the warm-up (test_step_vae_sampling.py:1639-1660): q = enc(X),
 * Z_old = loc + scale*eps_z, gen_old = dec(Z_old), z_old_prior_lp, x_old_lp. */
int cs_acmwg_init_state(const cs_mlp_t* enc, const cs_mlp_t* dec, const cs_acmwg_cfg* cfg, const cs_acmwg_bufs* b,
                        const cs_rng* rng, void* stream);
/* one MWG step (:1666-1755) including the history index draw (utils.py:190-237), the
 * availability update (utils.py:94-99) and the enqueue of the new imputations (utils.py:101-113)
 * into slot (hist_tail+1) % hist_len.  After the call the caller advances its tail/full. */
int cs_acmwg_step(const cs_mlp_t* enc, const cs_mlp_t* dec, const cs_acmwg_cfg* cfg, const cs_acmwg_bufs* b,
                  const cs_rng* rng, void* stream);
/* T MWG steps in one call: the loop of the samplers (:1221-1280, :1666-1766) for callers that need nothing on
 * the host between steps.  Step t is exactly cs_acmwg_step with rng->iteration + t and the history scalars the
 * reference's queue would hold then: hist_tail advances by one slot per step (mod hist_len) and hist_full turns 1
 * when the tail reaches the last slot (utils.py:41-44); the caller advances its own scalars T times afterwards.
 * With `out`, step t writes accepted / log_accept / slots into block t of the stores [T,B,K] (a NULL store leaves
 * the buffer of `b` in use, overwritten every step) and copies the chain latents Z_old into Z_store[t] ([T,B,K,Zd]).
 * Injected draws in `b`, when not NULL, hold T consecutive blocks of their per-step shapes.  `obs` as in
 * cs_lair_run_observed. */
typedef struct {
  uint8_t* accepted_store;
  float* log_accept_store;
  int64_t* slots_store;
  float* Z_store;
} cs_acmwg_run_out;
int cs_acmwg_run(const cs_mlp_t* enc, const cs_mlp_t* dec, const cs_acmwg_cfg* cfg, const cs_acmwg_bufs* b,
                 const cs_rng* rng, int32_t T, const cs_acmwg_run_out* out, const cs_run_observe* obs, void* stream);
/* enqueue_batch (utils.py:101-113): ages the counters when the ring was already full, then
 * writes X[B,K,D] into slot new_tail.  With was_full = 0 (avail may then be NULL) it is the
 * enqueue of the plain ImputationHistoryQueue of LAIR (utils.py:14-16). */
int cs_history_enqueue(float* ring, int64_t* avail, int32_t hist_len, int32_t new_tail, int32_t was_full,
                       const float* X, int64_t B, int32_t K, int32_t D, void* stream);
/* update_available_timesteps (utils.py:94-99): avail[mask] = len (mask NULL: all chains). */
/* The same two operations on a uint8 ring (binary data: the ring holds x != 0, reads give 0.0f / 1.0f). */
int cs_history_enqueue_u8(uint8_t* ring, int64_t* avail, int32_t hist_len, int32_t new_tail, int32_t was_full,
                          const float* X, int64_t B, int32_t K, int32_t D, void* stream);
int cs_history_concat_u8(const float* X, const uint8_t* ring, int64_t B, int32_t K, int32_t D, int32_t H,
                         const int32_t* slots, const int32_t* chain_of_proposal, const int64_t* chain_of_example,
                         const cs_rng* rng, float* Xc, void* stream);
int cs_history_set_available(int64_t* avail, const uint8_t* mask, int64_t chains, int64_t len, void* stream);

/* out[row] = sum_d log N(Z[row,d]; 0, 1): reduce(prior_distr.log_prob(Z), '... d -> ...', 'sum')
 * (test_step_vae_sampling.py:1655). */
int cs_stdnormal_logprob_rows(const float* Z, int64_t rows, int32_t Zd, float* out, void* stream);
/* out[b,k] = sum_d w_d log p(X[b,k,d] | gen[b,k]) with w = M[b,:] (NULL: all ones):
 * reduce(mis_distr.log_prob(X) [*M], '... d -> ...', 'sum') of the chain state after a pseudo-Gibbs
 * warm-up (test_step_vae_sampling.py:1217,1660).  gen as written by cs_decode_params. */
int cs_gen_loglik_rows(const float* gen, int32_t gen_dist, const float* X, const uint8_t* M, int64_t B, int32_t K,
                       int32_t D, float* out, void* stream);

/* Per-iteration imputation diagnostics (test_step_vae_sampling.py:431-445), on the device.
 * rmse[b,k] = sqrt(sum_{d missing} (X[b,k,d] - X_true[b,d])^2 / max(#missing_b, 1))
 * (imputation_rmse_metric, irwg/sampling/imputation_metrics.py:85-102); mae likewise with |.|
 * (imputation_mae_metric, :133-143).  Either output may be NULL. */
int cs_imputation_errors(const float* X, const float* X_true, const uint8_t* M, int64_t B, int32_t K, int32_t D,
                         float* rmse, float* mae, void* stream);
/* out[b,k] = 2-Wasserstein distance (squared) between the Gaussian posteriors q_imp[B,K,2,Zd] and
 * q_true[B,1,2,Zd] given as (loc, scale): imputation_latent_wass (imputation_metrics.py:200-213)
 * after the two encoder passes (cs_lair_encode). */
int cs_latent_wasserstein(const float* q_imp, const float* q_true, int64_t B, int32_t K, int32_t Zd, float* out,
                          void* stream);

/* Order statistics of the rows of x[rows, m] (device, float32): out[row, j] = the ranks[j]-th smallest value of the
 * row (0-based; `ranks` is a HOST array of nq <= 16 ranks common to all rows; rows <= 65535 per call).  Radix selection
 * in three passes over the data: the per-iteration quantiles of the point-wise diagnostics over a whole test set
 * (np.quantile in test_epoch_end, test_step_vae_sampling.py:824-858) without sorting it.  The value returned is an
 * element of the row; NaNs order above +inf (below -inf with the sign bit set), a caller that must answer NaN for a
 * row holding one checks for it itself.  `workspace`: cs_row_order_statistics_workspace(rows, nq) bytes of device
 * memory. */
int64_t cs_row_order_statistics_workspace(int64_t rows, int32_t nq);
int cs_row_order_statistics(const float* x, int64_t rows, int64_t m, const int64_t* ranks, int32_t nq, float* out,
                            void* workspace, int64_t workspace_bytes, void* stream);

/* HOST function (no device work, callable from any thread): one piece of a raw deflate stream (RFC 1951) for
 * src[0..n), for the `.npz` artefacts the samplers write with np.savez_compressed (test_step_vae_sampling.py:785-791,
 * 2039-2045).  Matches are looked for at distance `stride` only (the row length in bytes of a [.., K, D] array,
 * 1..32768: the observed entries of a row repeat the row before it), symbols are coded with the fixed Huffman code,
 * and the piece is closed by an empty stored block, so pieces concatenate into one valid stream that any inflate
 * reads.  `lookback`: how many bytes before src belong to the same stream and are readable at src - lookback.
 * Returns the number of bytes written to dst, or -1 (dst_cap too small: n + n/8 + 64 always suffices). */
int64_t cs_deflate_rows(const uint8_t* src, int64_t n, int32_t stride, int64_t lookback, uint8_t* dst, int64_t dst_cap);
/* HOST functions: CRC-32 of the zip container (zlib's crc32: cs_crc32(p, n, 0) of a whole buffer, or continued from
 * `crc`), and the checksum of two adjacent pieces from their own checksums and the length of the second, so that the
 * threads that deflate the pieces of a member also checksum them. */
uint32_t cs_crc32(const uint8_t* p, int64_t n, uint32_t crc);
uint32_t cs_crc32_combine(uint32_t crc1, uint32_t crc2, int64_t len2);

/* gen[rows,2,D] (mean, scale), [rows,D] (Bernoulli logits) or [rows,3,D] (Student-t: loc, scale, df) = generator parameterisation of
 * generator_network(Z[rows,Zd]) (vae.py:171-172,203-256). */
int cs_decode_params(const cs_mlp_t* dec, const float* Z, int64_t rows, int32_t gen_dist, float* gen, void* stream);
/* x_m ~ p(x | gen) and X = X*M + x_m*(1-M) for rows (b,k), in place: pseudo_gibbs_iteration
 * (vae_imputation.py:31-45), the 'vae_samples' initial imputation
 * (test_step_vae_sampling.py:279-287) and the final decode of LAIR (:2019-2026).
 * M == NULL: X = x_m.  eps[B*K,D] injected draws or NULL (-> counter RNG stream `stream_id`,
 * row = example_offset*K + row).  clip_min/clip_max[D] (optional) bound x_m before blending.
 * flags: CS_SI_ZERO_MISSING treats X as 0 where M = 0 (NaN-coded inputs);
 *        CS_SI_NONFINITE_KEEP keeps the previous value where x_m is not finite. */
#define CS_SI_ZERO_MISSING 1
#define CS_SI_NONFINITE_KEEP 2
int cs_sample_impute(const float* gen, int32_t gen_dist, const float* eps, const cs_rng* rng, uint32_t stream_id,
                     float* X, const uint8_t* M, int64_t B, int32_t K, int32_t D, const float* clip_min,
                     const float* clip_max, int32_t flags, void* stream);
/* The same with the second injected draw of the Student-t generators (chi2[B*K,D] chi-square(df) draws, or NULL). */
int cs_sample_impute2(const float* gen, int32_t gen_dist, const float* eps, const float* chi2, const cs_rng* rng,
                      uint32_t stream_id, float* X, const uint8_t* M, int64_t B, int32_t K, int32_t D,
                      const float* clip_min, const float* clip_max, int32_t flags, void* stream);

/* LAIR final resampling (test_step_vae_sampling.py:1998-2009): logw[TI,B,K'] and
 * Zprops[TI,B,K',Zd] hold the proposals of all TI = T*I stored iterations; per example a
 * Categorical over the S = TI*K' proposals (flattened '(t i k)') is sampled n times with the
 * float64 uniforms u[B,n] (NULL -> counter RNG) -> idx[B,n], Zout[B,n,Zd].
 * cdf_ws: workspace of B*S floats. */
int cs_final_resample(const float* logw, const float* Zprops, int64_t TI, int64_t B, int32_t Kp, int32_t Zd,
                      const double* u, const cs_rng* rng, int64_t n, float* cdf_ws, int64_t* idx, float* Zout,
                      void* stream);

/* cs_final_resample without a workspace argument, for callers that cannot allocate device memory (code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart
 * in the reference.  The CDF workspace is a fixed 32 MiB array in the library's device memory (each GPU the
 * library runs on holds its copy until the process ends), so B * TI * K' must not exceed 8 388 608;
 * larger problems are rejected with CS_EINVAL (use cs_final_resample).  The call allocates nothing; the
 * first call on a device looks the array's address up (make that call outside graph capture).  All
 * calls on one GPU share the array: calls that overlap on one GPU (different streams) race.  The array is
 * the one of the CURRENT device (hipGetDevice): the stream must belong to it.  The 32 MiB are part of the
 * library's code object: every GPU on which a process has loaded the library holds them whether or not this
 * entry point is ever called.  Same arguments and results as cs_final_resample otherwise. */
int cs_final_resample_static(const float* logw, const float* Zprops, int64_t TI, int64_t B, int32_t Kp,
                             int32_t Zd, const double* u, const cs_rng* rng, int64_t n, int64_t* idx, float* Zout,
                             void* stream);

/* Draws the built-in RNG would use, written out for tests: standard normals / uniforms of
 * `stream_id` (CS_STREAM_* in csrc/cs_common.h) for rows example_offset + [0,rows) and
 * elements [0,cols). */
int cs_rng_normal(const cs_rng* rng, uint32_t stream_id, int64_t rows, int32_t cols, float* out, void* stream);
int cs_rng_uniform(const cs_rng* rng, uint32_t stream_id, int64_t rows, int32_t cols, float* out, void* stream);
int cs_rng_uniform64(const cs_rng* rng, uint32_t stream_id, int64_t rows, int32_t cols, double* out, void* stream);

/* Arithmetic of the matrix products of the fused residual-MLP kernel (no counterpart in the
 * reference, whose torch.nn.Linear layers compute in float32):
 *   CS_ARITH_F16X2   (default) every float32 operand as two fp16 terms (11 + 11 significand bits), three products
 *                    per pair, float32 accumulation; the neglected term is below 2^-22 |a||b|.  fp16's 5 exponent
 *                    bits are dealt with by exact powers of two: every operand row is normalised (its largest
 *                    magnitude into [2^14, 2^15), fp16's top binade) and every layer's weights (largest into
 *                    [2^14, 2^15)), the exponents are carried along and taken out at the end.  An operand down to
 *                    2^-17 of its row's largest is represented to 2^-22 of itself, a smaller one to 2^-39 of the row's
 *                    largest.  Results are float32-equivalent: against a float64 evaluation the error is that of the
 *                    exact-float32 kernel or lower, for inputs and weights of any scale, outliers and in-row spreads
 *                    of 1e4 included (tests/test_gpu_f16x2_arith.py); they are not bit-identical to it.  A network
 *                    with a non-finite weight has no fp16 stream: under the process-wide setting it runs in
 *                    CS_ARITH_BF16X3, a handle told CS_ARITH_F16X2 itself (cs_mlp_set_arith) is refused.  In
 *                    layer-program networks (cs_net_create) the matrix convolutions run this arithmetic with ONE power
 *                    of two per image (all taps of an output pixel share a scale; an image's result does not depend on
 *                    the other images of a launch), taken from the largest magnitude of the image's activation, which
 *                    the producing layer leaves in the handle's workspace (tests/test_gpu_conv_f16x2.py); their linear
 *                    layers and convolutions to at most 4 channels keep CS_ARITH_BF16X3.
 *   CS_ARITH_BF16X3  every float32 operand split exactly into three bfloat16 terms (8 + 8 + 8 significand bits), a
 *                    product accumulated in float32 from the six leading term products; the neglected terms are below
 *                    2^-23 of |a||b|.  Float32-equivalent like the default (tests/test_gpu_split_arith.py) with
 *                    float32's exponent range in every term, at 1.5x the default's time.  Operands that are
 *                    infinite or within a factor 2^-8 of FLT_MAX give NaN where float32 gives +-inf.
 *   CS_ARITH_F32     exact float32 matrix instructions (2.7x the default's time).
 *   CS_ARITH_BF16    opt-in throughput mode, never a default: weights and layer inputs rounded to bfloat16
 *                    (nearest even), one matrix instruction per product, float32 accumulation, biases /
 *                    ReLU / residual sums / epilogues in float32.  NOT float32-equivalent: every operand carries
 *                    a relative rounding error of up to 2^-9; results follow a bfloat16-rounded evaluation of
 *                    the network (tests/test_gpu_bf16_arith.py), not the reference's float32 one.  Fused MLP
 *                    kernel only; layer-program networks keep CS_ARITH_BF16X3 under this setting.
 * The setting is process-wide and applies to launches made after the call; the environment variable
 * CSVAE_MLP_ARITH (f16x2 | bf16x3 | f32 | bf16) gives the initial value.  It also selects the kernels of the matrix-core
 * convolution / linear layers of layer-program networks (cs_net_create): exact float32 under CS_ARITH_F32, the fp16
 * split of the convolutions under CS_ARITH_F16X2, the bfloat16 split otherwise. */
enum { CS_ARITH_DEFAULT = -1, CS_ARITH_F32 = 0, CS_ARITH_BF16X3 = 1, CS_ARITH_BF16 = 2, CS_ARITH_F16X2 = 3 };
int cs_set_mlp_arith(int arith);
int cs_get_mlp_arith(void);
/* The same choice per network handle (fused MLP or layer program): it overrides the process-wide setting for
 * every launch that uses the handle; CS_ARITH_DEFAULT returns the handle to the process-wide setting.  A caller
 * that sets the arithmetic on its handles needs no process-wide state. */
int cs_mlp_set_arith(cs_mlp_t* net, int arith);
int cs_mlp_get_arith(const cs_mlp_t* net);

/* Profiling hook (no counterpart in the reference): when buf != NULL every fused-MLP launch
 * writes 16 int64 per 16-row tile into buf (device memory, 16*ceil(rows/16) int64): clock stamps in
 * slots 0..14, (XCC_ID << 32 | HW_ID) of the wavefront in slot 15. */
int cs_debug_set_trace(long long* buf);
/* Test hook (no counterpart in the reference): bound in bytes of the activation workspace of
 * layer-program networks, which decides how many rows one pass processes; 0 restores the default
 * (6 GiB). */
int cs_debug_set_workspace_budget(int64_t bytes);
/* Test hook (no counterpart in the reference): cs_lair_resample_impute handles proposal sets of up to 64 per
 * example (I*K') in one launch, one wavefront per group of examples, and larger sets in three launches; with
 * on = 1 every size takes the three-launch path.  Returns the previous setting. */
int cs_debug_force_large_resample(int on);
/* Test hook (no counterpart in the reference): 1 keeps the scalar log-weight kernel of the path for more than 64
 * proposals per example where its 16-byte-access form would run; both give the same bits. */
int cs_debug_force_scalar_logw(int on);
/* Test hook (no counterpart in the reference): a pass of a layer program over at least `min_rows` rows runs as two
 * halves, the second on a stream of the handle's own beside the caller's (which waits for it before the call returns);
 * 0 keeps every pass in the caller's stream.  Returns the previous value (default 8; a pass also needs about 40 GFLOP
 * of work to be divided; a negative value divides every pass of at least -min_rows rows whatever its work). */
int cs_debug_set_prog_two_streams(int min_rows);
/* Test hook (no counterpart in the reference): matrix-kernel convolutions of launches that would leave half the CUs
 * without a workgroup run in small blocks (32 output channels each) when a block has at least `steps` (tap,
 * 32-channel chunk) steps; 0 keeps the large blocks for every launch.  Returns the previous value (default 18). */
int cs_debug_set_small_launch_steps(int steps);
/* Test hook (no counterpart in the reference): launches of the fused MLP in the fp16 split of up to 512 row tiles run
 * in a kernel that gives a row tile to a workgroup and splits each layer's output features over its wavefronts;
 * mode 0 disables that kernel, 1 uses it for every launch it has an instance for, -1 restores the size rule.
 * Returns the previous mode. */
int cs_debug_set_narrow(int mode);

#ifdef __cplusplus
}
#endif
#endif
