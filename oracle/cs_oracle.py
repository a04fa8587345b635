"""CPU oracle for the conditional-sampling hot path (LAIR and AC-MWG iterations).

TEST INFRASTRUCTURE ONLY -- this file is the *checker*.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.
The product package (``vae_conditional_sampling_amd``) never does; its compute runs in
the HIP library and fails loudly when that library is missing.

It is a from-scratch restatement, in plain float32 torch-CPU tensor arithmetic, of what
the reference computes on the path, with every random draw made an explicit input
("injected draws") so that the reference, this oracle and the HIP engine can be fed the
same numbers.  Parity pinning: ``oracle/check_against_reference.py`` runs the reference
package itself (importable in the build container only) on seeded inputs with a recorded
draw tape and asserts this file reproduces every intermediate; the vectors it writes are
committed under ``tests/golden/`` and re-checked by ``tests/test_oracle_golden.py``.
The reference's own self-tests for the path (history-queue known-answer test, DM-MIS
vs nested loops, systematic-resampling histogram) are re-expressed in
``tests/test_oracle_reference_selftests.py``.

Reference locations restated here (paths relative to the reference root):
  * residual MLP forward                irwg/models/neural_nets.py:87-103,163-169
  * distribution parameterisation       irwg/models/vae.py:203-256  (softplus + 1e-8)
  * prior, unconditional sample         irwg/models/vae.py:158-178
  * DM-MIS log-weights                  irwg/utils/weights.py:37-85
  * grouped DM-MIS                      irwg/sampling/vae_imputation.py:380-458
  * gr / glr resampling                 irwg/sampling/vae_imputation.py:284-308,335-378
  * lr / ir resampling                  irwg/sampling/vae_imputation.py:310-333,460-482,730-738
  * standard importance resampling      irwg/sampling/vae_imputation.py:904-998
  * systematic resampling               irwg/sampling/vae_imputation.py:804-834
  * LAIR iteration                      irwg/sampling/vae_imputation.py:484-726
  * pseudo-Gibbs iteration              irwg/sampling/vae_imputation.py:19-45
  * AC-MWG step                         irwg/sampling/test_step_vae_sampling.py:1664-1768
  * restricted-availability history     irwg/sampling/utils.py:83-237
  * LAIR final resampling               irwg/sampling/test_step_vae_sampling.py:1979-2045
Arithmetic that lives in torch itself (Normal/Bernoulli log-pdf, softplus, multinomial)
is restated from the installed torch's published formulas.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

EPSILON = 1e-8  # irwg/models/vae.py:16
LOG_SQRT_2PI = math.log(math.sqrt(2 * math.pi))


# --------------------------------------------------------------------------------------
# networks
# --------------------------------------------------------------------------------------
def make_resfc_weights(in_dim, out_dim, num_blocks, hidden, seed, norm=None):
    """Random weights with torch's default nn.Linear initialisation (kaiming-uniform,
    a=sqrt(5)) for a residual MLP of the reference's shape.  Returns a dict of float32
    torch tensors: W0[H,in] b0[H]; blocks: list of (W1[H,H], b1[H], W2[H,H], b2[H]);
    Wf[out,H] bf[out].  ``norm`` = 'layer' / 'batch' adds 'norms': per block the parameters of the
    normalisation after the block's first linear layer (neural_nets.py:81-85), drawn at random so
    that the affine part and the running statistics matter."""
    g = torch.Generator().manual_seed(seed)

    def lin(o, i):
        bound = 1.0 / math.sqrt(i)
        W = (torch.rand(o, i, generator=g) * 2 - 1) * bound
        b = (torch.rand(o, generator=g) * 2 - 1) * bound
        return W.float(), b.float()

    W0, b0 = lin(hidden, in_dim)
    blocks = []
    for _ in range(num_blocks):
        W1, b1 = lin(hidden, hidden)
        W2, b2 = lin(hidden, hidden)
        blocks.append((W1, b1, W2, b2))
    Wf, bf = lin(out_dim, hidden)
    w = {"W0": W0, "b0": b0, "blocks": blocks, "Wf": Wf, "bf": bf}
    if norm is not None:
        norms = []
        for _ in range(num_blocks):
            n = {"kind": norm, "weight": 0.5 + torch.rand(hidden, generator=g), "bias": 0.2 * torch.randn(hidden, generator=g)}
            if norm == "batch":
                n["mean"] = 0.3 * torch.randn(hidden, generator=g)
                n["var"] = 0.5 + torch.rand(hidden, generator=g)
            norms.append(n)
        w["norms"] = norms
    return w


def net_input_dim(w):
    """Input width of a network weight dict (residual MLP or ResNet decoder)."""
    cfg = w.get("cfg")
    if isinstance(cfg, dict) and cfg.get("kind") == "resnet_dec":
        return cfg["latent_dim"]
    return w["W0"].shape[1]


def resfc_forward(w, x):
    """h = relu(W0 x + b0); per block h = relu(h + W2 relu(W1 h + b1) + b2); y = Wf h + bf.
    (neural_nets.py:163-169 with :87-103; dropout is identity in eval mode; no norm layers
    in the configurations on the path.)"""
    cfg = w.get("cfg") if isinstance(w, dict) else None
    if isinstance(cfg, dict) and str(cfg.get("kind", "")).startswith("resnet"):
        from . import cs_oracle_conv  # weight dicts of the convolutional nets carry their own forward
        return cs_oracle_conv.net_forward(w, x)
    h = F.relu(F.linear(x, w["W0"], w["b0"]))
    norms = w.get("norms") or [None] * len(w["blocks"])
    for (W1, b1, W2, b2), n in zip(w["blocks"], norms):
        t = F.linear(h, W1, b1)
        if n is not None:   # neural_nets.py:90-99, eval mode; applied on the 2-d view
            H = t.shape[-1]
            t2 = t.reshape(-1, H)
            if n["kind"] == "layer":
                t2 = F.layer_norm(t2, (H,), n["weight"], n["bias"], 1e-5)
            else:
                t2 = F.batch_norm(t2, n["mean"], n["var"], n["weight"], n["bias"], False, 0.1, 1e-5)
            t = t2.reshape(t.shape)
        t = F.relu(t)
        h = F.relu(h + F.linear(t, W2, b2))
    return F.linear(h, w["Wf"], w["bf"])


def resfc_forward_bf16(w, x):
    """The residual MLP of `resfc_forward` (no norm layers) in the opt-in plain-bf16 arithmetic of the fused
    kernel: the weights and the input of every layer rounded to bfloat16 (nearest even), products summed in
    float32, biases / ReLU / the residual sum in float32.  No counterpart in the reference (float32 throughout):
    this is the yardstick of tests/test_gpu_bf16_arith.py only."""
    rb = lambda t: t.to(torch.bfloat16).to(torch.float32)
    lin = lambda h, W, b: F.linear(rb(h), rb(W)) + b
    h = F.relu(lin(x, w["W0"], w["b0"]))
    for W1, b1, W2, b2 in w["blocks"]:
        t = F.relu(lin(h, W1, b1))
        h = F.relu((h + b2) + F.linear(rb(t), rb(W2)))
    return lin(h, w["Wf"], w["bf"])


# --------------------------------------------------------------------------------------
# distributions (vae.py:203-256 + torch.distributions formulas)
# --------------------------------------------------------------------------------------
def split_loc_scale(params, dist):
    """'... (2 d) -> loc, scale' with scale = softplus(raw) (+1e-8 for normal_with_eps)."""
    d = params.shape[-1] // 2
    loc, raw = params[..., :d], params[..., d:]
    scale = F.softplus(raw)
    if dist == "beta":  # (alpha, beta) = softplus of both halves (vae.py:239-243)
        return F.softplus(loc), scale
    if dist == "normal_with_eps":
        scale = scale + EPSILON
    elif dist != "normal":
        raise NotImplementedError(dist)
    return loc, scale


def normal_log_prob(x, loc, scale):
    """torch.distributions.Normal.log_prob: -(x-loc)^2/(2 var) - log(scale) - log(sqrt(2 pi))."""
    var = scale ** 2
    return -((x - loc) ** 2) / (2 * var) - scale.log() - LOG_SQRT_2PI


def beta_log_prob(z, alpha, beta):
    """torch.distributions.Beta(alpha, beta).log_prob(z): Dirichlet.log_prob of (z, 1 - z) with
    concentration (alpha, beta): xlogy(c - 1, v).sum + lgamma(c.sum) - lgamma(c).sum."""
    alpha, beta = torch.broadcast_tensors(alpha, beta)
    conc = torch.stack([alpha, beta], -1)
    v = torch.stack([z, 1.0 - z], -1)
    return torch.xlogy(conc - 1.0, v).sum(-1) + torch.lgamma(conc.sum(-1)) - torch.lgamma(conc).sum(-1)


def latent_log_prob(z, p1, p2, dist):
    """log q(z) per latent dimension: Normal(loc=p1, scale=p2) or Beta(p1, p2)."""
    return beta_log_prob(z, p1, p2) if dist == "beta" else normal_log_prob(z, p1, p2)


def uniform_prior_log_prob(z):
    """Uniform(0., 1. + 1e-8).log_prob(z) with float32 bounds (vae.py:160): the upper bound rounds
    to 1.0; log(1[low <= z] * 1[high > z]) - log(high - low)."""
    low, high = torch.tensor(0.), torch.tensor(1. + 1e-8)
    lb = low.le(z).type_as(low)
    ub = high.gt(z).type_as(low)
    return torch.log(lb.mul(ub)) - torch.log(high - low)


def std_normal_log_prob(z):
    """Normal(0.,1.).log_prob(z) (prior, vae.py:162).  The reference's prior holds 0-dim
    float32 tensors loc=0, scale=1: var=1, log(scale)=0."""
    one = torch.tensor(1.0)
    return -((z - torch.tensor(0.0)) ** 2) / (2 * one ** 2) - one.log() - LOG_SQRT_2PI


def bernoulli_log_prob(x, logits):
    """torch.distributions.Bernoulli(logits).log_prob = -BCEWithLogits(logits, x)."""
    logits, x = torch.broadcast_tensors(logits, x)  # as torch.distributions.Bernoulli.log_prob does
    return -F.binary_cross_entropy_with_logits(logits, x, reduction="none")


def studentt_params(params, dist):
    """'... (3 d) -> df, loc, scale' (vae.py:220-229): df = softplus(raw), scale = softplus(raw) (+ 2^-2 for
    studentt_with_eps22)."""
    d = params.shape[-1] // 3
    df, loc, scale = F.softplus(params[..., :d]), params[..., d:2 * d], F.softplus(params[..., 2 * d:])
    if dist == "studentt_with_eps22":
        scale = scale + 2 ** -2
    return df, loc, scale


def studentt_log_prob(x, df, loc, scale):
    """torch.distributions.StudentT.log_prob"""
    y = (x - loc) / scale
    Z = (scale.log() + 0.5 * df.log() + 0.5 * math.log(math.pi) + torch.lgamma(0.5 * df) - torch.lgamma(0.5 * (df + 1.)))
    return -0.5 * (df + 1.) * torch.log1p(y ** 2. / df) - Z


def logitnormal_params(params):
    """vae.py:244-247"""
    d = params.shape[-1] // 2
    return params[..., :d], F.softplus(params[..., d:]) + EPSILON


def logitnormal_log_prob(v, loc, scale):
    """LogitNormal (irwg/utils/logit_normal.py) = TransformedDistribution(Normal, SigmoidTransform).log_prob:
    u = SigmoidTransform.inv(v) with v clamped to [tiny, 1 - eps]; 0 - log|dv/du| + Normal.log_prob(u),
    log|dv/du| = -softplus(-u) - softplus(u)."""
    finfo = torch.finfo(v.dtype)
    y = v.clamp(min=finfo.tiny, max=1. - finfo.eps)
    u = y.log() - (-y).log1p()
    ladj = -F.softplus(-u) - F.softplus(u)
    return (0.0 - ladj) + normal_log_prob(u, loc, scale)


def gen_log_prob(x, params, dist):
    if dist == "bern":
        return bernoulli_log_prob(x, params)
    if dist in ("studentt", "studentt_with_eps22"):
        return studentt_log_prob(x, *studentt_params(params, dist))
    if dist == "logitnormal_with_eps":
        return logitnormal_log_prob(x, *logitnormal_params(params))
    loc, scale = split_loc_scale(params, dist)
    return normal_log_prob(x, loc, scale)


def gen_sample(params, dist, draw, draw2=None):
    """Sample of the generator distribution from an injected draw: standard-normal eps for
    the Normal families (torch.normal(loc, scale) == loc + scale*eps bitwise on CPU),
    float32 uniform u for Bernoulli (torch.bernoulli(p) == (u < p)).  Student-t (StudentT.rsample):
    loc + scale * (n * rsqrt(c / df)) from the standard normal n = draw and the chi-square(df) draw c = draw2.
    Logit-normal: the clipped sigmoid of the Normal sample (SigmoidTransform._call)."""
    if dist == "bern":
        return (draw < torch.sigmoid(params)).to(params.dtype)
    if dist in ("studentt", "studentt_with_eps22"):
        df, loc, scale = studentt_params(params, dist)
        return loc + scale * (draw * torch.rsqrt(draw2 / df))
    if dist == "logitnormal_with_eps":
        loc, scale = logitnormal_params(params)
        finfo = torch.finfo(params.dtype)
        return torch.clamp(torch.sigmoid(loc + scale * draw), min=finfo.tiny, max=1. - finfo.eps)
    loc, scale = split_loc_scale(params, dist)
    return loc + scale * draw


def softplus_inverse(x):
    """irwg/utils/softplus_inverse.py:3-7"""
    return x + torch.log(-torch.expm1(-x))


def replenish_params(B, R, Zd, dist):
    """Raw proposal parameters of the R prior-replenishing proposals
    (vae_imputation.py:563-577): loc 0, raw scale softplus^-1(1 [- 1e-8])."""
    means = torch.zeros(B, R, Zd)
    scales = torch.ones(B, R, Zd)
    if dist == "beta":  # Beta(1, 1) = the uniform prior (:578-585)
        return torch.cat([softplus_inverse(scales), softplus_inverse(scales)], dim=-1)
    if dist == "normal":
        raw = softplus_inverse(scales)
    elif dist == "normal_with_eps":
        raw = softplus_inverse(scales - EPSILON)
    else:
        raise NotImplementedError(dist)
    return torch.cat([means, raw], dim=-1)


def _join(a, b):
    return torch.cat([a, b], dim=-1)


def distribution_param_tempering(params, dist, temperature):
    """VAE.distribution_param_tempering (vae.py:258-306) on raw parameters '... (2 d)'."""
    d = params.shape[-1] // 2
    p1, p2 = params[..., :d], params[..., d:]
    if dist == "normal":
        scale = F.softplus(p2) * temperature
        return _join(p1, softplus_inverse(scale))
    if dist == "normal_with_eps":
        scale = F.softplus(p2) + EPSILON
        scale = (scale * temperature) - EPSILON
        return _join(p1, softplus_inverse(scale))
    if dist == "beta":
        return _join(softplus_inverse(F.softplus(p1) * temperature), softplus_inverse(F.softplus(p2) * temperature))
    raise NotImplementedError(dist)


def anneal_var_distribution_to_prior(params, M, dist, anneal_type, prior="std_normal"):
    """VAE.anneal_var_distribution_to_prior_based_on_missingness (vae.py:308-401).  M bool
    [B,1,D] broadcasts against params [B,K,2d]."""
    if anneal_type == "linear":
        const = None
    elif anneal_type == "linear_0.25":
        const = 0.25
    else:
        raise NotImplementedError(anneal_type)

    def anneal(k):
        mf = (M.shape[-1] - M.sum(dim=-1, keepdim=True)) / M.shape[-1]
        if const is None:
            return 1 * (1 - mf) + k * mf
        return 1 * (1 - mf) + k * const * mf

    d = params.shape[-1] // 2
    p1, p2 = params[..., :d], params[..., d:]
    if dist in ("normal", "normal_with_eps") and prior == "std_normal":
        eps = EPSILON if dist == "normal_with_eps" else None
        scale = F.softplus(p2)
        if eps is not None:
            scale = scale + eps
        k_m = anneal(1.0 / scale)
        scale = scale * k_m
        if eps is not None:
            scale = scale - eps
        raw = softplus_inverse(scale)
        loc = p1 * anneal(0)
        return _join(loc, raw)
    if dist == "beta" and prior == "uniform":
        alpha, beta = F.softplus(p1), F.softplus(p2)
        return _join(softplus_inverse(alpha * anneal(1.0 / alpha)), softplus_inverse(beta * anneal(1.0 / beta)))
    raise NotImplementedError((dist, prior))


def transform_proposal_params(params, M, dist, temperature=None, anneal_type=None, prior="std_normal"):
    """Tempering then annealing, as every call site of the reference orders them
    (vae_imputation.py:553-561, test_step_vae_sampling.py:1222-1230, 1672-1680)."""
    if temperature is not None:
        params = distribution_param_tempering(params, dist, temperature)
    if anneal_type is not None:
        params = anneal_var_distribution_to_prior(params, M, dist, anneal_type, prior)
    return params


# --------------------------------------------------------------------------------------
# categorical sampling with injected uniforms
# --------------------------------------------------------------------------------------
def categorical_probs(logits):
    """torch.distributions.Categorical(logits=l).probs: l - logsumexp(l), then softmax."""
    nl = logits - logits.logsumexp(dim=-1, keepdim=True)
    return F.softmax(nl, dim=-1)


def multinomial_indices(probs, u):
    """torch.multinomial(probs[rows,S], n>1, replacement=True) on CPU with its float64
    uniforms u[rows,n] made explicit: float32 running sum, divide by the total, force the
    last entry to 1, index = number of entries with cum < u (left bisection)."""
    p = probs.detach().to(torch.float32).numpy()
    u = np.asarray(u, dtype=np.float64)
    rows, S = p.shape
    out = np.empty(u.shape, dtype=np.int64)
    for r in range(rows):
        cum = np.cumsum(p[r], dtype=np.float32)  # sequential float32 accumulation
        cum = (cum / cum[-1]).astype(np.float32)
        cum[-1] = np.float32(1.0)
        out[r] = np.searchsorted(cum.astype(np.float64), u[r], side="left")
    return torch.from_numpy(out)


def systematic_indices(log_weights, U, num_dependent_samples):
    """vae_imputation.py:804-834 with its one uniform per row, U[..., 1] float32, injected."""
    lw = log_weights - torch.logsumexp(log_weights, dim=-1, keepdim=True)
    probs = torch.exp(lw)
    probs = probs / probs.sum(dim=-1, keepdim=True)
    m = num_dependent_samples
    n = probs.cumsum(-1).mul(m).add(U)
    n = n.floor().clamp(min=0, max=m).long()
    diff = probs.new_zeros(probs.shape[:-1] + (m + 1,))
    diff.scatter_add_(-1, n, torch.ones_like(probs))
    return diff[..., :-1].cumsum(-1).long()


# --------------------------------------------------------------------------------------
# importance weights
# --------------------------------------------------------------------------------------
def dmis_mixture_logprob(Z, q_loc, q_scale, latent_dist="normal"):
    """log of the deterministic-mixture proposal density at every proposal
    (weights.py:69-85).  Z[I,B,K,Zd], q_loc/q_scale[B,K,Zd] -> [I,B,K]:
    logsumexp_j sum_d log N(Z[i,b,k]; loc[b,j], scale[b,j]) - log K."""
    K = Z.shape[2]
    # comp[j,i,b,k] = sum_d log N(Z[i,b,k,d]; loc[b,j,d], scale[b,j,d])
    Zs = Z.unsqueeze(0).transpose(0, -2)  # [K(k),I,B,1,Zd]  (value k sits on axis 0)
    lp = latent_log_prob(Zs, q_loc, q_scale, latent_dist)  # broadcast -> [K(k),I,B,K(j),Zd]
    lp = lp.sum(-1)
    lp = torch.logsumexp(lp, dim=-1, keepdim=True)
    lp = lp - torch.log(torch.tensor(K))
    return lp.transpose(0, -1).squeeze(0)  # [I,B,K]


def grouped_dmis_mixture_logprob(Z, q_loc, q_scale, cluster_size, latent_dist="normal", pad_to=None):
    """Mixture restricted to the proposal's own group (vae_imputation.py:407-434).  Layout
    '(c k)': member j belongs to group j mod (K/c); a group has c members.  With ``pad_to`` the
    K proposals are laid out as if there were pad_to of them (group j mod (pad_to/c)); the
    missing members contribute nothing and the mixture is normalised by the number of members
    actually present (:395-405, 423-427)."""
    if pad_to is not None:
        return _grouped_dmis_padded(Z, q_loc, q_scale, cluster_size, latent_dist, pad_to)
    I, B, K, Zd = Z.shape
    c = cluster_size
    G = K // c
    Zg = Z.reshape(I, B, c, G, Zd).permute(0, 1, 3, 2, 4)        # i b g c d
    loc = q_loc.reshape(B, c, G, Zd).permute(0, 2, 1, 3)          # b g c d
    scale = q_scale.reshape(B, c, G, Zd).permute(0, 2, 1, 3)
    Zs = Zg.unsqueeze(0).transpose(0, -2)                        # [c(val),I,B,G,1,Zd]
    lp = latent_log_prob(Zs, loc, scale, latent_dist).sum(-1)     # [c(val),I,B,G,c(comp)]
    lp = torch.logsumexp(lp, dim=-1, keepdim=True)
    lp = lp - torch.log(torch.tensor(c))
    lp = lp.transpose(0, -1).squeeze(0)                           # [I,B,G,c]
    return lp.permute(0, 1, 3, 2).reshape(I, B, K)                # i b (c g)


def _grouped_dmis_padded(Z, q_loc, q_scale, c, latent_dist, pad_to):
    I, B, K, Zd = Z.shape
    G = pad_to // c
    out = torch.empty(I, B, K)
    for g in range(G):
        members = [j for j in range(g, pad_to, G) if j < K]
        if not members:
            continue
        mi = torch.tensor(members)
        Zg = Z[:, :, mi]                                              # i b m d
        lp = latent_log_prob(Zg.unsqueeze(3), q_loc[:, mi].unsqueeze(1), q_scale[:, mi].unsqueeze(1),
                             latent_dist).sum(-1)                     # i b m(value) m(component)
        lp = torch.logsumexp(lp, dim=-1) - torch.log(torch.tensor(len(members)))
        out[:, :, mi] = lp
    return out


def lair_log_weights(X, M, Z, q_loc, q_scale, gen_params, gen_dist, weighting, cluster_size=None,
                     latent_dist="normal", prior="std_normal", pad_to=None):
    """log w = sum_d M log p(x_d|z) + sum log p(z) - log q_mix(z)  (weights.py:46-57)."""
    gl = (gen_log_prob(X, gen_params, gen_dist) * M).sum(-1)
    pl = (uniform_prior_log_prob(Z) if prior == "uniform" else std_normal_log_prob(Z)).sum(-1)
    if weighting == "dmis":
        ql = dmis_mixture_logprob(Z, q_loc, q_scale, latent_dist)
    elif weighting == "dmis_within_groups":
        ql = grouped_dmis_mixture_logprob(Z, q_loc, q_scale, cluster_size, latent_dist, pad_to)
    elif weighting == "smis":
        ql = latent_log_prob(Z, q_loc, q_scale, latent_dist).sum(-1)
    else:
        raise NotImplementedError(weighting)
    return gl + pl - ql, gl, pl, ql


# --------------------------------------------------------------------------------------
# resampling
# --------------------------------------------------------------------------------------
def gr_resample(log_w, u, method="multinomial"):
    """Global resampling (vae_imputation.py:284-308).  log_w[I,B,K'] -> idx[B,K'] into the
    flattened '(i k)' axis, and normalised log-weights [B, I*K'].  ``u`` is float64
    [B,K'] for multinomial, float32 [B,1] for systematic."""
    I, B, K = log_w.shape
    lw = log_w.permute(1, 0, 2).reshape(B, I * K)
    if method == "multinomial":
        idx = multinomial_indices(categorical_probs(lw), u)
    elif method == "systematic":
        idx = systematic_indices(lw, u, K)
    else:
        raise NotImplementedError(method)
    return idx, lw - lw.logsumexp(dim=-1, keepdim=True)


def gather_gr(params, idx):
    """params[I,B,K',P] -> [B,n,P] taking flattened '(i k)' index idx[B,n]."""
    I, B, K = params.shape[:3]
    flat = params.permute(1, 0, 2, 3).reshape(B, I * K, -1)
    return flat[torch.arange(B)[:, None], idx]


def glr_resample(log_w, u, cluster_size, method="multinomial", pad_to=None):
    """Global-local resampling (vae_imputation.py:335-378).  With ``pad_to`` the K' proposals are
    padded with weight-zero members up to pad_to, which changes the group layout (G = pad_to/c);
    the returned slot order and normalised log-weights follow the reference's (:339-341,371-374).  Group g holds the
    members j with j mod G == g (G = K'/c); it draws c indices among its I*c members laid
    out '(i c)'.  Returns local idx[B,G,c] (draw t of group g), the flattened global
    '(i k)' index of every output slot in the reference's output order '(c k)'
    (slot t*G+g), and norm_log_weights[B, G*I*c]."""
    I, B, K = log_w.shape
    c = cluster_size
    if pad_to is not None:
        Kreal = K
        log_w = torch.cat([log_w, torch.ones(I, B, pad_to - K) * float("-inf")], dim=-1)
        K = pad_to
    else:
        Kreal = None
    G = K // c
    lw = log_w.reshape(I, B, c, G).permute(1, 3, 0, 2).reshape(B, G, I * c)  # b g (i c)
    if method == "multinomial":
        probs = categorical_probs(lw).reshape(B * G, I * c)
        idx = multinomial_indices(probs, np.asarray(u).reshape(B * G, c)).reshape(B, G, c)
    elif method == "systematic":
        idx = systematic_indices(lw, torch.as_tensor(u).reshape(B, G, 1), c)
    else:
        raise NotImplementedError(method)
    # local (i, cc) -> global proposal index cc*G + g within importance sample i
    i_loc = idx // c
    c_loc = idx % c
    g_ar = torch.arange(G)[None, :, None]
    kk = c_loc * G + g_ar                                     # proposal index inside importance sample i
    glob = i_loc * (Kreal or K) + kk                          # [B,G,c] flattened (i k) over the real proposals
    glob_out = glob.permute(0, 2, 1).reshape(B, c * G)        # slot t*G+g
    nlw = lw - lw.logsumexp(dim=-1, keepdim=True)
    if Kreal is not None:
        assert bool((kk < Kreal).all()), "a padded member was drawn"
        nlw = nlw.reshape(B, G, I, c).permute(2, 0, 3, 1).reshape(I, B, K)[:, :, :Kreal]   # i b (c k), real members
        return idx, glob_out, nlw.permute(1, 0, 2).reshape(B, I * Kreal)
    return idx, glob_out, nlw.reshape(B, G * I * c)


def single_indices(probs, e):
    """torch.multinomial(probs[rows,S], 1, replacement=True) with its Exp(1) variates e[rows,S]
    made explicit: argmax(probs / e) (ATen's single-sample path)."""
    return torch.argmax(probs.to(torch.float32) / torch.as_tensor(e, dtype=torch.float32), dim=-1)


def lr_resample(log_w, draw, method="multinomial"):
    """Local resampling (vae_imputation.py:310-333): chain k draws one of its I importance samples.
    ``draw``: Exp(1) variates [B*K, I] (multinomial) or uniforms [B,K,1] (systematic).  Returns
    idx[B,K] in [0,I), the flattened '(i k)' source of every output slot, and
    norm_log_weights[B, K*I] ('(k i)' order)."""
    I, B, K = log_w.shape
    lw = log_w.permute(1, 2, 0)                                # b k i
    if method == "multinomial":
        idx = single_indices(categorical_probs(lw).reshape(B * K, I), draw).reshape(B, K)
    elif method == "systematic":
        idx = systematic_indices(lw, torch.as_tensor(draw).reshape(B, K, 1), 1)[..., 0]
    else:
        raise NotImplementedError(method)
    glob = idx * K + torch.arange(K)[None, :]
    nlw = lw - lw.logsumexp(dim=-1, keepdim=True)
    return idx, glob, nlw.reshape(B, K * I)


def ir_resample(log_w, draw, method="multinomial"):
    """Independent resampling (vae_imputation.py:460-482 with select_gen_params_for_ir :730-738).
    The '(i k)' axis of an example is cut into K rows of I consecutive entries (a plain reshape);
    row r draws one of its entries, output slot r takes flattened proposal r*I + idx.  The
    "normalised" log-weights it returns are the log-weights themselves, '(i k)' order."""
    I, B, K = log_w.shape
    flat = log_w.permute(1, 0, 2).reshape(B, I * K)
    lw = flat.reshape(B, K, I)
    if method == "multinomial":
        idx = single_indices(categorical_probs(lw).reshape(B * K, I), draw).reshape(B, K)
    elif method == "systematic":
        idx = systematic_indices(lw, torch.as_tensor(draw).reshape(B, K, 1), 1)[..., 0]
    else:
        raise NotImplementedError(method)
    glob = torch.arange(K)[None, :] * I + idx
    return idx, glob, flat


# --------------------------------------------------------------------------------------
# LAIR iteration
# --------------------------------------------------------------------------------------
def lair_iteration(enc, dec, X, M, draws, *, num_imp_samples=1, weighting="dmis",
                   resampling_scheme="gr", resampling_method="multinomial",
                   glr_cluster_size=None, num_prior_replenish_proposals=0,
                   latent_dist="normal_with_eps", gen_dist="normal_with_eps",
                   return_resampled_latents=False, weight_alpha=None,
                   encoder=None, decoder=None, prior="std_normal",
                   var_proposal_temperature=None, var_proposal_anneal_type_to_prior=None,
                   prior_proposals=False, hist_X=None, glr_pad_to=None):
    """One LAIR step (vae_imputation.py:484-726) with injected draws.  ``hist_X`` [B,H,D]:
    historical imputations appended as extra proposals (:504-545); they take part in the weights
    and the resampling, only the first K chains are imputed (:677-680).  ``glr_pad_to``: the
    iterations before the history is used lay their groups out as if it were (:546-548).  ``prior_proposals``:
    standard_importance_resampling instead (:904-998) -- no encoder, every chain proposes from the
    prior (N(0,1); the uniform prior as Uniform(0, 1+1e-8), 'eps_z' then holds torch.rand draws).  For latent_dist='beta'
    'eps_z' holds the Beta samples themselves (torch draws them by rejection sampling).

    X[B,K,D] float32, M[B,1,D] bool.  draws: 'eps_z'[I,B,K',Zd], 'u_res' (float64
    [B,K'] for gr / [B*G,c] for glr multinomial; float32 [B,1] / [B,G,1] systematic),
    optional 'u_res2' (same shape, used when return_resampled_latents),
    'eps_x'[B,K,D] (uniforms for Bernoulli).  ``encoder``/``decoder`` override the MLP
    forward (used for the conv nets)."""
    encoder = encoder or (lambda x: resfc_forward(enc, x))
    decoder = decoder or (lambda z: resfc_forward(dec, z))
    B, K_keep, D = X.shape
    if hist_X is not None:
        X = torch.cat([X, hist_X], dim=1)                     # :545
    B, K, D = X.shape
    R = num_prior_replenish_proposals
    I = num_imp_samples
    Mf = M  # bool [B,1,D]
    out = {}

    if prior_proposals:
        assert R == 0
        Zd0 = net_input_dim(dec)
        if prior == "uniform":
            return _standard_ir_uniform(dec, X, M, draws, I, weighting, resampling_scheme, resampling_method,
                                        glr_cluster_size, gen_dist, weight_alpha, decoder)
        # torch.ones(shape) * prior.mean / prior.scale (:923-924): loc 0, scale exactly 1
        q_loc_p, q_scale_p = torch.zeros(B, K, Zd0), torch.ones(B, K, Zd0)
        latent_dist = "normal"
    var_params = None if prior_proposals else encoder(X)      # [B,K,2Zd]
    if prior_proposals:
        Zd = Zd0
        Xp = X
        q_loc, q_scale = q_loc_p, q_scale_p
        Kp = K
        Z = q_loc + q_scale * draws["eps_z"]
        gen_params = decoder(Z)
        log_w, gl, pl, ql = lair_log_weights(Xp, Mf, Z, q_loc, q_scale, gen_params, gen_dist,
                                             weighting, glr_cluster_size, latent_dist, prior)
        return _lair_finish(out, X, Mf, Z, gen_params, log_w, gl, pl, ql, q_loc, q_scale, var_params, draws, I, K, Kp,
                            resampling_scheme, resampling_method, glr_cluster_size, gen_dist, weight_alpha,
                            return_resampled_latents)
    var_params = transform_proposal_params(var_params, M, latent_dist, var_proposal_temperature,
                                           var_proposal_anneal_type_to_prior, prior)   # :553-561
    Zd = var_params.shape[-1] // 2
    Xp = X
    if R > 0:
        var_params = torch.cat([var_params, replenish_params(B, R, Zd, latent_dist)], dim=1)
        Xp = torch.cat([X, X[:, :1].expand(B, R, D)], dim=1)
    q_loc, q_scale = split_loc_scale(var_params, latent_dist)
    Kp = K + R
    if latent_dist == "beta":
        # Beta.sample() returns a stride-2 view into the Dirichlet sample (x, 1-x); torch's CPU linear
        # layer rounds differently (last bit) on that layout than on a contiguous tensor, so the
        # injected samples are given the same layout
        Z = torch.stack([draws["eps_z"], 1.0 - draws["eps_z"]], -1).select(-1, 0)   # [I,B,K',Zd]
    else:
        Z = q_loc + q_scale * draws["eps_z"]                  # [I,B,K',Zd]
    gen_params = decoder(Z)                                   # [I,B,K',P]
    log_w, gl, pl, ql = lair_log_weights(Xp, Mf, Z, q_loc, q_scale, gen_params, gen_dist,
                                         weighting, glr_cluster_size, latent_dist, prior, glr_pad_to)
    return _lair_finish(out, X[:, :K_keep], Mf, Z, gen_params, log_w, gl, pl, ql, q_loc, q_scale, var_params, draws, I,
                        K_keep, Kp, resampling_scheme, resampling_method, glr_cluster_size, gen_dist, weight_alpha,
                        return_resampled_latents, glr_pad_to)


def _lair_finish(out, X, Mf, Z, gen_params, log_w, gl, pl, ql, q_loc, q_scale, var_params, draws, I, K, Kp,
                 resampling_scheme, resampling_method, glr_cluster_size, gen_dist, weight_alpha,
                 return_resampled_latents, glr_pad_to=None):
    """Resampling, sampling of the missing values, imputation and the ESS / perplexity statistics
    (vae_imputation.py:642-726), shared by the LAIR iteration and standard importance resampling."""
    B = X.shape[0]
    t_log_w = log_w if weight_alpha is None else log_w * weight_alpha

    def resample(u):
        if resampling_scheme == "gr":
            idx, nlw = gr_resample(t_log_w, u, resampling_method)
            return idx, idx, nlw
        if resampling_scheme == "glr":
            loc_idx, glob, nlw = glr_resample(t_log_w, u, glr_cluster_size, resampling_method, glr_pad_to)
            return loc_idx, glob, nlw
        if resampling_scheme == "lr":
            return lr_resample(t_log_w, u, resampling_method)
        if resampling_scheme == "ir":
            return ir_resample(t_log_w, u, resampling_method)
        raise NotImplementedError(resampling_scheme)

    idx_raw, idx_glob, norm_log_w = resample(draws["u_res"])
    mis_params = gather_gr(gen_params, idx_glob)              # [B,K',P]
    latents_resampled = None
    if return_resampled_latents:
        _, idx2, _ = resample(draws["u_res2"])
        latents_resampled = gather_gr(Z, idx2)
        out["idx2"] = idx2
    mis_params = mis_params[:, :K]
    X_m = gen_sample(mis_params, gen_dist, draws["eps_x"], draws.get("chi2_x"))
    X_new = X * Mf + X_m * (~Mf)

    lwr = log_w.permute(1, 0, 2).reshape(B, I * Kp)
    lnw = lwr - torch.logsumexp(lwr, dim=-1, keepdim=True)
    ess = torch.exp(-torch.logsumexp(2 * lnw, -1))
    tlwr = t_log_w.permute(1, 0, 2).reshape(B, I * Kp)
    tlnw = tlwr - torch.logsumexp(tlwr, dim=-1, keepdim=True)
    tess = torch.exp(-torch.logsumexp(2 * tlnw, -1))
    perplexity = torch.exp(-torch.sum(lnw * torch.exp(lnw), dim=-1))

    out.update(dict(X_new=X_new, ess=ess, transformed_ess=tess, perplexity=perplexity,
                    total_num_props=I * Kp, norm_log_weights=norm_log_w, Z_imp=Z,
                    log_weights=log_w, q_loc=q_loc, q_scale=q_scale, var_params=var_params,
                    generator_params=gen_params, gen_loglik=gl, prior_logprob=pl,
                    mix_logprob=ql, idx=idx_raw, idx_global=idx_glob, mis_params=mis_params,
                    X_m=X_m, latents_resampled=latents_resampled))
    return out


def _standard_ir_uniform(dec, X, M, draws, I, weighting, scheme, method, cluster, gen_dist, weight_alpha, decoder):
    """standard_importance_resampling under the uniform prior (:925-928): proposals
    Uniform(0, 1+1e-8) for every chain; sample = rand*(high-low)+low."""
    B, K, D = X.shape
    low, high = torch.tensor(0.), torch.tensor(1. + 1e-8)
    Z = low + draws["eps_z"] * (high - low)
    gen_params = decoder(Z)
    gl = (gen_log_prob(X, gen_params, gen_dist) * M).sum(-1)
    pl = uniform_prior_log_prob(Z).sum(-1)
    comp = uniform_prior_log_prob(Z).sum(-1)          # every component is the same uniform density
    if weighting == "dmis":
        ql = torch.logsumexp(comp.unsqueeze(-1).expand(*comp.shape, K), dim=-1) - torch.log(torch.tensor(K))
    elif weighting == "smis":
        ql = comp
    else:
        raise NotImplementedError(weighting)
    log_w = gl + pl - ql
    Zd = Z.shape[-1]
    return _lair_finish({}, X, M, Z, gen_params, log_w, gl, pl, ql, torch.ones(B, K, Zd), torch.ones(B, K, Zd), None,
                        draws, I, K, K, scheme, method, cluster, gen_dist, weight_alpha, False)


def pseudo_gibbs_iteration(enc, dec, X, M, draws, *, latent_dist="normal_with_eps",
                           gen_dist="normal_with_eps", clip_min=None, clip_max=None,
                           encoder=None, decoder=None):
    """vae_imputation.py:19-45.  draws: 'eps_z'[B,K,Zd], 'eps_x'[B,K,D]."""
    encoder = encoder or (lambda x: resfc_forward(enc, x))
    decoder = decoder or (lambda z: resfc_forward(dec, z))
    q_loc, q_scale = split_loc_scale(encoder(X), latent_dist)
    Z = q_loc + q_scale * draws["eps_z"]
    params = decoder(Z)
    X_m = gen_sample(params, gen_dist, draws["eps_x"], draws.get("chi2_x"))
    if clip_min is not None:
        X_m = torch.max(X_m, clip_min)
    if clip_max is not None:
        X_m = torch.min(X_m, clip_max)
    bad = ~X_m.isfinite()
    X_m = torch.where(bad, X, X_m)
    return X * M + X_m * (~M), Z


def vae_samples_imputation(dec, X, M, draws, *, gen_dist="normal_with_eps", decoder=None, prior="std_normal"):
    """Initial imputation from unconditional model samples
    (test_step_vae_sampling.py:279-287 with vae.py:166-178).  draws: 'z'[B*K,Zd] standard
    normal, 'eps_x'[B*K,D]."""
    decoder = decoder or (lambda z: resfc_forward(dec, z))
    B, K, D = X.shape
    z = draws["z"] * (1. + 1e-8) if prior == "uniform" else draws["z"]  # 'z' = torch.rand under the uniform prior
    X_imp = gen_sample(decoder(z), gen_dist, draws["eps_x"], draws.get("chi2_x")).reshape(B, K, D)
    Xc = X.clone()
    Xc[(~M).expand(B, K, D)] = 0.0
    return Xc * M + X_imp * (~M)


# --------------------------------------------------------------------------------------
# per-iteration diagnostics (irwg/sampling/imputation_metrics.py)
# --------------------------------------------------------------------------------------
def imputation_rmse_metric(X_imp, X_true, M):
    """imputation_metrics.py:85-102.  X_imp[B,K,D], X_true[B,1,D], M bool [B,1,D] -> [B,K]."""
    M_not = ~M
    se = (((X_imp - X_true) ** 2) * M_not).sum(-1)
    count = M_not.sum(-1)
    count[count == 0] = 1
    return (se / count) ** 0.5


def imputation_mae_metric(X_imp, X_true, M):
    """imputation_metrics.py:133-143."""
    M_not = ~M
    ae = (torch.absolute(X_imp - X_true) * M_not).sum(-1)
    count = M_not.sum(-1)
    count[count == 0] = 1
    return ae / count


def imputation_latent_wass(enc, X_imp, X_true, latent_dist="normal_with_eps"):
    """imputation_metrics.py:200-213: squared 2-Wasserstein distance between the Gaussian
    posteriors of the imputed and the true datapoints."""
    li, si = split_loc_scale(resfc_forward(enc, X_imp), latent_dist)
    lt, st = split_loc_scale(resfc_forward(enc, X_true), latent_dist)
    vi, vt = si ** 2, st ** 2
    W = torch.sum((li - lt) ** 2, dim=-1)
    W += torch.sum(vi + vt - 2 * (vi * vt) ** 0.5, dim=-1)
    return W


def final_resample(latent_proposals, log_weights, u, num_samples):
    """LAIR final resampling over all T*I*K' stored proposals
    (test_step_vae_sampling.py:1998-2009).  latent_proposals[T,I,B,K',Zd],
    log_weights[T,I,B,K'], u float64 [B,num_samples] -> idx[B,n], Z[B,n,Zd]."""
    T, I, B, Kp, Zd = latent_proposals.shape
    lp = latent_proposals.permute(2, 0, 1, 3, 4).reshape(B, T * I * Kp, Zd)
    lw = log_weights.permute(2, 0, 1, 3).reshape(B, T * I * Kp)
    idx = multinomial_indices(categorical_probs(lw), u)
    Z = lp[torch.arange(B)[:, None], idx]
    return idx, Z


# --------------------------------------------------------------------------------------
# LAIR history of imputations (irwg/sampling/utils.py:5-68)
# --------------------------------------------------------------------------------------
class ImputationHistory:
    """ImputationHistoryQueue with its integer draws injected.  Ring [T,B,K,D]; stratified
    sampling: the stored history is cut into H windows, one index per window (the same for the
    whole batch), shifted by the tail; then one chain per example (gr) or one chain per
    historical proposal (glr)."""

    def __init__(self, max_len, batch_shape, dtype=torch.float32):
        self.T, self.tail, self.full = max_len, -1, False
        self.queue = torch.zeros((max_len,) + tuple(batch_shape), dtype=dtype)

    def __len__(self):
        assert self.tail > -1
        return self.T if self.full else self.tail + 1

    def enqueue(self, X):
        self.tail = (self.tail + 1) % self.T
        if self.tail == self.T - 1:
            self.full = True
        self.queue[self.tail] = X

    def window(self, H):
        """Exclusive upper bound of the H window draws (utils.py:33-36)."""
        return len(self) // H

    def slots(self, r, H):
        """Ring slots of the window draws r[H] (utils.py:37-39)."""
        w = self.window(H)
        return (torch.as_tensor(r) + torch.arange(0, H * w, w) + self.tail) % len(self)

    def sample(self, r, chain_of_example, H):
        """sample_history (utils.py:28-49) -> [H,B,D]."""
        hp = self.queue[self.slots(r, H)]
        B = self.queue.shape[1]
        return hp[:, torch.arange(B), torch.as_tensor(chain_of_example)]

    def sample_with_chain_idx(self, r, chain_of_proposal, H):
        """sample_history_with_chain_idx_for_each_sample (utils.py:51-68) -> [H,B,D]."""
        hp = self.queue[self.slots(r, H)]
        return hp[torch.arange(H), :, torch.as_tensor(chain_of_proposal)]


def glr_history_chain_idx(K, H, cluster_size):
    """Chain of every historical proposal when global-local resampling is combined with the
    history, so that each group only receives history of its own chains
    (vae_imputation.py:505-537)."""
    assert (K + H) % cluster_size == 0
    G = (K + H) // cluster_size
    assert G <= K
    num_rem = math.ceil(K / G) * G - K
    base = torch.cat([torch.arange(K), torch.arange(G - num_rem, G)])
    reps = (K + H) // len(base)
    chain = torch.cat(reps * [base])
    if (K + H) % len(base) != 0:
        chain = torch.cat([chain, base[:(K + H) - len(chain)]])
    return chain[K:]


# --------------------------------------------------------------------------------------
# AC-MWG
# --------------------------------------------------------------------------------------
class RestrictedHistory:
    """irwg/sampling/utils.py:83-237 with the per-chain integer draws injected.

    Ring of past imputations ``[T,B,K,D]`` and per-chain availability counters.  The
    reference draws ``randint(0, max(avail,1))`` per chain (b-major); here the caller passes
    float64 uniforms or the integers themselves."""

    def __init__(self, max_len, batch_shape, dtype=torch.float32):
        self.T = max_len
        self.tail = -1
        self.full = False
        self.queue = torch.zeros((max_len,) + tuple(batch_shape), dtype=dtype)
        self.avail = torch.zeros(tuple(batch_shape)[:2], dtype=torch.int64)

    def __len__(self):
        assert self.tail > -1
        return self.T if self.full else self.tail + 1

    def update_available(self, mask):
        self.avail[mask] = len(self)

    def enqueue(self, X):
        if self.full:
            self.avail -= 1
            self.avail[self.avail < 0] = 0
        self.tail = (self.tail + 1) % self.T
        if self.tail == self.T - 1:
            self.full = True
        self.queue[self.tail] = X

    def draw_bounds(self):
        """Exclusive upper bound of each chain's integer draw: max(avail, 1)."""
        return torch.clamp(self.avail, min=1)

    def slots_from_draws(self, r):
        """Ring slot for integer draws r[B,K] in [0, bound)."""
        if self.full:
            return (r + self.tail + 1) % self.T
        return r

    def gather(self, slots):
        B, K = slots.shape
        return self.queue[slots, torch.arange(B)[:, None], torch.arange(K)[None, :]]


def acmwg_step(enc, dec, state, M, hist_X, draws, *, prior_mixture_probability,
               latent_dist="normal_with_eps", gen_dist="normal_with_eps",
               encoder=None, decoder=None, joint_target_new=False, joint_target_old=False,
               var_proposal_temperature=None, var_proposal_anneal_type_to_prior=None):
    """One AC-MWG step (test_step_vae_sampling.py:1671-1755).  The plain Metropolis-within-Gibbs
    step (:1221-1269) is the same arithmetic with ``hist_X`` = the current imputations, no prior
    mixture, and both log-likelihoods summed over all dimensions (``joint_target_new/old``);
    AC-MWG's ``ablation_use_mwg_target`` sets ``joint_target_new`` only (:1722-1725 vs :1755).

    state: dict X[B,K,D], Z_old[B,K,Zd], mis_params_old[B,K,P], Z_old_prior_logprob[B,K],
    X_old_logprob[B,K].  hist_X[B,K,D] is the historical imputation the proposal conditions
    on.  draws: 'eps_z'[B,K,Zd], 'z_prior'[B,K,Zd] (standard normal), 'u_prior'[B,K]
    float32, 'u_accept'[B,K] float32, 'eps_x'[B,K,D]."""
    encoder = encoder or (lambda x: resfc_forward(enc, x))
    decoder = decoder or (lambda z: resfc_forward(dec, z))
    X, Z_old = state["X"], state["Z_old"]
    p = prior_mixture_probability
    var_params = transform_proposal_params(encoder(hist_X), M, latent_dist, var_proposal_temperature,
                                           var_proposal_anneal_type_to_prior)   # :1672-1680
    q_loc, q_scale = split_loc_scale(var_params, latent_dist)
    Z_new = q_loc + q_scale * draws["eps_z"]
    out = {}
    if p > 0.0:
        use_prior = draws["u_prior"] < torch.ones_like(draws["u_prior"]) * p
        Z_new = torch.where(use_prior.unsqueeze(-1), draws["z_prior"], Z_new)
        out["use_prior"] = use_prior
    zn_q = normal_log_prob(Z_new, q_loc, q_scale).sum(-1)
    zo_q = normal_log_prob(Z_old, q_loc, q_scale).sum(-1)
    if p > 0.0:
        zn_p = std_normal_log_prob(Z_new).sum(-1)
        zo_p = std_normal_log_prob(Z_old).sum(-1)
        lp = torch.log(torch.tensor(p))
        lq = torch.log(torch.tensor(1 - p))
        zn_q = torch.logsumexp(torch.stack([zn_q + lq, zn_p + lp]), dim=0)
        zo_q = torch.logsumexp(torch.stack([zo_q + lq, zo_p + lp]), dim=0)
    zn_prior = std_normal_log_prob(Z_new).sum(-1)
    mis_params = decoder(Z_new)
    x_new_lp = gen_log_prob(X, mis_params, gen_dist)
    x_new_lp = (x_new_lp if joint_target_new else x_new_lp * M).sum(-1)
    log_accept = (x_new_lp + zn_prior) - (state["X_old_logprob"] + state["Z_old_prior_logprob"]) \
        + (zo_q - zn_q)
    accepted = draws["u_accept"] < torch.exp(log_accept)
    na = ~accepted
    Z_upd = Z_old * na.unsqueeze(-1) + Z_new * accepted.unsqueeze(-1)
    zp_upd = state["Z_old_prior_logprob"] * na + zn_prior * accepted
    mp_upd = state["mis_params_old"] * na.unsqueeze(-1) + mis_params * accepted.unsqueeze(-1)
    X_m = gen_sample(mp_upd, gen_dist, draws["eps_x"], draws.get("chi2_x"))
    X_new = X * M + X_m * (~M)
    x_old_lp = gen_log_prob(X_new, mp_upd, gen_dist)
    x_old_lp = (x_old_lp if joint_target_old else x_old_lp * M).sum(-1)
    new_state = dict(X=X_new, Z_old=Z_upd, mis_params_old=mp_upd, Z_old_prior_logprob=zp_upd,
                     X_old_logprob=x_old_lp)
    out.update(dict(state=new_state, accepted=accepted, log_accept=log_accept, Z_new=Z_new,
                    q_loc=q_loc, q_scale=q_scale, mis_params=mis_params,
                    acceptance=accepted.float().mean()))
    return out
