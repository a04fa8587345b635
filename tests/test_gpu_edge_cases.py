"""GPU parity on the edges of the supported shapes: empty and single-example batches, one chain,
one feature / one latent dimension, the widest inputs and outputs of the fused kernel, the largest
proposal count a wavefront serves (I*K' = 64), ragged row counts, and the error behaviour beyond.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.helpers import assert_close, make_model, seeded_inputs

pytestmark = pytest.mark.gpu
DIAG_RTOL = 1e-4


def run_case(B, K, R, I, D, Zd, H, nbe, nbd, seed, **kw):
    from oracle import cs_oracle as O
    from vae_conditional_sampling_amd.sampling.vae_imputation import importance_resampling_gibbs_iteration
    enc = O.make_resfc_weights(D, 2 * Zd, nbe, H, seed)
    dec = O.make_resfc_weights(Zd, 2 * D, nbd, H, seed + 1)
    model = make_model(enc, dec)
    X, M = seeded_inputs(B, K, D, seed + 2)
    g = torch.Generator().manual_seed(seed + 3)
    Kp = K + R
    scheme = kw.get("resampling_scheme", "gr")
    c = kw.get("glr_cluster_size")
    G = Kp // c if scheme == "glr" else 1
    draws = {"eps_z": torch.randn(I, B, Kp, Zd, generator=g),
             "u_res": torch.rand(B * G, Kp // G, generator=g, dtype=torch.float64),
             "eps_x": torch.randn(B, K, D, generator=g)}
    o = O.lair_iteration(enc, dec, X, M, draws, num_imp_samples=I, num_prior_replenish_proposals=R, **kw)
    out = importance_resampling_gibbs_iteration(model, X.cuda(), M.cuda(), I, num_prior_replenish_proposals=R,
                                                draws=draws, **kw)
    ws = model._cs_workspaces[("lair", B, K, R, I, "cuda:0")]
    assert_close(out[7], o["log_weights"], "log_weights")
    assert torch.equal(ws.idx.cpu(), o["idx_global"]), "resampling indices differ"
    assert_close(out[0], o["X_new"], "X_new")
    assert_close(out[1], o["ess"], "ess", DIAG_RTOL)
    return out


@pytest.mark.parametrize("case", [
    dict(B=1, K=1, R=0, I=1, D=1, Zd=1, H=64, nbe=0, nbd=0),          # smallest everything, no residual blocks
    dict(B=1, K=1, R=1, I=1, D=3, Zd=2, H=64, nbe=1, nbd=1),          # one example, one chain
    dict(B=5, K=63, R=1, I=1, D=7, Zd=3, H=64, nbe=1, nbd=1),         # I*K' = 64: a full wavefront per example
    dict(B=4, K=15, R=1, I=4, D=6, Zd=5, H=64, nbe=1, nbd=1),         # I > 1 with I*K' = 64
    dict(B=3, K=2, R=1, I=1, D=128, Zd=7, H=128, nbe=1, nbd=1),       # widest fused-kernel input
    dict(B=3, K=2, R=1, I=1, D=64, Zd=64, H=256, nbe=1, nbd=1),       # widest output halves (2 x 64)
    dict(B=17, K=3, R=0, I=1, D=5, Zd=9, H=64, nbe=2, nbd=1),         # rows not a multiple of the 16-row tile
    dict(B=2049, K=1, R=1, I=1, D=4, Zd=3, H=64, nbe=1, nbd=1),       # just past the small-batch launch threshold
    dict(B=6, K=8, R=4, I=2, D=5, Zd=4, H=64, nbe=1, nbd=1, resampling_scheme="glr", weighting="dmis_within_groups",
         glr_cluster_size=4),                                          # grouped weights / resampling with I > 1
])
def test_lair_iteration_edge_shapes_match_oracle(case):
    run_case(seed=1000 + case["B"] * 7 + case["D"], **case)


def test_empty_batch_is_a_no_op():
    from oracle import cs_oracle as O
    from vae_conditional_sampling_amd.sampling.vae_imputation import importance_resampling_gibbs_iteration
    model = make_model(O.make_resfc_weights(5, 6, 1, 64, 1), O.make_resfc_weights(3, 10, 1, 64, 2))
    X = torch.zeros(0, 4, 5, device="cuda")
    M = torch.zeros(0, 1, 5, dtype=torch.bool, device="cuda")
    out = importance_resampling_gibbs_iteration(model, X, M, 1, num_prior_replenish_proposals=1)
    assert out[0].shape == (0, 4, 5) and out[7].shape == (1, 0, 5)


def test_unsupported_shapes_fail_loudly():
    from oracle import cs_oracle as O
    from vae_conditional_sampling_amd import _lib
    from vae_conditional_sampling_amd.sampling.vae_imputation import importance_resampling_gibbs_iteration
    model = make_model(O.make_resfc_weights(5, 6, 1, 64, 1), O.make_resfc_weights(3, 10, 1, 64, 2))
    X, M = seeded_inputs(2, 64, 5, 1)
    # I*K' = 65 proposals do not fit a wavefront: the path of the large proposal sets takes them
    out = importance_resampling_gibbs_iteration(model, X.cuda(), M.cuda(), 1, num_prior_replenish_proposals=1)
    assert out[7].shape == (1, 2, 65) and torch.isfinite(out[0]).all()
    with pytest.raises(_lib.CsError):      # clusters of 7 do not divide K' = 65
        importance_resampling_gibbs_iteration(model, X.cuda(), M.cuda(), 1, num_prior_replenish_proposals=1,
                                              resampling_scheme="glr", weighting="dmis_within_groups", glr_cluster_size=7)
    with pytest.raises(_lib.CsError):      # host tensors are refused: there is no CPU path
        importance_resampling_gibbs_iteration(model, X[:, :4], M, 1, num_prior_replenish_proposals=1)
    # C ABI: null arguments and mismatched networks return CS_EINVAL with a message
    L = _lib.lib()
    rc = L.cs_mlp_forward(None, None, 1, None, None)
    assert rc == -1 and b"null" in L.cs_last_error()
    enc = model.var_latent_network.packed()
    cfg = _lib.cs_lair_cfg(2, 4, 1, 1, 6, 3, 1, 1, 0, 0, 0, 0, 1.0, 0)   # D = 6 but the encoder takes 5
    Xg = torch.zeros(2, 4, 6, device="cuda")
    q = torch.zeros(2, 5, 2, 3, device="cuda")
    bufs = _lib.make_bufs(X=Xg, q=q)
    rc = L.cs_lair_encode(enc.handle, C.byref(cfg), C.byref(bufs), _lib.current_stream())
    assert rc == -1 and b"encoder shape" in L.cs_last_error()


def test_function_level_calls_return_fresh_tensors():
    """As in the reference, two calls of importance_resampling_gibbs_iteration return tensors that do not alias each
    other (the workspace views are an opt-in of inplace=True)."""
    import torch
    from oracle import cs_oracle as O
    from tests.helpers import make_model, seeded_inputs
    from vae_conditional_sampling_amd import _lib
    from vae_conditional_sampling_amd.sampling.vae_imputation import importance_resampling_gibbs_iteration
    model = make_model(O.make_resfc_weights(9, 8, 1, 64, 1), O.make_resfc_weights(4, 18, 1, 64, 2), device="cuda:0")
    X, M = seeded_inputs(16, 3, 9, 5)
    a = importance_resampling_gibbs_iteration(model, X.cuda(), M.cuda(), 1, num_prior_replenish_proposals=1, rng=_lib.make_rng(3, 0, 0))
    keep = [t.clone() for t in a if torch.is_tensor(t)]
    b = importance_resampling_gibbs_iteration(model, X.cuda(), M.cuda(), 1, num_prior_replenish_proposals=1, rng=_lib.make_rng(3, 1, 0))
    for t, k in zip([t for t in a if torch.is_tensor(t)], keep):
        assert torch.equal(t, k)                         # the first call's results survived the second call
    assert not torch.equal(a[7], b[7])
    v = importance_resampling_gibbs_iteration(model, X.cuda(), M.cuda(), 1, num_prior_replenish_proposals=1, rng=_lib.make_rng(3, 0, 0),
                                              inplace=True)
    w = importance_resampling_gibbs_iteration(model, X.cuda(), M.cuda(), 1, num_prior_replenish_proposals=1, rng=_lib.make_rng(3, 1, 0),
                                              inplace=True)
    assert v[7].data_ptr() == w[7].data_ptr()            # views of one workspace


@pytest.mark.parametrize("fn", ["rand_bin", "std_gauss"])
def test_initial_imputation_functions_of_the_reference(fn):
    """imputation_fn 'rand_bin' / 'std_gauss' (irwg/utils/basic_imputation.py:3-16): X*M + X_imp*(~M) with X_imp =
    (u > 0.5) or u for uniform draws u of X's shape; from injected draws exactly, from the counter RNG independent of
    batching."""
    from oracle import cs_oracle as O
    from oracle.cs_oracle_samplers import TapeSource
    from tests.test_gpu_samplers import BASE_KW
    from vae_conditional_sampling_amd.sampling.test_step_vae_sampling import TestVAEPseudoGibbs
    model = make_model(O.make_resfc_weights(5, 6, 1, 64, 1), O.make_resfc_weights(3, 10, 1, 64, 2))
    kw = dict(BASE_KW, imputation_fn=fn)
    s = TestVAEPseudoGibbs(num_iterations=1, clip_imputations=False, num_copies=3, **kw)
    s.set_model(model)
    s.on_test_start()
    X, M = seeded_inputs(6, 3, 5, 1)
    u = torch.rand(6, 3, 5, generator=torch.Generator().manual_seed(2))
    s.draw_source = TapeSource([("rand_like", u.numpy())])
    got = s.basic_imputation(X.cuda(), M.cuda()).cpu()
    want = X * M + ((u > 0.5).float() if fn == "rand_bin" else u) * (~M)
    assert torch.equal(got, want)
    s.draw_source = None
    s.rng_seed = 5
    s.example_offset = 0
    whole = s.basic_imputation(X.cuda(), M.cuda())
    s.example_offset = 4
    tail = s.basic_imputation(X[4:].cuda(), M[4:].cuda())
    assert torch.equal(whole[4:], tail) and torch.equal(whole.cpu()[M.expand_as(X)], X[M.expand_as(X)])
    if fn == "rand_bin":
        assert set(whole[~M.cuda().expand_as(whole)].unique().tolist()) <= {0.0, 1.0}
