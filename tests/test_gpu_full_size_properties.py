"""Size-independent properties at the benchmark's full size (MiniBooNE shape, 65 536 examples per
GPU, counter RNG): what must hold for any input without an oracle run of that size.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

B, K, R, D, Zd, H = 65536, 4, 1, 43, 32, 256


@pytest.fixture(scope="module")
def model():
    from vae_conditional_sampling_amd.models import ResidualFCNetwork, VAE
    torch.manual_seed(5)
    return VAE(ResidualFCNetwork(Zd, 2 * D, 2, H), "normal_with_eps", ResidualFCNetwork(D, 2 * Zd, 5, H), "normal_with_eps").cuda()


def inputs(seed):
    g = torch.Generator().manual_seed(seed)
    Xt = torch.randn(B, D, generator=g)
    M = torch.rand(B, D, generator=g) > 0.5
    M[:, 0] = True
    X = Xt.unsqueeze(1).repeat(1, K, 1) * M.unsqueeze(1) + torch.randn(B, K, D, generator=g) * (~M.unsqueeze(1))
    return X.cuda().contiguous(), M.unsqueeze(1).cuda().contiguous()


def test_lair_full_size_invariants_and_sharding(model):
    from vae_conditional_sampling_amd import _lib
    from vae_conditional_sampling_amd.sampling.vae_imputation import importance_resampling_gibbs_iteration as step
    X, M = inputs(1)
    out = step(model, X, M, 1, num_prior_replenish_proposals=R, rng=_lib.make_rng(9, 2, 0))
    Xn, ess, perp, nlw, Z, lw = out[0].clone(), out[1].clone(), out[3].clone(), out[5].clone(), out[6].clone(), out[7].clone()
    ws = model._cs_workspaces[("lair", B, K, R, 1, str(X.device))]
    idx = ws.idx.clone()
    Mb = M.expand_as(X)
    assert torch.equal(Xn[Mb], X[Mb]), "observed entries must be untouched bit for bit"
    assert torch.isfinite(Xn).all() and torch.isfinite(lw).all()
    assert (idx >= 0).all() and (idx < K + R).all()
    assert torch.allclose(torch.logsumexp(nlw, -1), torch.zeros(B, device=X.device), atol=1e-5)
    assert (ess > 0).all() and (ess <= K + R + 1e-3).all() and (perp > 0).all() and (perp <= K + R + 1e-3).all()
    # the imputed entries are the resampled proposal's generator sample: every chain's missing part changed
    assert (Xn[~Mb] != X[~Mb]).float().mean() > 0.999
    # two half batches keyed by their global offsets reproduce the whole batch bit for bit
    h = B // 2
    a = step(model, X[:h].contiguous(), M[:h].contiguous(), 1, num_prior_replenish_proposals=R, rng=_lib.make_rng(9, 2, 0))[0].clone()
    b = step(model, X[h:].contiguous(), M[h:].contiguous(), 1, num_prior_replenish_proposals=R, rng=_lib.make_rng(9, 2, h))[0].clone()
    assert torch.equal(torch.cat([a, b]), Xn)
    # a different iteration draws different numbers
    other = step(model, X, M, 1, num_prior_replenish_proposals=R, rng=_lib.make_rng(9, 3, 0))[0]
    assert not torch.equal(other, Xn)


def test_acmwg_full_size_invariants(model):
    from vae_conditional_sampling_amd.sampling.test_step_vae_sampling import \
        TestVAEAdaptiveCollapsedMetropolisWithinGibbs as Sampler
    from tests.test_gpu_samplers import BASE_KW
    Bn = 16384
    X, M = inputs(2)
    X, M = X[:Bn, :1].repeat(1, 5, 1).contiguous(), M[:Bn].contiguous()
    T = 6
    s = Sampler(num_pseudo_warmup_iterations=0, num_mwg_iterations=T, prior_mixture_probability=0.05, warmup_method="none",
                num_copies=5, **BASE_KW)
    s.set_model(model)
    s.on_test_start()
    s.rng_seed = 3
    Mb = M.expand_as(X)
    prev = X.clone()
    for t, o in enumerate(s.sampler(X.clone(), M)):
        Xt, acc = o["X"], o["_accepted"]
        assert set(torch.unique(acc).tolist()) <= {0, 1}
        assert torch.equal(Xt[Mb], prev[Mb]), "observed entries must be untouched"
        assert torch.isfinite(Xt).all()
        # the history ring holds this iteration's imputations at its tail
        assert torch.equal(s.history.history_queue[s.history.tail_idx], Xt)
        av = s.history.available_timesteps
        assert (av >= 0).all() and (av <= len(s.history)).all()
        # accepted chains may draw from everything that was in the queue when they were accepted
        # (update_available_timesteps runs before the enqueue, irwg/sampling/utils.py:94-113)
        assert (av[acc.bool()] == len(s.history) - 1).all()
        assert 0.0 <= float(o["acceptance"]) <= 1.0
        prev = Xt.clone()
    assert t == T - 1


@pytest.mark.parametrize("arith", ["bf16x3", "f16x2", "f32"])
def test_repeated_full_size_launches_are_bit_identical(model, arith):
    """The same encoder / decoder launches on the same inputs, over and over: any difference is a race or a
    missed hardware hazard inside a kernel.  (Regression: the epilogue of the fused MLP kernel once read the
    accumulators of the final matrix instruction too early, which showed as run-to-run differences in the last
    bits of two output columns, a few launches in a hundred.)"""
    import ctypes as C
    from vae_conditional_sampling_amd import _lib
    from vae_conditional_sampling_amd.sampling import engine
    prev = _lib.set_mlp_arith(arith)
    try:
        X, M = inputs(3)
        M8 = M[:, 0].to(torch.uint8).contiguous()
        ws = engine.lair_workspace(model, B, K, R, 1, X.device)
        cfg = engine.lair_cfg(model, ws, weighting="dmis", scheme="gr", method="multinomial", cluster_size=0, weight_alpha=None)
        bufs = _lib.make_bufs(X=X, M=M8, q=ws.q, Z=ws.Z, gen=ws.gen, gen_ll=ws.gen_ll)
        L, st = _lib.lib(), _lib.current_stream()
        he, hd = model.var_latent_network.packed().handle, model.generator_network.packed().handle
        rng = _lib.make_rng(4, 1, 0)

        def launch():
            ws.q.fill_(-7.0)
            ws.gen.fill_(-7.0)
            _lib.check(L.cs_lair_encode(he, C.byref(cfg), C.byref(bufs), st))
            _lib.check(L.cs_lair_decode(hd, C.byref(cfg), C.byref(bufs), C.byref(rng), st))
            torch.cuda.synchronize()
            return ws.q.clone(), ws.Z.clone(), ws.gen.clone(), ws.gen_ll.clone()

        ref = launch()
        for rep in range(25):
            out = launch()
            for name, a, b in zip(("q", "Z", "gen", "gen_ll"), out, ref):
                assert torch.equal(a, b), f"launch {rep}: {name} differs from the first launch in {int((a != b).sum())} entries"
    finally:
        _lib.set_mlp_arith(prev)
