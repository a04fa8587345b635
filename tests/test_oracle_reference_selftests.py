"""CPU: the reference's own executable self-tests for the path, re-expressed on the oracle.

  * restricted-availability history, known-answer test    irwg/sampling/utils.py:277-310
  * DM-MIS weights equal an explicit nested-loop evaluation irwg/utils/weights.py:88-134
  * systematic resampling histogram                         irwg/sampling/vae_imputation.py:1059-1079
"""
import numpy as np
import torch

from oracle import c_binding
from oracle import cs_oracle as O


def _history_script(enqueue, update, avail):
    """The scripted sequence of utils.py:283-308; `avail()` returns the counters as a list."""
    enqueue(); enqueue()
    update([[1, 0, 1], [0, 1, 1]])
    assert avail() == [[2, 0, 2], [0, 2, 2]]
    enqueue(); enqueue(); enqueue()
    update([[0, 0, 1], [0, 1, 0]])
    assert avail() == [[2, 0, 5], [0, 5, 2]]
    enqueue()
    assert avail() == [[2, 0, 5], [0, 5, 2]]
    enqueue(); enqueue()
    update([[1, 0, 1], [0, 0, 1]])
    assert avail() == [[8, 0, 8], [0, 5, 8]]
    enqueue(); enqueue()
    assert avail() == [[6, 0, 6], [0, 3, 6]]


def test_history_known_answer_python_oracle():
    h = O.RestrictedHistory(8, (2, 3, 4))
    c = [0]

    def enq():
        c[0] += 1
        h.enqueue(torch.ones(2, 3, 4) * c[0])
    _history_script(enq, lambda m: h.update_available(torch.tensor(m, dtype=torch.bool)), lambda: h.avail.tolist())
    # the ring holds the last 8 values at the expected slots
    assert h.full and h.tail == 1
    assert float(h.queue[1, 0, 0, 0]) == 10.0 and float(h.queue[2, 0, 0, 0]) == 3.0


def test_history_known_answer_c_oracle():
    L = c_binding.load()
    import ctypes as C
    h = c_binding.cs_hist()
    avail = np.zeros((2, 3), dtype=np.int64)
    L.history_init(C.byref(h), 8, 2, 3, avail.ctypes.data)

    def upd(m):
        m8 = np.asarray(m, dtype=np.uint8)
        L.history_update_available(C.byref(h), avail.ctypes.data, m8.ctypes.data)
    _history_script(lambda: L.history_enqueue(C.byref(h), avail.ctypes.data), upd, lambda: avail.tolist())
    assert h.full == 1 and h.tail == 1
    assert L.history_slot(C.byref(h), 0) == 2  # oldest entry sits right after the tail


def test_dmis_weights_equal_nested_loops():
    B, I, K, D, NZ, DZ = 7, 6, 5, 4, 3, 2
    g = torch.Generator().manual_seed(0)
    X = torch.randn(B, K, D, generator=g)
    Z = torch.randn(NZ, I, B, K, DZ, generator=g)
    gen_mean = torch.randn(NZ, I, B, K, D, generator=g)
    gen_scale = (torch.randn(NZ, I, B, K, D, generator=g) * 3) ** 2
    var_mean = torch.randn(B, K, DZ, generator=g)
    var_scale = (torch.randn(B, K, DZ, generator=g) * 2) ** 2
    for z in range(NZ):
        ql = O.dmis_mixture_logprob(Z[z], var_mean, var_scale)
        gl = O.normal_log_prob(X, gen_mean[z], gen_scale[z]).sum(-1)
        pl = O.std_normal_log_prob(Z[z]).sum(-1)
        lw = gl + pl - ql
        for i in range(I):
            for b in range(B):
                for k in range(K):
                    comps = torch.stack([O.normal_log_prob(Z[z, i, b, k], var_mean[b, k2], var_scale[b, k2]).sum()
                                         for k2 in range(K)])
                    ref = (O.normal_log_prob(X[b, k], gen_mean[z, i, b, k], gen_scale[z, i, b, k]).sum()
                           + O.std_normal_log_prob(Z[z, i, b, k]).sum()
                           - (torch.logsumexp(comps, 0) - torch.log(torch.tensor(K))))
                    assert abs(float(lw[i, b, k]) - float(ref)) <= 1e-4 * (1 + abs(float(ref)))


def test_systematic_resampling_histogram():
    g = torch.Generator().manual_seed(10112)
    w = torch.rand(20, generator=g)
    probs = w / w.sum()
    lw = torch.log(w)
    n = 20000
    L = c_binding.load()
    for m in (20, 5, 1):
        U = torch.rand(n, 1, generator=g)
        idx = O.systematic_indices(lw.expand(n, 20), U, m)
        assert idx.shape == (n, m) and int(idx.max()) < 20
        counts = torch.bincount(idx.reshape(-1), minlength=20).float() / (n * m)
        assert np.allclose(counts.numpy(), probs.numpy(), atol=0.01)
        # the C restatement agrees draw for draw on the same probabilities
        p32 = (torch.exp(lw - torch.logsumexp(lw, 0)))
        p32 = (p32 / p32.sum()).numpy().astype(np.float32)
        out = np.empty(m, dtype=np.int64)
        for r in range(0, 200):
            L.systematic_index(p32.ctypes.data, 20, float(U[r, 0]), m, out.ctypes.data)
            assert np.array_equal(out, idx[r].numpy())


def test_multinomial_restatement_matches_torch_multinomial():
    L = c_binding.load()
    for S_, n, rows, seed in [(5, 5, 7, 0), (25, 5, 3, 1), (17, 17, 11, 2), (3000, 12, 4, 3)]:
        g = torch.Generator().manual_seed(seed)
        p = torch.softmax(torch.randn(rows, S_, generator=g) * 3, -1)
        g1 = torch.Generator().manual_seed(100 + seed); g2 = torch.Generator().manual_seed(100 + seed)
        ref = torch.multinomial(p, n, True, generator=g1)
        u = torch.rand(rows, n, dtype=torch.float64, generator=g2)
        assert torch.equal(O.multinomial_indices(p, u.numpy()), ref)
        assert np.array_equal(c_binding.multinomial_left(L, p.numpy(), u.numpy()), ref.numpy())


def test_independent_resampling_indexing_known_answers():
    """The reference's two checks of select_gen_params_for_ir (vae_imputation.py:1004-1057): with
    I == K output (b, k) is X[k, b, idx[b, k]]; with I == 2K it is entry idx[b, k] of the chain-k row
    of '(i i2) b k p -> i b (i2 k) p'.  The oracle's flattened-source rule (slot k takes proposal
    k*I + idx of the '(i k)' axis) must give the same selections for the reference's index tables."""
    B, P, K = 6, 4, 5
    tables = {5: [[0, 1, 3, 2, 3], [4, 1, 1, 1, 1], [0, 0, 1, 2, 0], [0, 1, 3, 1, 1], [2, 2, 4, 2, 3], [1, 0, 3, 2, 0]],
              10: [[0, 8, 5, 2, 9], [4, 1, 1, 1, 1], [0, 0, 1, 2, 5], [0, 1, 3, 7, 1], [2, 8, 4, 2, 3], [1, 6, 6, 2, 0]]}
    for I, table in tables.items():
        idx = torch.tensor(table)
        X = torch.randn(I, B, K, P, generator=torch.Generator().manual_seed(I))
        glob = torch.arange(K)[None, :] * I + idx                    # the rule of cs_oracle.ir_resample
        picked = O.gather_gr(X, glob)                                # [B,K,P]
        if I == K:
            expect = torch.stack([torch.stack([X[k, b, idx[b, k]] for k in range(K)]) for b in range(B)])
        else:
            i2 = I // K
            Xr = X.reshape(K, i2, B, K, P).permute(0, 2, 1, 3, 4).reshape(K, B, i2 * K, P)
            expect = torch.stack([torch.stack([Xr[k, b, idx[b, k]] for k in range(K)]) for b in range(B)])
        assert torch.equal(picked, expect)


def test_single_sample_multinomial_restatement_matches_torch():
    """torch.multinomial(p, 1, replacement=True) on CPU = argmax(p / Exp(1)) with the variates drawn
    by exponential_ from the same generator (lr / ir resampling draw one index per chain)."""
    p = torch.softmax(torch.randn(40, 9, generator=torch.Generator().manual_seed(3)), -1)
    g = torch.Generator().manual_seed(4)
    real = torch.multinomial(p, 1, True, generator=g)[:, 0]
    g = torch.Generator().manual_seed(4)
    e = torch.empty_like(p).exponential_(1, generator=g)
    assert torch.equal(real, O.single_indices(p, e))
