#!/usr/bin/env python3
"""Repeats one full-size LAIR iteration (MiniBooNE shape, counter RNG) on the same inputs and compares every
output bit for bit with the first run, whole batch and two half batches: a race in a kernel shows up as a
mismatch.  python tools/determinism_check.py [repeats] [arith]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    from vae_conditional_sampling_amd import _lib
    from vae_conditional_sampling_amd.models import VAE, ResidualFCNetwork
    from vae_conditional_sampling_amd.sampling.vae_imputation import importance_resampling_gibbs_iteration as step
    if len(sys.argv) > 2:
        _lib.set_mlp_arith(sys.argv[2])
    B, K, R, D, Zd, H = 65536, 4, 1, 43, 32, 256
    torch.manual_seed(5)
    model = VAE(ResidualFCNetwork(Zd, 2 * D, 2, H), "normal_with_eps", ResidualFCNetwork(D, 2 * Zd, 5, H), "normal_with_eps").cuda()
    g = torch.Generator().manual_seed(1)
    Xt = torch.randn(B, D, generator=g)
    M = torch.rand(B, D, generator=g) > 0.5
    M[:, 0] = True
    X = (Xt.unsqueeze(1).repeat(1, K, 1) * M.unsqueeze(1) + torch.randn(B, K, D, generator=g) * (~M.unsqueeze(1))).cuda().contiguous()
    M = M.unsqueeze(1).cuda().contiguous()
    h = B // 2

    def run(x, m, off):
        out = step(model, x, m, 1, num_prior_replenish_proposals=R, rng=_lib.make_rng(9, 2, off))
        ws = model._cs_workspaces[("lair", x.shape[0], K, R, 1, str(x.device))]
        return {"X": out[0].clone(), "log_w": out[7].clone(), "q": ws.q.clone(), "Z": ws.Z.clone(), "gen": ws.gen.clone(),
                "gen_ll": ws.gen_ll.clone(), "idx": ws.idx.clone()}

    ref = run(X, M, 0)
    bad = 0
    for r in range(reps):
        full = run(X, M, 0)
        a, b = run(X[:h].contiguous(), M[:h].contiguous(), 0), run(X[h:].contiguous(), M[h:].contiguous(), h)
        for k in ref:
            if not torch.equal(full[k], ref[k]):
                n = int((full[k] != ref[k]).sum())
                print(f"rep {r}: whole batch differs from the first run in {k}: {n} entries")
                bad += 1
            halves = torch.cat([a[k].reshape(h, -1) if k not in ("Z", "gen", "gen_ll", "log_w") else a[k].reshape(-1, h, *a[k].shape[2:]).transpose(0, 1).reshape(h, -1),
                                b[k].reshape(h, -1) if k not in ("Z", "gen", "gen_ll", "log_w") else b[k].reshape(-1, h, *b[k].shape[2:]).transpose(0, 1).reshape(h, -1)])
            whole = ref[k].reshape(B, -1) if k not in ("Z", "gen", "gen_ll", "log_w") else ref[k].reshape(-1, B, *ref[k].shape[2:]).transpose(0, 1).reshape(B, -1)
            if halves.shape == whole.shape and not torch.equal(halves, whole):
                rows = (halves != whole).any(1).nonzero().flatten()
                print(f"rep {r}: halves differ from the whole batch in {k}: {len(rows)} examples, first {rows[:6].tolist()}")
                bad += 1
    print(f"{_lib.get_mlp_arith()}: {reps} repetitions, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
