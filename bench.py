#!/usr/bin/env python3
"""Throughput benchmark of the LAIR hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" is one LAIR iteration (encoder -> proposals -> decoder -> DM-MIS weights ->
resampling -> imputation) over one resident batch of synthetic data shaped like the UCI
MiniBooNE configuration (D=43, Zd=32, encoder 43->256 x5 blocks ->64, decoder 32->256 x2
blocks ->86, K=4 chains, R=1 prior-replenishing proposal, I=1, dmis / gr / multinomial).
Every rank (one process per GPU) owns its own shard of examples; there is no collective in
the loop.  The metric is conditional samples/s = (examples x K x steps) / seconds, summed
over ranks (weak scaling: the per-GPU batch is fixed).

Prints ONE JSON line on rank 0 with the roofline of the dominant kernel (the fused encoder
MLP on the matrix cores) and a CPU baseline (the torch-CPU oracle of this repository on a bounded
sample of the same workload).  The MLP kernel's default arithmetic builds each float32 product from
the bf16 terms of an exact 3-way split of both operands (float32-equivalent results); `--arith f32`
selects the exact-float32 matrix instructions, and at one GPU the line also carries that
arithmetic's throughput ("exact_f32").
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: D, Zd, H, encoder blocks, decoder blocks, K, R
    "miniboone": dict(D=43, Zd=32, H=256, nbe=5, nbd=2, K=4, R=1),
    "hepmass": dict(D=21, Zd=16, H=256, nbe=2, nbd=2, K=16, R=1),
    "gas": dict(D=8, Zd=16, H=256, nbe=2, nbd=2, K=16, R=1),
}
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X exact-f32 matrix peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 matrix peak
# The default arithmetic of the fused MLP kernel forms every float32 product from 6 bf16 term products
# (exact 3-way split of both operands, float32 accumulation): its ceiling in algorithmic (float32) FLOP/s
SPLIT_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0


def resfc_flops(i, H, nb, o):
    return 2 * (i * H + 2 * nb * H * H + H * o)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="miniboone", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=65536, help="examples per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--arith", default=None, choices=["bf16x3", "f32"],
                    help="arithmetic of the fused MLP kernel (default: the library's, bf16x3 = exact 3-way bf16 split of "
                         "the float32 operands, float32 accumulation, float32-equivalent results)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from vae_conditional_sampling_amd import _lib
    from vae_conditional_sampling_amd.models import VAE, ResidualFCNetwork
    from vae_conditional_sampling_amd.sampling import engine

    if args.arith:
        _lib.set_mlp_arith(args.arith)
    arith = _lib.get_mlp_arith()
    w = WORKLOADS[args.workload]
    D, Zd, H, K, R = w["D"], w["Zd"], w["H"], w["K"], w["R"]
    B = args.batch
    seed = 20220118
    torch.manual_seed(seed)  # random-init weights of the named architecture (torch's nn.Linear default)
    enc = ResidualFCNetwork(D, 2 * Zd, w["nbe"], H)
    dec = ResidualFCNetwork(Zd, 2 * D, w["nbd"], H)
    def cpu_copy(net):  # float32 CPU copies for the CPU baseline leg
        d = net.weight_dict()
        c = lambda t: t.detach().clone()
        return {"W0": c(d["W0"]), "b0": c(d["b0"]), "blocks": [tuple(c(t) for t in b) for b in d["blocks"]],
                "Wf": c(d["Wf"]), "bf": c(d["bf"])}

    enc_w, dec_w = cpu_copy(enc), cpu_copy(dec)
    model = VAE(dec, "normal_with_eps", enc, "normal_with_eps").to(dev)

    # synthetic standardised data, 50 % uniform missingness, this rank's shard
    g = torch.Generator(device="cpu").manual_seed(seed + 17 * rank)
    X_true = torch.randn(B, D, generator=g)
    M = torch.rand(B, D, generator=g) > 0.5
    M[:, 0] = True
    X = X_true.unsqueeze(1).repeat(1, K, 1)
    X = (X * M.unsqueeze(1) + torch.randn(B, K, D, generator=g) * (~M.unsqueeze(1))).to(dev).contiguous()
    M8 = M.to(torch.uint8).to(dev).contiguous()

    ws = engine.lair_workspace(model, B, K, R, 1, dev)
    cfg = engine.lair_cfg(model, ws, weighting="dmis", scheme="gr", method="multinomial", cluster_size=0,
                          weight_alpha=None)
    bufs = _lib.make_bufs(X=X, M=M8, q=ws.q, Z=ws.Z, gen=ws.gen, gen_ll=ws.gen_ll, log_w=ws.log_w,
                          norm_log_w=ws.norm_log_w, idx=ws.idx, ess=ws.stats[0], tess=ws.stats[1],
                          perplexity=ws.stats[2])
    L = _lib.lib()
    h_enc = model.var_latent_network.packed().handle
    h_dec = model.generator_network.packed().handle
    stream = _lib.current_stream()

    def step(t, events=None):
        rng = _lib.make_rng(seed, t, rank * B)
        if events is not None:
            events[0].record()
        _lib.check(L.cs_lair_encode(h_enc, C.byref(cfg), C.byref(bufs), stream))
        if events is not None:
            events[1].record()
        _lib.check(L.cs_lair_decode(h_dec, C.byref(cfg), C.byref(bufs), C.byref(rng), stream))
        if events is not None:
            events[2].record()
        _lib.check(L.cs_lair_resample_impute(C.byref(cfg), C.byref(bufs), C.byref(rng), stream))
        if events is not None:
            events[3].record()

    for t in range(args.warmup):
        step(t)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for t in range(args.steps):
        step(args.warmup + t, evs[t])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # the only collective of the path: final imputations and log-weights are gathered once
        gathered = torch.empty(world, *X.shape, device=dev)
        dist.all_gather_into_tensor(gathered, X)
        glw = torch.empty(world, *ws.log_w.shape, device=dev)
        dist.all_gather_into_tensor(glw, ws.log_w)
        torch.cuda.synchronize()
    assert torch.isfinite(X).all(), "non-finite imputations"

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # the same K steps on the exact-float32 matrix instructions, for comparison (outside the timed region)
    exact = None
    if arith != "f32" and world == 1:
        _lib.set_mlp_arith("f32")
        for t in range(args.warmup):
            step(t)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for t in range(args.steps):
            step(args.warmup + t)
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        _lib.set_mlp_arith(arith)
        exact = {"value": B * K * args.steps / el, "ms_per_step": el / args.steps * 1e3, "unit": "samples/s"}

    ms = lambda a, b: sum(e[a].elapsed_time(e[b]) for e in evs) / len(evs)
    enc_ms, dec_ms, res_ms = ms(0, 1), ms(1, 2), ms(2, 3)
    f_enc = resfc_flops(D, H, w["nbe"], 2 * Zd)
    f_dec = resfc_flops(Zd, H, w["nbd"], 2 * D)
    enc_flops = f_enc * B * K            # algorithmic FLOPs of one encoder launch
    dec_flops = f_dec * B * (K + R)
    enc_tf = enc_flops / (enc_ms * 1e-3) / 1e12
    dec_tf = dec_flops / (dec_ms * 1e-3) / 1e12
    value = world * B * K * args.steps / elapsed
    split = arith == "bf16x3"
    peak = SPLIT_PEAK_TFLOPS if split else F32_MFMA_PEAK_TFLOPS
    Kp = K + R
    res_bytes = (Kp * 2 * Zd * 4 + Kp * Zd * 4 + Kp * 4 + K * (2 * D * 4 + D * 4) + D) + (K * D * 4 + 2 * Kp * 4 + Kp * 8 + 12)

    # HBM bytes per encoder launch from the PMC counters of a profiled run of this command
    # (tools/pmc_summary.py writes the file; null when it has not been collected)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_summary.json")) as f:
            pmc = json.load(f)
        if pmc.get("workload") == args.workload and pmc.get("examples_per_gpu") == B:
            traffic = pmc["encoder_launch"]["hbm_bytes"]
    except (OSError, KeyError, ValueError):
        pass

    out = {
        "metric": "conditional samples/sec (N x K chains) for LAIR",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 as 3 bf16 terms per operand, f32 accumulation" if split else "f32", "data": "synthetic",
        "config": {"workload": f"uci_{args.workload} MLP-VAE LAIR (dmis/gr/multinomial), D={D} Zd={Zd} H={H} "
                               f"enc_blocks={w['nbe']} dec_blocks={w['nbd']} K={K} R={R} I=1",
                   "examples_per_gpu": B, "chains_per_gpu": B * K, "sharding": f"examples x{world}, no collective in loop"},
        "roofline": {"bound": "mfma", "kernel": ("cs_mlp_split_kernel" if split else "cs_mlp_kernel") + " (encoder launch)",
                     "achieved": enc_tf, "peak": peak, "unit": "TFLOP/s", "frac": enc_tf / peak,
                     "traffic": traffic, "flops_per_launch": enc_flops, "avg_launch_ms": enc_ms,
                     "note": ("algorithmic float32 FLOPs; each is executed as 6 bf16 matrix FLOPs, so peak = dense bf16 "
                              "peak 2500 / 6 and the executed bf16 rate is 6 x achieved") if split else
                             "algorithmic float32 FLOPs on the exact-f32 matrix instructions"},
        "kernels": {"encoder_ms": enc_ms, "decoder_ms": dec_ms, "resample_impute_ms": res_ms,
                    # HBM-bound stage: algorithmic bytes per example (q, Z, gen_ll, the K chosen generator rows,
                    # X and M read; X, log-weights, normalised weights, indices, statistics written)
                    "resample_impute_GBps": B * res_bytes / (res_ms * 1e-3) / 1e9, "resample_impute_bytes_per_example": res_bytes,
                    "decoder_tflops": dec_tf, "decoder_frac": dec_tf / peak,
                    "step_flops": enc_flops + dec_flops,
                    "step_mfma_frac": (enc_flops + dec_flops) / ((enc_ms + dec_ms + res_ms) * 1e-3) / 1e12 / peak,
                    "mlp_arith": arith},
    }
    if exact is not None:
        out["exact_f32"] = exact

    if not args.no_cpu_baseline and world == 1:
        # Accuracy of the arithmetic in force, measured here and now (left out, like the CPU baseline, of the
        # profiled --no-cpu-baseline runs, whose kernel statistics then hold the timed launches only): the encoder network on 4096 rows of the
        # batch against a float64 evaluation of the same weights (torch, CPU), beside the exact-f32 kernel's
        # error on the same rows.  max |y - y64| / max |y64|.
        def forward64(wd, x):
            F = torch.nn.functional
            h = F.relu(F.linear(x, wd["W0"].double(), wd["b0"].double()))
            for W1, b1, W2, b2 in wd["blocks"]:
                h = F.relu(h + F.linear(F.relu(F.linear(h, W1.double(), b1.double())), W2.double(), b2.double()))
            return F.linear(h, wd["Wf"].double(), wd["bf"].double())

        xs = X[:1024].reshape(-1, D).contiguous()
        y64 = forward64(enc_w, xs.cpu().double())
        acc = {}
        for name in ("bf16x3", "f32"):
            _lib.set_mlp_arith(name)
            y = model.var_latent_network(xs).cpu().double()
            acc[name] = float((y - y64).abs().max() / y64.abs().max())
        _lib.set_mlp_arith(arith)
        out["accuracy"] = {"what": "encoder outputs, 4096 rows, max |y - y_float64| / max |y_float64|",
                           "bf16x3": acc["bf16x3"], "exact_f32": acc["f32"]}

    if not args.no_cpu_baseline and world == 1:
        from oracle import cs_oracle as O  # the checker, used here only as the timed CPU baseline
        # CPU baseline: the torch-CPU oracle of this repository ("port") on a bounded sample of
        # the same workload: the reference configuration's batch of 512 examples, timed for ~10 s.
        nthreads = min(torch.get_num_threads(), 16)
        torch.set_num_threads(nthreads)
        Bc = 512
        Xc, Mc = X[:Bc].cpu(), M[:Bc].unsqueeze(1)
        gc = torch.Generator().manual_seed(1)
        draws = {"eps_z": torch.randn(1, Bc, K + R, Zd, generator=gc),
                 "u_res": torch.rand(Bc, K + R, generator=gc, dtype=torch.float64),
                 "eps_x": torch.randn(Bc, K, D, generator=gc)}
        with torch.no_grad():
            O.lair_iteration(enc_w, dec_w, Xc, Mc, draws, num_prior_replenish_proposals=R)
            n, tc0 = 0, time.perf_counter()
            while time.perf_counter() - tc0 < 10.0:
                Xc = O.lair_iteration(enc_w, dec_w, Xc, Mc, draws, num_prior_replenish_proposals=R)["X_new"]
                n += 1
            tc = time.perf_counter() - tc0
        out["cpu_baseline"] = {"value": Bc * K * n / tc, "unit": "samples/s", "cores": nthreads, "kind": "port",
                               "sample": f"{n} LAIR iterations of one batch of {Bc} examples x K={K} "
                                         f"(torch-CPU oracle, float32, {nthreads} threads, ~10 s)"}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
