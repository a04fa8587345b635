#!/usr/bin/env python3
"""Throughput benchmark of the LAIR hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" is one LAIR iteration (encoder -> proposals -> decoder -> DM-MIS weights ->
resampling -> imputation) over one resident batch of synthetic data shaped like the UCI
MiniBooNE configuration (D=43, Zd=32, encoder 43->256 x5 blocks ->64, decoder 32->256 x2
blocks ->86, K=4 chains, R=1 prior-replenishing proposal, I=1, dmis / gr / multinomial).
Every rank (one process per GPU) owns its own shard of examples; there is no collective in
the loop.  The metric is conditional samples/s = (examples x K x steps) / seconds, summed
over ranks (weak scaling: the per-GPU batch is fixed).  With --gpus N and no launcher
(WORLD_SIZE unset) the script starts its N rank processes itself.

Prints ONE JSON line on rank 0 with the roofline of the dominant kernel (the fused encoder
MLP on the matrix cores) and a CPU baseline (the torch-CPU oracle of this repository on a bounded
sample of the same workload).  The MLP kernel's default arithmetic builds each float32 product from
the bf16 terms of an exact 3-way split of both operands (float32-equivalent results); `--arith f32`
selects the exact-float32 matrix instructions, and at one GPU the line also carries that
arithmetic's throughput and roofline ("exact_f32").

At one GPU the line also carries `extra` legs, each timed the same way on its own resident batch
(`--no-extras` leaves them out): LAIR at the HEPMASS and GAS shapes (K=16), LAIR MiniBooNE through
the sampler class at the large batch and at the reference's own batch of 512 examples, AC-MWG at
the MiniBooNE configuration (K=5) and on the Omniglot ConvResNet VAE, LAIR on the MNIST-GMM
ConvResNet VAE.
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
    # more than 64 proposals per example (the multi-wave weights / resampling path): the MiniBooNE networks with
    # 96 chains + 4 prior proposals, as configs/mnist_gmm/samples/lair_k100_* have them
    "miniboone_k100": dict(D=43, Zd=32, H=256, nbe=5, nbd=2, K=96, R=4),
}
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X exact-f32 matrix peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 matrix peak
# The default arithmetic of the fused MLP kernel forms every float32 product from 6 bf16 term products
# (exact 3-way split of both operands, float32 accumulation): its ceiling in algorithmic (float32) FLOP/s
SPLIT_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0
# algorithmic FLOPs per image of the ConvResNet encoders / decoders (2 k^2 cin cout hout wout per convolution,
# 2 in out per linear layer)
CONV_FLOPS = {"omniglot": (911693824, 1230915584), "mnist_gmm": (60988672, 113640448)}
# CSVAE_BENCH_REHEARSAL=1: a rehearsal of the multi-rank control flow on a box with ONE GPU -- every rank computes on
# GPU 0 and the collectives run over gloo on host copies.  Its JSON line says so ("rehearsal") and is no measurement.
REHEARSAL = os.environ.get("CSVAE_BENCH_REHEARSAL") == "1"
SEED = 20220118
SETTLE_S = 0.4   # seconds of untimed load before the warm-up steps (clock settle)
BASE_KW = dict(store_last_n_iterations=-1, store_every_n_iterations=1, imputation_metric_quantiles=[],
               compute_f1_metric=False, compute_inv_f1_metric=False, compute_bin_accuracy_metric=False,
               compute_rmse_metric=False, compute_ssim_metric=False, compute_latent_wass_metric=False,
               imputation_fn="vae_samples")


# examples per GPU of the HEPMASS / GAS legs: the whole-test-set configurations (K = 16) shard ~1e5-row test sets,
# 131 072 synthetic examples per GPU under weak scaling
WHOLE_SET_BATCH = 131072


def resfc_flops(i, H, nb, o):
    return 2 * (i * H + 2 * nb * H * H + H * o)


# the fp16 split forms a float32 product from 3 fp16 term products (dense fp16 matrix peak = the bf16 one)
F16X2_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 3.0
DTYPES = {"bf16x3": "f32 as 3 bf16 terms per operand, f32 accumulation", "f16x2": "f32 as 2 fp16 terms per operand, f32 accumulation",
          "f32": "f32"}


def peak_of(arith):
    return {"bf16x3": SPLIT_PEAK_TFLOPS, "f16x2": F16X2_PEAK_TFLOPS, "bf16": BF16_MFMA_PEAK_TFLOPS}.get(arith, F32_MFMA_PEAK_TFLOPS)


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N rank processes here, one per GPU, before this
    process has touched a GPU (it never does), and exit with their status.  Rank 0 prints the JSON line."""
    import socket
    import subprocess
    have = torch.cuda.device_count()   # counts devices without initialising the GPU runtime
    if REHEARSAL and have >= 1:
        have = n                       # rehearsal: every rank uses GPU 0 (see REHEARSAL)
    if have < n:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible; not reporting a smaller run in its place",
              file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rcs = [p.wait() for p in procs]
    if any(rcs):
        print(f"bench.py: rank exit codes {rcs}", file=sys.stderr)
        return 1
    return 0


def dry_run(rank, world, local_rank, dev, dist):
    """`--dry-run`: every rank reports the device it is bound to and takes part in one all-reduce and one
    all-gather; rank 0 prints one JSON line {"dry_run": true, "ranks": [{rank, local_rank, device, name}, ...]}.
    A process group that cannot be set up has already ended the rank with a non-zero status (init_process_group
    raises), which launch_ranks / the launcher reports."""
    import json as _json
    cur = torch.cuda.current_device()
    me = {"rank": rank, "local_rank": local_rank, "device": f"cuda:{cur}", "name": torch.cuda.get_device_name(cur)}
    if (not REHEARSAL) and cur != local_rank:
        print(f"bench.py: rank {rank} is bound to cuda:{cur}, expected cuda:{local_rank}", file=sys.stderr)
        return 3
    everyone = [me]
    if dist is not None:
        cdev = torch.device("cpu") if REHEARSAL else dev
        t = torch.full((4,), float(rank + 1), device=cdev)
        dist.all_reduce(t)
        want = world * (world + 1) / 2
        g = torch.empty(world * 4, device=cdev)
        dist.all_gather_into_tensor(g, torch.full((4,), float(rank), device=cdev))
        if float(t[0]) != want or g.reshape(world, 4)[:, 0].tolist() != [float(r) for r in range(world)]:
            print(f"bench.py: rank {rank}: collective results are wrong", file=sys.stderr)
            return 4
        everyone = [None] * world
        dist.all_gather_object(everyone, me)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(_json.dumps({"dry_run": True, "n_gpus": world, "backend": "gloo (rehearsal)" if REHEARSAL else ("nccl" if world > 1 else None),
                           "ranks": everyone}))
    return 0


def synthetic_batch(B, K, D, seed, dev, binary=False, quadrants=False):
    """Standardised synthetic data with 50 % uniform missingness (UCI shapes), or Bernoulli(0.2) pixels with two
    of the four 14x14 quadrants missing (Omniglot shape).  Returns X[B,K,D] with missing entries filled by
    noise, and the mask M[B,D] (True = observed)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    if binary:
        X_true = (torch.rand(B, D, generator=g) < 0.2).float()
    else:
        X_true = torch.randn(B, D, generator=g)
    if quadrants:
        M = torch.ones(B, 28, 28, dtype=torch.bool)
        M[:, :14, :14] = False
        M[:, 14:, 14:] = False
        M = M.reshape(B, D)
    else:
        M = torch.rand(B, D, generator=g) > 0.5
    M[:, 0] = True
    X = X_true.unsqueeze(1).repeat(1, K, 1)
    fill = (torch.rand(B, K, D, generator=g) < 0.5).float() if binary else torch.randn(B, K, D, generator=g)
    X = (X * M.unsqueeze(1) + fill * (~M.unsqueeze(1))).to(dev).contiguous()
    return X, M


class LairCabi:
    """One resident LAIR batch driven through the three C-ABI stage calls (cs_lair_encode / _decode /
    _resample_impute), HIP events between the stages."""

    def __init__(self, workload, B, rank, dev):
        from vae_conditional_sampling_amd import _lib
        from vae_conditional_sampling_amd.models import VAE, ResidualFCNetwork
        from vae_conditional_sampling_amd.sampling import engine
        self._lib = _lib
        self.w = w = WORKLOADS[workload]
        self.workload, self.B, self.rank = workload, B, rank
        D, Zd, H, K, R = w["D"], w["Zd"], w["H"], w["K"], w["R"]
        torch.manual_seed(SEED)  # random-init weights of the named architecture (torch's nn.Linear default)
        enc = ResidualFCNetwork(D, 2 * Zd, w["nbe"], H)
        dec = ResidualFCNetwork(Zd, 2 * D, w["nbd"], H)

        def cpu_copy(net):  # float32 CPU copies for the CPU baseline leg
            d = net.weight_dict()
            c = lambda t: t.detach().clone()
            return {"W0": c(d["W0"]), "b0": c(d["b0"]), "blocks": [tuple(c(t) for t in b) for b in d["blocks"]],
                    "Wf": c(d["Wf"]), "bf": c(d["bf"])}

        self.enc_w, self.dec_w = cpu_copy(enc), cpu_copy(dec)
        self.model = model = VAE(dec, "normal_with_eps", enc, "normal_with_eps").to(dev)
        self.X, self.M = synthetic_batch(B, K, D, SEED + 17 * rank, dev)
        self.M8 = self.M.to(torch.uint8).to(dev).contiguous()
        self.ws = ws = engine.lair_workspace(model, B, K, R, 1, dev)
        self.cfg = engine.lair_cfg(model, ws, weighting="dmis", scheme="gr", method="multinomial", cluster_size=0,
                                   weight_alpha=None)
        self.bufs = _lib.make_bufs(X=self.X, M=self.M8, q=ws.q, Z=ws.Z, gen=ws.gen, gen_ll=ws.gen_ll, log_w=ws.log_w,
                                   norm_log_w=ws.norm_log_w, idx=ws.idx, ess=ws.stats[0], tess=ws.stats[1],
                                   perplexity=ws.stats[2])
        self.L = _lib.lib()
        self.h_enc = model.var_latent_network.packed().handle
        self.h_dec = model.generator_network.packed().handle
        self.stream = _lib.current_stream()
        self.f_enc = resfc_flops(D, H, w["nbe"], 2 * Zd)
        self.f_dec = resfc_flops(Zd, H, w["nbd"], 2 * D)

    def step(self, t, events=None):
        _lib, L = self._lib, self.L
        rng = _lib.make_rng(SEED, t, self.rank * self.B)
        cfg, bufs = C.byref(self.cfg), C.byref(self.bufs)
        if events is not None:
            events[0].record()
        _lib.check(L.cs_lair_encode(self.h_enc, cfg, bufs, self.stream))
        if events is not None:
            events[1].record()
        _lib.check(L.cs_lair_decode(self.h_dec, cfg, bufs, C.byref(rng), self.stream))
        if events is not None:
            events[2].record()
        _lib.check(L.cs_lair_resample_impute(cfg, bufs, C.byref(rng), self.stream))
        if events is not None:
            events[3].record()

    def timed(self, steps, warmup, dist=None, events=True):
        # The shader clock takes a few hundred milliseconds of load to leave its idle state (a launch measured cold
        # runs at ~1.4 GHz, after that at ~2.0 GHz): load the GPU for SETTLE_S before the W warm-up steps.
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < SETTLE_S:
            self.step(1 << 20 | n)
            torch.cuda.synchronize()
            n += 1
        for t in range(warmup):
            self.step(t)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)] if events else None
        t0 = time.perf_counter()
        for t in range(steps):
            self.step(warmup + t, evs[t] if events else None)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, evs

    def kernel_report(self, evs, arith):
        """Per-stage launch times (HIP events on the launch stream) and what they are as fractions of the
        arithmetic's ceiling."""
        w, B = self.w, self.B
        K, R, D, Zd = w["K"], w["R"], w["D"], w["Zd"]
        ms = lambda a, b: sum(e[a].elapsed_time(e[b]) for e in evs) / len(evs)
        enc_ms, dec_ms, res_ms = ms(0, 1), ms(1, 2), ms(2, 3)
        enc_flops = self.f_enc * B * K            # algorithmic FLOPs of one encoder launch
        dec_flops = self.f_dec * B * (K + R)
        enc_tf = enc_flops / (enc_ms * 1e-3) / 1e12
        dec_tf = dec_flops / (dec_ms * 1e-3) / 1e12
        peak = peak_of(arith)
        Kp = K + R
        res_bytes = (Kp * 2 * Zd * 4 + Kp * Zd * 4 + Kp * 4 + K * (2 * D * 4 + D * 4) + D) + (K * D * 4 + 2 * Kp * 4 + Kp * 8 + 12)
        split = arith == "bf16x3"
        roofline = {"bound": "mfma", "kernel": ((("cs_mlp_split_kernel" if os.environ.get("CSVAE_SPLIT_TILE") == "32" else "cs_mlp_split16_kernel")
                                                 if split else "cs_mlp_split16_kernel<TERMS=1>" if arith == "bf16"
                                                 else "cs_mlp_split16_kernel<TERMS=2>" if arith == "f16x2" else "cs_mlp_kernel")
                                                + " (encoder launch)"),
                    "achieved": enc_tf, "peak": peak, "unit": "TFLOP/s", "frac": enc_tf / peak,
                    "traffic": None, "flops_per_launch": enc_flops, "avg_launch_ms": enc_ms,
                    "note": ("algorithmic float32 FLOPs; each is executed as 6 bf16 matrix FLOPs, so peak = dense bf16 "
                             "peak 2500 / 6 and the executed bf16 rate is 6 x achieved") if split else
                            ("algorithmic float32 FLOPs; each is executed as 3 fp16 matrix FLOPs, so peak = dense fp16 "
                             "peak 2500 / 3 and the executed fp16 rate is 3 x achieved") if arith == "f16x2" else
                            "operands rounded to bf16, one bf16 matrix FLOP per algorithmic FLOP: peak = dense bf16 peak" if arith == "bf16" else
                            "algorithmic float32 FLOPs on the exact-f32 matrix instructions"}
        kernels = {"encoder_ms": enc_ms, "decoder_ms": dec_ms, "resample_impute_ms": res_ms,
                   # HBM-bound stage: algorithmic bytes per example (q, Z, gen_ll, the K chosen generator rows,
                   # X and M read; X, log-weights, normalised weights, indices, statistics written)
                   "resample_impute_GBps": B * res_bytes / (res_ms * 1e-3) / 1e9, "resample_impute_bytes_per_example": res_bytes,
                   "encoder_frac": enc_tf / peak, "decoder_tflops": dec_tf, "decoder_frac": dec_tf / peak,
                   "step_flops": enc_flops + dec_flops,
                   "step_mfma_frac": (enc_flops + dec_flops) / ((enc_ms + dec_ms + res_ms) * 1e-3) / 1e12 / peak,
                   "mlp_arith": arith}
        return roofline, kernels


def mlp_vae(workload, dev):
    from vae_conditional_sampling_amd.models import VAE, ResidualFCNetwork
    w = WORKLOADS[workload]
    torch.manual_seed(SEED)
    enc = ResidualFCNetwork(w["D"], 2 * w["Zd"], w["nbe"], w["H"])
    dec = ResidualFCNetwork(w["Zd"], 2 * w["D"], w["nbd"], w["H"])
    return VAE(dec, "normal_with_eps", enc, "normal_with_eps").to(dev)


def conv_vae(name, dev):
    from vae_conditional_sampling_amd.models import VAE, ResNetDecoder, ResNetEncoder
    torch.manual_seed(SEED)
    if name == "omniglot":
        enc = ResNetEncoder((1, 28, 28), [2, 2, 2, 2], [64, 128, 256, 512], 100)
        dec = ResNetDecoder((1, 28, 28), [2, 2, 2, 2], [7, None, None, None], [256, 128, 64, 64], (4, 4), 50, 7)
        return VAE(dec, "bern", enc, "normal").to(dev)
    enc = ResNetEncoder((1, 14, 14), [2, 2, 2, 2], [32, 64, 128, 256], 50)
    dec = ResNetDecoder((2, 14, 14), [2, 2, 2, 2], [4, 7, None, None], [128, 64, 32, 32], (1, 1), 25, 4)
    return VAE(dec, "normal_with_eps", enc, "normal_with_eps").to(dev)


def time_sampler(sampler, X0, Mk, total, skip):
    """Wall time per iteration of a sampler generator over iterations [skip, total), device work included."""
    t0 = None
    for t, _ in enumerate(sampler.sampler(X0, Mk)):
        if t == skip - 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (total - skip)


def leg_lair_class(model, B, K, R, D, T, dev, flops_per_example, arith, binary=False, final=True):
    """LAIR through TestVAELatentAdaptiveImportanceResampling.sampler (proposals kept for the final resampling
    as in the reference's configurations), counter RNG."""
    from vae_conditional_sampling_amd.sampling.test_step_vae_sampling import TestVAELatentAdaptiveImportanceResampling
    s = TestVAELatentAdaptiveImportanceResampling(
        num_iterations=T, num_imp_samples=1, weighting_scheme="dmis", resampling_scheme="gr", resampling_method="multinomial",
        num_prior_replenish_proposals=R, resample_final_imps=final, num_copies=K, **BASE_KW)
    s.set_model(model)
    s.on_test_start()
    s.rng_seed = SEED
    s.on_test_batch_start(None, 0)
    X, M = synthetic_batch(B, K, D, SEED + 5, dev, binary=binary)
    Mk = M.unsqueeze(1).to(dev)
    skip = max(2, T // 10)
    sec = time_sampler(s, X, Mk, T, skip)
    tf = B * flops_per_example / sec / 1e12
    return {"ms_per_step": sec * 1e3, "value": B * K / sec, "unit": "samples/s", "examples": B, "K": K, "R": R,
            "iterations_timed": T - skip, "roofline": {"bound": "mfma", "achieved": tf, "peak": peak_of(arith), "unit": "TFLOP/s",
                                                       "frac": tf / peak_of(arith), "what": "whole iteration (all kernels + host)"}}


def leg_lair_run(model, B, K, R, D, T, dev, flops_per_example, arith, per_call=64):
    """LAIR through cs_lair_run: `per_call` iterations per library call, proposals and statistics of every
    iteration written into device stores (what the sampler class does for the iterations whose state nobody
    reads).  This is the latency-bound regime of the reference's own batch size."""
    from vae_conditional_sampling_amd import _lib
    from vae_conditional_sampling_amd.sampling import engine
    X, M = synthetic_batch(B, K, D, SEED + 7, dev)
    M8 = M.to(torch.uint8).to(dev).contiguous()
    ws = engine.lair_workspace(model, B, K, R, 1, dev)
    cfg = engine.lair_cfg(model, ws, weighting="dmis", scheme="gr", method="multinomial", cluster_size=0, weight_alpha=None)
    Zs = torch.empty((T,) + tuple(ws.Z.shape), device=dev)
    lw = torch.empty((T,) + tuple(ws.log_w.shape), device=dev)
    st = torch.empty(3, T, B, device=dev)

    def run(t0, n):
        engine.lair_run(model, ws, X, M8, cfg, n, rng=_lib.make_rng(SEED, t0, 0), Z_store=Zs[t0:t0 + n],
                        log_w_store=lw[t0:t0 + n], stats_store=st[:, t0:t0 + n])
    run(0, per_call)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for c in range(per_call, T, per_call):
        run(c, min(per_call, T - c))
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / (T - per_call)
    assert torch.isfinite(X).all()
    tf = B * flops_per_example / sec / 1e12
    return {"ms_per_step": sec * 1e3, "value": B * K / sec, "unit": "samples/s", "examples": B, "K": K, "R": R,
            "iterations_timed": T - per_call, "iterations_per_call": per_call,
            "roofline": {"bound": "latency", "achieved": tf, "peak": peak_of(arith), "unit": "TFLOP/s", "frac": tf / peak_of(arith),
                         "what": "whole iteration; 2048 / 2560 network rows per launch = 128 / 160 row tiles, a workgroup (CU) "
                                 "each: the time is a CU streaming the network's weights from L2 for its one tile (2.7 MB + "
                                 "1.2 MB per iteration), three dependent launches per iteration"}}


def leg_acmwg_class(model, B, K, D, T, dev, flops_per_chain, arith, binary=False, quadrants=False):
    """AC-MWG steps through TestVAEAdaptiveCollapsedMetropolisWithinGibbs.sampler (no warm-up, prior mixture
    probability 0.05, restricted-availability history on the device), counter RNG."""
    from vae_conditional_sampling_amd.sampling.test_step_vae_sampling import TestVAEAdaptiveCollapsedMetropolisWithinGibbs
    s = TestVAEAdaptiveCollapsedMetropolisWithinGibbs(
        num_pseudo_warmup_iterations=0, num_mwg_iterations=T, prior_mixture_probability=0.05, warmup_method="none",
        num_copies=K, **BASE_KW)
    s.set_model(model)
    s.on_test_start()
    s.rng_seed = SEED
    X, M = synthetic_batch(B, K, D, SEED + 6, dev, binary=binary, quadrants=quadrants)
    Mk = M.unsqueeze(1).to(dev)
    X0 = s.basic_imputation(X, Mk)
    skip = max(2, T // 10)
    sec = time_sampler(s, X0, Mk, T, skip)
    tf = B * K * flops_per_chain / sec / 1e12
    return {"ms_per_step": sec * 1e3, "value": B * K / sec, "unit": "chain-iterations/s", "examples": B, "K": K,
            "iterations_timed": T - skip, "roofline": {"bound": "mfma", "achieved": tf, "peak": peak_of(arith), "unit": "TFLOP/s",
                                                       "frac": tf / peak_of(arith), "what": "whole iteration (all kernels + host)"}}


def extra_legs(args, arith, dev):
    legs = {}
    steps = max(5, args.steps // 2)
    for name in ("hepmass", "gas", "miniboone_k100"):
        if name == args.workload:
            continue
        nb = WHOLE_SET_BATCH if args.batch == 65536 else args.batch   # the whole-test-set configurations: 131 072 examples per GPU
        if name == "miniboone_k100":
            nb = 4096   # 393 216 encoder rows, 409 600 decoder rows
        b = LairCabi(name, nb, 0, dev)
        el, evs = b.timed(steps, args.warmup)
        roof, kern = b.kernel_report(evs, arith)
        w = b.w
        legs[f"lair_{name}"] = {"ms_per_step": el / steps * 1e3, "value": nb * w["K"] * steps / el, "unit": "samples/s",
                                "examples": nb, "K": w["K"], "R": w["R"], "steps": steps,
                                "roofline": {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms")},
                                "kernels": {k: kern[k] for k in ("encoder_ms", "decoder_ms", "resample_impute_ms", "decoder_frac",
                                                                 "step_mfma_frac", "resample_impute_GBps")}}
        del b
        torch.cuda.empty_cache()
    w = WORKLOADS["miniboone"]
    f_enc, f_dec = resfc_flops(w["D"], w["H"], w["nbe"], 2 * w["Zd"]), resfc_flops(w["Zd"], w["H"], w["nbd"], 2 * w["D"])
    model = mlp_vae("miniboone", dev)
    # LAIR MiniBooNE through the sampler class: at the large batch, and at the reference configuration's batch
    legs["lair_miniboone_sampler_class"] = leg_lair_class(model, args.batch, 4, 1, 43, 40, dev, 4 * f_enc + 5 * f_dec, arith)
    legs["lair_miniboone_sampler_class_b512"] = leg_lair_class(model, 512, 4, 1, 43, 300, dev, 4 * f_enc + 5 * f_dec, arith)
    legs["latency_b512"] = leg_lair_run(model, 512, 4, 1, 43, 640, dev, 4 * f_enc + 5 * f_dec, arith)
    legs["acmwg_miniboone"] = leg_acmwg_class(model, args.batch, 5, 43, 40, dev, f_enc + f_dec, arith)
    # AC-MWG at the reference configuration's own batch (B = 256, K = 5): 80 row tiles per launch
    legs["acmwg_miniboone_b256"] = leg_acmwg_class(model, 256, 5, 43, 300, dev, f_enc + f_dec, arith)
    del model
    torch.cuda.empty_cache()
    def layerwise(model, enc_rows, dec_rows, leg):
        # The ResNets' layers are not all matrix-bound (32-channel layers on 14x14 images, resizes, the strided 1x1
        # shortcuts move more bytes than they compute): price every layer at the larger of its matrix time at the
        # arithmetic's ceiling and its memory time at the HBM peak, and report the iteration against that sum.
        from vae_conditional_sampling_amd import _lib
        half = F16X2_PEAK_TFLOPS * 1e12 if lib_arith == "f16x2" else None
        ideal = (_lib.layerwise_ideal_seconds(model.var_latent_network._program().layers, enc_rows, peak_of(arith) * 1e12,
                                              fp16_split_flops_per_s=half) +
                 _lib.layerwise_ideal_seconds(model.generator_network._program().layers, dec_rows, peak_of(arith) * 1e12,
                                              fp16_split_flops_per_s=half))
        leg["layerwise_roofline"] = {"ideal_ms": ideal * 1e3, "frac": ideal * 1e3 / leg["ms_per_step"],
                                     "what": "sum over the layers of both networks of max(algorithmic FLOPs / the layer's matrix "
                                             "ceiling, algorithmic activation bytes / 8 TB/s), divided by the measured iteration"}
        leg["arithmetic"] = ("3x3 halo layers: fp16 split (2 terms, ceiling %.1f); other matrix layers: bf16 split (3 terms, "
                             "ceiling %.1f); roofline.frac is against the bf16-split ceiling" % (F16X2_PEAK_TFLOPS, SPLIT_PEAK_TFLOPS)
                             if half else arith)
        return leg

    # layer programs (the convolutional networks): the 3x3 halo layers follow the fp16 split, every other matrix layer
    # runs the bf16 split under every setting but exact f32; the legs' roofline.frac is against the bf16-split ceiling
    lib_arith = arith
    arith = "f32" if arith == "f32" else "bf16x3"
    fe, fd = CONV_FLOPS["omniglot"]
    m = conv_vae("omniglot", dev)
    legs["acmwg_omniglot_convresnet"] = layerwise(m, 400 * 5, 400 * 5, leg_acmwg_class(m, 400, 5, 784, 12, dev, fe + fd, arith,
                                                                                      binary=True, quadrants=True))
    del m
    torch.cuda.empty_cache()
    fe, fd = CONV_FLOPS["mnist_gmm"]
    m = conv_vae("mnist_gmm", dev)
    legs["lair_mnist_gmm_convresnet"] = layerwise(m, 2048 * 4, 2048 * 5, leg_lair_class(m, 2048, 4, 1, 196, 12, dev, 4 * fe + 5 * fd, arith))
    del m
    torch.cuda.empty_cache()
    return legs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="miniboone", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=65536, help="examples per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="start the ranks, bind each to cuda:LOCAL_RANK, set the process group up, run one all-reduce "
                         "and one all-gather of a few floats, print who sits where, and stop: a check of the launch, "
                         "not a measurement")
    ap.add_argument("--no-extras", action="store_true", help="only the headline workload (no `extra` legs)")
    ap.add_argument("--arith", default=None, choices=["bf16x3", "f16x2", "f32"],
                    help="arithmetic of the fused MLP kernel (default: the library's, bf16x3 = exact 3-way bf16 split of "
                         "the float32 operands, float32 accumulation, float32-equivalent results)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit(launch_ranks(args.gpus))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if REHEARSAL:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        import datetime
        # a rank that cannot join (no peer, no RCCL) raises here after the timeout and the process ends non-zero
        limit = datetime.timedelta(seconds=int(os.environ.get("CSVAE_BENCH_INIT_TIMEOUT", "600")))
        if REHEARSAL:
            dist.init_process_group("gloo", timeout=limit)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=limit)

    if args.dry_run:
        raise SystemExit(dry_run(rank, world, local_rank, dev, dist))

    from vae_conditional_sampling_amd import _lib

    if args.arith:
        _lib.set_mlp_arith(args.arith)
    arith = _lib.get_mlp_arith()
    if arith == "bf16":
        raise SystemExit("bench.py: the headline is measured in float32-equivalent arithmetic only (bf16x3 or f32); "
                         "the plain-bf16 mode is reported in the `plain_bf16` block of the same line")
    bench = LairCabi(args.workload, args.batch, rank, dev)
    w, B = bench.w, args.batch
    D, Zd, H, K, R = w["D"], w["Zd"], w["H"], w["K"], w["R"]

    elapsed, evs = bench.timed(args.steps, args.warmup, dist)
    if dist is not None:
        cdev = torch.device("cpu") if REHEARSAL else dev   # gloo collectives take host tensors
        tt = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # the only collective of the path: final imputations and log-weights are gathered once
        # (outputs in the concatenated form [world * n, ...], which every backend accepts)
        Xl, lwl = bench.X.to(cdev), bench.ws.log_w.reshape(-1).to(cdev)
        gathered = torch.empty(world * Xl.shape[0], *Xl.shape[1:], device=cdev)
        dist.all_gather_into_tensor(gathered, Xl)
        glw = torch.empty(world * lwl.shape[0], device=cdev)
        dist.all_gather_into_tensor(glw, lwl)
        torch.cuda.synchronize()
        assert torch.equal(gathered[rank * Xl.shape[0]:(rank + 1) * Xl.shape[0]], Xl) and torch.isfinite(glw).all(), \
            "gathered shard differs"
    assert torch.isfinite(bench.X).all(), "non-finite imputations"

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    roofline, kernels = bench.kernel_report(evs, arith)
    # HBM bytes per encoder launch from the PMC counters of a profiled run of this command
    # (tools/pmc_summary.py writes the file; null when it has not been collected for this workload and batch)
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_summary.json")) as f:
            pmc = json.load(f)
        if pmc.get("workload") == args.workload and pmc.get("examples_per_gpu") == B and pmc.get("mlp_arith", arith) == arith:
            roofline["traffic"] = pmc["encoder_launch"]["hbm_bytes"]
            roofline["traffic_source"] = ("profiles/pmc_summary.json: rocprofv3 --pmc passes of this command "
                                          "(FETCH_SIZE / WRITE_SIZE), not collected in this run")
    except (OSError, KeyError, ValueError):
        pass

    # the same K steps in the other float32-equivalent arithmetics, for comparison (outside the timed region)
    others = {}
    if world == 1:
        for name, key in (("f32", "exact_f32"), ("bf16x3", "split_bf16x3"), ("f16x2", "split_f16x2")):
            if name == arith:
                continue
            _lib.set_mlp_arith(name)
            el, evo = bench.timed(args.steps, args.warmup)
            ro, ko = bench.kernel_report(evo, name)
            others[key] = {"value": B * K * args.steps / el, "ms_per_step": el / args.steps * 1e3, "unit": "samples/s", "dtype": DTYPES[name],
                           "roofline": {k: ro[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "flops_per_launch", "avg_launch_ms")},
                           "kernels": {k: ko[k] for k in ("encoder_ms", "decoder_ms", "resample_impute_ms", "decoder_frac", "step_mfma_frac")}}
        _lib.set_mlp_arith(arith)

    # and in the opt-in plain-bf16 mode (operands ROUNDED to bf16: reduced precision, never the headline)
    plain = None
    if world == 1:
        _lib.set_mlp_arith("bf16")
        el, evsb = bench.timed(args.steps, args.warmup)
        rb, kb = bench.kernel_report(evsb, "bf16")
        _lib.set_mlp_arith(arith)
        plain = {"what": "opt-in CS_ARITH_BF16: weights and layer inputs rounded to bf16, one matrix instruction per product, "
                         "f32 accumulation; REDUCED PRECISION (about 3e-3 of the output scale from the float32 result), checked "
                         "against a bf16-rounded restatement (tests/test_gpu_bf16_arith.py); not comparable with `value`",
                 "value": B * K * args.steps / el, "ms_per_step": el / args.steps * 1e3, "unit": "samples/s",
                 "roofline": {k: rb[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "flops_per_launch", "avg_launch_ms")},
                 "kernels": {k: kb[k] for k in ("encoder_ms", "decoder_ms", "resample_impute_ms", "decoder_frac", "step_mfma_frac")}}

    value = world * B * K * args.steps / elapsed
    nprod = {"bf16x3": 6, "f16x2": 3}.get(arith)   # 16-bit matrix FLOPs executed per algorithmic FLOP
    split = nprod is not None
    out = {
        "metric": "conditional samples/sec (N x K chains) for LAIR",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "clock_settle_s": SETTLE_S,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPES[arith], "data": "synthetic",
        "config": {"workload": f"uci_{args.workload} MLP-VAE LAIR (dmis/gr/multinomial), D={D} Zd={Zd} H={H} "
                               f"enc_blocks={w['nbe']} dec_blocks={w['nbd']} K={K} R={R} I=1",
                   "examples_per_gpu": B, "chains_per_gpu": B * K, "sharding": f"examples x{world}, no collective in loop"},
        "roofline": roofline,
        "kernels": kernels,
    }
    if REHEARSAL:
        out["rehearsal"] = (f"{world} ranks sharing GPU 0, collectives over gloo: a rehearsal of the multi-rank control flow, "
                            "NOT a multi-GPU measurement")
    out.update(others)
    if plain is not None:
        out["plain_bf16"] = plain
    try:
        with open(os.path.join(ROOT, "roofline.json")) as f:
            out["measured_peaks"] = json.load(f)   # what a box of this kind sustains (tools/micro/peaks.hip), for context
        on_data = out["measured_peaks"].get("mfma_on_data", {}).get("f16_random_operands_tflops" if arith == "f16x2" else "random_operands_tflops")
        if split and on_data:
            # context only (`frac` stays priced on the nominal peak): what the matrix pipes alone sustain on random
            # operands of its instruction (tools/micro/mfma_data.hip), as algorithmic TFLOP/s of this arithmetic's product count
            roofline["matrix_pipes_on_data_tflops"] = on_data / nprod
            roofline["frac_of_matrix_pipes_on_data"] = roofline["achieved"] / (on_data / nprod)
    except (OSError, ValueError):
        pass

    if world == 1:
        # Yardstick measured in this run, on this box: the vendor library's dense GEMM (torch.matmul) on random normal
        # 8192^2 matrices, clock settled.  The split kernels execute 6 bf16 matrix FLOPs per algorithmic FLOP.
        lib = {}
        for name, dt, reps in (("bf16", torch.bfloat16, 40), ("f32", torch.float32, 8)):
            ga = torch.randn(8192, 8192, device=dev).to(dt)
            gb = torch.randn(8192, 8192, device=dev).to(dt)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:
                ga @ gb
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ga @ gb
            e1.record()
            torch.cuda.synchronize()
            lib[f"{name}_tflops"] = 2 * 8192 ** 3 / (e0.elapsed_time(e1) / reps) / 1e9
            del ga, gb
        torch.cuda.empty_cache()
        lib["what"] = (f"torch.matmul (vendor GEMM library) on random normal 8192^2 matrices in this run; the encoder launch "
                       f"executes {nprod} x roofline.achieved 16-bit matrix TFLOP/s" if split else
                       "torch.matmul (vendor GEMM library) on random normal 8192^2 matrices in this run")
        if split:
            lib["encoder_executed_16bit_tflops"] = nprod * roofline["achieved"]
        out["library_gemm_this_run"] = lib

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

        xs = bench.X[:1024].reshape(-1, D).contiguous()
        y64 = forward64(bench.enc_w, xs.cpu().double())
        acc = {}
        for name in ("bf16x3", "f16x2", "f32"):
            _lib.set_mlp_arith(name)
            y = bench.model.var_latent_network(xs).cpu().double()
            acc[name] = float((y - y64).abs().max() / y64.abs().max())
        _lib.set_mlp_arith(arith)
        out["accuracy"] = {"what": "encoder outputs, 4096 rows, max |y - y_float64| / max |y_float64|",
                           "bf16x3": acc["bf16x3"], "f16x2": acc["f16x2"], "exact_f32": acc["f32"]}

    enc_w, dec_w, Xs, Ms = bench.enc_w, bench.dec_w, bench.X[:512].cpu(), bench.M[:512]
    if not args.no_extras and world == 1:
        del bench
        torch.cuda.empty_cache()
        out["extra"] = extra_legs(args, arith, dev)

    if not args.no_cpu_baseline and world == 1:
        from oracle import cs_oracle as O  # the checker, used here only as the timed CPU baseline
        # CPU baseline: the torch-CPU oracle of this repository ("port") on a bounded sample of
        # the same workload: the reference configuration's batch of 512 examples, timed for ~10 s.
        nthreads = min(torch.get_num_threads(), 16)
        torch.set_num_threads(nthreads)
        Bc = 512
        Xc, Mc = Xs, Ms.unsqueeze(1)
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
