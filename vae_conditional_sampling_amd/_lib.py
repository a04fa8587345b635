"""ctypes binding of libcsvae.so (the C ABI declared in include/csvae.h).

There is no CPU fallback: if the shared library is missing or a call fails, an exception is
raised.  Tensors are passed as raw device pointers (``tensor.data_ptr()``) together with the
current HIP stream of PyTorch.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcsvae.so")

CS_DIST = {"normal": 0, "normal_with_eps": 1, "bern": 2, "beta": 3, "studentt": 4, "studentt_with_eps22": 5,
           "logitnormal_with_eps": 6}
# parameter vectors per dimension of a generator family (decoder output = GEN_NPARAMS * D, split likewise)
GEN_NPARAMS = {"bern": 1, "studentt": 3, "studentt_with_eps22": 3}


def gen_nparams(name):
    return GEN_NPARAMS.get(name, 2)
CS_PRIOR = {"std_normal": 0, "uniform": 1}
CS_WEIGHTING = {"dmis": 0, "dmis_within_groups": 1, "smis": 2}
CS_SCHEME = {"gr": 0, "glr": 1, "lr": 2, "ir": 3}
CS_METHOD = {"multinomial": 0, "systematic": 1}
CS_ANNEAL = {None: 0, "linear": 1, "linear_0.25": 2}


def proposal_transform_args(temperature, anneal_type):
    """(temperature, CS_ANNEAL_*) of var_proposal_temperature / var_proposal_anneal_type_to_prior."""
    if anneal_type not in CS_ANNEAL:
        raise NotImplementedError(f'Method not implemented for anneal_type = {anneal_type!r}')
    if temperature is not None and not temperature > 0:
        raise ValueError("var_proposal_temperature must be positive")
    return float(temperature or 0.0), CS_ANNEAL[anneal_type]


# stream ids of the counter RNG (csrc/cs_common.h, enum CsStream)
STREAM_Z, STREAM_X, STREAM_URES, STREAM_URES2 = 1, 2, 3, 4
STREAM_ZPRIOR, STREAM_UPRIOR, STREAM_UACC, STREAM_HIST = 5, 6, 7, 8
STREAM_INIT_Z, STREAM_INIT_X, STREAM_UFINAL = 9, 10, 11

# cs_acmwg_cfg.flags (include/csvae.h)
CS_MWG_PROPOSE_FROM_CURRENT, CS_MWG_JOINT_TARGET_NEW, CS_MWG_JOINT_TARGET_OLD = 1, 2, 4


class CsError(RuntimeError):
    pass


class cs_mlp_desc(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("out_dim", C.c_int32), ("num_blocks", C.c_int32),
                ("hidden", C.c_int32), ("out_split", C.c_int32)]


CS_OP_CONV, CS_OP_LINEAR, CS_OP_AVGPOOL, CS_OP_RESIZE, CS_OP_LAYERNORM = 1, 2, 3, 4, 5
CS_NET_BUF_IN, CS_NET_BUF_OUT = 0, 5


class cs_layer_desc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("op", "src", "dst", "res", "relu", "cin", "cout", "hin", "win", "hout",
                                         "wout", "ksize", "stride", "pad", "upsample", "in_nchw", "out_nchw")] + \
               [("w_off", C.c_int64), ("b_off", C.c_int64)]


def program_layer_costs(layers):
    """Per layer of a layer program: (algorithmic FLOPs, algorithmic HBM bytes) per input row (image).  FLOPs:
    2 k^2 cin cout hout wout of a convolution / linear layer.  Bytes: the float32 activations a layer must read
    (its source image once -- the pre-upsampling image of an upsampled convolution --, the residual it adds) and
    write; weights are shared by all rows and left out.  Used to price a network against a layer-wise roofline:
    every layer at the larger of its matrix time and its memory time."""
    out = []
    for L in layers:
        src = L.cin * min(L.hin * L.win, L.hout * L.wout * L.ksize * L.ksize)  # a strided 1x1 layer skips pixels
        dst = L.hout * L.wout * L.cout
        flops = 2 * L.ksize * L.ksize * L.cin * L.cout * L.hout * L.wout if L.op in (CS_OP_CONV, CS_OP_LINEAR) else 0
        out.append((flops, 4 * (src + dst + (dst if L.res >= 0 else 0))))
    return out


def layer_runs_fp16_split(L):
    """Whether a layer of a layer program runs the fp16-split (2-term) arithmetic when it is selected
    (layer_half_kind in csrc/cs_net.hip): the convolutions of the matrix kernels (16 input channels and up, NHWC
    input) except the convolutions to at most 4 channels; linear layers and those keep the 3-term bf16 split."""
    if L.op != CS_OP_CONV or L.in_nchw or L.cin < 16:
        return False
    return not (L.cout <= 4 and L.cin in (32, 64, 128, 256) and L.upsample == 1 and L.ksize in (1, 3))


def layerwise_ideal_seconds(layers, rows, flops_per_s, bytes_per_s=8.0e12, fp16_split_flops_per_s=None):
    """Sum over the layers of max(matrix time, memory time) for `rows` rows.  fp16_split_flops_per_s: the matrix
    ceiling of the layers that run the fp16 split (layer_runs_fp16_split), when that arithmetic is selected."""
    total = 0.0
    for L, (f, b) in zip(layers, program_layer_costs(layers)):
        peak = fp16_split_flops_per_s if fp16_split_flops_per_s and layer_runs_fp16_split(L) else flops_per_s
        total += max(f * rows / peak, b * rows / bytes_per_s)
    return total


class cs_rng(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("iteration", C.c_uint32), ("example_offset", C.c_int64)]


class cs_lair_cfg(C.Structure):
    _fields_ = [("B", C.c_int32), ("K", C.c_int32), ("R", C.c_int32), ("I", C.c_int32),
                ("D", C.c_int32), ("Zd", C.c_int32), ("latent_dist", C.c_int32),
                ("gen_dist", C.c_int32), ("weighting", C.c_int32), ("scheme", C.c_int32),
                ("method", C.c_int32), ("cluster_size", C.c_int32), ("weight_alpha", C.c_float),
                ("prior", C.c_int32), ("proposal_temperature", C.c_float), ("proposal_anneal", C.c_int32),
                ("K_impute", C.c_int32), ("glr_pad_to", C.c_int32)]


_LAIR_BUF_FIELDS = ["X", "M", "eps_z", "u_res", "u_res2", "u_sys", "u_sys2", "eps_x", "q", "Z", "gen",
                    "gen_ll", "log_w", "norm_log_w", "idx", "ess", "tess", "perplexity",
                    "latents_resampled", "idx2", "clip_min", "clip_max", "e_res", "e_res2", "X_out", "chi2_x"]


class cs_lair_bufs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _LAIR_BUF_FIELDS]


class cs_lair_run_out(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("Z_store", "log_w_store", "ess_store", "tess_store", "perplexity_store")]


class cs_run_observe(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("X_store", "X_true", "rmse_store", "mae_store", "q_true", "q_work", "wass_store")]


class cs_acmwg_run_out(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("accepted_store", "log_accept_store", "slots_store", "Z_store")]


class cs_acmwg_cfg(C.Structure):
    _fields_ = [("B", C.c_int32), ("K", C.c_int32), ("D", C.c_int32), ("Zd", C.c_int32),
                ("latent_dist", C.c_int32), ("gen_dist", C.c_int32), ("hist_len", C.c_int32),
                ("hist_tail", C.c_int32), ("hist_full", C.c_int32), ("flags", C.c_int32),
                ("prior_mixture_probability", C.c_double), ("proposal_temperature", C.c_float),
                ("proposal_anneal", C.c_int32), ("ring_u8", C.c_int32)]


_ACMWG_BUF_FIELDS = ["X", "M", "Z_old", "gen_old", "z_old_prior_lp", "x_old_lp", "ring", "avail", "r_hist",
                     "eps_z", "z_prior", "u_prior", "u_accept", "eps_x", "slots", "in_off", "use_prior", "z_alt",
                     "q", "Z_new", "gen_new", "ll_new", "accepted", "log_accept", "chi2_x"]


class cs_acmwg_bufs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _ACMWG_BUF_FIELDS]


def make_acmwg_bufs(**tensors):
    b = cs_acmwg_bufs()
    for k, v in tensors.items():
        if k not in _ACMWG_BUF_FIELDS:
            raise KeyError(k)
        setattr(b, k, ptr(v))
    return b


_lib = None


def lib():
    """Load the HIP library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CsError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(or `make -C vae_conditional_sampling_amd/csrc`). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.cs_last_error.restype = C.c_char_p
    L.cs_version.restype = C.c_int
    vp, i64, u32, i32 = C.c_void_p, C.c_int64, C.c_uint32, C.c_int32
    L.cs_mlp_create.argtypes = [C.POINTER(cs_mlp_desc), vp, vp, vp, vp, vp, vp, C.POINTER(vp)]
    L.cs_mlp_destroy.argtypes = [vp]
    L.cs_net_create.argtypes = [C.POINTER(cs_layer_desc), i32, vp, i64, i32, i32, i32, C.POINTER(vp)]
    L.cs_mlp_forward.argtypes = [vp, vp, i64, vp, vp]
    L.cs_proposal_transform.argtypes = [vp, vp, i64, i32, i32, i32, i32, i32, i32, C.c_float, i32, vp]
    L.cs_prior_proposals.argtypes = [vp, i64, i32, i32, i32, vp]
    L.cs_lair_encode.argtypes = [vp, C.POINTER(cs_lair_cfg), C.POINTER(cs_lair_bufs), vp]
    L.cs_lair_decode.argtypes = [vp, C.POINTER(cs_lair_cfg), C.POINTER(cs_lair_bufs), C.POINTER(cs_rng), vp]
    L.cs_lair_resample_impute.argtypes = [C.POINTER(cs_lair_cfg), C.POINTER(cs_lair_bufs), C.POINTER(cs_rng), vp]
    L.cs_lair_step.argtypes = [vp, vp, C.POINTER(cs_lair_cfg), C.POINTER(cs_lair_bufs), C.POINTER(cs_rng), vp]
    L.cs_lair_run.argtypes = [vp, vp, C.POINTER(cs_lair_cfg), C.POINTER(cs_lair_bufs), C.POINTER(cs_rng), i32,
                              C.POINTER(cs_lair_run_out), vp]
    L.cs_lair_run_observed.argtypes = [vp, vp, C.POINTER(cs_lair_cfg), C.POINTER(cs_lair_bufs), C.POINTER(cs_rng), i32,
                                       C.POINTER(cs_lair_run_out), C.POINTER(cs_run_observe), vp]
    L.cs_acmwg_run.argtypes = [vp, vp, C.POINTER(cs_acmwg_cfg), C.POINTER(cs_acmwg_bufs), C.POINTER(cs_rng), i32,
                               C.POINTER(cs_acmwg_run_out), C.POINTER(cs_run_observe), vp]
    L.cs_acmwg_init_state.argtypes = [vp, vp, C.POINTER(cs_acmwg_cfg), C.POINTER(cs_acmwg_bufs), C.POINTER(cs_rng), vp]
    L.cs_acmwg_step.argtypes = [vp, vp, C.POINTER(cs_acmwg_cfg), C.POINTER(cs_acmwg_bufs), C.POINTER(cs_rng), vp]
    L.cs_history_concat.argtypes = [vp, vp, i64, i32, i32, i32, C.POINTER(i32), C.POINTER(i32), vp,
                                    C.POINTER(cs_rng), vp, vp]
    L.cs_history_enqueue.argtypes = [vp, vp, i32, i32, i32, vp, i64, i32, i32, vp]
    L.cs_history_enqueue_u8.argtypes = L.cs_history_enqueue.argtypes
    L.cs_history_concat_u8.argtypes = L.cs_history_concat.argtypes
    L.cs_history_set_available.argtypes = [vp, vp, i64, i64, vp]
    L.cs_final_resample.argtypes = [vp, vp, i64, i64, i32, i32, vp, C.POINTER(cs_rng), i64, vp, vp, vp, vp]
    L.cs_final_resample_static.argtypes = [vp, vp, i64, i64, i32, i32, vp, C.POINTER(cs_rng), i64, vp, vp, vp]
    L.cs_stdnormal_logprob_rows.argtypes = [vp, i64, i32, vp, vp]
    L.cs_decode_params.argtypes = [vp, vp, i64, i32, vp, vp]
    L.cs_gen_loglik_rows.argtypes = [vp, i32, vp, vp, i64, i32, i32, vp, vp]
    L.cs_debug_set_workspace_budget.argtypes = [i64]
    L.cs_debug_force_large_resample.argtypes = [i32]
    L.cs_debug_force_scalar_logw.argtypes = [i32]
    L.cs_debug_set_narrow.argtypes = [i32]
    L.cs_debug_set_small_launch_steps.argtypes = [i32]
    L.cs_debug_set_prog_two_streams.argtypes = [i32]
    L.cs_set_mlp_arith.argtypes = [i32]
    L.cs_mlp_set_arith.argtypes = [vp, i32]
    L.cs_mlp_get_arith.argtypes = [vp]
    L.cs_net_reserve.argtypes = [vp, i64]
    L.cs_row_order_statistics_workspace.restype = i64
    L.cs_row_order_statistics_workspace.argtypes = [i64, i32]
    L.cs_row_order_statistics.argtypes = [vp, i64, i64, C.POINTER(i64), i32, vp, vp, i64, vp]
    L.cs_crc32.restype = u32
    L.cs_crc32.argtypes = [vp, i64, u32]
    L.cs_crc32_combine.restype = u32
    L.cs_crc32_combine.argtypes = [u32, u32, i64]
    L.cs_deflate_rows.restype = i64
    L.cs_deflate_rows.argtypes = [vp, i64, i32, i64, vp, i64]
    L.cs_imputation_errors.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp, vp]
    L.cs_latent_wasserstein.argtypes = [vp, vp, i64, i32, i32, vp, vp]
    L.cs_sample_impute.argtypes = [vp, i32, vp, C.POINTER(cs_rng), u32, vp, vp, i64, i32, i32, vp, vp, i32, vp]
    L.cs_sample_impute2.argtypes = [vp, i32, vp, vp, C.POINTER(cs_rng), u32, vp, vp, i64, i32, i32, vp, vp, i32, vp]
    for fn in ("cs_rng_normal", "cs_rng_uniform", "cs_rng_uniform64"):
        getattr(L, fn).argtypes = [C.POINTER(cs_rng), u32, i64, i32, vp, vp]
    _lib = L
    return L


CS_ARITH = {"f32": 0, "bf16x3": 1, "bf16": 2, "f16x2": 3}


def set_mlp_arith(name):
    """Arithmetic of the fused-MLP matrix products: 'f32' (exact float32 matrix instructions) or 'bf16x3'
    (float32 operands split exactly into three bfloat16 terms, six term products accumulated in float32:
    float32-equivalent results at 6/16 of the matrix time).  Process-wide; returns the previous setting."""
    prev = get_mlp_arith()
    check(lib().cs_set_mlp_arith(CS_ARITH[name]))
    return prev


def get_mlp_arith():
    v = lib().cs_get_mlp_arith()
    return next(k for k, c in CS_ARITH.items() if c == v)


def check(rc):
    if rc != 0:
        raise CsError(f"libcsvae error {rc}: {lib().cs_last_error().decode()}")


def current_stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise CsError("libcsvae operates on device tensors only (no CPU path)")
    if not t.is_contiguous():
        raise CsError("tensor must be contiguous")
    return t.data_ptr()


def make_rng(seed, iteration=0, example_offset=0):
    return cs_rng(int(seed) & 0xFFFFFFFFFFFFFFFF, int(iteration) & 0xFFFFFFFF, int(example_offset))


def make_bufs(**tensors):
    b = cs_lair_bufs()
    for k, v in tensors.items():
        if k not in _LAIR_BUF_FIELDS:
            raise KeyError(k)
        setattr(b, k, ptr(v))
    return b


class _PackedBase:
    handle = None
    out_dim = 0

    def set_arith(self, name):
        """Arithmetic of this network's matrix products ('f32', 'bf16x3', or None = the process-wide setting)."""
        check(lib().cs_mlp_set_arith(self.handle, -1 if name is None else CS_ARITH[name]))

    def get_arith(self):
        v = lib().cs_mlp_get_arith(self.handle)
        return next(k for k, c in CS_ARITH.items() if c == v)

    def reserve(self, rows):
        """Allocate the activation workspace of a layer-program network for launches of up to ``rows`` rows now,
        so that no forward call allocates (cs_net_reserve; nothing to do for the fused MLP kernels)."""
        check(lib().cs_net_reserve(self.handle, int(rows)))

    def forward(self, x):
        shape = x.shape
        x2 = x.reshape(-1, self.in_dim).contiguous().float()
        y = torch.empty(x2.shape[0], self.out_dim, device=x.device, dtype=torch.float32)
        check(lib().cs_mlp_forward(self.handle, ptr(x2), x2.shape[0], ptr(y), current_stream()))
        return y.reshape(*shape[:-1], self.out_dim) if shape[-1] == self.in_dim else y

    def __del__(self):
        try:
            if getattr(self, "handle", None) is not None and _lib is not None:
                _lib.cs_mlp_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class ProgramBuilder:
    """Accumulates the layer list and the weight blob of a layer-program network (cs_net_create).

    Buffer ids: 0 = network input, 1..4 scratch, 5 = network output.  Weights are appended in
    torch's layouts; every tensor starts at a multiple of 4 floats.
    """

    def __init__(self, in_dim, out_dim, out_split):
        self.in_dim, self.out_dim, self.out_split = int(in_dim), int(out_dim), int(out_split)
        self.layers, self.chunks, self.n = [], [], 0

    def _add(self, t):
        if t is None:
            return -1
        t = t.detach().to("cpu", torch.float32).contiguous().reshape(-1)
        off = self.n
        pad = (-t.numel()) % 4
        self.chunks.append(t)
        if pad:
            self.chunks.append(torch.zeros(pad))
        self.n += t.numel() + pad
        return off

    def conv(self, src, dst, weight, bias, *, hin, win, stride=1, upsample=1, relu=False, res=-1, in_nchw=False,
             out_nchw=False):
        cout, cin, k, _ = weight.shape
        pad = (k - 1) // 2
        hout = (hin * upsample + 2 * pad - k) // stride + 1
        wout = (win * upsample + 2 * pad - k) // stride + 1
        self.layers.append(cs_layer_desc(CS_OP_CONV, src, dst, res, int(relu), cin, cout, hin, win, hout, wout, k,
                                         stride, pad, upsample, int(in_nchw), int(out_nchw), self._add(weight),
                                         self._add(bias)))
        return hout, wout

    def linear(self, src, dst, weight, bias, *, relu=False, res=-1):
        out, inp = weight.shape
        self.layers.append(cs_layer_desc(CS_OP_LINEAR, src, dst, res, int(relu), inp, out, 1, 1, 1, 1, 1, 1, 0, 1, 0, 0,
                                         self._add(weight), self._add(bias)))

    def layernorm(self, src, dst, weight, bias, *, relu=False):
        n = weight.numel()
        self.layers.append(cs_layer_desc(CS_OP_LAYERNORM, src, dst, -1, int(relu), n, n, 1, 1, 1, 1, 1, 1, 0, 1, 0, 0,
                                         self._add(weight), self._add(bias)))

    def avgpool(self, src, dst, *, channels, hin, win):
        self.layers.append(cs_layer_desc(CS_OP_AVGPOOL, src, dst, -1, 0, channels, channels, hin, win, 1, 1, 1, 1, 0, 1,
                                         0, 0, 0, -1))

    def resize(self, src, dst, *, channels, hin, win, hout, wout):
        self.layers.append(cs_layer_desc(CS_OP_RESIZE, src, dst, -1, 0, channels, channels, hin, win, hout, wout, 1, 1,
                                         0, 1, 0, 0, 0, -1))

    def build(self):
        return PackedProgram(self)


class PackedProgram(_PackedBase):
    """Device handle of a layer-program network (cs_net_create)."""

    def __init__(self, b):
        L = lib()
        if not torch.cuda.is_available():
            raise CsError("no GPU visible: libcsvae has no CPU path")
        blob = torch.cat(b.chunks) if b.chunks else torch.zeros(4)
        arr = (cs_layer_desc * len(b.layers))(*b.layers)
        h = C.c_void_p()
        check(L.cs_net_create(arr, len(b.layers), blob.data_ptr(), blob.numel(), b.in_dim, b.out_dim, b.out_split,
                              C.byref(h)))
        self.handle = h
        self.in_dim, self.out_dim = b.in_dim, b.out_dim
        self.num_layers = len(b.layers)


class PackedMLP(_PackedBase):
    """Device handle of a residual MLP packed for the fused MFMA kernel (cs_mlp_create)."""

    def __init__(self, W0, b0, blocks, Wf, bf, out_split):
        L = lib()
        if not torch.cuda.is_available():
            raise CsError("no GPU visible: libcsvae has no CPU path")
        f = lambda t: t.detach().to("cpu", torch.float32).contiguous()
        W0, b0, Wf, bf = f(W0), f(b0), f(Wf), f(bf)
        H, in_dim = W0.shape
        out_dim = Wf.shape[0]
        nb = len(blocks)
        if nb:
            Wb = torch.stack([torch.stack([f(W1), f(W2)]) for (W1, b1, W2, b2) in blocks]).contiguous()
            bb = torch.stack([torch.stack([f(b1), f(b2)]) for (W1, b1, W2, b2) in blocks]).contiguous()
        else:
            Wb = torch.zeros(1); bb = torch.zeros(1)
        self.desc = cs_mlp_desc(in_dim, out_dim, nb, H, out_split)
        self.in_dim, self.out_dim, self.hidden, self.num_blocks = in_dim, out_dim, H, nb
        h = C.c_void_p()
        check(L.cs_mlp_create(C.byref(self.desc), W0.data_ptr(), b0.data_ptr(), Wb.data_ptr(), bb.data_ptr(),
                              Wf.data_ptr(), bf.data_ptr(), C.byref(h)))
        self.handle = h
