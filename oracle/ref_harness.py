"""Harness that imports the *reference* Python package (read-only at /root/reference).

TEST INFRASTRUCTURE ONLY.  This module is used in the build container to
(1) validate the CPU restatement in ``oracle/`` against the reference itself and
(2) generate the golden vectors committed under ``tests/golden/``.
It never runs on the GPU box (the reference tree does not exist there) and nothing in
the product package imports it.

The reference needs ``pytorch_lightning`` and ``torchmetrics`` which are not installed;
both are replaced by minimal in-memory stand-ins that only provide the few names the
reference's modules touch at import time (``LightningModule`` with
``save_hyperparameters`` / ``hparams`` / ``device``).  The package ``__init__`` files of
``irwg.models`` and ``irwg.data`` pull in unrelated third-party trees, so those two
packages are pre-registered as empty namespace modules.
"""
import contextlib
import inspect
import os
import sys
import types

import numpy as np
import torch

REFERENCE_ROOT = os.environ.get("CSVAE_REFERENCE_ROOT", "/root/reference")


class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _install_stubs():
    if "pytorch_lightning" in sys.modules and getattr(sys.modules["pytorch_lightning"], "_csvae_stub", False):
        return
    pl = types.ModuleType("pytorch_lightning")
    pl._csvae_stub = True

    class LightningModule(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self._hp = _AttrDict()

        @property
        def hparams(self):
            return self._hp

        def save_hyperparameters(self):
            # Record the arguments of the calling __init__ (including **kwargs), as
            # Lightning does; subclasses calling it again extend the same dict.
            frame = inspect.currentframe().f_back
            args = inspect.getargvalues(frame)
            for name in args.args:
                if name != "self":
                    self._hp[name] = args.locals[name]
            if args.keywords:
                for k, v in args.locals[args.keywords].items():
                    self._hp[k] = v

        @property
        def device(self):
            for p in self.parameters():
                return p.device
            return torch.device("cpu")

        def on_test_batch_start(self, *a, **k):
            pass

        def on_test_batch_end(self, *a, **k):
            pass

        def on_test_epoch_start(self, *a, **k):
            pass

    class LightningDataModule:
        def __init__(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    pl.LightningDataModule = LightningDataModule
    pl.Trainer = object
    util = types.ModuleType("pytorch_lightning.utilities")
    util_types = types.ModuleType("pytorch_lightning.utilities.types")
    util_types.STEP_OUTPUT = object
    util_types.EPOCH_OUTPUT = object
    pl.utilities = util
    util.types = util_types
    sys.modules["pytorch_lightning"] = pl
    sys.modules["pytorch_lightning.utilities"] = util
    sys.modules["pytorch_lightning.utilities.types"] = util_types
    if "torchmetrics" not in sys.modules:
        tm = types.ModuleType("torchmetrics")
        tmf = types.ModuleType("torchmetrics.functional")
        tm.functional = tmf
        sys.modules["torchmetrics"] = tm
        sys.modules["torchmetrics.functional"] = tmf


def load_reference():
    """Import the reference's hot-path modules and return them in a namespace."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference tree not present at {REFERENCE_ROOT}")
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import irwg  # noqa: F401  (plain package, empty __init__)
    for sub in ("models", "data"):
        name = f"irwg.{sub}"
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REFERENCE_ROOT, "irwg", sub)]
            sys.modules[name] = m
    import importlib
    ns = types.SimpleNamespace()
    ns.vae = importlib.import_module("irwg.models.vae")
    ns.neural_nets = importlib.import_module("irwg.models.neural_nets")
    ns.weights = importlib.import_module("irwg.utils.weights")
    ns.sampling_utils = importlib.import_module("irwg.sampling.utils")
    ns.vae_imputation = importlib.import_module("irwg.sampling.vae_imputation")
    try:
        ns.vae_resnet = importlib.import_module("irwg.models.vae_resnet")
    except Exception as e:  # pragma: no cover
        ns.vae_resnet = None
        ns.vae_resnet_error = e
    try:
        ns.test_step = importlib.import_module("irwg.sampling.test_step_vae_sampling")
    except Exception as e:  # pragma: no cover
        ns.test_step = None
        ns.test_step_error = e
    return ns


# ----------------------------------------------------------------------------------------
# Draw tape: record / replay every random draw the reference makes, so that the same draws
# can be injected into the restatement and into the HIP engine.
# ----------------------------------------------------------------------------------------
class DrawTape:
    """Records torch RNG entry points used by the reference's hot path.

    Recorded entries are ``(kind, ndarray)`` in call order:
      * ``normal``       eps such that the returned value is ``loc + scale*eps`` (bitwise on CPU)
      * ``randn``        the standard-normal tensor itself
      * ``multinomial``  float64 uniforms ``u[rows, n]`` such that the returned index is the
                         left bisection of the row CDF (n > 1, replacement=True)
      * ``exponential``  float32 Exp(1) variates ``e[rows, S]`` of a single-sample multinomial:
                         ATen returns ``argmax(probs / e)`` when one sample is drawn
      * ``rand`` / ``rand_like``  the uniforms
      * ``bernoulli``    float32 uniforms u; result is ``u < p``
      * ``randint``      the integers drawn
      * ``beta``         the samples of a Beta distribution as torch's Dirichlet sampler returned
                         them (rejection sampling inside ATen: the samples, not the uniforms, are
                         what can be replayed)
    """

    def __init__(self, seed=0):
        self.gen = torch.Generator().manual_seed(seed)
        self.entries = []

    def _rec(self, kind, t):
        self.entries.append((kind, t.detach().cpu().numpy().copy()))

    @contextlib.contextmanager
    def recording(self):
        tape = self
        orig = dict(normal=torch.normal, randn=torch.randn, multinomial=torch.multinomial,
                    rand=torch.rand, rand_like=torch.rand_like, bernoulli=torch.bernoulli,
                    randint=torch.randint, dirichlet=torch._sample_dirichlet)

        def sample_dirichlet(concentration, generator=None):
            out = orig["dirichlet"](concentration, generator=tape.gen)
            tape._rec("beta", out[..., 0])  # Beta.rsample = Dirichlet([a, b]).rsample().select(-1, 0)
            return out

        def normal(mean, std, *a, **k):
            if not torch.is_tensor(mean):
                mean = torch.as_tensor(mean, dtype=torch.float32)
            if not torch.is_tensor(std):
                std = torch.as_tensor(std, dtype=torch.float32)
            shape = torch.broadcast_shapes(mean.shape, std.shape)
            eps = orig["randn"](shape, generator=tape.gen, dtype=torch.float32)
            tape._rec("normal", eps)
            return mean + std * eps.to(mean.device)

        def randn(*size, generator=None, device=None, dtype=None, **k):
            if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
                size = tuple(size[0])
            out = orig["randn"](size, generator=tape.gen, dtype=torch.float32)
            tape._rec("randn", out)
            return out.to(device) if device is not None else out

        def rand(*size, generator=None, device=None, dtype=None, **k):
            if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
                size = tuple(size[0])
            out = orig["rand"](size, generator=tape.gen, dtype=torch.float32)
            tape._rec("rand", out)
            return out.to(device) if device is not None else out

        def rand_like(t, **k):
            out = orig["rand"](t.shape, generator=tape.gen, dtype=torch.float32)
            tape._rec("rand_like", out)
            return out.to(t.device)

        def bernoulli(p, *a, **k):
            u = orig["rand"](p.shape, generator=tape.gen, dtype=torch.float32)
            tape._rec("bernoulli", u)
            return (u < p).to(p.dtype)

        def multinomial(probs, num_samples, replacement=False, *, generator=None):
            assert replacement and probs.dim() == 2
            if num_samples == 1:
                e = torch.empty(probs.shape, dtype=torch.float32).exponential_(1, generator=tape.gen)
                tape._rec("exponential", e)
                return multinomial_single_from_exponentials(probs, e)
            u = orig["rand"]((probs.shape[0], num_samples), generator=tape.gen, dtype=torch.float64)
            tape._rec("multinomial", u)
            return multinomial_from_uniforms(probs, u)

        def randint(low, high=None, size=None, **k):
            if size is None:  # randint(high, size)
                low, high, size = 0, low, high
            high_i = int(high)
            out = orig["randint"](int(low), high_i, tuple(size), generator=tape.gen)
            tape._rec("randint", out)
            return out

        # StudentT.rsample: X = _standard_normal(...), Z = Chi2(df).rsample() = _standard_gamma(df / 2) / 0.5
        import torch.distributions.gamma as _gamma_mod
        import torch.distributions.studentT as _studentt_mod
        orig_sn, orig_sg = _studentt_mod._standard_normal, _gamma_mod._standard_gamma

        def standard_normal(shape, dtype, device):
            eps = orig["randn"](tuple(shape), generator=tape.gen, dtype=torch.float32)
            tape._rec("normal", eps)
            return eps.to(device)

        def standard_gamma(concentration):
            out = torch._standard_gamma(concentration, tape.gen)
            tape._rec("std_gamma", out)
            return out

        _studentt_mod._standard_normal, _gamma_mod._standard_gamma = standard_normal, standard_gamma
        torch.normal, torch.randn, torch.multinomial = normal, randn, multinomial
        torch.rand, torch.rand_like, torch.bernoulli, torch.randint = rand, rand_like, bernoulli, randint
        torch._sample_dirichlet = sample_dirichlet
        try:
            yield self
        finally:
            _studentt_mod._standard_normal, _gamma_mod._standard_gamma = orig_sn, orig_sg
            torch.normal, torch.randn, torch.multinomial = orig["normal"], orig["randn"], orig["multinomial"]
            torch.rand, torch.rand_like, torch.bernoulli = orig["rand"], orig["rand_like"], orig["bernoulli"]
            torch.randint = orig["randint"]
            torch._sample_dirichlet = orig["dirichlet"]


def multinomial_from_uniforms(probs, u):
    """torch.multinomial(probs, n, replacement=True) on CPU, with the uniforms made explicit.

    Follows ATen's CPU kernel: float32 running sum of the row, every entry divided by the
    total, last entry forced to 1, then for each float64 uniform the left-most index whose
    cumulative value is >= u (``cum < u`` moves the left pointer).  Equality with the real
    ``torch.multinomial`` under a shared generator is asserted in
    ``oracle/check_against_reference.py``.
    """
    p = probs.detach().cpu().to(torch.float32).numpy()
    u = u.detach().cpu().numpy() if torch.is_tensor(u) else np.asarray(u)
    rows, S = p.shape
    out = np.empty(u.shape, dtype=np.int64)
    for r in range(rows):
        cum = np.empty(S, dtype=np.float32)
        s = np.float32(0.0)
        for j in range(S):
            s = np.float32(s + p[r, j])
            cum[j] = s
        cum = (cum / s).astype(np.float32)
        cum[S - 1] = np.float32(1.0)
        out[r] = np.searchsorted(cum.astype(np.float64), u[r], side="left")
    return torch.from_numpy(out)


def multinomial_single_from_exponentials(probs, e):
    """torch.multinomial(probs[rows,S], 1, replacement=True) on CPU with its Exp(1) variates made
    explicit: ATen's single-sample path divides the probabilities by the variates and returns the
    argmax.  Equality with the real call under a shared generator is asserted in
    ``oracle/check_against_reference.py``."""
    q = probs.detach().cpu().to(torch.float32) / e
    return torch.argmax(q, dim=-1, keepdim=True)
