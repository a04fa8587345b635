"""Host-side VAE object with the interface the samplers use.

Mirrors the inference part of ``irwg.models.vae.VAE`` (irwg/models/vae.py:70-256): the
``DISTRIBUTION`` enum (:19-28), ``hparams``, ``generator_network`` / ``var_latent_network``,
``predict_var_latent_params`` (:180-188), ``split_params`` / ``join_params`` (:190-201),
``create_distribution`` (:203-256), ``get_prior`` (:158-164) and ``sample`` (:166-178).
Training (ELBO, optimisers) is not part of this package: weights are an input.
"""
from enum import Enum, auto

import torch
import torch.nn as nn

from .. import _lib
from ..utils.hparams import AttrDict

EPSILON = 1e-8


class DISTRIBUTION(Enum):
    normal = auto()
    bern = auto()
    normal_with_eps = auto()
    studentt_with_eps22 = auto()
    studentt = auto()
    disc_mix10_logistic_8bit = auto()
    beta = auto()
    logitnormal_with_eps = auto()
    beta_implrep = auto()


_SUPPORTED_LATENT = (DISTRIBUTION.normal, DISTRIBUTION.normal_with_eps, DISTRIBUTION.beta)
# Student-t and logit-normal likelihoods (vae.py:220-229, 244-247): LAIR path only, layer-program decoders
_SUPPORTED_GEN = (DISTRIBUTION.normal, DISTRIBUTION.normal_with_eps, DISTRIBUTION.bern, DISTRIBUTION.studentt,
                  DISTRIBUTION.studentt_with_eps22, DISTRIBUTION.logitnormal_with_eps)
_PROGRAM_ONLY_GEN = (DISTRIBUTION.studentt, DISTRIBUTION.studentt_with_eps22, DISTRIBUTION.logitnormal_with_eps)


def _as_distribution(d):
    if isinstance(d, DISTRIBUTION):
        return d
    if isinstance(d, str):
        return DISTRIBUTION[d.split(".")[-1]]
    raise TypeError(f"not a DISTRIBUTION: {d!r}")


class VAE(nn.Module):
    def __init__(self, generator_network: nn.Module, generator_distribution, var_latent_network: nn.Module,
                 var_latent_distribution, num_latent_samples: int = 1, lr_generator: float = 0.0,
                 lr_latent: float = 0.0, *args, encoder_use_mis_mask: bool = False,
                 use_uniform_prior: bool = False, **kwargs):
        super().__init__()
        self.hparams = AttrDict(generator_distribution=_as_distribution(generator_distribution),
                                var_latent_distribution=_as_distribution(var_latent_distribution),
                                num_latent_samples=num_latent_samples, lr_generator=lr_generator,
                                lr_latent=lr_latent, encoder_use_mis_mask=encoder_use_mis_mask,
                                use_uniform_prior=use_uniform_prior, **kwargs)
        self.generator_network = generator_network
        self.var_latent_network = var_latent_network
        if hasattr(generator_network, "out_split"):
            from .. import _lib
            generator_network.out_split = _lib.gen_nparams(self.hparams.generator_distribution.name)
            if self.hparams.generator_distribution in _PROGRAM_ONLY_GEN and hasattr(generator_network, "force_program"):
                generator_network.force_program = True   # the fused MLP kernel's epilogues do not know these families
        if hasattr(var_latent_network, "out_split"):
            var_latent_network.out_split = 2
        self.eval()

    # ---- engine-facing properties ------------------------------------------------------
    @property
    def device(self):
        for p in self.parameters():
            return p.device
        return torch.device("cpu")

    @property
    def latent_dim(self):
        return self.generator_network.input_dim

    def engine_dists(self):
        """(latent, generator) distribution codes of the C ABI; raises for families the HIP
        kernels do not implement."""
        hp = self.hparams
        if hp.encoder_use_mis_mask:
            raise NotImplementedError("encoder_use_mis_mask=True is not implemented")
        if hp.var_latent_distribution not in _SUPPORTED_LATENT:
            raise NotImplementedError(f"Method not implemented for {hp.var_latent_distribution = }")
        if hp.generator_distribution not in _SUPPORTED_GEN:
            raise NotImplementedError(f"Method not implemented for {hp.generator_distribution = }")
        return (_lib.CS_DIST[hp.var_latent_distribution.name], _lib.CS_DIST[hp.generator_distribution.name])

    def engine_prior(self):
        """Prior code of the C ABI (VAE.get_prior, vae.py:158-164)."""
        return _lib.CS_PRIOR["uniform" if self.hparams.use_uniform_prior else "std_normal"]

    # ---- reference interface -----------------------------------------------------------
    def get_prior(self):
        if self.hparams.use_uniform_prior:
            return torch.distributions.Uniform(low=torch.tensor(0.).to(self.device),
                                               high=torch.tensor(1. + 1e-8).to(self.device))
        return torch.distributions.Normal(loc=0., scale=1.)

    def predict_var_latent_params(self, X, M):
        if self.hparams.encoder_use_mis_mask:
            raise NotImplementedError("encoder_use_mis_mask=True is not implemented")
        return self.var_latent_network(X)

    def split_params(self, params, num_params):
        d = params.shape[-1] // num_params
        return params.reshape(*params.shape[:-1], num_params, d).movedim(-2, 0)

    def join_params(self, params):
        return torch.cat(list(params), dim=-1)

    def create_distribution(self, params, distribution, *, validate_args=None):
        """torch.distributions object for API compatibility; the samplers do not use it (the
        parameterisation is fused into the HIP kernels)."""
        distribution = _as_distribution(distribution)
        if distribution in (DISTRIBUTION.normal, DISTRIBUTION.normal_with_eps):
            loc, raw = self.split_params(params, 2)
            scale = torch.nn.functional.softplus(raw)
            if distribution is DISTRIBUTION.normal_with_eps:
                scale = scale + EPSILON
            return torch.distributions.Normal(loc=loc, scale=scale, validate_args=validate_args)
        if distribution is DISTRIBUTION.bern:
            return torch.distributions.Bernoulli(logits=params, validate_args=validate_args)
        if distribution is DISTRIBUTION.beta:
            a, b = self.split_params(params, 2)
            return torch.distributions.Beta(concentration1=torch.nn.functional.softplus(a),
                                            concentration0=torch.nn.functional.softplus(b), validate_args=validate_args)
        if distribution in (DISTRIBUTION.studentt, DISTRIBUTION.studentt_with_eps22):
            df_raw, loc, raw = self.split_params(params, 3)
            scale = torch.nn.functional.softplus(raw) + (2 ** -2 if distribution is DISTRIBUTION.studentt_with_eps22 else 0.)
            return torch.distributions.StudentT(df=torch.nn.functional.softplus(df_raw), loc=loc, scale=scale,
                                                validate_args=validate_args)
        if distribution is DISTRIBUTION.logitnormal_with_eps:
            loc, raw = self.split_params(params, 2)
            base = torch.distributions.Normal(loc, torch.nn.functional.softplus(raw) + EPSILON, validate_args=validate_args)
            return torch.distributions.TransformedDistribution(base, torch.distributions.SigmoidTransform(),
                                                               validate_args=validate_args)
        raise NotImplementedError(f"Method not implemented for {distribution = }")

    def sample(self, num_samples, rng=None, *, return_latents=False, draws=None):
        """Unconditional samples x ~ p(x|z), z ~ p(z) (vae.py:166-178).

        ``rng``: ``_lib.cs_rng`` key of the counter RNG; ``draws``: optional dict with injected
        'z' [n,Zd] and 'eps_x' [n,D]."""
        from ..sampling import engine
        return engine.vae_sample(self, num_samples, rng=rng, draws=draws, return_latents=return_latents)

    # ---- weights -----------------------------------------------------------------------
    @classmethod
    def load_from_checkpoint(cls, path, *, generator_network, generator_distribution, var_latent_network,
                             var_latent_distribution, map_location="cpu", **kwargs):
        """Load ``generator_network.*`` / ``var_latent_network.*`` from a Lightning ``.ckpt``
        (test.py:127) or a plain state-dict file, using a loader that executes nothing from
        the file."""
        model = cls(generator_network, generator_distribution, var_latent_network, var_latent_distribution, **kwargs)
        obj = torch.load(path, map_location=map_location, weights_only=True)
        sd = obj.get("state_dict", obj)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        if missing:
            raise KeyError(f"checkpoint lacks parameters: {missing}")
        return model
