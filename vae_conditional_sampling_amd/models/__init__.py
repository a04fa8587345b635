from .neural_nets import ResidualFCNetwork  # noqa: F401
from .vae import DISTRIBUTION, VAE  # noqa: F401
from .vae_resnet import ResNetDecoder, ResNetEncoder  # noqa: F401
