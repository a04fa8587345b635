from .neural_nets import ResidualFCNetwork  # noqa: F401
from .vae import DISTRIBUTION, VAE  # noqa: F401
