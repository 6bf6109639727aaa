"""polyphemus_amd — MI355X-native graph-VAE hot path of Polyphemus."""
from . import constants  # noqa: F401
