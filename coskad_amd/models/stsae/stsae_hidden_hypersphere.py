"""Legacy import name of the autoencoder: `from models.stsae.stsae_hidden_hypersphere import STSAE`
(reference models/euclidean_autoencoder.py:18); the shipped class is models/sts/ae.py::STSAE."""
from ..sts.ae import STSAE  # noqa: F401

__all__ = ["STSAE"]
