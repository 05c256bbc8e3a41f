"""Legacy import name of the variational model: `from models.stsve.stsve_hidden_hypersphere import STSVE`
(reference models/spherical_vae.py:16); the shipped class is models/sts/vae.py::STSVAE."""
from ..sts.vae import STSVAE as STSVE  # noqa: F401
from ..sts.vae import STSVAE  # noqa: F401

__all__ = ["STSVE", "STSVAE"]
