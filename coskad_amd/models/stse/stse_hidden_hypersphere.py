"""Legacy import name of the encoder-only model: the reference's Lightning wrappers still say
`from models.stse.stse_hidden_hypersphere import STSE` (models/euclidean_encoder_staticCenter.py:19,
euclidean_encoder_dynamicCenter.py:14, hyperbolic_encoder.py:16) although that package is absent from the
snapshot; the shipped class lives in models/sts/ae.py.  With `coskad_amd` on the path as `models`
 those imports resolve here."""
from ..sts.ae import STSE  # noqa: F401

__all__ = ["STSE"]
