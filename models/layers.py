"""Drop-in module name for reference models/layers.py."""
from infinite_texture_gans_amd.models.layers import *  # noqa: F401,F403
from infinite_texture_gans_amd.models import layers as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
