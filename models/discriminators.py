"""Drop-in module name for reference models/discriminators.py."""
from infinite_texture_gans_amd.models.discriminators import *  # noqa: F401,F403
from infinite_texture_gans_amd.models import discriminators as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
