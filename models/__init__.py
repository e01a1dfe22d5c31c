"""Drop-in package name: ``from models import generators, discriminators, layers`` (reference test_sample.py:8,
utils.py:12-13) resolves to the MI355X modules of infinite_texture_gans_amd.models."""
from infinite_texture_gans_amd.models import generators, discriminators, layers  # noqa: F401
