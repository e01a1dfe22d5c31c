from . import layers, generators, discriminators  # noqa: F401
