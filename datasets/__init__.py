"""Drop-in package name: ``from datasets import datasets_classes`` (reference utils.py:167,174)."""
from . import datasets_classes  # noqa: F401
