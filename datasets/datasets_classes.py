"""Drop-in module name for reference datasets/datasets_classes.py (torchvision-free)."""
from infinite_texture_gans_amd.data import single_image, multiple_images, CropLoader  # noqa: F401
