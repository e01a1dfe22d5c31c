"""Drop-in module name: the reference's callers do ``import utils`` / ``from utils import prepare_models, ...``
(reference train.py:10, test_sample.py:7).  Everything lives in infinite_texture_gans_amd.utils."""
from infinite_texture_gans_amd.utils import *  # noqa: F401,F403
from infinite_texture_gans_amd.utils import (  # noqa: F401  (names a star import would skip or callers name explicitly)
    prepare_parser, prepare_device, prepare_seed, prepare_data, prepare_models, prepare_filename, elapsed_time,
    init_weight, merge_patches_into_image, crop_images, crop_image, build_z, build_maps, tile_process,
    sample_from_gen, sample_from_gen_PatchByPatch_train, sample_from_gen_PatchByPatch_test, sample_latents_train)
