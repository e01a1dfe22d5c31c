"""MI355X-native patch-by-patch texture GAN (drop-in for the hot path of
ai4netzero/Infinite_Texture_GANs): hand-written HIP kernels behind the reference's
``models.generators`` / ``models.discriminators`` / ``utils`` / ``train.py`` surface."""
__version__ = "0.1.0"
