"""ResidualPatchGenerator with the reference's constructor signature, attributes and
state_dict keys (reference models/generators.py:4-124), running on the HIP kernels."""
import torch.nn as nn

from .. import ops
from ..ops import GT
from .layers import LocalPadder, Attention, conv2d_lp, ResBlockGenerator, _BNParams, up2_fold_enabled


class ResidualPatchGenerator(nn.Module):
    """Patch-by-patch residual generator.

    forward(z, maps=None, image_location='1st_row_1st_col'):
      z     merged latent (N, z_dim, nph*base_res+2, npw*base_res+2), randomly pre-padded
      maps  list of n_layers_G tensors (N*nph*npw, map_dim, r+4, r+4) for SSM, or Nones
      ->    patches (N*nph*npw, img_ch, P, P) in (-1, 1), P = base_res * 2**(n_layers_G-1)
    """

    def __init__(self, z_dim=128, G_ch=64, base_res=4, n_layers_G=4, attention=True, img_ch=3,
                 leak=0, SN=False, type_norm='BN', map_dim=1,
                 padding_mode='local', outer_padding='replicate',
                 num_patches_h=3, num_patches_w=3, padding_size=1, conv_reduction=2):
        super().__init__()
        self.z_dim, self.base_ch, self.base_res, self.n_layers_G = z_dim, G_ch, base_res, n_layers_G
        self.img_ch, self.leak, self.SN, self.type_norm, self.map_dim = img_ch, leak, SN, type_norm, map_dim
        self.padding_mode, self.outer_padding = padding_mode, outer_padding
        self.num_patches_h, self.num_patches_w = num_patches_h, num_patches_w
        self.padding_size, self.conv_reduction = padding_size, conv_reduction
        if padding_size != 1 or conv_reduction != 2:
            raise ValueError("local padding is defined for 3x3 convs: padding_size=1, conv_reduction=2")
        if n_layers_G not in (4, 5, 6):
            raise ValueError("n_layers_G must be 4, 5 or 6")
        # kept for API compatibility: the reference configures LocalPadder through class attributes
        LocalPadder.set_attributes(num_patches_h=num_patches_h, num_patches_w=num_patches_w,
                                   outer_padding=outer_padding, padding_size=padding_size,
                                   conv_reduction=conv_reduction)
        self.up = nn.Upsample(scale_factor=2, mode='nearest')
        self.activation = nn.LeakyReLU(leak) if leak > 0 else nn.ReLU()
        c = G_ch
        self.start = conv2d_lp(z_dim, c * 8, SN, padding_mode, merge_patches_into_image=False)
        widths = [c * 8, c * 8, c * 4, c * 2, c, c // 2, c // 4]
        for i in range(1, n_layers_G + 1):
            setattr(self, "block%d" % i, ResBlockGenerator(self, widths[i - 1], widths[i], padding_mode=padding_mode))
        # blocks 2.. sit behind the x2 upsample (forward_grid): their first conv folds it into its filter, and a step engine
        # keeps the folded panels for it (layers._ConvParams.up2); SSM modulates at the upsampled resolution: no fold there
        for i in range(2, n_layers_G + 1):
            getattr(self, "block%d" % i).conv1.conv.up2 = type_norm == 'BN' and up2_fold_enabled()
        final_chin = widths[n_layers_G]
        if type_norm == 'BN':
            self.bn = _BNParams(final_chin)
        self.attention = Attention(c * 2, SN=SN) if attention else attention
        self.final = conv2d_lp(final_chin, img_ch, SN, padding_mode)
        for m in self.modules():
            if isinstance(m, LocalPadder):
                m.pin(num_patches_h, num_patches_w, outer_padding)

    @property
    def execution_order(self):
        """Names of the parameter-carrying children in the order forward() runs them (registration order differs:
        `bn` and `attention` are registered behind the blocks as in the reference, generators.py:76-83); the bucketed
        gradient exchange only splits the flat gradient where both orders agree (engine.GradExchange)."""
        names = ["start"]
        for i in range(1, self.n_layers_G + 1):
            names.append("block%d" % i)
            if i == 3 and self.attention:
                names.append("attention")
        if self.type_norm == 'BN':
            names.append("bn")
        return names + ["final"]

    band_layout = None      # (patch rows, patch columns) of the band a row-sharded trainer holds in image layout (engine.BandTrainer)

    def set_sync(self, sync):
        """Make every BatchNorm (incl. SSM's inner one) take its statistics over ``sync``'s ranks."""
        for m in self.modules():
            if isinstance(m, _BNParams):
                m.sync = sync

    def reset_stream_state(self):
        for m in self.modules():
            if isinstance(m, LocalPadder):
                m.reset_state()

    def forward_grid(self, z, maps=None, image_location='1st_row_1st_col'):
        """z: NCHW merged latent.  Returns the patch-grid GT of the generated patches."""
        if maps is None:
            maps = [None] * self.n_layers_G
        A, s = ops.ACT_LRELU, float(self.leak)
        gh, gw = self.num_patches_h, self.num_patches_w
        # every conv whose output goes straight into a training-mode BatchNorm takes that BatchNorm's statistics in its
        # own epilogue (the block after an attention layer normalises the attention output instead: no fusion there)
        bn = self.type_norm == 'BN' and self.training
        h = self.start.forward_grid(ops.to_grid(z, 1, 1, merged=True), image_location, out_stats=bn)
        if self.padding_mode != 'local':
            gh, gw = 1, 1
        h = self.block1.forward_grid(h, maps[0], image_location, out_stats=bn)
        for i in range(2, self.n_layers_G + 1):
            h = getattr(self, "block%d" % i).forward_grid(h, maps[i - 1], image_location, upsample_input=True,
                                                          out_stats=bn and not (i == 3 and self.attention))
            if i == 3 and self.attention:
                if self.band_layout is not None:
                    # row-sharded training holds the band in image layout (one "patch" per image): attention mixes inside
                    # each generator patch, so it runs on the band's patch grid and the result goes back to image layout
                    rows, cols = self.band_layout
                    n, _, _, H, W, ld = h.t.shape
                    ph, pw = H // rows, W // cols
                    g = h.t.view(n, rows, ph, cols, pw, ld).permute(0, 1, 3, 2, 4, 5).contiguous()
                    o = self.attention.run(GT(g, h.c))
                    h = GT(o.t.permute(0, 1, 3, 2, 4, 5).reshape(n, 1, 1, H, W, ld), o.c)
                else:
                    h = self.attention.run(h)
        if self.type_norm == 'BN':
            return self.final.forward_grid(h, image_location, act=ops.ACT_TANH, bn=self.bn, bn_act=(A, s))
        h = ops.act(h, A, s)
        return self.final.forward_grid(h, image_location, act=ops.ACT_TANH)

    def forward(self, z, maps=None, image_location='1st_row_1st_col'):
        return ops.to_nchw(self.forward_grid(z, maps, image_location), merged=False)
