"""PatchDiscriminator with the reference's constructor signature and state_dict keys
(reference models/discriminators.py:156-210) on the HIP conv kernels.  The other
discriminator classes of the reference are dead code there (utils.py:205-207) and are
not provided."""
import torch
import torch.nn as nn

from .. import ops
from ..ops import GT
from .layers import conv4x4, conv3x3, _BNParams, _INParams, _ConvParams, batched_power_iteration


class PatchDiscriminator(nn.Module):
    """PatchGAN discriminator: conv(s2)+LReLU(0.2), n_layers_D-1 x [conv+LReLU], conv -> logit map."""

    def __init__(self, img_ch=1, base_ch=64, n_layers_D=4, kw=4, SN=False, norm_layer=None):
        super().__init__()
        self.img_ch = img_ch
        if kw == 4:
            conv_fun = conv4x4
        elif kw == 3:
            conv_fun = conv3x3
        else:
            raise ValueError("kw must be 3 or 4")
        if norm_layer not in (None, 'batch', 'instance'):
            raise ValueError("norm_layer must be None, 'batch' or 'instance', got %r" % (norm_layer,))
        nf = base_ch
        seq = [conv_fun(img_ch, base_ch, SN=SN, s=2, bias=True), nn.LeakyReLU(0.2, False)]
        for n in range(1, n_layers_D):
            nf_prev, nf = nf, min(nf * 2, 512)
            stride = 1 if n == n_layers_D - 1 else 2
            seq.append(conv_fun(nf_prev, nf, s=stride, SN=SN, bias=True))
            if norm_layer == 'batch':
                seq.append(_BNParams(nf, affine=True))
            elif norm_layer == 'instance':
                seq.append(_INParams(nf, affine=False))
            seq.append(nn.LeakyReLU(0.2, False))
        seq.append(conv_fun(nf, 1, s=1, SN=SN, bias=True))
        self.model = nn.Sequential(*seq)

    def forward_grid(self, x):
        """x: GT image (any patch grid).  Returns the logit map as a 1x1-grid GT."""
        mods = list(self.model)
        batched_power_iteration(self.model)      # every layer's spectral norm up front, 4 launches
        # conv -> LeakyReLU -> conv: the activation is fused into the first conv's epilogue, and its backward
        # into the second conv's input-gradient epilogue (the mask is the sign of the tensor both share)
        h, i, in_act = x, 0, None
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(nxt, nn.LeakyReLU):
                chained = i + 2 < len(mods) and isinstance(mods[i + 2], _ConvParams) and torch.is_grad_enabled()
                h = m.run(h, act=ops.ACT_LRELU, slope=nxt.negative_slope, out_grid=(1, 1), in_act=in_act,
                          defer_act_bwd=chained)
                in_act = (ops.ACT_LRELU, nxt.negative_slope) if chained else None
                i += 2
            elif isinstance(nxt, (_BNParams, _INParams)):
                h = m.run(h, out_grid=(1, 1), in_act=in_act)
                h = nxt.run(h, act=ops.ACT_LRELU, slope=mods[i + 2].negative_slope)
                in_act = None
                i += 3
            else:
                h = m.run(h, out_grid=(1, 1), in_act=in_act)
                in_act = None
                i += 1
        return h

    def forward(self, x):
        if isinstance(x, GT):
            return self.forward_grid(x)
        return ops.to_nchw(self.forward_grid(ops.to_grid(x, 1, 1, merged=True)), merged=True)
