"""Oracle: patch merge / overlapping crop / local padding (test infrastructure).

Restates, independently of the reference's Python loops:
  * ``utils.merge_patches_into_image``  (reference utils.py:577-613)
  * ``utils.crop_images`` / ``crop_image``  (reference utils.py:658-742)
  * ``LocalPadder.forward`` training branch  (reference models/layers.py:145-173,
    ``padding`` :78-82)
  * ``LocalPadder`` eval-mode streaming state machine
    (reference models/layers.py:84-143)

Patch ordering is image-major, then grid row, then grid column
(p = n*gh*gw + r*gw + c, reference utils.py:604-607).
"""
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- merge / crop
def merge(patches, gh, gw):
    """(N*gh*gw, C, ph, pw) -> (N, C, gh*ph, gw*pw).  reference utils.py:577-613."""
    np_, c, ph, pw = patches.shape
    n = np_ // (gh * gw)
    x = patches.reshape(n, gh, gw, c, ph, pw).permute(0, 3, 1, 4, 2, 5)
    return x.reshape(n, c, gh * ph, gw * pw)


def crop(img, size_h, size_w, stride):
    """Sliding-window crops, window (size_h,size_w), same stride on both axes.

    (N, C, H, W) -> (N*nh*nw, C, size_h, size_w); windows enumerated row-major per
    image, images outermost.  reference utils.py:658-742 (the while-loops at
    :721-740 admit a window iff its end index <= extent).
    """
    n, c, h, w = img.shape
    if h < size_h or w < size_w:
        return img.new_zeros((0,))
    u = img.unfold(2, size_h, stride).unfold(3, size_w, stride)  # N,C,nh,nw,sh,sw
    nh, nw = u.shape[2], u.shape[3]
    return u.permute(0, 2, 3, 1, 4, 5).reshape(n * nh * nw, c, size_h, size_w)


def merge_loops(patches, gh, gw):
    """Loop-faithful variant (torch.cat per patch) used for the 'reference-faithful'
    CPU baseline timing; same result as :func:`merge`.  reference utils.py:594-611."""
    np_ = patches.shape[0]
    per = gh * gw
    imgs = []
    for k in range(np_ // per):
        rows = []
        for r in range(gh):
            row = patches[k * per + r * gw]
            for c in range(1, gw):
                row = torch.cat((row, patches[k * per + r * gw + c]), dim=-1)
            rows.append(row)
        img = rows[0]
        for r in range(1, gh):
            img = torch.cat((img, rows[r]), dim=-2)
        imgs.append(img)
    out = torch.empty((len(imgs),) + tuple(imgs[0].shape), dtype=torch.float32)
    for k, im in enumerate(imgs):
        out[k] = im
    return out


def crop_loops(img, size_h, size_w, stride):
    """Loop-faithful variant of :func:`crop` (growing torch.cat), reference utils.py:679-742."""
    img = img.clone()
    out = torch.tensor([])
    for l in range(img.shape[0]):
        one = img[l]
        crops = torch.tensor([])
        y0 = 0
        while y0 + size_h <= one.shape[1]:
            x0 = 0
            while x0 + size_w <= one.shape[2]:
                crops = torch.cat((crops, one[:, y0:y0 + size_h, x0:x0 + size_w].unsqueeze(0)))
                x0 += stride
            y0 += stride
        out = torch.cat((out, crops), 0)
    return out


# --------------------------------------------------------------------------- local padding, training mode
def local_pad(x, gh, gw, outer="replicate", merged_input=False, loops=False):
    """Training-mode LocalPadder (also eval with image_location 1st_row & 1st_col).

    x: (N*gh*gw, C, P, P) patches, or, when ``merged_input`` (the generator's
    ``start`` layer, reference models/generators.py:59), the already merged and
    randomly padded latent (N, C, gh*b+2, gw*b+2) which is only cropped
    (reference models/layers.py:152-155,165-170).
    Returns (N*gh*gw, C, P+2, P+2).
    """
    mg = merge_loops if loops else merge
    cr = crop_loops if loops else crop
    if merged_input:
        p = x.shape[-1] // gw
        return cr(x, p + 2, p + 2, p)
    p = x.shape[-1]
    m = mg(x, gh, gw)
    m = F.pad(m, (1, 1, 1, 1), mode=outer)  # 'replicate' | 'constant' (zeros)
    return cr(m, p + 2, p + 2, p)


# --------------------------------------------------------------------------- eval-mode streaming padder
class StreamPadder:
    """Per-layer state of the eval-mode LocalPadder (reference models/layers.py:66-76,
    84-143).  One instance per conv2d_lp layer; call once per sub-image in raster order."""

    def __init__(self, gh, gw, outer="replicate", merge_input=True):
        self.gh, self.gw, self.outer, self.merge_input = gh, gw, outer, merge_input
        self.v = None          # left column used now
        self.v_next = None     # left column for the next sub-image
        self.h = None          # top row used now (1 x (gw*P+2))
        self.h_cur = None      # remaining top-row buffer of this sub-image row
        self.h_next = None     # row being assembled for the next sub-image row

    def _update(self, m, loc, ph, pw):
        gh, gw = self.gh, self.gw
        if self.v_next is not None:
            self.v = self.v_next
        self.v_next = None if "last_col" in loc else m[:, :, :, [pw * (gw - 1) - 1]]
        if "last_col" in loc:
            s = m[:, :, [ph * (gh - 1) - 1], :]
        else:
            s = m[:, :, [ph * (gh - 1) - 1], :pw * (gw - 1)]
        if "1st_col" in loc:
            if "1st_row" not in loc:
                self.h_cur = F.pad(self.h_next.clone(), (1, 1, 0, 0), mode=self.outer)
            self.h_next = s
        else:
            self.h_next = torch.cat((self.h_next, s), -1)
        if self.h_cur is not None:
            self.h = self.h_cur[:, :, :, :gw * pw + 2].clone()
            self.h_cur = None if "last_col" in loc else self.h_cur[:, :, :, (gw - 1) * pw:]

    def _pad(self, m, loc):
        o = self.outer
        if "1st_row" in loc and "1st_col" in loc:
            return F.pad(m, (1, 1, 1, 1), mode=o)
        if "1st_row" in loc:
            return F.pad(torch.cat((self.v, m), -1), (0, 1, 1, 1), mode=o)
        if "1st_col" in loc:
            return torch.cat((self.h, F.pad(m, (1, 1, 0, 1), mode=o)), -2)
        t = F.pad(torch.cat((self.v, m), -1), (0, 1, 0, 1), mode=o)
        return torch.cat((self.h, t), -2)

    def __call__(self, x, loc):
        if self.merge_input:
            ph, pw = x.shape[-2], x.shape[-1]
            m = merge(x, self.gh, self.gw)
        else:
            ph, pw = x.shape[-2] // self.gh, x.shape[-1] // self.gw
            m = x
        self._update(m, loc, ph, pw)
        if self.merge_input:
            m = self._pad(m, loc)
        return crop(m, pw + 2, pw + 2, pw)
