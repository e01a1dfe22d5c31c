"""Input pipeline without torchvision (reference datasets/datasets_classes.py, utils.prepare_data).

``single_image`` (reference :12-51): the texture is decoded once (PIL, or a .txt matrix), kept on the GPU as a
[-1, 1] float image; an item is the centre crop if ``center_crop`` is set, else a random crop if ``random_crop`` is
set, else the whole image (the reference's precedence, :27-34); ``len`` is ``sampling`` (10000 when unset).
``multiple_images`` (reference :54-128): a folder of images, optional resize -> crop -> [-1, 1].
``CropLoader`` stands in for the reference's ``DataLoader(shuffle=True, batch_size)``: it yields ``{0: batch}``
dicts; for ``single_image`` the whole batch of random crops is cut on the device in one go.
"""
import os
import random

import numpy as np
import torch


def _center_offsets(h, w, size):
    """torchvision CenterCrop's offsets: int(round((extent - size) / 2.0)) (banker's rounding, as Python's)."""
    if h < size or w < size:
        raise ValueError("center_crop %d exceeds the %dx%d image" % (size, h, w))
    return int(round((h - size) / 2.0)), int(round((w - size) / 2.0))


def _to_normalised(im):
    """PIL image -> (C, H, W) float in [-1, 1]: ToTensor (/255 for 8-bit modes) then Normalize(0.5, 0.5)."""
    if im.mode not in ("RGB", "L", "F"):      # palette / alpha / CMYK files: the models take RGB
        im = im.convert("RGB")
    a = np.asarray(im)
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(np.array(a)).permute(2, 0, 1)       # np.array: a writable copy of PIL's read-only buffer
    t = t.float() / 255.0 if a.dtype == np.uint8 else t.float()
    return (t - 0.5) / 0.5


class single_image:
    def __init__(self, path=None, ext="jpg", center_crop=None, random_crop=None, sampling=None, device="cpu"):
        self.img_path, self.ext = path, ext
        self.center_crop, self.random_crop, self.sampling = center_crop, random_crop, sampling
        if ext == "txt":        # simple binary geological images are stored as text matrices (already normalised)
            arr = np.loadtxt(path)
            img = (torch.from_numpy(arr).float()[None] - 0.5) / 0.5
        else:
            from PIL import Image
            with Image.open(path) as im:
                img = _to_normalised(im)
        self.img = img.contiguous().to(device)

    def __len__(self):
        return self.sampling if self.sampling else 10000

    def crop_batch(self, n, generator=None):
        """n items at once -> (n, C, h, w) on the image's device."""
        _, h, w = self.img.shape
        if self.center_crop:
            t, l = _center_offsets(h, w, self.center_crop)
            one = self.img[:, t:t + self.center_crop, l:l + self.center_crop]
            return one.unsqueeze(0).expand(n, -1, -1, -1).contiguous()
        if self.random_crop:
            c = self.random_crop
            if h < c or w < c:
                raise ValueError("random_crop %d exceeds the %dx%d image" % (c, h, w))
            ys = torch.randint(0, h - c + 1, (n,), generator=generator).tolist()
            xs = torch.randint(0, w - c + 1, (n,), generator=generator).tolist()
            return torch.stack([self.img[:, y:y + c, x:x + c] for y, x in zip(ys, xs)])
        return self.img.unsqueeze(0).expand(n, -1, -1, -1).contiguous()

    def __getitem__(self, idx):
        return {0: self.crop_batch(1)[0]}


class multiple_images:
    def __init__(self, path=None, ext="txt", center_crop=None, random_crop=None, resize=None, sampling=None,
                 device="cpu"):
        self.path, self.ext, self.device = path, ext, device
        self.center_crop, self.random_crop, self.resize, self.sampling = center_crop, random_crop, resize, sampling
        self.img_list = sorted(os.listdir(path))
        if sampling:
            self.img_list = random.sample(self.img_list, sampling)

    def __len__(self):
        return self.sampling if self.sampling else len(self.img_list)

    def __getitem__(self, idx):
        from PIL import Image
        with Image.open(os.path.join(self.path, self.img_list[idx])) as im:
            im.load()
            if self.resize is not None:
                h, w = self.resize
                im = im.resize((w, h), Image.BILINEAR)
            if self.center_crop:
                t, l = _center_offsets(im.size[1], im.size[0], self.center_crop)
                im = im.crop((l, t, l + self.center_crop, t + self.center_crop))
                # the reference follows its centre crop with Resize(64): smaller edge to 64, aspect kept (:79-81)
                w0, h0 = im.size
                s = 64.0 / min(w0, h0)
                im = im.resize((64, int(h0 * s)) if w0 <= h0 else (int(w0 * s), 64), Image.BILINEAR)
            elif self.random_crop:
                c = self.random_crop
                w0, h0 = im.size
                if h0 < c or w0 < c:
                    raise ValueError("random_crop %d exceeds the %dx%d image" % (c, h0, w0))
                t = int(torch.randint(0, h0 - c + 1, (1,)))
                l = int(torch.randint(0, w0 - c + 1, (1,)))
                im = im.crop((l, t, l + c, t + c))
            return {0: _to_normalised(im).to(self.device)}


class CropLoader:
    """for data in loader: data[0] is a (batch, C, H, W) tensor on the dataset's device; the last batch of an
    epoch may be short (DataLoader's drop_last=False)."""

    def __init__(self, dataset, batch_size, seed=None):
        self.dataset, self.batch_size = dataset, batch_size
        self.gen = torch.Generator().manual_seed(seed) if seed is not None else None

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.dataset)
        if isinstance(self.dataset, single_image):
            for b in range(0, n, self.batch_size):
                yield {0: self.dataset.crop_batch(min(self.batch_size, n - b), self.gen)}
            return
        order = torch.randperm(n, generator=self.gen).tolist()      # shuffle=True
        for b in range(0, n, self.batch_size):
            yield {0: torch.stack([self.dataset[i][0] for i in order[b:b + self.batch_size]])}


def SingleImageCrops(path, ext="jpg", sampling=8000, random_crop=None, center_crop=None, batch_size=8, device="cuda",
                     seed=None):
    """Loader over one texture (kept for callers of the round-1 name)."""
    return CropLoader(single_image(path, ext, center_crop, random_crop, sampling, device), batch_size, seed)
