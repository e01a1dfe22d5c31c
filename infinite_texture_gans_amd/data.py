"""Single-image random-crop input pipeline without torchvision (reference
datasets/datasets_classes.py:12-51): the texture is decoded once (PIL, or a .txt matrix), kept on
the GPU as a [-1, 1] float image, and every batch is `batch_size` random crops taken on the device.
`__len__` semantics follow the reference: an epoch is `sampling` crops."""
import numpy as np
import torch


class SingleImageCrops:
    def __init__(self, path, ext="jpg", sampling=8000, random_crop=None, center_crop=None, batch_size=8,
                 device="cuda", seed=None):
        if ext == "txt":
            arr = np.loadtxt(path, dtype=np.float32)
            arr = arr[None] if arr.ndim == 2 else arr
            img = torch.from_numpy(arr)
        else:
            from PIL import Image
            with Image.open(path) as im:
                arr = np.asarray(im.convert("RGB"), dtype=np.float32) / 255.0   # ToTensor
            img = torch.from_numpy(arr).permute(2, 0, 1)
        img = (img - 0.5) / 0.5                                                   # Normalize(0.5, 0.5)
        if center_crop:
            _, h, w = img.shape
            t, l = (h - center_crop) // 2, (w - center_crop) // 2
            img = img[:, t:t + center_crop, l:l + center_crop]
        self.img = img.contiguous().to(device)
        self.crop = random_crop
        self.sampling, self.batch_size = sampling, batch_size
        self.gen = torch.Generator().manual_seed(seed) if seed is not None else None

    def __len__(self):
        return self.sampling

    def __iter__(self):
        c, h, w = self.img.shape
        n_batches = (self.sampling + self.batch_size - 1) // self.batch_size
        for b in range(n_batches):
            bs = min(self.batch_size, self.sampling - b * self.batch_size)
            if self.crop is None:
                yield {0: self.img.unsqueeze(0).expand(bs, -1, -1, -1).contiguous()}
                continue
            ys = torch.randint(0, h - self.crop + 1, (bs,), generator=self.gen).tolist()
            xs = torch.randint(0, w - self.crop + 1, (bs,), generator=self.gen).tolist()
            yield {0: torch.stack([self.img[:, y:y + self.crop, x:x + self.crop] for y, x in zip(ys, xs)])}
