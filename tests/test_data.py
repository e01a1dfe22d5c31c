"""CPU tests of the host-side pipeline around the hot path: the torchvision-free datasets against the semantics of
reference datasets/datasets_classes.py:12-51 (ToTensor -> Normalize(0.5, 0.5); center_crop takes precedence over
random_crop; CenterCrop offsets int(round((extent - size) / 2)); len = sampling or 10000; items are {0: tensor}),
utils.prepare_data's (loader, dataset) contract (reference utils.py:158-191) and the learning-rate schedules against
torch.optim.lr_scheduler (reference train.py:60-70,183-185)."""
import numpy as np
import torch


def _png(tmp_path, h=37, w=53):
    from PIL import Image
    rng = np.random.RandomState(1)
    a = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
    p = tmp_path / "tex.png"
    Image.fromarray(a).save(p)
    return str(p), a


def test_single_image_decode_normalise_and_crop_semantics(tmp_path):
    from infinite_texture_gans_amd.data import single_image
    path, a = _png(tmp_path)
    want = (torch.from_numpy(a).permute(2, 0, 1).float() / 255.0 - 0.5) / 0.5        # ToTensor + Normalize(0.5, 0.5)
    ds = single_image(path, "png", sampling=None)
    assert len(ds) == 10000 and torch.equal(ds.img, want) and torch.equal(ds[0][0], want)
    ds = single_image(path, "png", random_crop=16, sampling=77)
    assert len(ds) == 77
    torch.manual_seed(0)
    for _ in range(20):
        c = ds[0][0]
        assert c.shape == (3, 16, 16)
        # every random crop is a window of the image
        hits = [(y, x) for y in range(37 - 15) for x in range(53 - 15) if torch.equal(want[:, y:y + 16, x:x + 16], c)]
        assert hits
    # center_crop wins over random_crop (reference :27-34) and uses torchvision's rounded offsets
    ds = single_image(path, "png", center_crop=20, random_crop=16)
    t, l = int(round((37 - 20) / 2.0)), int(round((53 - 20) / 2.0))
    assert torch.equal(ds[3][0], want[:, t:t + 20, l:l + 20])


def test_txt_images_and_multiple_images(tmp_path):
    from infinite_texture_gans_amd.data import single_image, multiple_images, CropLoader
    m = np.random.RandomState(2).rand(9, 11)
    np.savetxt(tmp_path / "geo.txt", m)
    ds = single_image(str(tmp_path / "geo.txt"), "txt")
    assert ds.img.shape == (1, 9, 11) and torch.allclose(ds.img[0], torch.from_numpy((m - 0.5) / 0.5).float(), atol=1e-6)
    from PIL import Image
    d = tmp_path / "many"
    d.mkdir()
    rng = np.random.RandomState(3)
    for i in range(5):
        Image.fromarray(rng.randint(0, 256, (24 + i, 30, 3), dtype=np.uint8)).save(d / ("%d.png" % i))
    ds = multiple_images(str(d), "png", random_crop=12)
    assert len(ds) == 5
    b = next(iter(CropLoader(ds, 4, seed=0)))
    assert set(b) == {0} and b[0].shape == (4, 3, 12, 12) and float(b[0].min()) >= -1 and float(b[0].max()) <= 1
    ds = multiple_images(str(d), "png", center_crop=20)          # the reference follows its centre crop with Resize(64)
    assert ds[0][0].shape == (3, 64, 64)


def test_prepare_data_contract(tmp_path):
    from infinite_texture_gans_amd import utils as U
    path, _ = _png(tmp_path)
    args = U.prepare_parser().parse_args(["--data_path", path, "--data_ext", "png", "--random_crop", "8", "--sampling", "10",
                                          "--batch_size", "4"])
    loader, ds = U.prepare_data(args, device="cpu", seed=1)
    assert len(ds) == 10 and len(loader) == 3
    sizes = [d_[0].shape[0] for d_ in loader]
    assert sizes == [4, 4, 2]                                    # DataLoader(drop_last=False)
    assert all(d_[0].shape[1:] == (3, 8, 8) for d_ in loader)


def test_lr_schedules_match_torch():
    from infinite_texture_gans_amd.train import lr_factor
    p = torch.nn.Parameter(torch.zeros(1))
    for kind, mk in (("exp", lambda o: torch.optim.lr_scheduler.ExponentialLR(o, gamma=0.99)),
                     ("step", lambda o: torch.optim.lr_scheduler.MultiStepLR(o, milestones=[40, 80, 120], gamma=0.5))):
        opt = torch.optim.Adam([p], lr=2e-4)
        sch = mk(opt)
        for epoch in range(1, 131):
            opt.step()
            sch.step()
            assert abs(opt.param_groups[0]["lr"] - 2e-4 * lr_factor(kind, epoch)) < 1e-12, (kind, epoch)
    assert lr_factor(None, 7) == 1.0
