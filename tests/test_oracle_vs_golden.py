"""Pins the CPU oracle to fixtures produced by the reference itself
(tests/golden/make_golden.py).  Forward tensors must agree to ~1e-6; the oracle calls
the same torch CPU primitives, so most comparisons are exact or 1-ulp."""
import numpy as np
import pytest
import torch

from oracle import patches as P, nets, step
from helpers import load, parse_flags, cfgs, state, rel_l2, crop_maps

torch.set_num_threads(4)


def test_merge_crop_localpad_match_reference():
    fx = load("patch_ops")
    for i in range(5):
        gh, gw, p, rep = [int(v) for v in fx["lp%d_cfg" % i]]
        outer = "replicate" if rep else "constant"
        x = torch.from_numpy(fx["lp%d_x" % i]).requires_grad_(True)
        assert torch.equal(P.merge(x.detach(), gh, gw), torch.from_numpy(fx["lp%d_merged" % i]))
        assert torch.equal(P.merge_loops(x.detach(), gh, gw), torch.from_numpy(fx["lp%d_merged" % i]))
        for loops in (False, True):
            y = P.local_pad(x, gh, gw, outer, loops=loops)
            assert torch.equal(y.detach(), torch.from_numpy(fx["lp%d_y" % i]))
        y = P.local_pad(x, gh, gw, outer)
        (dx,) = torch.autograd.grad(y, x, torch.from_numpy(fx["lp%d_dy" % i]))
        assert rel_l2(dx, fx["lp%d_dx" % i]) < 1e-6
    z = torch.from_numpy(fx["start_z"])
    assert torch.equal(P.local_pad(z, 3, 3, merged_input=True), torch.from_numpy(fx["start_y"]))
    assert torch.equal(P.crop(torch.from_numpy(fx["crop_img"]), 14, 14, 8), torch.from_numpy(fx["crop_out"]))
    assert torch.equal(P.crop_loops(torch.from_numpy(fx["crop_img"]), 14, 14, 8), torch.from_numpy(fx["crop_out"]))


@pytest.mark.parametrize("tag", ["bn_nl4", "bn_nl6_const", "bn_nl5_att"])
def test_forward_matches_reference(tag):
    fx = load("fwd_" + tag)
    a = parse_flags(fx["argv"])
    gcfg, dcfg = cfgs(a)
    gsd, dsd = state(fx, "G0/"), state(fx, "D0/")
    z = torch.from_numpy(fx["z"])
    with torch.no_grad():
        fake = step.g_sample_train(gsd, gcfg, z, None)
        logit = nets.d_forward(dsd, dcfg, fake, training=True)
    assert rel_l2(fake, fx["fake"]) < 2e-6
    assert rel_l2(logit, fx["d_fake"]) < 5e-6
    for k, v in state(fx, "G1/").items():   # BN running stats / counters after the forward
        assert rel_l2(gsd[k].double(), v.double()) < 1e-5, k
    for k, v in state(fx, "D1/").items():   # SN u, v after one power iteration
        assert rel_l2(dsd[k].double(), v.double()) < 1e-5, k


def fixture_latents(fx, a, gcfg, s):
    """(z, maps) of train step ``s`` as train_step takes them: tensors for disc_iters == 1, lists otherwise."""
    def one(sfx):
        maps = None
        if gcfg.type_norm == "SSM":
            maps = crop_maps(gcfg, [torch.from_numpy(fx["map%s_%d" % (sfx, i)]) for i in range(gcfg.n_layers_G)])
        return torch.from_numpy(fx["z" + sfx]), maps
    if a["disc_iters"] == 1:
        return one("%d" % s)
    zs, ms = zip(*[one("%d_%d" % (s, d)) for d in range(a["disc_iters"])])
    return list(zs), list(ms)


def _run_train(tag, loops=False, steps=None):
    fx = load("train_" + tag)
    a = parse_flags(fx["argv"])
    gcfg, dcfg = cfgs(a)
    gsd = step.as_leaf_params(state(fx, "G0/"))
    dsd = step.as_leaf_params(state(fx, "D0/"))
    esd = {k: v.detach().clone() for k, v in gsd.items()} if a["ema"] else None
    optD = step.Adam([dsd[k] for k in step.trainable(dsd)])
    optG = step.Adam([gsd[k] for k in step.trainable(gsd)])
    outs = []
    for s in range(int(fx["steps"]) if steps is None else steps):
        z, maps = fixture_latents(fx, a, gcfg, s)
        r = step.train_step(gsd, dsd, gcfg, dcfg, optG, optD, torch.from_numpy(fx["real_x%d" % s]), z, maps,
                            smooth=a["smooth"], loops=loops, ema_sd=esd, ema_decay=a["ema_decay"])
        outs.append(r)
    if esd is not None:
        gsd["__ema__"] = esd
    return fx, gsd, dsd, outs


ZERO_GRAD_G = ("mlp_shared.0.bias", "embed.bias", "attention.phi.bias", "attention.g.bias", "attention.o.bias")


def zero_grad_bias(k):
    """G parameters whose gradient is mathematically zero."""
    return (k.endswith("bias") and "conv" in k and k != "final.conv.bias") or any(t in k for t in ZERO_GRAD_G)


@pytest.mark.parametrize("tag", ["bn_nl4_sn", "ssm_nl4", "bn_nl5_att", "bn_nl4_g44", "bn_nl4_di2_ema", "bn_nl4_nosn",
                                 "bn_nl4_zeros"])
def test_train_step_matches_reference(tag):
    fx, gsd, dsd, outs = _run_train(tag)
    esd = gsd.pop("__ema__", None)
    for s, r in enumerate(outs):
        want = fx["loss%d" % s]
        got = np.array([r["d_loss_real"], r["d_loss_fake"], r["g_loss"]])
        assert np.allclose(got, want, rtol=2e-5, atol=1e-6), (s, got, want)
        if "dloss%d" % s in fx:      # every D iteration of --disc_iters > 1
            assert np.allclose(r["d_losses"], fx["dloss%d" % s], rtol=2e-5, atol=1e-6), (s, r["d_losses"])
    assert rel_l2(outs[-1]["fake"], fx["fake_last"]) < 1e-4
    # post-step parameters, BN buffers and SN vectors.  Conv biases that feed a BatchNorm
    # have mathematically zero gradient: with Adam(beta1=0) their update is
    # sign-of-noise * lr, so they are compared only to within the 2*lr*steps they can drift.
    steps = int(fx["steps"])
    for name, sd in (("G1/", gsd), ("D1/", dsd), ("E1/", esd)):
        if sd is None:
            continue
        for k, v in state(fx, name).items():
            got = sd[k].detach().double()
            if name in ("G1/", "E1/") and zero_grad_bias(k):
                assert (got - v.double()).abs().max() <= 2 * 2e-4 * steps + 1e-7, k
                continue
            if v.dtype == torch.int64:
                assert torch.equal(sd[k], v), (k, sd[k], v)       # incl. the EMA's truncated counters
                continue
            assert rel_l2(got, v.double()) < 2e-3, (k, rel_l2(got, v.double()))


@pytest.mark.parametrize("tag", ["bn_nl4_sn", "ssm_nl4", "bn_nl5_att", "bn_nl4_g44", "bn_nl4_nosn"])
def test_first_step_generator_and_discriminator_gradients_match_reference(tag):
    """Gradient MAGNITUDES (the post-Adam parameters of a beta1=0 first step only carry signs)."""
    fx = load("train_" + tag)
    a = parse_flags(fx["argv"])
    gcfg, dcfg = cfgs(a)
    gsd, dsd = step.as_leaf_params(state(fx, "G0/")), step.as_leaf_params(state(fx, "D0/"))
    optD = step.Adam([dsd[k] for k in step.trainable(dsd)])
    optG = step.Adam([gsd[k] for k in step.trainable(gsd)])
    z, maps = fixture_latents(fx, a, gcfg, 0)
    lt = 0.9 if a["smooth"] else 1.0
    _, _, fake, _, _ = step.d_step(gsd, dsd, gcfg, dcfg, optD, torch.from_numpy(fx["real_x0"]), z, maps, lt)
    for k in step.trainable(dsd):       # D's gradients of the D step (the G step adds to them afterwards)
        assert rel_l2(dsd[k].grad, fx["gradD0/" + k]) < 1e-4, k
    step.g_step(gsd, dsd, dcfg, optG, fake, lt)
    for k in step.trainable(gsd):
        want = torch.from_numpy(fx["gradG0/" + k])
        if float(want.abs().max()) < 1e-6:      # rounding noise around a mathematically zero gradient (F11)
            assert zero_grad_bias(k), k
            continue
        assert rel_l2(gsd[k].grad, want) < 1e-4, (k, rel_l2(gsd[k].grad, want))


@pytest.mark.parametrize("tag", ["bn_nl4_sn", "ssm_nl4", "bn_nl4_g44"])
def test_sampler_rng_order_matches_reference(tag):
    """z first, then the SSM maps of layer 0..nl-1, from the global CPU generator (reference utils.py:503-519);
    the fixture recorded the reference sampler's draws under the same seed."""
    fx = load("train_" + tag)
    a = parse_flags(fx["argv"])
    gcfg, _ = cfgs(a)
    seed = {"bn_nl4_sn": 201, "ssm_nl4": 202, "bn_nl4_g44": 204}[tag] + 100
    torch.manual_seed(seed)
    z, maps = step.sample_latents(gcfg, a["num_images"])
    assert torch.equal(z, torch.from_numpy(fx["z0"]))
    if gcfg.type_norm == "SSM":
        want = crop_maps(gcfg, [torch.from_numpy(fx["map0_%d" % i]) for i in range(gcfg.n_layers_G)])
        for m, w in zip(maps, want):
            assert torch.equal(m, w)


def test_zeros_mode_sampler_and_tiling_match_reference():
    """padding_mode='zeros' baseline: sample_from_gen plain and with --tiles (reference utils.py:530-575, 401-470)."""
    fx = load("infer_bn_nl4_zeros_tiles")
    a = parse_flags(fx["argv"])
    gcfg, _ = cfgs(a)
    gsd = state(fx, "G0/")
    z = torch.from_numpy(fx["z"])
    assert rel_l2(step.sample_zeros(gsd, gcfg, z), fx["image"]) < 2e-6
    assert rel_l2(step.sample_zeros(gsd, gcfg, z, tiles=True), fx["image_tiles"]) < 2e-6


def test_loop_faithful_padder_gives_same_step():
    fx, gsd, dsd, outs = _run_train("bn_nl4_sn", loops=True)
    want = fx["loss%d" % (len(outs) - 1)]
    r = outs[-1]
    assert np.allclose([r["d_loss_real"], r["d_loss_fake"], r["g_loss"]], want, rtol=2e-5, atol=1e-6)


def test_first_step_grads_match_reference():
    fx = load("train_bn_nl4_sn")
    a = parse_flags(fx["argv"])
    gcfg, dcfg = cfgs(a)
    gsd = step.as_leaf_params(state(fx, "G0/"))
    dsd = step.as_leaf_params(state(fx, "D0/"))
    real_x, z = torch.from_numpy(fx["real_x0"]), torch.from_numpy(fx["z0"])
    step.bce_logits(nets.d_forward(dsd, dcfg, real_x), 0.9).backward()
    fake = step.g_sample_train(gsd, gcfg, z, None)
    step.bce_logits(nets.d_forward(dsd, dcfg, fake.detach()), 0.0).backward()
    for k in step.trainable(dsd):
        assert rel_l2(dsd[k].grad, fx["gradD0/" + k]) < 1e-4, k


@pytest.mark.parametrize("tag", ["bn_nl4", "ssm_nl4", "bn_nl4_att"])
def test_streamed_inference_matches_reference(tag):
    fx = load("infer_" + tag)
    a = parse_flags(fx["argv"])
    gcfg, _ = cfgs(a)
    gsd = state(fx, "G0/")
    out_h, out_w = [int(v) for v in fx["out_hw"]]
    zf = torch.from_numpy(fx["z_full"])
    maps = None
    if gcfg.type_norm == "SSM":
        maps = [torch.from_numpy(fx["map_full%d" % i]) for i in range(gcfg.n_layers_G)]
    img = step.infer_streamed(gsd, gcfg, zf, maps, out_h, out_w)
    assert img.shape == fx["image"].shape
    assert rel_l2(img, fx["image"]) < 2e-6
    if not gcfg.attention:
        one = step.infer_oneshot(gsd, gcfg, zf, maps, out_h, out_w)
        assert rel_l2(one, fx["image"]) < 2e-6   # streamed == one-shot grid


def test_activation_replay_hook_reproduces_the_unhooked_step():
    """oracle.nets.ACT_RECORD / ACT_REPLAY (the hook behind the flip-free gradient comparison of tests/test_gpu_fullsize.py): a
    step whose LeakyReLUs take their branch from the masks recorded in a plain run of the same step is that run, bit for bit
    (losses and every gradient of G and D); the masks are consumed in order, all of them; a flipped mask changes the result."""
    fx = load("train_bn_nl4_sn")
    a = parse_flags(fx["argv"])
    gcfg, dcfg = cfgs(a)
    real_x, z = torch.from_numpy(fx["real_x0"]), torch.from_numpy(fx["z0"])

    def run(replay=None, record=None):
        gsd = step.as_leaf_params(state(fx, "G0/"))
        dsd = step.as_leaf_params(state(fx, "D0/"))
        optD = step.Adam([dsd[k] for k in step.trainable(dsd)])
        optG = step.Adam([gsd[k] for k in step.trainable(gsd)])
        nets.ACT_REPLAY, nets.ACT_RECORD = replay, record
        try:
            r = step.train_step(gsd, dsd, gcfg, dcfg, optG, optD, real_x, z, None, smooth=True)
        finally:
            left = None if replay is None else len(replay)
            nets.ACT_REPLAY = nets.ACT_RECORD = None
        return r, {k: gsd[k].grad.clone() for k in step.trainable(gsd)}, left

    masks = []
    r0, g0, _ = run(record=masks)
    n_d = len(nets.d_strides(dcfg.n_layers_D)) - 1
    assert len(masks) == 3 * n_d + 2 * gcfg.n_layers_G + 1          # three D passes, two per block + the final BatchNorm
    r1, g1, left = run(replay=[m.clone() for m in masks])
    assert left == 0
    assert (r0["d_loss_real"], r0["d_loss_fake"], r0["g_loss"]) == (r1["d_loss_real"], r1["d_loss_fake"], r1["g_loss"])
    assert all(torch.equal(g0[k], g1[k]) for k in g0) and all(torch.equal(r0["gradD"][k], r1["gradD"][k]) for k in r0["gradD"])
    flipped = [m.clone() for m in masks]
    flipped[n_d + 1] = ~flipped[n_d + 1]                                 # an activation inside G's first block
    r2, g2, _ = run(replay=flipped)
    assert any(not torch.equal(g0[k], g2[k]) for k in g0)


def test_hinge_is_unpinned_but_sane():
    r, f = torch.tensor([[0.5, 2.0]]), torch.tensor([[-2.0, 0.5]])
    dr, df = step.hinge_d(r, f)
    assert abs(float(dr) - 0.25) < 1e-7 and abs(float(df) - 0.75) < 1e-7
    assert abs(float(step.hinge_g(f)) - 0.75) < 1e-7
