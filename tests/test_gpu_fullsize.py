"""GPU parity at BASELINE config 1's FULL layer sizes through size-independent properties (the oracle
would need minutes per layer here): the three kernels of a convolution are mutually adjoint,

    <dy, conv(x; w)>  ==  <dgrad(dy; w), x>  ==  <wgrad(x, dy), w>      (bias-free, linear activation)

and the forward is linear in x.  Together with the small-size oracle comparisons of test_gpu_ops.py
(same kernels, same code paths: split-K, parity classes, halo tiles, taps-as-rows, replicate fold) this
pins the full-size launches - tile plans, 32-bit offsets, persistent grids - that the small cases cannot reach."""
import pytest
import torch

pytestmark = pytest.mark.gpu
cuda = torch.device("cuda")

# name, n, (gh, gw), patch, cin, cout, k, stride, pad, mode   -- shapes of SURVEY.md section 8 (a2, a10), batch 8
LAYERS = [
    ("D0_fake_3_64_s2", 8, (3, 3), 128, 3, 64, 4, 2, 1, "zero"),
    ("D1_fake_64_128_s2", 8, (1, 1), 192, 64, 128, 4, 2, 1, "zero"),
    ("D3_fake_256_512", 8, (1, 1), 48, 256, 512, 4, 1, 1, "zero"),
    ("D4_fake_512_1", 8, (1, 1), 47, 512, 1, 4, 1, 1, "zero"),
    ("G_b1_416_416_P4", 8, (3, 3), 4, 416, 416, 3, 1, 1, "rep"),
    ("G_b4c1_104_52_P32", 8, (3, 3), 32, 104, 52, 3, 1, 1, "rep"),
    ("G_b6c1_26_13_P128", 8, (3, 3), 128, 26, 13, 3, 1, 1, "rep"),
    ("G_b6c2_13_13_P128", 8, (3, 3), 128, 13, 13, 3, 1, 1, "rep"),
    ("G_final_13_3_P128", 8, (3, 3), 128, 13, 3, 3, 1, 1, "rep"),
]


def dot(a, b):
    return float((a.detach().double() * b.detach().double()).sum())


@pytest.mark.parametrize("prec,tol", [("f32", 2e-5), ("bf16", 2e-2)])
@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
def test_conv_kernels_are_mutually_adjoint_at_full_size(layer, prec, tol):
    from infinite_texture_gans_amd import ops
    name, n, (gh, gw), p, ci, co, k, s, pad, mode = layer
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(n, gh, gw, p, p, ops.ld_for(ci), device=cuda, generator=g)
    x[..., ci:] = 0
    x2 = torch.randn_like(x)
    x2[..., ci:] = 0
    w = torch.randn(co, ci, k, k, device=cuda, generator=g) / (ci * k * k) ** 0.5
    pm = ops.PAD_REPLICATE if mode == "rep" else ops.PAD_ZERO
    og = (gh, gw) if k == 3 else (1, 1)

    def conv(xt, wt):
        return ops.conv(ops.GT(xt, ci), wt, None, k, k, s, pad, pm, out_grid=og).t

    with ops.mfma_precision(prec):
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y = conv(xr, wr)
        dy = torch.randn(y.shape, device=cuda, generator=g)
        dy[..., co:] = 0
        gx, gw_ = torch.autograd.grad(y, (xr, wr), dy)
        lhs = dot(dy, y)
        assert abs(dot(gx, x) - lhs) <= tol * (dot(dy, dy) * dot(y, y)) ** 0.5, (name, "dgrad")
        assert abs(dot(gw_, w) - lhs) <= tol * (dot(dy, dy) * dot(y, y)) ** 0.5, (name, "wgrad")
        if prec == "f32":      # linearity of the forward (bf16 rounds the operands, so only for the fp32 path)
            with torch.no_grad():
                y2, y12 = conv(x2, w), conv(0.5 * x - 2.0 * x2, w)
            err = float((y12 - (0.5 * y.detach() - 2.0 * y2)).norm() / y12.norm())
            assert err < 1e-5, (name, err)
    assert torch.isfinite(y).all() and torch.isfinite(gx).all() and torch.isfinite(gw_).all()


@pytest.mark.parametrize("c,p", [(13, 128), (104, 16)])
def test_fused_local_padding_equals_explicit_padder_at_full_size(c, p):
    """The halo resolved inside the conv loader (training path) == LocalPadder materialised by the standalone
    NHWC kernel followed by a valid conv per padded patch, on the 8 x (3x3) patch grid of config 1."""
    from infinite_texture_gans_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    ld = ops.ld_for(c)
    x = torch.randn(8, 3, 3, p, p, ld, device=cuda, generator=g)
    x[..., c:] = 0
    w = torch.randn(c, c, 3, 3, device=cuda, generator=g) / (9 * c) ** 0.5
    b = torch.randn(c, device=cuda, generator=g)
    with torch.no_grad():
        fused = ops.conv(ops.GT(x, c), w, b, 3, 3, 1, 1, ops.PAD_REPLICATE).t
        padded = ops.local_pad_grid(ops.GT(x, c), ops.PAD_REPLICATE).t          # (8,3,3,p+2,p+2,ld)
        per_patch = ops.conv(ops.GT(padded.reshape(72, 1, 1, p + 2, p + 2, ld), c), w, b, 3, 3, 1, 0, ops.PAD_ZERO).t
    ref = per_patch.reshape(8, 3, 3, p, p, ld)
    assert float((fused - ref).norm() / ref.norm()) < 2e-6


def test_batchnorm_invariants_at_full_size():
    """Training-mode BatchNorm over all 72 patches of (13, 128, 128): normalised output has per-channel mean
    beta and variance gamma^2, and the input gradient is orthogonal to 1 and to x-hat in every channel."""
    from infinite_texture_gans_amd import ops
    g = torch.Generator(device="cuda").manual_seed(4)
    c, ld = 13, 16
    x = (torch.randn(8, 3, 3, 128, 128, ld, device=cuda, generator=g) * 3 + 1.5)
    x[..., c:] = 0
    x.requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(c, device=cuda, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(c, device=cuda, generator=g)).requires_grad_(True)
    rm, rv, nbt = torch.zeros(c, device=cuda), torch.ones(c, device=cuda), torch.zeros((), dtype=torch.int64, device=cuda)
    y = ops.bn_act(ops.GT(x, c), gamma, beta, rm, rv, nbt, training=True).t
    yd = y.detach().double()[..., :c].reshape(-1, c)
    assert float((yd.mean(0) - beta.detach().double()).abs().max()) < 1e-5
    assert float((yd.var(0, unbiased=False) - gamma.detach().double() ** 2).abs().max()) < 1e-4
    dy = torch.randn(y.shape, device=cuda, generator=g)
    dy[..., c:] = 0
    (dx,) = torch.autograd.grad(y, x, dy)
    dxd = dx.double()[..., :c].reshape(-1, c)
    xhat = (yd - beta.detach().double()) / gamma.detach().double()
    scale = float(dxd.abs().sum(0).max())
    assert float(dxd.sum(0).abs().max()) < 1e-5 * scale
    assert float((dxd * xhat).sum(0).abs().max()) < 1e-5 * scale
