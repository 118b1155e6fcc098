"""HIP conv kernels (through the C ABI) vs the CPU oracle ops on identical inputs.

fp32 tolerance: rtol 2e-4, atol 2e-5 * max|ref| (summation order differs; the
MFMA path is an exact fp32 fma chain)."""
import pytest
import torch
import torch.nn.functional as F

from gpu_helpers import assert_close, from_cl, t3, to_cl

pytestmark = pytest.mark.gpu

# (dims, cin, cout, k, stride, pad, spatial, batch)  -- every layer type of G and D
CONV_CASES = [
    (2, 1, 16, 3, 2, 1, (20, 24), 2),      # G down0.unit0 / residual (Cin = 1, scalar gather)
    (2, 16, 16, 3, 1, 1, (20, 24), 3),     # G down0.unit1
    (2, 16, 32, 3, 2, 1, (20, 24), 2),     # G down1.unit0
    (2, 32, 64, 3, 2, 1, (12, 16), 2),
    (2, 32, 32, 3, 1, 1, (24, 20), 2),     # G down1.unit1 (2-D patch weight gradient, 32 x 32 channels)
    (2, 16, 32, 3, 2, 1, (21, 27), 2),     # odd extents under stride 2
    (2, 64, 128, 3, 1, 1, (9, 7), 2),      # G bottom.unit0
    (2, 128, 128, 3, 1, 1, (8, 8), 2),     # G bottom.unit1
    (2, 64, 128, 1, 1, 0, (8, 8), 2),      # G bottom.residual (1x1)
    (2, 1, 1, 3, 1, 1, (16, 20), 2),       # G up0.ru.unit0 (1 -> 1)
    (2, 1, 64, 3, 1, 0, (20, 18), 2),      # D conv1
    (2, 64, 128, 3, 1, 0, (20, 18), 2),    # D conv2
    (2, 128, 256, 4, 2, 0, (18, 16), 2),   # D conv3
    (2, 256, 256, 4, 2, 0, (15, 13), 2),   # D conv4 (odd input: uneven phases in dgrad)
    (2, 64, 128, 3, 1, 0, (126, 126), 1),   # D conv2 at the C1 size
    (2, 128, 256, 4, 2, 0, (124, 124), 1),  # D conv3 at the C1 size: many m-tiles x 4 phases x 64-wide tiles in dgrad
    (2, 256, 256, 4, 2, 0, (61, 61), 1),    # D conv4 at the C1 size (weights > 3 MiB: phase-outer block order)
    (3, 1, 16, 3, 2, 1, (8, 10, 12), 2),   # 3-D variants (reference's true shape)
    (3, 16, 32, 3, 2, 1, (8, 10, 12), 1),
    (3, 32, 32, 3, 1, 1, (6, 5, 7), 2),
    (3, 16, 16, 3, 1, 1, (5, 12, 11), 2),  # 3-D patch kernel (16 -> 16): ragged 2x8x8 tiles, padding on every side
    (3, 16, 16, 3, 1, 0, (6, 10, 18), 1),  # ... pad-free
    (3, 64, 128, 3, 1, 0, (6, 6, 6), 1),
    (3, 128, 256, 4, 2, 0, (8, 8, 10), 1),
    (3, 256, 512, 3, 1, 0, (5, 5, 5), 2),  # variant-B D conv4
    (3, 512, 512, 3, 2, 1, (4, 4, 4), 1),  # generator_test.py's 7-level U-Net: levels 5-6 on 2^3 voxels ...
    (3, 512, 512, 3, 1, 1, (2, 2, 2), 1),  # ... its bottom layer
    (3, 256, 512, 3, 2, 1, (8, 8, 8), 1),
]

CONVT_CASES = [
    (2, 192, 32, (6, 5), 2),
    (2, 64, 16, (10, 12), 2),
    (2, 32, 1, (12, 10), 3),
    (2, 16, 1, (7, 9), 2),                 # quad kernel, 4 lanes per pixel, odd extents
    (2, 64, 1, (5, 6), 2),                 # 16 lanes per pixel
    (3, 192, 32, (3, 4, 5), 1),
    (3, 32, 1, (5, 4, 6), 2),              # octet kernel of ConvTranspose3d(C -> 1): 8 lanes per input voxel
    (3, 64, 1, (3, 4, 5), 1),              # ... 16 lanes
    (3, 1024, 512, (2, 2, 2), 1),          # 7-level U-Net: deepest up-convolution (cat of 512 + 512)
    (3, 512, 128, (4, 4, 4), 1),
]


_MIN_BLOCKS = [0]       # the geometry's big-tile threshold for the geometries this module builds (fixture `dma_form`)


def _geom(dims, n, cin, cout, k, s, p, spatial, transposed=False):
    from mpgan_amd.ops import ConvGeom
    return ConvGeom(n, t3(spatial, dims, 1), cin, cout, t3(k, dims, 1), t3(s, dims, 1), t3(p, dims, 0),
                    transposed, t3(s - 1, dims, 0) if transposed else (0, 0, 0), min_blocks=_MIN_BLOCKS[0])


def _conv(dims):
    return F.conv2d if dims == 2 else F.conv3d


def _convt(dims):
    return F.conv_transpose2d if dims == 2 else F.conv_transpose3d


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "d{}_{}to{}_k{}s{}p{}".format(*c[:6]))
def test_conv_forward_dgrad_wgrad(case):
    from mpgan_amd import ops
    dims, cin, cout, k, s, p, spatial, n = case
    gen = torch.Generator().manual_seed(1000 + cin * 7 + cout)
    x = (torch.rand(n, cin, *spatial, generator=gen) * 2 - 1).requires_grad_(True)
    w = ((torch.rand(cout, cin, *([k] * dims), generator=gen) * 2 - 1) / (cin * k ** dims) ** 0.5).requires_grad_(True)
    b = (torch.rand(cout, generator=gen) - 0.5).requires_grad_(True)
    y_ref = _conv(dims)(x, w, b, stride=s, padding=p)
    gy = torch.rand(y_ref.shape, generator=gen) * 2 - 1
    y_ref.backward(gy)

    g = _geom(dims, n, cin, cout, k, s, p, spatial)
    assert g.out_dhw[3 - dims:] == tuple(y_ref.shape[2:])
    xc, wc = to_cl(x.detach()), w.detach().cuda()
    y = torch.empty(n, *g.out_dhw, cout, device="cuda")
    ops.conv_forward(g, xc, ops.pack_weight(wc), b.detach().cuda(), y)
    assert_close(from_cl(y, dims), y_ref, what="forward")

    dx = torch.full((n, *g.in_dhw, cin), float("nan"), device="cuda")
    ops.conv_backward_data(g, to_cl(gy), ops.pack_weight(wc, for_dgrad=True), dx)
    assert_close(from_cl(dx, dims), x.grad, what="dgrad")

    ws = torch.empty(ops.conv_wgrad_workspace(g) // 4, device="cuda")
    dw = torch.full_like(wc, float("nan"))
    db = torch.full((cout,), float("nan"), device="cuda")
    ops.conv_backward_weight(g, xc, to_cl(gy), dw, ws, dbias=db)
    assert_close(dw.cpu(), w.grad, what="wgrad")
    assert_close(db.cpu(), b.grad, what="fused bias grad")
    # accumulate (beta = 1) doubles it
    ops.conv_backward_weight(g, xc, to_cl(gy), dw, ws, beta=1.0, dbias=db)
    assert_close(dw.cpu(), 2 * w.grad, what="wgrad beta=1")
    assert_close(db.cpu(), 2 * b.grad, what="fused bias grad beta=1")


@pytest.mark.parametrize("case", CONVT_CASES, ids=lambda c: "d{}_{}to{}".format(*c[:3]))
def test_conv_transpose_forward_dgrad_wgrad(case):
    from mpgan_amd import ops
    dims, cin, cout, spatial, n = case
    k, s, p = 3, 2, 1
    gen = torch.Generator().manual_seed(2000 + cin + cout)
    x = (torch.rand(n, cin, *spatial, generator=gen) * 2 - 1).requires_grad_(True)
    w = ((torch.rand(cin, cout, *([k] * dims), generator=gen) * 2 - 1) / (cin * k ** dims) ** 0.5).requires_grad_(True)
    b = (torch.rand(cout, generator=gen) - 0.5).requires_grad_(True)
    y_ref = _convt(dims)(x, w, b, stride=s, padding=p, output_padding=s - 1)
    gy = torch.rand(y_ref.shape, generator=gen) * 2 - 1
    y_ref.backward(gy)

    g = _geom(dims, n, cin, cout, k, s, p, spatial, transposed=True)
    assert g.out_dhw[3 - dims:] == tuple(y_ref.shape[2:])
    xc, wc = to_cl(x.detach()), w.detach().cuda()
    y = torch.full((n, *g.out_dhw, cout), float("nan"), device="cuda")
    ops.conv_forward(g, xc, ops.pack_weight(wc, transposed=True), b.detach().cuda(), y)
    assert_close(from_cl(y, dims), y_ref, what="convT forward")

    dx = torch.full((n, *g.in_dhw, cin), float("nan"), device="cuda")
    ops.conv_backward_data(g, to_cl(gy), ops.pack_weight(wc, transposed=True, for_dgrad=True), dx)
    assert_close(from_cl(dx, dims), x.grad, what="convT dgrad")

    ws = torch.empty(ops.conv_wgrad_workspace(g) // 4, device="cuda")
    dw = torch.full_like(wc, float("nan"))
    ops.conv_backward_weight(g, xc, to_cl(gy), dw, ws)
    assert_close(dw.cpu(), w.grad, what="convT wgrad")


@pytest.mark.parametrize("instance", [False, True])
@pytest.mark.parametrize("cin,cout,k,s,p", [(16, 32, 3, 2, 1), (64, 128, 3, 1, 0), (1, 16, 3, 1, 1)])
def test_conv_prologue_bias_resid_tanh_and_channel_slices(cin, cout, k, s, p, instance):
    """Producer norm + PReLU applied on load; input and output are channel slices
    of wider buffers (concat elision); residual add and tanh in the epilogue."""
    from mpgan_amd import ops
    n, spatial = 3, (14, 10)
    gen = torch.Generator().manual_seed(77 + cin)
    z = torch.rand(n, cin, *spatial, generator=gen) * 2 - 1
    sc = torch.rand(n if instance else 1, cin, generator=gen) + 0.5
    sh = torch.rand(n if instance else 1, cin, generator=gen) - 0.5
    alpha = 0.3
    a = z * sc[:, :, None, None] + sh[:, :, None, None]
    a = torch.where(a > 0, a, alpha * a)
    w = (torch.rand(cout, cin, k, k, generator=gen) * 2 - 1) / (cin * k * k) ** 0.5
    b = torch.rand(cout, generator=gen) - 0.5
    y0 = F.conv2d(a, w, b, stride=s, padding=p)
    r = torch.rand(y0.shape, generator=gen) * 2 - 1
    y_ref = torch.tanh(y0 + r)

    g = _geom(2, n, cin, cout, k, s, p, spatial)
    pad_in = 4 if cin % 4 == 0 else 3
    xbuf = torch.zeros(n, 1, *spatial, cin + 2 * pad_in, device="cuda")
    xbuf[..., pad_in:pad_in + cin] = to_cl(z)
    ybuf = torch.full((n, *g.out_dhw, cout + 8), float("nan"), device="cuda")
    rbuf = torch.zeros(n, *g.out_dhw, cout + 4, device="cuda")
    rbuf[..., 4:] = to_cl(r)
    pro = ops.Prologue(sc.flatten().cuda(), sh.flatten().cuda(), cin if instance else 0, ops.ACT_LEAKY, 1.0,
                       torch.tensor([alpha], device="cuda"))
    ops.conv_forward(g, xbuf[..., pad_in:pad_in + cin], ops.pack_weight(w.cuda()), b.cuda(), ybuf[..., 8:],
                     pro=pro, resid=rbuf[..., 4:], tanh_out=True)
    assert_close(from_cl(ybuf[..., 8:], 2), y_ref, what="fused forward")
    assert torch.isnan(ybuf[..., :8]).all(), "wrote outside its channel slice"

    # weight gradient sees the same prologue on its gathered operand
    gy = torch.rand(y0.shape, generator=gen) * 2 - 1
    a_ = a.clone().requires_grad_(True)
    w_ = w.clone().requires_grad_(True)
    F.conv2d(a_, w_, None, stride=s, padding=p).backward(gy)
    ws = torch.empty(ops.conv_wgrad_workspace(g) // 4, device="cuda")
    dw = torch.empty_like(w, device="cuda")
    ops.conv_backward_weight(g, xbuf[..., pad_in:pad_in + cin], to_cl(gy), dw, ws, pro=pro)
    assert_close(dw.cpu(), w_.grad, what="wgrad with prologue")
    # dgrad with residual accumulate into a slice
    dxbuf = torch.full((n, 1, *spatial, cin + 4), float("nan"), device="cuda")
    res = torch.rand(n, cin, *spatial, generator=gen)
    ops.conv_backward_data(g, to_cl(gy), ops.pack_weight(w.cuda(), for_dgrad=True), dxbuf[..., :cin],
                           resid=to_cl(res))
    assert_close(from_cl(dxbuf[..., :cin], 2), a_.grad + res, what="dgrad + resid")


def test_linear_as_conv_matches_flatten_linear():
    """Variant-B head Linear(512*8^3 -> 64) is a conv whose kernel spans the whole
    (8,8,8) grid; the packed weight realises the NCDHW-flatten permutation."""
    from mpgan_amd import ops
    n, c, sp, o = 5, 32, (4, 4, 4), 64
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(n, c, *sp, generator=gen) * 2 - 1
    lin = torch.nn.Linear(c * 64, o)
    y_ref = lin(x.flatten(1))
    g = ops.ConvGeom(n, sp, c, o, sp, (1, 1, 1), (0, 0, 0))
    y = torch.empty(n, 1, 1, 1, o, device="cuda")
    wconv = lin.weight.detach().reshape(o, c, *sp).cuda()
    ops.conv_forward(g, to_cl(x), ops.pack_weight(wconv), lin.bias.detach().cuda(), y)
    assert_close(y.reshape(n, o).cpu(), y_ref, what="linear as conv")


def test_unsupported_and_invalid_arguments_raise():
    from mpgan_amd import ops
    g = ops.ConvGeom(1, (1, 8, 8), 4, 4, (1, 3, 3), (1, 3, 3), (0, 1, 1))
    x = torch.zeros(1, 1, 8, 8, 4, device="cuda")
    y = torch.zeros(1, *g.out_dhw, 4, device="cuda")
    with pytest.raises(RuntimeError, match="stride"):
        ops.conv_forward(g, x, torch.zeros(4 * 4 * 9, device="cuda"), None, y)
    g2 = ops.ConvGeom(1, (1, 8, 8), 4, 4, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    with pytest.raises(ValueError):
        ops.conv_forward(g2, x.cpu(), torch.zeros(144, device="cuda"), None, y)


@pytest.mark.parametrize("cin,cout,k,s,p,spatial,transposed", [
    (16, 32, 3, 2, 1, (20, 24), False), (64, 128, 3, 1, 0, (20, 18), False), (256, 256, 4, 2, 0, (15, 13), False),
    (192, 32, 3, 2, 1, (7, 5), True),
    (1, 16, 3, 2, 1, (38, 26), False), (1, 32, 3, 1, 1, (17, 23), False),     # all-channel 1 -> C stencil
    (32, 1, 3, 2, 1, (19, 14), True)])                                         # quad kernel of ConvTranspose2d(C -> 1)
def test_fused_batchnorm_statistics_from_conv_epilogue(cin, cout, k, s, p, spatial, transposed):
    """The conv's own epilogue leaves per-tile (sum, sum^2) rows; finalize turns them
    into the same scale/shift/running stats as F.batch_norm on the conv output."""
    from mpgan_amd import ops
    n = 3
    gen = torch.Generator().manual_seed(123 + cout)
    x = torch.rand(n, cin, *spatial, generator=gen) * 2 - 1
    if transposed:
        w = (torch.rand(cin, cout, k, k, generator=gen) - 0.5) / (cin * k * k) ** 0.5
        b = torch.rand(cout, generator=gen) - 0.5
        z_ref = F.conv_transpose2d(x, w, b, stride=s, padding=p, output_padding=s - 1)
    else:
        w = (torch.rand(cout, cin, k, k, generator=gen) - 0.5) / (cin * k * k) ** 0.5
        b = torch.rand(cout, generator=gen) - 0.5
        z_ref = F.conv2d(x, w, b, stride=s, padding=p)
    gamma, beta = torch.rand(cout, generator=gen) + 0.5, torch.rand(cout, generator=gen) - 0.5
    rm, rv = torch.zeros(cout), torch.ones(cout)
    y_ref = F.batch_norm(z_ref, rm, rv, gamma, beta, True, 0.1, 1e-5)

    g = _geom(2, n, cin, cout, k, s, p, spatial, transposed=transposed)
    rows = ops.conv_stats_rows(g, False)
    assert rows > 0
    part = torch.full(((rows + 32) * 2 * cout,), float("nan"), device="cuda")   # + finalize's fold scratch
    z = torch.empty(n, *g.out_dhw, cout, device="cuda")
    ops.conv_forward(g, to_cl(x), ops.pack_weight(w.cuda(), transposed=transposed), b.cuda(), z, stats_partials=part)
    assert_close(from_cl(z, 2), z_ref, what="conv output")
    assert torch.isfinite(part[:rows * 2 * cout]).all()
    scale, shift, mean, invstd = (torch.empty(cout, device="cuda") for _ in range(4))
    rmd, rvd = torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    P = n * g.out_dhw[0] * g.out_dhw[1] * g.out_dhw[2]
    ops.norm_finalize(part, 1, rows, cout, P, False, gamma.cuda(), beta.cuda(), 1e-5, 0.1, rmd, rvd, nbt, scale, shift,
                      mean, invstd)
    y = from_cl(z, 2) * scale.cpu()[None, :, None, None] + shift.cpu()[None, :, None, None]
    assert_close(y, y_ref, rtol=1e-4, what="normalised output")
    assert_close(rmd.cpu(), rm, what="running_mean")
    assert_close(rvd.cpu(), rv, rtol=1e-4, what="running_var")


# ---- patch kernel (2-D, <= 32 output channels, 16/32/64 gathered channels) ----------------
PATCH_CASES = [
    # cin, cout, k, s, p, spatial, transposed
    (16, 16, 3, 1, 1, (37, 29), False),     # ragged tiles on both axes
    (32, 32, 3, 1, 1, (16, 48), False),
    (64, 32, 2, 1, 0, (11, 19), False),     # even kernel, no padding (3x3 at 64->32 exceeds the LDS budget: K-stepped kernel)
    (16, 32, 3, 2, 1, (26, 34), False),     # stride-2 patch (input stride inside LDS)
    (32, 24, 1, 1, 0, (9, 17), False),      # 1x1, Cout not a power of two
    (64, 16, 3, 2, 1, (13, 9), True),       # transposed: 4 phases, odd extents
    (32, 8, 3, 2, 1, (8, 16), True),
]


@pytest.mark.parametrize("case", PATCH_CASES, ids=lambda c: "{}to{}_k{}s{}p{}_{}".format(c[0], c[1], c[2], c[3], c[4], "T" if c[6] else "F"))
def test_patch_kernel_forward_stats_prologue_and_dgrad(case):
    """Layers the dispatcher hands to gather_patch_kernel: forward with the producer's
    BatchNorm+PReLU prologue and fused statistics, plain forward + residual + tanh, and the
    backward-data gather of a conv whose INPUT has <= 32 channels."""
    import ctypes
    from mpgan_amd import ops
    from mpgan_amd._lib import lib
    cin, cout, k, s, p, spatial, tr = case
    n = 3
    gen = torch.Generator().manual_seed(4242 + cin * 3 + cout)
    z = torch.rand(n, cin, *spatial, generator=gen) * 2 - 1
    sc, sh = torch.rand(cin, generator=gen) + 0.5, torch.rand(cin, generator=gen) - 0.5
    alpha = 0.25
    a = z * sc[None, :, None, None] + sh[None, :, None, None]
    a = torch.where(a > 0, a, alpha * a)
    if tr:
        w = (torch.rand(cin, cout, k, k, generator=gen) * 2 - 1) / (cin * k * k) ** 0.5
    else:
        w = (torch.rand(cout, cin, k, k, generator=gen) * 2 - 1) / (cin * k * k) ** 0.5
    b = torch.rand(cout, generator=gen) - 0.5
    conv = (lambda t: F.conv_transpose2d(t, w, b, stride=s, padding=p, output_padding=s - 1)) if tr else \
           (lambda t: F.conv2d(t, w, b, stride=s, padding=p))
    g = _geom(2, n, cin, cout, k, s, p, spatial, transposed=tr)
    gc = g.c()
    assert lib().mpgan_conv_variant(ctypes.byref(gc), 0, 1) == 16, "expected the patch kernel for this geometry"
    assert lib().mpgan_conv_variant(ctypes.byref(gc), 0, 0) == (17 if tr else 16)     # merged phases without a prologue
    wp = ops.pack_weight(w.cuda(), transposed=tr)

    # (1) prologue + fused statistics
    y_ref = conv(a)
    rows = ops.conv_stats_rows(g, True)
    assert rows > 0
    part = torch.full(((rows + 32) * 2 * cout,), float("nan"), device="cuda")
    y = torch.full((n, *g.out_dhw, cout), float("nan"), device="cuda")
    pro = ops.Prologue(sc.cuda(), sh.cuda(), 0, ops.ACT_LEAKY, 1.0, torch.tensor([alpha], device="cuda"))
    ops.conv_forward(g, to_cl(z), wp, b.cuda(), y, pro=pro, stats_partials=part)
    assert_close(from_cl(y, 2), y_ref, what="patch forward (prologue)")
    sums = part[:rows * 2 * cout].reshape(rows, 2, cout).double().sum(0).cpu()
    ref64 = y_ref.double()
    assert_close(sums[0].float(), ref64.sum((0, 2, 3)).float(), rtol=1e-4, what="fused sum")
    assert_close(sums[1].float(), (ref64 ** 2).sum((0, 2, 3)).float(), rtol=1e-4, what="fused sum of squares")

    # (2) no prologue, residual + tanh, output into a channel slice
    y0 = conv(z)
    r = torch.rand(y0.shape, generator=gen) * 2 - 1
    ybuf = torch.full((n, *g.out_dhw, cout + 8), float("nan"), device="cuda")
    ops.conv_forward(g, to_cl(z), wp, b.cuda(), ybuf[..., 4:4 + cout], resid=to_cl(r), tanh_out=True)
    assert_close(from_cl(ybuf[..., 4:4 + cout], 2), torch.tanh(y0 + r), what="patch forward (resid+tanh)")
    assert torch.isnan(ybuf[..., :4]).all() and torch.isnan(ybuf[..., 4 + cout:]).all()

    # (2b) no prologue + fused statistics (for transposed layers: the merged-phase form, one partial row per
    #      (tile, phase))
    rows0 = ops.conv_stats_rows(g, False)
    part0 = torch.full(((rows0 + 32) * 2 * cout,), float("nan"), device="cuda")
    y2 = torch.full((n, *g.out_dhw, cout), float("nan"), device="cuda")
    ops.conv_forward(g, to_cl(z), wp, b.cuda(), y2, stats_partials=part0)
    assert_close(from_cl(y2, 2), y0, what="patch forward (stats, no prologue)")
    sums0 = part0[:rows0 * 2 * cout].reshape(rows0, 2, cout).double().sum(0).cpu()
    assert_close(sums0[0].float(), y0.double().sum((0, 2, 3)).float(), rtol=1e-4, what="fused sum (no prologue)")
    assert_close(sums0[1].float(), (y0.double() ** 2).sum((0, 2, 3)).float(), rtol=1e-4,
                 what="fused sum of squares (no prologue)")

    # (3) backward-data of the mirrored layer (gathers `cin` channels of dy, lands on `cout`)
    if tr:
        # dgrad of a ConvNd(cout -> cin, stride s) is this transposed gather
        wf = ((torch.rand(cin, cout, k, k, generator=gen) * 2 - 1) / (cout * k * k) ** 0.5)
        gf = _geom(2, n, cout, cin, k, s, p, g.out_dhw[1:])
        if gf.out_dhw[1:] != tuple(spatial):
            return
        xf = torch.rand(n, cout, *g.out_dhw[1:], generator=gen).requires_grad_(True)
        gy = torch.rand(n, cin, *spatial, generator=gen) * 2 - 1
        F.conv2d(xf, wf, None, stride=s, padding=p).backward(gy)
        gfc = gf.c()
        assert lib().mpgan_conv_variant(ctypes.byref(gfc), 1, 0) == 17      # all phases in one block
        dx = torch.full((n, *gf.in_dhw, cout), float("nan"), device="cuda")
        ops.conv_backward_data(gf, to_cl(gy), ops.pack_weight(wf.cuda(), for_dgrad=True), dx)
        assert_close(from_cl(dx, 2), xf.grad, what="patch dgrad (strided conv)")
    elif s == 1:
        wf = ((torch.rand(cin, cout, k, k, generator=gen) * 2 - 1) / (cout * k * k) ** 0.5)   # ConvNd(cout -> cin)
        osp = tuple(v + 2 * p - k + 1 for v in spatial)
        gf = _geom(2, n, cout, cin, k, 1, p, spatial)
        xf = torch.rand(n, cout, *spatial, generator=gen).requires_grad_(True)
        gy = torch.rand(n, cin, *osp, generator=gen) * 2 - 1
        F.conv2d(xf, wf, None, stride=1, padding=p).backward(gy)
        gfc = gf.c()
        assert lib().mpgan_conv_variant(ctypes.byref(gfc), 1, 0) == 16
        dx = torch.full((n, *gf.in_dhw, cout), float("nan"), device="cuda")
        ops.conv_backward_data(gf, to_cl(gy), ops.pack_weight(wf.cuda(), for_dgrad=True), dx)
        assert_close(from_cl(dx, 2), xf.grad, what="patch dgrad (stride 1)")


@pytest.mark.parametrize("cin,cout,k,pro_code", [(128, 128, 3, 1), (64, 128, 3, 0), (64, 64, 3, 1)])
def test_small_grid_layers_split_k_inside_the_block(cin, cout, k, pro_code):
    """The U-Net's 32 x 32 levels at config C2/C3's size (bs 16: 128 pixel tiles): about one block per CU, so the
    dispatcher picks the form whose K axis is split over two 4-wave groups per block (variant 2032 / 2064);
    forward with prologue + fused statistics and the backward-data gather must match torch."""
    import ctypes
    from mpgan_amd import ops
    from mpgan_amd._lib import lib
    n, spatial = 16, (32, 32)
    g = _geom(2, n, cin, cout, k, 1, 1, spatial)
    gc = g.c()
    v = lib().mpgan_conv_variant(ctypes.byref(gc), 0, pro_code)
    assert v in (2032, 2064), v
    gen = torch.Generator().manual_seed(900 + cin + cout)
    z = torch.rand(n, cin, *spatial, generator=gen) * 2 - 1
    w = (torch.rand(cout, cin, k, k, generator=gen) * 2 - 1) / (cin * k * k) ** 0.5
    b = torch.rand(cout, generator=gen) - 0.5
    pro = None
    a = z
    if pro_code:
        sc, sh = torch.rand(cin, generator=gen) + 0.5, torch.rand(cin, generator=gen) - 0.5
        a = z * sc[None, :, None, None] + sh[None, :, None, None]
        a = torch.where(a > 0, a, 0.25 * a)
        pro = ops.Prologue(sc.cuda(), sh.cuda(), 0, ops.ACT_LEAKY, 1.0, torch.tensor([0.25], device="cuda"))
    a_ = a.clone().requires_grad_(True)
    y_ref = F.conv2d(a_, w, b, padding=1)
    rows = ops.conv_stats_rows(g, pro_code)
    part = torch.full(((rows + 32) * 2 * cout,), float("nan"), device="cuda")
    y = torch.full((n, *g.out_dhw, cout), float("nan"), device="cuda")
    ops.conv_forward(g, to_cl(z), ops.pack_weight(w.cuda()), b.cuda(), y, pro=pro, stats_partials=part)
    assert_close(from_cl(y, 2), y_ref.detach(), what="forward (K split in block)")
    sums = part[:rows * 2 * cout].reshape(rows, 2, cout).double().sum(0).cpu()
    assert_close(sums[0].float(), y_ref.detach().double().sum((0, 2, 3)).float(), rtol=1e-4, what="fused sum")
    assert_close(sums[1].float(), (y_ref.detach().double() ** 2).sum((0, 2, 3)).float(), rtol=1e-4, what="fused sum of squares")
    gy = torch.rand(y_ref.shape, generator=gen) * 2 - 1
    y_ref.backward(gy)
    assert lib().mpgan_conv_variant(ctypes.byref(gc), 1, 0) in (2032, 2064)
    dx = torch.full((n, *g.in_dhw, cin), float("nan"), device="cuda")
    ops.conv_backward_data(g, to_cl(gy), ops.pack_weight(w.cuda(), for_dgrad=True), dx)
    assert_close(from_cl(dx, 2), a_.grad, what="dgrad (K split in block)")


@pytest.mark.parametrize("cin,c,spatial", [(1, 16, (64, 48)), (16, 32, (40, 56)), (32, 64, (24, 32))])
def test_residual_unit_first_conv_and_residual_conv_as_one_launch(cin, c, spatial):
    """MONAI ResidualUnit(stride 2): `conv.unit0.conv` and `residual` read the same input with the same geometry;
    the engine runs them as ONE conv over their concatenated output channels (ParamStore.register_fused packs
    [W_unit0; W_res] and the two biases) and takes BatchNorm statistics of the FIRST half only
    (mpgan_norm_finalize_strided).  Checked against the two torch convs and F.batch_norm."""
    import torch.nn as nn
    from mpgan_amd import engine, ops
    n = 4
    torch.manual_seed(3 + cin)
    net = nn.ModuleDict({"a": nn.Conv2d(cin, c, 3, 2, 1), "b": nn.Conv2d(cin, c, 3, 2, 1), "bn": nn.BatchNorm2d(c)}).cuda()
    with torch.no_grad():
        net["bn"].weight.uniform_(0.5, 1.5)
        net["bn"].bias.uniform_(-0.5, 0.5)
    store = engine.ParamStore(net)
    fr = store.register_fused(net["a"], net["b"])
    prog = engine.Program()
    store.emit_pack(prog)
    gen = torch.Generator().manual_seed(17)
    x = torch.rand(n, cin, *spatial, generator=gen) * 2 - 1
    g = _geom(2, n, cin, 2 * c, 3, 2, 1, spatial)
    zr = torch.full((n, *g.out_dhw, 2 * c), float("nan"), device="cuda")
    nb = engine.NormBuf(n, c, False, torch.device("cuda"))
    rows = ops.conv_stats_rows(g, 0)
    assert rows > 0
    part = torch.empty((rows + 32) * 2 * 2 * c, device="cuda")
    engine.emit_conv_fwd_norm(prog, g, to_cl(x), store.wp_fused(fr), store.bias_fused(fr), zr, nb, net["bn"], part, c_norm=c)
    prog.run()
    wa, wb = net["a"].weight.detach().cpu(), net["b"].weight.detach().cpu()
    za = F.conv2d(x, wa, net["a"].bias.detach().cpu(), stride=2, padding=1)
    zb = F.conv2d(x, wb, net["b"].bias.detach().cpu(), stride=2, padding=1)
    assert_close(from_cl(zr[..., :c], 2), za, what="first conv (fused launch)")
    assert_close(from_cl(zr[..., c:], 2), zb, what="residual conv (fused launch)")
    rm, rv = torch.zeros(c), torch.ones(c)
    y_ref = F.batch_norm(za, rm, rv, net["bn"].weight.detach().cpu(), net["bn"].bias.detach().cpu(), True, 0.1, 1e-5)
    y = from_cl(zr[..., :c], 2) * nb.scale.cpu()[None, :, None, None] + nb.shift.cpu()[None, :, None, None]
    assert_close(y, y_ref, rtol=1e-4, what="normalised first half")
    assert_close(net["bn"].running_mean.cpu(), rm, what="running_mean")
    assert_close(net["bn"].running_var.cpu(), rv, rtol=1e-4, what="running_var")
    assert int(net["bn"].num_batches_tracked) == 1


@pytest.mark.parametrize("cin,c,k,s,spatial,transposed", [
    (16, 16, 3, 1, (40, 56), False), (32, 32, 3, 1, (24, 40), False), (1, 32, 3, 2, (48, 64), False),
    (64, 128, 3, 1, (16, 16), False), (64, 16, 3, 2, (12, 20), True), (32, 1, 3, 2, (20, 28), True)])
def test_batchnorm_statistics_through_accumulators_and_fold_on_load(cin, c, k, s, spatial, transposed):
    """csrc/norm_fold.h: the producing conv adds its per-block (sum, sum^2) to int64 fixed-point accumulators;
    the FIRST consumer folds them at block start -- here (a) the residual-sum pass, (b) a following 3x3 conv with
    the load prologue where the persistent patch kernel serves it -- publishes scale / shift / mean / invstd and
    advances the running statistics.  Same numbers as F.batch_norm (train) on the conv output; order-independent,
    so two runs agree bit for bit."""
    import torch.nn as nn
    from mpgan_amd import ops
    n = 4
    gen = torch.Generator().manual_seed(31 + cin + c)
    x = torch.rand(n, cin, *spatial, generator=gen) * 2 - 1
    if transposed:
        w = (torch.rand(cin, c, k, k, generator=gen) - 0.5) / (cin * k * k) ** 0.5
        b = torch.rand(c, generator=gen) - 0.5
        z_ref = F.conv_transpose2d(x, w, b, stride=s, padding=1, output_padding=s - 1)
    else:
        w = (torch.rand(c, cin, k, k, generator=gen) - 0.5) / (cin * k * k) ** 0.5
        b = torch.rand(c, generator=gen) - 0.5
        z_ref = F.conv2d(x, w, b, stride=s, padding=1)
    g = _geom(2, n, cin, c, k, s, 1, spatial, transposed=transposed)
    assert ops.conv_acc_supported(g, 0)
    bn = nn.BatchNorm2d(c).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    gamma, beta = bn.weight.detach().cpu(), bn.bias.detach().cpu()
    alpha = 0.25
    y_ref = F.batch_norm(z_ref, torch.zeros(c), torch.ones(c), gamma, beta, True, 0.1, 1e-5)
    a_ref = torch.where(y_ref > 0, y_ref, alpha * y_ref)
    wp = ops.pack_weight(w.cuda(), transposed=transposed)
    slope = torch.tensor([alpha], device="cuda")
    outs = []
    for rep in range(2):
        z = torch.full((n, *g.out_dhw, c), float("nan"), device="cuda")
        acc = ops.acc_buffer(c, "cuda")
        ops.conv_forward_fold(g, to_cl(x), wp, b.cuda(), z, stats_acc=acc)
        assert_close(from_cl(z, 2), z_ref, what="conv output")
        vec = [torch.full((c,), float("nan"), device="cuda") for _ in range(4)]
        bn.running_mean.zero_(); bn.running_var.fill_(1.0); bn.num_batches_tracked.zero_()
        P = n * g.out_dhw[1] * g.out_dhw[2]
        fold = ops.make_fold(acc, ops.ACC_REPLICAS, c, P, bn, *vec)
        pro = ops.Prologue(vec[0], vec[1], 0, ops.ACT_LEAKY, 1.0, slope)
        out = torch.full_like(z, float("nan"))
        r = torch.rand(z_ref.shape, generator=torch.Generator().manual_seed(5)) - 0.5
        ops.norm_act_add_fold(z, pro, fold, to_cl(r), None, out)
        assert_close(from_cl(out, 2), a_ref + r, rtol=1e-4, what="act(bn(z)) + r through the fold")
        y = from_cl(z, 2) * vec[0].cpu()[None, :, None, None] + vec[1].cpu()[None, :, None, None]
        assert_close(y, y_ref, rtol=1e-4, what="published scale/shift")
        rm, rv = torch.zeros(c), torch.ones(c)
        F.batch_norm(z_ref, rm, rv, gamma, beta, True, 0.1, 1e-5)
        assert_close(bn.running_mean.cpu(), rm, what="running_mean")
        assert_close(bn.running_var.cpu(), rv, rtol=1e-4, what="running_var")
        assert int(bn.num_batches_tracked) == 1
        outs.append((acc.clone(), vec[0].clone(), vec[1].clone(), out.clone()))
    for t0, t1 in zip(outs[0], outs[1]):
        assert torch.equal(t0, t1), "accumulator statistics must not depend on block arrival order"
    # (b) the next conv folds on load
    g2 = _geom(2, n, c, c, 3, 1, 1, g.out_dhw[1:])
    if c >= 16 and ops.conv_fold_supported(g2):
        w2 = (torch.rand(c, c, 3, 3, generator=gen) - 0.5) / (c * 9) ** 0.5
        z2_ref = F.conv2d(a_ref, w2, None, padding=1)
        acc = ops.acc_buffer(c, "cuda")
        z = torch.empty(n, *g.out_dhw, c, device="cuda")
        ops.conv_forward_fold(g, to_cl(x), wp, b.cuda(), z, stats_acc=acc)
        vec = [torch.full((c,), float("nan"), device="cuda") for _ in range(4)]
        fold = ops.make_fold(acc, ops.ACC_REPLICAS, c, n * g.out_dhw[1] * g.out_dhw[2], bn, *vec)
        pro = ops.Prologue(vec[0], vec[1], 0, ops.ACT_LEAKY, 1.0, slope)
        acc2 = ops.acc_buffer(c, "cuda")
        z2 = torch.full((n, *g2.out_dhw, c), float("nan"), device="cuda")
        ops.conv_forward_fold(g2, z, ops.pack_weight(w2.cuda()), None, z2, pro=pro, fold=fold, stats_acc=acc2)
        assert_close(from_cl(z2, 2), z2_ref, rtol=5e-4, what="conv with fold-on-load prologue")
        assert torch.isfinite(vec[2]).all() and torch.isfinite(vec[3]).all()
        # ... and its own accumulators hold the sums of its output
        A = acc2.view(ops.ACC_REPLICAS, 4, c).sum(0).double().cpu()
        s1 = A[0] / 256.0 + A[1] / 2.0 ** 56
        s2 = A[2] / 256.0 + A[3] / 2.0 ** 56
        assert_close(s1.float(), z2_ref.double().sum((0, 2, 3)).float(), rtol=1e-4, what="accumulated sum")
        assert_close(s2.float(), (z2_ref.double() ** 2).sum((0, 2, 3)).float(), rtol=1e-4, what="accumulated sum of squares")


@pytest.mark.parametrize("cin,cout,k,s,spatial,n", [(64, 128, 3, 1, (30, 26), 3), (128, 256, 4, 2, (22, 26), 2),
                                                    (256, 256, 4, 2, (13, 11), 2), (64, 72, 3, 1, (9, 9, 10), 1)])
def test_backward_data_with_fused_norm_backward_sums(cin, cout, k, s, spatial, n):
    """mpgan_conv_backward_data_stats: the same dx as the plain launch, bit for bit, and partial rows whose column
    sums equal what mpgan_norm_bwd_reduce forms by re-reading dx and z (BatchNorm + LeakyReLU(0.2) in front of the
    conv's input: the discriminator's Conv -> BN -> LeakyReLU chain, GAN_final.py:167-189)."""
    from mpgan_amd import ops
    dims = len(spatial)
    g = _geom(dims, n, cin, cout, k, s, 0, spatial)
    rows = ops.conv_bwd_stats_rows(g)
    assert rows > 0
    gen = torch.Generator(device="cuda").manual_seed(31 + cin)
    R = lambda *shape: torch.rand(*shape, device="cuda", generator=gen) * 2 - 1
    dy = R(n, *g.out_dhw, cout)
    z = R(n, *g.in_dhw, cin)
    w = R(cout, cin, *([k] * dims)) / (cin * k ** dims) ** 0.5
    wpb = ops.pack_weight(w, for_dgrad=True)
    scale, shift = R(cin) * 0.5 + 1.0, R(cin) * 0.3
    mean, invstd = R(cin) * 0.2, R(cin) * 0.3 + 1.0
    dx0 = torch.empty_like(z)
    ops.conv_backward_data(g, dy, wpb, dx0)
    dx1 = torch.full_like(z, float("nan"))
    part = torch.full((rows * 3 * cin,), float("nan"), device="cuda")
    assert ops.conv_backward_data_stats(g, dy, wpb, dx1, z, scale, shift, mean, invstd, ops.ACT_LEAKY, 0.2, part) == rows
    assert torch.equal(dx0, dx1)
    got = part.view(rows, 3, cin).double().sum(0).cpu()
    y = z.double() * scale.double() + shift.double()
    zh = (z.double() - mean.double()) * invstd.double()
    neg = y < 0
    gy = torch.where(neg, dx0.double() * 0.2, dx0.double())
    want = torch.stack([gy.reshape(-1, cin).sum(0), (gy * zh).reshape(-1, cin).sum(0),
                        torch.where(neg, dx0.double() * y, torch.zeros_like(y)).reshape(-1, cin).sum(0)]).cpu()
    scale_ = (gy.abs().reshape(-1, cin).sum(0).max().item())
    assert (got - want).abs().max().item() <= 2e-5 * scale_, ((got - want).abs().max().item(), scale_)


@pytest.mark.parametrize("spatial,n,p", [((5, 12, 11), 2, 1), ((6, 10, 18), 1, 0)])
def test_patch3d_prologue_residual_statistics_and_slices(spatial, n, p):
    """The 3-D 16 -> 16 patch kernel with everything the U-Net hangs on it: producer BatchNorm + PReLU (device
    slope) applied on load, zero padding of the ACTIVATED tensor, bias, residual add, input / output / residual as
    channel slices of wider buffers, fused statistics rows -- and its backward-data gather with a residual."""
    import ctypes
    from mpgan_amd import ops
    from mpgan_amd._lib import lib
    cin = cout = 16
    g = _geom(3, n, cin, cout, 3, 1, p, spatial)
    gc = g.c()
    assert lib().mpgan_conv_variant(ctypes.byref(gc), 0, 1) == 18 and lib().mpgan_conv_variant(ctypes.byref(gc), 1, 0) == 18
    gen = torch.Generator().manual_seed(99)
    z = torch.rand(n, cin, *spatial, generator=gen) * 2 - 1
    sc, sh = torch.rand(cin, generator=gen) + 0.5, torch.rand(cin, generator=gen) - 0.5
    alpha = 0.3
    a = z * sc[None, :, None, None, None] + sh[None, :, None, None, None]
    a = torch.where(a > 0, a, alpha * a)
    w = (torch.rand(cout, cin, 3, 3, 3, generator=gen) - 0.5) / (cin * 27) ** 0.5
    b = torch.rand(cout, generator=gen) - 0.5
    y_ref = F.conv3d(a, w, b, padding=p)
    res = torch.rand(y_ref.shape, generator=gen) - 0.5
    # operands as channel slices of wider buffers
    zbuf = torch.full((n, *spatial, cin + 8), float("nan"), device="cuda")
    zbuf[..., 4:4 + cin] = to_cl(z)
    ybuf = torch.full((n, *g.out_dhw, cout + 16), float("nan"), device="cuda")
    rbuf = torch.zeros(n, *g.out_dhw, cout + 4, device="cuda")
    rbuf[..., :cout] = to_cl(res)
    pro = ops.Prologue(sc.cuda(), sh.cuda(), 0, ops.ACT_LEAKY, 1.0, torch.tensor([alpha], device="cuda"))
    rows = ops.conv_stats_rows(g, 1)
    assert rows == n * ((g.out_dhw[0] + 1) // 2) * ((g.out_dhw[1] + 7) // 8) * ((g.out_dhw[2] + 7) // 8)
    part = torch.full(((rows + 32) * 2 * cout,), float("nan"), device="cuda")
    ops.conv_forward(g, zbuf[..., 4:4 + cin], ops.pack_weight(w.cuda()), b.cuda(), ybuf[..., 8:8 + cout], pro=pro,
                     stats_partials=part)
    assert_close(from_cl(ybuf[..., 8:8 + cout], 3), y_ref, what="conv output")
    assert torch.isnan(ybuf[..., :8]).all() and torch.isnan(ybuf[..., 8 + cout:]).all()      # nothing written outside
    got = part[:rows * 2 * cout].view(rows, 2, cout).double().sum(0).cpu()
    want = torch.stack([y_ref.double().transpose(0, 1).reshape(cout, -1).sum(1),
                        (y_ref.double() ** 2).transpose(0, 1).reshape(cout, -1).sum(1)])
    assert (got - want).abs().max().item() <= 1e-4 * want.abs().max().item()
    # residual in the epilogue (statistics are not asked for together with it)
    y2 = torch.empty(n, *g.out_dhw, cout, device="cuda")
    ops.conv_forward(g, zbuf[..., 4:4 + cin], ops.pack_weight(w.cuda()), b.cuda(), y2, pro=pro, resid=rbuf[..., :cout])
    assert_close(from_cl(y2, 3), y_ref + res, what="conv output + residual")
    # weight gradient (the patch form's pixel-contracting kernel): prologue on x, bias gradient, accumulate
    ar = a.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    gy = torch.rand(y_ref.shape, generator=gen) * 2 - 1
    F.conv3d(ar, wr, br, padding=p).backward(gy)
    dw = torch.ones_like(w).cuda()
    db = torch.ones(cout, device="cuda")
    ws = torch.empty(ops.conv_wgrad_workspace(g) // 4, device="cuda")
    gybuf = torch.zeros(n, *g.out_dhw, cout + 4, device="cuda")
    gybuf[..., :cout] = to_cl(gy)
    ops.conv_backward_weight(g, zbuf[..., 4:4 + cin], gybuf[..., :cout], dw, ws, pro=pro, beta=1.0, dbias=db)
    assert_close(dw.cpu() - 1.0, wr.grad, what="wgrad (patch form)")
    assert_close(db.cpu() - 1.0, br.grad, what="bias grad (patch form)")
    # backward-data with an accumulate-into residual
    dy = torch.rand(y_ref.shape, generator=gen) * 2 - 1
    dx_ref = torch.nn.grad.conv3d_input(a.shape, w, dy, padding=p)
    acc0 = torch.rand(n, *spatial, cin, generator=gen)
    dx = acc0.cuda()
    ops.conv_backward_data(g, to_cl(dy), ops.pack_weight(w.cuda(), for_dgrad=True), dx, resid=dx)
    assert_close(from_cl(dx, 3), dx_ref + acc0.permute(0, 4, 1, 2, 3), what="backward data + residual")


@pytest.mark.parametrize("cd,cg,s,spatial,n,transposed", [
    (16, 16, 1, (136, 200), 3, False),      # 17 x 13 x 3 = 663 tiles over 512 persistent blocks: two tiles per block
    (32, 32, 1, (72, 144), 4, False),       # 9 x 9 x 4 = 324 tiles
    (32, 16, 2, (140, 260), 2, False),      # stride 2: 70 x 130 coarse pixels, ragged tiles
    (64, 32, 2, (66, 150), 3, False),       # 8-wide tiles: 5 x 10 x 3
    (64, 16, 2, (40, 52), 3, True),         # ConvTranspose2d(64 -> 16): dense = x (coarse grid), gathered = dy
])
def test_patch2d_weight_gradient_prologue_slices_and_many_tiles(cd, cg, s, spatial, n, transposed):
    """The 2-D patch form of the weight gradient (wgrad_patch2d_kernel) at sizes where a persistent block walks
    several tiles: prologue (producer BatchNorm + PReLU with a device slope, zero padding of the ACTIVATED tensor)
    on the gathered operand, both operands as channel slices of wider buffers, fused bias gradient, accumulate."""
    from mpgan_amd import ops
    gen = torch.Generator().manual_seed(500 + cd + cg + s)
    if not transposed:
        cin, cout = cg, cd
        g = _geom(2, n, cin, cout, 3, s, 1, spatial)
        z = torch.rand(n, cin, *spatial, generator=gen) * 2 - 1
        sc, sh = torch.rand(cin, generator=gen) + 0.5, torch.rand(cin, generator=gen) - 0.5
        alpha = 0.3
        a = z * sc[None, :, None, None] + sh[None, :, None, None]
        a = torch.where(a > 0, a, alpha * a).requires_grad_(True)
        w = ((torch.rand(cout, cin, 3, 3, generator=gen) - 0.5) / (cin * 9) ** 0.5).requires_grad_(True)
        b = (torch.rand(cout, generator=gen) - 0.5).requires_grad_(True)
        y = F.conv2d(a, w, b, stride=s, padding=1)
        gy = torch.rand(y.shape, generator=gen) * 2 - 1
        y.backward(gy)
        zbuf = torch.full((n, 1, *spatial, cin + 8), float("nan"), device="cuda")
        zbuf[..., 4:4 + cin] = to_cl(z)
        gybuf = torch.full((n, *g.out_dhw, cout + 4), float("nan"), device="cuda")
        gybuf[..., :cout] = to_cl(gy)
        pro = ops.Prologue(sc.cuda(), sh.cuda(), 0, ops.ACT_LEAKY, 1.0, torch.tensor([alpha], device="cuda"))
        dw = torch.ones_like(w.detach()).cuda()
        db = torch.ones(cout, device="cuda")
        ws = torch.empty(ops.conv_wgrad_workspace(g) // 4, device="cuda")
        ops.conv_backward_weight(g, zbuf[..., 4:4 + cin], gybuf[..., :cout], dw, ws, pro=pro, beta=1.0, dbias=db)
        assert_close(dw.cpu() - 1.0, w.grad, what="wgrad (2-D patch form)")
        assert_close(db.cpu() - 1.0, b.grad, what="bias grad (2-D patch form)")
        ops.conv_backward_weight(g, zbuf[..., 4:4 + cin], gybuf[..., :cout], dw, ws, pro=pro, beta=0.0)
        assert_close(dw.cpu(), w.grad, what="wgrad (2-D patch form), overwrite, no bias")
    else:
        cin, cout = cd, cg
        g = _geom(2, n, cin, cout, 3, 2, 1, spatial, transposed=True)
        x = (torch.rand(n, cin, *spatial, generator=gen) * 2 - 1).requires_grad_(True)
        w = ((torch.rand(cin, cout, 3, 3, generator=gen) - 0.5) / (cin * 9) ** 0.5).requires_grad_(True)
        y = F.conv_transpose2d(x, w, None, stride=2, padding=1, output_padding=1)
        gy = torch.rand(y.shape, generator=gen) * 2 - 1
        y.backward(gy)
        xbuf = torch.full((n, 1, *spatial, cin + 32), float("nan"), device="cuda")      # x = a slice of the concat buffer
        xbuf[..., :cin] = to_cl(x.detach())
        dw = torch.full_like(w.detach(), float("nan")).cuda()
        ws = torch.empty(ops.conv_wgrad_workspace(g) // 4, device="cuda")
        ops.conv_backward_weight(g, xbuf[..., :cin], to_cl(gy), dw, ws)
        assert_close(dw.cpu(), w.grad, what="convT wgrad (2-D patch form)")


@pytest.fixture
def dma_form():
    """Run small shapes through gather_conv_dma_kernel (it otherwise serves prologue-free launches of >= 1024 blocks):
    the threshold travels with the geometry (mpgan_conv_geom.min_blocks)."""
    _MIN_BLOCKS[0] = 1
    yield
    _MIN_BLOCKS[0] = 0


DMA_CASES = [   # enough tiles for the 64- / 128-wide K-stepped variants (select_variant), prologue-free
    (2, 64, 128, 3, 1, 0, (62, 66), 14),     # D conv2's class: backward-data on 64-wide tiles (> 384 blocks: no in-block split-K)
    (2, 128, 256, 4, 2, 0, (72, 72), 8),     # D conv3's: four phases of 2 x 2 taps, 128-wide tiles
    (2, 256, 256, 4, 2, 0, (45, 45), 8),     # D conv4's, odd input: uneven phases, two channel tiles
    (2, 96, 128, 3, 1, 0, (62, 66), 8),      # three K chunks per tap; ragged 128-wide tile (96 channels) in backward-data
    (2, 64, 160, 3, 1, 1, (40, 44), 12),     # padded forward gather (masked pieces), ragged channel tiles
    (3, 64, 128, 3, 1, 0, (20, 20, 20), 6),  # 27 taps
]


@pytest.mark.parametrize("case", DMA_CASES, ids=lambda c: "d{}_{}to{}_k{}s{}p{}".format(*c[:6]))
def test_dma_staged_form_forward_and_backward_data(dma_form, case):
    """The DMA-staged form of the K-stepped kernel (prologue-free gathers: D's backward-data launches at C3) on the
    discriminator's layer classes: the forward (no prologue, bias) and / or the backward-data launch route through it
    and must agree with torch like the pipelined kernel does."""
    from mpgan_amd import ops
    dims, cin, cout, k, s, p, spatial, n = case
    g = _geom(dims, n, cin, cout, k, s, p, spatial)
    assert ops.conv_variant(g, False, 0) >= 3000 or ops.conv_variant(g, True, 0) >= 3000
    test_conv_forward_dgrad_wgrad(case)


@pytest.mark.parametrize("cin,cout,k,s,spatial,n", [(64, 128, 3, 1, (62, 66), 14), (128, 256, 4, 2, (72, 72), 8),
                                                    (256, 256, 4, 2, (45, 45), 8)])
def test_dma_staged_form_with_fused_norm_backward_sums(dma_form, cin, cout, k, s, spatial, n):
    from mpgan_amd import ops
    g = _geom(len(spatial), n, cin, cout, k, s, 0, spatial)
    assert ops.conv_variant(g, True, 0) >= 3000
    test_backward_data_with_fused_norm_backward_sums(cin, cout, k, s, spatial, n)


@pytest.mark.parametrize("case", [(3, 64, 16, (8, 9, 10), 2), (3, 192, 32, (5, 6, 4), 2), (2, 192, 32, (12, 10), 3)],
                         ids=lambda c: "d{}_{}to{}".format(*c[:3]))
def test_dma_staged_form_32_wide_transposed_layers(dma_form, case):
    """The 32-wide instance on the generator's ConvTranspose layers without a prologue: forward
    (1 - 8 taps per phase, 8 / 4 phases) and backward-data (27 / 9 taps)."""
    from mpgan_amd import ops
    dims, cin, cout, spatial, n = case
    g = _geom(dims, n, cin, cout, 3, 2, 1, spatial, transposed=True)
    assert ops.conv_variant(g, False, 0) == 3032 or ops.conv_variant(g, True, 0) == 3032
    test_conv_transpose_forward_dgrad_wgrad(case)


@pytest.mark.parametrize("case", [(3, 16, 32, 3, 2, 1, (12, 14, 10), 2), (3, 32, 64, 3, 2, 1, (8, 8, 10), 2)],
                         ids=lambda c: "d{}_{}to{}_k{}s{}p{}".format(*c[:6]))
def test_dma_staged_form_32_wide_strided_backward_data(dma_form, case):
    """Backward-data of the generator's stride-2 3-D convs (16 / 32 produced channels, eight phases of 1 - 8 taps)."""
    from mpgan_amd import ops
    dims, cin, cout, k, s, p, spatial, n = case
    g = _geom(dims, n, cin, cout, k, s, p, spatial)
    assert ops.conv_variant(g, True, 0) == 3032
    test_conv_forward_dgrad_wgrad(case)


def test_dma_staged_form_serves_the_discriminator_backward_data_at_c3():
    from mpgan_amd import ops
    for cin, cout, k, s, e in ((64, 128, 3, 1, 254), (128, 256, 4, 2, 252), (256, 256, 4, 2, 125)):
        g = _geom(2, 16, cin, cout, k, s, 0, (e, e))
        assert ops.conv_variant(g, True, 0) >= 3000, (cin, cout)
        assert ops.conv_variant(g, False, 3) < 3000            # the forward keeps its normalise-on-load prologue


@pytest.mark.parametrize("case", [(3, 64, 128, 3, 1, 0, (10, 9, 12), 6), (2, 64, 64, 3, 1, 0, (9, 10), 8),
                                  (3, 32, 64, 4, 1, 0, (7, 8, 9), 4), (3, 32, 32, 3, 1, 0, (5, 5, 5), 3)],
                         ids=lambda c: "d{}_{}to{}_k{}s{}p{}".format(*c[:6]))
@pytest.mark.parametrize("dma", [False, True], ids=["pipe", "dma"])
def test_small_map_valid_conv_backward_data_runs_as_border_class_phases(case, dma):
    """Backward-data of a pad-free stride-1 conv on a map only a few kernels wide (variant B's patch discriminator:
    8^3 -> 10^3, 10^3 -> 12^3): csrc/conv_geom.h splits the produced grid into <= 27 border-class phases with their own
    tap ranges (tools/check_phase_classes.hip checks the construction exhaustively on the host); here the K-stepped
    kernels -- pipelined and DMA-staged -- run them: forward, backward-data and weight gradient against torch."""
    saved = _MIN_BLOCKS[0]
    _MIN_BLOCKS[0] = 1 if dma else 0
    try:
        test_conv_forward_dgrad_wgrad(case)
    finally:
        _MIN_BLOCKS[0] = saved


# ---- eval-mode inference convs (mpgan_conv_forward_act): BatchNorm affine + PReLU + residual in the epilogue ----
# (dims, cin, cout, k, stride, pad, spatial, batch, transposed): one case per kernel class of the generator's eval program
ACT_CASES = [
    (2, 1, 32, 3, 2, 1, (20, 24), 2, False),       # thin_cin1_full: the fused unit0 || residual conv of level 0 (c_norm = 16)
    (2, 16, 16, 3, 1, 1, (40, 24), 2, False),      # persistent 2-D patch kernel, 16-wide MFMA form
    (2, 32, 32, 3, 1, 1, (24, 20), 2, False),      # ... 32-wide form
    (2, 16, 64, 3, 2, 1, (20, 24), 2, False),      # K-stepped, 16 gathered channels (fused 16 -> 32 || 32: c_norm = 32)
    (2, 64, 64, 3, 1, 1, (16, 16), 2, False),      # K-stepped, in-block split-K form
    (2, 128, 128, 3, 1, 1, (8, 8), 2, False),
    (2, 64, 128, 1, 1, 0, (8, 8), 2, False),       # 1 x 1 residual conv (linear epilogue: c_norm = 0)
    (2, 192, 32, 3, 2, 1, (6, 5), 2, True),        # transposed conv, four phases, K-stepped
    (2, 64, 16, 3, 2, 1, (10, 12), 2, True),       # transposed conv on the phase-merged patch kernel
    (2, 32, 1, 3, 2, 1, (12, 10), 3, True),        # quad kernel (C -> 1)
    (2, 1, 1, 3, 1, 1, (16, 20), 2, False),        # 1 -> 1 (generic thin kernel: the rows kernel has no activation)
    (3, 1, 32, 3, 2, 1, (8, 10, 12), 2, False),
    (3, 16, 16, 3, 1, 1, (5, 12, 11), 2, False),   # 3-D patch kernel
    (3, 32, 32, 3, 1, 1, (6, 5, 7), 2, False),
    (3, 192, 32, 3, 2, 1, (3, 4, 5), 1, True),
    (3, 32, 1, 3, 2, 1, (5, 4, 6), 2, True),       # octet kernel (C -> 1)
    (3, 16, 1, 3, 2, 1, (5, 4, 6), 1, True),       # ... four lanes per voxel
]


@pytest.mark.parametrize("case", ACT_CASES, ids=lambda c: "d{}_{}to{}_k{}s{}p{}{}".format(*c[:6], "T" if c[8] else ""))
@pytest.mark.parametrize("with_resid,tanh", [(False, False), (True, False), (True, True)], ids=["plain", "resid", "resid_tanh"])
def test_conv_forward_act_epilogue(case, with_resid, tanh):
    """y = prelu(conv(x) * scale + shift, slope) (+ resid) (tanh) against torch, the vectors produced by
    mpgan_epi_vectors_multi from a BatchNorm's running statistics, the conv's bias and a PReLU weight (channels beyond
    c_norm stay linear); input, output and residual as channel slices of wider buffers."""
    import ctypes as C
    import struct
    from mpgan_amd import ops
    from mpgan_amd._lib import lib
    dims, cin, cout, k, s, p, spatial, n, transposed = case
    gen = torch.Generator().manual_seed(3000 + cin * 5 + cout + dims)
    x = torch.rand(n, cin, *spatial, generator=gen) * 2 - 1
    wshape = (cin, cout) if transposed else (cout, cin)
    w = (torch.rand(*wshape, *([k] * dims), generator=gen) * 2 - 1) / (cin * k ** dims) ** 0.5
    b = torch.rand(cout, generator=gen) - 0.5
    c_norm = 0 if (k == 1) else (cout // 2 if cout in (32, 64) and not transposed and s == 2 else cout)
    gamma, beta = torch.rand(cout, generator=gen) + 0.5, torch.rand(cout, generator=gen) - 0.5
    rm, rv = torch.rand(cout, generator=gen) - 0.5, torch.rand(cout, generator=gen) + 0.25
    alpha, eps = torch.tensor([0.3]), 1e-5
    if transposed:
        z = _convt(dims)(x, w, b, stride=s, padding=p, output_padding=s - 1)
    else:
        z = _conv(dims)(x, w, b, stride=s, padding=p)
    shp = [1, -1] + [1] * dims
    zn = (z - rm.view(shp)) / torch.sqrt(rv.view(shp) + eps) * gamma.view(shp) + beta.view(shp)
    a = torch.where(zn > 0, zn, alpha * zn)
    ref = torch.cat([a[:, :c_norm], z[:, c_norm:]], 1)
    res = torch.rand(ref.shape, generator=gen) - 0.5
    if with_resid:
        ref = ref + res
    if tanh:
        ref = torch.tanh(ref)

    g = _geom(dims, n, cin, cout, k, s, p, spatial, transposed=transposed)
    dev = "cuda"
    vec = torch.full((3, (cout + 3) // 4 * 4), float("nan"), device=dev)
    keep = [t.to(dev) for t in (gamma, beta, rm, rv, b, alpha)]
    eps_bits = struct.unpack("<I", struct.pack("<f", eps))[0]
    row = [keep[0].data_ptr() if c_norm else 0, keep[1].data_ptr() if c_norm else 0, keep[2].data_ptr() if c_norm else 0,
           keep[3].data_ptr() if c_norm else 0, keep[4].data_ptr(), keep[5].data_ptr() if c_norm else 0,
           vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(), c_norm, cout, eps_bits]
    table = torch.tensor([row], dtype=torch.int64, device=dev)
    assert lib().mpgan_epi_vectors_multi(table.data_ptr(), 1, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    # channel slices of wider buffers (multiples of four channels aside, as the U-Net's concatenation buffers are)
    xw = torch.zeros(n, *g.in_dhw, cin + (4 if cin > 1 else 0), device=dev)
    xs = xw[..., :cin] if cin > 1 else xw
    xs.copy_(to_cl(x))
    yw = torch.full((n, *g.out_dhw, cout + (8 if cout > 1 else 0)), float("nan"), device=dev)
    ys = yw[..., 4:4 + cout] if cout > 1 else yw
    rw = torch.zeros(n, *g.out_dhw, 2 * cout, device=dev) if cout > 1 else torch.zeros(n, *g.out_dhw, 1, device=dev)
    rs = rw[..., cout:] if cout > 1 else rw
    rs.copy_(to_cl(res))
    wp = ops.pack_weight(w.to(dev), transposed=transposed)
    ops.conv_forward_act(g, xs, wp, vec[0], vec[1], vec[2], ys, resid=rs if with_resid else None, tanh_out=tanh)
    assert_close(from_cl(ys.contiguous(), dims), ref, what="conv + BatchNorm(eval) + PReLU + residual epilogue")
    if cout > 1:
        assert torch.isnan(yw[..., :4]).all() and torch.isnan(yw[..., 4 + cout:]).all()      # nothing outside the slice
