"""MPGAN_CONV_MM_BF16 (config C5's generator): matrix operands rounded to bf16 on their way into LDS, fp32 storage and
accumulation (include/mpgan_hip.h; csrc/conv_mm16.hip, the MM16 forms of the 3-D patch kernels).

What pins the kernels:
  * operands that are exactly representable in bf16 (sparse small integers, power-of-two prologue vectors): every
    product and every partial sum is exact in fp32, so forward / backward-data / backward-weight must equal torch's
    fp32 result BIT FOR BIT -- this pins fragment maps, LDS images, tap walks, k-sub order, split-K folds;
  * random operands against torch fp32 on the SAME bf16-rounded operands (oracle/mm16_emul.py's contract: products
    exact, only the summation order differs): the fp32 kernels' own tolerance, 2e-4;
  * the generator's layer classes at the reference's true shape family (3-D, 16 / 32 / 64 / 128 channels, 1 -> 1 U-Net).
Whole-network checks (forward, step) live in tests/test_c5_step_gpu.py."""
import dataclasses

import pytest
import torch
import torch.nn.functional as F

from gpu_helpers import assert_close, from_cl, t3, to_cl

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rb(t):
    return t.to(BF).float()


def _geom(dims, n, cin, cout, k, s, p, spatial, transposed=False):
    from mpgan_amd.ops import ConvGeom
    return ConvGeom(n, t3(spatial, dims, 1), cin, cout, t3(k, dims, 1), t3(s, dims, 1), t3(p, dims, 0), transposed,
                    t3(s - 1, dims, 0) if transposed else (0, 0, 0), mm_bf16=True)


def _ints(shape, gen, density, lo=-2, hi=2):
    v = torch.randint(lo, hi + 1, shape, generator=gen).float()
    return v * (torch.rand(shape, generator=gen) < density).float()


# (dims, cin, cout, k, s, p, spatial, n): the generator's K-stepped and patch layer classes at C5, small extents
CASES = [
    (3, 32, 32, 3, 1, 1, (12, 10, 14), 2),      # 32 -> 32 @32^3 class: 32-wide tile, 27 K-steps
    (3, 64, 64, 3, 1, 1, (8, 8, 8), 2),         # 64 -> 64 @16^3: 64-wide, in-block split-K (<= 384 blocks)
    (3, 64, 128, 3, 1, 1, (8, 6, 8), 2),        # bottom unit0
    (3, 128, 128, 3, 1, 1, (8, 8, 8), 4),       # bottom unit1: 108 K-steps
    (3, 64, 128, 1, 1, 0, (8, 8, 8), 2),        # bottom residual: 1x1x1, two K-steps
    (3, 16, 64, 3, 2, 1, (16, 12, 16), 2),      # down1 unit0 || residual fused (Cin = 16: two taps per K-step)
    (3, 32, 128, 3, 2, 1, (12, 12, 8), 2),      # down2 fused
    (3, 16, 16, 3, 1, 1, (6, 16, 16), 2),       # 3-D patch kernels (16 -> 16): several 2x8x8 tiles
    (3, 16, 16, 3, 1, 1, (5, 9, 11), 1),        # ... ragged tiles
    (2, 128, 128, 3, 1, 1, (32, 32), 2),        # 2-D K-stepped layer (the flag is honoured there as well)
    (3, 128, 256, 4, 2, 0, (10, 10, 10), 2),    # pad-free k4 s2 (a discriminator class): 64 taps, four phases in dgrad
]
CONVT = [(3, 64, 16, (8, 6, 8), 2), (3, 192, 32, (4, 6, 4), 2)]


def _run_conv(case, exact):
    from mpgan_amd import ops
    dims, cin, cout, k, s, p, spatial, n = case
    gen = torch.Generator().manual_seed(31 * cin + cout + k)
    conv = F.conv2d if dims == 2 else F.conv3d
    if exact:
        x = _ints((n, cin, *spatial), gen, 0.25).requires_grad_(True)
        w = _ints((cout, cin, *([k] * dims)), gen, 0.25).requires_grad_(True)
        b = _ints((cout,), gen, 0.5).requires_grad_(True)
        gy_fn = lambda shape: _ints(shape, gen, 0.25)
    else:
        x = (torch.rand(n, cin, *spatial, generator=gen) * 2 - 1).requires_grad_(True)
        w = ((torch.rand(cout, cin, *([k] * dims), generator=gen) * 2 - 1) / (cin * k ** dims) ** 0.5).requires_grad_(True)
        b = (torch.rand(cout, generator=gen) - 0.5).requires_grad_(True)
        gy_fn = lambda shape: torch.rand(shape, generator=gen) * 2 - 1
    # the contract: conv(rb(x), rb(w)) + b;  dgrad conv^T(rb(gy), rb(w));  wgrad corr(rb(x), rb(gy));  db = sum(gy)
    xr, wr = rb(x.detach()).requires_grad_(True), rb(w.detach()).requires_grad_(True)
    y_ref = conv(xr, wr, b, stride=s, padding=p)
    gy = gy_fn(y_ref.shape)
    gx_ref, gw_ref = torch.autograd.grad(conv(xr, wr, None, stride=s, padding=p), (xr, wr), rb(gy))
    db_ref = gy.sum([0] + list(range(2, gy.dim())))

    g = _geom(dims, n, cin, cout, k, s, p, spatial)
    xc, wc = to_cl(x.detach()), w.detach().cuda()
    y = torch.full((n, *g.out_dhw, cout), float("nan"), device="cuda")
    ops.conv_forward(g, xc, ops.pack_weight(wc), b.detach().cuda(), y)
    dx = torch.full((n, *g.in_dhw, cin), float("nan"), device="cuda")
    ops.conv_backward_data(g, to_cl(gy), ops.pack_weight(wc, for_dgrad=True), dx)
    ws = torch.empty(ops.conv_wgrad_workspace(g) // 4, device="cuda")
    dw = torch.full_like(wc, float("nan"))
    db = torch.full((cout,), float("nan"), device="cuda")
    ops.conv_backward_weight(g, xc, to_cl(gy), dw, ws, dbias=db)
    got = dict(forward=from_cl(y, dims), dgrad=from_cl(dx, dims), wgrad=dw.cpu(), dbias=db.cpu())
    want = dict(forward=y_ref.detach(), dgrad=gx_ref, wgrad=gw_ref, dbias=db_ref)
    for name in got:
        if exact:
            assert torch.equal(got[name], want[name]), (name, (got[name] - want[name]).abs().max().item())
        else:
            assert_close(got[name], want[name], what=name)
    # and the flag really selects another arithmetic: against fp32 operands the random case must differ
    if not exact:
        y32 = conv(x.detach(), w.detach(), b.detach(), stride=s, padding=p)
        assert (from_cl(y, dims) - y32).abs().max().item() > 1e-4 * y32.abs().max().item()


@pytest.mark.parametrize("case", CASES, ids=lambda c: "d{}_{}to{}_k{}s{}p{}".format(*c[:6]))
def test_mm16_conv_is_exact_on_bf16_representable_operands(case):
    _run_conv(case, exact=True)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "d{}_{}to{}_k{}s{}p{}".format(*c[:6]))
def test_mm16_conv_matches_fp32_conv_of_rounded_operands(case):
    _run_conv(case, exact=False)


@pytest.mark.parametrize("exact", [True, False], ids=["ints", "random"])
@pytest.mark.parametrize("case", CONVT, ids=lambda c: "d{}_{}to{}".format(*c[:3]))
def test_mm16_conv_transpose(case, exact):
    from mpgan_amd import ops
    dims, cin, cout, spatial, n = case
    k, s, p = 3, 2, 1
    gen = torch.Generator().manual_seed(900 + cin + cout)
    convt = F.conv_transpose3d
    if exact:
        x, w = _ints((n, cin, *spatial), gen, 0.25), _ints((cin, cout, k, k, k), gen, 0.25)
        b = _ints((cout,), gen, 0.5)
    else:
        x = torch.rand(n, cin, *spatial, generator=gen) * 2 - 1
        w = (torch.rand(cin, cout, k, k, k, generator=gen) * 2 - 1) / (cin * 27) ** 0.5
        b = torch.rand(cout, generator=gen) - 0.5
    xr, wr = rb(x).requires_grad_(True), rb(w).requires_grad_(True)
    y_ref = convt(xr, wr, b, stride=s, padding=p, output_padding=s - 1)
    gy = _ints(y_ref.shape, gen, 0.25) if exact else torch.rand(y_ref.shape, generator=gen) * 2 - 1
    gx_ref, gw_ref = torch.autograd.grad(convt(xr, wr, None, stride=s, padding=p, output_padding=s - 1), (xr, wr), rb(gy))
    g = _geom(dims, n, cin, cout, k, s, p, spatial, transposed=True)
    xc, wc = to_cl(x), w.cuda()
    y = torch.full((n, *g.out_dhw, cout), float("nan"), device="cuda")
    ops.conv_forward(g, xc, ops.pack_weight(wc, transposed=True), b.cuda(), y)
    dx = torch.full((n, *g.in_dhw, cin), float("nan"), device="cuda")
    ops.conv_backward_data(g, to_cl(gy), ops.pack_weight(wc, transposed=True, for_dgrad=True), dx)
    ws = torch.empty(ops.conv_wgrad_workspace(g) // 4, device="cuda")
    dw = torch.full_like(wc, float("nan"))
    ops.conv_backward_weight(g, xc, to_cl(gy), dw, ws)
    for name, got, want in (("convT forward", from_cl(y, dims), y_ref.detach()), ("convT dgrad", from_cl(dx, dims), gx_ref),
                            ("convT wgrad", dw.cpu(), gw_ref)):
        if exact:
            assert torch.equal(got, want), (name, (got - want).abs().max().item())
        else:
            assert_close(got, want, what=name)


@pytest.mark.parametrize("cin,cout,k,s,spatial", [(16, 16, 3, 1, (6, 12, 16)), (32, 32, 3, 1, (8, 8, 12)),
                                                   (16, 64, 3, 2, (8, 12, 16)), (128, 128, 3, 1, (8, 8, 8))])
def test_mm16_prologue_residual_statistics(cin, cout, k, s, spatial):
    """The producer's BatchNorm + PReLU on load (fp32 arithmetic, THEN the rounding), residual add, fused statistics
    rows (taken from the fp32 accumulators) and the weight gradient's prologue on its gathered operand; channel slices.
    The prologue's scale vector holds powers of two: z * scale is then exact, so the kernel's fused multiply-add and
    torch's multiply-then-add round `z * scale + shift` identically and both sides round the SAME fp32 activation to
    bf16.  (With a general scale the two differ by one fp32 ulp on about half the elements, one in ~2^16 of which
    then falls on the other side of a bf16 rounding boundary: a 2^-9 relative change of one operand, 6.7e-5 of the
    output scale in the first run of this test -- a property of comparing two fp32 prologues, not of the kernel.)"""
    from mpgan_amd import ops
    n, p = 2, 1
    gen = torch.Generator().manual_seed(5 + cin + cout)
    z = torch.rand(n, cin, *spatial, generator=gen) * 2 - 1
    sc = 2.0 ** torch.randint(-1, 2, (cin,), generator=gen).float()
    sh, alpha = torch.rand(cin, generator=gen) - 0.5, 0.25
    a = z * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1)
    a = torch.where(a > 0, a, alpha * a)
    w = (torch.rand(cout, cin, k, k, k, generator=gen) * 2 - 1) / (cin * k ** 3) ** 0.5
    b = torch.rand(cout, generator=gen) - 0.5
    ar, wr = rb(a).requires_grad_(True), rb(w).requires_grad_(True)
    y0 = F.conv3d(ar, wr, b, stride=s, padding=p)
    r = torch.rand(y0.shape, generator=gen) * 2 - 1
    g = _geom(3, n, cin, cout, k, s, p, spatial)
    xbuf = torch.zeros(n, *spatial, cin + 8, device="cuda")
    xbuf[..., 4:4 + cin] = to_cl(z)
    ybuf = torch.full((n, *g.out_dhw, cout + 8), float("nan"), device="cuda")
    pro = ops.Prologue(sc.cuda(), sh.cuda(), 0, ops.ACT_LEAKY, 1.0, torch.tensor([alpha], device="cuda"))
    rows = ops.conv_stats_rows(g, 1)
    assert rows == ops.conv_stats_rows(dataclasses.replace(g, mm_bf16=False), 1)      # sizing does not depend on the flag
    stats = torch.full((max(rows, 1) * 2 * cout,), float("nan"), device="cuda")
    ops.conv_forward(g, xbuf[..., 4:4 + cin], ops.pack_weight(w.cuda()), b.cuda(), ybuf[..., 8:], pro=pro, resid=to_cl(r))
    assert_close(from_cl(ybuf[..., 8:], 3), (y0 + r).detach(), what="forward + residual")
    assert torch.isnan(ybuf[..., :8]).all()
    if rows:                               # (fused statistics describe the raw conv output: a launch without the residual)
        ops.conv_forward(g, xbuf[..., 4:4 + cin], ops.pack_weight(w.cuda()), b.cuda(), ybuf[..., 8:], pro=pro,
                         stats_partials=stats)
        assert_close(from_cl(ybuf[..., 8:], 3), y0.detach(), what="forward with statistics")
        st = stats.view(rows, 2, cout).double().sum(0).cpu()
        zf = y0.detach().double()
        assert_close(st[0].float(), zf.sum((0, 2, 3, 4)).float(), rtol=1e-4, what="fused sum(z)")
        assert_close(st[1].float(), (zf * zf).sum((0, 2, 3, 4)).float(), rtol=1e-4, what="fused sum(z^2)")
    gy = torch.rand(y0.shape, generator=gen) * 2 - 1
    gw_ref, = torch.autograd.grad(F.conv3d(ar, wr, None, stride=s, padding=p), (wr,), rb(gy))
    ws = torch.empty(ops.conv_wgrad_workspace(g) // 4, device="cuda")
    dw = torch.empty_like(w, device="cuda")
    ops.conv_backward_weight(g, xbuf[..., 4:4 + cin], to_cl(gy), dw, ws, pro=pro)
    assert_close(dw.cpu(), gw_ref, what="wgrad with prologue")


def test_mm16_flag_leaves_thin_layers_and_row_counts_alone():
    """1-channel layers run on the vector ALUs in fp32 whatever the flag says; kernel families and statistics-row
    counts are the same with and without it (the caller's buffers are sized once)."""
    from mpgan_amd import ops
    gen = torch.Generator().manual_seed(3)
    for cin, cout, k, s in ((1, 16, 3, 2), (1, 1, 3, 1)):
        x = torch.rand(2, cin, 8, 8, 8, generator=gen) * 2 - 1
        w = torch.rand(cout, cin, k, k, k, generator=gen) - 0.5
        g = _geom(3, 2, cin, cout, k, s, 1, (8, 8, 8))
        y = torch.empty(2, *g.out_dhw, cout, device="cuda")
        ops.conv_forward(g, to_cl(x), ops.pack_weight(w.cuda()), None, y)
        assert_close(from_cl(y, 3), F.conv3d(x, w, None, stride=s, padding=1), what="thin layer stays fp32")
    for case in CASES:
        dims, cin, cout, k, s, p, spatial, n = case
        g = _geom(dims, n, cin, cout, k, s, p, spatial)
        g0 = dataclasses.replace(g, mm_bf16=False)
        for code in (0, 1):
            assert ops.conv_stats_rows(g, code) == ops.conv_stats_rows(g0, code)
        assert ops.conv_wgrad_workspace(g) == ops.conv_wgrad_workspace(g0)
