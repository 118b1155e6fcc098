"""Norm / activation / loss / optimiser kernels vs the CPU oracle ops."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_helpers import assert_close, from_cl, to_cl

pytestmark = pytest.mark.gpu


def _norm_forward(z_cl, gamma, beta, instance, running=None, eps=1e-5, momentum=0.1):
    from mpgan_amd import ops
    n, d, h, w, c = z_cl.shape
    P = d * h * w
    chunks = ops.stats_chunks(P, c)
    partials = torch.empty((n * chunks + 32) * 2 * c, device="cuda")   # + finalize's fold scratch
    ops.channel_stats(z_cl, partials)
    m = n * c if instance else c
    scale, shift, mean, invstd = (torch.empty(m, device="cuda") for _ in range(4))
    rm, rv, nbt = running if running is not None else (None, None, None)
    ops.norm_finalize(partials, n, chunks, c, P, instance, gamma, beta, eps, momentum, rm, rv, nbt, scale, shift,
                      mean, invstd)
    return scale, shift, mean, invstd


@pytest.mark.parametrize("c,spatial,n", [(16, (1, 40, 36), 3), (1, (1, 64, 64), 2), (128, (1, 9, 7), 2),
                                         (8, (1, 300, 280), 8),   # > 512 partial rows: folded before finalize
                                         (16, (1, 128, 128), 16), (1, (1, 256, 256), 4),   # 1024 rows: block-per-channel
                                         (6, (1, 160, 160), 8),                            # backward finalize, scalar slope sum
                                         (192, (1, 8, 8), 2), (32, (6, 10, 12), 2), (512, (2, 2, 2), 4)])
@pytest.mark.parametrize("instance", [False, True])
def test_norm_prelu_forward_backward(c, spatial, n, instance):
    """stats -> finalize -> apply, and the three-kernel backward, against
    F.batch_norm / F.instance_norm + PReLU autograd (train mode, running stats)."""
    from mpgan_amd import ops
    gen = torch.Generator().manual_seed(31 + c)
    d, h, w = spatial
    z = (torch.rand(n, c, d, h, w, generator=gen) * 3 - 1).requires_grad_(True)
    gamma = (torch.rand(c, generator=gen) + 0.5).requires_grad_(True)
    beta = (torch.rand(c, generator=gen) - 0.5).requires_grad_(True)
    alpha = torch.tensor([0.25]).requires_grad_(True)
    rm, rv = torch.zeros(c), torch.ones(c)
    if instance:
        y = F.instance_norm(z, None, None, gamma, beta, True, 0.1, 1e-5)
    else:
        y = F.batch_norm(z, rm, rv, gamma, beta, True, 0.1, 1e-5)
    a = F.prelu(y, alpha)
    g = torch.rand(a.shape, generator=gen) * 2 - 1
    a.backward(g)

    z_cl = to_cl(z.detach())
    run = None
    if not instance:
        run = (torch.zeros(c, device="cuda"), torch.ones(c, device="cuda"),
               torch.zeros((), dtype=torch.int64, device="cuda"))
    scale, shift, mean, invstd = _norm_forward(z_cl, gamma.detach().cuda(), beta.detach().cuda(), instance, run)
    alpha_d = alpha.detach().cuda()
    pro = ops.Prologue(scale, shift, c if instance else 0, ops.ACT_LEAKY, 1.0, alpha_d)
    out = torch.empty_like(z_cl)
    ops.norm_act_add(z_cl, pro, None, None, out)
    assert_close(from_cl(out, 3), a, what="norm+prelu forward")
    if not instance:
        assert_close(run[0].cpu(), rm, what="running_mean")
        assert_close(run[1].cpu(), rv, what="running_var (unbiased)")
        assert int(run[2].item()) == 1

    P = d * h * w
    chunks = ops.stats_chunks(P, c)
    partials = torch.empty(n * chunks * (3 * c + 1), device="cuda")    # [3][C] rows + one slope scalar per row
    g_cl = to_cl(g)
    ops.norm_bwd_reduce(g_cl, z_cl, pro, mean, invstd, partials)
    dgamma, dbeta = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
    dslope = torch.zeros(1, device="cuda")
    m = n * c if instance else c
    c1, c2 = torch.empty(m, device="cuda"), torch.empty(m, device="cuda")
    ops.norm_bwd_finalize(partials, n, chunks, c, P, instance, dgamma, dbeta, dslope, c1, c2)
    dz = torch.empty_like(z_cl)
    ops.norm_bwd_apply(g_cl, z_cl, pro, mean, invstd, c1, c2, dz)
    assert_close(from_cl(dz, 3), z.grad, rtol=1e-3, what="dz")
    assert_close(dgamma.cpu(), gamma.grad, rtol=1e-3, what="dgamma")
    assert_close(dbeta.cpu(), beta.grad, rtol=1e-3, what="dbeta")
    assert_close(dslope.cpu(), alpha.grad, rtol=1e-3, what="dalpha")


def test_norm_act_add_virtual_residual_and_tanh():
    from mpgan_amd import ops
    gen = torch.Generator().manual_seed(9)
    n, c, h, w = 2, 32, 10, 12
    z, r = torch.rand(n, c, h, w, generator=gen) - 0.5, torch.rand(n, c, h, w, generator=gen) - 0.5
    s1, t1, s2, t2 = (torch.rand(c, generator=gen) for _ in range(4))
    lrelu = lambda x, a: torch.where(x > 0, x, a * x)
    ref = torch.tanh(lrelu(z * s1[None, :, None, None] + t1[None, :, None, None], 0.25)
                     + lrelu(r * s2[None, :, None, None] + t2[None, :, None, None], 0.1))
    buf = torch.full((n, 1, h, w, 2 * c), float("nan"), device="cuda")
    ops.norm_act_add(to_cl(z), ops.Prologue(s1.cuda(), t1.cuda(), 0, ops.ACT_LEAKY, 0.25), to_cl(r),
                     ops.Prologue(s2.cuda(), t2.cuda(), 0, ops.ACT_LEAKY, 0.1), buf[..., c:], tanh_out=True)
    assert_close(from_cl(buf[..., c:], 2), ref, what="virtual residual")
    assert torch.isnan(buf[..., :c]).all()


def test_sigmoid_bce_matches_oracle_incl_clamp(golden_dir):
    from mpgan_amd import ops
    from oracle import refmodel as R
    logits = torch.tensor([-200.0, -20.0, -1.5, 0.0, 0.3, 4.0, 30.0, 200.0])
    for target in (1.0, 0.9, 0.0):
        x = logits.clone().requires_grad_(True)
        p = torch.sigmoid(x).unsqueeze(1)
        loss_ref = R.adversarial_loss(p, torch.full_like(p, target))
        (0.5 * loss_ref).backward()
        prob, loss, dl = torch.empty(8, device="cuda"), torch.empty(1, device="cuda"), torch.empty(8, device="cuda")
        ops.sigmoid_bce(logits.cuda(), target, 0.5, prob, loss, dl)
        assert_close(prob.cpu(), p.detach().flatten(), what="prob")
        assert_close(loss.cpu(), loss_ref.detach().reshape(1), what=f"bce target {target}")
        assert_close(dl.cpu(), x.grad, rtol=1e-4, atol=1e-9, what="dlogit")
    # saturated: the reference's checkpoints show g_loss = 100.03 (log clamp at -100)
    loss = torch.empty(1, device="cuda")
    ops.sigmoid_bce(torch.full((4,), -500.0, device="cuda"), 1.0, 1.0, None, loss, None)
    assert loss.item() == 100.0


def test_l1_loss_and_grad(golden_dir):
    import os
    from mpgan_amd import ops
    fx = np.load(os.path.join(golden_dir, "losses.npz"))
    a, b = torch.from_numpy(fx["a"]), torch.from_numpy(fx["b"])
    part = torch.empty(ops.l1_partials(), device="cuda")
    loss, grad = torch.empty(1, device="cuda"), torch.empty_like(a, device="cuda")
    ops.l1_loss(a.cuda(), b.cuda(), part, loss, grad, 1.0)
    np.testing.assert_allclose(loss.item(), float(fx["l1"]), rtol=1e-6)  # reference's own value
    a2 = a.clone().requires_grad_(True)
    F.l1_loss(a2, b).backward()
    assert torch.equal(grad.cpu(), a2.grad)
    big_a, big_b = torch.rand(3, 1, 300, 301), torch.rand(3, 1, 300, 301)
    ops.l1_loss(big_a.cuda(), big_b.cuda(), part, loss)
    np.testing.assert_allclose(loss.item(), F.l1_loss(big_a, big_b).item(), rtol=1e-5)


def test_adam_matches_torch_adam_over_steps():
    from mpgan_amd import ops
    gen = torch.Generator().manual_seed(4)
    p0 = torch.rand(10007, generator=gen) - 0.5
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=5e-4, betas=(0.5, 0.999))
    p, m, v = p0.cuda(), torch.zeros(10007, device="cuda"), torch.zeros(10007, device="cuda")
    for step in range(1, 6):
        g = torch.rand(10007, generator=gen) - 0.5
        p_ref.grad = g.clone()
        opt.step()
        ops.adam_step(p, g.cuda(), m, v, 5e-4, 0.5, 0.999, 1e-8, step)
        assert_close(p.cpu(), p_ref.detach(), rtol=1e-6, atol=1e-7, what=f"adam step {step}")
    st = opt.state[p_ref]
    assert_close(m.cpu(), st["exp_avg"], rtol=1e-6, atol=1e-9, what="exp_avg")
    assert_close(v.cpu(), st["exp_avg_sq"], rtol=1e-6, atol=1e-12, what="exp_avg_sq")


def test_linear1_head_forward_backward():
    """Flatten + Linear(F,1) on the last conv's RAW output with BN+LeakyReLU on load."""
    from mpgan_amd import ops
    gen = torch.Generator().manual_seed(8)
    n, c, h, w = 3, 64, 9, 7
    z = torch.rand(n, c, h, w, generator=gen) - 0.5
    sc, sh = torch.rand(c, generator=gen) + 0.5, torch.rand(c, generator=gen) - 0.5
    lin = torch.nn.Linear(c * h * w, 1)
    a = (z * sc[None, :, None, None] + sh[None, :, None, None])
    a = F.leaky_relu(a, 0.2).requires_grad_(True)
    logit_ref = lin(a.flatten(1))
    dl = torch.rand(n, 1, generator=gen) - 0.5
    logit_ref.backward(dl)

    wperm = ops.pack_weight(lin.weight.detach().reshape(1, c, 1, h, w).cuda())
    pro = ops.Prologue(sc.cuda(), sh.cuda(), 0, ops.ACT_LEAKY, 0.2)
    z_cl = to_cl(z)
    part = torch.empty(ops.linear1_partials(n), device="cuda")
    logit = torch.empty(n, device="cuda")
    ops.linear1_forward(z_cl, pro, wperm, lin.bias.detach().cuda(), part, logit)
    assert_close(logit.cpu(), logit_ref.detach().flatten(), what="logit")
    g_a = torch.empty_like(z_cl)
    dw = torch.zeros(c * h * w, device="cuda")
    db = torch.zeros(1, device="cuda")
    ops.linear1_backward(z_cl, pro, wperm, dl.flatten().cuda(), g_a, dw, db, beta=0.0)
    assert_close(from_cl(g_a, 2), a.grad, what="g_a")
    assert_close(dw.cpu(), lin.weight.grad.flatten(), what="dW (torch flatten order)")
    assert_close(db.cpu(), lin.bias.grad, what="dbias")


def test_patch_gather_is_bit_exact_and_scatter_is_its_adjoint():
    from mpgan_amd import ops
    from oracle import refmodel as R
    rs = np.random.RandomState(3)
    B, S, roi = 2, 7, 4
    vol = torch.rand(B, 1, 10, 12, 14)
    corners = R.draw_corners(rs, B, S, (10, 12, 14), roi)
    want = R.crop_patches(vol, corners, roi)
    c_dev = torch.from_numpy(corners.reshape(-1, 3).astype(np.int32)).cuda()
    got = torch.empty(B * S, 1, roi, roi, roi, device="cuda")
    ops.patch_gather(vol.cuda(), c_dev, S, (roi,) * 3, got)
    assert torch.equal(got.cpu(), want)  # index op: bit-exact
    v = vol.clone().requires_grad_(True)
    gp = torch.rand(want.shape)
    R.crop_patches(v, corners, roi).backward(gp)
    dvol = torch.zeros_like(vol, device="cuda")
    ops.patch_scatter_add(gp.cuda(), c_dev, S, (roi,) * 3, dvol)
    assert_close(dvol.cpu(), v.grad, rtol=1e-6, what="scatter-add")


def test_eval_metrics_match_numpy_restatement():
    """inferrence.py:188-204 / psnr_ssim_metric.py:88-106 restated in numpy: min/max rescale to
    0..255, round, MAE; MSE and PSNR with data_range 256."""
    from mpgan_amd import metrics
    gen = torch.Generator().manual_seed(2)
    a = torch.rand(1, 1, 40, 48, 44, generator=gen) * 2 - 1
    b = (a + 0.1 * torch.randn(a.shape, generator=gen)).clamp(-1, 1)

    def rescale(x):
        x = x.numpy().astype(np.float64)
        lo, hi = np.percentile(x, 0), np.percentile(x, 100)
        return np.round(np.clip((x - lo) / (hi - lo) * 255.0, 0, 255))

    ra, rb = rescale(a), rescale(b)
    got = metrics.rescale_0_255(a.cuda()).cpu().numpy()
    assert (np.abs(got - ra) > 0).mean() < 1e-4          # .5 ties may round differently in fp32
    s = metrics.score_volume(a.cuda(), b.cuda())
    mse = ((ra - rb) ** 2).mean()
    np.testing.assert_allclose(s["mae"].item(), np.abs(ra - rb).mean(), rtol=1e-3)
    np.testing.assert_allclose(s["mse"].item(), mse, rtol=1e-3)
    np.testing.assert_allclose(s["psnr"].item(), 10 * np.log10(256.0 ** 2 / mse), rtol=1e-4)


@pytest.mark.parametrize("shape", [(40, 48, 44), (9, 35, 71), (64, 50), (7, 7, 7), (128, 128, 128)])
def test_ssim_matches_restated_skimage_algorithm(shape):
    """psnr_ssim_metric.py:91-92: structural_similarity(t2, t2_gen, data_range=256) on 0..255 volumes (and,
    for the 2-D instantiation, slices).  Oracle: oracle/metrics_ref.py (skimage's algorithm on
    scipy.ndimage.uniform_filter, float64; skimage itself is absent: parity unpinned against it).
    Tolerance 1e-6: the kernel accumulates the window sums in double."""
    from mpgan_amd import metrics
    from oracle.metrics_ref import structural_similarity
    gen = torch.Generator().manual_seed(31 + len(shape))
    a = torch.round(torch.rand(shape, generator=gen) * 255)
    b = torch.round((a + 25 * torch.randn(shape, generator=gen)).clamp(0, 255))
    want = structural_similarity(a.numpy(), b.numpy(), data_range=256)
    got = metrics.ssim(a.cuda(), b.cuda(), 256.0).item()
    assert abs(got - want) < 1e-6, (got, want)
    assert abs(metrics.ssim(a.cuda(), a.cuda()).item() - 1.0) < 1e-6          # identical images
    if len(shape) == 3 and shape[0] >= 7:
        s = metrics.score_volume(a.cuda(), b.cuda())
        assert "ssim" in s


def test_ssim_rejects_small_or_mismatched_inputs():
    from mpgan_amd import metrics
    with pytest.raises(ValueError):
        metrics.ssim(torch.zeros(5, 5, device="cuda"), torch.zeros(5, 5, device="cuda"))
    with pytest.raises(ValueError):
        metrics.ssim(torch.zeros(8, 8, device="cuda"), torch.zeros(8, 9, device="cuda"))
    with pytest.raises(RuntimeError, match="neither a slice"):
        metrics.ssim(torch.zeros(3, 8, 8, device="cuda"), torch.zeros(3, 8, 8, device="cuda"))


@pytest.mark.parametrize("n,kind", [(1, "rand"), (2, "rand"), (1000, "rand"), (64 * 64 * 64, "mri"),
                                    (100003, "dups"), (128 * 128 * 128, "mri")])
def test_percentiles_are_numpy_percentiles(n, kind):
    """The radix select returns the exact order statistics: np.percentile (linear interpolation) on the same
    float32 data, relative 1e-6 (the interpolation itself runs in double on both sides)."""
    from mpgan_amd import preprocess
    rng = np.random.RandomState(n % 9973)
    if kind == "rand":
        x = (rng.rand(n) * 2000 - 1000).astype(np.float32)
    elif kind == "dups":
        x = rng.randint(-5, 6, size=n).astype(np.float32)            # heavy ties, negative values, zeros
    else:                                                             # 60 % background, skewed foreground
        x = np.where(rng.rand(n) < 0.6, 0.0, rng.gamma(2.0, 300.0, size=n)).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    for q in ((1.0, 99.0), (0.0, 100.0), (50.0,), (12.5, 87.25)):
        got = preprocess.percentiles(xt, q).cpu().numpy().astype(np.float64)
        want = np.percentile(x.astype(np.float64), q)
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6 * max(1.0, np.abs(want).max()), err_msg=str(q))


def test_scale_intensity_range_percentiles_matches_monai_restatement():
    """GAN_final.py:386-394: lower=1, upper=99, b_min=-1, b_max=1, clip=True on a 96^3 volume."""
    from mpgan_amd import preprocess
    from oracle.metrics_ref import scale_intensity_range_percentiles
    rng = np.random.RandomState(5)
    vol = np.where(rng.rand(96, 96, 96) < 0.55, 0.0, rng.gamma(2.0, 250.0, size=(96, 96, 96))).astype(np.float32)
    want = scale_intensity_range_percentiles(vol.astype(np.float64))
    got = preprocess.scale_intensity_range_percentiles(torch.from_numpy(vol).cuda()).cpu().numpy()
    assert got.min() >= -1.0 and got.max() <= 1.0
    np.testing.assert_allclose(got, want, atol=2e-6)
    # no clipping, other target range
    want2 = scale_intensity_range_percentiles(vol.astype(np.float64), 5, 95, 0.0, 255.0, clip=False)
    got2 = preprocess.scale_intensity_range_percentiles(torch.from_numpy(vol).cuda(), 5, 95, 0.0, 255.0, False)
    np.testing.assert_allclose(got2.cpu().numpy(), want2, rtol=1e-5, atol=1e-3)


def test_resample_to_identity_grid_matches_restatement():
    """Next-row N3, second half: `ResampleT1T2d` (code/GAN/transforms.py:79-213) on arrays.  ITK is absent, so the
    kernel is held to an independent numpy restatement of ITK's published algorithm (oracle/resample_ref.py;
    PARITY UNPINNED against ITK itself) on an anisotropic, rotated, off-centre geometry, plus two properties that
    need no reference: an input that already lives on the reference grid comes back unchanged, and everything
    outside the input's half-voxel border is the default pixel 0."""
    from mpgan_amd import preprocess
    from oracle import resample_ref
    gen = torch.Generator().manual_seed(21)
    vol = torch.rand(36, 48, 40, generator=gen) * 100
    th = 0.2
    direction = [[np.cos(th), -np.sin(th), 0.0], [np.sin(th), np.cos(th), 0.0], [0.0, 0.0, 1.0]]
    origin, spacing = (-70.0, -95.0, -60.0), (3.4, 4.1, 3.2)
    for out_size in ((32, 32, 32), (48, 40, 24)):
        ref = resample_ref.resample_to_identity_grid(vol.numpy(), origin, spacing, direction, out_size)
        got = preprocess.resample_to_identity_grid(vol.cuda(), origin, spacing, direction, out_size).cpu().numpy()
        assert got.shape == tuple(reversed(out_size))
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-4)
        assert (ref == 0).any() and (ref != 0).any()                 # the geometry exercises both sides of the border
    size = (24, 20, 16)                                              # (x, y, z) on the reference grid itself
    v2 = torch.rand(size[2], size[1], size[0], generator=gen)
    sp = [256.0 / s for s in size]
    same = preprocess.resample_to_identity_grid(v2.cuda(), [-s / 2 for s in size], sp, np.eye(3).tolist(), size).cpu()
    np.testing.assert_allclose(same.numpy(), v2.numpy(), rtol=0, atol=1e-6)
    far = preprocess.resample_to_identity_grid(v2.cuda(), (1000.0, 1000.0, 1000.0), sp, np.eye(3).tolist(), size)
    assert float(far.abs().max()) == 0.0
