"""Variant B (test_runs/GAN.py): patch discriminator with 16 perceptual taps, the
perceptual-loss hook, the random-crop gather and the full G / D steps, against the CPU
oracle (whose PatchDiscriminator / perceptual_loss are pinned by the reference's own code,
tests/golden/disc_variant_b.npz and perceptual.npz)."""
import os

import numpy as np
import pytest
import torch

from gpu_helpers import assert_close

pytestmark = pytest.mark.gpu


def _pair():
    from mpgan_amd.networks import PatchDiscriminator
    from oracle import refmodel as R
    ref = R.PatchDiscriminator((1, 16, 16, 16))
    R.closed_form_fill_(ref)
    ref.train()
    ours = PatchDiscriminator((1, 16, 16, 16))
    ours.load_state_dict(ref.state_dict())
    ours.cuda().train()
    return ours, ref, R


def _l2rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _kink_flips(tapset, ref, x):
    """Number of BatchNorm outputs whose sign differs between the HIP pass and the oracle:
    a LeakyReLU gradient jumps by 5x there, and with |y| within fp32 rounding of 0 two
    correct fp32 implementations may land on opposite sides (tools/debug_patchd.py shows one
    such element of 524,288 moving dL/dx by 1.6e-2 in L2)."""
    flips, h = 0, x
    with torch.no_grad():
        for i, m in enumerate(ref.model_conv):
            h = m(h)
            if i % 3 == 1:
                flips += int(((tapset.materialize(i).cpu() > 0) != (h > 0)).sum())
    return flips


def test_patch_discriminator_matches_reference_fixture(golden_dir):
    """Same input and closed-form weights as the fixture produced by the REFERENCE's own
    Discriminator class: validity, all 16 taps, BCE and BatchNorm buffers against the
    fixture; gradients against the oracle (itself held to the same fixture), to 2e-3 in L2
    when the two passes agree on every activation sign, 5e-2 when a kink flip occurred."""
    from mpgan_amd.gan import adversarial_loss
    from oracle.make_golden import summarize
    fx = np.load(os.path.join(golden_dir, "disc_variant_b.npz"))
    ours, ref, R = _pair()
    x = torch.from_numpy(fx["x"]).cuda().requires_grad_(True)
    val, taps = ours(x)
    np.testing.assert_allclose(val.detach().cpu().numpy(), fx["validity"], rtol=0, atol=2e-6)
    assert sorted(taps.keys()) == list(range(16))
    for k in range(16):
        t = taps.tapset.materialize(k)
        assert tuple(t.shape) == tuple(fx[f"tap{k}_shape"]), k
        np.testing.assert_allclose(summarize(t.cpu()), fx[f"tap{k}"], rtol=2e-4, atol=2e-4, err_msg=f"tap {k}")
    for name, b in ours.named_buffers():
        np.testing.assert_allclose(summarize(b.float().cpu()), fx["buf__" + name], rtol=1e-4, atol=1e-5, err_msg=name)
    ref2 = R.PatchDiscriminator((1, 16, 16, 16))
    R.closed_form_fill_(ref2)
    ref2.train()
    flips = _kink_flips(taps.tapset, ref2, torch.from_numpy(fx["x"]))
    assert flips <= 3, flips
    tol = 2e-3 if flips == 0 else 5e-2
    loss = adversarial_loss(val, torch.full_like(val, 0.9))
    np.testing.assert_allclose(loss.item(), float(fx["bce_smooth"]), rtol=1e-5)
    loss.backward()
    assert _l2rel(x.grad, torch.from_numpy(fx["grad_x"])) < tol
    xr = torch.from_numpy(fx["x"]).requires_grad_(True)
    vr, _ = ref(xr)
    R.adversarial_loss(vr, torch.full_like(vr, 0.9)).backward()
    rp = dict(ref.named_parameters())
    for name, p in ours.named_parameters():
        if name in ("model_conv.0.bias", "model_conv.3.bias", "model_conv.6.bias", "model_conv.9.bias"):
            continue   # pre-norm conv biases: exactly-zero true gradient, rounding noise on both sides
        assert _l2rel(p.grad, rp[name].grad) < tol, (name, _l2rel(p.grad, rp[name].grad), flips)
        np.testing.assert_allclose(summarize(rp[name].grad)[:3], fx["grad__" + name][:3], rtol=2e-3, atol=1e-6)


def test_perceptual_loss_value_matches_reference_fixture(golden_dir):
    from mpgan_amd.gan_patch import perceptual_loss
    fx = np.load(os.path.join(golden_dir, "perceptual.npz"))
    ours, _, _ = _pair()
    _, ta = ours(torch.from_numpy(fx["xa"]).cuda())
    _, tb = ours(torch.from_numpy(fx["xb"]).cuda())
    out = perceptual_loss(ta, tb)
    assert out.shape == (1,)
    np.testing.assert_allclose(out.detach().cpu().numpy(), fx["loss"], rtol=2e-4)


def test_perceptual_loss_gradient_matches_oracle():
    """d(perceptual + BCE)/d(fake patches) with the discriminator frozen (the G step)."""
    from mpgan_amd.gan import adversarial_loss
    from mpgan_amd.gan_patch import perceptual_loss
    ours, ref, R = _pair()
    for p in list(ours.parameters()) + list(ref.parameters()):
        p.requires_grad_(False)
    gen = torch.Generator().manual_seed(21)
    xf = (torch.rand(3, 1, 16, 16, 16, generator=gen) * 2 - 1).requires_grad_(True)
    xr = torch.rand(3, 1, 16, 16, 16, generator=gen) * 2 - 1
    vf, af = ref(xf)
    _, ar = ref(xr)
    # the perceptual term is ~1e-7 of the BCE term in magnitude: weight it up so both are exercised
    loss_ref = 1e6 * R.perceptual_loss(af, ar).sum() + R.adversarial_loss(vf, torch.ones_like(vf))
    loss_ref.backward()
    xfc = xf.detach().cuda().requires_grad_(True)
    v, tf = ours(xfc)
    _, tr = ours(xr.cuda())
    perc = perceptual_loss(tf, tr)
    loss = 1e6 * perc.sum() + adversarial_loss(v, torch.ones_like(v))
    # the head taps (Linear outputs: differences of two 262,144-term fp32 dot products) carry ~3e-4 of rounding
    assert_close(loss.reshape(1), loss_ref.detach().reshape(1), rtol=1e-3, what="loss")
    flips = _kink_flips(tf.tapset, ref, xf.detach())
    loss.backward()
    assert flips <= 3 and _l2rel(xfc.grad, xf.grad) < (2e-3 if flips == 0 else 5e-2), (flips, _l2rel(xfc.grad, xf.grad))


def test_variant_b_steps_match_oracle():
    """Full variant-B G step and D step (small volumes: 2 x 32^3, 3 crops each)."""
    from mpgan_amd.gan_patch import GAN
    from oracle import refmodel as R
    kw = dict(n_unet_blocks=1, channels=(8, 16, 32), strides=(2, 2), num_samples=3, crop_seed=5)
    ref = R.PatchGAN((1, 32, 32, 32), **kw)
    R.closed_form_fill_(ref.generator)
    R.closed_form_fill_(ref.discriminator)
    ref.train()
    ours = GAN(1, 32, 32, 32, n_unet_blocks=1, unet_channels=(8, 16, 32), unet_strides=(2, 2), num_samples=3,
               crop_seed=5)
    ours.generator.load_state_dict(ref.generator.state_dict())
    ours.discriminator.load_state_dict(ref.discriminator.state_dict())
    ours.train()
    gen = torch.Generator().manual_seed(8)
    batch = {"t1w": torch.rand(2, 1, 32, 32, 32, generator=gen) * 2 - 1,
             "t2w": torch.rand(2, 1, 32, 32, 32, generator=gen) * 2 - 1}
    cb = {k: v.cuda() for k, v in batch.items()}
    # G step: losses and generator gradients
    for p in list(ref.discriminator.parameters()) + list(ours.discriminator.parameters()):
        p.requires_grad_(False)
    l_ref = ref.training_step(batch, 0, 0)
    l_ref.backward()
    l = ours.training_step(cb, 0, 0)
    l.backward()
    for k in ("g_perceptual_loss", "g_adv_loss", "g_recon_loss", "g_loss"):
        got, want = float(ours.logged[k]), ref.logged[k]
        assert abs(got - want) <= 2e-3 * abs(want) + 1e-9, (k, got, want)
    # gradients: L2-relative per tensor, 5e-2 (a single LeakyReLU/PReLU kink flip among ~1e6
    # activations moves small tensors by percents; see tools/debug_patchd.py)
    rp = dict(ref.generator.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in rp.values())
    keys = set(rp)
    for name, p in ours.generator.named_parameters():
        stem = name[:-len("conv.bias")]
        if name.endswith("conv.bias") and (stem + "adn.N.weight") in keys:
            assert p.grad.abs().max().item() <= 1e-4 * gmax + 1e-6
            continue
        # scalar PReLU slopes are sums of ~1e6 signed terms that cancel to ~1e-3 of the largest gradient:
        # a kink flip moves them by percents of their own size, so they also pass on absolute error
        tiny = (p.grad.cpu() - rp[name].grad).abs().max().item() <= 5e-4 * gmax
        assert _l2rel(p.grad, rp[name].grad) < 5e-2 or tiny, ("G grad " + name, _l2rel(p.grad, rp[name].grad), gmax)
    # D step
    for net in (ref, ours):
        for p in net.discriminator.parameters():
            p.requires_grad_(True)
        for p in net.generator.parameters():
            p.requires_grad_(False)
    ours.discriminator.zero_grad()
    ref.discriminator.zero_grad()
    ref.R = np.random.RandomState(6)
    ours.patch_transform.set_random_state(6)
    d_ref = ref.training_step(batch, 0, 1)
    d_ref.backward()
    d = ours.training_step(cb, 0, 1)
    d.backward()
    assert abs(d.item() - d_ref.item()) <= 2e-3 * abs(d_ref.item()) + 1e-7
    rdp = dict(ref.discriminator.named_parameters())
    for name, p in ours.discriminator.named_parameters():
        if name in ("model_conv.0.bias", "model_conv.3.bias", "model_conv.6.bias", "model_conv.9.bias"):
            continue
        assert _l2rel(p.grad, rdp[name].grad) < 5e-2, ("D grad " + name, _l2rel(p.grad, rdp[name].grad))


def _reference_perceptual_loss(y_hat_activations, y_activations):
    """The reference's own loss body, test_runs/GAN.py:288-298, verbatim in behaviour (torch.Tensor([0]).type_as,
    F.l1_loss / numel per key, summed)."""
    import torch.nn.functional as F
    assert set(y_activations.keys()) == set(y_hat_activations.keys())
    running_sum = torch.Tensor([0]).type_as(y_hat_activations[0])
    for key in y_activations.keys():
        layer_contribution = F.l1_loss(y_activations[key], y_hat_activations[key]) / y_activations[key].numel()
        running_sum = running_sum + layer_contribution
    return running_sum


def test_reference_perceptual_loss_body_runs_on_the_returned_dict():
    """Boundary: `Discriminator.forward` returns `(validity, {idx: Tensor})` in the reference
    (test_runs/GAN.py:183-198) and its own `perceptual_loss` (:288-298) indexes that dict.  Here the dict's values
    are materialised on demand as differentiable tensors, so the reference's loss body runs UNCHANGED on it: same
    value as the fused path and as the oracle, and the same gradient w.r.t. the fake patches (the gradients of all
    16 taps travel back through the BatchNorm / conv chain of the pass that produced them)."""
    from mpgan_amd.gan import adversarial_loss
    from mpgan_amd.gan_patch import perceptual_loss
    ours, ref, R = _pair()
    for p in list(ours.parameters()) + list(ref.parameters()):
        p.requires_grad_(False)
    gen = torch.Generator().manual_seed(22)
    xf = torch.rand(2, 1, 16, 16, 16, generator=gen) * 2 - 1
    xr = torch.rand(2, 1, 16, 16, 16, generator=gen) * 2 - 1
    # oracle
    xfr = xf.clone().requires_grad_(True)
    vf, af = ref(xfr)
    _, ar = ref(xr)
    loss_ref = 1e6 * R.perceptual_loss(af, ar).sum() + R.adversarial_loss(vf, torch.ones_like(vf))
    loss_ref.backward()
    # fused path
    x1 = xf.cuda().requires_grad_(True)
    v1, t1 = ours(x1)
    _, t1r = ours(xr.cuda())
    fused = perceptual_loss(t1, t1r)
    (1e6 * fused.sum() + adversarial_loss(v1, torch.ones_like(v1))).backward()
    # the reference's body on the returned dicts
    x2 = xf.cuda().requires_grad_(True)
    v2, t2 = ours(x2)
    _, t2r = ours(xr.cuda())
    assert isinstance(t2[0], torch.Tensor) and tuple(t2[0].shape) == (2, 64, 14, 14, 14) and t2[15].shape == (2, 1)
    body = _reference_perceptual_loss(t2, t2r)
    assert body.shape == (1,)
    assert_close(body, fused.detach(), rtol=1e-5, what="reference body vs fused perceptual loss")
    assert_close(body.detach().cpu(), R.perceptual_loss(af, ar).detach().reshape(1), rtol=1e-3, what="vs oracle")   # head taps: ~3e-4 of rounding (see above)
    flips = _kink_flips(t2.tapset, ref, xf)                                    # (before backward releases the pass)
    (1e6 * body.sum() + adversarial_loss(v2, torch.ones_like(v2))).backward()
    assert _l2rel(x2.grad, x1.grad) < 2e-4, _l2rel(x2.grad, x1.grad)          # same kernels, same stored tensors
    assert flips <= 3 and _l2rel(x2.grad, xfr.grad) < (2e-3 if flips == 0 else 5e-2), (flips, _l2rel(x2.grad, xfr.grad))
    # a second, plain backward through the same discriminator still works (external buffers were cleared)
    x3 = xf.cuda().requires_grad_(True)
    v3, _ = ours(x3)
    adversarial_loss(v3, torch.ones_like(v3)).backward()
    assert torch.isfinite(x3.grad).all()


def test_perceptual_loss_on_plain_mappings_runs_through_the_library():
    """`perceptual_loss` given anything but two TapDicts (taps a caller materialised, detached or re-keyed) restates the
    reference's body (test_runs/GAN.py:288-298) over the library's own L1 / axpby kernels: same value as torch's
    arithmetic on the same tensors, shape (1,), gradients to BOTH arguments (F.l1_loss is differentiable in both)."""
    from mpgan_amd.gan_patch import perceptual_loss
    gen = torch.Generator().manual_seed(4)
    shapes = [(2, 8, 5, 5, 5), (2, 4, 3, 3, 3), (2, 7), (2, 1)]
    fake = {k: (torch.rand(*s, generator=gen) * 2 - 1).cuda().requires_grad_(True) for k, s in enumerate(shapes)}
    real = {k: (torch.rand(*s, generator=gen) * 2 - 1).cuda().requires_grad_(True) for k, s in enumerate(shapes)}
    got = perceptual_loss(fake, real)
    assert got.shape == (1,)
    want = _reference_perceptual_loss({k: v.detach().cpu().requires_grad_(True) for k, v in fake.items()},
                                      {k: v.detach().cpu().requires_grad_(True) for k, v in real.items()})
    assert_close(got.detach().cpu(), want.detach(), rtol=1e-5, what="value")
    got.sum().backward()
    fc = {k: v.detach().cpu().requires_grad_(True) for k, v in fake.items()}
    rc = {k: v.detach().cpu().requires_grad_(True) for k, v in real.items()}
    _reference_perceptual_loss(fc, rc).sum().backward()
    for k in fake:
        assert_close(fake[k].grad.cpu(), fc[k].grad, rtol=1e-5, atol=1e-12, what=f"d/d fake[{k}]")
        assert_close(real[k].grad.cpu(), rc[k].grad, rtol=1e-5, atol=1e-12, what=f"d/d real[{k}]")
