"""Whole-network parity: the HIP engine behind the reference's module API vs the
CPU oracle, same closed-form weights (loaded through state_dict), same inputs.

Tolerances (fp32): forward 1e-4 abs on [-1,1] outputs (north_star: G-output
L1 < 1e-4); gradients rtol 2e-3 of the tensor's max (six cascaded train-mode
BatchNorm U-Nets amplify rounding).  Conv biases that feed a norm layer have a
mathematically ZERO gradient (the norm removes the mean): both sides hold
rounding noise there, so those are only checked to be tiny.
"""
import pytest
import torch

from gpu_helpers import assert_close

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import refmodel as R
    return R


def _pre_norm_bias(name, keys):
    if not name.endswith("bias"):
        return False
    if name.startswith("model_conv."):
        return name in ("model_conv.0.bias", "model_conv.3.bias", "model_conv.6.bias", "model_conv.9.bias")
    stem = name[:-len("conv.bias")]
    # followed by a norm layer: its PReLU is always there, gamma only when the norm is affine
    return name.endswith("conv.bias") and ((stem + "adn.N.weight") in keys or (stem + "adn.A.weight") in keys)


def _check_param_grads(ours, ref, rtol=2e-3):
    keys = set(dict(ref.named_parameters()).keys())
    ref_p = dict(ref.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in ref_p.values() if p.grad is not None)
    flipped = []
    for name, p in ours.named_parameters():
        r = ref_p[name]
        assert p.grad is not None, name
        if _pre_norm_bias(name, keys):
            assert p.grad.abs().max().item() <= 1e-4 * gmax + 1e-6, (name, p.grad.abs().max().item())
            continue
        try:
            assert_close(p.grad, r.grad, rtol=rtol, atol=rtol * (r.grad.abs().max().item() + 1e-12), what="grad " + name,
                         outliers=0.005)
        except AssertionError:
            # One activation within fp32 rounding of a PReLU kink (or |y - t| of the L1 loss's sign) lands on the
            # other side in the two implementations; BatchNorm's batch coupling spreads that flip over the whole
            # layer's gradient as a smooth perturbation of up to ~1e-2 in L2.  A wrong kernel
            # gives O(1).  Such a tensor is held to 2e-2 relative L2 instead of elementwise 2e-3.
            rel = ((p.grad.detach().double().cpu() - r.grad.double()).norm() / (r.grad.double().norm() + 1e-300)).item()
            assert rel <= 2e-2, (name, rel)
            flipped.append((name, rel))
    assert len(flipped) <= max(2, len(ref_p) // 10), flipped       # ... and only a few tensors may need it


@pytest.mark.parametrize("dims,spatial,n,norm,nblocks", [
    (2, (32, 48), 2, "batch", 2),
    (2, (64, 64), 3, "batch", 6),       # the reference's 6-U-Net cascade
    (2, (32, 32), 2, "instance", 2),    # north_star's InstanceNorm variant (MONAI Norm.INSTANCE: no gamma / beta)
    (2, (32, 32), 2, "instance_affine", 2),
    (3, (16, 16, 24), 2, "batch", 2),   # the reference's real (3-D) graph
])
def test_generator_forward_backward_matches_oracle(dims, spatial, n, norm, nblocks):
    R = _oracle()
    from mpgan_amd.networks import CasNetGenerator
    ref = R.CasNetGenerator((1, *spatial), nblocks, dimensions=dims, norm=norm)
    R.closed_form_fill_(ref)
    ref.train()
    ours = CasNetGenerator((1, *spatial), nblocks, dimensions=dims, norm=norm)
    ours.load_state_dict(ref.state_dict())
    ours.cuda().train()
    gen = torch.Generator().manual_seed(7)
    x = (torch.rand(n, 1, *spatial, generator=gen) * 2 - 1).requires_grad_(True)
    t = torch.rand(n, 1, *spatial, generator=gen) * 2 - 1
    y_ref = ref(x)
    loss_ref = R.reconstruction_loss(y_ref, t) + 0.1 * (y_ref * y_ref).mean()
    loss_ref.backward()

    xc = x.detach().cuda().requires_grad_(True)
    y = ours(xc)
    l1 = (y.cpu() - y_ref.detach()).abs().mean().item()
    assert l1 < 1e-4, f"G-output L1 vs CPU oracle = {l1:.3e}"
    assert_close(y, y_ref, rtol=0, atol=5e-4, what="G output")
    from mpgan_amd.gan import reconstruction_loss
    loss = reconstruction_loss(y, t.cuda()) + 0.1 * (y * y).mean()
    assert_close(loss.reshape(1), loss_ref.reshape(1), rtol=1e-4, what="loss")
    loss.backward()
    assert_close(xc.grad, x.grad, rtol=2e-3, atol=2e-3 * x.grad.abs().max().item(), what="dL/dx", outliers=0.005)
    _check_param_grads(ours, ref)
    if norm == "batch":   # running statistics are part of parity (updated on every train-mode forward)
        sd, sr = ours.state_dict(), ref.state_dict()
        for k in sr:
            if "running_" in k:
                assert_close(sd[k], sr[k], rtol=1e-4, atol=1e-6, what=k)
            if k.endswith("num_batches_tracked"):
                assert int(sd[k]) == int(sr[k]) == 1


@pytest.mark.parametrize("dims,spatial,n", [(2, (32, 40), 3), (2, (128, 128), 1), (3, (28, 28, 32), 2)])
def test_discriminator_forward_backward_matches_oracle(dims, spatial, n):
    R = _oracle()
    from mpgan_amd.gan import adversarial_loss
    from mpgan_amd.networks import Discriminator
    ref = R.Discriminator((1, *spatial), dimensions=dims)
    R.closed_form_fill_(ref)
    ref.train()
    ours = Discriminator((1, *spatial), dimensions=dims)
    ours.load_state_dict(ref.state_dict())
    ours.cuda().train()
    gen = torch.Generator().manual_seed(17)
    x = (torch.rand(n, 1, *spatial, generator=gen) * 2 - 1).requires_grad_(True)
    acts = {}
    hooks = [ref.model_conv[i].register_forward_hook(lambda m, a, out, i=i: acts.__setitem__(i, out.detach()))
             for i in (1, 4, 7, 10)]
    p_ref = ref(x)
    for h in hooks:
        h.remove()
    loss_ref = R.adversarial_loss(p_ref, torch.full_like(p_ref, 0.9))
    loss_ref.backward()
    xc = x.detach().cuda().requires_grad_(True)
    p = ours(xc)
    assert p.shape == (n, 1)
    assert_close(p, p_ref, rtol=1e-4, atol=1e-6, what="validity")
    loss = adversarial_loss(p, torch.full_like(p, 0.9))
    assert_close(loss.reshape(1), loss_ref.reshape(1), rtol=1e-5, what="bce")
    loss.backward()
    # LeakyReLU kinks: an activation within fp32 rounding of 0 may fall on opposite sides in two correct
    # implementations; its gradient then differs by the slope ratio over its whole receptive field
    # (tools/debug_dgrad_noise.py shows the HIP forward agreeing with a float64 graph to
    # 1e-5 while one flipped element of conv4 moves dL/dx by 8 % on 28 input rows).  Count them.
    plan = [pl for pool in ours._plans.values() for pl in pool][0]
    flips = 0
    for k, i in enumerate((1, 4, 7, 10)):
        z = plan.zs[k].squeeze(1) if dims == 2 else plan.zs[k]
        a = (z * plan.nbs[k].scale + plan.nbs[k].shift).cpu()
        a = a.permute(0, 3, 1, 2) if dims == 2 else a.permute(0, 4, 1, 2, 3)
        flips += int(((a > 0) != (acts[i] > 0)).sum())
    assert flips <= 4, flips
    if flips == 0:
        # four train-mode BatchNorms over few pixels amplify fp32 rounding: 5e-3 of the gradient's max
        assert_close(xc.grad, x.grad, rtol=5e-3, atol=5e-3 * x.grad.abs().max().item(), what="dL/dx", outliers=0.005)
        _check_param_grads(ours, ref, rtol=5e-3)
    else:
        l2 = lambda a, b: ((a.cpu() - b).norm() / (b.norm() + 1e-30)).item()
        # measured with 2 flips at 128x128, bs 1: dL/dx 4e-2, BatchNorm weight gradients (sums that cancel) 7e-2
        assert l2(xc.grad, x.grad) < 1e-1, (flips, l2(xc.grad, x.grad))
        rp = dict(ref.named_parameters())
        for name, p in ours.named_parameters():
            if not _pre_norm_bias(name, set(rp)):
                assert l2(p.grad, rp[name].grad) < 1e-1, (name, flips, l2(p.grad, rp[name].grad))
    sd, sr = ours.state_dict(), ref.state_dict()
    for k in sr:
        if "running_" in k:
            assert_close(sd[k], sr[k], rtol=1e-4, atol=1e-6, what=k)


def test_discriminator_frozen_params_only_input_grad():
    """G-step: D's parameters are toggled off; only dL/d(input) flows."""
    from mpgan_amd.networks import Discriminator
    d = Discriminator((1, 32, 32), dimensions=2, device="cuda")
    for p in d.parameters():
        p.requires_grad_(False)
    x = torch.rand(2, 1, 32, 32, device="cuda", requires_grad=True)
    d(x).sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all() and x.grad.abs().sum() > 0
    assert all(p.grad is None or p.grad.abs().sum() == 0 for p in d.parameters())


def test_two_optimizer_step_matches_oracle_c1():
    """BASELINE config C1: one 128x128 slice pair, bs 1, full G+D steps vs the
    CPU oracle.

    What can and cannot match: Adam's first steps move EVERY parameter by ~lr in
    the direction of sign(grad), so parameters whose gradient is rounding noise
    (pre-norm conv biases: exactly zero in exact arithmetic) or merely tiny take
    unrelated +-lr steps in two fp32 implementations, and that perturbation feeds
    the next forward.  The test therefore pins, per iteration, (a) the generator
    losses, computed BEFORE any update of that iteration, tightly; (b) d_loss,
    computed after that iteration's G update, to 5 %; (c) the updated parameters
    elementwise to 2*lr with >= 98 % of all of them within 2e-4; (d) BatchNorm
    bookkeeping (G: 2 forwards/step, D: 3).  The oracle is re-synchronised to
    our parameters after each iteration so that iteration 2 starts level."""
    R = _oracle()
    from mpgan_amd.gan import GAN
    ref = R.GAN((1, 128, 128), dimensions=2)
    R.closed_form_fill_(ref.generator)
    R.closed_form_fill_(ref.discriminator)
    ref.train()
    ours = GAN(1, 128, 128, dimensions=2)
    ours.generator.load_state_dict(ref.generator.state_dict())
    ours.discriminator.load_state_dict(ref.discriminator.state_dict())
    ours.train()
    opts_ref, _ = ref.configure_optimizers()
    opts, _ = ours.configure_optimizers()
    gen = torch.Generator().manual_seed(1234)
    for it in range(2):
        t1 = torch.rand(1, 1, 128, 128, generator=gen) * 2 - 1
        t2 = torch.rand(1, 1, 128, 128, generator=gen) * 2 - 1
        log_ref = ref.step({"t1w": t1, "t2w": t2}, it, opts_ref)
        log = ours.fit_batch({"t1w": t1.cuda(), "t2w": t2.cuda()}, it, opts)
        for k, tol in (("g_adv_loss", 2e-3), ("g_recon_loss", 2e-3), ("g_loss", 2e-3), ("d_loss", 5e-2)):
            got, want = float(log[k]), log_ref[k]
            assert abs(got - want) <= tol * abs(want) + 1e-5, (it, k, got, want)
        for net, rnet in ((ours.generator, ref.generator), (ours.discriminator, ref.discriminator)):
            keys = set(dict(rnet.named_parameters()).keys())
            rp = dict(rnet.named_parameters())
            n_bad = n_all = 0
            for name, p in net.named_parameters():
                diff = (p.detach().cpu() - rp[name].detach()).abs()
                assert diff.max().item() <= 2 * 5e-4 + 1e-6, (it, name, diff.max().item())
                if not _pre_norm_bias(name, keys):
                    n_bad += int((diff > 2e-4).sum())
                    n_all += diff.numel()
            assert n_bad / n_all < 0.02, (it, n_bad, n_all)
            sd = net.state_dict()
            for k, v in rnet.state_dict().items():
                if k.endswith("num_batches_tracked"):
                    expect = (2 if net is ours.generator else 3) * (it + 1)
                    assert int(sd[k]) == expect, (k, int(sd[k]), expect)
            rnet.load_state_dict({k: v.cpu() for k, v in sd.items()})   # level the field for the next iteration
            for k, v in rnet.state_dict().items():
                if k.endswith("num_batches_tracked"):
                    v.fill_((2 if net is ours.generator else 3) * (it + 1))


def test_missing_library_or_cpu_tensor_fails_loudly():
    from mpgan_amd.networks import CasNetGenerator
    g = CasNetGenerator((1, 32, 32), 1, dimensions=2)     # parameters on the CPU
    with pytest.raises(RuntimeError, match="no CPU path"):
        g(torch.zeros(1, 1, 32, 32))


@pytest.mark.parametrize("dims,size", [(2, 64), (3, 32)], ids=["2d", "3d"])
def test_generator_eval_mode_uses_running_statistics(dims, size):
    """inferrence.py:97-110,169-170: model.eval(), no_grad, batch 1 -- BatchNorm normalises with its running statistics and
    updates nothing.  The eval plan is the fused inference program: every conv applies its layer's
    BatchNorm + PReLU + residual add in its epilogue, so the program holds convs only -- and it follows parameter and
    running-statistics changes made after it was built (its vectors are re-derived on the device by every forward)."""
    R = _oracle()
    from mpgan_amd.networks import CasNetGenerator
    shape = (1,) + (size,) * dims
    ref = R.CasNetGenerator(shape, 2, dimensions=dims)
    R.closed_form_fill_(ref)
    ours = CasNetGenerator(shape, 2, dimensions=dims)
    ours.load_state_dict(ref.state_dict())
    ours.cuda()
    gen = torch.Generator().manual_seed(3)
    warm = torch.rand(2, *shape, generator=gen) * 2 - 1
    ref.train(); ours.train()
    with torch.no_grad():                       # one train-mode pass so running stats are non-trivial
        ref(warm); ours(warm.cuda())
    ref.eval(); ours.eval()
    x = torch.rand(1, *shape, generator=gen) * 2 - 1
    before = {k: v.clone() for k, v in ours.state_dict().items()}
    with torch.no_grad():
        y_ref, y = ref(x), ours(x.cuda())
    assert_close(y, y_ref, rtol=0, atol=5e-4, what="eval-mode G output")
    assert (y.cpu() - y_ref).abs().mean().item() < 1e-4
    after = ours.state_dict()
    assert all(torch.equal(before[k], after[k]) for k in before), "eval forward must not touch parameters or buffers"
    plan = [pl for key, pool in ours._plans.items() for pl in pool if key[4] is False][0]      # key[4]: gen.training
    names = [name for name in plan.fwd.names if name]
    assert "norm_act_add" not in names and "norm_finalize" not in names, names
    assert names.count("conv_forward") == 2 * 15, names                                        # 15 convs per U-Net
    # the same plan after a checkpoint-style change of running statistics, BatchNorm and PReLU parameters
    with torch.no_grad():
        for (k, v), (_, vr) in zip(ours.state_dict().items(), ref.state_dict().items()):
            if k.endswith("running_var") or k.endswith("adn.A.weight"):
                v.mul_(1.25); vr.mul_(1.25)
            elif k.endswith("running_mean") or k.endswith("adn.N.bias"):
                v.add_(0.05); vr.add_(0.05)
        y_ref2, y2 = ref(x), ours(x.cuda())
    assert (y_ref2 - y_ref).abs().max().item() > 1e-3
    assert_close(y2, y_ref2, rtol=0, atol=5e-4, what="eval-mode G output after a parameter change")


def test_stream_overlap_is_bitwise_identical_to_single_stream():
    """The second HIP stream (the generator's weight gradients beside its norm-backward / backward-data chain)
    only re-orders independent work: every kernel is deterministic, so two G+D steps must leave bit-identical
    parameters and losses with and without it.  A missing event/join shows up here as a difference (or as NaNs)."""
    from mpgan_amd import engine
    from mpgan_amd.gan import GAN
    gen = torch.Generator().manual_seed(99)
    batch = {"t1w": (torch.rand(4, 1, 64, 64, generator=gen) * 2 - 1).cuda(),
             "t2w": (torch.rand(4, 1, 64, 64, generator=gen) * 2 - 1).cuda()}
    results = []
    for single in (True, False):
        torch.manual_seed(7)
        gan = GAN(1, 64, 64, dimensions=2, n_unet_blocks=3, g_lr=1e-3, d_lr=1e-3)
        gan.train()
        saved = engine._SINGLE_STREAM
        engine._SINGLE_STREAM = single
        try:
            opts, _ = gan.configure_optimizers()
            logs = [gan.fit_batch(batch, i, opts) for i in range(2)]
            torch.cuda.synchronize()
        finally:
            engine._SINGLE_STREAM = saved
        results.append((gan.generator.store.flat.clone(), gan.discriminator.store.flat.clone(),
                        {k: v.clone() for k, v in logs[-1].items()}))
    (g1, d1, l1), (g2, d2, l2) = results
    assert torch.isfinite(g2).all() and torch.isfinite(d2).all()
    assert torch.equal(g1, g2), (g1 - g2).abs().max().item()
    assert torch.equal(d1, d2), (d1 - d2).abs().max().item()
    for k in l1:
        assert torch.equal(l1[k], l2[k]), k


def test_generator_output_l1_target_at_c2_shape():
    """north_star: "G-output L1 vs CPU reference < 1e-4" on 256x256 slices through the full 6-U-Net cascade
    (config C2's shape at batch 4 so that the CPU oracle finishes in seconds), train-mode BatchNorm."""
    R = _oracle()
    from mpgan_amd.networks import CasNetGenerator
    torch.manual_seed(3)
    ref = R.CasNetGenerator((1, 256, 256), 6, dimensions=2)
    ref.train()
    ours = CasNetGenerator((1, 256, 256), 6, dimensions=2)
    ours.load_state_dict(ref.state_dict())
    ours.cuda().train()
    gen = torch.Generator().manual_seed(1234)
    x = torch.rand(4, 1, 256, 256, generator=gen) * 2 - 1
    with torch.no_grad():
        y_ref = ref(x)
        y = ours(x.cuda()).cpu()
    l1 = (y - y_ref).abs().mean().item()
    mse = ((y - y_ref) ** 2).mean().item()
    psnr = 10 * torch.log10(torch.tensor(4.0 / max(mse, 1e-30))).item()      # data range 2 ([-1, 1])
    assert l1 < 1e-4, l1
    assert (y - y_ref).abs().max().item() < 2e-3
    assert psnr > 80.0, psnr


def test_fused_adam_state_round_trips_in_torch_adam_format():
    """FusedAdam.state_dict() has torch.optim.Adam's layout (what a Lightning checkpoint's
    `optimizer_states` holds): saving after two steps and loading into a fresh optimizer resumes with
    the same moments and step count, and moving the module (.float(): new flat store) keeps them."""
    from mpgan_amd.gan import FusedAdam
    from mpgan_amd.networks import Discriminator
    torch.manual_seed(0)
    d = Discriminator((1, 32, 32), dimensions=2, device="cuda")
    d2 = Discriminator((1, 32, 32), dimensions=2, device="cuda")
    d2.load_state_dict(d.state_dict())
    x = torch.rand(2, 1, 32, 32, device="cuda")
    opt = FusedAdam(d, lr=1e-3, betas=(0.5, 0.999))
    for _ in range(2):
        opt.zero_grad()
        d(x).sum().backward()
        opt.step()
    sd = opt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 2.0
    ref_opt = torch.optim.Adam(d2.parameters(), lr=1e-3, betas=(0.5, 0.999))
    ref_opt.load_state_dict(sd)                                    # torch accepts it as its own format
    d2.load_state_dict(d.state_dict())
    opt2 = FusedAdam(d2, lr=1e-3, betas=(0.5, 0.999))
    opt2.load_state_dict(sd)
    assert opt2.step_count == 2
    d.float()                                                      # re-creates parameter storage: store.version changes
    for o, net in ((opt, d), (opt2, d2)):
        o.zero_grad()
        net(x).sum().backward()
        o.step()
    assert opt.step_count == 3
    assert torch.equal(d.store.flat, d2.store.flat), (d.store.flat - d2.store.flat).abs().max().item()
    assert torch.equal(opt.exp_avg, opt2.exp_avg) and torch.equal(opt.exp_avg_sq, opt2.exp_avg_sq)
    # a state of a differently shaped network is refused BEFORE anything is overwritten
    other = Discriminator((1, 40, 40), dimensions=2, device="cuda")          # another Linear width
    o3 = FusedAdam(other, lr=1e-3, betas=(0.5, 0.999))
    o3.zero_grad()
    other(torch.rand(2, 1, 40, 40, device="cuda")).sum().backward()
    o3.step()
    m_before, steps_before = opt.exp_avg.clone(), opt.step_count
    with pytest.raises(ValueError, match="model_linear.1.weight"):
        opt.load_state_dict(o3.state_dict())
    assert torch.equal(opt.exp_avg, m_before) and opt.step_count == steps_before


def test_boundary_tensors_are_used_in_place_and_guarded():
    """The generator / discriminator read the caller's input and write the returned tensor directly (1-channel
    NC(D)HW == channels-last; engine.IoSlots) -- no staging copies.  Same numbers through the staging path that a
    misaligned view takes; autograd's version counter rejects an in-place edit of the input between forward and
    backward (the first conv's weight gradient reads it again)."""
    from mpgan_amd.networks import CasNetGenerator, Discriminator
    from mpgan_amd.engine import plan_io
    R = _oracle()
    torch.manual_seed(3)
    g = CasNetGenerator((1, 32, 32), 2, dimensions=2).cuda().train()
    d = Discriminator((1, 32, 32), dimensions=2).cuda().train()
    base = torch.rand(2 * 32 * 32 + 1, device="cuda") * 2 - 1
    x_al = base[:-1].clone().view(2, 1, 32, 32).requires_grad_(True)
    x_mis = base[1:].view(2, 1, 32, 32)                       # 4-byte offset: not usable in place
    x_mis = x_mis.detach().requires_grad_(True)
    with torch.no_grad():
        x_mis.copy_(x_al)
    outs = []
    for x in (x_al, x_mis):
        for m in (g, d):
            m.zero_grad()
        # same BatchNorm running statistics for both passes
        y = g(x)
        p = d(y)
        (p.sum() + y.abs().mean()).backward()
        outs.append((y.detach().clone(), p.detach().clone(), x.grad.clone(), g.store.flat_grad.clone(),
                     d.store.flat_grad.clone()))
    plan = next(iter(g._plans.values()))[0]
    io = plan_io(plan)
    assert set(io.slot) == {"x", "y", "g_y", "g_x"}, set(io.slot)
    assert io.slot["x"].value == plan.x_in.data_ptr()          # slots are back on the plan's buffers after a pass
    assert not io.usable(x_mis) and io.usable(x_al)
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    x = x_al.detach().clone().requires_grad_(True)
    y = g(x)
    with torch.no_grad():
        x.add_(1.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        y.sum().backward()
    # the plan is returned to the pool and usable again
    y = g(x_al)
    y.sum().backward()


def test_parameter_writes_after_a_forward_reach_the_packed_weights():
    """The conv weights are re-packed whenever a parameter changed, whichever way it was written: after a forward
    has packed once, `load_state_dict`, an in-place `weight.mul_()` / `nn.init` and a stock `torch.optim.Adam`
    step (all of which bump the Parameter's own version counter, not the flat buffer's) must be seen by the next
    forward.  Checked against a fresh module holding the same weights, and one Adam step against the oracle."""
    R = _oracle()
    from mpgan_amd.networks import CasNetGenerator, Discriminator
    torch.manual_seed(11)
    x = (torch.rand(2, 1, 32, 32) * 2 - 1).cuda()
    for make in (lambda: CasNetGenerator((1, 32, 32), 2, dimensions=2), lambda: Discriminator((1, 32, 32), dimensions=2)):
        a = make().cuda().train()
        with torch.no_grad():
            y0 = a(x).clone()                                   # packs once
        torch.manual_seed(12)
        other = make()                                          # different weights
        sd = {k: v.clone() for k, v in other.state_dict().items()}
        a.load_state_dict(sd)
        fresh = make()
        fresh.load_state_dict(sd)
        fresh.cuda().train()
        with torch.no_grad():
            y1, y_f = a(x), fresh(x)
        assert not torch.equal(y0, y1)
        assert torch.equal(y1, y_f), (y1 - y_f).abs().max().item()
        # in-place edit of one conv weight through the Parameter
        name, w = next((n, p) for n, p in a.named_parameters() if n.endswith("weight") and p.dim() >= 4)
        with torch.no_grad():
            w.mul_(1.5)
        fresh2 = make()
        fresh2.load_state_dict({k: v.clone() for k, v in a.state_dict().items()})   # weights AND BatchNorm buffers
        fresh2.cuda().train()
        with torch.no_grad():
            y2, y3 = a(x), fresh2(x)
        assert not torch.equal(y1, y2)
        assert torch.equal(y2, y3), (name, (y2 - y3).abs().max().item())
    # a stock torch.optim.Adam on the engine's parameters: one step, then a forward, vs the oracle
    torch.manual_seed(13)
    ref = R.Discriminator((1, 32, 32), dimensions=2)
    R.closed_form_fill_(ref)
    ours = Discriminator((1, 32, 32), dimensions=2)
    ours.load_state_dict(ref.state_dict())
    ours.cuda().train()
    opt_r = torch.optim.Adam(ref.parameters(), lr=1e-3, betas=(0.5, 0.999))
    opt_o = torch.optim.Adam(ours.parameters(), lr=1e-3, betas=(0.5, 0.999))
    xc = x.cpu()
    for it in range(2):                                         # two steps: zero_grad() (set_to_none, torch's default)
        for net, opt, inp in ((ref, opt_r, xc), (ours, opt_o, x)):   # must not leave the first step's gradients behind
            opt.zero_grad()
            net(inp).sum().backward()
            if it == 1:
                g_first = dict(net.named_parameters())["model_linear.1.weight"].grad.detach().cpu().clone()
                if net is ref:
                    g_ref = g_first
                else:
                    # (a gradient left over from the first step would double it; the two sides' weights differ by
                    #  Adam's +-lr sign noise on near-zero gradients after step one, hence L2 and not elementwise)
                    rel = ((g_first.double() - g_ref.double()).norm() / g_ref.double().norm()).item()
                    assert rel <= 2e-2, ("head weight gradient on the second step", rel)
            opt.step()
    with torch.no_grad():
        p_r, p_o = ref(xc), ours(x).cpu()
    assert_close(p_o, p_r, rtol=2e-3, atol=2e-4, what="D output after a torch.optim.Adam step")
    # the step really moved the conv weights the second forward used
    w_r = dict(ref.named_parameters())["model_conv.3.weight"]
    w_o = dict(ours.named_parameters())["model_conv.3.weight"].detach().cpu()
    assert (w_o - w_r).abs().max().item() <= 4.5e-3             # two Adam steps of +-lr: sign noise on ~zero gradients aside
