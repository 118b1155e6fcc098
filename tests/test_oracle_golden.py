"""The oracle restatement against fixtures produced by the REFERENCE'S OWN
code (oracle/make_golden.py ran code/GAN/GAN_final.py and test_runs/GAN.py on
CPU in the build container).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import refmodel as R
from oracle.make_golden import summarize


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_param_counts_match_reference():
    # counts verified against the imported reference modules
    assert R.param_count(R.Discriminator((1, 128, 128, 128))) == 12760065
    assert R.param_count(R.PatchDiscriminator((1, 16, 16, 16))) == 21426817
    assert R.param_count(R.Discriminator((1, 256, 256), dimensions=2)) == 2601857
    assert R.param_count(R.CasNetGenerator((1, 128, 128, 128))) == 6 * 1186972
    assert R.param_count(R.CasNetGenerator((1, 256, 256), dimensions=2)) == 6 * 402442


def test_state_dict_keys_follow_reference_tree():
    d = R.Discriminator((1, 32, 32), dimensions=2)
    keys = set(d.state_dict().keys())
    for j in (0, 3, 6, 9):
        assert f"model_conv.{j}.weight" in keys and f"model_conv.{j}.bias" in keys
    for j in (1, 4, 7, 10):
        assert f"model_conv.{j}.running_var" in keys
    assert "model_linear.1.weight" in keys
    g = R.CasNetGenerator((1, 32, 32), 1, dimensions=2)
    keys = set(g.state_dict().keys())
    for k in ("model.0.model.0.conv.unit0.conv.weight",
              "model.0.model.0.conv.unit1.adn.N.running_mean",
              "model.0.model.0.conv.unit0.adn.A.weight",
              "model.0.model.0.residual.weight",
              "model.0.model.1.submodule.1.submodule.1.submodule.conv.unit1.conv.weight",
              "model.0.model.1.submodule.1.submodule.1.submodule.residual.weight",
              "model.0.model.2.0.conv.weight", "model.0.model.2.0.adn.N.weight",
              "model.0.model.2.1.conv.unit0.conv.weight"):
        assert k in keys, k
    assert "model.0.model.2.1.conv.unit0.adn.N.weight" not in keys  # top RU is conv-only
    # transposed conv weight is (Cin, Cout, k, k)
    assert g.state_dict()["model.0.model.2.0.conv.weight"].shape == (32, 1, 3, 3)


def test_variant_b_discriminator_matches_reference(golden_dir):
    fx = _load(golden_dir, "disc_variant_b.npz")
    d = R.PatchDiscriminator((1, 16, 16, 16))
    R.closed_form_fill_(d)
    d.train()
    x = torch.from_numpy(fx["x"]).requires_grad_(True)
    val, taps = d(x)
    np.testing.assert_allclose(val.detach().numpy(), fx["validity"], rtol=0, atol=1e-6)
    assert len(taps) == 16
    for k, t in taps.items():
        assert tuple(fx[f"tap{k}_shape"]) == tuple(t.shape)
        np.testing.assert_allclose(summarize(t), fx[f"tap{k}"], rtol=1e-5, atol=1e-5)
    for name, lbl in (("one", 1.0), ("smooth", 0.9), ("zero", 0.0)):
        l = R.adversarial_loss(val, torch.full_like(val, lbl))
        np.testing.assert_allclose(l.item(), fx[f"bce_{name}"], rtol=1e-6)
    R.adversarial_loss(val, torch.full_like(val, 0.9)).backward()
    np.testing.assert_allclose(x.grad.numpy(), fx["grad_x"], rtol=1e-4, atol=1e-7)
    for name, p in d.named_parameters():
        np.testing.assert_allclose(summarize(p.grad), fx["grad__" + name],
                                   rtol=2e-4, atol=1e-6, err_msg=name)
    for name, b in d.named_buffers():
        np.testing.assert_allclose(summarize(b.float()), fx["buf__" + name],
                                   rtol=1e-5, atol=1e-6, err_msg=name)


def test_perceptual_loss_matches_reference(golden_dir):
    fx = _load(golden_dir, "perceptual.npz")
    d = R.PatchDiscriminator((1, 16, 16, 16))
    R.closed_form_fill_(d)
    d.train()
    _, ta = d(torch.from_numpy(fx["xa"]))
    _, tb = d(torch.from_numpy(fx["xb"]))
    out = R.perceptual_loss(ta, tb)
    assert out.shape == (1,)
    np.testing.assert_allclose(out.detach().numpy(), fx["loss"], rtol=1e-5)


def test_losses_match_reference_including_bce_clamp(golden_dir):
    fx = _load(golden_dir, "losses.npz")
    y_hat = torch.from_numpy(fx["y_hat"])
    for lbl in (0.0, 0.9, 1.0):
        l = R.adversarial_loss(y_hat, torch.full_like(y_hat, lbl)).item()
        np.testing.assert_allclose(l, fx[f"bce_{lbl}"], rtol=1e-6)
    # y_hat == 0 with label 1 is the -100 clamp the reference's checkpoints show
    assert R.adversarial_loss(torch.zeros(1, 1), torch.ones(1, 1)).item() == 100.0
    l1 = R.reconstruction_loss(torch.from_numpy(fx["a"]), torch.from_numpy(fx["b"])).item()
    np.testing.assert_allclose(l1, fx["l1"], rtol=1e-6)


@pytest.mark.slow
def test_variant_a_discriminator_matches_reference_at_128cubed(golden_dir):
    fx = _load(golden_dir, "disc_variant_a_128.npz")
    torch.set_num_threads(8)
    d = R.Discriminator((1, 128, 128, 128))
    R.closed_form_fill_(d)
    d.train()
    g = torch.Generator().manual_seed(int(fx["seed"]))
    x = (torch.rand(1, 1, 128, 128, 128, generator=g) * 2 - 1).requires_grad_(True)
    v = d(x)
    np.testing.assert_allclose(v.detach().numpy(), fx["validity"], atol=1e-6)
    l = R.adversarial_loss(v, torch.full_like(v, 0.9))
    np.testing.assert_allclose(l.item(), fx["loss"], rtol=1e-5)
    l.backward()
    np.testing.assert_allclose(summarize(x.grad), fx["grad_x"], rtol=1e-4, atol=1e-8)
    for name, p in d.named_parameters():
        np.testing.assert_allclose(summarize(p.grad), fx["grad__" + name],
                                   rtol=1e-3, atol=1e-6, err_msg=name)


def test_custom_dataloader_batching_contract(golden_dir):
    """test_runs/GAN.py:204-233: sequential batches of 2, wraps to index 0
    when the next batch would overrun (drops the tail)."""
    fx = _load(golden_dir, "custom_dataloader.npz")
    t1, t2 = fx["items_t1"], fx["items_t2"]
    idx = 0
    for i in range(4):
        if idx + 2 > len(t1):
            idx = 0
        np.testing.assert_array_equal(fx[f"b{i}_t1"], t1[idx:idx + 2])
        np.testing.assert_array_equal(fx[f"b{i}_t2"], t2[idx:idx + 2])
        idx += 2


def test_crop_patches_gather_is_exact():
    rs = np.random.RandomState(3)
    vols = torch.arange(2 * 12 * 12 * 12, dtype=torch.float32).reshape(2, 1, 12, 12, 12)
    corners = R.draw_corners(rs, 2, 5, (12, 12, 12), 4)
    p = R.crop_patches(vols, corners, 4)
    assert p.shape == (10, 1, 4, 4, 4)
    b, s = 1, 3
    z, y, x = corners[b, s]
    assert torch.equal(p[b * 5 + s, 0], vols[b, 0, z:z + 4, y:y + 4, x:x + 4])


def test_ssim_restatement_properties():
    """oracle/metrics_ref.py (skimage's SSIM restated on scipy): identical images score 1, the score is
    symmetric, and a constant offset lowers only the luminance term by the closed form
    (2 u (u+d) + C1) / (u^2 + (u+d)^2 + C1) on a flat image."""
    import numpy as np
    from oracle.metrics_ref import structural_similarity
    rng = np.random.RandomState(0)
    a = np.round(rng.rand(12, 14, 16) * 255)
    b = np.round(np.clip(a + 20 * rng.randn(*a.shape), 0, 255))
    assert abs(structural_similarity(a, a) - 1.0) < 1e-12
    assert abs(structural_similarity(a, b) - structural_similarity(b, a)) < 1e-12
    flat = np.full((9, 9, 9), 100.0)
    d = 30.0
    c1, c2 = (0.01 * 256) ** 2, (0.03 * 256) ** 2
    want = (2 * 100 * 130 + c1) / (100 ** 2 + 130 ** 2 + c1)      # variances are 0: contrast term = C2/C2
    assert abs(structural_similarity(flat, flat + d) - want) < 1e-12
