"""Host-side logic that needs no GPU: geometry, module tree / state_dict keys,
batching contract, gradient sharding."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F


def test_conv_geom_matches_torch_shapes():
    from mpgan_amd.ops import ConvGeom
    for dims, spatial, k, s, p in [(2, (20, 24), 3, 2, 1), (2, (254, 254), 3, 1, 0), (2, (252, 252), 4, 2, 0),
                                   (3, (8, 10, 12), 3, 2, 1), (3, (124, 124, 124), 4, 2, 0)]:
        x = torch.zeros(1, 1, *spatial)
        conv = F.conv2d if dims == 2 else F.conv3d
        w = torch.zeros(1, 1, *([k] * dims))
        want = tuple(conv(x, w, stride=s, padding=p).shape[2:]) if max(spatial) < 64 else None
        pad3 = lambda v, f: (f,) * (3 - dims) + ((v,) * dims if isinstance(v, int) else tuple(v))
        g = ConvGeom(1, pad3(spatial, 1), 1, 1, pad3(k, 1), pad3(s, 1), pad3(p, 0))
        got = g.out_dhw[3 - dims:]
        assert got == tuple((n + 2 * p - k) // s + 1 for n in spatial)
        if want is not None:
            assert got == want
    gt = ConvGeom(1, (1, 16, 16), 4, 2, (1, 3, 3), (1, 2, 2), (0, 1, 1), True, (0, 1, 1))
    assert gt.out_dhw == (1, 32, 32) and gt.taps == 9


def test_module_tree_has_the_reference_state_dict_keys():
    """Same keys and shapes as the oracle's restatement of the reference tree, so
    a reference checkpoint's generator.* / discriminator.* entries load unchanged."""
    from mpgan_amd.networks import CasNetGenerator, Discriminator
    from oracle import refmodel as R
    for dims, shape in ((2, (1, 64, 64)), (3, (1, 32, 32, 32))):
        ours, ref = CasNetGenerator(shape, 2, dimensions=dims), R.CasNetGenerator(shape, 2, dimensions=dims)
        a, b = ours.state_dict(), ref.state_dict()
        assert list(a.keys()) == list(b.keys())
        assert all(a[k].shape == b[k].shape for k in a)
        ours.load_state_dict(b)                       # strict
    d, dr = Discriminator((1, 128, 128, 128)), R.Discriminator((1, 128, 128, 128))
    assert {k: tuple(v.shape) for k, v in d.state_dict().items()} == \
           {k: tuple(v.shape) for k, v in dr.state_dict().items()}
    assert d.model_linear[1].in_features == 256 * 29 * 29 * 29       # GAN_final.py:201
    assert hasattr(d, "model_conv") and d.model_conv[0].weight.shape == (64, 1, 3, 3, 3)


def test_networks_refuse_cpu_and_eval():
    from mpgan_amd.networks import CasNetGenerator
    g = CasNetGenerator((1, 32, 32), 1, dimensions=2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        g(torch.zeros(1, 1, 32, 32))


def test_unet_rejects_unsupported_configurations():
    from mpgan_amd.networks import UNet
    with pytest.raises(ValueError):
        UNet(2, 3, 1)
    with pytest.raises(ValueError):
        UNet(2, 1, 1, strides=(1, 2, 2))


def test_batching_contract_matches_reference_fixture(golden_dir):
    """CustomDataLoader (test_runs/GAN.py:204-233) against batches produced by the
    reference's own class; BatchLoader = DataLoader(shuffle=True) keeping the tail."""
    from mpgan_amd.data import BatchLoader, CustomDataLoader, SyntheticPairs, collate
    fx = np.load(os.path.join(golden_dir, "custom_dataloader.npz"))
    items = [{"t1w": torch.from_numpy(a), "t2w": torch.from_numpy(b)} for a, b in zip(fx["items_t1"], fx["items_t2"])]
    dl = CustomDataLoader(items, 2)
    for i in range(4):
        b = next(dl)
        np.testing.assert_array_equal(b["t1w"].numpy(), fx[f"b{i}_t1"])
        np.testing.assert_array_equal(b["t2w"].numpy(), fx[f"b{i}_t2"])
    ds = SyntheticPairs(10, (8, 8), seed=3)
    assert ds[0]["t1w"].shape == (1, 8, 8) and -1 <= float(ds[0]["t1w"].min()) and float(ds[0]["t1w"].max()) < 1
    batches = list(BatchLoader(ds, batch_size=4, shuffle=True, seed=1))
    assert [b["t1w"].shape[0] for b in batches] == [4, 4, 2]            # last partial batch kept
    seen = torch.cat([b["t1w"] for b in batches])
    assert sorted(seen.flatten(1).sum(1).tolist()) == sorted(torch.stack([d["t1w"] for d in ds.items]).flatten(1).sum(1).tolist())
    assert collate(ds.items[:3])["t2w"].shape == (3, 1, 8, 8)


def test_shard_batch_partitions_without_overlap():
    from mpgan_amd.parallel import shard_batch
    b = {"t1w": torch.arange(10).reshape(10, 1), "t2w": torch.arange(10, 20).reshape(10, 1)}
    parts = [shard_batch(b, r, 4) for r in range(4)]
    assert [p["t1w"].shape[0] for p in parts] == [3, 3, 2, 2]
    assert torch.equal(torch.cat([p["t1w"] for p in parts]), b["t1w"])
    assert torch.equal(torch.cat([p["t2w"] for p in parts]), b["t2w"])


def test_reference_checkpoint_loads_into_modules(tmp_path):
    """inferrence.py:97-106: a Lightning checkpoint's `state_dict` (keys `generator.*`, `discriminator.*`)
    loads into the whole GAN and, prefix stripped, into a bare generator; extra entries are ignored
    (strict=False, as the reference does).  The file is a synthetic one in the reference's format."""
    from mpgan_amd.gan import load_reference_checkpoint
    from mpgan_amd.networks import CasNetGenerator
    from oracle import refmodel as R
    ref = R.GAN((1, 32, 32), dimensions=2, n_unet_blocks=2)
    R.closed_form_fill_(ref.generator)
    sd = {"generator." + k: v for k, v in ref.generator.state_dict().items()}
    sd.update({"discriminator." + k: v for k, v in ref.discriminator.state_dict().items()})
    sd["some_metric.total"] = torch.zeros(1)
    path = tmp_path / "epoch=3.ckpt"
    torch.save({"epoch": 3, "global_step": 120, "state_dict": sd, "hparams_name": "kwargs"}, path)
    g = CasNetGenerator((1, 32, 32), 2, dimensions=2, device="cpu")
    res = load_reference_checkpoint(g, str(path))
    assert not res.missing_keys
    for k, v in ref.generator.state_dict().items():
        assert torch.equal(g.state_dict()[k], v), k

