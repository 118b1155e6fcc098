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



def test_reference_checkpoint_with_lightning_pickle_globals(tmp_path):
    """A checkpoint in the REAL format of the reference's trainer (Lightning 1.2.1, three ModelCheckpoint
    callbacks, GAN_final.py:448-484): `callbacks` is keyed by the callback CLASS, so the pickle holds a
    GLOBAL of a module that is not importable here; `optimizer_states` / `hyper_parameters` ride along.
    The weights-only loader must read it without importing or executing anything from the file."""
    import sys
    import types
    from mpgan_amd.gan import load_checkpoint_blob, load_reference_checkpoint
    from mpgan_amd.networks import CasNetGenerator
    from oracle import refmodel as R
    # stand-ins for the trainer's classes, present only while the file is WRITTEN
    names = ["pytorch_lightning", "pytorch_lightning.callbacks", "pytorch_lightning.callbacks.model_checkpoint",
             "pytorch_lightning.utilities", "pytorch_lightning.utilities.parsing"]
    mods = {n: types.ModuleType(n) for n in names}
    ModelCheckpoint = type("ModelCheckpoint", (), {"__module__": names[2]})
    AttributeDict = type("AttributeDict", (dict,), {"__module__": names[4]})
    mods[names[2]].ModelCheckpoint = ModelCheckpoint
    mods[names[4]].AttributeDict = AttributeDict
    ref = R.GAN((1, 32, 32), dimensions=2, n_unet_blocks=2)
    R.closed_form_fill_(ref.generator)
    sd = {"generator." + k: v for k, v in ref.generator.state_dict().items()}
    sd.update({"discriminator." + k: v for k, v in ref.discriminator.state_dict().items()})
    opt = torch.optim.Adam(ref.generator.parameters(), lr=5e-4, betas=(0.5, 0.999))
    for p in ref.generator.parameters():
        p.grad = torch.full_like(p, 0.25)
    opt.step()
    hp = AttributeDict(channels=1, width=32, height=32, g_lr=5e-4)
    blob = {"epoch": 30, "global_step": 14000, "pytorch-lightning_version": "1.2.1",
            "callbacks": {ModelCheckpoint: {"monitor": "g_loss_step", "best_model_score": torch.tensor(100.03),
                                            "best_model_path": "epoch=30-g_loss_step=100.03.ckpt"}},
            "optimizer_states": [opt.state_dict()], "lr_schedulers": [], "state_dict": sd,
            "hparams_name": "kwargs", "hyper_parameters": hp}
    path = tmp_path / "epoch=30-g_loss_step=100.03-g_recon_loss_step=0.03-d_loss_step=45.00.ckpt"
    sys.modules.update(mods)
    try:
        torch.save(blob, path)
    finally:
        for n in names:
            sys.modules.pop(n, None)
    with pytest.raises(Exception):                      # plain weights_only load refuses the class-keyed entry
        torch.load(path, map_location="cpu", weights_only=True)
    got = load_checkpoint_blob(str(path))
    assert got["epoch"] == 30 and got["hyper_parameters"]["width"] == 32
    assert not any(n in sys.modules for n in names)     # nothing was imported on the way
    (cb_key, cb_val), = got["callbacks"].items()
    assert cb_key.__name__ == "ModelCheckpoint" and cb_val["monitor"] == "g_loss_step"
    g = CasNetGenerator((1, 32, 32), 2, dimensions=2, device="cpu")
    res = load_reference_checkpoint(g, str(path))
    assert not res.missing_keys
    for k, v in ref.generator.state_dict().items():
        assert torch.equal(g.state_dict()[k], v), k
    assert len(got["optimizer_states"][0]["state"]) == len(list(ref.generator.parameters()))


def test_bench_launcher_starts_n_ranks_from_a_bare_shell():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must itself start 2 ranks (the
    driver's command form) and report n_gpus 2; a mismatching WORLD_SIZE is refused with rc != 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["dist"]["world_size"] == 2 and line["rank_sum"] == 3.0
    env_bad = dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       env=env_bad, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)


def test_border_class_phases_cover_every_tap_pair_exactly_once(tmp_path):
    """csrc/conv_geom.h splits a stride-1 transposed gather on a small map (the backward-data of variant B's valid 3^3
    convs on 10^3 / 12^3 maps) into up to 27 border-class phases; tools/check_phase_classes.hip enumerates 100+
    geometries on the host and requires every in-range (pixel, tap) pair to be issued exactly once."""
    import shutil
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "check_phase_classes")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), "-I",
                    os.path.join(ROOT, "cross-modality-minipig-gan_amd", "csrc"),
                    os.path.join(ROOT, "tools", "check_phase_classes.hip"), "-o", exe], check=True, capture_output=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert " 0 errors" in out and "geometries used classes" in out, out
