"""Generate tests/golden/*.npz by running the REFERENCE'S OWN CODE on CPU.

Run in the build container only (needs /root/reference; the GPU box never
sees it):   python oracle/make_golden.py

The reference files `code/GAN/GAN_final.py` and `test_runs/GAN.py` import
third-party packages that are neither vendored nor installed here (monai,
pytorch_lightning, itk, torchvision, ...).  They are replaced by inert stubs
in sys.modules so the files execute as modules (their `__main__` guards keep
the trainers from running).  What then runs is the reference's real
`Discriminator` (variants A and B), `GAN.adversarial_loss`,
`GAN.reconstruction_loss`, `GAN.perceptual_loss` and `CustomDataLoader` --
pure torch.  `CasNetGenerator` cannot run (needs real monai UNet): generator
parity stays unpinned.

Weights are closed-form (oracle.refmodel.closed_form_fill_), so fixtures hold
only inputs and expected outputs.  Large tensors are stored as
(checksum, first 64 values, strided sample).
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


class _Meta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything


class _Anything(metaclass=_Meta):
    """Inert stand-in for any third-party symbol the reference imports."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        return _Anything()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything


def _install_stubs():
    names = [
        "transforms", "itk", "torchvision", "torchvision.transforms",
        "matplotlib", "matplotlib.pyplot",
        "monai", "monai.apps", "monai.config", "monai.data", "monai.inferers",
        "monai.losses", "monai.metrics", "monai.networks",
        "monai.networks.layers", "monai.networks.nets", "monai.transforms",
        "monai.utils", "monai.visualize", "monai.visualize.img2tensorboard",
        "pytorch_lightning", "pytorch_lightning.loggers",
        "pytorch_lightning.callbacks",
        "pytorch_lightning.callbacks.model_checkpoint",
    ]
    for n in names:
        if n in sys.modules and not isinstance(sys.modules[n], _StubModule):
            if n.startswith("matplotlib"):
                continue
        m = _StubModule(n)
        m.__path__ = []
        sys.modules[n] = m
    pl = sys.modules["pytorch_lightning"]

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            self.hparams = types.SimpleNamespace()

        def log(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    pl.LightningDataModule = object


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def summarize(t: torch.Tensor):
    """(sum, abs-sum, first 64, 64 strided samples) in float64."""
    f = t.detach().double().flatten()
    stride = max(1, f.numel() // 64)
    return np.concatenate([
        np.array([f.sum().item(), f.abs().sum().item(), float(f.numel())]),
        f[:64].numpy() if f.numel() >= 64 else np.pad(f.numpy(), (0, 64 - f.numel())),
        f[::stride][:64].numpy() if f.numel() >= 64 else np.pad(f[::stride].numpy(), (0, 64 - len(f[::stride]))),
    ])


def main():
    from oracle.refmodel import closed_form_fill_

    torch.manual_seed(0)
    torch.set_num_threads(8)
    _install_stubs()
    ref_a = _load(os.path.join(REF, "code/GAN/GAN_final.py"), "ref_gan_final")
    ref_b = _load(os.path.join(REF, "test_runs/GAN.py"), "ref_gan_b")
    os.makedirs(OUT, exist_ok=True)

    # ---- G1: variant-B discriminator on (2,1,16,16,16) --------------------
    dB = ref_b.Discriminator((1, 16, 16, 16))
    assert sum(p.numel() for p in dB.parameters()) == 21426817
    closed_form_fill_(dB)
    dB.train()
    g = torch.Generator().manual_seed(11)
    xB = (torch.rand(2, 1, 16, 16, 16, generator=g) * 2 - 1).requires_grad_(True)
    val, taps = dB(xB)
    gan_b = ref_b.GAN.__new__(ref_b.GAN)
    nn.Module.__init__(gan_b)
    fx = {"x": xB.detach().numpy(), "validity": val.detach().numpy()}
    for k, t in taps.items():
        fx[f"tap{k}"] = summarize(t)
        fx[f"tap{k}_shape"] = np.array(t.shape)
    for lbl_name, lbl in (("one", 1.0), ("smooth", 0.9), ("zero", 0.0)):
        loss = gan_b.adversarial_loss(val, torch.full_like(val, lbl))
        fx[f"bce_{lbl_name}"] = np.array(loss.item())
    loss = gan_b.adversarial_loss(val, torch.full_like(val, 0.9))
    loss.backward()
    fx["grad_x"] = xB.grad.numpy()
    for name, p in dB.named_parameters():
        fx["grad__" + name] = summarize(p.grad)
    for name, b in dB.named_buffers():
        fx["buf__" + name] = summarize(b.float())
    np.savez_compressed(os.path.join(OUT, "disc_variant_b.npz"), **fx)

    # ---- G5: perceptual loss on the taps of two inputs --------------------
    dB2 = ref_b.Discriminator((1, 16, 16, 16))
    closed_form_fill_(dB2)
    dB2.train()
    g = torch.Generator().manual_seed(12)
    xa = torch.rand(2, 1, 16, 16, 16, generator=g) * 2 - 1
    xb = torch.rand(2, 1, 16, 16, 16, generator=g) * 2 - 1
    _, ta = dB2(xa)
    _, tb = dB2(xb)
    pl_val = gan_b.perceptual_loss(ta, tb)
    np.savez_compressed(os.path.join(OUT, "perceptual.npz"), xa=xa.numpy(),
                        xb=xb.numpy(), loss=pl_val.detach().numpy())

    # ---- G6/L1: BCE incl. saturation, L1 ----------------------------------
    y_hat = torch.tensor([[0.0], [1.0], [0.3], [1e-45], [0.999999]])
    out = {}
    for lbl in (0.0, 0.9, 1.0):
        out[f"bce_{lbl}"] = gan_b.adversarial_loss(
            y_hat, torch.full_like(y_hat, lbl)).item()
    gan_a = ref_a.GAN.__new__(ref_a.GAN)
    nn.Module.__init__(gan_a)
    out["bce_a_1.0"] = gan_a.adversarial_loss(y_hat, torch.ones_like(y_hat)).item()
    g = torch.Generator().manual_seed(13)
    a = torch.rand(3, 1, 8, 8, 8, generator=g) * 2 - 1
    b = torch.rand(3, 1, 8, 8, 8, generator=g) * 2 - 1
    out["l1"] = gan_a.reconstruction_loss(a, b).item()
    np.savez_compressed(os.path.join(OUT, "losses.npz"), y_hat=y_hat.numpy(),
                        a=a.numpy(), b=b.numpy(),
                        **{k: np.array(v) for k, v in out.items()})

    # ---- G2: variant-A discriminator (only valid at 128^3) ----------------
    dA = ref_a.Discriminator((1, 128, 128, 128))
    assert sum(p.numel() for p in dA.parameters()) == 12760065
    closed_form_fill_(dA)
    dA.train()
    g = torch.Generator().manual_seed(14)
    xA = (torch.rand(1, 1, 128, 128, 128, generator=g) * 2 - 1).requires_grad_(True)
    vA = dA(xA)
    lA = gan_a.adversarial_loss(vA, torch.full_like(vA, 0.9))
    lA.backward()
    fa = {"seed": np.array(14), "validity": vA.detach().numpy(),
          "loss": np.array(lA.item()), "grad_x": summarize(xA.grad)}
    for name, p in dA.named_parameters():
        fa["grad__" + name] = summarize(p.grad)
    for name, b in dA.named_buffers():
        fa["buf__" + name] = summarize(b.float())
    np.savez_compressed(os.path.join(OUT, "disc_variant_a_128.npz"), **fa)

    # ---- a11: CustomDataLoader batching (test_runs/GAN.py:204-233) --------
    class DS:
        def __init__(self, n):
            g = torch.Generator().manual_seed(15)
            self.items = [{"t1w": torch.rand(1, 4, 4, 4, generator=g),
                           "t2w": torch.rand(1, 4, 4, 4, generator=g)}
                          for _ in range(n)]

        def __len__(self):
            return len(self.items)

        def __getitem__(self, i):
            return self.items[i]

    ds = DS(5)
    dl = ref_b.CustomDataLoader(ds, 2)
    batches = [next(dl) for _ in range(4)]
    np.savez_compressed(
        os.path.join(OUT, "custom_dataloader.npz"),
        items_t1=torch.stack([d["t1w"] for d in ds.items]).numpy(),
        items_t2=torch.stack([d["t2w"] for d in ds.items]).numpy(),
        **{f"b{i}_t1": b["t1w"].numpy() for i, b in enumerate(batches)},
        **{f"b{i}_t2": b["t2w"].numpy() for i, b in enumerate(batches)})
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
