"""Variant B of the reference trainer (test_runs/GAN.py:236-464): 4-U-Net CasNet
(32..256 channels), patch discriminator on 128 random 16^3 crops per volume, G loss =
BCE + L1(patches) + perceptual (sum over the 16 discriminator taps)."""
from __future__ import annotations

import types
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .gan import FusedAdam, adversarial_loss, reconstruction_loss, scalar_axpby
from .networks import CasNetGenerator, PatchDiscriminator, TapDict, _EngineModule


class _CropFn(torch.autograd.Function):
    """The gather half of RandSpatialCropSamplesd (same corners for fake and real,
    test_runs/GAN.py:263-272,313-337): bit-exact copy forward, scatter-add backward."""

    @staticmethod
    def forward(ctx, vol, corners, samples, roi):
        b = vol.shape[0]
        dims = vol.dim() - 2
        roi3 = (1,) * (3 - dims) + tuple(roi)
        v = vol.contiguous()
        out = torch.empty(b * samples, 1, *roi, device=vol.device)
        ops.patch_gather(v if dims == 3 else v.unsqueeze(2), corners, samples, roi3, out)
        ctx.corners, ctx.samples, ctx.roi3, ctx.shape, ctx.dims = corners, samples, roi3, vol.shape, dims
        return out

    @staticmethod
    def backward(ctx, g):
        dvol = torch.zeros(ctx.shape, device=g.device)
        ops.patch_scatter_add(g.contiguous(), ctx.corners, ctx.samples, ctx.roi3,
                              dvol if ctx.dims == 3 else dvol.unsqueeze(2))
        return dvol, None, None, None


class PatchSampler:
    """Corner stream of MONAI's RandSpatialCropSamplesd(random_size=False): per volume,
    per sample, per spatial dim one RandomState.randint(0, size-roi+1) (restated; the RNG
    stream is parity-unpinned, the gather is exact)."""

    def __init__(self, roi_size: Sequence[int], num_samples: int, seed: Optional[int] = None):
        self.roi, self.num_samples = tuple(roi_size), num_samples
        self.R = np.random.RandomState(seed)

    def set_random_state(self, seed):
        self.R = np.random.RandomState(seed)
        return self

    def draw(self, batch: int, spatial: Sequence[int]) -> np.ndarray:
        c = np.zeros((batch, self.num_samples, 3), dtype=np.int32)
        off = 3 - len(spatial)
        for b in range(batch):
            for s in range(self.num_samples):
                for d, n in enumerate(spatial):
                    c[b, s, off + d] = self.R.randint(0, n - self.roi[d] + 1)
        return c

    def __call__(self, fake: torch.Tensor, real: torch.Tensor):
        spatial = tuple(fake.shape[2:])
        corners = torch.from_numpy(self.draw(fake.shape[0], spatial).reshape(-1, 3)).to(fake.device)
        return (_CropFn.apply(fake, corners, self.num_samples, self.roi),
                _CropFn.apply(real, corners, self.num_samples, self.roi))


class _PerceptualFn(torch.autograd.Function):
    """sum_k L1mean(real_k, fake_k) / numel_k over the 16 taps (test_runs/GAN.py:288-298),
    shape (1,).  Values and gradients come from the two passes' raw conv outputs."""

    @staticmethod
    def forward(ctx, h_fake, h_real, taps_fake, taps_real):
        pf, pr = taps_fake.plan, taps_real.plan
        dev = h_fake.device
        part = torch.empty(ops.tap_l1_partials(), device=dev)
        nl = len(pf.zs)
        vals = torch.empty(3 * nl + 3, device=dev)     # the 16 L1 means ([z, y, a] per layer; a of the last layer
        for i in range(nl):                            # stands for keys 11 and 12), weighted by pf.perc_w_fwd
            ops.tap_l1(pf.zs[i], pf.lrelu(pf.nbs[i]), pr.zs[i], pr.lrelu(pr.nbs[i]), part, vals[3 * i:3 * i + 3])
        l1part = torch.empty(ops.l1_partials(), device=dev)
        for k, (a, b) in enumerate(((pf.h, pr.h), (pf.logit, pr.logit), (pf.prob, pr.prob))):
            ops.l1_loss(a.reshape(-1), b.reshape(-1), l1part, vals[3 * nl + k])
        total = torch.empty(1, device=dev)
        ops.weighted_sum(vals, pf.perc_w_fwd, total)
        ctx.taps = (taps_fake, taps_real)
        return total

    @staticmethod
    def backward(ctx, gout):
        taps_fake, taps_real = ctx.taps
        grads = []
        for need, mine, other in ((ctx.needs_input_grad[0], taps_fake, taps_real),
                                  (ctx.needs_input_grad[1], taps_real, taps_fake)):
            if not need:
                grads.append(None)
                continue
            pm, po = mine.plan, other.plan
            dev = gout.device
            g = gout.reshape(1).contiguous()
            ops.scale_by_device_scalar(pm.perc_w_bwd, g, pm.coef_all)      # every layer's (z, y, a) coefficients
            l1part = torch.empty(ops.l1_partials(), device=dev)
            dummy = torch.empty((), device=dev)
            for a, b, dst in ((pm.h, po.h, pm.tap_g_h), (pm.logit, po.logit, pm.tap_g_logit),
                              (pm.prob, po.prob, pm.tap_g_prob)):
                tmp = torch.empty(a.numel(), device=dev)
                ops.l1_loss(a.reshape(-1), b.reshape(-1), l1part, dummy, tmp, 1.0 / a.numel())   # sign/numel^2
                ops.scale_by_device_scalar(tmp, g, tmp)
                ops.axpby(dst.view(-1), 1.0, tmp, 1.0, dst.view(-1))
            pm.peer = po
            pm.peer_lease = other.lease           # keep the peer's buffers alive until our backward has run
            grads.append(torch.zeros(1, device=dev))
        return grads[0], grads[1], None, None


def perceptual_loss(y_hat_activations, y_activations):
    """test_runs/GAN.py:288-298 (y_hat = fake taps, y = real taps)."""
    assert set(y_activations.keys()) == set(y_hat_activations.keys())
    if isinstance(y_hat_activations, TapDict) and isinstance(y_activations, TapDict):
        tf, tr = y_hat_activations.tapset, y_activations.tapset      # fused: the 16 taps are never materialised
        return _PerceptualFn.apply(tf.handle, tr.handle, tf, tr)
    # any other mapping of tensors (e.g. taps a caller materialised or detached): the reference's body statement by
    # statement, through the library's own L1 and axpby kernels (no eager-torch arithmetic on the path)
    first = next(iter(y_hat_activations.values()))
    running_sum = torch.zeros(1, device=first.device, dtype=first.dtype)
    for key in y_activations.keys():
        term = reconstruction_loss(y_activations[key], y_hat_activations[key])              # F.l1_loss(y, y_hat), mean
        running_sum = scalar_axpby(running_sum, 1.0, term.reshape(1), 1.0 / y_activations[key].numel())
    return running_sum


class GAN(nn.Module):
    """test_runs/GAN.py:236-464 without Lightning."""

    def __init__(self, channels, width, height, depth=None, latent_dim: int = 100, lr: float = 0.0002,
                 b1: float = 0.5, b2: float = 0.999, batch_size: int = 64, example_data=None,
                 one_sided_label_value=0.9, *, dimensions: Optional[int] = None, n_unet_blocks: int = 4,
                 unet_channels=(32, 64, 128, 256), unet_strides=(2, 2, 2, 2), roi_size=None, num_samples: int = 128,
                 crop_seed: Optional[int] = None, use_perceptual: bool = True, device="cuda", **kwargs):
        super().__init__()
        if dimensions is None:
            dimensions = 3 if depth is not None else 2
        self.hparams = types.SimpleNamespace(latent_dim=latent_dim, lr=lr, b1=b1, b2=b2, batch_size=batch_size,
                                             one_sided_label_value=one_sided_label_value)
        data_shape = (channels, width, height) + ((depth,) if dimensions == 3 else ())
        roi_size = tuple(roi_size) if roi_size is not None else (16,) * dimensions
        self.generator = CasNetGenerator(data_shape, n_unet_blocks, dimensions=dimensions, channels=unet_channels,
                                         strides=unet_strides, device=device)
        self.discriminator = PatchDiscriminator(data_shape, use_perceptual=use_perceptual, dimensions=dimensions,
                                                patch=roi_size[0], device=device)
        self.patch_transform = PatchSampler(roi_size, num_samples, crop_seed)
        self.logged: Dict[str, torch.Tensor] = {}
        self.ddp = None

    def forward(self, x):
        return self.generator(x)

    def adversarial_loss(self, y_hat, y):
        return adversarial_loss(y_hat, y)

    def reconstruction_loss(self, y_hat, y):
        return reconstruction_loss(y_hat, y)

    def perceptual_loss(self, y_hat_activations, y_activations):
        return perceptual_loss(y_hat_activations, y_activations)

    def log(self, name, value, **kw):
        self.logged[name] = value.detach()

    def training_step(self, batch, batch_idx, optimizer_idx):
        t1w_images, t2w_images = batch["t1w"], batch["t2w"]
        generated_imgs = self(t1w_images)                                   # :308 (always, before the branch)
        self.generated_imgs = generated_imgs
        t2_generated_batch, t2_ground_truth_batch = self.patch_transform(generated_imgs, t2w_images)   # :313-337
        dev, dt = t1w_images.device, t1w_images.dtype
        if optimizer_idx == 0:                                              # :340-390
            valid = torch.ones(t2_ground_truth_batch.shape[0], 1, device=dev, dtype=dt)
            disc_output_fake, disc_activations_fake = self.discriminator(t2_generated_batch)
            _, disc_activations_real = self.discriminator(t2_ground_truth_batch)
            g_loss_extra = None
            if self.discriminator.use_perceptual:
                g_perceptual_loss = self.perceptual_loss(disc_activations_fake, disc_activations_real)
                self.log("g_perceptual_loss", g_perceptual_loss)
                g_loss_extra = g_perceptual_loss
            g_adv_loss = self.adversarial_loss(disc_output_fake, valid)
            self.log("g_adv_loss", g_adv_loss)
            g_recon_loss = self.reconstruction_loss(t2_generated_batch, t2_ground_truth_batch)
            self.log("g_recon_loss", g_recon_loss)
            g_loss = g_adv_loss + g_recon_loss
            if g_loss_extra is not None:
                g_loss = g_loss + g_loss_extra
            self.log("g_loss", g_loss)
            return g_loss
        if optimizer_idx == 1:                                              # :393-438
            valid = torch.full((t2_ground_truth_batch.shape[0], 1), float(self.hparams.one_sided_label_value),
                               device=dev, dtype=dt)
            real_loss = self.adversarial_loss(self.discriminator(t2_ground_truth_batch)[0], valid)
            fake = torch.zeros(t2_generated_batch.shape[0], 1, device=dev, dtype=dt)
            fake_loss = self.adversarial_loss(self.discriminator(t2_generated_batch)[0], fake)
            d_loss = (real_loss + fake_loss) / 2
            self.log("d_loss", d_loss)
            return d_loss

    def configure_optimizers(self):                                         # :440-447
        h = self.hparams
        return [FusedAdam(self.generator, lr=h.lr, betas=(h.b1, h.b2)),
                FusedAdam(self.discriminator, lr=h.lr, betas=(h.b1, h.b2))], []

    def fit_batch(self, batch, batch_idx, optimizers) -> Dict[str, torch.Tensor]:
        nets: List[_EngineModule] = [self.generator, self.discriminator]
        for idx, opt in enumerate(optimizers):
            other = nets[1 - idx]
            for p in other.parameters():
                p.requires_grad_(False)
            opt.zero_grad()
            loss = self.training_step(batch, batch_idx, idx)
            loss.backward()
            if self.ddp is not None:
                self.ddp.reduce_gradients(nets[idx], opt)
            opt.step()
            for p in other.parameters():
                p.requires_grad_(True)
        return dict(self.logged)
