"""The reference's GAN LightningModule without Lightning: same methods, same
two-optimizer step (code/GAN/GAN_final.py:212-317; Lightning 1.2.1 loop restated
in SURVEY.md Appendix B), every FLOP in HIP kernels.
"""
from __future__ import annotations

import re
import types
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from .networks import CasNetGenerator, Discriminator, _EngineModule


# --------------------------------------------------------------------------
# loss hooks (autograd nodes around the loss kernels)
# --------------------------------------------------------------------------
class _BCEFn(torch.autograd.Function):
    """F.binary_cross_entropy(y_hat, y) (mean), log clamped at -100."""

    @staticmethod
    def forward(ctx, y_hat, y):
        p = y_hat.contiguous()
        t = y.contiguous()
        loss = torch.empty((), device=p.device)
        ops.bce_forward(p, t, loss)
        ctx.save_for_backward(p, t)
        return loss

    @staticmethod
    def backward(ctx, gout):
        p, t = ctx.saved_tensors
        dprob = torch.empty_like(p)
        ops.bce_backward(p, t, gout.contiguous(), dprob)
        return dprob, None


class _L1Fn(torch.autograd.Function):
    """F.l1_loss(y_hat, y) (mean); backward = gout * sign(y_hat - y) / numel."""

    @staticmethod
    def forward(ctx, y_hat, y):
        a, b = y_hat.contiguous(), y.contiguous()
        loss = torch.empty((), device=a.device)
        part = torch.empty(ops.l1_partials(), device=a.device)
        grad = torch.empty_like(a) if (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) else None
        ops.l1_loss(a, b, part, loss, grad, 1.0)
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, gout):
        g = torch.empty_like(ctx.grad)          # a fresh tensor: a second backward (retain_graph) sees the unscaled sign
        ops.scale_by_device_scalar(ctx.grad, gout.contiguous(), g)
        gb = ops.axpby(g, -1.0, None, 0.0, torch.empty_like(g)) if ctx.needs_input_grad[1] else None   # d/dy = -d/dy_hat
        return (g if ctx.needs_input_grad[0] else None), gb


class _AxpbyFn(torch.autograd.Function):
    """alpha*a + beta*b on device scalars (the loss sums of training_step) through the library's own
    pointwise kernel, so that no torch elementwise kernel sits on the step path."""

    @staticmethod
    def forward(ctx, a, alpha, b, beta):
        ctx.alpha, ctx.beta = alpha, beta
        out = torch.empty_like(a)
        ops.axpby(a.contiguous(), alpha, b.contiguous(), beta, out)
        return out

    @staticmethod
    def backward(ctx, gout):
        gout = gout.contiguous()
        ga = ops.axpby(gout, ctx.alpha, None, 0.0, torch.empty_like(gout)) if ctx.needs_input_grad[0] else None
        gb = ops.axpby(gout, ctx.beta, None, 0.0, torch.empty_like(gout)) if ctx.needs_input_grad[2] else None
        return ga, None, gb, None


def scalar_axpby(a, alpha, b, beta):
    return _AxpbyFn.apply(a, float(alpha), b, float(beta))


def adversarial_loss(y_hat, y):
    """code/GAN/GAN_final.py:244-245."""
    return _BCEFn.apply(y_hat, y)


def reconstruction_loss(y_hat, y):
    """code/GAN/GAN_final.py:247-248."""
    return _L1Fn.apply(y_hat, y)


# --------------------------------------------------------------------------
# optimiser
# --------------------------------------------------------------------------
class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(lr, betas, eps=1e-8, weight_decay=0) as ONE kernel over
    the network's flat parameter buffer (code/GAN/GAN_final.py:306-307).
    `grad_scale` (e.g. 1/world_size after a sum all-reduce) is applied on load."""

    def __init__(self, net: _EngineModule, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.net = net
        super().__init__(list(net.parameters()), dict(lr=lr, betas=betas, eps=eps))
        self._version = -1
        self.step_count = 0
        self.grad_scale = 1.0

    def _state(self):
        store = self.net.store
        if self._version != store.version:
            # the flat layout is a function of the module tree only: a re-created store (.to()/.cuda()/.float()
            # after the first step) keeps the moments; a different layout restarts Adam as a whole
            old_m, old_v = getattr(self, "exp_avg", None), getattr(self, "exp_avg_sq", None)
            self.exp_avg = torch.zeros_like(store.flat)
            self.exp_avg_sq = torch.zeros_like(store.flat)
            if old_m is not None and old_m.numel() == store.flat.numel():
                self.exp_avg.copy_(old_m)
                self.exp_avg_sq.copy_(old_v)
            elif old_m is not None:
                import warnings
                warnings.warn("FusedAdam: parameter layout changed; moments and step count reset")
                self.step_count = 0
            self._version = store.version
        return store

    # torch.optim.Adam's state_dict format (per-parameter step / exp_avg / exp_avg_sq), so that a
    # Lightning-style 'optimizer_states' entry round-trips and a torch Adam state loads here
    def state_dict(self):
        store = self._state()
        params = list(self.net.parameters())
        state = {}
        if self.step_count > 0:
            for i, p in enumerate(params):
                o = store.offset(p)
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[o:o + p.numel()].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + p.numel()].view(p.shape).clone()}
        g = self.param_groups[0]
        return {"state": state,
                "param_groups": [{"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": 0,
                                  "amsgrad": False, "params": list(range(len(params)))}]}

    def load_state_dict(self, sd):
        store = self._state()
        params = list(self.net.parameters())
        g = sd["param_groups"][0]
        # validate everything before touching the moments: a checkpoint of a differently shaped network must leave
        # this optimizer as it was
        names = [n for n, _ in self.net.named_parameters()]
        for i, st in sd["state"].items():
            if not 0 <= int(i) < len(params):
                raise ValueError(f"FusedAdam.load_state_dict: state entry {i} but the network has {len(params)} parameters")
            p = params[int(i)]
            for key in ("exp_avg", "exp_avg_sq"):
                if key not in st or st[key].numel() != p.numel():
                    got = tuple(st[key].shape) if key in st else None
                    raise ValueError(f"FusedAdam.load_state_dict: {key} of parameter {int(i)} ({names[int(i)]}, shape "
                                     f"{tuple(p.shape)}) has shape {got}")
        self.param_groups[0].update(lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"])
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.step_count = 0
        for i, st in sd["state"].items():
            p = params[int(i)]
            o = store.offset(p)
            self.exp_avg[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
            self.step_count = max(self.step_count, int(float(st["step"])))

    def zero_grad(self, set_to_none: bool = False):
        store = self._state()
        store.flat_grad.zero_()
        store.attach_grads()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        store = self._state()
        g = self.param_groups[0]
        self.step_count += 1
        ops.adam_step(store.flat, store.flat_grad, self.exp_avg, self.exp_avg_sq, g["lr"], g["betas"][0],
                      g["betas"][1], g["eps"], self.step_count, self.grad_scale)
        store.touch()                               # the packed weights are stale now
        return loss


# --------------------------------------------------------------------------
# the GAN module
# --------------------------------------------------------------------------
class GAN(nn.Module):
    """code/GAN/GAN_final.py:212-317.  `training_step(batch, batch_idx,
    optimizer_idx)` has the reference's body; `fit_batch` is Lightning's
    per-batch loop (toggle_optimizer -> zero_grad -> training_step -> backward
    -> optimizer.step for optimizer 0 (G) then 1 (D))."""

    def __init__(self, channels, width, height, depth=None, latent_dim: int = 100, d_lr: float = 0.0005,
                 g_lr: float = 0.0005, b1: float = 0.5, b2: float = 0.999, batch_size: int = 64, example_data=None,
                 one_sided_label_value=0.9, *, dimensions: Optional[int] = None, norm: str = "batch",
                 n_unet_blocks: int = 6, unet_channels=(16, 32, 64, 128), unet_strides=(2, 2, 2), device="cuda",
                 storage_dtype: str = "f32", **kwargs):
        super().__init__()
        if dimensions is None:
            dimensions = 3 if depth is not None else 2
        self.hparams = types.SimpleNamespace(latent_dim=latent_dim, g_lr=g_lr, d_lr=d_lr, b1=b1, b2=b2,
                                             batch_size=batch_size, one_sided_label_value=one_sided_label_value)
        data_shape = (channels, width, height) + ((depth,) if dimensions == 3 else ())
        # storage_dtype="bf16" is config C5: the discriminator stores its activations in bf16 and the generator's
        # matrix products take bf16 operands (fp32 storage: its launches are matrix-bound, not byte-bound)
        self.generator = CasNetGenerator(data_shape, n_unet_blocks, dimensions=dimensions, norm=norm,
                                         channels=unet_channels, strides=unet_strides, device=device,
                                         matmul_dtype="bf16" if storage_dtype == "bf16" else "f32")
        self.discriminator = Discriminator(data_shape, dimensions=dimensions, device=device,
                                           storage_dtype=storage_dtype)
        self.logged: Dict[str, torch.Tensor] = {}
        self.ddp = None  # set by parallel.DataParallelGAN

    def forward(self, x):
        return self.generator(x)

    def adversarial_loss(self, y_hat, y):
        return adversarial_loss(y_hat, y)

    def reconstruction_loss(self, y_hat, y):
        return reconstruction_loss(y_hat, y)

    def log(self, name, value, **kw):
        """Lightning's self.log: scalars stay on the device (no per-step sync)."""
        self.logged[name] = value.detach()

    def training_step(self, batch, batch_idx, optimizer_idx):
        t1w_images, t2w_images = batch["t1w"], batch["t2w"]
        if optimizer_idx == 0:                      # GAN_final.py:254-273
            generated_imgs = self(t1w_images)
            self.generated_imgs = generated_imgs
            valid = torch.ones(t1w_images.shape[0], 1, device=t1w_images.device, dtype=t1w_images.dtype)
            g_adv_loss = self.adversarial_loss(self.discriminator(generated_imgs), valid)
            self.log("g_adv_loss", g_adv_loss)
            g_recon_loss = self.reconstruction_loss(generated_imgs, t2w_images)
            self.log("g_recon_loss", g_recon_loss)
            g_loss = scalar_axpby(g_adv_loss, 1.0, g_recon_loss, 1.0)
            self.log("g_loss", g_loss)
            return g_loss
        if optimizer_idx == 1:                      # GAN_final.py:276-296
            valid = torch.full((t1w_images.shape[0], 1), float(self.hparams.one_sided_label_value),
                               device=t1w_images.device, dtype=t1w_images.dtype)
            # (D(real) and G(t1w) are independent, and rounds 1-3 carried an opt-in mode that ran the generator's
            #  forward on a high-priority stream underneath D(real).  It measured nothing beyond run-to-run spread
            #  -- 59.1 vs 58.6 ms at C3, 174.3 vs 175.8 ms at C5 -- while stretching D's kernels up to 3x, and its first
            #  use in a process once ended in a host SIGSEGV inside hipStreamWaitEvent on that stream: removed in
            #  round 4)
            real_loss = self.adversarial_loss(self.discriminator(t2w_images), valid)
            generated = self(t1w_images).detach()
            fake = torch.zeros(t1w_images.shape[0], 1, device=t1w_images.device, dtype=t1w_images.dtype)
            fake_loss = self.adversarial_loss(self.discriminator(generated), fake)
            d_loss = scalar_axpby(real_loss, 0.5, fake_loss, 0.5)
            self.log("d_loss", d_loss)
            return d_loss

    def configure_optimizers(self):                 # GAN_final.py:298-308
        h = self.hparams
        opt_g = FusedAdam(self.generator, lr=h.g_lr, betas=(h.b1, h.b2))
        opt_d = FusedAdam(self.discriminator, lr=h.d_lr, betas=(h.b1, h.b2))
        return [opt_g, opt_d], []

    def fit_batch(self, batch, batch_idx, optimizers) -> Dict[str, torch.Tensor]:
        # (Measured and dropped: pulling the D step's real branch forward onto a background stream
        #  underneath the generator's backward gains 0.7 ms of 60 -- matrix-bound kernels and chains
        #  of small launches contend for the same CUs -- and stretches D's forward kernels by 20 %.)
        nets: List[_EngineModule] = [self.generator, self.discriminator]
        for idx, opt in enumerate(optimizers):
            other = nets[1 - idx]
            for p in other.parameters():            # toggle_optimizer
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


_UNSUPPORTED_GLOBAL = re.compile(r"GLOBAL ([\w\.]+) was not an allowed global")


def _inert_stub(full_path: str):
    """A placeholder class carrying the module / name of a global the checkpoint's pickle refers to
    (e.g. `pytorch_lightning.callbacks.model_checkpoint.ModelCheckpoint`, which Lightning 1.2.1 uses as
    a KEY of checkpoint['callbacks']).  Nothing is imported and nothing from the file runs: the
    weights-only unpickler resolves the name to this class, whose construction ignores its arguments.
    Dict-like names come back as plain dicts so that their items can still be read."""
    module, _, name = full_path.rpartition(".")
    ns = {"__module__": module, "__init__": lambda self, *a, **k: None, "__setstate__": lambda self, state: None}
    if name.endswith("Dict") or name.endswith("dict"):
        ns["__new__"] = staticmethod(lambda cls, *a, **k: {})
    return type(name, (object,), ns)


def load_checkpoint_blob(path: str):
    """`torch.load(path, weights_only=True)` that also accepts the reference's real checkpoint format:
    Lightning writes class objects and small helper types into the pickle next to the tensors, which the
    weights-only unpickler refuses by name.  Each refused name is mapped to an inert stub (above) through
    `torch.serialization.safe_globals` and the load is retried; the stubs never leave this function's scope."""
    stubs = []
    for _ in range(64):
        try:
            with torch.serialization.safe_globals(stubs):
                return torch.load(path, map_location="cpu", weights_only=True)
        except Exception as e:          # pickle.UnpicklingError from the weights-only unpickler
            m = _UNSUPPORTED_GLOBAL.search(str(e))
            if m is None or any(f"{c.__module__}.{c.__name__}" == m.group(1) for c in stubs):
                raise
            stubs.append(_inert_stub(m.group(1)))
    raise RuntimeError(f"{path}: more than 64 distinct non-tensor globals in the pickle")


def load_reference_checkpoint(model: nn.Module, path: str, strict: bool = False, optimizers=None):
    """Load a Lightning checkpoint written by the reference's trainer (code/GAN/inferrence.py:97-106:
    `torch.load(ckpt)['state_dict']`, `load_state_dict(..., strict=False)`) into `model` (a GAN, or a
    bare generator / discriminator: then the `generator.` / `discriminator.` prefix is stripped).
    The file is read with `weights_only=True`: nothing in it is executed (see load_checkpoint_blob for
    how the trainer's non-tensor entries -- `callbacks` keyed by the ModelCheckpoint class,
    `hyper_parameters`, `optimizer_states` -- are tolerated).  `optimizers` (the list
    `configure_optimizers` returned) additionally restores Adam's moments from `optimizer_states`."""
    blob = load_checkpoint_blob(path)
    sd = blob["state_dict"] if isinstance(blob, dict) and "state_dict" in blob else blob
    if not isinstance(model, GAN):
        want = set(model.state_dict().keys())
        for prefix in ("generator.", "discriminator."):
            sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
            if sub and set(sub) & want:
                sd = sub
                break
    res = model.load_state_dict(sd, strict=strict)
    if optimizers is not None and isinstance(blob, dict) and blob.get("optimizer_states"):
        states = blob["optimizer_states"]
        if len(states) != len(optimizers):
            raise ValueError(f"{path}: {len(states)} optimizer_states for {len(optimizers)} optimizers "
                             "(configure_optimizers order: generator, discriminator)")
        for opt, st in zip(optimizers, states):
            opt.load_state_dict(st)
    return res
