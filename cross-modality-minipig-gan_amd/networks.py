"""The reference's module API on top of the HIP engine.

`CasNetGenerator(img_shape, n_unet_blocks=6)` and
`Discriminator(img_shape, use_perceptual=True)` keep the constructor / forward
signatures, attribute names (`.model`, `.model_conv`, `.model_linear`) and
state_dict keys of code/GAN/GAN_final.py:92-122,159-209 (and the MONAI 0.4.0
U-Net tree beneath the generator), so a reference
checkpoint's `generator.*` / `discriminator.*` entries load unchanged.
Keyword-only additions: `dimensions` (2|3), `norm` ("batch"|"instance"|"instance_affine"),
`channels`, `strides`, `device`.

The torch.nn leaf modules below are parameter CONTAINERS only: their own
forward is never called.  `forward` runs a pre-built plan of HIP kernels
(engine.py) inside one autograd node; parameter gradients are accumulated by
the kernels straight into the store's flat gradient buffer (the `.grad` of each
parameter is a view of it).  Without libmpgan_hip.so or a GPU this raises.
"""
from __future__ import annotations

import weakref
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .engine import DiscPlan, DiscPlanBF16, GeneratorPlan, ParamStore, plan_io

_CONV = {2: nn.Conv2d, 3: nn.Conv3d}
_CONVT = {2: nn.ConvTranspose2d, 3: nn.ConvTranspose3d}
_BN = {2: nn.BatchNorm2d, 3: nn.BatchNorm3d}
_IN = {2: nn.InstanceNorm2d, 3: nn.InstanceNorm3d}


class _Container(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: executed only by the HIP engine (call the top-level network)")


def _norm(norm: str, dims: int, ch: int) -> nn.Module:
    if norm == "batch":
        return _BN[dims](ch)
    if norm == "instance":                       # MONAI Norm.INSTANCE: torch's default, no gamma / beta
        return _IN[dims](ch)
    if norm == "instance_affine":                # same parameter set (and state_dict keys) as the batch variant
        return _IN[dims](ch, affine=True)
    raise ValueError(f"norm must be 'batch', 'instance' or 'instance_affine', got {norm!r}")


def _convolution(dims, cin, cout, stride, *, conv_only=False, transposed=False, norm="batch", k=3) -> nn.Sequential:
    pad = (k - 1) // 2
    seq = nn.Sequential()
    if transposed:
        seq.add_module("conv", _CONVT[dims](cin, cout, k, stride=stride, padding=pad, output_padding=stride - 1))
    else:
        seq.add_module("conv", _CONV[dims](cin, cout, k, stride=stride, padding=pad))
    if not conv_only:
        adn = nn.Sequential()
        adn.add_module("N", _norm(norm, dims, cout))
        adn.add_module("D", nn.Dropout(0.0))
        adn.add_module("A", nn.PReLU())
        seq.add_module("adn", adn)
    return seq


class ResidualUnit(_Container):
    def __init__(self, dims, cin, cout, stride, subunits, last_conv_only=False, norm="batch"):
        super().__init__()
        self.conv = nn.Sequential()
        self.residual = nn.Identity()
        sch, sst = cin, stride
        for su in range(max(1, subunits)):
            self.conv.add_module(f"unit{su:d}", _convolution(dims, sch, cout, sst, norm=norm,
                                                             conv_only=last_conv_only and su == subunits - 1))
            sch, sst = cout, 1
        if stride != 1 or cin != cout:
            rk, rp = (3, 1) if stride != 1 else (1, 0)
            self.residual = _CONV[dims](cin, cout, rk, stride, rp)


class SkipConnection(_Container):
    def __init__(self, submodule):
        super().__init__()
        self.submodule = submodule


class UNet(_Container):
    """Parameter tree of MONAI 0.4.0 UNet(num_res_units=2, kernel 3, PReLU)."""

    def __init__(self, dimensions=3, in_channels=1, out_channels=1, channels=(16, 32, 64, 128), strides=(2, 2, 2),
                 num_res_units=2, norm="batch"):
        super().__init__()
        if in_channels != 1 or out_channels != 1 or num_res_units != 2:
            raise ValueError("the HIP engine implements the reference's U-Net: 1->1 channels, 2 residual units")
        if any(s != 2 for s in list(strides)[:len(channels) - 1]):
            raise ValueError("the HIP engine implements stride-2 levels only (the reference's setting)")
        self.dimensions, self.channels, self.strides = dimensions, tuple(channels), tuple(strides)
        self.norm = norm
        d = dimensions

        def up(inc, outc, s, is_top):
            return nn.Sequential(_convolution(d, inc, outc, s, transposed=True, norm=norm),
                                 ResidualUnit(d, outc, outc, 1, 1, last_conv_only=is_top, norm=norm))

        def block(inc, outc, chs, sts, is_top):
            c, s = chs[0], sts[0]
            if len(chs) > 2:
                sub, upc = block(c, c, chs[1:], sts[1:], False), c * 2
            else:
                sub, upc = ResidualUnit(d, c, chs[1], 1, 2, norm=norm), c + chs[1]
            return nn.Sequential(ResidualUnit(d, inc, c, s, 2, norm=norm), SkipConnection(sub), up(upc, outc, s, is_top))

        self.model = block(1, 1, tuple(channels), tuple(strides), True)


class _Lease:
    """Returns a plan to its pool when the autograd node that borrowed it dies."""

    def __init__(self, plan):
        self.plan = plan
        plan.busy = True

    def release(self):
        if self.plan is not None:
            self.plan.busy = False
            self.plan = None

    def __del__(self):
        self.release()


class _EngineModule(nn.Module):
    """Shared plumbing: flat parameter store, plan pool, grad anchor."""

    def __init__(self):
        super().__init__()
        self._store: Optional[ParamStore] = None
        self._plans: Dict[tuple, list] = {}
        self._anchor: Optional[torch.Tensor] = None

    # the store is created lazily on first use so construction / load_state_dict / .cuda() work as usual
    @property
    def store(self) -> ParamStore:
        if self._store is None:
            p = next(self.parameters())
            if not p.is_cuda:
                raise RuntimeError(f"{type(self).__name__} runs only on an MI355X: move it to a CUDA/HIP device first "
                                   "(there is no CPU path)")
            ops.lib()  # raises if libmpgan_hip.so is missing
            self._store = ParamStore(self)
            self._register_extra(self._store)
            self._anchor = torch.zeros(1, device=p.device, requires_grad=True)
        return self._store

    def _register_extra(self, store):
        pass

    def _apply(self, fn, *a, **k):
        # .to()/.cuda()/.float() re-create parameter storage: drop the flat store and the plans
        self._store = None
        self._plans = {}
        return super()._apply(fn, *a, **k)

    def _acquire(self, key, make):
        pool = self._plans.setdefault(key, [])
        for pl in pool:
            if not pl.busy:
                return pl
        pl = make()
        pool.append(pl)
        return pl

    def _params_require_grad(self) -> bool:
        return any(p.requires_grad for p in self.parameters())

    def zero_grad(self, set_to_none: bool = False):  # keep the flat gradient views attached
        if self._store is not None:
            self._store.flat_grad.zero_()
            self._store.attach_grads()
        else:
            super().zero_grad(set_to_none=False)


def _spatial(x: torch.Tensor, dims: int) -> Tuple[int, ...]:
    if x.dim() != dims + 2 or x.shape[1] != 1:
        raise ValueError(f"expected a (B,1,{'D,' if dims == 3 else ''}H,W) tensor, got {tuple(x.shape)}")
    if x.dtype != torch.float32 or not x.is_cuda:
        raise ValueError(f"expected a float32 CUDA tensor, got {x.dtype} on {x.device} (no CPU path)")
    return tuple(x.shape[2:])


class _GenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, gen):
        need_bwd = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        spatial = _spatial(x, gen.dimensions)
        n = x.shape[0]
        mm = gen.matmul_dtype == "bf16"
        key = (n, spatial, need_bwd, bool(ctx.needs_input_grad[0]), gen.training, mm)
        store = gen.store
        plan = gen._acquire(key, lambda: GeneratorPlan(gen, store, n, spatial, want_backward=need_bwd,
                                                      want_input_grad=bool(ctx.needs_input_grad[0]),
                                                      instance=gen.norm.startswith("instance"), training=gen.training,
                                                      mm_bf16=mm))
        lease = _Lease(plan)
        # C == 1: NC(D)HW and channels-last coincide, so the kernels read the caller's tensor and write the returned
        # one in place (engine.IoSlots); a misaligned or strided tensor goes through the plan's staging buffer.
        io = plan_io(plan)
        if not io.bind("x", x):
            plan.x_in.view(-1).copy_(x.reshape(-1))
        y = torch.empty_like(x)
        direct_y = io.bind("y", y)
        try:
            plan.fwd.run()
        finally:
            if not need_bwd:
                io.reset()
        if not direct_y:
            y.view(-1).copy_(plan.y.view(-1))
        if need_bwd:
            ctx.lease = lease
            ctx.shape = x.shape
            ctx.save_for_backward(x, y)          # read again by the backward (first conv's wgrad, tanh'): autograd's
            store.attach_grads()                 # version check guards them against in-place edits
        else:
            lease.release()
        return y

    @staticmethod
    def backward(ctx, gy):
        plan = ctx.lease.plan
        _ = ctx.saved_tensors                    # raises if x or y were modified in place since the forward
        io = plan_io(plan)
        gy = gy.contiguous()
        if not io.bind("g_y", gy):
            plan.g_y.view(-1).copy_(gy.reshape(-1))
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(ctx.shape, device=gy.device, dtype=torch.float32)
            direct = io.bind("g_x", gx)
        try:
            plan.bwd.run()
        finally:
            io.reset()
        if gx is not None and not direct:
            gx.view(-1).copy_(plan.g_x.view(-1))
        ctx.lease.release()
        return gx, None, None


class CasNetGenerator(_EngineModule):
    """code/GAN/GAN_final.py:92-122 (variant B: test_runs/GAN.py:94-129 via
    n_unet_blocks=4, channels=(32,64,128,256), strides=(2,2,2,2))."""

    def __init__(self, img_shape, n_unet_blocks=6, *, dimensions=3, norm="batch", channels=(16, 32, 64, 128),
                 strides=(2, 2, 2), device=None, matmul_dtype="f32"):
        super().__init__()
        if matmul_dtype not in ("f32", "bf16"):
            raise ValueError(f"matmul_dtype must be 'f32' or 'bf16', got {matmul_dtype!r}")
        self.img_shape = img_shape
        self.dimensions, self.norm = dimensions, norm
        # "bf16": the matrix products of the MFMA-served convs take bf16-rounded operands and accumulate in fp32
        # (config C5); parameters, activations, statistics, gradients and Adam stay fp32.  May be switched on a
        # live module: plans are keyed by it.
        self.matmul_dtype = matmul_dtype
        nets = [UNet(dimensions, 1, 1, channels, strides, 2, norm) for _ in range(n_unet_blocks)]
        nets.append(nn.Tanh())
        self.model = nn.Sequential(*nets)
        if device is not None:
            self.to(device)

    def forward(self, x):
        """Train mode: batch statistics + running-stat updates (what the reference's training
        loop runs).  Eval mode (code/GAN/inferrence.py:97-110,169-170: `.eval()`, `no_grad`):
        BatchNorm uses its running statistics; no gradients are offered in eval mode."""
        store = self.store
        if not self.training and torch.is_grad_enabled() and (x.requires_grad or self._params_require_grad()):
            with torch.no_grad():
                return _GenFn.apply(x.contiguous(), self._anchor.detach(), self)
        self._anchor.requires_grad_(self.training and torch.is_grad_enabled() and self._params_require_grad())
        return _GenFn.apply(x.contiguous(), self._anchor, self)


class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, disc):
        want_in, want_par = bool(ctx.needs_input_grad[0]), bool(ctx.needs_input_grad[1])
        need_bwd = want_in or want_par
        spatial = _spatial(x, disc.dimensions)
        n = x.shape[0]
        store = disc.store
        key = (n, spatial, need_bwd, want_in, want_par)
        plan_cls = DiscPlanBF16 if disc.storage_dtype == "bf16" else DiscPlan
        plan = disc._acquire(key, lambda: plan_cls(disc, store, n, spatial, want_backward=need_bwd,
                                                   want_input_grad=want_in, want_param_grads=want_par))
        lease = _Lease(plan)
        io = plan_io(plan)
        if not io.bind("x", x):                  # (see _GenFn: the first conv reads the caller's tensor in place)
            plan.x_in.view(-1).copy_(x.reshape(-1))
        try:
            plan.fwd.run()
        finally:
            if not need_bwd:
                io.reset()
        prob = plan.prob.view(n, 1).clone()
        if need_bwd:
            ctx.lease = lease
            ctx.shape = x.shape
            ctx.save_for_backward(x)
            if want_par:
                store.attach_grads()
        else:
            lease.release()
        return prob

    @staticmethod
    def backward(ctx, gprob):
        plan = ctx.lease.plan
        _ = ctx.saved_tensors
        io = plan_io(plan)
        plan.g_prob.copy_(gprob.reshape(-1))
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(ctx.shape, device=gprob.device, dtype=torch.float32)
            direct = io.bind("g_x", gx)
        try:
            plan.bwd.run()
        finally:
            io.reset()
        if gx is not None and not direct:
            gx.view(-1).copy_(plan.g_x.view(-1))
        ctx.lease.release()
        return gx, None, None


def disc_feature_sizes(spatial: Sequence[int]):
    sizes = [tuple(spatial)]
    for k, s in ((3, 1), (3, 1), (4, 2), (4, 2)):
        sizes.append(tuple((n - k) // s + 1 for n in sizes[-1]))
    return sizes


class Discriminator(_EngineModule):
    """code/GAN/GAN_final.py:159-209.  The reference hard-codes
    Linear(256*29*29*29, 1) for its 128^3 input (:201); in_features here is
    computed from img_shape (the same number for 128^3)."""

    def __init__(self, img_shape, use_perceptual=True, *, dimensions=3, device=None, storage_dtype="f32"):
        super().__init__()
        if storage_dtype not in ("f32", "bf16"):
            raise ValueError(f"storage_dtype must be 'f32' or 'bf16', got {storage_dtype!r}")
        self.use_perceptual = use_perceptual
        self.img_shape = img_shape
        self.dimensions = dimensions
        self.storage_dtype = storage_dtype      # activations / packed weights in HBM; parameters, statistics, Adam: fp32
        Cv, Bn = _CONV[dimensions], _BN[dimensions]
        self.model_conv = nn.Sequential(
            Cv(1, 64, 3, 1), Bn(64), nn.LeakyReLU(0.2, inplace=True),
            Cv(64, 128, 3, 1), Bn(128), nn.LeakyReLU(0.2, inplace=True),
            Cv(128, 256, 4, 2), Bn(256), nn.LeakyReLU(0.2, inplace=True),
            Cv(256, 256, 4, 2), Bn(256), nn.LeakyReLU(0.2, inplace=True))
        spatial = tuple(img_shape)[-dimensions:]
        last = disc_feature_sizes(spatial)[-1]
        if min(last) < 1:
            raise ValueError(f"img_shape {img_shape} too small for the discriminator")
        self._last = last
        self.model_linear = nn.Sequential(nn.Flatten(), nn.Linear(256 * int(np.prod(last)), 1), nn.Sigmoid())
        if device is not None:
            self.to(device)

    def _register_extra(self, store):
        lin = self.model_linear[1]
        store.register_conv(lin, cout=1, cin=256, taps=lin.in_features // 256)

    def forward(self, img):
        if not self.training:
            raise NotImplementedError("eval-mode discriminator is not built (the reference never evaluates D)")
        store = self.store
        self._anchor.requires_grad_(torch.is_grad_enabled() and self._params_require_grad())
        return _DiscFn.apply(img.contiguous(), self._anchor, self)


# --------------------------------------------------------------------------
# variant B: patch discriminator with perceptual taps (test_runs/GAN.py:136-198)
# --------------------------------------------------------------------------
class TapSet:
    """The 16 activation taps of one PatchDiscriminator pass.  The reference returns
    16 cloned tensors; here they stay implicit (raw conv outputs + norm vectors inside the
    plan) and `perceptual_loss` evaluates their L1 terms and gradients from those.
    `handle` is the autograd edge standing for all of them."""

    KEYS = tuple(range(16))

    def __init__(self, handle, lease):
        self.handle, self.lease = handle, lease

    @property
    def plan(self):
        return self.lease.plan

    def materialize(self, key: int) -> torch.Tensor:
        """The tap as the reference would hold it (NC(D)HW, detached) -- for inspection / tests."""
        plan, dims = self.plan, self.plan.dims
        n = plan.n
        if key < 12:
            i, kind = divmod(key, 3)
            z = plan.zs[i]
            if kind == 0:
                t = z
            else:
                pro = plan.lrelu(plan.nbs[i]) if kind == 2 else plan.nbs[i].prologue(ops.ACT_NONE)
                t = torch.empty_like(z)
                ops.norm_act_add(z, pro, None, None, t)
            t = t.permute(0, 4, 1, 2, 3)
            return (t.squeeze(2) if dims == 2 else t).contiguous()
        if key == 12:
            return self.materialize(11).flatten(1)
        if key == 13:
            return plan.h.reshape(n, -1).clone()
        if key == 14:
            return plan.logit.reshape(n, 1).clone()
        return plan.prob.reshape(n, 1).clone()


class _TapFn(torch.autograd.Function):
    """One perceptual tap as a real tensor (what the reference's Discriminator.forward returns with
    `.clone()`, test_runs/GAN.py:183-198), connected to the discriminator pass through the TapSet's handle:
    its gradient is deposited into the plan's external-tap buffers, which the pass's backward then feeds
    through the BatchNorm / conv chain (PatchDiscPlan.backward_program_ext)."""

    @staticmethod
    def forward(ctx, handle, tapset, key):
        ctx.tapset, ctx.key = tapset, key
        return tapset.materialize(key)

    @staticmethod
    def backward(ctx, g):
        ctx.tapset.plan.deposit_tap_grad(ctx.key, g)
        return torch.zeros(1, device=g.device), None, None


class TapDict(dict):
    """The reference's `perceptual_dict` (16 int keys).  Values are produced on demand: indexing a key
    materialises that tap as a differentiable tensor (`_TapFn`), so the reference's own `perceptual_loss` body
    (test_runs/GAN.py:288-298: `F.l1_loss(y_activations[key], y_hat_activations[key])`) runs unchanged on it.
    `mpgan_amd.gan_patch.perceptual_loss`, given two TapDicts, never materialises anything (fused path)."""

    def __init__(self, tapset: TapSet):
        super().__init__({k: None for k in TapSet.KEYS})
        self.tapset = tapset

    def __getitem__(self, key):
        if key not in TapSet.KEYS:
            raise KeyError(key)
        v = super().__getitem__(key)
        if v is None:
            v = _TapFn.apply(self.tapset.handle, self.tapset, key)
            super().__setitem__(key, v)
        return v

    def get(self, key, default=None):
        return self[key] if key in TapSet.KEYS else default

    def values(self):
        return [self[k] for k in TapSet.KEYS]

    def items(self):
        return [(k, self[k]) for k in TapSet.KEYS]


class _PatchDiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, disc):
        want_in, want_par = bool(ctx.needs_input_grad[0]), bool(ctx.needs_input_grad[1])
        need_bwd = want_in or want_par
        spatial = _spatial(x, disc.dimensions)
        n = x.shape[0]
        store = disc.store
        from .engine import PatchDiscPlan
        key = ("patch", n, spatial, need_bwd, want_in, want_par)
        plan = disc._acquire(key, lambda: PatchDiscPlan(disc, store, n, spatial, want_backward=need_bwd,
                                                        want_input_grad=want_in, want_param_grads=want_par))
        lease = _Lease(plan)
        plan.clear_taps()
        plan.x_in.view(-1).copy_(x.reshape(-1))
        plan.fwd.run()
        prob = plan.prob.view(n, 1).clone()
        handle = torch.zeros(1, device=x.device)
        ctx.lease = lease
        ctx.shape = x.shape
        ctx.need_bwd = need_bwd
        disc._last_lease = lease
        if want_par:
            store.attach_grads()
        return prob, handle

    @staticmethod
    def backward(ctx, gprob, ghandle):
        plan = ctx.lease.plan
        if gprob is None:
            plan.g_prob.zero_()
        else:
            plan.g_prob.copy_(gprob.reshape(-1))
        if getattr(plan, "ext_used", False):
            plan.backward_program_ext(getattr(plan, "peer", None)).run()
            plan.clear_ext()
        else:
            plan.backward_program(getattr(plan, "peer", None)).run()
        gx = plan.g_x.view(ctx.shape).clone() if ctx.needs_input_grad[0] else None
        plan.peer = None
        plan.peer_lease = None
        ctx.lease.release()
        return gx, None, None


class PatchDiscriminator(_EngineModule):
    """test_runs/GAN.py:136-198: returns (validity, perceptual_dict)."""

    def __init__(self, img_shape, use_perceptual=True, *, dimensions=3, patch=16, device=None):
        super().__init__()
        self.use_perceptual = use_perceptual
        self.img_shape = img_shape
        self.dimensions = dimensions
        Cv, Bn = _CONV[dimensions], _BN[dimensions]
        self.model_conv = nn.Sequential(
            Cv(1, 64, 3, 1), Bn(64), nn.LeakyReLU(0.2, inplace=True),
            Cv(64, 128, 3, 1), Bn(128), nn.LeakyReLU(0.2, inplace=True),
            Cv(128, 256, 3, 1), Bn(256), nn.LeakyReLU(0.2, inplace=True),
            Cv(256, 512, 3, 1), Bn(512), nn.LeakyReLU(0.2, inplace=True))
        last = patch - 8
        if last < 1:
            raise ValueError("patch too small")
        self.model_linear = nn.Sequential(nn.Flatten(), nn.Linear(512 * last ** dimensions, 64), nn.Linear(64, 1),
                                          nn.Sigmoid())
        self._last_lease = None
        if device is not None:
            self.to(device)

    def _register_extra(self, store):
        lin1, lin2 = self.model_linear[1], self.model_linear[2]
        store.register_conv(lin1, cout=lin1.out_features, cin=512, taps=lin1.in_features // 512, tco=True)
        store.register_conv(lin2, cout=1, cin=lin1.out_features, taps=1)

    def forward(self, x):
        if not self.training:
            raise NotImplementedError("eval-mode discriminator is not built (the reference never evaluates D)")
        store = self.store
        self._anchor.requires_grad_(torch.is_grad_enabled() and self._params_require_grad())
        prob, handle = _PatchDiscFn.apply(x.contiguous(), self._anchor, self)
        taps = TapDict(TapSet(handle, self._last_lease)) if self.use_perceptual else {}
        self._last_lease = None
        return prob, taps
