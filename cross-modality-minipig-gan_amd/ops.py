"""Thin Python wrappers over the C ABI: shape/pitch checks on the host, raw
device pointers and the current torch stream into the HIP library.

Activations are channels-last tensors of shape (N, D, H, W, C) whose last
dimension may be a slice of a wider buffer (pitch = stride of W).  Nothing in
here computes: every function ends in exactly one C call.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import torch

from ._lib import ConvGeomC, NormFoldC, PeerTapsC, PrologueC, check, lib

ACT_NONE = 0
ACT_LEAKY = 1


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _cl(t: torch.Tensor, what: str) -> Tuple[int, int, int]:
    """Validate a channels-last (N,D,H,W,C) view; return (n, pixels/sample, ld)."""
    if t.dim() != 5 or t.dtype != torch.float32 or not t.is_cuda:
        raise ValueError(f"{what}: need a CUDA float32 (N,D,H,W,C) tensor, got {tuple(t.shape)} {t.dtype} {t.device}")
    n, d, h, w, c = t.shape
    st = t.stride()
    # pitch from the innermost dimension that actually has extent (size-1 dims carry arbitrary strides)
    if w > 1:
        ld = st[3]
    elif h > 1:
        ld = st[2]
    elif d > 1:
        ld = st[1]
    elif n > 1:
        ld = st[0]
    else:
        ld = c
    want = (d * h * w * ld, h * w * ld, w * ld, ld, 1)
    ok = ld >= c and all(sz == 1 or a == b for sz, a, b in zip(t.shape, st, want))
    if not ok:
        raise ValueError(f"{what}: not a pixel-contiguous channels-last view: shape {tuple(t.shape)} strides {st}")
    return n, d * h * w, ld


@dataclass(frozen=True)
class ConvGeom:
    """Geometry of one nn.ConvNd / nn.ConvTransposeNd (depth dim unused in 2-D)."""
    n: int
    in_dhw: Tuple[int, int, int]
    cin: int
    cout: int
    k: Tuple[int, int, int]
    stride: Tuple[int, int, int]
    pad: Tuple[int, int, int]
    transposed: bool = False
    out_pad: Tuple[int, int, int] = (0, 0, 0)
    mm_bf16: bool = False      # MPGAN_CONV_MM_BF16: matrix operands rounded to bf16, fp32 storage and accumulation
    min_blocks: int = 0        # launch size from which the big-tile forms serve this conv (0: the library's 1024)

    @property
    def out_dhw(self) -> Tuple[int, int, int]:
        if not self.transposed:
            return tuple((i + 2 * p - k) // s + 1 for i, p, k, s in zip(self.in_dhw, self.pad, self.k, self.stride))
        return tuple((i - 1) * s - 2 * p + k + op
                     for i, p, k, s, op in zip(self.in_dhw, self.pad, self.k, self.stride, self.out_pad))

    @property
    def taps(self) -> int:
        return self.k[0] * self.k[1] * self.k[2]

    def c(self) -> ConvGeomC:
        g = ConvGeomC()
        g.n = self.n
        g.in_dhw[:] = self.in_dhw
        g.out_dhw[:] = self.out_dhw
        g.cin, g.cout = self.cin, self.cout
        g.k[:] = self.k
        g.stride[:] = self.stride
        g.pad[:] = self.pad
        g.transposed = 1 if self.transposed else 0
        g.flags = 1 if self.mm_bf16 else 0
        g.min_blocks = int(self.min_blocks)
        return g

    def with_n(self, n: int) -> "ConvGeom":
        return dataclasses.replace(self, n=n)


@dataclass
class Prologue:
    """a = act(z*scale + shift) applied on load (norm + PReLU/LeakyReLU of the producer)."""
    scale: torch.Tensor
    shift: torch.Tensor
    n_stride: int = 0
    act: int = ACT_NONE
    slope: float = 1.0
    slope_t: Optional[torch.Tensor] = None  # device scalar (PReLU alpha)

    def c(self) -> PrologueC:
        p = PrologueC()
        p.scale = self.scale.data_ptr()
        p.shift = self.shift.data_ptr()
        p.n_stride = self.n_stride
        p.act = self.act
        p.slope = self.slope
        p.slope_ptr = None if self.slope_t is None else self.slope_t.data_ptr()
        return p


def _pro(p: Optional[Prologue]):
    return None if p is None else C.byref(p.c())


@dataclass
class PeerTaps:
    """The other pass of a perceptual-loss pair (see mpgan_peer_taps)."""
    z: torch.Tensor
    scale: torch.Tensor
    shift: torch.Tensor
    coef: torch.Tensor  # device float[3]

    def c(self) -> PeerTapsC:
        t = PeerTapsC()
        t.z_peer = self.z.data_ptr()
        t.ld_peer = _cl(self.z, "peer z")[2]
        t.scale_peer = self.scale.data_ptr()
        t.shift_peer = self.shift.data_ptr()
        t.coef = self.coef.data_ptr()
        return t


def _peer(t: Optional[PeerTaps]):
    return None if t is None else C.byref(t.c())


ACC_WORDS = 4
ACC_REPLICAS = 8


def conv_acc_supported(g: ConvGeom, has_prologue) -> bool:
    gc = g.c()
    return bool(lib().mpgan_conv_acc_supported(C.byref(gc), int(has_prologue)))


def conv_fold_supported(g: ConvGeom) -> bool:
    gc = g.c()
    return bool(lib().mpgan_conv_fold_supported(C.byref(gc)))


def make_fold(acc, replicas, cstride, count, norm_mod, scale, shift, mean, invstd) -> NormFoldC:
    """mpgan_norm_fold for a torch BatchNorm module's parameters / buffers and a plan's output vectors."""
    f = NormFoldC()
    f.acc = acc.data_ptr()
    f.replicas, f.cstride, f.count = int(replicas), int(cstride), int(count)
    f.gamma, f.beta = _ptr(norm_mod.weight), _ptr(norm_mod.bias)
    f.eps, f.momentum = float(norm_mod.eps), float(norm_mod.momentum if norm_mod.momentum is not None else 0.1)
    f.running_mean = _ptr(getattr(norm_mod, "running_mean", None))
    f.running_var = _ptr(getattr(norm_mod, "running_var", None))
    f.num_batches_tracked = _ptr(getattr(norm_mod, "num_batches_tracked", None))
    f.scale, f.shift, f.mean, f.invstd = scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr()
    return f


def _check_in_out(g: ConvGeom, x: torch.Tensor, y: torch.Tensor, what: str):
    if tuple(x.shape) != (g.n, *g.in_dhw, g.cin):
        raise ValueError(f"{what}: input shape {tuple(x.shape)} != {(g.n, *g.in_dhw, g.cin)}")
    if tuple(y.shape) != (g.n, *g.out_dhw, g.cout):
        raise ValueError(f"{what}: output shape {tuple(y.shape)} != {(g.n, *g.out_dhw, g.cout)}")


def conv_stats_rows(g: ConvGeom, has_prologue) -> int:
    """Partial rows the conv's fused statistics leave (0: none).  has_prologue: False/True, or the C
    code 0 none / 1 per-channel / 2 per-(sample, channel) / 3 per-channel fast LeakyReLU."""
    gc = g.c()
    return int(lib().mpgan_conv_stats_rows(C.byref(gc), int(has_prologue)))


def conv_variant(g: ConvGeom, backward_data: bool, has_prologue) -> int:
    gc = g.c()
    return int(lib().mpgan_conv_variant(C.byref(gc), int(backward_data), int(has_prologue)))


def conv_forward(g: ConvGeom, x, w_packed, bias, y, *, pro: Optional[Prologue] = None, resid=None,
                 tanh_out: bool = False, stats_partials=None):
    _check_in_out(g, x, y, "conv_forward")
    _, _, ldx = _cl(x, "conv_forward x")
    _, _, ldy = _cl(y, "conv_forward y")
    ldr = 0
    if resid is not None:
        if resid.shape != y.shape:
            raise ValueError("conv_forward: resid shape mismatch")
        _, _, ldr = _cl(resid, "conv_forward resid")
    if w_packed.numel() < g.cout * g.cin * g.taps:
        raise ValueError("conv_forward: packed weight too small")
    gc = g.c()
    if stats_partials is not None and stats_partials.numel() < conv_stats_rows(
            g, 0 if pro is None else (2 if pro.n_stride else 1)) * 2 * g.cout:
        raise ValueError("conv_forward: stats_partials too small")
    check(lib().mpgan_conv_forward(C.byref(gc), x.data_ptr(), ldx, w_packed.data_ptr(), _ptr(bias), _pro(pro),
                                   _ptr(resid), ldr, int(tanh_out), _ptr(stats_partials), y.data_ptr(), ldy,
                                   _stream()), "conv_forward")
    return y


def conv_forward_act(g: ConvGeom, x, w_packed, scale, shift, slope, y, *, resid=None, tanh_out: bool = False):
    """Eval-mode inference conv (code/GAN/inferrence.py:97-110): y = prelu(conv(x) * scale + shift, slope) (+ resid) (tanh),
    everything behind the matrix product applied in the conv's epilogue (include/mpgan_hip.h: mpgan_conv_forward_act)."""
    _check_in_out(g, x, y, "conv_forward_act")
    _, _, ldx = _cl(x, "conv_forward_act x")
    _, _, ldy = _cl(y, "conv_forward_act y")
    ldr = 0
    if resid is not None:
        if resid.shape != y.shape:
            raise ValueError("conv_forward_act: resid shape mismatch")
        _, _, ldr = _cl(resid, "conv_forward_act resid")
    if w_packed.numel() < g.cout * g.cin * g.taps:
        raise ValueError("conv_forward_act: packed weight too small")
    for v in (scale, shift, slope):
        if v.numel() < g.cout or v.dtype != torch.float32 or not v.is_contiguous():
            raise ValueError("conv_forward_act: scale / shift / slope must be contiguous fp32 vectors of cout elements")
    gc = g.c()
    check(lib().mpgan_conv_forward_act(C.byref(gc), x.data_ptr(), ldx, w_packed.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                       slope.data_ptr(), _ptr(resid), ldr, int(tanh_out), y.data_ptr(), ldy, _stream()),
          "conv_forward_act")
    return y


def conv_backward_data(g: ConvGeom, dy, w_packed_bwd, dx, *, resid=None):
    _check_in_out(g, dx, dy, "conv_backward_data")
    _, _, lddy = _cl(dy, "conv_backward_data dy")
    _, _, lddx = _cl(dx, "conv_backward_data dx")
    ldr = 0
    if resid is not None:
        if resid.shape != dx.shape:
            raise ValueError("conv_backward_data: resid shape mismatch")
        _, _, ldr = _cl(resid, "conv_backward_data resid")
    gc = g.c()
    check(lib().mpgan_conv_backward_data(C.byref(gc), dy.data_ptr(), lddy, w_packed_bwd.data_ptr(), _ptr(resid), ldr,
                                         dx.data_ptr(), lddx, _stream()), "conv_backward_data")
    return dx


def conv_bwd_stats_rows(g: ConvGeom) -> int:
    """Partial rows conv_backward_data_stats leaves for this geometry (0: no fused sums for it)."""
    gc = g.c()
    return int(lib().mpgan_conv_bwd_stats_rows(C.byref(gc)))


def conv_backward_data_stats(g: ConvGeom, dy, w_packed_bwd, dx, z, scale, shift, mean, invstd, act: int, slope: float,
                             partials):
    """dx = conv_backward_data(dy) and, from the same launch, the norm-backward partial sums of dx against z
    (partials[rows][3][Cin]; see include/mpgan_hip.h)."""
    _check_in_out(g, dx, dy, "conv_backward_data_stats")
    _, _, lddy = _cl(dy, "conv_backward_data_stats dy")
    _, _, lddx = _cl(dx, "conv_backward_data_stats dx")
    _, _, ldz = _cl(z, "conv_backward_data_stats z")
    rows = conv_bwd_stats_rows(g)
    if z.shape != dx.shape or partials.numel() < rows * 3 * g.cin:
        raise ValueError("conv_backward_data_stats: z must match dx and partials hold rows*3*Cin floats")
    gc = g.c()
    check(lib().mpgan_conv_backward_data_stats(C.byref(gc), dy.data_ptr(), lddy, w_packed_bwd.data_ptr(), dx.data_ptr(),
                                               lddx, z.data_ptr(), ldz, scale.data_ptr(), shift.data_ptr(),
                                               mean.data_ptr(), invstd.data_ptr(), int(act), float(slope),
                                               partials.data_ptr(), _stream()), "conv_backward_data_stats")
    return rows


def conv_wgrad_workspace(g: ConvGeom) -> int:
    gc = g.c()
    return int(lib().mpgan_conv_wgrad_workspace(C.byref(gc)))


def conv_backward_weight(g: ConvGeom, x, dy, dw, workspace, *, pro: Optional[Prologue] = None, beta: float = 0.0,
                         dbias=None):
    _check_in_out(g, x, dy, "conv_backward_weight")
    _, _, ldx = _cl(x, "conv_backward_weight x")
    _, _, lddy = _cl(dy, "conv_backward_weight dy")
    if dw.numel() != g.cout * g.cin * g.taps or not dw.is_contiguous():
        raise ValueError("conv_backward_weight: dw must be the contiguous torch-layout weight gradient")
    gc = g.c()
    check(lib().mpgan_conv_backward_weight(C.byref(gc), x.data_ptr(), ldx, _pro(pro), dy.data_ptr(), lddy,
                                           dw.data_ptr(), _ptr(dbias), float(beta), workspace.data_ptr(),
                                           workspace.numel() * workspace.element_size(), _stream()),
          "conv_backward_weight")
    return dw


def pack_weights(flat_params, packed, table, n_entries: int, max_elems: int):
    check(lib().mpgan_pack_weights(flat_params.data_ptr(), packed.data_ptr(), table.data_ptr(), n_entries, max_elems,
                                   _stream()), "pack_weights")


def stats_chunks(pixels_per_sample: int, c: int) -> int:
    return int(lib().mpgan_stats_chunks(pixels_per_sample, c))


def channel_stats(z, partials):
    n, P, ld = _cl(z, "channel_stats")
    c = z.shape[-1]
    if partials.numel() < n * stats_chunks(P, c) * 2 * c:
        raise ValueError("channel_stats: partials too small")
    check(lib().mpgan_channel_stats(z.data_ptr(), ld, n, P, c, partials.data_ptr(), _stream()), "channel_stats")


def norm_finalize(partials, n, chunks, c, P, instance, gamma, beta, eps, momentum, running_mean, running_var, nbt,
                  scale, shift, mean, invstd):
    check(lib().mpgan_norm_finalize(partials.data_ptr(), n, chunks, c, P, int(instance), _ptr(gamma), _ptr(beta),
                                    float(eps), float(momentum), _ptr(running_mean), _ptr(running_var), _ptr(nbt),
                                    scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                    _stream()), "norm_finalize")


def norm_act_add(z, pz: Optional[Prologue], r, pr: Optional[Prologue], out, *, tanh_out=False):
    n, P, ldz = _cl(z, "norm_act_add z")
    _, _, ldo = _cl(out, "norm_act_add out")
    ldr = 0
    if r is not None:
        _, _, ldr = _cl(r, "norm_act_add r")
        if r.shape != z.shape:
            raise ValueError("norm_act_add: r shape mismatch")
    if out.shape != z.shape:
        raise ValueError("norm_act_add: out shape mismatch")
    check(lib().mpgan_norm_act_add(z.data_ptr(), ldz, _pro(pz), _ptr(r), ldr, _pro(pr), n, P, z.shape[-1],
                                   int(tanh_out), out.data_ptr(), ldo, _stream()), "norm_act_add")
    return out


def norm_bwd_reduce(g, z, p: Prologue, mean, invstd, partials, peer: Optional[PeerTaps] = None):
    n, P, ldz = _cl(z, "norm_bwd_reduce z")
    _, _, ldg = _cl(g, "norm_bwd_reduce g")
    if g.shape != z.shape:
        raise ValueError("norm_bwd_reduce: shape mismatch")
    c = z.shape[-1]
    if partials.numel() < n * stats_chunks(P, c) * (3 * c + 1):
        raise ValueError("norm_bwd_reduce: partials too small (need n*chunks*(3*c + 1) floats)")
    check(lib().mpgan_norm_bwd_reduce(g.data_ptr(), ldg, z.data_ptr(), ldz, _pro(p), mean.data_ptr(),
                                      invstd.data_ptr(), _peer(peer), n, P, c, partials.data_ptr(), _stream()),
          "norm_bwd_reduce")


def norm_bwd_finalize(partials, n, chunks, c, P, instance, dgamma, dbeta, dslope, c1, c2):
    check(lib().mpgan_norm_bwd_finalize(partials.data_ptr(), n, chunks, c, P, int(instance), _ptr(dgamma), _ptr(dbeta),
                                        _ptr(dslope), c1.data_ptr(), c2.data_ptr(), _stream()), "norm_bwd_finalize")


def norm_bwd_apply(g, z, p: Prologue, mean, invstd, c1, c2, dz, peer: Optional[PeerTaps] = None):
    n, P, ldz = _cl(z, "norm_bwd_apply z")
    _, _, ldg = _cl(g, "norm_bwd_apply g")
    _, _, lddz = _cl(dz, "norm_bwd_apply dz")
    if g.shape != z.shape or dz.shape != z.shape:
        raise ValueError("norm_bwd_apply: shape mismatch")
    check(lib().mpgan_norm_bwd_apply(g.data_ptr(), ldg, z.data_ptr(), ldz, _pro(p), mean.data_ptr(),
                                     invstd.data_ptr(), c1.data_ptr(), c2.data_ptr(), _peer(peer), n, P, z.shape[-1],
                                     dz.data_ptr(), lddz, _stream()), "norm_bwd_apply")
    return dz


def tap_l1_partials() -> int:
    return int(lib().mpgan_tap_l1_partials())


def tap_l1(za, pa: Prologue, zb, pb: Prologue, partials, out3):
    """out3 = mean |z_a-z_b|, |y_a-y_b|, |a_a-a_b| of one conv+BN+act layer of two passes."""
    n, P, lda = _cl(za, "tap_l1 a")
    _, _, ldb = _cl(zb, "tap_l1 b")
    if za.shape != zb.shape:
        raise ValueError("tap_l1: shape mismatch")
    check(lib().mpgan_tap_l1(za.data_ptr(), lda, _pro(pa), zb.data_ptr(), ldb, _pro(pb), n * P, za.shape[-1],
                             partials.data_ptr(), out3.data_ptr(), _stream()), "tap_l1")
    return out3


def conv_splitk_workspace(g: ConvGeom) -> int:
    gc = g.c()
    return int(lib().mpgan_conv_splitk_workspace(C.byref(gc)))


def conv_forward_splitk(g: ConvGeom, x, w_packed, bias, y, workspace, *, pro: Optional[Prologue] = None):
    _check_in_out(g, x, y, "conv_forward_splitk")
    _, _, ldx = _cl(x, "conv_forward_splitk x")
    _, _, ldy = _cl(y, "conv_forward_splitk y")
    gc = g.c()
    check(lib().mpgan_conv_forward_splitk(C.byref(gc), x.data_ptr(), ldx, w_packed.data_ptr(), _ptr(bias), _pro(pro),
                                          workspace.data_ptr(), workspace.numel() * 4, y.data_ptr(), ldy, _stream()),
          "conv_forward_splitk")
    return y


def sigmoid_forward(logit, prob):
    check(lib().mpgan_sigmoid_forward(logit.data_ptr(), logit.numel(), prob.data_ptr(), _stream()), "sigmoid_forward")
    return prob


def reduce_partials(partials, rows, row_stride, c, out, beta=0.0):
    check(lib().mpgan_reduce_partials(partials.data_ptr(), rows, row_stride, c, out.data_ptr(), float(beta),
                                      _stream()), "reduce_partials")


def add_tanh(a, b, y, apply_tanh: bool):
    check(lib().mpgan_add_tanh(a.data_ptr(), _ptr(b), a.numel(), int(apply_tanh), y.data_ptr(), _stream()),
          "add_tanh")
    return y


def tanh_backward(g, y, dx):
    check(lib().mpgan_tanh_backward(g.data_ptr(), y.data_ptr(), y.numel(), dx.data_ptr(), _stream()),
          "tanh_backward")
    return dx


def axpby(a, alpha, b, beta, y):
    check(lib().mpgan_axpby(a.data_ptr(), float(alpha), _ptr(b), float(beta), a.numel(), y.data_ptr(), _stream()),
          "axpby")
    return y


def copy_slice(src, dst, accumulate=False):
    n, P, lds = _cl(src, "copy_slice src")
    _, _, ldd = _cl(dst, "copy_slice dst")
    if src.shape != dst.shape:
        raise ValueError("copy_slice: shape mismatch")
    check(lib().mpgan_copy_slice(src.data_ptr(), lds, dst.data_ptr(), ldd, n * P, src.shape[-1], int(accumulate),
                                 _stream()), "copy_slice")
    return dst


def linear1_partials(n: int) -> int:
    return int(lib().mpgan_linear1_partials(n))


def linear1_forward(z, p: Optional[Prologue], w_perm, bias, partials, logit, prob=None):
    n, P, ld = _cl(z, "linear1_forward z")
    c = z.shape[-1]
    if ld != c:
        raise ValueError("linear1_forward: z must be dense")
    check(lib().mpgan_linear1_forward(z.data_ptr(), _pro(p), n, P, c, w_perm.data_ptr(), _ptr(bias),
                                      partials.data_ptr(), logit.data_ptr(), _ptr(prob), _stream()),
          "linear1_forward")
    return logit


def linear1_backward(z, p: Optional[Prologue], w_perm, dlogit, g_a, dw, dbias, beta=0.0):
    n, P, ld = _cl(z, "linear1_backward z")
    c = z.shape[-1]
    if ld != c or (g_a is not None and (g_a.shape != z.shape or not g_a.is_contiguous())):
        raise ValueError("linear1_backward: z / g_a must be dense and same shape")
    check(lib().mpgan_linear1_backward(z.data_ptr(), _pro(p), n, P, c, w_perm.data_ptr(), dlogit.data_ptr(),
                                       _ptr(g_a), _ptr(dw), _ptr(dbias), float(beta), _stream()), "linear1_backward")


def sigmoid_bce(logit, target: float, loss_scale: float, prob, loss, dlogit):
    check(lib().mpgan_sigmoid_bce(logit.data_ptr(), logit.numel(), float(target), float(loss_scale), _ptr(prob),
                                  _ptr(loss), _ptr(dlogit), _stream()), "sigmoid_bce")


def bce_forward(prob, target, loss):
    check(lib().mpgan_bce_forward(prob.data_ptr(), target.data_ptr(), prob.numel(), loss.data_ptr(), _stream()),
          "bce_forward")


def bce_backward(prob, target, gout, dprob):
    check(lib().mpgan_bce_backward(prob.data_ptr(), target.data_ptr(), prob.numel(), gout.data_ptr(),
                                   dprob.data_ptr(), _stream()), "bce_backward")


def sigmoid_backward(dprob, prob, dlogit):
    check(lib().mpgan_sigmoid_backward(dprob.data_ptr(), prob.data_ptr(), prob.numel(), dlogit.data_ptr(), _stream()),
          "sigmoid_backward")


def scale_by_device_scalar(x, scalar, y):
    check(lib().mpgan_scale_by_device_scalar(x.data_ptr(), scalar.data_ptr(), x.numel(), y.data_ptr(), _stream()),
          "scale_by_device_scalar")
    return y


def weighted_sum(v, w, out):
    """out[0] = sum_i v[i] * w[i] (device vectors of equal length, index order)."""
    if v.numel() != w.numel() or not v.is_contiguous() or not w.is_contiguous():
        raise ValueError("weighted_sum: v and w must be contiguous and of equal length")
    check(lib().mpgan_weighted_sum(v.data_ptr(), w.data_ptr(), v.numel(), out.data_ptr(), _stream()), "weighted_sum")
    return out


def l1_partials() -> int:
    return int(lib().mpgan_l1_partials())


def l1_loss(a, b, partials, loss, grad_a=None, grad_scale: float = 1.0):
    if a.shape != b.shape or not a.is_contiguous() or not b.is_contiguous():
        raise ValueError("l1_loss: a and b must be contiguous and same shape")
    check(lib().mpgan_l1_loss(a.data_ptr(), b.data_ptr(), a.numel(), float(grad_scale), partials.data_ptr(),
                              loss.data_ptr(), _ptr(grad_a), _stream()), "l1_loss")


def adam_step(p, g, m, v, lr, b1, b2, eps, step: int, grad_scale: float = 1.0):
    if not (p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()):
        raise ValueError("adam_step: flat buffers must be contiguous")
    if not (p.numel() == g.numel() == m.numel() == v.numel()):
        raise ValueError("adam_step: size mismatch")
    check(lib().mpgan_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr),
                                float(b1), float(b2), float(eps), int(step), float(grad_scale), _stream()),
          "adam_step")


def _i3(v: Sequence[int]):
    return (C.c_int32 * 3)(*[int(x) for x in v])


def patch_gather(vol, corners, samples: int, roi, patches):
    """vol (B,1,D,H,W) or (B,D,H,W,1); corners int32 (B*S,3)."""
    b = vol.shape[0]
    dhw = vol.shape[2:] if vol.shape[1] == 1 and vol.dim() == 5 and vol.shape[-1] != 1 else vol.shape[1:4]
    if corners.dtype != torch.int32 or corners.numel() != b * samples * 3:
        raise ValueError("patch_gather: corners must be int32 (B*S,3)")
    check(lib().mpgan_patch_gather(vol.data_ptr(), b, _i3(dhw), corners.data_ptr(), samples, _i3(roi),
                                   patches.data_ptr(), _stream()), "patch_gather")
    return patches


def patch_scatter_add(dpatches, corners, samples: int, roi, dvol):
    b = dvol.shape[0]
    dhw = dvol.shape[2:] if dvol.shape[1] == 1 and dvol.dim() == 5 and dvol.shape[-1] != 1 else dvol.shape[1:4]
    check(lib().mpgan_patch_scatter_add(dpatches.data_ptr(), b, _i3(dhw), corners.data_ptr(), samples, _i3(roi),
                                        dvol.data_ptr(), _stream()), "patch_scatter_add")
    return dvol


def pack_weight(w: torch.Tensor, *, transposed: bool = False, for_dgrad: bool = False) -> torch.Tensor:
    """Pack ONE torch-layout conv / convT / linear weight (convenience for tests
    and small callers; networks pack all their weights with one table launch)."""
    w = w.contiguous()
    if transposed:
        cin, cout = w.shape[0], w.shape[1]
    else:
        cout, cin = w.shape[0], w.shape[1]
    taps = w.numel() // (cin * cout)
    table = torch.tensor([[0, 0, cout, cin, taps, int(transposed), int(for_dgrad), 0]], dtype=torch.int64,
                         device=w.device)
    packed = torch.empty(w.numel(), dtype=torch.float32, device=w.device)
    pack_weights(w, packed, table, 1, w.numel())
    return packed


# --------------------------------------------------------------------------
# bf16 storage path (BASELINE config C5): activations / packed weights bf16, fp32 accumulate + statistics
# --------------------------------------------------------------------------
def _clx(t: torch.Tensor, dtype, what: str) -> Tuple[int, int, int]:
    """_cl for a channels-last tensor of the given dtype."""
    if t.dim() != 5 or t.dtype != dtype or not t.is_cuda:
        raise ValueError(f"{what}: need a CUDA {dtype} (N,D,H,W,C) tensor, got {tuple(t.shape)} {t.dtype} {t.device}")
    n, d, h, w, c = t.shape
    st = t.stride()
    ld = st[3] if w > 1 else (st[2] if h > 1 else (st[1] if d > 1 else (st[0] if n > 1 else c)))
    want = (d * h * w * ld, h * w * ld, w * ld, ld, 1)
    if not (ld >= c and all(sz == 1 or a == b for sz, a, b in zip(t.shape, st, want))):
        raise ValueError(f"{what}: not a pixel-contiguous channels-last view: shape {tuple(t.shape)} strides {st}")
    return n, d * h * w, ld


BF16 = torch.bfloat16


def _check_shapes(g: ConvGeom, x, y, what):
    if tuple(x.shape) != (g.n, *g.in_dhw, g.cin) or tuple(y.shape) != (g.n, *g.out_dhw, g.cout):
        raise ValueError(f"{what}: shapes {tuple(x.shape)} / {tuple(y.shape)} do not match the geometry")


def conv_stats_rows_bf16(g: ConvGeom) -> int:
    gc = g.c()
    return int(lib().mpgan_conv_stats_rows_bf16(C.byref(gc)))


def conv_forward_bf16(g: ConvGeom, x, w_packed, bias, y, *, stats_partials=None):
    _check_shapes(g, x, y, "conv_forward_bf16")
    _, _, ldx = _clx(x, BF16, "conv_forward_bf16 x")
    _, _, ldy = _clx(y, BF16, "conv_forward_bf16 y")
    if w_packed.dtype != BF16 or w_packed.numel() < g.cout * g.cin * g.taps:
        raise ValueError("conv_forward_bf16: packed weight must be bf16 and complete")
    if stats_partials is not None and stats_partials.numel() < conv_stats_rows_bf16(g) * 2 * g.cout:
        raise ValueError("conv_forward_bf16: stats_partials too small")
    gc = g.c()
    check(lib().mpgan_conv_forward_bf16(C.byref(gc), x.data_ptr(), ldx, w_packed.data_ptr(), _ptr(bias),
                                        _ptr(stats_partials), y.data_ptr(), ldy, _stream()), "conv_forward_bf16")
    return y


def conv_backward_data_bf16(g: ConvGeom, dy, w_packed_bwd, dx):
    _check_shapes(g, dx, dy, "conv_backward_data_bf16")
    _, _, lddy = _clx(dy, BF16, "conv_backward_data_bf16 dy")
    _, _, lddx = _clx(dx, BF16, "conv_backward_data_bf16 dx")
    if w_packed_bwd.dtype != BF16:
        raise ValueError("conv_backward_data_bf16: packed weight must be bf16")
    gc = g.c()
    check(lib().mpgan_conv_backward_data_bf16(C.byref(gc), dy.data_ptr(), lddy, w_packed_bwd.data_ptr(), dx.data_ptr(),
                                              lddx, _stream()), "conv_backward_data_bf16")
    return dx


def conv_wgrad_workspace_bf16(g: ConvGeom) -> int:
    gc = g.c()
    return int(lib().mpgan_conv_wgrad_workspace_bf16(C.byref(gc)))


def conv_backward_weight_bf16(g: ConvGeom, x, dy, dw, workspace, *, beta: float = 0.0):
    _check_shapes(g, x, dy, "conv_backward_weight_bf16")
    _, _, ldx = _clx(x, BF16, "conv_backward_weight_bf16 x")
    _, _, lddy = _clx(dy, BF16, "conv_backward_weight_bf16 dy")
    if dw.dtype != torch.float32 or dw.numel() != g.cout * g.cin * g.taps or not dw.is_contiguous():
        raise ValueError("conv_backward_weight_bf16: dw must be the contiguous fp32 torch-layout weight gradient")
    gc = g.c()
    check(lib().mpgan_conv_backward_weight_bf16(C.byref(gc), x.data_ptr(), ldx, dy.data_ptr(), lddy, dw.data_ptr(),
                                                float(beta), workspace.data_ptr(),
                                                workspace.numel() * workspace.element_size(), _stream()),
          "conv_backward_weight_bf16")
    return dw


def conv_forward_f32_to_bf16(g: ConvGeom, x, w_packed, bias, y, *, stats_partials=None):
    _check_shapes(g, x, y, "conv_forward_f32_to_bf16")
    _, _, ldx = _cl(x, "conv_forward_f32_to_bf16 x")
    _, _, ldy = _clx(y, BF16, "conv_forward_f32_to_bf16 y")
    gc = g.c()
    check(lib().mpgan_conv_forward_f32_to_bf16(C.byref(gc), x.data_ptr(), ldx, w_packed.data_ptr(), _ptr(bias),
                                               _ptr(stats_partials), y.data_ptr(), ldy, _stream()),
          "conv_forward_f32_to_bf16")
    return y


def conv_backward_data_bf16_to_f32(g: ConvGeom, dy, w_packed_bwd, dx):
    _check_shapes(g, dx, dy, "conv_backward_data_bf16_to_f32")
    _, _, lddy = _clx(dy, BF16, "conv_backward_data_bf16_to_f32 dy")
    _, _, lddx = _cl(dx, "conv_backward_data_bf16_to_f32 dx")
    gc = g.c()
    check(lib().mpgan_conv_backward_data_bf16_to_f32(C.byref(gc), dy.data_ptr(), lddy, w_packed_bwd.data_ptr(),
                                                     dx.data_ptr(), lddx, _stream()), "conv_backward_data_bf16_to_f32")
    return dx


def conv_wgrad_workspace_bf16dy(g: ConvGeom) -> int:
    gc = g.c()
    return int(lib().mpgan_conv_wgrad_workspace_bf16dy(C.byref(gc)))


def conv_backward_weight_bf16dy(g: ConvGeom, x, dy, dw, workspace, *, beta: float = 0.0, dbias=None):
    _check_shapes(g, x, dy, "conv_backward_weight_bf16dy")
    _, _, ldx = _cl(x, "conv_backward_weight_bf16dy x")
    _, _, lddy = _clx(dy, BF16, "conv_backward_weight_bf16dy dy")
    gc = g.c()
    check(lib().mpgan_conv_backward_weight_bf16dy(C.byref(gc), x.data_ptr(), ldx, dy.data_ptr(), lddy, dw.data_ptr(),
                                                  _ptr(dbias), float(beta), workspace.data_ptr(),
                                                  workspace.numel() * workspace.element_size(), _stream()),
          "conv_backward_weight_bf16dy")
    return dw


def pack_weight_bf16(w: torch.Tensor, *, for_dgrad: bool = False) -> torch.Tensor:
    """One fp32 ConvNd weight -> bf16 [Cout][tap][Cin] (or [Cin][tap][Cout] for backward-data)."""
    w = w.contiguous()
    cout, cin = w.shape[0], w.shape[1]
    taps = w.numel() // (cin * cout)
    table = torch.tensor([[0, 0, cout, cin, taps, 0, int(for_dgrad), 0]], dtype=torch.int64, device=w.device)
    packed = torch.empty(w.numel(), dtype=BF16, device=w.device)
    check(lib().mpgan_pack_weights_bf16(w.data_ptr(), packed.data_ptr(), table.data_ptr(), 1, w.numel(), _stream()),
          "pack_weights_bf16")
    return packed


def norm_act_bf16(z, scale, shift, slope: float, out):
    n, P, ldz = _clx(z, BF16, "norm_act_bf16 z")
    if out.shape != z.shape or out.dtype not in (BF16, torch.float32):
        raise ValueError("norm_act_bf16: out must match z and be bf16 or fp32")
    _, _, ldo = _clx(out, out.dtype, "norm_act_bf16 out")
    check(lib().mpgan_norm_act_bf16(z.data_ptr(), ldz, scale.data_ptr(), shift.data_ptr(), float(slope), n * P,
                                    z.shape[-1], out.data_ptr(), ldo, int(out.dtype == torch.float32), _stream()),
          "norm_act_bf16")
    return out


def norm_bwd_rows_bf16(rows: int, c: int) -> int:
    return int(lib().mpgan_norm_bwd_rows_bf16(rows, c))


def norm_bwd_reduce_bf16(g, z, scale, shift, mean, invstd, slope: float, partials):
    n, P, ldz = _clx(z, BF16, "norm_bwd_reduce_bf16 z")
    _, _, ldg = _clx(g, g.dtype, "norm_bwd_reduce_bf16 g")
    c = z.shape[-1]
    if partials.numel() < norm_bwd_rows_bf16(n * P, c) * 3 * c + c:
        raise ValueError("norm_bwd_reduce_bf16: partials too small")
    check(lib().mpgan_norm_bwd_reduce_bf16(g.data_ptr(), int(g.dtype == torch.float32), ldg, z.data_ptr(), ldz,
                                           scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                           float(slope), n * P, c, partials.data_ptr(), _stream()),
          "norm_bwd_reduce_bf16")


def norm_bwd_apply_bf16(g, z, scale, shift, mean, invstd, c1, c2, slope: float, dz, bias_partials=None):
    n, P, ldz = _clx(z, BF16, "norm_bwd_apply_bf16 z")
    _, _, ldg = _clx(g, g.dtype, "norm_bwd_apply_bf16 g")
    _, _, lddz = _clx(dz, BF16, "norm_bwd_apply_bf16 dz")
    c = z.shape[-1]
    check(lib().mpgan_norm_bwd_apply_bf16(g.data_ptr(), int(g.dtype == torch.float32), ldg, z.data_ptr(), ldz,
                                          scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                          c1.data_ptr(), c2.data_ptr(), float(slope), n * P, c, dz.data_ptr(), lddz,
                                          _ptr(bias_partials), _stream()), "norm_bwd_apply_bf16")
    return dz


# --------------------------------------------------------------------------
# BatchNorm statistics through accumulators + fold-on-load (csrc/norm_fold.h)
# --------------------------------------------------------------------------
def acc_buffer(c: int, device) -> torch.Tensor:
    """Zeroed int64 accumulators [ACC_REPLICAS][ACC_WORDS][c] for one norm layer fed by a c-channel conv."""
    return torch.zeros(ACC_REPLICAS * ACC_WORDS * c, dtype=torch.int64, device=device)


def conv_forward_fold(g: ConvGeom, x, w_packed, bias, y, *, pro: Optional[Prologue] = None, fold: Optional[NormFoldC] = None,
                      resid=None, tanh_out: bool = False, stats_acc=None):
    _check_in_out(g, x, y, "conv_forward_fold")
    _, _, ldx = _cl(x, "conv_forward_fold x")
    _, _, ldy = _cl(y, "conv_forward_fold y")
    ldr = _cl(resid, "conv_forward_fold resid")[2] if resid is not None else 0
    gc = g.c()
    if stats_acc is not None and (stats_acc.dtype != torch.int64 or stats_acc.numel() < ACC_REPLICAS * ACC_WORDS * g.cout):
        raise ValueError("conv_forward_fold: stats_acc must be int64 [ACC_REPLICAS][4][Cout]")
    check(lib().mpgan_conv_forward_fold(C.byref(gc), x.data_ptr(), ldx, w_packed.data_ptr(), _ptr(bias), _pro(pro),
                                        C.byref(fold) if fold is not None else None, _ptr(resid), ldr, int(tanh_out),
                                        None, _ptr(stats_acc), ACC_REPLICAS if stats_acc is not None else 0,
                                        y.data_ptr(), ldy, _stream()), "conv_forward_fold")
    return y


def norm_act_add_fold(z, pz: Prologue, fold: NormFoldC, r, pr: Optional[Prologue], out, *, tanh_out=False):
    n, P, ldz = _cl(z, "norm_act_add_fold z")
    _, _, ldo = _cl(out, "norm_act_add_fold out")
    ldr = _cl(r, "norm_act_add_fold r")[2] if r is not None else 0
    check(lib().mpgan_norm_act_add_fold(z.data_ptr(), ldz, _pro(pz), C.byref(fold), _ptr(r), ldr, _pro(pr), n, P,
                                        z.shape[-1], int(tanh_out), out.data_ptr(), ldo, _stream()), "norm_act_add_fold")
    return out
