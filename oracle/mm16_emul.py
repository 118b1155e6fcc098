"""TEST INFRASTRUCTURE (imported by tests/ only, never by the product path).

CPU restatement of the generator (code/GAN/GAN_final.py:92-122 over oracle.refmodel's MONAI-0.4.0 U-Net restatement)
under the "bf16 matrix operands" contract of BASELINE config C5 as this repo defines it (include/mpgan_hip.h,
MPGAN_CONV_MM_BF16): every tensor, the BatchNorm / PReLU arithmetic, bias, residual sums,
statistics and the accumulation of every convolution stay fp32; the TWO operands of each matrix product that an MFMA
kernel serves are rounded to bf16 (round-to-nearest-even) as they enter it --
  forward:          conv(rb(a), rb(W)) + b              a = the activated input the kernel's prologue forms
  backward-data:    conv^T(rb(dy), rb(W))
  backward-weight:  corr(rb(a), rb(dy));   the bias gradient sums the UNROUNDED dy (fp32 column sums)
and bf16 x bf16 products are exact in fp32, so only the summation order separates two implementations of this
contract.  The layers with one input or one output channel run on the vector ALUs in fp32 and are left alone
(`served`: cin >= 16 and cout >= 16 -- the generator's K-stepped and 3-D patch kernels).

Parity: UNPINNED like the generator itself (monai is absent; oracle/__init__.py).  The distance between this model and
oracle.refmodel is what the tests report as the precision cost of the contract."""
import torch
import torch.nn as nn
import torch.nn.functional as F

BF = torch.bfloat16


def rb(t: torch.Tensor) -> torch.Tensor:
    return t.to(BF).to(t.dtype)


class _MMConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, mod):
        ctx.mod = mod
        ctx.save_for_backward(x, w)
        return _apply(mod, rb(x), rb(w), b)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        mod = ctx.mod
        xr, wr = rb(x).requires_grad_(True), rb(w).requires_grad_(True)
        with torch.enable_grad():
            out = _apply(mod, xr, wr, None)
        gx, gw = torch.autograd.grad(out, (xr, wr), rb(gy))
        gb = gy.sum([0] + list(range(2, gy.dim()))) if mod.bias is not None else None
        return gx, gw, gb, None


def _apply(mod, x, w, b):
    if isinstance(mod, (nn.ConvTranspose2d, nn.ConvTranspose3d)):
        f = F.conv_transpose2d if isinstance(mod, nn.ConvTranspose2d) else F.conv_transpose3d
        return f(x, w, b, stride=mod.stride, padding=mod.padding, output_padding=mod.output_padding)
    f = F.conv2d if isinstance(mod, nn.Conv2d) else F.conv3d
    return f(x, w, b, stride=mod.stride, padding=mod.padding)


def served(mod) -> bool:
    """Does an MFMA kernel (and hence the bf16-operand contract) serve this conv?"""
    return mod.in_channels >= 16 and mod.out_channels >= 16


def apply_mm16(net: nn.Module) -> nn.Module:
    """Switch every served conv of `net` (an oracle.refmodel generator / U-Net, any float dtype) to the contract, in place."""
    for mod in net.modules():
        if isinstance(mod, (nn.Conv2d, nn.Conv3d, nn.ConvTranspose2d, nn.ConvTranspose3d)) and served(mod):
            mod.forward = (lambda m: (lambda x: _MMConv.apply(x, m.weight, m.bias, m)))(mod)
    return net
