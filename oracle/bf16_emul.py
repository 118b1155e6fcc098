"""TEST INFRASTRUCTURE (imported by tests/ only, never by the product path).

CPU restatement of the discriminator step of code/GAN/GAN_final.py:159-209,244-245 under the bf16-STORAGE
contract of BASELINE config C5 as this repo defines it ("bf16 storage"): fp32 arithmetic everywhere,
with a round-to-nearest-even to bf16 at exactly the places where the HIP path stores a tensor in HBM:
  * packed weights of the three dense convs (the first conv and the Linear head read fp32 weights),
  * every raw conv output z_i (its BatchNorm statistics are taken BEFORE the rounding, from the fp32 sums),
  * every materialised activation a_i = LeakyReLU(BN(z_i)) except the last one (fp32, read by the fp32 head),
  * every activation gradient written by a backward-data kernel and every dz written by the BatchNorm backward.
A wrong index, tap, swizzle or coefficient in a kernel shows up against this model at fp32-noise level, which
a comparison against the pure-fp32 oracle (whose distance to bf16 storage is 1e-2 .. 1e-1 on gradients) cannot
resolve.  The distance between this model and oracle.refmodel is reported by the tests as the precision cost."""
import torch
import torch.nn.functional as F

BF = torch.bfloat16


def rb(t: torch.Tensor) -> torch.Tensor:
    return t.to(BF).float()


def disc_step(disc, x: torch.Tensor, target: float, eps: float = 1e-5, slope: float = 0.2, acc64: bool = False):
    """disc: an oracle.refmodel.Discriminator (fp32 parameters; BatchNorm running statistics are left alone).
    Returns dict(validity, loss, grad_x, grads{name: tensor}) of loss = BCE(D(x), target) (mean).
    acc64: every convolution (forward, backward-data, backward-weight) accumulates in fp64 and is rounded to fp32 ONCE
    -- the same contract with another summation order.  Two correct fp32 implementations differ by their accumulation
    order (1e-5 .. 1e-4 of a conv output through the cancellation of a 1,700 - 6,900-term sum), which moves a few per
    cent of the stored elements across a bf16 rounding boundary; BatchNorm's backward amplifies that like any other
    bf16-sized perturbation.  The distance between the acc64 run and the plain one is the yardstick the step test
    holds the HIP path to (a one-ulp change of the INPUT is 100 x smaller than that and under-states it)."""
    dims = x.dim() - 2
    conv_ = F.conv2d if dims == 2 else F.conv3d
    if acc64:
        def conv(a, w, b, stride):
            return conv_(a.double(), w.double(), None if b is None else b.double(), stride=stride).float()
    else:
        conv = conv_
    convs = [disc.model_conv[i] for i in (0, 3, 6, 9)]
    bns = [disc.model_conv[i] for i in (1, 4, 7, 10)]
    lin = disc.model_linear[1]
    red = [0] + list(range(2, 2 + dims))
    shp = [1, -1] + [1] * dims
    a = x
    saved = []
    for i, (cv, bn) in enumerate(zip(convs, bns)):
        w = cv.weight.detach() if i == 0 else rb(cv.weight.detach())
        z32 = conv(a, w, cv.bias.detach(), stride=cv.stride)
        mean = z32.mean(red)
        var = z32.var(red, unbiased=False)
        invstd = 1.0 / torch.sqrt(var + eps)
        scale = bn.weight.detach() * invstd
        shift = bn.bias.detach() - mean * scale
        z = rb(z32)
        y = z * scale.view(shp) + shift.view(shp)
        act = torch.where(y < 0, y * slope, y)
        a_next = act if i == 3 else rb(act)
        saved.append((a, w, z, y, mean, invstd, scale, cv))
        a = a_next
    n = x.shape[0]
    logit = a.reshape(n, -1) @ lin.weight.detach().t() + lin.bias.detach()
    prob = torch.sigmoid(logit)
    t = torch.full_like(prob, target)
    loss = F.binary_cross_entropy(prob, t)
    # ---- backward (torch's BCE backward: (p - t) / max((1-p) p, 1e-12) / n, then the sigmoid) ----
    dprob = (prob - t) / torch.clamp((1 - prob) * prob, min=1e-12) / prob.numel()
    dlogit = dprob * prob * (1 - prob)
    grads = {"model_linear.1.weight": dlogit.t() @ a.reshape(n, -1), "model_linear.1.bias": dlogit.sum(0)}
    g = (dlogit @ lin.weight.detach()).reshape(a.shape)           # fp32: gradient of the fp32 last activation
    for i in range(3, -1, -1):
        a_in, w, z, y, mean, invstd, scale, cv = saved[i]
        gy = torch.where(y < 0, g * slope, g)
        zh = (z - mean.view(shp)) * invstd.view(shp)
        cnt = z.numel() / z.shape[1]
        s1, s2 = gy.sum(red), (gy * zh).sum(red)
        grads[f"model_conv.{3 * i + 1}.weight"] = s2
        grads[f"model_conv.{3 * i + 1}.bias"] = s1
        dz = rb(scale.view(shp) * (gy - (s1 / cnt).view(shp) - zh * (s2 / cnt).view(shp)))
        grads[f"model_conv.{3 * i}.bias"] = dz.sum(red)
        a_req = (a_in.double() if acc64 else a_in).clone().requires_grad_(True)
        w_req = (w.double() if acc64 else w).clone().requires_grad_(True)
        out = conv_(a_req, w_req, None, stride=cv.stride)
        ga, gw = torch.autograd.grad(out, (a_req, w_req), dz.double() if acc64 else dz)
        ga, gw = ga.float(), gw.float()
        grads[f"model_conv.{3 * i}.weight"] = gw
        g = ga if i == 0 else rb(ga)
    return {"validity": prob, "loss": loss, "grad_x": g, "grads": grads, "zs": [sv[2] for sv in saved]}
