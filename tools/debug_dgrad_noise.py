#!/usr/bin/env python3
"""Is a dL/dx mismatch of the discriminator rounding noise or a bug?  Compares the HIP path and the
fp32 CPU oracle against the same graph evaluated in float64 (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import copy
import torch
from oracle import refmodel as R
from mpgan_amd.gan import adversarial_loss
from mpgan_amd.networks import Discriminator

spatial, n = (128, 128), 1
ref = R.Discriminator((1, *spatial), dimensions=2)
R.closed_form_fill_(ref)
ref.train()
ref64 = copy.deepcopy(ref).double()
ours = Discriminator((1, *spatial), dimensions=2)
ours.load_state_dict(ref.state_dict())
ours.cuda().train()
gen = torch.Generator().manual_seed(17)
x = (torch.rand(n, 1, *spatial, generator=gen) * 2 - 1)


def run(net, xin, lossf):
    xin = xin.clone().requires_grad_(True)
    p = net(xin)
    loss = lossf(p, torch.full_like(p, 0.9))
    loss.backward()
    return xin.grad.detach().double().cpu()


g64 = run(ref64, x.double(), R.adversarial_loss)
g32 = run(ref, x, R.adversarial_loss)
gh = run(ours, x.cuda(), adversarial_loss)
sc = g64.abs().max().item()
for name, g in (("oracle fp32", g32), ("hip", gh)):
    e = (g - g64).abs()
    print(f"{name:12s} max err {e.max().item() / sc:.3e} of max|g|, mean {e.mean().item() / sc:.3e}, "
          f">5e-3: {(e > 5e-3 * sc).sum().item()} of {e.numel()}")
e = (gh - g32).abs()
print(f"hip vs oracle fp32: max {e.max().item() / sc:.3e}, >5e-3: {(e > 5e-3 * sc).sum().item()}")
idx = (e > 5e-3 * sc).nonzero()
print("rows of offenders:", sorted(set(idx[:, 2].tolist()))[:40])

# ---- per-layer: raw conv outputs and their gradients against the fp64 graph ----
feats = {}
convs64 = [ref64.model_conv[i] for i in (0, 3, 6, 9)]
hooks = []
for i, cv in enumerate(convs64):
    def hook(mod, inp, out, i=i):
        out.retain_grad()
        feats[i] = out
    hooks.append(cv.register_forward_hook(hook))
x64 = x.double().clone().requires_grad_(True)
p64 = ref64(x64)
R.adversarial_loss(p64, torch.full_like(p64, 0.9)).backward()
plan = [pl for pool in ours._plans.values() for pl in pool][0]
for i in range(4):
    z = plan.zs[i].double().cpu().squeeze(1).permute(0, 3, 1, 2)     # (N,1,H,W,C) -> (N,C,H,W)
    zr = feats[i].detach()
    ez = (z - zr).abs()
    g = plan.gas[i].double().cpu().squeeze(1).permute(0, 3, 1, 2)
    gr = feats[i].grad
    eg = (g - gr).abs()
    bad = (eg > 1e-3 * gr.abs().max()).nonzero()
    print(f"conv{i + 1}: z max err {ez.max().item() / zr.abs().max().item():.2e}   dz max err "
          f"{eg.max().item() / gr.abs().max().item():.2e}  bad rows {sorted(set(bad[:, 2].tolist()))[:30]} "
          f"bad ch {len(set(bad[:, 1].tolist()))}")
