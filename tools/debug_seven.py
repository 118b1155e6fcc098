#!/usr/bin/env python3
"""Where does the 7-level generator's gradient differ from the oracle, and is the oracle itself stable there?
Prints per-tensor relative L2: ours vs oracle(fp32), oracle(fp32) vs oracle(fp64)."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpgan_amd.networks import CasNetGenerator
from mpgan_amd.gan import reconstruction_loss
from oracle import refmodel as R

CH = tuple(int(c) for c in os.environ.get("CH", "64,128,256,512,512,512,512").split(","))
S = int(os.environ.get("S", "128"))
N = int(os.environ.get("N", "1"))
D = int(os.environ.get("DIMS", "3"))
SP = (S,) * D
rl2 = lambda a, b: ((a.double() - b.double()).norm() / (b.double().norm() + 1e-300)).item()
ref = R.CasNetGenerator((1, *SP), 1, dimensions=D, channels=CH, strides=(2,) * len(CH))
R.closed_form_fill_(ref); ref.train()
ref64 = copy.deepcopy(ref).double()
ours = CasNetGenerator((1, *SP), 1, dimensions=D, channels=CH, strides=(2,) * len(CH))
ours.load_state_dict(ref.state_dict()); ours.cuda().train()
gen = torch.Generator().manual_seed(5)
x = (torch.rand(N, 1, *SP, generator=gen) * 2 - 1).requires_grad_(True)
t = torch.rand(N, 1, *SP, generator=gen) * 2 - 1
def run(m, x, t):
    y = m(x); l = R.reconstruction_loss(y, t) + 0.1 * (y * y).mean(); l.backward(); return y.detach()
y32 = run(ref, x, t)
x64 = x.detach().double().requires_grad_(True)
y64 = run(ref64, x64, t.double())
xc = x.detach().cuda().requires_grad_(True)
y = ours(xc); l = reconstruction_loss(y, t.cuda()) + 0.1 * (y * y).mean(); l.backward()
print("y: ours-vs-32 %.2e  32-vs-64 %.2e ours-vs-64 %.2e" % (rl2(y.cpu(), y32), rl2(y32, y64), rl2(y.cpu(), y64)))
print("dx: ours-vs-32 %.2e  32-vs-64 %.2e ours-vs-64 %.2e" % (rl2(xc.grad.cpu(), x.grad), rl2(x.grad, x64.grad), rl2(xc.grad.cpu(), x64.grad)))
p32, p64 = dict(ref.named_parameters()), dict(ref64.named_parameters())
for name, p in ours.named_parameters():
    if p.numel() < 8: continue
    print("%-60s %9d  ours/32 %.2e  32/64 %.2e  ours/64 %.2e  |g| %.2e" % (name, p.numel(), rl2(p.grad.cpu(), p32[name].grad),
          rl2(p32[name].grad, p64[name].grad), rl2(p.grad.cpu(), p64[name].grad), p64[name].grad.norm().item()))
