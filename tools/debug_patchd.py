import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import refmodel as R
from mpgan_amd.networks import PatchDiscriminator
from mpgan_amd.gan import adversarial_loss
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ref = R.PatchDiscriminator((1, 16, 16, 16)); R.closed_form_fill_(ref); ref.train()
ours = PatchDiscriminator((1, 16, 16, 16)); ours.load_state_dict(ref.state_dict()); ours.cuda().train()
g = torch.Generator().manual_seed(11)
x = (torch.rand(n, 1, 16, 16, 16, generator=g) * 2 - 1).requires_grad_(True)
# oracle with retained grads on conv outputs and bn outputs
acts = {}
h = x
for i, m in enumerate(ref.model_conv):
    h = m(h)
    if i % 3 != 2:
        h.retain_grad(); acts[i] = h
    else:
        acts[i] = h
h2 = h
for i, m in enumerate(ref.model_linear):
    h2 = m(h2)
    if i == 1: h2.retain_grad(); acts["lin1"] = h2
v = h2
l = R.adversarial_loss(v, torch.full_like(v, 0.9)); l.backward()
xc = x.detach().cuda().requires_grad_(True)
vo, taps = ours(xc)
plan = taps.tapset.plan
mism = [int(((taps.tapset.materialize(3*i+1).cpu() > 0) != (acts[3*i+1].detach() > 0)).sum()) for i in range(4)]
lo = adversarial_loss(vo, torch.full_like(vo, 0.9)); lo.backward()
def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item(), ((a-b).norm()/(b.norm()+1e-30)).item()
def cl2nc(t): return t.permute(0, 4, 1, 2, 3).contiguous()
print("dh (lin1 out grad)  max-rel %.3e l2-rel %.3e" % rel(plan.dh.reshape(n, -1), acts["lin1"].grad))
for i in range(4):
    zref = acts[3 * i]
    print(f"layer {i}: dz max-rel %.3e l2-rel %.3e" % rel(cl2nc(plan.gas[i]), zref.grad),
          "| y sign mismatches:", mism[i], "of", acts[3*i+1].numel())
print("x.grad max-rel %.3e l2-rel %.3e" % rel(xc.grad, x.grad))
