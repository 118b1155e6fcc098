import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import refmodel as R
from mpgan_amd.networks import PatchDiscriminator
from mpgan_amd.gan_patch import perceptual_loss
ref = R.PatchDiscriminator((1, 16, 16, 16)); R.closed_form_fill_(ref); ref.train()
ours = PatchDiscriminator((1, 16, 16, 16)); ours.load_state_dict(ref.state_dict()); ours.cuda().train()
gen = torch.Generator().manual_seed(21)
xf = (torch.rand(3, 1, 16, 16, 16, generator=gen) * 2 - 1)
xr = torch.rand(3, 1, 16, 16, 16, generator=gen) * 2 - 1
vf, af = ref(xf); _, ar = ref(xr)
v, tf = ours(xf.cuda()); _, tr = ours(xr.cuda())
tot_o = tot_r = 0.0
for k in range(16):
    a, b = tf.tapset.materialize(k).cpu().double(), tr.tapset.materialize(k).cpu().double()
    mine = ((a - b).abs().mean() / a.numel()).item()
    want = (F.l1_loss(ar[k].double(), af[k].double()) / ar[k].numel()).item()
    want32 = (F.l1_loss(ar[k], af[k]) / ar[k].numel()).item()
    tot_o += mine; tot_r += want
    print(k, f"ours(from taps, f64 sum) {mine:.6e} oracle f64 {want:.6e} oracle f32 {want32:.6e} rel {abs(mine-want)/want:.2e}")
print("sum ours", tot_o, "oracle", tot_r, "kernel value", perceptual_loss(tf, tr).item(), "oracle fn", R.perceptual_loss(af, ar).item())
