"""One rank of the 2-rank data-parallel test (tests/test_parallel_gpu.py starts two fresh interpreters of
this file; they share cuda:0 and talk over gloo).  Builds the GAN with a RANK-DEPENDENT seed, lets
DataParallelGAN broadcast rank 0's replica, runs one fit_batch on this rank's shard of a global batch and
dumps what the parent needs to check the exchange: parameters before, local and reduced flat gradients,
parameters after, logged scalars."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mpgan_amd.gan import GAN
    from mpgan_amd.parallel import DataParallelGAN, shard_batch
    torch.cuda.set_device(0)
    torch.manual_seed(100 + rank)                       # replicas differ until the broadcast
    gan = GAN(1, 64, 64, dimensions=2, n_unet_blocks=2, device="cuda", g_lr=5e-4, d_lr=5e-4)
    with torch.no_grad():                               # keep the head out of sigmoid saturation: non-zero D gradients
        gan.discriminator.model_linear[1].weight.mul_(0.05)
    gan.train()
    ddp = DataParallelGAN(gan)
    opts, _ = gan.configure_optimizers()
    g = torch.Generator().manual_seed(7)                # the same global batch on both ranks
    glob = {"t1w": torch.rand(4, 1, 64, 64, generator=g) * 2 - 1, "t2w": torch.rand(4, 1, 64, 64, generator=g) * 2 - 1}
    batch = {k: v.cuda() for k, v in shard_batch(glob, rank, world).items()}
    rec = {}
    orig = ddp.reduce_gradients

    def recording(net, opt):
        name = "G" if net is gan.generator else "D"
        rec[name + "_param_before"] = net.store.flat.clone().cpu()
        rec[name + "_grad_local"] = net.store.flat_grad.clone().cpu()
        orig(net, opt)
        rec[name + "_grad_reduced"] = net.store.flat_grad.clone().cpu()
        rec[name + "_grad_scale"] = opt.grad_scale

    ddp.reduce_gradients = recording
    gan.fit_batch(batch, 0, opts)
    torch.cuda.synchronize()
    rec["local_logged"] = {k: float(v) for k, v in gan.logged.items()}
    rec["synced_logged"] = {k: float(v) for k, v in ddp.sync_logged().items()}
    for name, net in (("G", gan.generator), ("D", gan.discriminator)):
        rec[name + "_param_after"] = net.store.flat.clone().cpu()
    torch.save(rec, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
