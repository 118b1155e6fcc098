"""Data-parallel training over the GPUs of one node: one process per GPU,
`torch.distributed` backend "nccl" (= RCCL over xGMI on ROCm).

The reference has no multi-GPU path (accelerator='dp' pinned to one device,
code/GAN/GAN_final.py:480-485); this is new design.  Slices are independent,
so the step shards over ranks by sample; the only exchange is ONE sum
all-reduce of the flat fp32 gradient buffer of the network being optimised,
right after its backward (G: ~9.7 MB, D: ~10.4 MB at 2-D 256^2), followed by
the fused Adam that folds the 1/world_size in.  BatchNorm statistics stay
per-rank (what DP / DDP without SyncBatchNorm do); running statistics are
rank-local and rank 0's are the ones a checkpoint would hold.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class DataParallelGAN:
    def __init__(self, gan, process_group=None):
        self.gan = gan
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        gan.ddp = self
        if self.world > 1:
            self.broadcast_parameters()

    def broadcast_parameters(self, src: int = 0):
        """Identical replicas: rank `src`'s flat parameter buffers and norm buffers."""
        for net in (self.gan.generator, self.gan.discriminator):
            dist.broadcast(net.store.flat, src, group=self.pg)
            for b in net.buffers():
                dist.broadcast(b, src, group=self.pg)

    def reduce_gradients(self, net, opt):
        if self.world > 1:
            dist.all_reduce(net.store.flat_grad, op=dist.ReduceOp.SUM, group=self.pg)
        opt.grad_scale = 1.0 / self.world

    def sync_logged(self):
        """Mean over ranks of the logged scalars (`self.log(..., sync_dist=True)` of the reference,
        code/GAN/GAN_final.py:266): ONE all-reduce of the four values, on request -- not per step."""
        names = sorted(self.gan.logged)
        if self.world > 1 and names:
            t = torch.stack([self.gan.logged[k].reshape(()).float() for k in names])
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
            t /= self.world
            for i, k in enumerate(names):
                self.gan.logged[k] = t[i]
        return dict(self.gan.logged)


def reduce_flat_gradient(flat_grad: torch.Tensor, world: int, group=None) -> float:
    """The collective on its own (testable with gloo on CPU): sum all-reduce in
    place, returns the scale the optimiser must apply."""
    if world > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


def shard_batch(batch: dict, rank: int, world: int) -> dict:
    """Split a global batch dict along dim 0 (contiguous shards, remainder to the
    low ranks) -- the data-parallel counterpart of the reference's DataLoader
    (code/GAN/GAN_final.py:421-425)."""
    out = {}
    for k, v in batch.items():
        n = v.shape[0]
        base, rem = divmod(n, world)
        lo = rank * base + min(rank, rem)
        hi = lo + base + (1 if rank < rem else 0)
        out[k] = v[lo:hi]
    return out
