"""The N>1 path on CPU: two gloo ranks run the data-parallel exchange (one sum
all-reduce of a flat gradient buffer + the 1/world scale the fused Adam applies)
and must end with identical, correctly averaged updates."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mpgan_amd.parallel import reduce_flat_gradient, shard_batch
    torch.manual_seed(0)
    flat_param = torch.rand(1000)                       # identical replicas (seeded)
    batch = {"t1w": torch.arange(8, dtype=torch.float32).reshape(8, 1)}
    local = shard_batch(batch, rank, world)["t1w"]      # per-rank samples
    grad = torch.full((1000,), float(local.sum()))      # "gradient" depends on the shard
    scale = reduce_flat_gradient(grad, world)
    flat_param -= 0.1 * grad * scale                    # what the optimiser does with grad_scale
    q.put((rank, flat_param[:4].tolist(), float(grad[0]), scale))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_exchange_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, p0, g0, s0), (r1, p1, g1, s1) = out
    assert s0 == s1 == 0.5
    assert g0 == g1 == float(sum(range(8)))             # sum over both shards
    assert p0 == p1                                     # replicas stay identical
