"""Data-parallel step on real kernels: two fresh processes share cuda:0 over gloo (the N>1 code path of
bench.py / DataParallelGAN with the collective's transport swapped; config C4's logic).  Checks
broadcast_parameters, reduce_gradients and FusedAdam.grad_scale end to end:
  * replicas identical before and bit-identical after the step,
  * the reduced buffer is the sum of the two ranks' local flat gradients,
  * the update equals Adam applied to their mean,
  * sync_logged returns the rank mean of the logged scalars."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_fit_batch_data_parallel(tmp_path):
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp_worker.py"), str(tmp_path)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    for net in ("G", "D"):
        assert torch.equal(r0[net + "_param_before"], r1[net + "_param_before"]), "broadcast left replicas different"
        assert not torch.equal(r0[net + "_grad_local"], r1[net + "_grad_local"]), "shards produced the same gradient"
        want_sum = r0[net + "_grad_local"] + r1[net + "_grad_local"]
        assert torch.equal(r0[net + "_grad_reduced"], want_sum) and torch.equal(r1[net + "_grad_reduced"], want_sum)
        assert r0[net + "_grad_scale"] == r1[net + "_grad_scale"] == 0.5
        assert torch.equal(r0[net + "_param_after"], r1[net + "_param_after"]), "replicas diverged"
        # the update is torch.optim.Adam's first step on the MEAN gradient
        p = r0[net + "_param_before"].clone().requires_grad_(True)
        opt = torch.optim.Adam([p], lr=5e-4, betas=(0.5, 0.999))
        p.grad = want_sum * 0.5
        opt.step()
        moved = (r0[net + "_param_after"] - r0[net + "_param_before"]).abs().max().item()
        assert moved > 1e-4, "optimizer did not move the parameters"
        err = (r0[net + "_param_after"] - p.detach()).abs().max().item()
        assert err <= 1e-7 + 1e-6 * p.detach().abs().max().item(), (net, err)
    for k in r0["synced_logged"]:
        mean = 0.5 * (r0["local_logged"][k] + r1["local_logged"][k])
        assert abs(r0["synced_logged"][k] - mean) <= 1e-6 * max(1.0, abs(mean)), k
        assert r0["synced_logged"][k] == r1["synced_logged"][k]
