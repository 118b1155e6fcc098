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


def test_bench_line_contract_small_size():
    """bench.py at a small size (one rank, no CPU baseline): ONE JSON line on stdout with the driver's keys, the
    roofline object and a dominant kernel named as rocprofv3 would name it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--size", "64", "--batch", "4", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline", "--no-gfwd"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["dtype"] == "f32" and j["unit"] == "slices/s"
    assert abs(j["value"] - 4 * 1000.0 / j["ms_per_step"]) < 1e-6 * j["value"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "dense_families"):
        assert k in r, k
    # (at this size no launch reaches the probe's 20-GFLOP floor, so `achieved` may be 0: the line still names a kernel)
    assert r["bound"] == "mfma" and r["peak"] == 157.3 and 0.0 <= r["frac"] < 1.0 and "kernel" in r["kernel"]
    assert "workload" in j["config"] and "model" not in j["config"]
