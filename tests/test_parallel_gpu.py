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


@pytest.mark.parametrize("extra,unit,dtype,peak,per_step", [
    (["--size", "64", "--batch", "4"], "slices/s", "f32", 157.3, 4),
    (["--dims", "3", "--size", "40", "--batch", "2", "--dtype", "bf16"], "volumes/s", "bf16", 2500.0, 2),   # C5's path
])
def test_bench_line_contract_small_size(extra, unit, dtype, peak, per_step):
    """bench.py at a small size (one rank, no CPU baseline): ONE JSON line on stdout with the driver's keys, the
    roofline object -- its kernel chosen by measured time and named as rocprofv3 would name it, the best kernel
    beside it -- the per-pass `phases` of one step and the rank count seen by a collective.  Second case: the
    3-D bf16-storage configuration (BASELINE config C5's code path)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *extra, "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-gfwd", "--dense-min-gflop", "0.05"], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "phases", "dist"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["dtype"] == dtype and j["unit"] == unit
    assert abs(j["value"] - per_step * 1000.0 / j["ms_per_step"]) < 1e-6 * j["value"]
    assert j["dist"]["ranks_seen"] == 1 and j["vs_baseline"] is None
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_role", "best_kernel",
              "dense_kernels", "dense_families", "algorithmic_gb_per_s"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["peak"] == peak and 0.0 < r["frac"] < 1.0 and "kernel" in r["kernel"]
    # the named kernel is the one with the most measured time; the best one is at least as fast
    by_time = max(r["dense_kernels"].items(), key=lambda kv: kv[1]["ms_per_step"])
    assert by_time[0].replace("dgrad of ", "") == r["kernel"], (by_time[0], r["kernel"])
    assert r["best_kernel"]["frac"] >= r["frac"] - 1e-12
    for fam in ("fwd", "dgrad", "wgrad"):
        assert r["dense_families"][fam]["algorithmic_gb_per_s"] > 0.0, fam
    # phases: every network pass of the step, times adding up to about one step
    names = [ph["phase"] for ph in j["phases"]]
    assert sum(n.startswith("G fwd") for n in names) == 2 and sum(n.startswith("D fwd") for n in names) == 3
    assert "G bwd" in names and any(n.startswith("D bwd x2") for n in names)
    total = sum(ph["ms"] for ph in j["phases"])
    assert 0.5 * j["ms_per_step"] <= total <= 2.0 * j["ms_per_step"], (total, j["ms_per_step"])
    assert all(ph["roofline_frac"] is None or 0.0 < ph["roofline_frac"] < 1.0 for ph in j["phases"])
    assert "workload" in j["config"] and "model" not in j["config"]
