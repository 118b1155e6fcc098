"""Host-side sanitizer run of whole training steps, on the CPU container (no GPU).

    make -C cross-modality-minipig-gan_amd/csrc DRYRUN=1
    LD_PRELOAD=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so \
    ASAN_OPTIONS=detect_leaks=0 python tools/asan_dryrun.py [c3] [c5] [vb] [small] | --pytest [pytest args]

`libmpgan_hip_dry.so` is the library's HOST code alone, built with AddressSanitizer; every kernel launch is a check
of its launch geometry with the arguments still marshalled (csrc/mpgan_common.h).  This script drives the real Python
host path -- plan building, ctypes marshalling, autograd Functions, the two-optimizer `fit_batch`, both stream lanes
of `Program.run` -- through it with CPU tensors standing in for device memory (never dereferenced: nothing launches)
and inert stand-ins for the torch.cuda stream / event objects.  What it can find: heap / stack overruns and
use-after-free in the library's argument handling and geometry builders, ctypes structures that outlive their
Python owners, launches outside the hardware's limits.  What it cannot: anything on the device.

Written for the round-3 review's item 1 (a SIGSEGV inside hipStreamWaitEvent after 117 passing tests, DESIGN.md
section 8): it rules the library's host side in or out as the corruptor.
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MPGAN_LIB_PATH"] = os.path.join(ROOT, "cross-modality-minipig-gan_amd", "libmpgan_hip_dry.so")

import torch  # noqa: E402


class _Ev:
    def __init__(self, *a, **k):
        pass

    def record(self, *a):
        pass

    def wait(self, *a):
        pass

    def synchronize(self):
        pass

    def elapsed_time(self, other):
        return 0.0


class _St:
    cuda_stream = 0
    device = torch.device("cpu")

    def __init__(self, *a, **k):
        pass

    def wait_event(self, e):
        pass

    def wait_stream(self, s):
        pass

    def record_event(self, e=None):
        return e or _Ev()

    def synchronize(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_CUR = _St()
torch.Tensor.is_cuda = property(lambda self: True)
torch.Tensor.record_stream = lambda self, s: None
torch.cuda.current_stream = lambda *a, **k: _CUR
torch.cuda.Stream = _St
torch.cuda.Event = _Ev
torch.cuda.stream = lambda s: _St()
torch.cuda.synchronize = lambda *a, **k: None



class _CudaIsCpu(torch.overrides.TorchFunctionMode):
    """`device="cuda"` / `.cuda()` / `.to("cuda")` keep their tensors on the CPU (--pytest: the -m gpu tests' own
    tensor set-up then runs here)."""

    @staticmethod
    def _is_cuda(d):
        return (isinstance(d, str) and d.startswith("cuda")) or (isinstance(d, torch.device) and d.type == "cuda")

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if self._is_cuda(kwargs.get("device")):
            kwargs["device"] = "cpu"
        name = getattr(func, "__name__", "")
        if name == "cuda":
            return args[0]
        if name == "to" and len(args) > 1 and self._is_cuda(args[1]):
            args = (args[0], "cpu") + tuple(args[2:])
        return func(*args, **kwargs)


import mpgan_amd  # noqa: E402
from mpgan_amd import _lib  # noqa: E402


def counters():
    h = _lib.lib()
    h.mpgan_dry_launches.restype = ctypes.c_long
    h.mpgan_dry_bad_launches.restype = ctypes.c_long
    return h.mpgan_dry_launches(), h.mpgan_dry_bad_launches()


def steps(name, make, batch, n=2):
    t0 = time.time()
    l0, _ = counters()
    gan = make()
    gan.train()
    opts, _ = gan.configure_optimizers()
    for i in range(n):
        gan.fit_batch(batch, i, opts)
    l1, bad = counters()
    print(f"[dry-run] {name}: {n} G+D steps, {l1 - l0} launches checked, {bad} bad, {time.time() - t0:.1f} s", flush=True)
    return gan


def run_pytest(argv):
    """The -m gpu tests themselves through the dry-run library: their numerical assertions fail (nothing computes), but
    every host call they make up to the first failed assertion is marshalled under the sanitizer."""
    import pytest
    torch.cuda.is_available = lambda: True
    torch.cuda.device_count = lambda: 1
    torch.nn.Module.cuda = lambda self, *a, **k: self
    with _CudaIsCpu():
        rc = pytest.main(["-m", "gpu", "-q", "--no-header", "-p", "no:cacheprovider", "--tb=line"] + (argv or ["tests"]))
    total, bad = counters()
    print(f"[dry-run] pytest rc {rc} (assertion failures are expected: nothing computes); {total} launches marshalled, "
          f"{bad} outside the hardware limits; no AddressSanitizer report above means the host side wrote nothing "
          "out of bounds", flush=True)
    sys.exit(1 if bad else 0)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--pytest":
        run_pytest(sys.argv[2:])
    from mpgan_amd import engine
    from mpgan_amd.gan import GAN
    what = set(sys.argv[1:]) or {"small", "c3", "c5", "vb"}

    def pair(*shape):
        return {"t1w": torch.empty(*shape), "t2w": torch.empty(*shape)}

    for single in (False, True):                       # both stream modes of Program.run
        engine._SINGLE_STREAM = single
        tag = "single-stream" if single else "two lanes"
        if "small" in what:
            steps(f"64^2 bs 2 ({tag})", lambda: GAN(1, 64, 64, dimensions=2, n_unet_blocks=2, device=None), pair(2, 1, 64, 64))
            steps(f"24^3 bs 2, instance norm ({tag})",
                  lambda: GAN(1, 40, 40, 40, dimensions=3, n_unet_blocks=1, norm="instance", device=None), pair(2, 1, 40, 40, 40))
        if "c3" in what:
            steps(f"C3 256^2 bs 16 fp32 ({tag})", lambda: GAN(1, 256, 256, dimensions=2, device=None), pair(16, 1, 256, 256))
        if "c5" in what:
            steps(f"C5 128^3 bs 4, bf16 storage in D ({tag})",
                  lambda: GAN(1, 128, 128, 128, dimensions=3, storage_dtype="bf16", device=None), pair(4, 1, 128, 128, 128), n=1)
    if "vb" in what:
        from mpgan_amd import gan_patch
        engine._SINGLE_STREAM = False
        g = gan_patch.GAN(1, 64, 64, 64, device=None, num_samples=8, crop_seed=1)
        g.train()
        opts, _ = g.configure_optimizers()
        l0, _ = counters()
        g.fit_batch(pair(2, 1, 64, 64, 64), 0, opts)
        l1, bad = counters()
        print(f"[dry-run] variant B 64^3 bs 2 x 8 patches: {l1 - l0} launches checked, {bad} bad", flush=True)
    total, bad = counters()
    print(f"[dry-run] done: {total} launches, {bad} outside the hardware limits; no AddressSanitizer report above "
          "means the host side wrote nothing out of bounds", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
