"""Execution plans for the GAN hot path: pre-bound sequences of C-ABI calls.

A *plan* owns every activation / gradient buffer one forward (+ backward) of a
network needs for a fixed (batch, spatial) shape, and two *programs*: lists of
(C function, argument tuple) built once.  Running a network is a tight loop of
ctypes calls on the current stream -- no per-op Python logic, no per-op
allocation, graph-capturable.  Data layout in HBM: channels-last
(N, D, H, W, C) fp32; concatenations are channel slices of one buffer that the
producers write directly; only RAW conv outputs + per-channel norm vectors are
kept (BatchNorm/InstanceNorm + PReLU/LeakyReLU are applied by the consumer's
load prologue).

Graph of one U-Net follows MONAI 0.4.0 `UNet(num_res_units=2)` as the reference
instantiates it (code/GAN/GAN_final.py:106-114); the
discriminator follows code/GAN/GAN_final.py:159-209.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import ops
from ._lib import lib
from .ops import ACT_LEAKY, ACT_NONE, ConvGeom, Prologue


# --------------------------------------------------------------------------
# programs
# --------------------------------------------------------------------------
class KernelProbe:
    """HIP-event timing of tagged launches (bench.py's live roofline figure):
    events are recorded on the stream the kernels are launched on."""

    def __init__(self, want=None, detail=False, min_flops=0.0):
        self.want = want            # None = every tagged call, else a set of kernel names
        self.detail = detail        # True: time EVERY call, keyed "<c entry point> <geometry>" (tools/profile_step.py)
        self.min_flops = min_flops  # only launches with at least this much algorithmic work (the dense families)
        self.samples = []           # (kernel, flops, ev0, ev1, algorithmic bytes)

    def summary(self):
        """kernel -> dict(calls, ms, flops): durations read after a synchronize."""
        out = {}
        for k, fl, e0, e1, by in self.samples:
            d = out.setdefault(k, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
            d["calls"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += by
        return out


_PROBE: Optional[KernelProbe] = None


def set_probe(p: Optional[KernelProbe]):
    global _PROBE
    _PROBE = p


_SIDE = {}


def side_stream(device) -> "torch.cuda.Stream":
    """The second HIP stream of this device: weight-gradient kernels of the generator run here,
    next to the (latency-bound) norm-backward / backward-data chain on the caller's stream."""
    key = torch.device(device).index or 0
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


_SINGLE_STREAM = bool(os.environ.get("MPGAN_SINGLE_STREAM"))
_ALWAYS_PACK = bool(os.environ.get("MPGAN_DBG_ALWAYS_PACK"))
# BatchNorm statistics through integer accumulators + fold-on-load in the first consumer (csrc/norm_fold.h) instead of
# partial rows + a finalize launch.  Built, tested and MEASURED: the 66 finalize launches it
# removes from a generator forward (4.7-6.3 us each + a ~1.5 us boundary) are paid back in full by what it adds to
# every block of the neighbouring kernels -- the fold's load -> barrier -> double-precision finalize -> barrier sits on
# each consumer block's critical path (+3.6-5 us per launch) and the atomics on the producer's tail (+1-4 us) -- and
# same-address atomics make a 1-channel layer 8x slower (convt_quad: 28 -> 225 us).  Off unless MPGAN_ACC_STATS=1.
_ACC_STATS = bool(os.environ.get("MPGAN_ACC_STATS"))
_FUSE_DOWN = not os.environ.get("MPGAN_DBG_NO_FUSE_DOWN")   # ResidualUnit: first conv + residual conv as one launch
# The discriminator's weight gradients run on the caller's stream, one after the other with the backward-data
# launches (lane 0).  Both families are matrix-bound: side by side on two streams each only gets half the chip --
# A/B on one box, 20-step benches: 55.25 / 55.27 ms (second stream) vs 55.18 / 55.35 ms (one stream) -- and the
# per-launch figures the bench line reports are then contention-free.  MPGAN_D_WGRAD_SIDE=1 puts them back on lane 1.
_D_WGRAD_LANE = 1 if os.environ.get("MPGAN_D_WGRAD_SIDE") else 0


def same_geom(a, b) -> bool:
    return (a.n, a.in_dhw, a.cin, a.k, a.stride, a.pad, a.transposed) == (b.n, b.in_dhw, b.cin, b.k, b.stride, b.pad,
                                                                         b.transposed)


class Mark:
    """A point on the side stream another program section can wait for."""
    __slots__ = ("event", "armed")

    def __init__(self):
        self.event = torch.cuda.Event()
        self.armed = False


class Program:
    """A frozen list of C calls; `run` appends the stream and checks status.

    Calls carry a lane: 0 = the caller's stream, 1 = the side stream.  A lane-1 call first makes
    the side stream wait for everything enqueued so far on the caller's stream (event), so it
    may read anything produced before it; `join()` makes the caller's stream wait for the side
    stream.  The emitter must place a join before any buffer a lane-1 call reads is rewritten
    and before its results are consumed."""
    __slots__ = ("calls", "keep", "names", "tags", "descs", "lanes", "events")

    def __init__(self):
        self.calls = []
        self.keep = []
        self.names = []
        self.tags = []
        self.descs = []
        self.lanes = []
        self.events = {}

    def add(self, name, fn, *args, keep=(), tag=None, desc="", lane=0):
        self.calls.append((fn, args))
        self.names.append(name)
        self.keep.append(keep)
        self.tags.append(tag)
        self.descs.append(desc)
        self.lanes.append(lane)

    def join(self):
        """The caller's stream waits for everything enqueued so far on the side stream."""
        if self.calls and self.calls[-1] == (None, ("join",)):
            return
        self.add("join", None, "join")

    def mark(self, m: "Mark"):
        """Record `m` on the side stream here (a later `wait(m)` orders the caller's stream after it)."""
        self.add("mark", None, "mark", m)

    def wait(self, m: "Mark"):
        self.add("wait", None, "wait", m)

    def extend(self, other: "Program"):
        self.calls += other.calls
        self.names += other.names
        self.keep += other.keep
        self.tags += other.tags
        self.descs += other.descs
        self.lanes += other.lanes

    def rebind(self, ptr: int, slot: "C.c_void_p") -> int:
        """Replace every pointer argument equal to `ptr` by the mutable `slot` (a ctypes pointer object whose
        .value the caller sets before `run`): the caller's own tensor is then read / written in place instead
        of a plan-owned staging buffer.  Returns the number of calls touched."""
        hits = 0
        for i, (fn, args) in enumerate(self.calls):
            if fn is None or not any(type(a) is int and a == ptr for a in args):
                continue
            self.calls[i] = (fn, tuple(slot if (type(a) is int and a == ptr) else a for a in args))
            hits += 1
        return hits

    def _event(self, i):
        ev = self.events.get(i)
        if ev is None:
            ev = self.events[i] = torch.cuda.Event()
        return ev

    def run(self, stream=None):
        main = torch.cuda.current_stream()
        s = main.cuda_stream if stream is None else stream
        probe = _PROBE
        lanes = self.lanes
        multi = (stream is None and not _SINGLE_STREAM and not (probe is not None and probe.detail)
                 and 1 in lanes)
        if probe is None and not multi:            # plain program on one stream: the tight loop
            for i, (fn, args) in enumerate(self.calls):
                if fn is not None:
                    rc = fn(*args, s)
                    if rc:
                        raise RuntimeError(f"{self.names[i]} failed (status {rc}): {lib().mpgan_last_error().decode()}")
            return
        side = side_stream(main.device) if multi else None
        s1 = side.cuda_stream if multi else s
        side_busy = False
        main_dirty = True                          # the caller's stream has work the side stream has not waited for
        for i, (fn, args) in enumerate(self.calls):
            if fn is None:                         # stream bookkeeping
                kind = args[0]
                if not multi:
                    continue
                if kind == "join":
                    if side_busy:
                        ev = self._event(i)
                        ev.record(side)
                        main.wait_event(ev)
                        side_busy = False
                elif kind == "mark":
                    args[1].event.record(side)
                    args[1].armed = True
                elif args[1].armed:                # wait
                    main.wait_event(args[1].event)
                continue
            on_side = multi and lanes[i] == 1
            if on_side:
                if main_dirty:
                    ev = self._event(i)
                    ev.record(main)
                    side.wait_event(ev)
                    main_dirty = False
                side_busy = True
            else:
                main_dirty = True
            tag = self.tags[i]
            timed = False
            if probe is not None:
                if probe.detail:
                    tag = (f"{self.names[i]} {self.descs[i]}".strip(), tag[1] if tag else 0.0,
                           tag[2] if tag and len(tag) > 2 else 0.0)
                timed = (tag is not None and (probe.want is None or tag[0] in probe.want)
                         and tag[1] >= probe.min_flops)
            if timed:                              # events go on the stream the kernel is launched on
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side if on_side else main)
            rc = fn(*args, s1 if on_side else s)
            if rc:
                raise RuntimeError(f"{self.names[i]} failed (status {rc}): {lib().mpgan_last_error().decode()}")
            if timed:
                e1.record(side if on_side else main)
                probe.samples.append((tag[0], tag[1], e0, e1, tag[2] if len(tag) > 2 else 0.0))
        if side_busy:                              # never leave work un-joined behind a program
            ev = self._event(-1)
            ev.record(side)
            main.wait_event(ev)

    def __len__(self):
        return len(self.calls)


def conv_macs(g: ConvGeom) -> int:
    """Algorithmic multiply-accumulates of one conv (transposed conv counted input-side)."""
    grid = g.in_dhw if g.transposed else g.out_dhw
    return g.n * grid[0] * grid[1] * grid[2] * g.cin * g.cout * g.taps


def conv_bytes(g: ConvGeom, esize: int = 4) -> int:
    """Algorithmic HBM bytes of one conv in any direction: the gathered side read or written once
    plus the dense side read or written once; weights and norm vectors are not counted."""
    i, o = g.in_dhw, g.out_dhw
    return g.n * (i[0] * i[1] * i[2] * g.cin + o[0] * o[1] * o[2] * g.cout) * esize


def gather_kernel_name(g: ConvGeom, backward_data: bool, has_pro: bool, per_sample_norm: bool = False,
                       fast_leaky: bool = False) -> str:
    """The kernel symbol (as rocprofv3 prints it, minus `void mpgan::` and the argument list)
    the C dispatcher picks for this conv (mpgan_conv_variant + launch_gather's rules)."""
    gc = g.c()
    v = int(lib().mpgan_conv_variant(C.byref(gc), int(backward_data),
                                     (2 if per_sample_norm else (3 if fast_leaky else 1)) if has_pro else 0))
    dma = v >= 3000                              # DMA-staged form (prologue-free gathers with many tiles)
    v = v - 3000 if dma else v
    ks2 = v >= 2000                              # K axis split over two wave groups inside the block
    v = v - 2000 if ks2 else v
    fast = v >= 1000                             # mask-free instance of the pipelined kernel
    v = v - 1000 if fast else v
    if v == 1:
        return "thin_cin1_kernel"
    if v == 2:
        return "thin_cout1_kernel"
    if v == 18:
        return f"gather_patch3d_c16_kernel<{'true' if has_pro else 'false'}, {'true' if g.mm_bf16 else 'false'}>"
    cin_eff = g.cout if backward_data else g.cin
    if v in (16, 17):
        cout_eff = g.cin if backward_data else g.cout
        return (f"gather_patch_kernel<{cin_eff}, {1 if has_pro else 0}, {'true' if cout_eff <= 16 else 'false'}, "
                f"{'true' if v == 17 else 'false'}>")
    tm, tn, wn = {128: (2, 2, 2), 64: (1, 2, 1), 32: (1, 1, 1)}[v]
    if dma:
        return f"gather_conv_dma_kernel<{v}, {tm}, {tn}, {wn}, 2>"
    if cin_eff % 32 == 0 or cin_eff == 16:      # software-pipelined main kernel <BN, TM, TN, WN, WRAPS, PRO>
        pro = 0 if not has_pro else (2 if per_sample_norm else (3 if fast_leaky else 1))
        return (f"gather_conv_pipe_kernel<{v}, {tm}, {tn}, {wn}, {1 if cin_eff % 32 == 0 else 2}, {pro}, "
                f"{'true' if fast else 'false'}, {2 if ks2 else 1}, {'true' if g.mm_bf16 else 'false'}>")
    return f"gather_conv_kernel<{v}, {tm}, {tn}, {wn}, {'false' if cin_eff % 4 == 0 else 'true'}>"


def _ld(t):
    return 0 if t is None else ops._cl(t, "plan tensor")[2]


def _p(t):
    return None if t is None else t.data_ptr()


# --------------------------------------------------------------------------
# parameter storage
# --------------------------------------------------------------------------
class ConvRec:
    """One conv / transposed conv / linear-as-conv weight in the flat store."""
    __slots__ = ("mod", "transposed", "cout", "cin", "taps", "w_off", "fwd_off", "bwd_off", "tco", "tco_off")


class FusedRec:
    """Two ConvNd of one geometry reading the same input (a ResidualUnit's first conv and its strided
    residual conv), packed as ONE conv over their concatenated output channels: forward weights
    [CoutA + CoutB][tap][Cin] and the concatenated biases, both in the packed buffer."""
    __slots__ = ("a", "b", "w_off", "b_off", "cout", "cin", "taps")


class ParamStore:
    """All parameters of a network as views of ONE flat fp32 buffer (+ a flat
    gradient buffer the kernels accumulate into, the buffer RCCL all-reduces and
    the fused Adam steps over) and the packed [Cout][tap][Cin] conv weights."""

    def __init__(self, module: nn.Module):
        self.module = module
        self.convs: List[ConvRec] = []
        self.fused: List[FusedRec] = []
        self._by_mod = {}
        self.flat = None
        self.flat_grad = None
        self.packed = None
        self.table = None
        self.version = 0
        self.frozen = False
        self._touched = 0
        self._pack_state = {}
        self.flatten()
        for m in module.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv3d, nn.ConvTranspose2d, nn.ConvTranspose3d)):
                tr = isinstance(m, (nn.ConvTranspose2d, nn.ConvTranspose3d))
                taps = 1
                for k in m.kernel_size:
                    taps *= k
                self.register_conv(m, cout=m.out_channels, cin=m.in_channels, taps=taps, transposed=tr)

    def flatten(self):
        params = list(self.module.parameters())
        dev = params[0].device
        offs, total = [], 0
        for p in params:
            total = (total + 3) // 4 * 4  # keep every tensor 16-byte aligned
            offs.append(total)
            total += p.numel()
        total = (total + 3) // 4 * 4
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self._off = {}
        with torch.no_grad():
            for p, o in zip(params, offs):
                flat[o:o + p.numel()].copy_(p.detach().reshape(-1))
                p.data = flat[o:o + p.numel()].view(p.shape)
                p.grad = grad[o:o + p.numel()].view(p.shape)
                self._off[id(p)] = o
        self.flat, self.flat_grad = flat, grad
        self._params = params
        self.version += 1
        self._build_pack_table()

    def attach_grads(self):
        """Re-point .grad at the flat gradient views (an optimizer's
        zero_grad(set_to_none=True) detaches them)."""
        lost = [p for p in self._params
                if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * self._off[id(p)]]
        if not lost:
            return
        # A gradient that was set to None (torch.optim's zero_grad default) counts as zeroed: the kernels ACCUMULATE
        # into the flat buffer, so its slice must not keep the previous step's values.
        all_none = len(lost) == len(self._params) and all(p.grad is None for p in lost)
        if all_none:
            self.flat_grad.zero_()
        for p in lost:
            o = self._off[id(p)]
            view = self.flat_grad[o:o + p.numel()].view(p.shape)
            if p.grad is None:
                if not all_none:
                    view.zero_()
            else:
                view.copy_(p.grad)                 # a foreign gradient tensor: adopt its values
            p.grad = view

    def offset(self, p) -> int:
        return self._off[id(p)]

    def grad_view(self, p) -> Optional[torch.Tensor]:
        if p is None:                            # e.g. the gamma / beta of a non-affine instance norm
            return None
        o = self._off[id(p)]
        return self.flat_grad[o:o + p.numel()].view(p.shape)

    def register_conv(self, mod, *, cout, cin, taps, transposed=False, tco=False):
        """Register a weight for packing.  All weights must be registered before
        the first plan is built (packed offsets are baked into the programs)."""
        if id(mod) in self._by_mod:
            r = self._by_mod[id(mod)]
            assert (r.cout, r.cin, r.taps, r.transposed) == (cout, cin, taps, transposed)
            return r
        if self.frozen:
            raise RuntimeError("ParamStore: conv registered after plans were built")
        r = ConvRec()
        r.mod, r.transposed, r.cout, r.cin, r.taps = mod, transposed, cout, cin, taps
        r.tco, r.tco_off = tco, -1
        self.convs.append(r)
        self._by_mod[id(mod)] = r
        if self.flat is not None:
            self._build_pack_table()
        return r

    def register_fused(self, conv_a, conv_b) -> FusedRec:
        """The concatenated forward pack of two registered ConvNd with equal input channels and kernel."""
        for f in self.fused:
            if f.a.mod is conv_a and f.b.mod is conv_b:
                return f
        if self.frozen:
            raise RuntimeError("ParamStore: fused pair registered after plans were built")
        ra, rb = self._by_mod[id(conv_a)], self._by_mod[id(conv_b)]
        assert (ra.cin, ra.taps, ra.transposed, rb.transposed) == (rb.cin, rb.taps, False, False)
        assert conv_a.bias is not None and conv_b.bias is not None
        f = FusedRec()
        f.a, f.b, f.cout, f.cin, f.taps = ra, rb, ra.cout + rb.cout, ra.cin, ra.taps
        self.fused.append(f)
        self._build_pack_table()
        return f

    def wp_fused(self, f: FusedRec) -> torch.Tensor:
        return self.packed[f.w_off:f.w_off + f.cout * f.cin * f.taps]

    def bias_fused(self, f: FusedRec) -> torch.Tensor:
        return self.packed[f.b_off:f.b_off + f.cout]

    def _build_pack_table(self):
        if not self.convs:
            return
        rows, off = [], 0
        for r in self.convs:
            n = r.cout * r.cin * r.taps
            assert r.mod.weight.numel() == n, (r.mod, n)
            r.w_off = self.offset(r.mod.weight)
            r.fwd_off = off
            off += (n + 3) // 4 * 4
            r.bwd_off = off
            off += (n + 3) // 4 * 4
            rows.append([r.w_off, r.fwd_off, r.cout, r.cin, r.taps, int(r.transposed), 0, 0])
            rows.append([r.w_off, r.bwd_off, r.cout, r.cin, r.taps, int(r.transposed), 1, 0])
            if r.tco:
                r.tco_off = off
                off += (n + 3) // 4 * 4
                rows.append([r.w_off, r.tco_off, r.cout, r.cin, r.taps, int(r.transposed), 2, 0])
        for f in self.fused:
            na, nb_ = f.a.cout * f.cin * f.taps, f.b.cout * f.cin * f.taps
            f.w_off = off
            rows.append([f.a.w_off, off, f.a.cout, f.cin, f.taps, 0, 0, 0])
            rows.append([f.b.w_off, off + na, f.b.cout, f.cin, f.taps, 0, 0, 0])
            off += (na + nb_ + 3) // 4 * 4
            f.b_off = off                      # biases: a "conv" with one tap and one input channel copies them
            rows.append([self.offset(f.a.mod.bias), off, f.a.cout, 1, 1, 0, 0, 0])
            rows.append([self.offset(f.b.mod.bias), off + f.a.cout, f.b.cout, 1, 1, 0, 0, 0])
            off += (f.cout + 3) // 4 * 4
        dev = self.flat.device
        self.packed = torch.empty(off, dtype=torch.float32, device=dev)
        self.table = torch.tensor(rows, dtype=torch.int64, device=dev)
        self._max_elems = max(r.cout * r.cin * r.taps for r in self.convs)
        self._pack_state.clear()               # a new packed buffer holds nothing yet

    def emit_pack(self, prog: Program):
        """The repack launch, skipped while the parameters are known to be unchanged since the last pack.
        Every Parameter keeps its OWN version counter (`p.data = flat[...]` aliases the storage, not the counter),
        so the key holds the counters of all packed parameters -- `load_state_dict`, `nn.init.*`, `weight.mul_()`
        and a stock `torch.optim.Adam` bump those -- next to the flat buffer's counter (writes through `store.flat`)
        and `touch()` (the fused Adam writes through a raw pointer).  Back-to-back forwards on constant weights
        (inference, the discriminator's real/fake pair of one step) then pack once."""
        self.frozen = True
        fn = lib().mpgan_pack_weights
        args = (self.flat.data_ptr(), self.packed.data_ptr(), self.table.data_ptr(), self.table.shape[0], self._max_elems)
        state = self._pack_state
        packed_params = [r.mod.weight for r in self.convs]
        for f in self.fused:
            packed_params += [f.a.mod.bias, f.b.mod.bias]

        def pack_if_stale(stream):
            key = (self.flat._version, self.version, self._touched, sum(p._version for p in packed_params))
            if state.get("key") == key and not _ALWAYS_PACK:
                return 0
            rc = fn(*args, stream)
            if rc == 0:
                state["key"] = key
            return rc

        prog.add("pack_weights", pack_if_stale, keep=(self.flat, self.packed, self.table))

    def touch(self):
        """Parameters were written behind torch's back (raw-pointer kernel)."""
        self._touched += 1

    def wp(self, rec: ConvRec) -> torch.Tensor:
        return self.packed[rec.fwd_off:rec.fwd_off + rec.cout * rec.cin * rec.taps]

    def wp_bwd(self, rec: ConvRec) -> torch.Tensor:
        return self.packed[rec.bwd_off:rec.bwd_off + rec.cout * rec.cin * rec.taps]

    def wp_tco(self, rec: ConvRec) -> torch.Tensor:
        return self.packed[rec.tco_off:rec.tco_off + rec.cout * rec.cin * rec.taps]


# --------------------------------------------------------------------------
# emit helpers
# --------------------------------------------------------------------------
class NormBuf:
    """Device vectors of one norm layer in one plan."""

    def __init__(self, n, c, instance, dev):
        m = n * c if instance else c
        mm = (m + 3) // 4 * 4
        buf = torch.zeros(6, mm, device=dev)
        self.scale, self.shift, self.mean, self.invstd, self.c1, self.c2 = (buf[i, :m] for i in range(6))
        self.n, self.c, self.instance = n, c, instance
        # accumulator statistics (csrc/norm_fold.h): set by the plan for layers whose producer and first consumer
        # both support it; `folded` flips when the first consumer has been emitted
        self.acc = None            # int64 view [replicas][4][cstride]
        self.acc_cstride = 0
        self.acc_count = 0
        self.norm_mod = None
        self.folded = False

    def fold(self):
        """The mpgan_norm_fold the FIRST consumer of these statistics passes; None once they are published."""
        if self.acc is None or self.folded:
            return None
        self.folded = True
        return ops.make_fold(self.acc, ops.ACC_REPLICAS, self.acc_cstride, self.acc_count, self.norm_mod, self.scale,
                             self.shift, self.mean, self.invstd)

    def prologue(self, act, slope=1.0, slope_t=None) -> Prologue:
        return Prologue(self.scale, self.shift, self.c if self.instance else 0, act, slope, slope_t)


def _fast_leaky(pro) -> bool:
    """launch_pipe_pro's rule for the max(y, slope*y) form: per-channel norm + LeakyReLU whose slope the
    host knows (no device PReLU weight) to lie in [0, 1]."""
    return bool(pro is not None and not pro.n_stride and pro.act == ACT_LEAKY and pro.slope_t is None
                and 0.0 <= pro.slope <= 1.0)


def _gdesc(g: ConvGeom) -> str:
    return (f"{g.cin}->{g.cout} k{'x'.join(map(str, g.k))} s{g.stride[-1]}{'T' if g.transposed else ''} "
            f"in{'x'.join(map(str, g.in_dhw))}")


def emit_conv_fwd(prog, g: ConvGeom, x, wp, bias, y, pro=None, resid=None, tanh=False, stats=None, lane=0, fold=None,
                  stats_acc=None):
    """fold: mpgan_norm_fold of the producer of x (its statistics are finalised by this launch);
    stats_acc: int64 accumulators that receive this conv's own statistics instead of partial rows."""
    ops._check_in_out(g, x, y, "plan conv_forward")
    gc = g.c()
    pc = pro.c() if pro is not None else None
    tag = (gather_kernel_name(g, False, pro is not None, bool(pro is not None and pro.n_stride), _fast_leaky(pro)),
           2.0 * conv_macs(g), conv_bytes(g))
    if fold is None and stats_acc is None:
        prog.add("conv_forward", lib().mpgan_conv_forward, C.byref(gc), x.data_ptr(), _ld(x), wp.data_ptr(), _p(bias),
                 C.byref(pc) if pc is not None else None, _p(resid), _ld(resid), int(tanh), _p(stats), y.data_ptr(),
                 _ld(y), keep=(gc, pc, x, wp, bias, y, resid, pro), desc=_gdesc(g), tag=tag, lane=lane)
        return
    assert stats is None or stats_acc is None
    prog.add("conv_forward_fold", lib().mpgan_conv_forward_fold, C.byref(gc), x.data_ptr(), _ld(x), wp.data_ptr(),
             _p(bias), C.byref(pc) if pc is not None else None, C.byref(fold) if fold is not None else None, _p(resid),
             _ld(resid), int(tanh), _p(stats), _p(stats_acc), ops.ACC_REPLICAS if stats_acc is not None else 0,
             y.data_ptr(), _ld(y), keep=(gc, pc, fold, x, wp, bias, y, resid, pro, stats_acc), desc=_gdesc(g), tag=tag,
             lane=lane)


def emit_conv_dgrad(prog, g: ConvGeom, dy, wp_bwd, dx, resid=None):
    ops._check_in_out(g, dx, dy, "plan conv_backward_data")
    gc = g.c()
    prog.add("conv_backward_data", lib().mpgan_conv_backward_data, C.byref(gc), dy.data_ptr(), _ld(dy),
             wp_bwd.data_ptr(), _p(resid), _ld(resid), dx.data_ptr(), _ld(dx), keep=(gc, dy, wp_bwd, dx, resid),
             desc=_gdesc(g), tag=("dgrad:" + gather_kernel_name(g, True, False), 2.0 * conv_macs(g), conv_bytes(g)))


_FUSE_BWD_STATS = not os.environ.get("MPGAN_DBG_NO_FUSE_BWD_STATS")
# The bf16 path's counterpart (mpgan_conv_backward_data_stats_bf16) is OFF by default: measured at config C5 on one box
# (tools/phase_times.py, round 4), D's backward passes take 57.9 ms per step with the sums fused into the backward-data
# epilogues and 57.2 ms with the separate norm_bwd_reduce_bf16 launches -- the bf16 kernels run one block per CU, so the
# epilogue's extra 16-byte z loads are exposed, while the separate pass streams g and z with the whole chip at the HBM rate.
_FUSE_BWD_STATS_BF16 = bool(os.environ.get("MPGAN_FUSE_BWD_STATS_BF16"))


def emit_conv_dgrad_stats(prog, g: ConvGeom, dy, wp_bwd, dx, z, nb: "NormBuf", slope: float, partials) -> int:
    """Backward-data whose epilogue also leaves the norm-backward partial sums of dx against z (the raw output of
    the layer in front, BatchNorm + LeakyReLU(slope)): returns the number of rows, 0 when this geometry has no
    fused sums (then the plain launch is emitted and the caller runs the reduce pass)."""
    rows = ops.conv_bwd_stats_rows(g) if _FUSE_BWD_STATS and not nb.instance else 0
    if not rows or partials.numel() < rows * 3 * g.cin:
        emit_conv_dgrad(prog, g, dy, wp_bwd, dx)
        return 0
    ops._check_in_out(g, dx, dy, "plan conv_backward_data_stats")
    gc = g.c()
    prog.add("conv_backward_data", lib().mpgan_conv_backward_data_stats, C.byref(gc), dy.data_ptr(), _ld(dy),
             wp_bwd.data_ptr(), dx.data_ptr(), _ld(dx), z.data_ptr(), _ld(z), nb.scale.data_ptr(), nb.shift.data_ptr(),
             nb.mean.data_ptr(), nb.invstd.data_ptr(), ACT_LEAKY, float(slope), partials.data_ptr(),
             keep=(gc, dy, wp_bwd, dx, z, nb, partials), desc=_gdesc(g),
             tag=("dgrad:" + gather_kernel_name(g, True, False), 2.0 * conv_macs(g), conv_bytes(g)))
    return rows


def emit_conv_wgrad(prog, g: ConvGeom, x, dy, dw, ws, pro=None, dbias=None, lane=0):
    """dW += wgrad; for a ConvNd, dbias += colsum(dy) rides along in the same kernel."""
    ops._check_in_out(g, x, dy, "plan conv_backward_weight")
    gc = g.c()
    pc = pro.c() if pro is not None else None
    need = ops.conv_wgrad_workspace(g)
    assert ws.numel() * 4 >= need, "wgrad workspace too small"
    prog.add("conv_backward_weight", lib().mpgan_conv_backward_weight, C.byref(gc), x.data_ptr(), _ld(x),
             C.byref(pc) if pc is not None else None, dy.data_ptr(), _ld(dy), dw.data_ptr(), _p(dbias), 1.0,
             ws.data_ptr(), ws.numel() * 4, keep=(gc, pc, x, dy, dw, dbias, ws, pro), desc=_gdesc(g),
             tag=("wgrad_kernel", 2.0 * conv_macs(g), conv_bytes(g)), lane=lane)


def emit_bias_grad(prog, dy, db, partials):
    """db += column sums of dy (two-stage, deterministic)."""
    n, P, ld = ops._cl(dy, "bias grad")
    c = dy.shape[-1]
    chunks = ops.stats_chunks(P, c)
    assert partials.numel() >= n * chunks * 2 * c
    L = lib()
    prog.add("channel_stats", L.mpgan_channel_stats, dy.data_ptr(), ld, n, P, c, partials.data_ptr(),
             keep=(dy, partials))
    prog.add("reduce_partials", L.mpgan_reduce_partials, partials.data_ptr(), n * chunks, 2 * c, c, db.data_ptr(), 1.0,
             keep=(db,))


def emit_norm_stats(prog, z, nb: NormBuf, norm_mod, partials, eps=1e-5, momentum=0.1):
    """channel_stats + finalize: scale/shift for the consumer's prologue; running
    stats updated in place (BatchNorm, train mode)."""
    n, P, ld = ops._cl(z, "norm stats")
    c = z.shape[-1]
    chunks = ops.stats_chunks(P, c)
    assert partials.numel() >= n * chunks * 2 * c
    L = lib()
    rm = getattr(norm_mod, "running_mean", None)
    rv = getattr(norm_mod, "running_var", None)
    nbt = getattr(norm_mod, "num_batches_tracked", None)
    eps = getattr(norm_mod, "eps", eps)
    momentum = getattr(norm_mod, "momentum", momentum)
    if nb.instance:
        rm = rv = nbt = None
    prog.add("channel_stats", L.mpgan_channel_stats, z.data_ptr(), ld, n, P, c, partials.data_ptr(),
             keep=(z, partials))
    prog.add("norm_finalize", L.mpgan_norm_finalize, partials.data_ptr(), n, chunks, c, P, int(nb.instance),
             _p(norm_mod.weight), _p(norm_mod.bias), float(eps), float(momentum), _p(rm), _p(rv), _p(nbt),
             nb.scale.data_ptr(), nb.shift.data_ptr(), nb.mean.data_ptr(), nb.invstd.data_ptr(),
             keep=(norm_mod.weight, norm_mod.bias, rm, rv, nbt, nb))


def emit_conv_fwd_norm(prog, g: ConvGeom, x, wp, bias, y, nb: NormBuf, norm_mod, partials, pro=None, training=True,
                       eval_norms=None, c_norm=None, fold=None):
    """Conv whose raw output feeds a norm layer: BatchNorm statistics come out of the
    conv's own epilogue (one finalize launch follows); so do InstanceNorm's when the partial rows
    fall into per-sample groups; otherwise a separate statistics pass runs.  In eval
    mode BatchNorm's scale/shift come from the running statistics instead.
    c_norm: the norm covers only the first c_norm output channels (a conv fused with its
    ResidualUnit's residual conv: the trailing channels are the un-normalised residual branch)."""
    c = g.cout if c_norm is None else c_norm
    y_norm = y if c_norm is None else y[..., :c]
    if training and nb.acc is not None:
        # accumulator statistics: no finalize launch; the first consumer of `nb` folds them (NormBuf.fold)
        nb.norm_mod = norm_mod
        nb.acc_cstride = g.cout
        n_, P_, _ = ops._cl(y, "conv+norm")
        nb.acc_count = n_ * P_
        emit_conv_fwd(prog, g, x, wp, bias, y, pro=pro, fold=fold, stats_acc=nb.acc)
        return
    if not training and not nb.instance:
        emit_conv_fwd(prog, g, x, wp, bias, y, pro=pro)
        if eval_norms is not None:       # input-independent: the plan folds all of them into one launch up front
            eval_norms.append((norm_mod, nb, c))
            return
        prog.add("norm_from_running", lib().mpgan_norm_from_running, _p(norm_mod.weight), _p(norm_mod.bias),
                 norm_mod.running_mean.data_ptr(), norm_mod.running_var.data_ptr(), float(norm_mod.eps), c,
                 nb.scale.data_ptr(), nb.shift.data_ptr(), nb.mean.data_ptr(), nb.invstd.data_ptr(),
                 keep=(norm_mod.weight, norm_mod.bias, norm_mod.running_mean, norm_mod.running_var, nb))
        return
    code = 0 if pro is None else (2 if pro.n_stride else 1)
    rows = ops.conv_stats_rows(g, code)
    n, P, _ = ops._cl(y, "conv+norm")
    if nb.instance and rows:
        # InstanceNorm needs the partial rows grouped by sample: true for the patch kernel (sample is the
        # slowest block index), for 128-pixel tiles of a non-transposed conv when 128 | P, and for the
        # thin kernel's 256-pixel blocks when 256 | P
        v = ops.conv_variant(g, False, code)
        v = v - 3000 if v >= 3000 else v
        v = v - 2000 if v >= 2000 else v
        grouped = (v in (16, 17) or (v == 1 and P % 256 == 0)
                   or (v in (32, 64, 128) and not g.transposed and P % 128 == 0))
        if not grouped or rows % n:
            rows = 0
    if rows == 0:
        emit_conv_fwd(prog, g, x, wp, bias, y, pro=pro, fold=fold)
        emit_norm_stats(prog, y_norm, nb, norm_mod, partials)
        return
    W = g.cout                                  # partial rows are [2][W] wide
    assert partials.numel() >= (rows + 32) * 2 * W
    emit_conv_fwd(prog, g, x, wp, bias, y, pro=pro, stats=partials, fold=fold)
    if nb.instance:
        prog.add("norm_finalize", lib().mpgan_norm_finalize_strided, partials.data_ptr(), n, rows // n, c, W, P, 1,
                 _p(norm_mod.weight), _p(norm_mod.bias), float(norm_mod.eps), 0.0, None, None, None,
                 nb.scale.data_ptr(), nb.shift.data_ptr(), nb.mean.data_ptr(), nb.invstd.data_ptr(),
                 keep=(norm_mod.weight, norm_mod.bias, nb, partials))
        return
    rm = getattr(norm_mod, "running_mean", None)
    rv = getattr(norm_mod, "running_var", None)
    nbt = getattr(norm_mod, "num_batches_tracked", None)
    prog.add("norm_finalize", lib().mpgan_norm_finalize_strided, partials.data_ptr(), 1, rows, c, W, n * P, 0,
             _p(norm_mod.weight), _p(norm_mod.bias), float(norm_mod.eps), float(norm_mod.momentum), _p(rm), _p(rv),
             _p(nbt), nb.scale.data_ptr(), nb.shift.data_ptr(), nb.mean.data_ptr(), nb.invstd.data_ptr(),
             keep=(norm_mod.weight, norm_mod.bias, rm, rv, nbt, nb, partials), desc=f"C{c} rows{rows}")


def emit_eval_norms(prog, eval_norms, dev):
    """One launch computing scale/shift of every eval-mode BatchNorm layer of a plan (they depend on
    parameters and running statistics only, not on the input)."""
    if not eval_norms:
        return
    import struct
    rows = []
    for norm_mod, nb, c in eval_norms:
        eps_bits = struct.unpack("<I", struct.pack("<f", float(norm_mod.eps)))[0]
        rows.append([_p(norm_mod.weight) or 0, _p(norm_mod.bias) or 0, norm_mod.running_mean.data_ptr(),
                     norm_mod.running_var.data_ptr(), nb.scale.data_ptr(), nb.shift.data_ptr(), nb.mean.data_ptr(),
                     nb.invstd.data_ptr(), c, eps_bits])
    table = torch.tensor(rows, dtype=torch.int64, device=dev)
    prog.add("norm_from_running_multi", lib().mpgan_norm_from_running_multi, table.data_ptr(), len(rows),
             keep=(table, eval_norms))


def emit_conv_fwd_act(prog, rows, g: ConvGeom, x, wp, bias, y, norm_mod=None, act_mod=None, c_norm=None, resid=None,
                      tanh=False):
    """Eval-mode inference conv (code/GAN/inferrence.py:97-110,169-170): running-statistics BatchNorm, PReLU, the conv's
    bias and the residual add all in the conv's epilogue (mpgan_conv_forward_act); `rows` collects the plan's table rows
    for the one up-front mpgan_epi_vectors_multi launch that fills this conv's scale / shift / slope vectors.
    c_norm: the norm + activation cover only the first c_norm output channels (a conv fused with its unit's residual conv)."""
    import struct
    ops._check_in_out(g, x, y, "plan conv_forward_act")
    c = g.cout if c_norm is None else c_norm
    if norm_mod is None:
        c = 0
    else:
        assert hasattr(norm_mod, "running_mean") and norm_mod.running_mean is not None, "eval plan: BatchNorm layers only"
    if act_mod is not None:
        assert act_mod.weight.numel() == 1, "eval plan: PReLU with one parameter (MONAI 0.4.0's default)"
    cpad = (g.cout + 3) // 4 * 4
    vec = torch.empty(3, cpad, device=x.device)              # scale / shift / slope, each row 16-byte aligned
    eps_bits = struct.unpack("<I", struct.pack("<f", float(norm_mod.eps if norm_mod is not None else 0.0)))[0]
    nm = norm_mod
    rows.append([_p(nm.weight) or 0 if nm is not None else 0, _p(nm.bias) or 0 if nm is not None else 0,
                 nm.running_mean.data_ptr() if nm is not None else 0, nm.running_var.data_ptr() if nm is not None else 0,
                 _p(bias) or 0, _p(act_mod.weight) if act_mod is not None else 0,
                 vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(), c, g.cout, eps_bits])
    gc = g.c()
    tag = (gather_kernel_name(g, False, False), 2.0 * conv_macs(g), conv_bytes(g))
    prog.add("conv_forward", lib().mpgan_conv_forward_act, C.byref(gc), x.data_ptr(), _ld(x), wp.data_ptr(),
             vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(), _p(resid), _ld(resid), int(tanh), y.data_ptr(), _ld(y),
             keep=(gc, x, wp, bias, y, resid, vec, norm_mod, act_mod), desc=_gdesc(g), tag=tag)


def emit_epi_vectors(prog, rows, dev):
    """One launch filling the epilogue-activation vectors of every conv of an eval-mode plan (they depend on parameters
    and running statistics only, not on the input)."""
    if not rows:
        return
    table = torch.tensor(rows, dtype=torch.int64, device=dev)
    prog.add("epi_vectors_multi", lib().mpgan_epi_vectors_multi, table.data_ptr(), len(rows), keep=(table,))


def emit_norm_act_add(prog, z, pz, r, pr, out, tanh=False, fold=None):
    """fold: mpgan_norm_fold of z's producer (this launch finalises its statistics)."""
    n, P, ldz = ops._cl(z, "norm_act_add z")
    assert out.shape == z.shape and (r is None or r.shape == z.shape)
    pzc = pz.c() if pz is not None else None
    prc = pr.c() if pr is not None else None
    if fold is None:
        prog.add("norm_act_add", lib().mpgan_norm_act_add, z.data_ptr(), ldz, C.byref(pzc) if pzc else None, _p(r),
                 _ld(r), C.byref(prc) if prc else None, n, P, z.shape[-1], int(tanh), out.data_ptr(), _ld(out),
                 keep=(z, pzc, pz, r, prc, pr, out), desc=f"C{z.shape[-1]} n{n} P{P}")
        return
    prog.add("norm_act_add_fold", lib().mpgan_norm_act_add_fold, z.data_ptr(), ldz, C.byref(pzc), C.byref(fold), _p(r),
             _ld(r), C.byref(prc) if prc else None, n, P, z.shape[-1], int(tanh), out.data_ptr(), _ld(out),
             keep=(z, pzc, pz, fold, r, prc, pr, out))


def emit_norm_bwd(prog, g, z, nb: NormBuf, pro: Prologue, dz, partials, dgamma, dbeta, dslope, peer=None, reduced_rows=0):
    """reduce -> finalize -> apply.  dgamma/dbeta/dslope are gradient views
    (accumulated into) or None when the owner's parameters are frozen.
    reduced_rows > 0: `partials` already holds that many [3][C] rows, left by the backward-data launch that
    produced g (emit_conv_dgrad_stats) -- no reduce pass."""
    n, P, ldz = ops._cl(z, "norm_bwd z")
    assert g.shape == z.shape and dz.shape == z.shape
    c = z.shape[-1]
    chunks = ops.stats_chunks(P, c)
    assert partials.numel() >= n * chunks * (3 * c + 1)              # [3][C] rows + one slope scalar per row (ops.norm_bwd_reduce)
    L = lib()
    pc = pro.c()
    pe = peer.c() if peer is not None else None
    pe_ref = C.byref(pe) if pe is not None else None
    if reduced_rows:
        assert peer is None and not nb.instance and partials.numel() >= reduced_rows * 3 * c
        # rows left by mpgan_conv_backward_data_stats carry no per-row slope scalars (include/mpgan_hip.h): a learnable
        # PReLU slope needs the stand-alone reduce pass
        assert dslope is None, "fused norm-backward rows cannot feed a slope gradient"
        prog.add("norm_bwd_finalize", L.mpgan_norm_bwd_finalize, partials.data_ptr(), 1, reduced_rows, c, n * P, 0,
                 _p(dgamma), _p(dbeta), _p(dslope), nb.c1.data_ptr(), nb.c2.data_ptr(), keep=(dgamma, dbeta, dslope))
        prog.add("norm_bwd_apply", L.mpgan_norm_bwd_apply, g.data_ptr(), _ld(g), z.data_ptr(), ldz, C.byref(pc),
                 nb.mean.data_ptr(), nb.invstd.data_ptr(), nb.c1.data_ptr(), nb.c2.data_ptr(), pe_ref, n, P, c,
                 dz.data_ptr(), _ld(dz), keep=(dz, g, z, pc, pro, nb, partials), desc=f"C{c} n{n} P{P}")
        return
    prog.add("norm_bwd_reduce", L.mpgan_norm_bwd_reduce, g.data_ptr(), _ld(g), z.data_ptr(), ldz, C.byref(pc),
             nb.mean.data_ptr(), nb.invstd.data_ptr(), pe_ref, n, P, c, partials.data_ptr(),
             keep=(g, z, pc, pro, nb, partials, pe, peer), desc=f"C{c} n{n} P{P}")
    prog.add("norm_bwd_finalize", L.mpgan_norm_bwd_finalize, partials.data_ptr(), n, chunks, c, P, int(nb.instance),
             _p(dgamma), _p(dbeta), _p(dslope), nb.c1.data_ptr(), nb.c2.data_ptr(), keep=(dgamma, dbeta, dslope),
             desc=f"C{c} rows{n * chunks}")
    prog.add("norm_bwd_apply", L.mpgan_norm_bwd_apply, g.data_ptr(), _ld(g), z.data_ptr(), ldz, C.byref(pc),
             nb.mean.data_ptr(), nb.invstd.data_ptr(), nb.c1.data_ptr(), nb.c2.data_ptr(), pe_ref, n, P, c,
             dz.data_ptr(), _ld(dz), keep=(dz,), desc=f"C{c} n{n} P{P}")


def _t3(v, dims, fill):
    v = (v,) * dims if isinstance(v, int) else tuple(v)
    return (fill,) * (3 - dims) + tuple(v)


_GEOM_DEFAULTS = dict(mm_bf16=False)


class geom_defaults:
    """Plan-wide geometry flags while a plan is being built: `with geom_defaults(mm_bf16=True):` makes every
    conv_geom_of inside carry MPGAN_CONV_MM_BF16 (the C5 generator: bf16 matrix operands, fp32 storage)."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.saved = dict(_GEOM_DEFAULTS)
        _GEOM_DEFAULTS.update(self.kw)

    def __exit__(self, *a):
        _GEOM_DEFAULTS.clear()
        _GEOM_DEFAULTS.update(self.saved)
        return False


def conv_geom_of(mod, n, in_dhw, dims) -> ConvGeom:
    """Geometry of an nn.ConvNd / nn.ConvTransposeNd module at a given input size."""
    tr = isinstance(mod, (nn.ConvTranspose2d, nn.ConvTranspose3d))
    k, s, p = _t3(mod.kernel_size, dims, 1), _t3(mod.stride, dims, 1), _t3(mod.padding, dims, 0)
    op = _t3(mod.output_padding, dims, 0) if tr else (0, 0, 0)
    return ConvGeom(n, tuple(in_dhw), mod.in_channels, mod.out_channels, k, s, p, tr, op, **_GEOM_DEFAULTS)


class Scratch:
    """Grow-only scratch shared by the plans of one network (partials, wgrad slabs)."""

    def __init__(self, dev):
        self.dev = dev
        self.partials_need = 0
        self.ws_need = 0
        self.partials = None
        self.ws = None

    def want_partials(self, n, P, c):
        self.partials_need = max(self.partials_need, (n * ops.stats_chunks(P, c) + 32) * (3 * c + 1) + c)

    def want_ws(self, g: ConvGeom):
        self.ws_need = max(self.ws_need, ops.conv_wgrad_workspace(g) // 4)
        # fused-statistics partial rows of the forward conv (with or without prologue: same count)
        rows = max(ops.conv_stats_rows(g, code) for code in (0, 1, 2))
        self.partials_need = max(self.partials_need, (rows + 32) * 2 * g.cout)   # + finalize's fold scratch

    def alloc(self):
        self.partials = torch.empty(max(self.partials_need, 4), device=self.dev)
        self.ws = torch.empty(max(self.ws_need, 4), device=self.dev)


# --------------------------------------------------------------------------
# residual U-Net
# --------------------------------------------------------------------------
class _RU:
    """Handles into one ResidualUnit's parameter modules."""

    def __init__(self, ru):
        self.units = []
        for unit in ru.conv:
            adn = getattr(unit, "adn", None)
            self.units.append((unit.conv, adn.N if adn is not None else None, adn.A if adn is not None else None))
        self.res = ru.residual if not isinstance(ru.residual, nn.Identity) else None


class UNetPlan:
    """Forward/backward programs of one residual U-Net for a fixed input shape."""

    def __init__(self, unet, store: ParamStore, n: int, spatial: Sequence[int], x_in, y_out, *, tanh_out: bool,
                 instance: bool, want_backward: bool, gbufs: Optional[dict], scratch: Scratch, training: bool = True,
                 own_mark: Optional[Mark] = None, wait_mark: Optional[Mark] = None):
        self.unet, self.store, self.n = unet, store, n
        self.training = training
        dims = unet.dimensions
        self.dims = dims
        dev = x_in.device
        chans = list(unet.channels)
        L = len(chans)                      # levels incl. bottom
        strides = list(unet.strides)[:L - 1]
        dhw0 = _t3(spatial, dims, 1)
        for d in range(3 - dims, 3):
            if dhw0[d] % (2 ** (L - 1)) != 0:
                raise ValueError(f"U-Net input extent {dhw0[d]} is not divisible by {2 ** (L - 1)}")
        # walk the module tree: level l = (down RU, up Sequential); last = bottom RU
        downs, ups = [], []
        blk = unet.model
        while True:
            downs.append(_RU(blk[0]))
            ups.append(blk[2])
            sub = blk[1].submodule
            if isinstance(sub, nn.Sequential):
                blk = sub
            else:
                bottom = _RU(sub)
                break
        assert len(downs) == L - 1
        self.fwd, self.bwd = Program(), Program()
        f = self.fwd
        E = lambda *shape: torch.empty(*shape, device=dev)

        def reg(conv):
            tr = isinstance(conv, (nn.ConvTranspose2d, nn.ConvTranspose3d))
            taps = 1
            for k in conv.kernel_size:
                taps *= k
            return store.register_conv(conv, cout=conv.out_channels, cin=conv.in_channels, taps=taps, transposed=tr)

        def prelu_pro(nb, A):
            return nb.prologue(ACT_LEAKY, 1.0, A.weight)

        # ---- shapes ----
        sizes = [dhw0]
        in_ch = [1] + chans[:L - 2]          # input channels of down level l
        geoms_down = []
        for l in range(L - 1):
            g0 = conv_geom_of(downs[l].units[0][0], n, sizes[l], dims)
            geoms_down.append(g0)
            sizes.append(g0.out_dhw)
        sub_out = [chans[l] for l in range(L - 2)] + [chans[L - 1]]  # channels appended to cat_l
        self.saved = []
        cats = [E(n, *sizes[l + 1], chans[l] + sub_out[l]) for l in range(L - 1)]
        self.cats = cats
        recs = {}

        def R(conv):
            if id(conv) not in recs:
                recs[id(conv)] = reg(conv)
            return recs[id(conv)]

        # ================= forward =================
        down_state = []
        for l in range(L - 1):
            ru = downs[l]
            xin = x_in if l == 0 else cats[l - 1][..., :chans[l - 1]]
            c = chans[l]
            (cv0, N0, A0), (cv1, N1, A1) = ru.units
            g0 = geoms_down[l]
            g1 = conv_geom_of(cv1, n, sizes[l + 1], dims)
            gr = conv_geom_of(ru.res, n, sizes[l], dims)
            # unit0 and the strided residual conv read the same input with the same geometry: ONE conv over
            # their concatenated output channels writes [z0 | r] (one launch, the input read once)
            fuse = _FUSE_DOWN and same_geom(g0, gr)
            fr = None
            if fuse:
                zr = E(n, *sizes[l + 1], 2 * c)
                z0, r = zr[..., :c], zr[..., c:]
                gf = dataclasses.replace(g0, cout=2 * c)
                reg(cv0), reg(ru.res)
                fr = store.register_fused(cv0, ru.res)       # before any packed offset is taken (emit)
            else:
                zr, gf = None, None
                z0, r = E(n, *sizes[l + 1], c), E(n, *sizes[l + 1], c)
            z1 = E(n, *sizes[l + 1], c)
            nb0, nb1 = NormBuf(n, c, instance, dev), NormBuf(n, c, instance, dev)
            P = sizes[l + 1][0] * sizes[l + 1][1] * sizes[l + 1][2]
            scratch.want_partials(n, P, c)
            for g in (g0, g1, gr) + ((gf,) if fuse else ()):
                scratch.want_ws(g)
            down_state.append(dict(xin=xin, z0=z0, z1=z1, r=r, nb0=nb0, nb1=nb1, g0=g0, g1=g1, gr=gr, ru=ru, c=c,
                                   zr=zr, gf=gf, fr=fr))
        # bottom
        cb_in, cb = chans[L - 2], chans[L - 1]
        (bc0, BN0, BA0), (bc1, BN1, BA1) = bottom.units
        sb = sizes[L - 1]
        gb0 = conv_geom_of(bc0, n, sb, dims)
        gb1 = conv_geom_of(bc1, n, sb, dims)
        # MONAI's ResidualUnit has a residual conv only when the stride or the channel count changes: a bottom
        # layer with chans[-2] == chans[-1] (generator_test.py's (…, 512, 512)) adds its input unchanged.
        gbr = conv_geom_of(bottom.res, n, sb, dims) if bottom.res is not None else None
        zb0, zb1 = E(n, *sb, cb), E(n, *sb, cb)
        rb = E(n, *sb, cb) if gbr is not None else None
        nbb0, nbb1 = NormBuf(n, cb, instance, dev), NormBuf(n, cb, instance, dev)
        scratch.want_partials(n, sb[0] * sb[1] * sb[2], cb)
        for g in (gb0, gb1, gbr):
            if g is not None:
                scratch.want_ws(g)
        # up
        up_state = []
        out_ch = in_ch                        # output channels of up level l == input channels of down level l
        for l in range(L - 1):
            convT_blk, ru_mod = ups[l][0], _RU(ups[l][1])
            ct, NT, AT = convT_blk.conv, convT_blk.adn.N, convT_blk.adn.A
            co = out_ch[l]
            gt = conv_geom_of(ct, n, sizes[l + 1], dims)
            assert gt.out_dhw == tuple(sizes[l]), (gt.out_dhw, sizes[l])
            cu, NU, AU = ru_mod.units[0]
            gu = conv_geom_of(cu, n, sizes[l], dims)
            zt = E(n, *sizes[l], co)
            nbt = NormBuf(n, co, instance, dev)
            P = sizes[l][0] * sizes[l][1] * sizes[l][2]
            scratch.want_partials(n, P, co)
            scratch.want_ws(gt)
            scratch.want_ws(gu)
            st = dict(ct=ct, NT=NT, AT=AT, cu=cu, NU=NU, AU=AU, gt=gt, gu=gu, zt=zt, nbt=nbt, co=co)
            if NU is not None:
                st["zu"] = E(n, *sizes[l], co)
                st["nbu"] = NormBuf(n, co, instance, dev)
            else:
                st["ua"] = E(n, *sizes[l], co)
            up_state.append(st)
        self._late = (down_state, up_state)
        # Accumulator statistics (csrc/norm_fold.h) for the norm layers whose producing kernel can leave them and
        # whose first consumer can fold them: no finalize launch for those.  (nb, accumulator row width)
        self.acc_wants = []
        if training and not instance and _ACC_STATS:
            def want(nb, g_prod, code, consumer_ok):
                if consumer_ok and ops.conv_acc_supported(g_prod, code):
                    self.acc_wants.append((nb, g_prod.cout))
            for l in range(L - 1):
                sdn = down_state[l]
                want(sdn["nb0"], sdn["gf"] if sdn["gf"] is not None else sdn["g0"], 0, ops.conv_fold_supported(sdn["g1"]))
                want(sdn["nb1"], sdn["g1"], 1, True)                     # consumer: the residual-sum pass
            want(nbb0, gb0, 0, ops.conv_fold_supported(gb1))
            want(nbb1, gb1, 1, True)
            for l in range(L - 1):
                u = up_state[l]
                if "zu" in u:
                    want(u["nbt"], u["gt"], 0, ops.conv_fold_supported(u["gu"]))
                    want(u["nbu"], u["gu"], 1, True)
                else:
                    want(u["nbt"], u["gt"], 0, True)                     # top level: materialised by the residual-sum pass
        self.scratch = scratch
        self._marks = (own_mark, wait_mark)
        self.eval_norms = [] if not training else None   # eval mode: (norm module, NormBuf, C) of every layer
        self.epi_rows = []                                # eval mode, BatchNorm: table rows of emit_conv_fwd_act
        self.eval_fused = (not training) and (not instance) and (not want_backward)
        self._build_args = dict(x_in=x_in, y_out=y_out, tanh_out=tanh_out, want_backward=want_backward, gbufs=gbufs,
                                chans=chans, L=L, sizes=sizes, in_ch=in_ch, sub_out=sub_out, cats=cats, R=R,
                                bottom=dict(bc0=bc0, BN0=BN0, BA0=BA0, bc1=bc1, BN1=BN1, BA1=BA1, res=bottom.res,
                                            gb0=gb0, gb1=gb1, gbr=gbr, zb0=zb0, zb1=zb1, rb=rb, nbb0=nbb0, nbb1=nbb1,
                                            cb=cb, cb_in=cb_in),
                                prelu_pro=prelu_pro, instance=instance)

    def _emit_eval(self, a, down_state, up_state):
        """Eval-mode inference program of one U-Net (code/GAN/inferrence.py:97-110): every
        conv emits its ACTIVATED output -- running-statistics BatchNorm, PReLU, bias and the unit's residual add in its
        epilogue -- so no norm_act_add launch and no normalise-on-load prologue remain: 15 launches per U-Net instead of
        22, all of them convs.  Same buffers as the training plan (a raw-output buffer now holds the activation)."""
        f, store, rows = self.fwd, self.store, self.epi_rows
        R = a["R"]
        chans, L, cats = a["chans"], a["L"], a["cats"]
        y_out = a["y_out"]
        bt = a["bottom"]
        wp = store.wp
        for l in range(L - 1):
            s = down_state[l]
            (cv0, N0, A0), (cv1, N1, A1) = s["ru"].units
            if s["gf"] is not None:       # unit0 || residual in one conv: the first c channels are normed + activated
                fr = s["fr"]
                emit_conv_fwd_act(f, rows, s["gf"], s["xin"], store.wp_fused(fr), store.bias_fused(fr), s["zr"], N0, A0,
                                  c_norm=s["c"])
            else:
                emit_conv_fwd_act(f, rows, s["g0"], s["xin"], wp(R(cv0)), cv0.bias, s["z0"], N0, A0)
                emit_conv_fwd(f, s["gr"], s["xin"], wp(R(s["ru"].res)), s["ru"].res.bias, s["r"])
            emit_conv_fwd_act(f, rows, s["g1"], s["z0"], wp(R(cv1)), cv1.bias, cats[l][..., :s["c"]], N1, A1, resid=s["r"])
        d_last = cats[L - 2][..., :bt["cb_in"]]
        emit_conv_fwd_act(f, rows, bt["gb0"], d_last, wp(R(bt["bc0"])), bt["bc0"].bias, bt["zb0"], bt["BN0"], bt["BA0"])
        if bt["res"] is not None:
            emit_conv_fwd(f, bt["gbr"], d_last, wp(R(bt["res"])), bt["res"].bias, bt["rb"])
        emit_conv_fwd_act(f, rows, bt["gb1"], bt["zb0"], wp(R(bt["bc1"])), bt["bc1"].bias, cats[L - 2][..., bt["cb_in"]:],
                          bt["BN1"], bt["BA1"], resid=bt["rb"] if bt["res"] is not None else d_last)
        for l in range(L - 2, -1, -1):
            u = up_state[l]
            if "zu" in u:
                emit_conv_fwd_act(f, rows, u["gt"], cats[l], wp(R(u["ct"])), u["ct"].bias, u["zt"], u["NT"], u["AT"])
                dst = cats[l - 1][..., chans[l - 1]:] if l > 0 else y_out
                emit_conv_fwd_act(f, rows, u["gu"], u["zt"], wp(R(u["cu"])), u["cu"].bias, dst, u["NU"], u["AU"],
                                  resid=u["zt"], tanh=(a["tanh_out"] and l == 0))
            else:                          # top level: the transposed conv leaves act(bn(.)) itself, then the conv-only unit
                emit_conv_fwd_act(f, rows, u["gt"], cats[l], wp(R(u["ct"])), u["ct"].bias, u["ua"], u["NT"], u["AT"])
                emit_conv_fwd(f, u["gu"], u["ua"], wp(R(u["cu"])), u["cu"].bias, y_out, resid=u["ua"], tanh=a["tanh_out"])

    # programs are emitted after the shared scratch has been sized and allocated
    def emit(self):
        a = self._build_args
        down_state, up_state = self._late
        store, scratch = self.store, self.scratch
        part, ws = scratch.partials, scratch.ws
        f, b = self.fwd, self.bwd
        R, prelu_pro = a["R"], a["prelu_pro"]
        chans, L, sizes, cats = a["chans"], a["L"], a["sizes"], a["cats"]
        x_in, y_out = a["x_in"], a["y_out"]
        tr = self.training
        bt = a["bottom"]
        wp, wpb = store.wp, store.wp_bwd
        if self.eval_fused:
            self._emit_eval(a, down_state, up_state)
            return

        # ================= forward =================
        for l in range(L - 1):
            s = down_state[l]
            (cv0, N0, A0), (cv1, N1, A1) = s["ru"].units
            if s["gf"] is not None:
                fr = s["fr"]
                emit_conv_fwd_norm(f, s["gf"], s["xin"], store.wp_fused(fr), store.bias_fused(fr), s["zr"], s["nb0"], N0,
                                   part, training=tr, eval_norms=self.eval_norms, c_norm=s["c"])
            else:
                emit_conv_fwd_norm(f, s["g0"], s["xin"], wp(R(cv0)), cv0.bias, s["z0"], s["nb0"], N0, part, training=tr,
                                   eval_norms=self.eval_norms)
                # (the residual conv could run on the side stream; measured: the event hand-offs cost more
                #  than the ~20 us kernels they would overlap -- G forward 4.42 -> 4.56 ms)
                emit_conv_fwd(f, s["gr"], s["xin"], wp(R(s["ru"].res)), s["ru"].res.bias, s["r"])
            emit_conv_fwd_norm(f, s["g1"], s["z0"], wp(R(cv1)), cv1.bias, s["z1"], s["nb1"], N1, part,
                               pro=prelu_pro(s["nb0"], A0), training=tr, eval_norms=self.eval_norms,
                               fold=s["nb0"].fold())
            emit_norm_act_add(f, s["z1"], prelu_pro(s["nb1"], A1), s["r"], None, cats[l][..., :s["c"]],
                              fold=s["nb1"].fold())
        d_last = cats[L - 2][..., :bt["cb_in"]]
        emit_conv_fwd_norm(f, bt["gb0"], d_last, wp(R(bt["bc0"])), bt["bc0"].bias, bt["zb0"], bt["nbb0"], bt["BN0"],
                           part, training=tr, eval_norms=self.eval_norms)
        if bt["res"] is not None:
            emit_conv_fwd(f, bt["gbr"], d_last, wp(R(bt["res"])), bt["res"].bias, bt["rb"])
        emit_conv_fwd_norm(f, bt["gb1"], bt["zb0"], wp(R(bt["bc1"])), bt["bc1"].bias, bt["zb1"], bt["nbb1"],
                           bt["BN1"], part, pro=prelu_pro(bt["nbb0"], bt["BA0"]), training=tr, eval_norms=self.eval_norms,
                           fold=bt["nbb0"].fold())
        emit_norm_act_add(f, bt["zb1"], prelu_pro(bt["nbb1"], bt["BA1"]), bt["rb"] if bt["res"] is not None else d_last,
                          None, cats[L - 2][..., bt["cb_in"]:], fold=bt["nbb1"].fold())
        for l in range(L - 2, -1, -1):
            u = up_state[l]
            emit_conv_fwd_norm(f, u["gt"], cats[l], wp(R(u["ct"])), u["ct"].bias, u["zt"], u["nbt"], u["NT"], part,
                               training=tr, eval_norms=self.eval_norms)
            pt = prelu_pro(u["nbt"], u["AT"])
            if "zu" in u:
                emit_conv_fwd_norm(f, u["gu"], u["zt"], wp(R(u["cu"])), u["cu"].bias, u["zu"], u["nbu"], u["NU"], part,
                                   pro=pt, training=tr, eval_norms=self.eval_norms, fold=u["nbt"].fold())
                dst = cats[l - 1][..., chans[l - 1]:] if l > 0 else y_out
                emit_norm_act_add(f, u["zu"], prelu_pro(u["nbu"], u["AU"]), u["zt"], pt, dst,
                                  tanh=(a["tanh_out"] and l == 0), fold=u["nbu"].fold())
            else:
                # top level: conv-only residual unit on the materialised act(bn(zt))
                emit_norm_act_add(f, u["zt"], pt, None, None, u["ua"], fold=u["nbt"].fold())
                emit_conv_fwd(f, u["gu"], u["ua"], wp(R(u["cu"])), u["cu"].bias, y_out, resid=u["ua"],
                              tanh=a["tanh_out"])
        if not a["want_backward"]:
            return

        # ================= backward =================
        # Weight gradients run on the side stream (lane 1), all of them AFTER this U-Net's norm-backward /
        # backward-data chain has been enqueued: they only read (activation, dz) pairs that nothing
        # rewrites inside this U-Net's backward, and write parameter gradients / the split-K workspace
        # that only lane 1 touches.  One event hands the whole batch over; it then runs next to the
        # NEXT U-Net's chain, which works in the other set of gradient scratch (GeneratorPlan keeps two
        # and rotates).  `wait_mark` orders this U-Net after the side-stream work of the U-Net that
        # used the same set last.
        own_mark, wait_mark = self._marks
        bw = Program()
        if wait_mark is not None:
            b.wait(wait_mark)
        G = a["gbufs"]                       # shared gradient scratch (see GeneratorPlan)
        gv = store.grad_view
        g_out, g_x = G["g_out"], G["g_x"]
        # ---- up path, top to bottom of the U ----
        for l in range(0, L - 1):
            u = up_state[l]
            pt = prelu_pro(u["nbt"], u["AT"])
            g_u = g_out if l == 0 else G["gcat"][l - 1][..., chans[l - 1]:]
            if "zu" in u:
                dzu, gta = G["dzu"][l], G["gta"][l]
                emit_norm_bwd(b, g_u, u["zu"], u["nbu"], prelu_pro(u["nbu"], u["AU"]), dzu, part,
                              gv(u["NU"].weight), gv(u["NU"].bias), gv(u["AU"].weight))
                emit_conv_wgrad(bw, u["gu"], u["zt"], dzu, gv(u["cu"].weight), ws, pro=pt, dbias=gv(u["cu"].bias), lane=1)
                emit_conv_dgrad(b, u["gu"], dzu, wpb(R(u["cu"])), gta, resid=g_u)
            else:
                gta = G["gta"][l]
                emit_conv_wgrad(bw, u["gu"], u["ua"], g_u, gv(u["cu"].weight), ws, dbias=gv(u["cu"].bias), lane=1)
                emit_conv_dgrad(b, u["gu"], g_u, wpb(R(u["cu"])), gta, resid=g_u)
            emit_norm_bwd(b, gta, u["zt"], u["nbt"], pt, gta, part, gv(u["NT"].weight), gv(u["NT"].bias),
                          gv(u["AT"].weight))
            emit_bias_grad(b, gta, gv(u["ct"].bias), part)
            emit_conv_wgrad(bw, u["gt"], cats[l], gta, gv(u["ct"].weight), ws, lane=1)
            emit_conv_dgrad(b, u["gt"], gta, wpb(R(u["ct"])), G["gcat"][l])
        # ---- bottom ----
        gcat_last = G["gcat"][L - 2]
        g_b = gcat_last[..., bt["cb_in"]:]
        gd_last = gcat_last[..., :bt["cb_in"]]
        d_last = cats[L - 2][..., :bt["cb_in"]]
        dzb1, gab0 = G["dzb1"], G["gab0"]
        emit_norm_bwd(b, g_b, bt["zb1"], bt["nbb1"], prelu_pro(bt["nbb1"], bt["BA1"]), dzb1, part,
                      gv(bt["BN1"].weight), gv(bt["BN1"].bias), gv(bt["BA1"].weight))
        emit_conv_wgrad(bw, bt["gb1"], bt["zb0"], dzb1, gv(bt["bc1"].weight), ws, pro=prelu_pro(bt["nbb0"], bt["BA0"]),
                        dbias=gv(bt["bc1"].bias), lane=1)
        emit_conv_dgrad(b, bt["gb1"], dzb1, wpb(R(bt["bc1"])), gab0)
        emit_norm_bwd(b, gab0, bt["zb0"], bt["nbb0"], prelu_pro(bt["nbb0"], bt["BA0"]), gab0, part,
                      gv(bt["BN0"].weight), gv(bt["BN0"].bias), gv(bt["BA0"].weight))
        emit_conv_wgrad(bw, bt["gb0"], d_last, gab0, gv(bt["bc0"].weight), ws, dbias=gv(bt["bc0"].bias), lane=1)
        emit_conv_dgrad(b, bt["gb0"], gab0, wpb(R(bt["bc0"])), gd_last, resid=gd_last)
        if bt["res"] is not None:
            emit_conv_wgrad(bw, bt["gbr"], d_last, g_b, gv(bt["res"].weight), ws, dbias=gv(bt["res"].bias), lane=1)
            emit_conv_dgrad(b, bt["gbr"], g_b, wpb(R(bt["res"])), gd_last, resid=gd_last)
        else:
            emit_norm_act_add(b, g_b, None, gd_last, None, gd_last)           # identity residual: gd_last += g_b
        # ---- down path, bottom to top ----
        for l in range(L - 2, -1, -1):
            s = down_state[l]
            (cv0, N0, A0), (cv1, N1, A1) = s["ru"].units
            g_d = G["gcat"][l][..., :s["c"]]
            dz1, ga0 = G["dz1"][l], G["ga0"][l]
            emit_norm_bwd(b, g_d, s["z1"], s["nb1"], prelu_pro(s["nb1"], A1), dz1, part, gv(N1.weight), gv(N1.bias),
                          gv(A1.weight))
            emit_conv_wgrad(bw, s["g1"], s["z0"], dz1, gv(cv1.weight), ws, pro=prelu_pro(s["nb0"], A0),
                            dbias=gv(cv1.bias), lane=1)
            emit_conv_dgrad(b, s["g1"], dz1, wpb(R(cv1)), ga0)
            emit_norm_bwd(b, ga0, s["z0"], s["nb0"], prelu_pro(s["nb0"], A0), ga0, part, gv(N0.weight), gv(N0.bias),
                          gv(A0.weight))
            emit_conv_wgrad(bw, s["g0"], s["xin"], ga0, gv(cv0.weight), ws, dbias=gv(cv0.bias), lane=1)
            emit_conv_wgrad(bw, s["gr"], s["xin"], g_d, gv(s["ru"].res.weight), ws, dbias=gv(s["ru"].res.bias), lane=1)
            if l > 0:
                tgt = G["gcat"][l - 1][..., :chans[l - 1]]
                emit_conv_dgrad(b, s["g0"], ga0, wpb(R(cv0)), tgt, resid=tgt)
                emit_conv_dgrad(b, s["gr"], g_d, wpb(R(s["ru"].res)), tgt, resid=tgt)
            elif g_x is not None:
                emit_conv_dgrad(b, s["g0"], ga0, wpb(R(cv0)), g_x)
                emit_conv_dgrad(b, s["gr"], g_d, wpb(R(s["ru"].res)), g_x, resid=g_x)
        b.extend(bw)
        if own_mark is not None:
            b.mark(own_mark)
        else:
            b.join()


class GeneratorPlan:
    """CasNet: chain of U-Nets + Tanh (code/GAN/GAN_final.py:92-122)."""

    def __init__(self, gen, store: ParamStore, n: int, spatial: Sequence[int], *, want_backward: bool,
                 want_input_grad: bool, instance: bool, training: bool = True, mm_bf16: bool = False):
        # mm_bf16: every MFMA-served conv of this plan rounds its matrix operands to bf16 (MPGAN_CONV_MM_BF16;
        # config C5's generator): activations, weights, statistics and gradients in HBM stay fp32
        self.mm_bf16 = mm_bf16
        unets = [m for m in gen.model if not isinstance(m, nn.Tanh)]
        dims = unets[0].dimensions
        dev = store.flat.device
        dhw = _t3(spatial, dims, 1)
        self.n, self.dhw, self.dims = n, dhw, dims
        self.store = store
        E = lambda *shape: torch.empty(*shape, device=dev)
        self.acts = [E(n, *dhw, 1) for _ in range(len(unets) + 1)]   # x0 .. y
        self.x_in, self.y = self.acts[0], self.acts[-1]
        self.scratch = Scratch(dev)
        self.want_backward = want_backward
        chans = list(unets[0].channels)
        L = len(chans)
        gb = None
        if want_backward:
            # gradient scratch shared by all U-Nets (their backwards run one after another)
            sizes = [dhw]
            for l in range(L - 1):
                sizes.append(tuple((s + 1) // 2 if i >= 3 - dims else s for i, s in enumerate(sizes[-1])))
            in_ch = [1] + chans[:L - 2]
            sub_out = [chans[l] for l in range(L - 2)] + [chans[L - 1]]
            def grad_scratch():
                return dict(
                    gcat=[E(n, *sizes[l + 1], chans[l] + sub_out[l]) for l in range(L - 1)],
                    dzu=[E(n, *sizes[l], in_ch[l]) for l in range(L - 1)],
                    gta=[E(n, *sizes[l], in_ch[l]) for l in range(L - 1)],
                    dz1=[E(n, *sizes[l + 1], chans[l]) for l in range(L - 1)],
                    ga0=[E(n, *sizes[l + 1], chans[l]) for l in range(L - 1)],
                    dzb1=E(n, *sizes[L - 1], chans[L - 1]), gab0=E(n, *sizes[L - 1], chans[L - 1]))
            # two sets, rotated: U-Net u's weight gradients (side stream) still read set u % 2 while
            # U-Net u-1's chain fills the other one
            gb = [grad_scratch(), grad_scratch()]
            self.g_acts = [E(n, *dhw, 1) for _ in range(len(unets) + 1)]   # dL/d(acts[u]); never shared
            self.g_y = E(n, *dhw, 1)          # upstream gradient dL/dy is copied here
        self.unet_plans: List[UNetPlan] = []
        nU = len(unets)
        marks = [Mark() for _ in range(nU)]
        for u, unet in enumerate(unets):
            g = None
            if want_backward:
                g = dict(gb[u % 2])
                g["g_out"] = self.g_acts[u + 1]
                g["g_x"] = self.g_acts[u] if (u > 0 or want_input_grad) else None
            with geom_defaults(mm_bf16=mm_bf16):
                self.unet_plans.append(UNetPlan(unet, store, n, spatial, self.acts[u], self.acts[u + 1],
                                                tanh_out=(u == nU - 1), instance=instance, want_backward=want_backward,
                                                gbufs=g, scratch=self.scratch, training=training,
                                                own_mark=marks[u], wait_mark=marks[u + 2] if u + 2 < nU else None))
        self.scratch.alloc()
        wants = [w for p in self.unet_plans for w in p.acc_wants]
        self.acc_all = None
        if wants:
            total = sum(ops.ACC_REPLICAS * ops.ACC_WORDS * cw for _, cw in wants)
            self.acc_all = torch.zeros(total, dtype=torch.int64, device=dev)
            off = 0
            for nb, cw in wants:
                sz = ops.ACC_REPLICAS * ops.ACC_WORDS * cw
                nb.acc = self.acc_all[off:off + sz]
                off += sz
        for p in self.unet_plans:
            p.emit()
        self.fwd = Program()
        store.emit_pack(self.fwd)
        if self.acc_all is not None:      # one memset for every norm layer of the forward
            self.fwd.add("zero_bytes", lib().mpgan_zero_bytes, self.acc_all.data_ptr(), self.acc_all.numel() * 8,
                         keep=(self.acc_all,))
        if not training:
            emit_eval_norms(self.fwd, [e for p in self.unet_plans for e in p.eval_norms], dev)
            emit_epi_vectors(self.fwd, [r for p in self.unet_plans for r in p.epi_rows], dev)
        for p in self.unet_plans:
            self.fwd.extend(p.fwd)
        self.bwd = Program()
        if want_backward:
            L_ = lib()
            last = self.g_acts[nU]            # = g_out of the last U-Net
            self.bwd.add("tanh_backward", L_.mpgan_tanh_backward, self.g_y.data_ptr(), self.y.data_ptr(),
                         self.y.numel(), last.data_ptr(), keep=(self.g_y, self.y, last))
            for p in reversed(self.unet_plans):
                self.bwd.extend(p.bwd)
            self.bwd.join()                   # parameter gradients are complete when the program returns
            self.g_x = self.g_acts[0] if want_input_grad else None
        self.busy = False


class IoSlots:
    """The plan's boundary tensors (network input, output, upstream gradient, input gradient) as rebindable pointer
    slots: `bind` points the programs at the caller's tensors for one pass, `reset` returns them to the plan's own
    buffers (which tools and tests that drive `plan.fwd.run()` directly keep using)."""

    def __init__(self, programs, **tensors):
        self.home, self.slot = {}, {}
        for name, t in tensors.items():
            if t is None:
                continue
            sl = C.c_void_p(t.data_ptr())
            if sum(pr.rebind(t.data_ptr(), sl) for pr in programs) == 0:
                continue
            self.home[name], self.slot[name] = t, sl

    @staticmethod
    def usable(t: torch.Tensor) -> bool:
        return t.is_contiguous() and t.dtype == torch.float32 and t.data_ptr() % 256 == 0

    def bind(self, name, t: torch.Tensor) -> bool:
        sl = self.slot.get(name)
        if sl is None or not self.usable(t) or t.numel() != self.home[name].numel():
            return False
        sl.value = t.data_ptr()
        return True

    def reset(self):
        for name, sl in self.slot.items():
            sl.value = self.home[name].data_ptr()


def plan_io(plan) -> IoSlots:
    io = getattr(plan, "io", None)
    if io is None:
        io = plan.io = IoSlots((plan.fwd, plan.bwd), x=plan.x_in, y=getattr(plan, "y", None),
                               g_y=getattr(plan, "g_y", None), g_x=getattr(plan, "g_x", None))
    return io


# --------------------------------------------------------------------------
# discriminator (variant A)
# --------------------------------------------------------------------------
class DiscPlan:
    """4 x (valid conv -> BN -> LeakyReLU 0.2) -> Flatten -> Linear(F,1) -> Sigmoid
    (code/GAN/GAN_final.py:159-209)."""

    def __init__(self, disc, store: ParamStore, n: int, spatial: Sequence[int], *, want_backward: bool,
                 want_input_grad: bool, want_param_grads: bool):
        dims = disc.dimensions
        dev = store.flat.device
        dhw = _t3(spatial, dims, 1)
        self.n, self.dhw, self.dims, self.store = n, dhw, dims, store
        E = lambda *shape: torch.empty(*shape, device=dev)
        convs = [disc.model_conv[i] for i in (0, 3, 6, 9)]
        bns = [disc.model_conv[i] for i in (1, 4, 7, 10)]
        lin = disc.model_linear[1]
        self.x_in = E(n, *dhw, 1)
        geoms, zs, nbs, recs = [], [], [], []
        size = dhw
        scratch = Scratch(dev)
        for cv in convs:
            g = conv_geom_of(cv, n, size, dims)
            geoms.append(g)
            size = g.out_dhw
            if min(size) < 1:
                raise ValueError(f"discriminator input {spatial} too small")
            zs.append(E(n, *size, cv.out_channels))
            nbs.append(NormBuf(n, cv.out_channels, False, dev))
            taps = 1
            for k in cv.kernel_size:
                taps *= k
            recs.append(store.register_conv(cv, cout=cv.out_channels, cin=cv.in_channels, taps=taps))
            scratch.want_partials(n, size[0] * size[1] * size[2], cv.out_channels)
            scratch.want_ws(g)
            if g.cin > 1:          # rows of the backward-data launch that also reduces the previous layer's norm backward
                scratch.partials_need = max(scratch.partials_need, ops.conv_bwd_stats_rows(g) * 3 * g.cin)
        P_last = size[0] * size[1] * size[2]
        c_last = convs[-1].out_channels
        if lin.in_features != P_last * c_last:
            raise ValueError(f"Linear.in_features {lin.in_features} != {c_last}*{P_last} for input {spatial}")
        rlin = store.register_conv(lin, cout=1, cin=c_last, taps=P_last)
        scratch.alloc()
        part, ws = scratch.partials, scratch.ws
        self.logit, self.prob = E(n), E(n)
        lin_part = E(ops.linear1_partials(n))
        L = lib()
        f = self.fwd = Program()
        store.emit_pack(f)
        lrelu = lambda nb: nb.prologue(ACT_LEAKY, 0.2, None)
        src, pro = self.x_in, None
        for i, cv in enumerate(convs):
            emit_conv_fwd_norm(f, geoms[i], src, store.wp(recs[i]), cv.bias, zs[i], nbs[i], bns[i], part, pro=pro)
            src, pro = zs[i], lrelu(nbs[i])
        pc = pro.c()
        f.add("linear1_forward", L.mpgan_linear1_forward, zs[-1].data_ptr(), C.byref(pc), n, P_last, c_last,
              store.wp(rlin).data_ptr(), lin.bias.data_ptr(), lin_part.data_ptr(), self.logit.data_ptr(),
              self.prob.data_ptr(), keep=(pc, pro, lin_part))
        self.busy = False
        self.bwd = Program()
        self.g_x = None
        if not want_backward:
            return
        b = self.bwd
        gv = store.grad_view if want_param_grads else (lambda p: None)
        self.g_prob = E(n)
        dlogit = E(n)
        gas = [E(*z.shape) for z in zs]
        self.zs, self.gas, self.nbs = zs, gas, nbs   # raw conv outputs, their gradients, norm vectors (tests, tools)
        b.add("sigmoid_backward", L.mpgan_sigmoid_backward, self.g_prob.data_ptr(), self.prob.data_ptr(), n,
              dlogit.data_ptr(), keep=(dlogit,))
        b.add("linear1_backward", L.mpgan_linear1_backward, zs[-1].data_ptr(), C.byref(pc), n, P_last, c_last,
              store.wp(rlin).data_ptr(), dlogit.data_ptr(), gas[-1].data_ptr(), _p(gv(lin.weight)), _p(gv(lin.bias)),
              1.0, keep=(gas,))
        reduced = 0                                  # rows the previous backward-data launch left in `part`
        for i in range(3, -1, -1):
            pro_i = lrelu(nbs[i])
            emit_norm_bwd(b, gas[i], zs[i], nbs[i], pro_i, gas[i], part, gv(bns[i].weight), gv(bns[i].bias), None,
                          reduced_rows=reduced)
            src = zs[i - 1] if i > 0 else self.x_in
            pro_in = lrelu(nbs[i - 1]) if i > 0 else None
            if want_param_grads:
                emit_conv_wgrad(b, geoms[i], src, gas[i], gv(convs[i].weight), ws, pro=pro_in,
                                dbias=gv(convs[i].bias), lane=_D_WGRAD_LANE)
            if i > 0:
                # the gradient w.r.t. a_{i-1} = LeakyReLU(BN(z_{i-1})): the epilogue of this launch also forms the
                # norm-backward sums of layer i-1, so that layer needs no reduce pass over g and z
                reduced = emit_conv_dgrad_stats(b, geoms[i], gas[i], store.wp_bwd(recs[i]), gas[i - 1], zs[i - 1],
                                                nbs[i - 1], 0.2, part)
            elif want_input_grad:
                self.g_x = E(n, *dhw, 1)
                emit_conv_dgrad(b, geoms[0], gas[0], store.wp_bwd(recs[0]), self.g_x)
        b.join()


# --------------------------------------------------------------------------
# discriminator (variant A), bf16 storage (BASELINE config C5)
# --------------------------------------------------------------------------
def _bf16_kernel_name(g: ConvGeom, backward_data: bool) -> str:
    """rocprofv3's name of the bf16 kernel that serves this layer (labels of bench.py's probe)."""
    gc = g.c()
    bn = 128 if (g.cin if backward_data else g.cout) > 64 else 64
    v = lib().mpgan_conv_variant_bf16(C.byref(gc), int(backward_data))
    if v == 1:
        return f"gather_patch_bf16_kernel<{bn}>"
    if v == 5:
        return f"gather_patch8_bf16_kernel<{bn}>"
    if v in (2, 3, 4):
        masked = "true" if backward_data or any(g.pad) else "false"
        return f"gather_conv_bf16_wide_kernel<{'4, 2' if v == 3 else '2, 4'}, {masked}, true, {'true' if v == 4 else 'false'}>"
    return f"gather_conv_bf16_kernel<{bn}, {'true' if backward_data else 'false'}, 8>"


def _bf16_wgrad_kernel_name(g: ConvGeom) -> str:
    """rocprofv3's name of the bf16 weight-gradient kernel of this layer, asked of the library that makes the choice."""
    gc = g.c()
    return "wgrad_bf16_wide_kernel" if lib().mpgan_conv_wgrad_variant_bf16(C.byref(gc)) == 1 else "wgrad_bf16_kernel<8>"


class DiscPlanBF16:
    """The same network as DiscPlan with bf16 activations, activation gradients and packed weights in HBM
    (code/GAN/GAN_final.py:159-209 at the reference's 3-D shape); fp32 accumulation, statistics, parameters,
    weight gradients and Adam.  Per layer: raw conv output z (bf16) + fused fp32 statistics -> finalize ->
    a = LeakyReLU(BN(z)) materialised once (bf16; fp32 for the last layer, which the fp32 Linear head reads).
    The first layer (1 input channel) runs on the HBM-bound VALU kernels with fp32 weights."""

    def __init__(self, disc, store: ParamStore, n: int, spatial: Sequence[int], *, want_backward: bool,
                 want_input_grad: bool, want_param_grads: bool):
        dims = disc.dimensions
        dev = store.flat.device
        dhw = _t3(spatial, dims, 1)
        self.n, self.dhw, self.dims, self.store = n, dhw, dims, store
        bf = torch.bfloat16
        E = lambda *shape: torch.empty(*shape, device=dev)
        H = lambda *shape: torch.empty(*shape, device=dev, dtype=bf)
        convs = [disc.model_conv[i] for i in (0, 3, 6, 9)]
        bns = [disc.model_conv[i] for i in (1, 4, 7, 10)]
        lin = disc.model_linear[1]
        self.x_in = E(n, *dhw, 1)
        geoms, zs, acts, nbs, recs = [], [], [], [], []
        size = dhw
        L = lib()
        part_need, ws_need = 4, 4
        fused_rows, prev_bwd_rows = [], 0
        for i, cv in enumerate(convs):
            g = conv_geom_of(cv, n, size, dims)
            geoms.append(g)
            size = g.out_dhw
            if min(size) < 1:
                raise ValueError(f"discriminator input {spatial} too small")
            zs.append(H(n, *size, cv.out_channels))
            acts.append(E(n, *size, cv.out_channels) if i == 3 else H(n, *size, cv.out_channels))
            nbs.append(NormBuf(n, cv.out_channels, False, dev))
            taps = 1
            for k in cv.kernel_size:
                taps *= k
            recs.append(store.register_conv(cv, cout=cv.out_channels, cin=cv.in_channels, taps=taps))
            rows_total = n * size[0] * size[1] * size[2]
            fwd_rows = (rows_total + 255) // 256 if i == 0 else ops.conv_stats_rows_bf16(g)
            bwd_rows = ops.norm_bwd_rows_bf16(rows_total, cv.out_channels)
            # (rows of the fused norm-backward sums the NEXT layer's backward-data launch leaves for this layer's norm)
            fused_rows.append(0)
            if i > 0 and _FUSE_BWD_STATS_BF16:
                fused_rows[i - 1] = max(0, int(lib().mpgan_conv_bwd_stats_rows_bf16(C.byref(g.c()))))
                part_need = max(part_need, (max(fused_rows[i - 1], prev_bwd_rows) * 4 + 1) * cv.in_channels)
            prev_bwd_rows = bwd_rows
            part_need = max(part_need, (fwd_rows + 32) * 2 * cv.out_channels, bwd_rows * 4 * cv.out_channels + cv.out_channels)
            ws_need = max(ws_need, (ops.conv_wgrad_workspace_bf16dy(g) if i == 0 else ops.conv_wgrad_workspace_bf16(g)) // 4)
        P_last = size[0] * size[1] * size[2]
        c_last = convs[-1].out_channels
        if lin.in_features != P_last * c_last:
            raise ValueError(f"Linear.in_features {lin.in_features} != {c_last}*{P_last} for input {spatial}")
        rlin = store.register_conv(lin, cout=1, cin=c_last, taps=P_last)
        part = E(part_need)
        ws = E(ws_need)
        # bf16 packs of the three dense layers: [Cout][tap][Cin] (forward) and [Cin][tap][Cout] (backward-data)
        rows16, off = [], 0
        self._w16 = {}
        for i in (1, 2, 3):
            r = recs[i]
            nel = r.cout * r.cin * r.taps
            self._w16[i] = (off, off + (nel + 7) // 8 * 8, nel)
            rows16.append([r.w_off, off, r.cout, r.cin, r.taps, 0, 0, 0])
            rows16.append([r.w_off, off + (nel + 7) // 8 * 8, r.cout, r.cin, r.taps, 0, 1, 0])
            off += 2 * ((nel + 7) // 8 * 8)
        packed16 = torch.empty(off, device=dev, dtype=bf)
        table16 = torch.tensor(rows16, dtype=torch.int64, device=dev)
        w16 = lambda i: packed16[self._w16[i][0]:self._w16[i][0] + self._w16[i][2]]
        w16b = lambda i: packed16[self._w16[i][1]:self._w16[i][1] + self._w16[i][2]]
        self.logit, self.prob = E(n), E(n)
        lin_part = E(ops.linear1_partials(n))
        f = self.fwd = Program()
        store.emit_pack(f)                                   # fp32 packs: first layer, Linear head
        f.add("pack_weights_bf16", L.mpgan_pack_weights_bf16, store.flat.data_ptr(), packed16.data_ptr(),
              table16.data_ptr(), table16.shape[0], max(r.cout * r.cin * r.taps for r in recs[1:]),
              keep=(packed16, table16))
        src = self.x_in
        for i, cv in enumerate(convs):
            g, z, nb, bn = geoms[i], zs[i], nbs[i], bns[i]
            gc = g.c()
            rows_total = n * g.out_dhw[0] * g.out_dhw[1] * g.out_dhw[2]
            if i == 0:
                rows = (rows_total + 255) // 256
                f.add("conv_forward_f32_to_bf16", L.mpgan_conv_forward_f32_to_bf16, C.byref(gc), src.data_ptr(), 1,
                      store.wp(recs[0]).data_ptr(), cv.bias.data_ptr(), part.data_ptr(), z.data_ptr(), g.cout,
                      keep=(gc, src, z, part), desc=_gdesc(g), tag=("thin_cin1_full_kernel<16, true>", 2.0 * conv_macs(g),
                           conv_bytes(g, 2) + 2 * src.numel()))
            else:
                rows = ops.conv_stats_rows_bf16(g)
                f.add("conv_forward_bf16", L.mpgan_conv_forward_bf16, C.byref(gc), src.data_ptr(), g.cin,
                      w16(i).data_ptr(), cv.bias.data_ptr(), part.data_ptr(), z.data_ptr(), g.cout,
                      keep=(gc, src, z, part), desc=_gdesc(g),
                      tag=(_bf16_kernel_name(g, False), 2.0 * conv_macs(g), conv_bytes(g, 2)))
            f.add("norm_finalize", L.mpgan_norm_finalize, part.data_ptr(), 1, rows, g.cout, rows_total, 0,
                  _p(bn.weight), _p(bn.bias), float(bn.eps), float(bn.momentum), _p(bn.running_mean),
                  _p(bn.running_var), _p(bn.num_batches_tracked), nb.scale.data_ptr(), nb.shift.data_ptr(),
                  nb.mean.data_ptr(), nb.invstd.data_ptr(), keep=(bn, nb))
            a = acts[i]
            f.add("norm_act_bf16", L.mpgan_norm_act_bf16, z.data_ptr(), g.cout, nb.scale.data_ptr(), nb.shift.data_ptr(),
                  0.2, rows_total, g.cout, a.data_ptr(), g.cout, int(a.dtype == torch.float32), keep=(a,))
            src = a
        f.add("linear1_forward", L.mpgan_linear1_forward, acts[3].data_ptr(), None, n, P_last, c_last,
              store.wp(rlin).data_ptr(), lin.bias.data_ptr(), lin_part.data_ptr(), self.logit.data_ptr(),
              self.prob.data_ptr(), keep=(lin_part,))
        self.zs, self.acts, self.nbs = zs, acts, nbs
        self.busy = False
        self.bwd = Program()
        self.g_x = None
        if not want_backward:
            return
        b = self.bwd
        gv = store.grad_view if want_param_grads else (lambda p: None)
        self.g_prob = E(n)
        dlogit = E(n)
        # gradients w.r.t. the activations; the BatchNorm backward overwrites them in place with dz
        # (disc.debug_keep_intermediates: separate dz buffers, so that tests can check every layer of the backward
        #  against the CPU restatement on the SAME inputs)
        gas = [H(*z.shape) for z in zs[:3]] + [E(*zs[3].shape)]
        dz4 = H(*zs[3].shape)
        keep_all = bool(getattr(disc, "debug_keep_intermediates", False))
        self.dzs = [H(*z.shape) for z in zs[:3]] + [dz4] if keep_all else [gas[0], gas[1], gas[2], dz4]
        self.gas = gas
        b.add("sigmoid_backward", L.mpgan_sigmoid_backward, self.g_prob.data_ptr(), self.prob.data_ptr(), n,
              dlogit.data_ptr(), keep=(dlogit,))
        b.add("linear1_backward", L.mpgan_linear1_backward, acts[3].data_ptr(), None, n, P_last, c_last,
              store.wp(rlin).data_ptr(), dlogit.data_ptr(), gas[3].data_ptr(), _p(gv(lin.weight)), _p(gv(lin.bias)),
              1.0, keep=(gas, dz4))
        for i in range(3, -1, -1):
            g, z, nb, bn, cv = geoms[i], zs[i], nbs[i], bns[i], convs[i]
            gc = g.c()
            c = g.cout
            rows_total = n * g.out_dhw[0] * g.out_dhw[1] * g.out_dhw[2]
            brow = ops.norm_bwd_rows_bf16(rows_total, c)
            gin = gas[i]
            dz = self.dzs[i]
            g32 = int(gin.dtype == torch.float32)
            frow = fused_rows[i] if i < 3 else 0          # > 0: the backward-data launch that produced `gin` left the sums
            base = max(brow, frow) * 3 * c + c            # the apply pass's bias partials sit behind the rows finalize reads
            bias_part = part[base:base + brow * c] if (want_param_grads and i > 0) else None
            if not frow:
                b.add("norm_bwd_reduce_bf16", L.mpgan_norm_bwd_reduce_bf16, gin.data_ptr(), g32, c, z.data_ptr(), c,
                      nb.scale.data_ptr(), nb.shift.data_ptr(), nb.mean.data_ptr(), nb.invstd.data_ptr(), 0.2, rows_total,
                      c, part.data_ptr(), keep=(gin, z, nb, part))
            b.add("norm_bwd_finalize", L.mpgan_norm_bwd_finalize, part.data_ptr(), 1, frow or brow, c, rows_total, 0,
                  _p(gv(bn.weight)), _p(gv(bn.bias)), None, nb.c1.data_ptr(), nb.c2.data_ptr(), keep=(bn,))
            b.add("norm_bwd_apply_bf16", L.mpgan_norm_bwd_apply_bf16, gin.data_ptr(), g32, c, z.data_ptr(), c,
                  nb.scale.data_ptr(), nb.shift.data_ptr(), nb.mean.data_ptr(), nb.invstd.data_ptr(),
                  nb.c1.data_ptr(), nb.c2.data_ptr(), 0.2, rows_total, c, dz.data_ptr(), c, _p(bias_part), keep=(dz,))
            if want_param_grads:
                if i > 0:
                    b.add("reduce_partials", L.mpgan_reduce_partials, bias_part.data_ptr(), brow, c, c,
                          gv(cv.bias).data_ptr(), 1.0, keep=(bias_part,))
                    b.add("conv_backward_weight_bf16", L.mpgan_conv_backward_weight_bf16, C.byref(gc),
                          acts[i - 1].data_ptr(), g.cin, dz.data_ptr(), c, gv(cv.weight).data_ptr(), 1.0,
                          ws.data_ptr(), ws.numel() * 4, keep=(gc, ws), desc=_gdesc(g),
                          tag=(_bf16_wgrad_kernel_name(g), 2.0 * conv_macs(g), conv_bytes(g, 2)))
                else:
                    b.add("conv_backward_weight_bf16dy", L.mpgan_conv_backward_weight_bf16dy, C.byref(gc),
                          self.x_in.data_ptr(), 1, dz.data_ptr(), c, gv(cv.weight).data_ptr(), gv(cv.bias).data_ptr(),
                          1.0, ws.data_ptr(), ws.numel() * 4, keep=(gc, ws), desc=_gdesc(g),
                          tag=("wgrad_thin_kernel", 2.0 * conv_macs(g), conv_bytes(g, 2) + 2 * self.x_in.numel()))
            if i > 0 and fused_rows[i - 1]:
                # ... and the reduce pass of the layer in front (its norm-backward sums against z_{i-1}) in the same launch
                zp, nbp = zs[i - 1], nbs[i - 1]
                b.add("conv_backward_data_bf16", L.mpgan_conv_backward_data_stats_bf16, C.byref(gc), dz.data_ptr(), c,
                      w16b(i).data_ptr(), gas[i - 1].data_ptr(), g.cin, zp.data_ptr(), g.cin, nbp.scale.data_ptr(),
                      nbp.shift.data_ptr(), nbp.mean.data_ptr(), nbp.invstd.data_ptr(), 0.2, part.data_ptr(),
                      keep=(gc, zp, nbp), desc=_gdesc(g),
                      tag=("dgrad:" + _bf16_kernel_name(g, True), 2.0 * conv_macs(g), conv_bytes(g, 2)))
            elif i > 0:
                b.add("conv_backward_data_bf16", L.mpgan_conv_backward_data_bf16, C.byref(gc), dz.data_ptr(), c,
                      w16b(i).data_ptr(), gas[i - 1].data_ptr(), g.cin, keep=(gc,), desc=_gdesc(g),
                      tag=("dgrad:" + _bf16_kernel_name(g, True), 2.0 * conv_macs(g), conv_bytes(g, 2)))
            elif want_input_grad:
                self.g_x = E(n, *dhw, 1)
                b.add("conv_backward_data_bf16_to_f32", L.mpgan_conv_backward_data_bf16_to_f32, C.byref(gc), dz.data_ptr(),
                      c, store.wp_bwd(recs[0]).data_ptr(), self.g_x.data_ptr(), 1, keep=(gc, self.g_x), desc=_gdesc(g))


# --------------------------------------------------------------------------
# patch discriminator (variant B)
# --------------------------------------------------------------------------
class PatchDiscPlan:
    """4 x (valid k3 conv -> BN -> LeakyReLU 0.2) -> Flatten -> Linear(F,64) -> Linear(64,1) ->
    Sigmoid on small patches (test_runs/GAN.py:136-198).  The 16 perceptual taps are never
    materialised: their L1 terms and gradients come from the raw conv outputs of the two
    passes (`peer`), see mpgan_peer_taps."""

    def __init__(self, disc, store: ParamStore, n: int, spatial: Sequence[int], *, want_backward: bool,
                 want_input_grad: bool, want_param_grads: bool):
        dims = disc.dimensions
        dev = store.flat.device
        dhw = _t3(spatial, dims, 1)
        self.n, self.dhw, self.dims, self.store, self.disc = n, dhw, dims, store, disc
        self.want_input_grad, self.want_param_grads = want_input_grad, want_param_grads
        E = lambda *shape: torch.empty(*shape, device=dev)
        Z = lambda *shape: torch.zeros(*shape, device=dev)
        convs = [disc.model_conv[i] for i in (0, 3, 6, 9)]
        bns = [disc.model_conv[i] for i in (1, 4, 7, 10)]
        lin1, lin2 = disc.model_linear[1], disc.model_linear[2]
        self.convs, self.bns, self.lin1, self.lin2 = convs, bns, lin1, lin2
        self.x_in = E(n, *dhw, 1)
        self.geoms, self.zs, self.nbs, self.recs = [], [], [], []
        size = dhw
        scratch = Scratch(dev)
        for cv in convs:
            g = conv_geom_of(cv, n, size, dims)
            self.geoms.append(g)
            size = g.out_dhw
            if min(size) < 1:
                raise ValueError(f"patch discriminator input {spatial} too small")
            self.zs.append(E(n, *size, cv.out_channels))
            self.nbs.append(NormBuf(n, cv.out_channels, False, dev))
            self.recs.append(store.register_conv(cv, cout=cv.out_channels, cin=cv.in_channels,
                                                 taps=int(torch.tensor(cv.kernel_size).prod())))
            scratch.want_partials(n, size[0] * size[1] * size[2], cv.out_channels)
            scratch.want_ws(g)
        c_last = convs[-1].out_channels
        P_last = size[0] * size[1] * size[2]
        if lin1.in_features != P_last * c_last:
            raise ValueError(f"Linear.in_features {lin1.in_features} != {c_last}*{P_last} for patches {spatial}")
        # Linear(F, 64) as a conv whose kernel spans the last feature map (packing realises the flatten order)
        self.g_l1 = ConvGeom(n, tuple(size), c_last, lin1.out_features, tuple(size), (1, 1, 1), (0, 0, 0))
        self.g_l2 = ConvGeom(n, (1, 1, 1), lin1.out_features, 1, (1, 1, 1), (1, 1, 1), (0, 0, 0))
        self.r_l1 = store.register_conv(lin1, cout=lin1.out_features, cin=c_last, taps=P_last, tco=True)
        # its data gradient as ONE GEMM: (P x 64) * (64 x taps*C) -> the channels-last gradient of the last map
        self.g_l1_bwd = ConvGeom(n, (1, 1, 1), lin1.out_features, P_last * c_last, (1, 1, 1), (1, 1, 1), (0, 0, 0))
        self.r_l2 = store.register_conv(lin2, cout=1, cin=lin1.out_features, taps=1)
        scratch.want_ws(self.g_l1)
        scratch.want_ws(self.g_l2)
        scratch.alloc()
        self.scratch = scratch
        part, ws = scratch.partials, scratch.ws
        self.h = E(n, 1, 1, 1, lin1.out_features)
        self.logit, self.prob = E(n, 1, 1, 1, 1), E(n)
        self.splitk_ws = E(max(ops.conv_splitk_workspace(self.g_l1) // 4, 4))
        L = lib()
        self.lrelu = lambda nb: nb.prologue(ACT_LEAKY, 0.2, None)
        f = self.fwd = Program()
        store.emit_pack(f)
        src, pro = self.x_in, None
        for i, cv in enumerate(convs):
            emit_conv_fwd_norm(f, self.geoms[i], src, store.wp(self.recs[i]), cv.bias, self.zs[i], self.nbs[i], bns[i],
                               part, pro=pro)
            src, pro = self.zs[i], self.lrelu(self.nbs[i])
        gc1, pc = self.g_l1.c(), pro.c()
        f.add("conv_forward_splitk", L.mpgan_conv_forward_splitk, C.byref(gc1), self.zs[-1].data_ptr(), _ld(self.zs[-1]),
              store.wp(self.r_l1).data_ptr(), lin1.bias.data_ptr(), C.byref(pc), self.splitk_ws.data_ptr(),
              self.splitk_ws.numel() * 4, self.h.data_ptr(), _ld(self.h), keep=(gc1, pc, pro),
              tag=("gather_conv_pipe_kernel<splitK>", 2.0 * conv_macs(self.g_l1)))
        emit_conv_fwd(f, self.g_l2, self.h, store.wp(self.r_l2), lin2.bias, self.logit)
        f.add("sigmoid_forward", L.mpgan_sigmoid_forward, self.logit.data_ptr(), n, self.prob.data_ptr())
        self.busy = False
        self.g_x = None
        self._bwd_cache = {}
        if not want_backward:
            return
        self.g_prob = E(n)
        self.dlogit = E(n, 1, 1, 1, 1)
        self.dh = E(n, 1, 1, 1, lin1.out_features)
        self.gas = [E(*z.shape) for z in self.zs]
        # gradients arriving through the perceptual taps of the three head tensors (zero unless a
        # perceptual loss deposited them) and the per-layer (z, y, a) coefficients
        self.tap_g_h, self.tap_g_logit, self.tap_g_prob = Z(*self.h.shape), Z(n), Z(n)
        self.coef_all = Z(4 * len(convs))                      # one buffer: the perceptual loss fills it in one launch
        self.coef = [self.coef_all[4 * i:4 * i + 4] for i in range(len(convs))]
        # constant weights of the perceptual loss (test_runs/GAN.py:288-298: every tap's L1 mean / its numel; Flatten
        # repeats the last activation, key 12): forward terms in the order [layer0 z, y, a, layer1 ..., h, logit, prob]
        nl = len(convs)
        wf, wb = [], []
        for i, z in enumerate(self.zs):
            nel = float(z.numel())
            last = 2.0 if i == nl - 1 else 1.0
            wf += [1.0 / nel, 1.0 / nel, last / nel]
            wb += [1.0 / (nel * nel), 1.0 / (nel * nel), last / (nel * nel), 0.0]
        wf += [1.0 / self.h.numel(), 1.0 / self.logit.numel(), 1.0 / self.prob.numel()]
        self.perc_w_fwd = torch.tensor(wf, device=dev)
        self.perc_w_bwd = torch.tensor(wb, device=dev)
        if want_input_grad:
            self.g_x = E(n, *dhw, 1)

    def backward_program(self, peer: Optional["PatchDiscPlan"]) -> Program:
        key = id(peer) if peer is not None else 0
        if key in self._bwd_cache:
            return self._bwd_cache[key]
        store, L = self.store, lib()
        part, ws = self.scratch.partials, self.scratch.ws
        gv = store.grad_view if self.want_param_grads else (lambda p: None)
        b = Program()
        n = self.n
        # sigmoid: dlogit = (g_prob + tap_prob) * p(1-p) + tap_logit
        b.add("axpby", L.mpgan_axpby, self.g_prob.data_ptr(), 1.0, self.tap_g_prob.data_ptr(), 1.0, n,
              self.g_prob.data_ptr())
        b.add("sigmoid_backward", L.mpgan_sigmoid_backward, self.g_prob.data_ptr(), self.prob.data_ptr(), n,
              self.dlogit.data_ptr())
        b.add("axpby", L.mpgan_axpby, self.dlogit.data_ptr(), 1.0, self.tap_g_logit.data_ptr(), 1.0, n,
              self.dlogit.data_ptr())
        if self.want_param_grads:
            emit_conv_wgrad(b, self.g_l2, self.h, self.dlogit, gv(self.lin2.weight), ws, dbias=gv(self.lin2.bias))
        emit_conv_dgrad(b, self.g_l2, self.dlogit, store.wp_bwd(self.r_l2), self.dh, resid=self.tap_g_h)
        pro4 = self.lrelu(self.nbs[-1])
        if self.want_param_grads:
            emit_conv_wgrad(b, self.g_l1, self.zs[-1], self.dh, gv(self.lin1.weight), ws, pro=pro4,
                            dbias=gv(self.lin1.bias))
        emit_conv_fwd(b, self.g_l1_bwd, self.dh, store.wp_tco(self.r_l1), None,
                      self.gas[-1].view(n, 1, 1, 1, -1))
        for i in range(len(self.convs) - 1, -1, -1):
            pr = None
            if peer is not None:
                pr = ops.PeerTaps(peer.zs[i], peer.nbs[i].scale, peer.nbs[i].shift, self.coef[i])
            emit_norm_bwd(b, self.gas[i], self.zs[i], self.nbs[i], self.lrelu(self.nbs[i]), self.gas[i], part,
                          gv(self.bns[i].weight), gv(self.bns[i].bias), None, peer=pr)
            src = self.zs[i - 1] if i > 0 else self.x_in
            pro_in = self.lrelu(self.nbs[i - 1]) if i > 0 else None
            if self.want_param_grads:
                emit_conv_wgrad(b, self.geoms[i], src, self.gas[i], gv(self.convs[i].weight), ws, pro=pro_in,
                                dbias=gv(self.convs[i].bias))
            if i > 0:
                emit_conv_dgrad(b, self.geoms[i], self.gas[i], store.wp_bwd(self.recs[i]), self.gas[i - 1])
            elif self.want_input_grad:
                emit_conv_dgrad(b, self.geoms[0], self.gas[0], store.wp_bwd(self.recs[0]), self.g_x)
        self._bwd_cache[key] = b
        return b

    def clear_taps(self):
        if hasattr(self, "tap_g_h"):
            self.tap_g_h.zero_()
            self.tap_g_logit.zero_()
            self.tap_g_prob.zero_()
        self.peer = None
        if getattr(self, "ext_used", False):
            self.clear_ext()

    # ---- external gradients on materialised taps (the reference's own perceptual_loss body on TapDict values) ----
    def _ext_buffers(self):
        if not hasattr(self, "gext"):
            # per conv layer: gradients arriving on (z, y, a) = (conv out, norm out, activation) as channels-last tensors
            self.gext = [[torch.zeros_like(z) for _ in range(3)] for z in self.zs]
            self.ext_used = False
        return self.gext

    def deposit_tap_grad(self, key: int, g: torch.Tensor):
        """Gradient of one of the 16 taps (NC(D)HW, as TapSet.materialize returned it)."""
        if not hasattr(self, "gas"):
            raise RuntimeError("this discriminator pass was run without a backward (no gradient can flow into its taps)")
        ge = self._ext_buffers()
        n = self.n
        if key < 12 or key == 12:
            i, kind = (3, 2) if key == 12 else divmod(key, 3)
            z = self.zs[i]
            gg = g.reshape(n, z.shape[-1], *z.shape[1:4])                     # NCDHW (2-D taps: D = 1)
            ge[i][kind].add_(gg.permute(0, 2, 3, 4, 1))
        elif key == 13:
            self.tap_g_h.view(n, -1).add_(g.reshape(n, -1))
        elif key == 14:
            self.tap_g_logit.add_(g.reshape(-1))
        else:
            self.tap_g_prob.add_(g.reshape(-1))
        self.ext_used = True

    def clear_ext(self):
        if hasattr(self, "gext"):
            for trio in self.gext:
                for t in trio:
                    t.zero_()
        self.ext_used = False

    def backward_program_ext(self, peer: Optional["PatchDiscPlan"]) -> Program:
        """backward_program with the external tap gradients folded in, layer by layer:
            g_a += G_a;  dz = BNbwd_act(g_a) + BNbwd_identity(G_y) + G_z
        (BatchNorm's backward is linear in the gradient of its output, so the y-tap's gradient goes through a
        second norm-backward with an identity activation; both add into dgamma / dbeta).  A compatibility path
        built from the existing kernels: three extra element-wise passes per layer."""
        key = ("ext", id(peer) if peer is not None else 0)
        if key in self._bwd_cache:
            return self._bwd_cache[key]
        ge = self._ext_buffers()
        store, L = self.store, lib()
        part, ws = self.scratch.partials, self.scratch.ws
        gv = store.grad_view if self.want_param_grads else (lambda p: None)
        b = Program()
        n = self.n
        add = lambda dst, src: b.add("axpby", L.mpgan_axpby, dst.data_ptr(), 1.0, src.data_ptr(), 1.0, dst.numel(),
                                     dst.data_ptr(), keep=(dst, src))
        b.add("axpby", L.mpgan_axpby, self.g_prob.data_ptr(), 1.0, self.tap_g_prob.data_ptr(), 1.0, n,
              self.g_prob.data_ptr())
        b.add("sigmoid_backward", L.mpgan_sigmoid_backward, self.g_prob.data_ptr(), self.prob.data_ptr(), n,
              self.dlogit.data_ptr())
        b.add("axpby", L.mpgan_axpby, self.dlogit.data_ptr(), 1.0, self.tap_g_logit.data_ptr(), 1.0, n,
              self.dlogit.data_ptr())
        if self.want_param_grads:
            emit_conv_wgrad(b, self.g_l2, self.h, self.dlogit, gv(self.lin2.weight), ws, dbias=gv(self.lin2.bias))
        emit_conv_dgrad(b, self.g_l2, self.dlogit, store.wp_bwd(self.r_l2), self.dh, resid=self.tap_g_h)
        pro4 = self.lrelu(self.nbs[-1])
        if self.want_param_grads:
            emit_conv_wgrad(b, self.g_l1, self.zs[-1], self.dh, gv(self.lin1.weight), ws, pro=pro4,
                            dbias=gv(self.lin1.bias))
        emit_conv_fwd(b, self.g_l1_bwd, self.dh, store.wp_tco(self.r_l1), None, self.gas[-1].view(n, 1, 1, 1, -1))
        for i in range(len(self.convs) - 1, -1, -1):
            pr = None
            if peer is not None:
                pr = ops.PeerTaps(peer.zs[i], peer.nbs[i].scale, peer.nbs[i].shift, self.coef[i])
            gz, gy, ga = ge[i]
            add(self.gas[i], ga)
            emit_norm_bwd(b, self.gas[i], self.zs[i], self.nbs[i], self.lrelu(self.nbs[i]), self.gas[i], part,
                          gv(self.bns[i].weight), gv(self.bns[i].bias), None, peer=pr)
            emit_norm_bwd(b, gy, self.zs[i], self.nbs[i], self.nbs[i].prologue(ACT_NONE), gy, part,
                          gv(self.bns[i].weight), gv(self.bns[i].bias), None)
            add(self.gas[i], gy)
            add(self.gas[i], gz)
            src = self.zs[i - 1] if i > 0 else self.x_in
            pro_in = self.lrelu(self.nbs[i - 1]) if i > 0 else None
            if self.want_param_grads:
                emit_conv_wgrad(b, self.geoms[i], src, self.gas[i], gv(self.convs[i].weight), ws, pro=pro_in,
                                dbias=gv(self.convs[i].bias))
            if i > 0:
                emit_conv_dgrad(b, self.geoms[i], self.gas[i], store.wp_bwd(self.recs[i]), self.gas[i - 1])
            elif self.want_input_grad:
                emit_conv_dgrad(b, self.geoms[0], self.gas[0], store.wp_bwd(self.recs[0]), self.g_x)
        self._bwd_cache[key] = b
        return b
