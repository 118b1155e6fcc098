"""The C-ABI library loads on a machine without a GPU and exports every symbol
include/mpgan_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mpgan_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpgan_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    names = _declared()
    for must in ("mpgan_conv_forward", "mpgan_conv_backward_data", "mpgan_conv_backward_weight", "mpgan_norm_finalize",
                 "mpgan_norm_bwd_apply", "mpgan_adam_step", "mpgan_l1_loss", "mpgan_bce_forward",
                 "mpgan_patch_gather", "mpgan_pack_weights"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from mpgan_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libmpgan_hip.so not built (run __graft_entry__.build())")
    handle = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(handle, n)]
    assert not missing, missing


def test_python_binding_covers_every_declared_symbol():
    from mpgan_amd import _lib
    declared = set(_declared())
    bound = set(_lib.SIGNATURES)
    assert declared == bound, (sorted(declared - bound), sorted(bound - declared))


def test_error_reporting_without_gpu():
    """Argument validation happens on the host before any launch."""
    from mpgan_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    lib = _lib.lib()
    assert lib.mpgan_abi_version() == 2
    # the geometry struct of the Python binding has the header's fields (ABI 2 added flags + min_blocks at its end)
    header = open(os.path.join(ROOT, "include", "mpgan_hip.h")).read()
    body = re.sub(r"/\*.*?\*/", "", header[header.index("typedef struct {"):header.index("} mpgan_conv_geom;")], flags=re.S)
    fields = re.findall(r"int32_t\s+([a-z_, ]+?)(?:\[3\])?;", body)
    names = [n.strip() for f in fields for n in f.split(",")]
    assert names == [n for n, _ in _lib.ConvGeomC._fields_], (names, _lib.ConvGeomC._fields_)
    rc = lib.mpgan_adam_step(None, None, None, None, 0, 1e-3, 0.5, 0.999, 1e-8, 1, 1.0, None)
    assert rc == -1 and b"adam_step" in lib.mpgan_last_error()
    g = _lib.ConvGeomC()
    assert lib.mpgan_conv_variant(ctypes.byref(g), 0, 0) == -1       # zeroed geometry is rejected
    g.n, g.cin, g.cout, g.flags = 1, 16, 16, 4                        # unknown flag bits are rejected, not ignored
    for d in range(3):
        g.in_dhw[d] = g.out_dhw[d] = 8
        g.k[d], g.stride[d], g.pad[d] = 3, 1, 1
    assert lib.mpgan_conv_variant(ctypes.byref(g), 0, 0) == -1 and b"flags" in lib.mpgan_last_error()
    g.flags = 1
    assert lib.mpgan_conv_variant(ctypes.byref(g), 0, 0) == 18        # the 3-D patch kernel, bf16 operands or not
