"""Helpers shared by the -m gpu parity tests (layout conversion, tolerances)."""
import torch


def to_cl(x: torch.Tensor) -> torch.Tensor:
    """NC(D)HW cpu/cuda tensor -> channels-last (N,D,H,W,C) contiguous CUDA tensor."""
    if x.dim() == 4:
        x = x.unsqueeze(2)
    return x.permute(0, 2, 3, 4, 1).contiguous().cuda()


def from_cl(x: torch.Tensor, dims: int) -> torch.Tensor:
    """(N,D,H,W,C) cuda -> NC(D)HW cpu."""
    y = x.permute(0, 4, 1, 2, 3).contiguous().cpu()
    return y.squeeze(2) if dims == 2 else y


def t3(v, dims, fill):
    """Per-dimension tuple for a dims-D op, depth padded with `fill`."""
    v = (v,) * dims if isinstance(v, int) else tuple(v)
    return (fill,) * (3 - dims) + v


def assert_close(got, want, rtol=2e-4, atol=None, what="", outliers=0.0, outlier_cap=0.05):
    """|got-want| <= atol + rtol*|want| elementwise.  `outliers` > 0 tolerates that
    fraction of elements (at least one) beyond it, each still within outlier_cap *
    max|want|: gradients through LeakyReLU/PReLU jump when an activation sits
    within rounding of the kink and the two implementations land on opposite sides."""
    got = got.detach().double().cpu()
    want = want.detach().double().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = want.abs().max().item() + 1e-30
    atol = atol if atol is not None else 2e-5 * scale
    err = (got - want).abs()
    bad = err > (atol + rtol * want.abs())
    if outliers > 0 and bad.any():
        allowed = max(1, int(outliers * bad.numel()))
        if bad.sum().item() <= allowed and err.max().item() <= outlier_cap * scale:
            return
    if bad.any():
        i = err.argmax()
        raise AssertionError(f"{what}: max abs err {err.max().item():.3e} (scale {scale:.3e}) at flat index {i.item()} "
                             f"got {got.flatten()[i].item():.6e} want {want.flatten()[i].item():.6e}; "
                             f"{bad.sum().item()}/{bad.numel()} out of tolerance")
