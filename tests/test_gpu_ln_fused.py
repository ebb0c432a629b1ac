"""LayerNorm fused into the epilogue of the 384-column panel GEMM (a3t_gemm_desc::ln_*): the GEMM that closes a Conformer
sub-layer (x + a * dropout(branch), encoder_layer.py:117-181) also writes the next sub-layer's norm (layer_norm.py:28-42).
Checked against the two launches it replaces (a3t_gemm + a3t_layernorm_fwd) on the same operands."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mk(rs, *shape, sc=1.0):
    return torch.from_numpy((rs.standard_normal(shape) * sc).astype(np.float32)).to(DEV)


@pytest.mark.parametrize("ydt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,Cin,taps,T,drop,alpha", [
    (160, 128, 1, 0, None, 1.0),
    (1000, 384, 1, 0, (0.2, 99), 1.0),
    (3 * 333, 256, 3, 333, (0.1, 7), 0.5),          # conv over time, ragged last panel
    (32 * 1120, 1536, 3, 1120, (0.2, 1234), 0.5),   # the second FFN conv of configs[1]
    (41000, 384, 1, 0, None, 1.0),                  # more panels than CUs: two tiles per workgroup
])
def test_panel_gemm_writes_the_next_layernorm(M, Cin, taps, T, drop, alpha, ydt):
    from a3t_amd import ops
    from a3t_amd._lib import BF16
    rs = np.random.RandomState(M + Cin)
    N = 384
    x = _mk(rs, M, Cin).bfloat16()
    W = _mk(rs, N, taps, Cin, sc=0.05).bfloat16() if taps > 1 else _mk(rs, N, Cin, sc=0.05).bfloat16()
    bias, R = _mk(rs, N), _mk(rs, M, N, sc=2.0) + 3.0            # (a mean far from zero: the variance is taken about the mean)
    g, b = _mk(rs, N) + 1.0, _mk(rs, N)
    K = taps * Cin
    assert ops.gemm_pn_supported(M, N, K, taps, ops.PN_LN)

    def run(ln):
        out = torch.empty(M, N, device=DEV)
        if taps > 1:
            ops.conv_fwd(x, W, out, T, (taps - 1) // 2, bias=bias, R=R, alpha=alpha, compute=BF16, drop=drop, ln=ln)
        else:
            ops.linear_fwd(x, W, out, bias=bias, R=R, alpha=alpha, compute=BF16, drop=drop, ln=ln)
        return out

    y = torch.full((M, N), 7.0, device=DEV, dtype=ydt)
    mean, rstd = torch.zeros(M, device=DEV), torch.zeros(M, device=DEV)
    out = run((g, b, y, mean, rstd, 1e-12))
    out2 = run(None)
    y2 = torch.empty(M, N, device=DEV, dtype=ydt)
    mean2, rstd2 = torch.zeros(M, device=DEV), torch.zeros(M, device=DEV)
    ops.layernorm_fwd(out2, g, b, y2, mean2, rstd2, 1e-12)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)                                  # the GEMM output itself is untouched
    assert float(((mean - mean2).abs() / (mean2.abs() + 1e-3)).max()) < 1e-5
    assert float(((rstd - rstd2).abs() / rstd2.abs()).max()) < 1e-5
    ref = torch.nn.functional.layer_norm(out2.double(), (N,), g.double(), b.double(), 1e-12)
    tol = 2.0 ** -7 if ydt == torch.bfloat16 else 2e-5
    scale = float(ref.abs().max())
    assert float((y.double() - ref).abs().max()) <= tol * scale
    assert float((y.double() - y2.double()).abs().max()) <= tol * scale
    if ydt == torch.bfloat16:                                       # same rounding of (almost always) the same fp32 value
        assert float((y != y2).float().mean()) < 2e-3


def test_fused_layernorm_needs_the_panel_kernel():
    """a3t_gemm refuses a descriptor with ln_y that the panel kernel does not take (N != 384) instead of dropping the norm."""
    from a3t_amd import ops
    from a3t_amd._lib import BF16
    rs = np.random.RandomState(0)
    M, N, K = 512, 256, 128
    x, W = _mk(rs, M, K).bfloat16(), _mk(rs, N, K).bfloat16()
    out = torch.empty(M, N, device=DEV)
    g, b = _mk(rs, N), _mk(rs, N)
    y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    mean, rstd = torch.zeros(M, device=DEV), torch.zeros(M, device=DEV)
    assert not ops.gemm_pn_supported(M, N, K, 1, ops.PN_LN)
    with pytest.raises(RuntimeError):
        ops.linear_fwd(x, W, out, compute=BF16, ln=(g, b, y, mean, rstd, 1e-12))
