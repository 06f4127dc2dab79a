"""Weight re-layouts the kernels expect (done once at model load, on the host or GPU)."""
import torch


def interleave_gate_up(w_gate, w_up):
    """[I,K] gate and [I,K] up -> [2I,K] with rows interleaved in blocks of 16:
    rows 32j..32j+15 = gate features 16j..16j+15, rows 32j+16..32j+31 = up features 16j..16j+15
    (layout required by TCAVT_EPI_SILU_MUL, include/tcavt.h)."""
    I, K = w_gate.shape
    assert w_up.shape == (I, K) and I % 16 == 0
    g = w_gate.reshape(I // 16, 16, K)
    u = w_up.reshape(I // 16, 16, K)
    return torch.stack([g, u], dim=1).reshape(2 * I, K).contiguous()
