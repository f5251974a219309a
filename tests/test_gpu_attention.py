"""Single-query attention with the K and V projections absorbed (product2vec.py:48-68; csrc/attention.hip): saved tensors and parity
with non-zero biases.  Needs an MI355X."""
import pytest
import torch

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ absorbed attention
def test_attention_saves_no_kv_buffer_and_matches_the_oracle_with_bias():
    """The K|V projections are absorbed into the per-sample side: the saved tensors are [B,4,D]-sized, not [B*N,2D], and
    forward / every gradient equal the oracle's nn.MultiheadAttention restatement with NON-ZERO in_proj / out_proj biases
    (torch initialises them to 0, which would hide an error in the bias algebra: the key bias leaves the softmax, the
    value bias enters through sum_n p_n)."""
    from oracle import p2v_oracle
    from p_companion_amd import ops
    g = torch.Generator().manual_seed(2)
    B, N, D = 37, 11, 128
    st = p2v_oracle.init_state(3)
    st["attention.in_proj_bias"] = torch.randn(3 * D, generator=g) * 0.3
    st["attention.out_proj.bias"] = torch.randn(D, generator=g) * 0.3
    q = torch.randn(B, D, generator=g)
    kv = torch.randn(B, N, D, generator=g)
    params = {k: v.cuda() for k, v in st.items()}
    out, sv = ops.attention_forward(params, q.cuda(), kv.cuda())
    assert "kv" not in sv and sv["qt"].shape == (B, 4, D) and sv["c"].shape == (B, 4, D)
    assert sum(v.numel() for v in sv.values() if torch.is_tensor(v)) < B * N * 2 * D
    qr, kr = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    pr = {k: v.clone().requires_grad_(True) for k, v in st.items() if k.startswith("attention.")}
    ref = p2v_oracle.attention(qr, kr, dict(st, **pr))
    assert float((out.cpu() - ref.detach()).abs().max()) < 2e-5
    w = torch.randn(B, D, generator=g)
    (ref * w).sum().backward()
    grads, dq, dk = ops.attention_backward(params, q.cuda(), kv.cuda(), w.cuda(), sv)
    tol = lambda r: 2e-6 + 2e-4 * float(r.abs().max())
    assert float((dq.cpu() - qr.grad).abs().max()) <= tol(qr.grad)
    assert float((dk.cpu() - kr.grad).abs().max()) <= tol(kr.grad)
    for k, p in pr.items():
        assert float((grads[k].cpu() - p.grad).abs().max()) <= tol(p.grad), k
    # the key-bias gradient is analytically zero: the reference holds rounding noise there, this path an exact 0
    assert float(grads["attention.in_proj_bias"][D:2 * D].abs().max()) == 0.0
    assert float(pr["attention.in_proj_bias"].grad[D:2 * D].abs().max()) < 1e-5
