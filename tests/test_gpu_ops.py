"""HIP kernels vs the oracle, op by op, through the C ABI.  Needs an MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import p2v_oracle


@pytest.fixture(scope="module")
def ops():
    from p_companion_amd import ops as o
    assert torch.cuda.is_available()
    return o


def dev(x, dtype=None):
    t = torch.as_tensor(np.asarray(x)) if not isinstance(x, torch.Tensor) else x
    if dtype is not None:
        t = t.to(dtype)
    return t.contiguous().cuda()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(a, b, atol, rtol=0.0, what=""):
    a = a.detach().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(a)
    b = b.detach().cpu() if isinstance(b, torch.Tensor) else torch.as_tensor(b)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f"{what}: max err {err.max().item():.3e} (tol {atol:.1e}+{rtol:.1e}*|ref|)"


# ------------------------------------------------------------------ shared MFMA GEMMs
@pytest.mark.parametrize("rows,in_dim,out_dim,act", [(1, 128, 128, 0), (130, 64, 100, 2), (257, 128, 256, 1),
                                                      (64, 32, 64, 0), (300, 100, 36, 0), (5, 64, 34800, 0),
                                                      # two column tiles per row tile (paired tile numbering, padding tiles):
                                                      (1000, 64, 300, 1), (700, 128, 512, 0),          # 8-wave kernel, 256 < N <= 512
                                                      (26000, 128, 200, 2), (24700, 64, 256, 0)])      # 4-wave column tiles, ragged N / rows
def test_linear_forward(ops, rows, in_dim, out_dim, act):
    x, w, b = rnd(rows, in_dim, seed=1), rnd(out_dim, in_dim, seed=2, scale=0.1), rnd(out_dim, seed=3)
    y = ops.linear_forward(dev(x), dev(w), dev(b), act=act)
    ref = x.double() @ w.double().T + b.double()
    ref = torch.tanh(ref) if act == 1 else torch.relu(ref) if act == 2 else ref
    close(y, ref.float(), 2e-5, what="linear_forward")


def test_linear_forward_gather_zero_rows(ops):
    table, w = rnd(50, 128, seed=4), rnd(256, 128, seed=5, scale=0.1)
    idx = torch.tensor([3, -1, 49, 0, -1, 7] * 30, dtype=torch.int32)
    y = ops.linear_forward(dev(table), dev(w), None, idx=dev(idx))
    tab = torch.cat([table, torch.zeros(1, 128)])
    ref = tab[idx.long()] @ w.T
    close(y, ref, 2e-5, what="gather")
    assert float(y[1].abs().max()) == 0.0          # a -1 slot is an exact zero row


@pytest.mark.parametrize("rows,out_dim,in_dim", [(1, 128, 128), (1000, 256, 128), (4099, 32, 64), (77, 128, 256),
                                                 (513, 100, 64)])
def test_linear_backward_weight(ops, rows, out_dim, in_dim):
    dy, x = rnd(rows, out_dim, seed=6), rnd(rows, in_dim, seed=7)
    dw, db = ops.linear_backward_weight(dev(dy), dev(x), out_dim, in_dim)
    close(dw, (dy.double().T @ x.double()).float(), 1e-4 * max(1.0, rows ** 0.5 / 8), what="dW")
    close(db, dy.double().sum(0).float(), 1e-4 * max(1.0, rows ** 0.5 / 8), what="db")
    dw2, _ = ops.linear_backward_weight(dev(dy), dev(x), out_dim, in_dim)
    assert torch.equal(dw, dw2)                      # fixed-order split-K: bitwise reproducible


def test_linear_backward_input(ops):
    dy, w = rnd(200, 256, seed=8), rnd(256, 128, seed=9, scale=0.1)
    close(ops.linear_backward_input(dev(dy), dev(w)), (dy.double() @ w.double()).float(), 2e-5, what="dx")


@pytest.mark.parametrize("rows,k,n", [(20000, 256, 256), (30000, 128, 256), (50000, 256, 128)])
def test_matrix_core_products_are_fp32_grade(ops, rows, k, n):
    """The persistent NT / full-tile TN kernels evaluate each fp32 product as six bf16 matrix-core products of a
    three-way split (common.h split3).  Their error against an fp64 product must not exceed what a plain fp32
    product (torch's fp32 matmul on the same GPU) shows on the same operands."""
    x, w = dev(rnd(rows, k, seed=21)), dev(rnd(n, k, seed=22, scale=0.1))
    ref = x.double() @ w.double().T
    y = ops.linear_forward(x, w, None)
    e_hip = float((y.double() - ref).abs().max())
    e_f32 = float(((x @ w.T).double() - ref).abs().max())
    assert e_hip <= 1.25 * e_f32 + 1e-7, f"NT: {e_hip:.3e} vs fp32 {e_f32:.3e}"
    dy = dev(rnd(rows, n, seed=23))
    dw, _ = ops.linear_backward_weight(dy, x, n, k)
    refw = dy.double().T @ x.double()
    e_hip = float((dw.double() - refw).abs().max())
    e_f32 = float(((dy.T @ x).double() - refw).abs().max())
    assert e_hip <= 1.25 * e_f32 + 1e-6, f"TN: {e_hip:.3e} vs fp32 {e_f32:.3e}"


def _adversarial(kind, rows, k, n):
    g = torch.Generator().manual_seed(77)
    u = lambda *s: torch.rand(*s, generator=g)
    if kind == "wide_exponents":          # every row mixes magnitudes 1e-15 .. 1e15; weights 1e-3 .. 1e3
        x = torch.randn(rows, k, generator=g) * 10.0 ** (u(rows, k) * 30 - 15)
        w = torch.randn(n, k, generator=g) * 10.0 ** (u(n, k) * 6 - 3)
    elif kind == "cancellation":          # +a, -a pairs along K against pairwise-equal weights: exact sum 0 + a small term
        a = torch.randn(rows, k // 2, generator=g) * 1e3
        x = torch.stack([a, -a], 2).reshape(rows, k) + torch.randn(rows, k, generator=g) * 1e-3
        wh = torch.randn(n, k // 2, generator=g)
        w = torch.stack([wh, wh], 2).reshape(n, k)
    elif kind == "tiny":                  # operands near 2^-120: the third bf16 piece falls into bf16's subnormal range
        x = torch.randn(rows, k, generator=g) * 2.0 ** -120
        w = torch.randn(n, k, generator=g)
    elif kind == "big_integers":          # integers beyond 2^16: exact in fp32, need all three pieces
        x = torch.randint(-(1 << 22), 1 << 22, (rows, k), generator=g).float()
        w = torch.randint(-64, 64, (n, k), generator=g).float()
    return x.cuda().contiguous(), w.cuda().contiguous()


@pytest.mark.parametrize("kind", ["wide_exponents", "cancellation", "tiny", "big_integers"])
@pytest.mark.parametrize("rows,k,n", [(26000, 256, 256), (26000, 128, 256)])
def test_matrix_core_products_fp32_grade_on_adversarial_operands(ops, kind, rows, k, n):
    """The six-product bf16 form on operands chosen against it: per-row exponent spreads of 30 decades, exact
    cancellation along K, values whose third piece is a bf16 subnormal, integers that need all 24 mantissa bits.
    Bound: the error of a plain fp32 product (torch matmul, same GPU) x 1.25, plus 4e-7 of the row's sum |x w|
    (what an exact-fp32 fma chain is allowed: MI355X_MICROARCH, 'FP32-input MFMA'), plus fp32's smallest normal."""
    x, w = _adversarial(kind, rows, k, n)
    ref = x.double() @ w.double().T
    mag = x.double().abs() @ w.double().abs().T
    y = ops.linear_forward(x, w, None)
    assert torch.isfinite(y).all()
    err = (y.double() - ref).abs()
    e32 = ((x @ w.T).double() - ref).abs()
    tol = 1.25 * e32.max() + 4e-7 * mag + 1.2e-38
    worst = float((err - tol).max())
    assert worst <= 0.0, f"NT {kind}: error exceeds the fp32-grade bound by {worst:.3e} (max err {float(err.max()):.3e}, fp32 {float(e32.max()):.3e})"
    # the weight-gradient (TN) form: contraction over the rows
    dy = torch.randn(rows, n, generator=torch.Generator().manual_seed(5)).cuda()
    if kind == "tiny":
        dy = dy * 2.0 ** -60                                   # products ~2^-180 would underflow fp32 altogether: keep them representable
        x = x * 2.0 ** 60
    dw, _ = ops.linear_backward_weight(dy, x, n, k)
    refw = dy.double().T @ x.double()
    magw = dy.double().abs().T @ x.double().abs()
    errw = (dw.double() - refw).abs()
    e32w = ((dy.T @ x).double() - refw).abs()
    tolw = 1.25 * e32w.max() + 4e-7 * magw + 1.2e-38
    worst = float((errw - tolw).max())
    assert worst <= 0.0, f"TN {kind}: error exceeds the fp32-grade bound by {worst:.3e} (max err {float(errw.max()):.3e}, fp32 {float(e32w.max()):.3e})"


# ------------------------------------------------------------------ P6 FFN
def _state(seed=11):
    st = p2v_oracle.init_state(seed)
    st["ffn.1.weight"] = 1.0 + 0.1 * rnd(256, seed=seed + 1)
    st["ffn.1.bias"] = 0.1 * rnd(256, seed=seed + 2)
    st["attention.in_proj_bias"] = 0.05 * rnd(384, seed=seed + 3)
    st["attention.out_proj.bias"] = 0.05 * rnd(128, seed=seed + 4)
    return st


def _to_dev(st):
    return {k: v.clone().cuda() for k, v in st.items()}


SEGS = [([0], 37), ([0, 8, 8 + 48, 8 + 48 + 8], 8 + 48 + 8 + 40), ([0, 130, 130 + 300], 130 + 300 + 129),
        # >= 192 row tiles: the N = 256 products run as paired 128-wide column tiles (three workgroups per CU), the weight
        # gradients through the full-tile kernels; ragged segment ends, a padding row tile in the paired numbering
        ([0, 9001, 9001 + 15000], 9001 + 15000 + 1703)]


@pytest.mark.parametrize("starts,rows", SEGS)
def test_ffn_forward_train(ops, starts, rows):
    st = _state()
    table = rnd(90, 128, seed=20)
    g = torch.Generator().manual_seed(21)
    idx = torch.randint(-1, 90, (rows,), generator=g, dtype=torch.int32)
    tab = torch.cat([table, torch.zeros(1, 128)])
    x = tab[idx.long()]
    ref_st = {k: v.clone() for k, v in st.items()}
    bounds = list(starts) + [rows]
    ref = torch.cat([p2v_oracle.ffn(x[bounds[i]:bounds[i + 1]], ref_st, True) for i in range(len(starts))])
    dst = _to_dev(st)
    y, sv = ops.ffn_forward_train(dst, dev(table), dev(idx), rows, starts)
    close(y, ref, 5e-6, what="ffn y")
    close(dst["ffn.1.running_mean"], ref_st["ffn.1.running_mean"], 1e-6, what="running_mean")
    close(dst["ffn.1.running_var"], ref_st["ffn.1.running_var"], 1e-6, what="running_var")
    assert int(dst["ffn.1.num_batches_tracked"]) == len(starts)


def test_ffn_forward_eval(ops):
    st = _state()
    st["ffn.1.running_mean"] = 0.1 * rnd(256, seed=30)
    st["ffn.1.running_var"] = 0.5 + torch.rand(256, generator=torch.Generator().manual_seed(31))
    x = rnd(200, 128, seed=32)
    close(ops.ffn_forward_eval(_to_dev(st), dev(x), None, 200), p2v_oracle.ffn(x, st, False), 5e-6, what="eval")


@pytest.mark.parametrize("starts,rows", SEGS)
def test_ffn_backward(ops, starts, rows):
    st = _state()
    x = rnd(rows, 128, seed=40)
    dy = rnd(rows, 128, seed=41, scale=0.1)
    leaves = {k: st[k].clone().requires_grad_(True) for k in p2v_oracle.TRAINABLE if k.startswith("ffn")}
    work = {k: v.clone() for k, v in st.items()}
    work.update(leaves)
    xin = x.clone().requires_grad_(True)
    bounds = list(starts) + [rows]
    y = torch.cat([p2v_oracle.ffn(xin[bounds[i]:bounds[i + 1]], work, True, update_running=False)
                   for i in range(len(starts))])
    (y * dy).sum().backward()
    dst = _to_dev(st)
    _, sv = ops.ffn_forward_train(dst, dev(x), None, rows, starts, update_running=False)
    grads, dx = ops.ffn_backward(dst, dev(x), None, dev(dy), sv, need_dx=True)
    for k, leaf in leaves.items():
        ref = leaf.grad
        tol = 2e-5 * max(1.0, float(ref.abs().max())) * max(1.0, (rows / 1000) ** 0.5)
        close(grads[k], ref, tol, what=k)
    close(dx, xin.grad, 2e-5, what="dx")


# ------------------------------------------------------------------ P7 attention
@pytest.mark.parametrize("B,N", [(1, 1), (8, 6), (70, 33), (5, 48)])
def test_attention(ops, B, N):
    st = _state()
    q, kv = rnd(B, 128, seed=50), rnd(B, N, 128, seed=51)
    dout = rnd(B, 128, seed=52)
    names = [k for k in p2v_oracle.TRAINABLE if k.startswith("attention")]
    leaves = {k: st[k].clone().requires_grad_(True) for k in names}
    work = dict(st); work.update(leaves)
    qi, ki = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    ref = p2v_oracle.attention(qi, ki, work)
    (ref * dout).sum().backward()
    dst = _to_dev(st)
    out, sv = ops.attention_forward(dst, dev(q), dev(kv))
    close(out, ref, 3e-6, what="attention out")
    grads, dq, dk = ops.attention_backward(dst, dev(q), dev(kv), dev(dout), sv)
    close(dq, qi.grad, 1e-5, what="dquery")
    close(dk, ki.grad, 1e-5, what="dkeys")
    for k in names:
        close(grads[k], leaves[k].grad, 3e-5 * max(1.0, float(leaves[k].grad.abs().max())), what=k)


# ------------------------------------------------------------------ P9 loss, P10 Adam
@pytest.mark.parametrize("B,K", [(1, 1), (8, 5), (300, 5), (17, 8)])
def test_triplet_loss(ops, B, K):
    a, p, n = rnd(B, 128, seed=60), rnd(B, 128, seed=61), rnd(B, K, 128, seed=62)
    p[0] = a[0] + 0.01                                  # a pair inside the margin and one far outside
    ai, pi, ni = (t.clone().requires_grad_(True) for t in (a, p, n))
    loss, dpos, dneg = p2v_oracle.triplet_loss(ai, pi, ni, 1.0)
    loss.backward()
    out = ops.triplet_loss(dev(a), dev(p), dev(n), 1.0)
    close(out["loss"], loss.reshape(1), 1e-6, what="loss")
    close(out["d_pos"], dpos, 1e-5, what="d_pos")
    close(out["d_neg"], dneg, 1e-5, what="d_neg")
    close(out["da"], ai.grad, 1e-6, what="da"); close(out["dp"], pi.grad, 1e-6, what="dp")
    close(out["dn"], ni.grad, 1e-6, what="dn")


def test_adam_matches_torch(ops):
    n = 1003 * 4
    p0, gs = rnd(n, seed=70), [rnd(n, seed=71 + i, scale=0.01) for i in range(4)]
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3)
    p = dev(p0); m = torch.zeros_like(p); v = torch.zeros_like(p)
    step = torch.zeros(1, dtype=torch.int64, device="cuda"); scal = torch.zeros(2, device="cuda")
    for g in gs:
        ref.grad = g.clone(); opt.step()
        ops.adam_step(p, dev(g), m, v, step, scal)
    assert int(step) == 4
    close(p, ref.detach(), 2e-7, what="adam param")
    close(m, opt.state[ref]["exp_avg"], 1e-8, what="exp_avg")
    close(v, opt.state[ref]["exp_avg_sq"], 1e-10, what="exp_avg_sq")
