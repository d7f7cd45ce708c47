"""Kernel-level parity: HIP engine (through the C ABI) vs the CPU oracle restatement.

Parity statement: "vs CPU restatement of TF-1.13.1 semantics; TF parity unpinned" (SURVEY.md 8c)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,ta,tb", [
    (128, 128, 64, 0, 0), (200, 72, 100, 0, 0), (131, 31, 52, 0, 1), (64, 1024, 80, 0, 0),
    (48, 36, 1000, 1, 0), (256, 256, 256, 0, 1), (7, 5, 3, 0, 0), (33, 130, 17, 1, 0),
])
def test_gemm_shapes(M, N, K, ta, tb):
    from avsr_tf1_amd import ops
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    B = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    ref = 0.5 * ((A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)) + 2.0 * C0 + bias
    a, b, c, bi = dev(A), dev(B), dev(C0), dev(bias)
    ops.gemm(ops.mat(a, A.shape[1]), ops.mat(b, B.shape[1]), ops.mat(c, N), M, N, K,
             trans_a=ta, trans_b=tb, alpha=0.5, beta=2.0, bias=bi)
    torch.cuda.synchronize()
    err = np.abs(c.cpu().numpy() - ref).max()
    assert err < 1e-4 * max(1.0, np.abs(ref).max()), err


def test_gemm_splitk_and_two_level_rows():
    from avsr_tf1_amd import ops
    rng = np.random.default_rng(5)
    Bn, T, F, N = 3, 37, 20, 48
    X = rng.standard_normal((Bn, T + 2, F)).astype(np.float32)      # slot layout [B, T+2, F]
    G = rng.standard_normal((Bn, T, N)).astype(np.float32)
    x, g = dev(X), dev(G)
    # dW = sum_{b,t} X[b, t+1, :]^T G[b, t, :]   (TN, K = B*T rows, two-level row addressing on A)
    ref = np.einsum("btf,btn->fn", X[:, 1:T + 1].astype(np.float64), G.astype(np.float64))
    out = torch.zeros(F, N, device="cuda")
    ws = torch.empty(8 * F * N, device="cuda")
    ops.gemm(ops.mat(x, F, T=T, ldo=(T + 2) * F, offset=F), ops.mat(g, N), ops.mat(out, N), F, N, Bn * T,
             trans_a=1, splitk=8, workspace=ws)
    torch.cuda.synchronize()
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-4
    # C with two-level rows: Y[b, t+1, :] = G[b,t,:] @ W
    W = rng.standard_normal((N, F)).astype(np.float32)
    y = torch.zeros(Bn, T + 2, F, device="cuda")
    w = dev(W)
    ops.gemm(ops.mat(g, N), ops.mat(w, F), ops.mat(y, F, T=T, ldo=(T + 2) * F, offset=F), Bn * T, F, N)
    torch.cuda.synchronize()
    yr = np.zeros((Bn, T + 2, F))
    yr[:, 1:T + 1] = G.astype(np.float64) @ W.astype(np.float64)
    assert np.abs(y.cpu().numpy() - yr).max() < 1e-4


@pytest.mark.parametrize("M,N,K,splitk,ta", [(256, 1024, 3000, None, 1), (80, 1024, 777, 4, 1), (40, 136, 50, 1, 1), (300, 260, 1000, 6, 0),
                                             (20, 48, 111, 8, 1)])
def test_gemm_with_column_sums_of_b(M, N, K, splitk, ta):
    """avsr_gemm_desc.colsum: the bias gradient (column sums of d gates) taken from the B tiles of the weight-gradient GEMM, with and
    without split-K, several row tiles (only the first owns the sums), ragged N and K, accumulating onto the destination; alone and
    as an entry of a grouped launch."""
    from avsr_tf1_amd import ops
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    B = rng.standard_normal((K, N)).astype(np.float32)
    c0 = rng.standard_normal(N).astype(np.float32)
    ref = (A.T if ta else A).astype(np.float64) @ B.astype(np.float64)
    cref = B.astype(np.float64).sum(0) + c0
    a, b = dev(A), dev(B)
    ws = torch.empty(1 << 22, device="cuda")
    for grouped in (False, True):
        c = torch.zeros(M, N, device="cuda")
        buf = torch.zeros(7 + N, device="cuda")
        buf[7:] = dev(c0)
        if grouped:
            other = torch.zeros(M, N, device="cuda")
            with ops.gemm_group():
                ops.gemm(ops.mat(a, A.shape[1]), ops.mat(b, N), ops.mat(other, N), M, N, K, trans_a=ta, splitk=splitk, workspace=ws)
                ops.gemm(ops.mat(a, A.shape[1]), ops.mat(b, N), ops.mat(c, N), M, N, K, trans_a=ta, splitk=splitk, workspace=ws,
                         colsum=(buf, 7), colsum_beta=1.0)
        else:
            ops.gemm(ops.mat(a, A.shape[1]), ops.mat(b, N), ops.mat(c, N), M, N, K, trans_a=ta, splitk=splitk, workspace=ws,
                     colsum=(buf, 7), colsum_beta=1.0)
        torch.cuda.synchronize()
        assert np.abs(c.cpu().numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
        assert np.abs(buf[7:].cpu().numpy() - cref).max() < 1e-4 * max(1.0, np.abs(cref).max()), grouped
        assert (buf[:7] == 0).all()
        if grouped:
            assert np.abs(other.cpu().numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


def test_gemm_batched():
    from avsr_tf1_amd import ops
    rng = np.random.default_rng(9)
    nb, M, N, K = 5, 40, 24, 12
    A = rng.standard_normal((nb, K, M)).astype(np.float32)
    B = rng.standard_normal((nb, K, N)).astype(np.float32)
    c = torch.zeros(nb, M, N, device="cuda")
    a, b = dev(A), dev(B)          # keep alive: ops.mat only captures the raw device pointer
    ops.gemm(ops.mat(a, M), ops.mat(b, N), ops.mat(c, N), M, N, K, trans_a=1, batch=nb,
             strides=(K * M, K * N, M * N))
    torch.cuda.synchronize()
    ref = np.einsum("bkm,bkn->bmn", A.astype(np.float64), B.astype(np.float64))
    assert np.abs(c.cpu().numpy() - ref).max() < 1e-4


# ------------------------------------------------------------------------------------------------
def _oracle_stack(x, lens, Ws, bs, reverse, R_out, R_h, R_c):
    """Oracle forward+backward of one LSTM stack (oracle.dynamic_rnn) with a linear probe loss."""
    from oracle import avsr_oracle as O
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    lt = torch.tensor(lens, dtype=torch.int64)
    P = {}
    for l, (W, b) in enumerate(zip(Ws, bs)):
        P[f"s/l{l}/kernel"] = torch.tensor(W, dtype=torch.float64, requires_grad=True)
        P[f"s/l{l}/bias"] = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    cells = [O._Cell(P, f"s/l{l}", "lstm", W.shape[1] // 4) for l, W in enumerate(Ws)]
    xin = O._reverse_sequence(xt, lt) if reverse else xt
    outs, st = O.dynamic_rnn(O._stack_step(cells), tuple(c.zero_state(x.shape[0], torch.float64) for c in cells), xin, lt)
    if reverse:
        outs = O._reverse_sequence(outs, lt)
    c_f, h_f = st[-1]
    loss = (outs * torch.tensor(R_out)).sum() + (h_f * torch.tensor(R_h)).sum() + (c_f * torch.tensor(R_c)).sum()
    loss.backward()
    return (outs.detach().numpy(), h_f.detach().numpy(), c_f.detach().numpy(), xt.grad.numpy(),
            [P[f"s/l{l}/kernel"].grad.numpy() for l in range(len(Ws))],
            [P[f"s/l{l}/bias"].grad.numpy() for l in range(len(Ws))])


@pytest.mark.parametrize("B", [19, 70])      # 70: more than one 64-row slice of the persistent kernels
@pytest.mark.parametrize("persistent", [0, 1, 2, 6, 10])     # 6: + split persistent BPTT; 10: pair-layout forward
@pytest.mark.parametrize("reverse", [0, 1])
@pytest.mark.parametrize("units", [(32,), (32, 48, 32)])
def test_rnn_stack_fwd_bwd(units, reverse, persistent, B):
    from avsr_tf1_amd import ops, params as PR
    from avsr_tf1_amd._lib import RnnLayer, RnnStack
    rng = np.random.default_rng(11 + reverse + len(units))
    T, F = 23, 20
    lens = rng.integers(T // 2, T + 1, size=B).astype(np.int32)
    lens[0] = T
    lens[1] = 1
    x = rng.standard_normal((B, T, F)).astype(np.float32)
    x *= (np.arange(T)[None, :, None] < lens[:, None, None])
    Ws, bs, ins = [], [], []
    i = F
    for u in units:
        Ws.append((rng.standard_normal((i + u, 4 * u)) * 0.3).astype(np.float32))
        bs.append((rng.standard_normal(4 * u) * 0.1).astype(np.float32))
        ins.append(i)
        i = u
    Htop = units[-1]
    R_out = rng.standard_normal((B, T, Htop))
    R_h, R_c = rng.standard_normal((B, Htop)), rng.standard_normal((B, Htop))
    o_out, o_h, o_c, o_dx, o_dW, o_db = _oracle_stack(x, lens, Ws, bs, reverse, R_out, R_h, R_c)

    xd, ld = dev(x), dev(lens, torch.int32)
    st = RnnStack()
    st.B, st.T, st.reverse, st.n_layers, st.cell = B, T, reverse, len(units), 0
    st.len = ld.data_ptr()
    keep = []
    bufs = []
    for l, u in enumerate(units):
        We = dev(PR.lstm_kernel_to_engine(Ws[l]))
        Wt = We.t().contiguous()
        be = dev(PR.lstm_bias_to_engine(bs[l]))
        gates = torch.zeros(B, T, u, 4, device="cuda")
        if l == 0:   # hoisted input projection
            ops.gemm(ops.mat(xd, F), ops.mat(We, 4 * u), ops.mat(gates, 4 * u), B * T, 4 * u, F)
        cs = torch.zeros(B, T, u, device="cuda")
        out = torch.zeros(B, T + 2, u, device="cuda")
        state = torch.empty(4 * B * u, device="cuda")
        hf, cf = torch.zeros(B, u, device="cuda"), torch.zeros(B, u, device="cuda")
        dgates = torch.full((B, T, u, 4), 7.0, device="cuda")
        dstate = torch.empty(12 * B * u, device="cuda")
        L = st.layer[l]
        L.units, L.in_dim, L.hoisted, L.out_col = u, ins[l], int(l == 0), 0
        L.wt, L.w, L.bias = Wt.data_ptr(), We.data_ptr(), be.data_ptr()
        L.gates, L.cs, L.out, L.ld_out = gates.data_ptr(), cs.data_ptr(), out.data_ptr(), u
        L.state, L.h_final, L.c_final = state.data_ptr(), hf.data_ptr(), cf.data_ptr()
        L.dgates, L.dstate = dgates.data_ptr(), dstate.data_ptr()
        bufs.append(dict(We=We, gates=gates, cs=cs, out=out, hf=hf, cf=cf, dgates=dgates))
        keep += [We, Wt, be, gates, cs, out, state, hf, cf, dgates, dstate]
    dout = torch.zeros(B, T + 2, Htop, device="cuda")
    dout[:, 1:T + 1] = dev(R_out)
    dhf, dcf = dev(R_h), dev(R_c)
    top = st.layer[len(units) - 1]
    top.dout, top.ld_dout, top.dout_col = dout.data_ptr(), Htop, 0
    st.dh_final, st.dc_final = dhf.data_ptr(), dcf.data_ptr()

    # one launch per wavefront step | persistent agent-scope kernel | persistent XCD-local kernel
    ops.rnn_set_persistent(bool(persistent), mode=max(persistent, 1))
    try:
        ops.rnn_fwd([st])
        torch.cuda.synchronize()
        assert not ops.rnn_persistent_error()
    finally:
        ops.rnn_set_persistent(False)
    out = bufs[-1]["out"][:, 1:T + 1].cpu().numpy()
    assert np.abs(out - o_out).max() < 2e-5
    assert np.abs(bufs[-1]["hf"].cpu().numpy() - o_h).max() < 2e-5
    assert np.abs(bufs[-1]["cf"].cpu().numpy() - o_c).max() < 2e-5
    assert float(bufs[-1]["out"][:, 0].abs().max()) == 0.0 and float(bufs[-1]["out"][:, T + 1].abs().max()) == 0.0

    ops.rnn_set_persistent(bool(persistent), mode=max(persistent, 1))
    try:
        ops.rnn_bwd([st])
        torch.cuda.synchronize()
        assert not ops.rnn_persistent_error()
    finally:
        ops.rnn_set_persistent(False)
    for l, u in enumerate(units):
        dg = bufs[l]["dgates"]
        i = ins[l]
        dW = torch.zeros(i + u, 4 * u, device="cuda")
        ws = torch.empty(4 * (i + u) * 4 * u, device="cuda")
        if l == 0:
            a_x = ops.mat(xd, F)
        else:
            o = bufs[l - 1]["out"]
            a_x = ops.mat(o, units[l - 1], T=T, ldo=(T + 2) * units[l - 1], offset=units[l - 1])
        ops.gemm(a_x, ops.mat(dg, 4 * u), ops.mat(dW, 4 * u), i, 4 * u, B * T, trans_a=1, splitk=4, workspace=ws)
        o = bufs[l]["out"]
        a_h = ops.mat(o, u, T=T, ldo=(T + 2) * u, offset=(2 * u if reverse else 0))
        ops.gemm(a_h, ops.mat(dg, 4 * u), ops.mat(dW, 4 * u, offset=i * 4 * u), u, 4 * u, B * T, trans_a=1,
                 splitk=4, workspace=ws)
        torch.cuda.synchronize()
        dW_tf = PR.lstm_kernel_from_engine(dW.cpu().numpy())
        db_tf = PR.lstm_bias_from_engine(dg.sum(dim=(0, 1)).reshape(-1).cpu().numpy())
        scale = max(1.0, np.abs(o_dW[l]).max())
        assert np.abs(dW_tf - o_dW[l]).max() < 2e-4 * scale, (l, np.abs(dW_tf - o_dW[l]).max())
        assert np.abs(db_tf - o_db[l]).max() < 2e-4 * max(1.0, np.abs(o_db[l]).max())
    # dX = dgates0 @ Wx0^T
    u0 = units[0]
    dx = torch.zeros(B, T, F, device="cuda")
    ops.gemm(ops.mat(bufs[0]["dgates"], 4 * u0), ops.mat(bufs[0]["We"], 4 * u0), ops.mat(dx, F), B * T, F, 4 * u0, trans_b=1)
    torch.cuda.synchronize()
    assert np.abs(dx.cpu().numpy() - o_dx).max() < 2e-4 * max(1.0, np.abs(o_dx).max())


@pytest.mark.parametrize("start", [0, 7, 10, 25, 31])
def test_adam_step_with_cosine_restarts(start):
    """avsr_adam_step_decay against the oracle's lr_at + Adam formula at several global steps (seq2seq.py:259-280)."""
    from avsr_tf1_amd import ops
    from oracle import avsr_oracle as O
    cfg = O.OracleConfig(architecture="unimodal", video_units=None, audio_units=(8,), warmup_steps=4, lr_decay_steps=10)
    rng = np.random.default_rng(start)
    n = 1000
    p, g = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    m, v = (0.1 * rng.standard_normal(n)).astype(np.float32), (0.01 * rng.random(n)).astype(np.float32)
    dp, dg, dm, dv = (torch.tensor(a, device="cuda") for a in (p, g, m, v))
    step = torch.tensor([start], dtype=torch.int32, device="cuda")
    gn = torch.tensor([float(np.linalg.norm(g))], device="cuda")
    ops.adam_step(dp, dg, dm, dv, n, gn, step, cfg.learning_rate, cfg.warmup_steps, 1.0, first_decay_steps=cfg.lr_decay_steps)
    torch.cuda.synchronize()
    t = start + 1
    gc = g.astype(np.float64) * min(1.0, 1.0 / np.linalg.norm(g.astype(np.float64)))
    m2, v2 = 0.9 * m + 0.1 * gc, 0.999 * v + 0.001 * gc * gc
    lr_t = O.lr_at(cfg, start) * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
    want = p - lr_t * m2 / (np.sqrt(v2) + 1e-8)
    assert int(step.item()) == t
    assert np.abs(dp.cpu().numpy() - want).max() < 2e-6 * max(1.0, lr_t / 1e-3)
    assert np.abs((dp.cpu().numpy() - p) - (want - p)).max() < 1e-3 * np.abs(want - p).max() + 1e-9


def test_unit_partitioned_bptt_kernel_stays_covered():
    """The K-split persistent BPTT is the default; the round-1 kernel partitioned by units takes what it declines (and AVSR_RNN_BWD_KSPLIT=0).
    The switch is read once per process, so the same stack tests run in a child with the K-split form off."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AVSR_RNN_BWD_KSPLIT="0")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_kernels.py"), "-q", "-x", "-k",
                        "test_rnn_stack_fwd_bwd and (2 or 6)"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert " passed" in p.stdout
