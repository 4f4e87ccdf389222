"""CPU model of the lane / register bookkeeping of attention_reg_kernel (csrc/attention.hip): the MFMA 32x32x16
operand and accumulator layouts are modelled per lane (the same model reproduces the LDS kernel's bookkeeping, which
runs on hardware), and the register-resident kernel's permuted key order -- MFMA row rho fed with key pi(rho), P^T
consumed straight from the accumulator registers, V^T fragments as eight consecutive keys -- must reproduce plain
softmax(Q K^T) V for every length 1..128.  No GPU involved: this pins the index arithmetic, not the kernel."""
import numpy as np
import pytest

HD = 64


def mfma_32x32x16(a_frag, b_frag, acc):
    """a_frag, b_frag: [64 lanes, 8]; acc: [64 lanes, 16].  A[rho][8 g + e] sits in lane (i = rho, g) slot e,
    B[8 g + e][j] in lane (j, g) slot e, D[rho][j] in lane (j, g) register r with rho = (r & 3) + 8 (r >> 2) + 4 g."""
    A = np.zeros((32, 16))
    B = np.zeros((16, 32))
    for lane in range(64):
        i, g = lane & 31, lane >> 5
        A[i, 8 * g:8 * g + 8] = a_frag[lane]
        B[8 * g:8 * g + 8, i] = b_frag[lane]
    D = A @ B
    out = acc.copy()
    for lane in range(64):
        j, g = lane & 31, lane >> 5
        for r in range(16):
            out[lane, r] += D[(r & 3) + 8 * (r >> 2) + 4 * g, j]
    return out


def key_perm(i):
    return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1)


def attend_reg_model(q, k, v, cls_only=False):
    """q, k, v: [T, 64] (q in the log2 domain).  Mirrors attend_reg<N> lane by lane."""
    T = q.shape[0]
    N = (T + 31) >> 5
    lanes = np.arange(64)
    g, i = lanes >> 5, lanes & 31
    kf = np.zeros((N, 4, 64, 8))
    vf = np.zeros((N, 2, 2, 64, 8))
    for kb in range(N):
        for lane in range(64):
            key = kb * 32 + key_perm(i[lane])
            row = min(key, T - 1)
            for sx in range(4):
                kf[kb, sx, lane] = k[row, 32 * g[lane] + 8 * sx:32 * g[lane] + 8 * sx + 8]
            for u in range(2):
                key0 = kb * 32 + 16 * u + 8 * g[lane]
                for e in range(8):
                    if key0 + e < T:
                        vf[kb, u, 0, lane, e] = v[key0 + e, i[lane]]
                        vf[kb, u, 1, lane, e] = v[key0 + e, i[lane] + 32]
    out = np.zeros((1 if cls_only else T, HD))
    q_end = 1 if cls_only else T
    for qb0 in range(0, q_end, 32):
        qf = np.zeros((4, 64, 8))
        for lane in range(64):
            row = min(qb0 + i[lane], T - 1)
            for sx in range(4):
                qf[sx, lane] = q[row, 32 * g[lane] + 8 * sx:32 * g[lane] + 8 * sx + 8]
        m_run = np.full(64, -np.inf)
        l_run = np.zeros(64)
        o0 = np.zeros((64, 16))
        o1 = np.zeros((64, 16))
        for kb in range(N):
            st = np.zeros((64, 16))
            for sx in range(4):
                st = mfma_32x32x16(kf[kb, sx], qf[sx], st)
            if kb == N - 1:
                for lane in range(64):
                    for r in range(16):
                        if kb * 32 + 16 * (r >> 3) + 8 * g[lane] + (r & 7) >= T:
                            st[lane, r] = -np.inf
            bm = st.max(1)
            bm = np.maximum(bm, bm[lanes ^ 32])
            m_new = bm if kb == 0 else np.maximum(m_run, bm)
            if kb > 0:
                alpha = np.exp2(m_run - m_new)
                l_run *= alpha
                o0 *= alpha[:, None]
                o1 *= alpha[:, None]
            m_run = m_new
            p = np.exp2(st - m_new[:, None])
            l_run += p.sum(1)
            for u in range(2):
                pf = p[:, 8 * u:8 * u + 8]
                o0 = mfma_32x32x16(vf[kb, u, 0], pf, o0)
                o1 = mfma_32x32x16(vf[kb, u, 1], pf, o1)
        l_tot = l_run + l_run[lanes ^ 32]
        for lane in range(64):
            if qb0 + i[lane] < q_end:
                for db, o in enumerate((o0, o1)):
                    for rq in range(4):
                        d0 = db * 32 + 8 * rq + 4 * g[lane]
                        out[qb0 + i[lane], d0:d0 + 4] = o[lane, 4 * rq:4 * rq + 4] / l_tot[lane]
    return out


def reference(q, k, v):
    s = q @ k.T
    p = np.exp2(s - s.max(1, keepdims=True))
    return (p / p.sum(1, keepdims=True)) @ v


@pytest.mark.parametrize("T", [1, 2, 7, 8, 9, 31, 32, 33, 40, 63, 64, 65, 74, 95, 96, 97, 100, 127, 128])
def test_register_kernel_bookkeeping(T):
    rng = np.random.default_rng(T)
    q, k, v = (rng.standard_normal((T, HD)) for _ in range(3))
    q *= 0.5
    got = attend_reg_model(q, k, v)
    want = reference(q, k, v)
    assert np.abs(got - want).max() < 1e-12
    got1 = attend_reg_model(q, k, v, cls_only=True)
    assert np.abs(got1[0] - want[0]).max() < 1e-12
