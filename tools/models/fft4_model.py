#!/usr/bin/env python3
"""numpy model of the wave-level transform of k_fused_rev.hip (R = 16) / k_fused_team.hip: a 1024-point complex FFT as four
256-point ones (decimation in time; 16 lanes x 16 registers each: radix-16 in registers, LDS transpose inside the 16-lane
group, radix-16 in registers -- no cross-lane stage), joined by a radix-4 pass that is fused with the real-FFT untangle
(and, for a team of S waves, with the team's radix-S join).  Checks the index maps and twiddles against numpy.fft.
    python tools/models/fft4_model.py"""
import numpy as np

rng = np.random.default_rng(0)


def W(n, k):
    return np.exp(-2j * np.pi * (k % n) / n)


def wave_fft_4x256(z):
    """z[1024] complex -> E[u][k'] (4 x 256): lane = 16 u + l, register r holds z[4 (l + 16 r) + u]."""
    E = np.zeros((4, 256), dtype=complex)
    lane_regs = np.zeros((64, 16), dtype=complex)
    for lane in range(64):
        u, l = lane >> 4, lane & 15
        for r in range(16):
            lane_regs[lane, r] = z[4 * l + u + 64 * r]
    # stage 1: radix-16 over r, twiddle W_256^(l q)
    lds = np.zeros((4, 16, 16), dtype=complex)           # [u][q][l]
    for lane in range(64):
        u, l = lane >> 4, lane & 15
        A = np.fft.fft(lane_regs[lane])                   # A[q] = sum_r z W_16^(r q)
        for q in range(16):
            lds[u, q, l] = A[q] * W(256, l * q)
    # exchange: lane (u, q') reads lds[u][q'][l2]; stage 2: radix-16 over l2 -> E_u[q' + 16 t]
    for lane in range(64):
        u, qp = lane >> 4, lane & 15
        zz = lds[u, qp, :]
        out = np.fft.fft(zz)
        for t in range(16):
            E[u, qp + 16 * t] = out[t]
    return E


def join4_untangle(E, N):
    """E[u][k1] (4 x Lq), the DIT parts of a complex FFT of length M = 4 Lq = N/2 of packed real data -> X[k], k < M
    (bins of the N-point real FFT), as the kernel does it: per (k1, Lq - k1) set, 8 values in, 8 bins out, in place."""
    Lq = E.shape[1]
    M = 4 * Lq
    X = np.zeros(M, dtype=complex)
    done = np.zeros(M, dtype=int)

    def untangle(za, zb, w):
        S = za + np.conj(zb)
        D = za - np.conj(zb)
        O = -0.5j * D
        Pk = O * w
        return 0.5 * S + Pk, np.conj(0.5 * S - Pk)

    for k1 in range(Lq // 2):
        kb = (Lq - k1) % Lq
        a = np.array([E[u, k1] * W(M, u * k1) for u in range(4)])
        b = np.array([E[u, kb] * np.conj(W(M, u * k1)) for u in range(4)])
        A = np.fft.fft(a)
        B = np.fft.fft(b)
        wu = W(N, k1)
        x0 = [None] * 4
        x1 = [None] * 4
        for t in range(4):
            x0[t], x1[t] = untangle(A[t], B[(4 - t) % 4], wu * W(8, t))
        if k1 == 0:
            # the k1 = Lq/2 family in the mirrored slots
            c = np.array([E[u, Lq // 2] * W(8, u) for u in range(4)])
            Zc = np.fft.fft(c)                                  # Z[Lq/2 + Lq u]
            sp = [None] * 4
            sp[0], sp[3] = untangle(Zc[0], Zc[3], W(16, 1))
            sp[1], sp[2] = untangle(Zc[1], Zc[2], W(16, 3))
            for t in range(4):
                x1[t] = sp[3 - t]
            kbb = Lq // 2
        else:
            kbb = kb
        for t in range(4):
            X[k1 + Lq * t] = x0[t]; done[k1 + Lq * t] += 1
            X[kbb + Lq * (3 - t)] = x1[t]; done[kbb + Lq * (3 - t)] += 1
    assert np.all(done == 1), (done.min(), done.max())
    return X


def check_single_wave():
    N = 2048
    x = rng.standard_normal(N)
    z = x[0::2] + 1j * x[1::2]
    E = wave_fft_4x256(z)
    # E_u must be the FFT of z[4 i + u]
    for u in range(4):
        assert np.allclose(E[u], np.fft.fft(z[u::4]))
    X = join4_untangle(E, N)
    ref = np.fft.fft(x)[:N // 2]
    assert np.allclose(X, ref), np.abs(X - ref).max()
    print("single wave (nfft 2048): ok, max err %.2e" % np.abs(X - ref).max())


def join8_untangle(Ec, N):
    """Ec[c][k1] (8 x 256), c = 2 u + s: radix-8 join fused with the untangle (team of 2 waves, nfft 4096)."""
    Lq = Ec.shape[1]
    R8 = Ec.shape[0]
    M = R8 * Lq
    X = np.zeros(M, dtype=complex)
    done = np.zeros(M, dtype=int)

    def untangle(za, zb, w):
        S = za + np.conj(zb); D = za - np.conj(zb)
        Pk = -0.5j * D * w
        return 0.5 * S + Pk, np.conj(0.5 * S - Pk)

    for k1 in range(Lq // 2):
        kb = (Lq - k1) % Lq
        a = np.array([Ec[c, k1] * W(M, c * k1) for c in range(R8)])
        b = np.array([Ec[c, kb] * np.conj(W(M, c * k1)) for c in range(R8)])
        A = np.fft.fft(a); B = np.fft.fft(b)
        wu = W(N, k1)
        x0 = [None] * R8; x1 = [None] * R8
        for t in range(R8):
            x0[t], x1[t] = untangle(A[t], B[(R8 - t) % R8], wu * W(2 * R8, t))
        kbb = kb
        if k1 == 0:
            c = np.array([Ec[cc, Lq // 2] * W(2 * R8, cc) for cc in range(R8)])
            Zc = np.fft.fft(c)
            sp = [None] * R8
            for t in range(R8 // 2):
                sp[t], sp[R8 - 1 - t] = untangle(Zc[t], Zc[R8 - 1 - t], W(4 * R8, 1 + 2 * t))
            for t in range(R8):
                x1[t] = sp[R8 - 1 - t]
            kbb = Lq // 2
        for t in range(R8):
            X[k1 + Lq * t] = x0[t]; done[k1 + Lq * t] += 1
            X[kbb + Lq * (R8 - 1 - t)] = x1[t]; done[kbb + Lq * (R8 - 1 - t)] += 1
    assert np.all(done == 1)
    return X


def check_team(S):
    N = 2048 * S
    M = N // 2
    x = rng.standard_normal(N)
    z = x[0::2] + 1j * x[1::2]
    Es = [wave_fft_4x256(z[s::S]) for s in range(S)]            # wave s: z[S i' + s]; its E[u] = FFT of z[S (4 i + u) + s]
    ref = np.fft.fft(x)[:M]
    if S == 2:
        Ec = np.zeros((8, 256), dtype=complex)
        for s in range(S):
            for u in range(4):
                Ec[S * u + s] = Es[s][u]
        X = join8_untangle(Ec, N)
        assert np.allclose(X, ref), np.abs(X - ref).max()
        print("team S=2 (nfft 4096), radix-8 join fused with the untangle: ok, max err %.2e" % np.abs(X - ref).max())
    # two levels: in-wave radix-4 join (plain), then the team's radix-S join fused with the untangle
    Ew = []
    for s in range(S):
        Z = np.zeros(1024, dtype=complex)
        for k1 in range(256):
            a = np.array([Es[s][u, k1] * W(1024, u * k1) for u in range(4)])
            A = np.fft.fft(a)
            for t in range(4):
                Z[k1 + 256 * t] = A[t]
        assert np.allclose(Z, np.fft.fft(z[s::S]))
        Ew.append(Z)
    print("team S=%d: in-wave radix-4 join gives the 1024-point sub-transforms: ok" % S)


if __name__ == "__main__":
    check_single_wave()
    check_team(2)
    check_team(4)
