"""Address-level emulation of the windowed time-varying FIR (csrc/noise.hip, fir_win_core) in numpy:
checks the geometry (image / noise layout in LDS, per-lane pointers, frame wraps) against a direct evaluation."""
import sys
import numpy as np


def fdiv(a, b):
    return a // b            # python floor division


def cdiv(a, b):
    return -((-a) // b)


def geometry(U, Lw, delay, OPL, D=32):
    assert U % 4 == 0 and OPL % 4 == 0 and U % OPL == 0
    bpf = U // 4
    NP = U // OPL
    AQ = (OPL + 3 + 3) // 4
    q_hi0 = fdiv(delay + OPL - 1, 4)
    q_lo0 = cdiv(delay - Lw - 2, 4)
    nsteps = q_hi0 - q_lo0 + 1
    RL = -fdiv(q_lo0, bpf)
    RH = fdiv(q_hi0 + (OPL // 4) * (NP - 1), bpf)
    W = D - RL - RH
    padl = OPL + 3 - delay + 4 * q_hi0
    while (padl + delay - 3) % 4 != 0:
        padl += 1
    gap = max(padl, delay + OPL - 3 - 4 * q_lo0 + 4 * AQ - Lw)
    gs = Lw + gap
    while gs % 4 != 0 or (gs // 4) % 2 == 0:
        gs += 1
    return dict(bpf=bpf, NP=NP, AQ=AQ, q_hi0=q_hi0, nsteps=nsteps, RL=RL, RH=RH, W=W, padl=padl, gs=gs, D=D)


def emulate(U, K, delay, OPL, T, seed=0, D=32):
    Lw = 2 * (K - 1)
    g = geometry(U, Lw, delay, OPL, D)
    bpf, NP, AQ, W, RL, gs, padl, nsteps = g['bpf'], g['NP'], g['AQ'], g['W'], g['RL'], g['gs'], g['padl'], g['nsteps']
    assert W >= 1
    rng = np.random.default_rng(seed)
    N = T * U
    x = rng.uniform(-1, 1, N)
    h = rng.normal(0, 1, [T, Lw])
    # direct
    ref = np.zeros(N)
    for j in range(N):
        f = j // U
        lo = max(j - delay, 0)
        hi = min(j - delay + Lw - 1, N - 1)
        if hi >= lo:
            n = np.arange(lo, hi + 1)
            ref[n] += x[j] * h[f, n + delay - j]
    out = np.full(N, np.nan)
    wpr = cdiv(T, W)
    RS = AQ + 1
    for win in range(wpr):
        F0 = win * W
        G = np.zeros(D * gs + 32)
        for s in range(D):
            f = min(max(F0 - RL + s, 0), T - 1)
            G[s * gs + padl: s * gs + padl + Lw] = h[f]
        Gnz = G != 0
        s_chk = True
        Xs = np.zeros((bpf + 1) * 4 * D)
        for b in range(bpf * D):
            jb = bpf * (F0 - RL) + b
            if 0 <= jb < N // 4:
                o = 4 * (b + b // bpf)
                Xs[o:o + 4] = x[4 * jb: 4 * jb + 4]
        for ph in range(NP):
            for fr in range(W):
                if F0 + fr >= T:
                    continue
                tb = OPL * ph + delay
                phA = ph & ~1
                qA = g['q_hi0'] + (OPL // 4) * phA
                q_hi, q_lo = qA + OPL // 4, qA - nsteps + 1
                gl = (fr + RL) * gs + padl + tb - 3
                xl = 4 * (bpf + 1) * (fr + RL)
                acc = np.zeros(OPL)
                for gg in range(fdiv(q_hi, bpf), fdiv(q_lo, bpf) - 1, -1):
                    fq0 = gg * bpf
                    qs = min(q_hi, fq0 + bpf - 1)
                    ln = qs - max(q_lo, fq0) + 1
                    gp = gl + gg * gs - 4 * qs
                    xp = xl + 4 * ((bpf + 1) * gg + qs - fq0)
                    assert gp % 4 == 0
                    for i in range(ln):
                        a0 = gp + 4 * i
                        x0 = xp - 4 * i
                        assert a0 >= (padl & ~3) and a0 + 4 * AQ <= D * gs, (a0, len(G))
                        assert x0 >= 0 and x0 + 4 <= len(Xs)
                        tp = G[a0: a0 + 4 * AQ]
                        if s_chk: assert not Gnz[a0: a0 + 4 * AQ][[k for k in range(4 * AQ) if not (0 <= a0 + k - ((fr + RL + gg) * gs + padl) < Lw)]].any()
                        xs = Xs[x0: x0 + 4]
                        for d in range(4):
                            for e in range(OPL):
                                acc[e] += xs[d] * tp[e - d + 3]
                n0 = U * (F0 + fr) + OPL * ph
                out[n0:n0 + OPL] = acc
    err = np.abs(out - ref).max()
    return err, g


if __name__ == '__main__':
    cases = [(96, 96, None, 12, 70), (96, 96, 95, 12, 61), (96, 64, None, 12, 35), (128, 32, None, 16, 40),
             (192, 96, None, 12, 17), (192, 96, None, 24, 17), (64, 64, None, 8, 45), (64, 64, None, 16, 45),
             (96, 96, 0, 12, 33), (96, 96, 40, 12, 33), (32, 32, None, 4, 100), (128, 96, None, 16, 20)]
    for U, K, delay, OPL, T in cases:
        Lw = 2 * (K - 1)
        d = (Lw - 1) // 2 - 1 if delay is None else delay
        err, g = emulate(U, K, d, OPL, T)
        lds = (g['D'] * g['gs'] + 16 + (g['bpf'] + 1) * 4 * g['D'] + g['D'] * (K + 4)) * 4
        print(f'U={U} K={K} delay={d} OPL={OPL}: err {err:.2e}  W={g["W"]} RL={g["RL"]} RH={g["RH"]} nsteps={g["nsteps"]} gs={g["gs"]} '
              f'padl={g["padl"]} lds={lds}')
        assert err < 1e-9
