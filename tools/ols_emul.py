"""numpy emulation of the partitioned overlap-save convolution (csrc/reverb_part.hip): Stockham radix-16 passes of a
4096-point complex FFT, real <-> complex packing with the Nyquist bin in imag(bin 0), uniform partitions."""
import numpy as np

M = 4096          # complex FFT size = block (hop) of real samples
R = 16


def expand(j, Ns, r):
    return (j // Ns) * Ns * r + (j % Ns)


def fft_stockham(z, inverse=False):
    """Govindaraju-style Stockham passes, radix 16 x 3, thread j = one butterfly per pass."""
    n = z.shape[-1]
    assert n == M
    sign = +1.0 if inverse else -1.0
    a = z.astype(np.complex64)
    Ns = 1
    j = np.arange(n // R)
    dft = np.exp(sign * 2j * np.pi * np.outer(np.arange(R), np.arange(R)) / R).astype(np.complex64)
    while Ns < n:
        k = j % Ns
        v = np.stack([a[..., j + r * (n // R)] for r in range(R)], axis=-1)               # [..., 256, 16]
        tw = np.exp(sign * 2j * np.pi * np.outer(k, np.arange(R)) / (Ns * R)).astype(np.complex64)
        v = v * tw
        v = v @ dft.T
        out = np.empty_like(a)
        idx = expand(j, Ns, R)
        for r in range(R):
            out[..., idx + r * Ns] = v[..., r]
        a = out
        Ns *= R
    return a


def rfft_packed(w):
    """w: [..., 2M] real -> [..., M] complex: bins 0..M-1, Nyquist in imag(bin 0)."""
    z = (w[..., 0::2] + 1j * w[..., 1::2]).astype(np.complex64)
    Z = fft_stockham(z)
    k = np.arange(M)
    Zr = np.conj(Z[..., (-k) % M])                     # conj(Z[M - k]), Z[M] = Z[0]
    E = 0.5 * (Z + Zr)
    O = -0.5j * (Z - Zr)
    W = E + np.exp(-2j * np.pi * k / (2 * M)).astype(np.complex64) * O
    nyq = (E[..., 0] - O[..., 0]).real                 # k = M: E[0] + e^{-i pi} O[0]
    W = W.astype(np.complex64)
    W[..., 0] = W[..., 0].real + 1j * nyq
    return W


def irfft_packed(Y):
    """inverse of rfft_packed up to the factor 2M (returns 2M * irfft): [..., M] packed -> [..., 2M] real."""
    k = np.arange(M)
    Yf = Y.copy()
    dc, nyq = Y[..., 0].real, Y[..., 0].imag
    Yf[..., 0] = dc
    Ym = np.conj(Yf[..., (-k) % M])
    Ym[..., 0] = nyq                                    # conj(Y[M]) = Nyquist (real)
    E = 0.5 * (Yf + Ym)
    O = 0.5 * (Yf - Ym) * np.exp(2j * np.pi * k / (2 * M)).astype(np.complex64)
    Zp = (E + 1j * O).astype(np.complex64)
    z = fft_stockham(Zp, inverse=True)
    w = np.empty(Y.shape[:-1] + (2 * M,), np.float32)
    w[..., 0::2] = z.real
    w[..., 1::2] = z.imag
    return w * 2.0                                      # = 2M * irfft(Y)


def mul_packed(X, H):
    out = X * H
    out[..., 0] = X[..., 0].real * H[..., 0].real + 1j * (X[..., 0].imag * H[..., 0].imag)
    return out


def ols_convolve(x, h, start, out_len, mask_dry=False):
    N, L = len(x), len(h)
    h = h.copy()
    if mask_dry:
        h[0] = 0
    Pn = -(-L // M)
    nb_out = -(-(start + out_len) // M)
    i0 = start // M
    jmax = min(nb_out - 1, -(-N // M))                  # X_j is zero beyond
    def win(sig, lo):                                   # sig[lo : lo + 2M], zero outside
        w = np.zeros(2 * M, np.float32)
        a, b = max(lo, 0), min(lo + 2 * M, len(sig))
        if b > a:
            w[a - lo:b - lo] = sig[a:b]
        return w
    X = np.stack([rfft_packed(win(x, (j - 1) * M)) for j in range(jmax + 1)])
    Hs = []
    for p in range(Pn):
        w = np.zeros(2 * M, np.float32)
        seg = h[p * M:(p + 1) * M]
        w[:len(seg)] = seg
        Hs.append(rfft_packed(w))
    Hs = np.stack(Hs)
    out = np.zeros(out_len, np.float32)
    for i in range(i0, nb_out):
        acc = np.zeros(M, np.complex64)
        for p in range(max(0, i - jmax), min(i, Pn - 1) + 1):
            acc = acc + mul_packed(X[i - p], Hs[p])
        w = irfft_packed(acc) / (2.0 * M)
        y = w[M:]                                       # y[i M .. (i + 1) M)
        lo, hi = max(i * M, start), min((i + 1) * M, start + out_len)
        if hi > lo:
            out[lo - start:hi - start] = y[lo - i * M:hi - i * M]
    return out


if __name__ == '__main__':
    rng = np.random.default_rng(0)
    z = (rng.normal(size=M) + 1j * rng.normal(size=M)).astype(np.complex64)
    print('fft err', np.abs(fft_stockham(z) - np.fft.fft(z)).max() / np.abs(np.fft.fft(z)).max())
    print('ifft err', np.abs(fft_stockham(z, True) - np.fft.ifft(z) * M).max() / M)
    w = rng.normal(size=2 * M).astype(np.float32)
    ref = np.fft.rfft(w)
    got = rfft_packed(w)
    print('rfft err', np.abs(got[1:] - ref[1:M]).max(), abs(got[0].real - ref[0].real), abs(got[0].imag - ref[M].real))
    print('irfft err', np.abs(irfft_packed(got) / (2 * M) - w).max())
    for (N, L, start, out_len, md) in [(20000, 9000, 0, 20000, True), (9000, 20000, 0, 9000, False), (5000, 3000, 1498, 5000, False),
                                       (12000, 4096, 0, 12000 + 4095, False), (4096, 8192, 0, 4096, True)]:
        x = rng.normal(size=N).astype(np.float32)
        h = (rng.normal(size=L) * np.exp(-np.arange(L) / (0.3 * L))).astype(np.float32)
        hh = h.copy()
        if md:
            hh[0] = 0
        ref = np.convolve(x.astype(np.float64), hh.astype(np.float64))[start:start + out_len]
        got = ols_convolve(x, h, start, out_len, md)
        print(N, L, start, out_len, 'err', np.abs(got - ref).max() / np.abs(ref).max())
