#!/usr/bin/env python3
"""Error budget of Winograd F(4x4,3x3) in fp32 (conv_wino4.hip) per choice of interpolation points: numpy
emulation of the transforms and the sequential fp32 accumulation over the input channels on the ResNet's layer
shapes against an fp64 convolution, next to direct fp32 and F(2x2,3x3).  Output of `python wino4_points.py`
(max error / output scale, Cin = 64 / 256 / 512):
    direct fp32                 7.7e-07  1.9e-06  3.0e-06
    F(2x2,3x3)                  2.9e-07  6.5e-07  6.7e-07
    F(4x4,3x3) (0,1,-1,2,-2)    5.7e-06  8.3e-06  4.2e-06     (textbook points)
    F(4x4,3x3) (0,1,-1,2,-1/2)  1.9e-06  4.0e-06  3.5e-06     (used by the kernel)
`python wino4_points.py search` ranks 120 point sets; (0,1,-1,2,-1/2) is the best of them."""
import numpy as np, sys
from fractions import Fraction as Fr
def cook_toom(points, m, r):
    # returns AT (m x n), G (n x r), BT (n x n) for F(m, r) with n = m + r - 1, last point = infinity
    n = m + r - 1
    pts = [Fr(p) for p in points]  # n-1 finite points
    # Build via Vandermonde method (wincnn style)
    import sympy as sp
    a = [sp.Rational(p.numerator, p.denominator) for p in pts]
    x = sp.symbols('x')
    def At(a, m, n):
        return sp.Matrix(m, n, lambda i, j: a[j]**i)
    def A(a, m, n):
        M = At(a, m-1, n).T
        M = M.row_insert(m-1, sp.Matrix(1, n, lambda i, j: 1 if j == n-1 else 0)) if False else M
        return M
    # wincnn
    def T(a, n):
        return sp.Matrix(sp.Matrix.eye(n).col_insert(n, sp.Matrix(n, 1, lambda i, j: -a[i]**n)))
    def Lx(a, n):
        f = sp.prod([(x - a[i]) for i in range(n)]) if False else None
    from functools import reduce
    import operator
    def fdiag(a, n):
        f = [reduce(operator.mul, [(a[i]-a[j]) for j in range(n) if j != i], 1) for i in range(n)]
        return f
    al = len(a)
    f = fdiag(a, al)
    # AT: m x n
    AT = sp.Matrix(m, n, lambda i, j: (a[j]**i if j < al else (1 if i == m-1 else 0)))
    G = sp.Matrix(n, r, lambda i, j: (a[i]**j / f[i] if i < al else (1 if j == r-1 else 0)))
    # BT from polynomial: rows = coefficients of L_i(x) * f[i]... use wincnn formula
    def poly_coeffs(p, deg):
        P = sp.Poly(sp.expand(p), x); c = P.all_coeffs()[::-1]
        return c + [0]*(deg+1-len(c))
    M = reduce(operator.mul, [(x - a[i]) for i in range(al)], 1)
    rows = []
    for i in range(al):
        Li = sp.cancel(M / (x - a[i]))
        rows.append(poly_coeffs(Li, al))
    rows.append(poly_coeffs(M, al))
    BT = sp.Matrix(rows)
    # scale: BT rows for finite points correspond to f[i]*L_i normalised... verify numerically below
    return np.array(AT.tolist(), dtype=np.float64), np.array(G.tolist(), dtype=np.float64), np.array(BT.tolist(), dtype=np.float64)

def check(AT, G, BT, m, r):
    rng = np.random.default_rng(0)
    d = rng.standard_normal(m + r - 1); g = rng.standard_normal(r)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])
    return np.abs(y - ref).max()

def conv_ref(x, w):  # x (C,H,W) f64, w (M,C,3,3); pad 1
    C, H, W = x.shape; M = w.shape[0]
    xp = np.zeros((C, H + 2, W + 2)); xp[:, 1:-1, 1:-1] = x
    y = np.zeros((M, H, W))
    for a in range(3):
        for b in range(3):
            y += np.einsum('mc,chw->mhw', w[:, :, a, b], xp[:, a:a + H, b:b + W])
    return y

def conv_wino(x, w, AT, G, BT, m, dt=np.float32, kchain=True):
    # 2D F(m x m, 3x3) in dtype dt, accumulation over C as sequential fma-like (np sum in dt, chunked)
    C, H, W = x.shape; M = w.shape[0]; n = m + 2
    AT_, G_, BT_ = AT.astype(dt), G.astype(dt), BT.astype(dt)
    TH, TW = -(-H // m), -(-W // m)
    xp = np.zeros((C, TH * m + 2, TW * m + 2), dt); xp[:, 1:H + 1, 1:W + 1] = x.astype(dt)
    U = np.einsum('ia,mcab,jb->mcij', G_, w.astype(dt), G_).astype(dt)  # (M,C,n,n)
    # tiles
    V = np.zeros((C, TH, TW, n, n), dt)
    for i in range(TH):
        for j in range(TW):
            d = xp[:, i * m:i * m + n, j * m:j * m + n]
            t = np.einsum('ia,cab->cib', BT_, d).astype(dt)
            V[:, i, j] = np.einsum('cib,jb->cij', t, BT_).astype(dt)
    # sequential accumulation over c in dt
    Mm = np.zeros((M, TH, TW, n, n), dt)
    for c in range(C):
        Mm = (Mm + U[:, c][:, None, None] * V[c][None]).astype(dt)
    t = np.einsum('ia,mhwab->mhwib', AT_, Mm).astype(dt)
    Y = np.einsum('mhwib,jb->mhwij', t, AT_).astype(dt)
    y = Y.transpose(0, 1, 3, 2, 4).reshape(M, TH * m, TW * m)[:, :H, :W]
    return y

def conv_direct32(x, w):
    C, H, W = x.shape; M = w.shape[0]
    xp = np.zeros((C, H + 2, W + 2), np.float32); xp[:, 1:-1, 1:-1] = x
    y = np.zeros((M, H, W), np.float32)
    for c in range(C):
        for a in range(3):
            for b in range(3):
                y = (y + w[:, c, a, b].astype(np.float32)[:, None, None] * xp[c, a:a + H, b:b + W][None]).astype(np.float32)
    return y

if __name__ == "__main__":
    sets = {"F2 std": ([0, 1, -1], 2), "F4 std(0,1,-1,2,-2)": ([0, 1, -1, 2, -2], 4),
            "F4 (0,1,-1,1/2,-1/2)": ([0, 1, -1, Fr(1, 2), Fr(-1, 2)], 4),
            "F4 (0,1,-1,1/2,-2)": ([0, 1, -1, Fr(1, 2), -2], 4),
            "F4 (0,1,-1,2,-1/2)": ([0, 1, -1, 2, Fr(-1, 2)], 4)}
    rng = np.random.default_rng(1)
    for C, H, W, M in [(64, 18, 64, 32), (256, 5, 48, 32), (512, 3, 48, 16)]:
        x = rng.standard_normal((C, H, W)); w = rng.standard_normal((M, C, 3, 3)) * np.sqrt(2.0 / (9 * M))
        ref = conv_ref(x, w); scale = np.abs(ref).max()
        e = np.abs(conv_direct32(x.astype(np.float32), w.astype(np.float32)) - ref).max() / scale
        print("C=%d H=%d: direct fp32 err/scale %.2e" % (C, H, e))
        for name, (pts, m) in sets.items():
            AT, G, BT = cook_toom(pts, m, 3)
            ok = check(AT, G, BT, m, 3)
            y = conv_wino(x, w, AT, G, BT, m)
            err = np.abs(y - ref)
            print("   %-24s selfcheck %.1e  max err/scale %.2e  rms err/rms %.2e" % (name, ok, err.max() / scale, np.sqrt((err**2).mean()) / np.sqrt((ref**2).mean())))

    if len(sys.argv) > 1 and sys.argv[1] == "search":
        import itertools
        cands = [Fr(1,2), Fr(-1,2), 2, -2, Fr(3,2), Fr(-3,2), Fr(2,3), Fr(-2,3), 3, Fr(1,3), Fr(-1,3), -3, Fr(3,4), Fr(-3,4), Fr(4,3), Fr(-4,3)]
        data = []
        for C, H, W, M in [(64, 18, 48, 16), (256, 5, 48, 16)]:
            x = rng.standard_normal((C, H, W)); w = rng.standard_normal((M, C, 3, 3)) * np.sqrt(2.0 / (9 * M))
            data.append((x, w, conv_ref(x, w)))
        res = []
        for a, b in itertools.combinations(cands, 2):
            pts = [0, 1, -1, a, b]
            AT, G, BT = cook_toom(pts, 4, 3)
            errs = [np.abs(conv_wino(x, w, AT, G, BT, 4) - ref).max() / np.abs(ref).max() for x, w, ref in data]
            res.append((max(errs), errs, pts))
        for r in sorted(res, key=lambda r: r[0])[:8]:
            print(r[2], ["%.2e" % e for e in r[1]])
