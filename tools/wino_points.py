#!/usr/bin/env python3
"""Rounding error of Winograd F(4x4,3x3) in float32 as a function of the interpolation points (CPU simulation).

The device kernel (csrc/wino4_kernel.hip) computes  Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A  with U = G g G^T made in
float64 at model load (rounded once to float32), the input transform as float32 fma chains, the channel sum on the fp32
matrix cores (one fma per channel, sequential) and the output transform in float32.  This tool derives the three matrices
for a symmetric point set {0, +-1, +-a, inf} (or any six points) with sympy, replays that arithmetic in numpy with exactly
rounded fmas and reports the error against a float64 direct convolution, next to the error of a direct float32
convolution (sequential fma over 9 * Cin terms) and of F(2x2,3x3).

    python tools/wino_points.py [--cin 64 128 512] [--points 2 0.5 ...]
"""
import argparse
import json
from fractions import Fraction

import numpy as np
import sympy as sp


def cook_toom(points, m=4, r=3):
    """points: n - 1 finite rationals (n = m + r - 1; the last point is infinity) -> (AT m x n, G n x r, BT n x n) as
    sympy rational matrices with  y = AT [(G g) .* (BT d)]  for the correlation y[i] = sum_k d[i + k] g[k]."""
    n = m + r - 1
    pts = [sp.Rational(str(Fraction(p).limit_denominator(64))) for p in points]
    assert len(pts) == n - 1
    AT = sp.zeros(m, n)
    G = sp.zeros(n, r)
    for j, p in enumerate(pts):
        N = sp.prod([p - q for k, q in enumerate(pts) if k != j])
        for i in range(m):
            AT[i, j] = p ** i
        for k in range(r):
            G[j, k] = p ** k / N
    AT[m - 1, n - 1] = 1
    G[n - 1, r - 1] = 1
    # BT from linearity: sum_j AT[i, j] G[j, k] BT[j, l] = [l == i + k]
    BT = sp.zeros(n, n)
    for l in range(n):
        rows, rhs = [], []
        for i in range(m):
            for k in range(r):
                rows.append([AT[i, j] * G[j, k] for j in range(n)])
                rhs.append(1 if l == i + k else 0)
        sol = sp.Matrix(rows).gauss_jordan_solve(sp.Matrix(rhs))[0]
        for j in range(n):
            BT[j, l] = sol[j]
    return AT, G, BT


def normalise(AT, G, BT):
    """Scale every row of BT so that its LAST non-zero entry is +1 (the kernel's 'd[r3] comes in without a multiply' form);
    the inverse factor goes into G (free: U is computed in float64 on the host)."""
    n = BT.shape[0]
    BT, G = BT.copy(), G.copy()
    for j in range(n):
        nz = [BT[j, l] for l in range(n) if BT[j, l] != 0]
        s = nz[-1]
        BT[j, :] = BT[j, :] / s
        G[j, :] = G[j, :] * s
    return AT, G, BT


def f32(x):
    return np.asarray(x, np.float64).astype(np.float32).astype(np.float64)


def fma(a, b, c):
    """float32 fma of float32-valued float64 arrays: the product is exact in float64, one rounding of the sum to 53 bits
    and one to 24 (double rounding at 2^-29 relative: negligible for error statistics)."""
    return f32(a * b + c)


def lin32(coeffs, vals):
    """float32 evaluation of sum_i coeffs[i] * vals[i] as the fma chain the kernel uses: terms with |c| == 1 are folded
    as adds, innermost = last term."""
    terms = [(float(c), v) for c, v in zip(coeffs, vals) if c != 0]
    c, v = terms[-1]
    acc = v if c == 1 else f32(c * v)
    for c, v in reversed(terms[:-1]):
        acc = fma(c, v, acc) if abs(c) != 1 else f32(acc + c * v)
    return acc


def simulate(AT, G, BT, cin, cout=16, tiles=64, seed=0, m=4, pairwise_out=False):
    rng = np.random.default_rng(seed)
    n = AT.shape[1]
    ATf = np.array(AT.tolist(), np.float64)
    Gf = np.array(G.tolist(), np.float64)
    BTf = np.array(BT.tolist(), np.float64)
    d = f32(np.maximum(rng.normal(0, 1, (tiles, cin, n, n)), 0))            # ReLU-like activations, O(1)
    g = f32(rng.normal(0, np.sqrt(2.0 / (9 * cin)), (cout, cin, 3, 3)))     # He-normal
    # float64 truth: direct correlation
    truth = np.zeros((tiles, cout, m, m))
    for ky in range(3):
        for kx in range(3):
            truth += np.einsum('tcyx,oc->toyx', d[:, :, ky:ky + m, kx:kx + m], g[:, :, ky, kx])
    # direct float32: sequential fma over (channel, tap)
    acc = np.zeros((tiles, cout, m, m))
    for c in range(cin):
        for ky in range(3):
            for kx in range(3):
                acc = fma(d[:, None, c, ky:ky + m, kx:kx + m], g[None, :, c, ky, kx, None, None], acc)
    e_direct = acc - truth
    # Winograd: U in float64 -> float32
    U = f32(np.einsum('ik,ockl,jl->ocij', Gf, g, Gf))
    # input transform: rows first (t = BT d), then columns (V = t B), float32 fma chains
    t = np.stack([lin32(BTf[i], [d[:, :, r, :] for r in range(n)]) for i in range(n)], axis=2)          # (tiles, cin, n, n)
    V = np.stack([lin32(BTf[j], [t[:, :, :, cidx] for cidx in range(n)]) for j in range(n)], axis=3)
    M = np.zeros((tiles, cout, n, n))
    for c in range(cin):
        M = fma(V[:, None, c], U[None, :, c], M)
    # output transform: R = M A (per row), Y = A^T R
    R = np.stack([lin32(ATf[i], [M[:, :, :, j] for j in range(n)]) for i in range(m)], axis=3)           # (tiles, cout, n, m)
    Y = np.stack([lin32(ATf[i], [R[:, :, j, :] for j in range(n)]) for i in range(m)], axis=2)
    e_w = Y - truth
    scale = np.sqrt((truth ** 2).mean())
    return {'rms_direct': float(np.sqrt((e_direct ** 2).mean()) / scale), 'rms_wino': float(np.sqrt((e_w ** 2).mean()) / scale),
            'max_direct': float(np.abs(e_direct).max() / scale), 'max_wino': float(np.abs(e_w).max() / scale)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cin', type=int, nargs='+', default=[64, 256])
    ap.add_argument('--sets', nargs='+', default=['0,1,-1,2,-2', '0,1,-1,1/2,-1/2', '0,1,-1,1/2,-2', '0,1,-1,2,-1/2',
                                                  '0,1/2,-1/2,2,-2', '0,1,-1,3/2,-3/2', '0,1,-1,3/4,-3/4', '0,1/2,-1/2,1,-1',
                                                  '0,1,-1,1/2,-3', '0,3/4,-3/4,3/2,-3/2', '0,1/2,-1/2,3/2,-3/2'])
    ap.add_argument('--m', type=int, default=4)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    res = []
    for s in a.sets:
        pts = [Fraction(x) for x in s.split(',')]
        AT, G, BT = normalise(*cook_toom(pts, m=a.m))
        row = {'points': s + ',inf', 'BT': str(BT.tolist()), 'AT': str(AT.tolist())}
        for cin in a.cin:
            r = simulate(AT, G, BT, cin, m=a.m)
            row['cin%d' % cin] = r
        res.append(row)
        print(s, {k: ('%.3g / %.3g (x%.2f)' % (v['rms_wino'], v['rms_direct'], v['rms_wino'] / v['rms_direct'])) for k, v in row.items() if k.startswith('cin')}, flush=True)
    if a.out:
        json.dump(res, open(a.out, 'w'), indent=1)


if __name__ == '__main__':
    main()


# ---------------------------------------------------------------------------------------------------------------------
# The same experiment with the EXACT operation order of csrc/wino4_kernel.hip for a symmetric point set {0, +-a, +-b, inf}
# (row transform as fma chains, column transform with shared even / odd parts, one fma per input channel on the matrix
# core, output folds  s12, d12, s34, d34  as in write_R / the combine step).  Used to choose (a, b):  python -c
# "import tools.wino_points as w; print(w.kernel_order_error(5/8, 3/2, 64))"
# ---------------------------------------------------------------------------------------------------------------------
def kernel_order_error(a, b, cin, cout=16, tiles=64, seed=0):
    rng = np.random.default_rng(seed)
    d = f32(np.maximum(rng.normal(0, 1, (tiles, cin, 6, 6)), 0))
    g = f32(rng.normal(0, np.sqrt(2.0 / (9 * cin)), (cout, cin, 3, 3)))
    truth = np.zeros((tiles, cout, 4, 4))
    for ky in range(3):
        for kx in range(3):
            truth += np.einsum('tcyx,oc->toyx', d[:, :, ky:ky + 4, kx:kx + 4], g[:, :, ky, kx])
    a2, b2 = a * a, b * b
    G = np.array([[1 / (a2 * b2), 0, 0],
                  [1 / (2 * a2 * (a2 - b2)), a / (2 * a2 * (a2 - b2)), a2 / (2 * a2 * (a2 - b2))],
                  [1 / (2 * a2 * (a2 - b2)), -a / (2 * a2 * (a2 - b2)), a2 / (2 * a2 * (a2 - b2))],
                  [1 / (2 * b2 * (b2 - a2)), b / (2 * b2 * (b2 - a2)), b2 / (2 * b2 * (b2 - a2))],
                  [1 / (2 * b2 * (b2 - a2)), -b / (2 * b2 * (b2 - a2)), b2 / (2 * b2 * (b2 - a2))],
                  [0, 0, 1]])
    U = f32(np.einsum('ik,ockl,jl->ocij', G, g, G))
    rows = [(0, 2, 4, None, a2 * b2, -(a2 + b2), None), (1, 2, 3, 4, -a * b2, -b2, a), (1, 2, 3, 4, a * b2, -b2, -a),
            (1, 2, 3, 4, -a2 * b, -a2, b), (1, 2, 3, 4, a2 * b, -a2, -b), (1, 3, 5, None, a2 * b2, -(a2 + b2), None)]
    t = []
    for r0, r1, r2, r3, c0, c1, c2 in rows:                         # t[xi] = row transform, (tiles, cin, 6 columns)
        if r3 is None:
            t.append(fma(c0, d[:, :, r0, :], fma(c1, d[:, :, r1, :], d[:, :, r2, :])))
        else:
            t.append(fma(c0, d[:, :, r0, :], fma(c1, d[:, :, r1, :], fma(c2, d[:, :, r2, :], d[:, :, r3, :]))))
    t = np.stack(t, axis=2)                                          # (tiles, cin, xi, j)
    u = [t[..., j] for j in range(6)]
    ea, oa = fma(-b2, u[2], u[4]), fma(-b2, u[1], u[3])
    eb, ob = fma(-a2, u[2], u[4]), fma(-a2, u[1], u[3])
    V = np.stack([fma(a2 * b2, u[0], fma(-(a2 + b2), u[2], u[4])), fma(a, oa, ea), fma(-a, oa, ea), fma(b, ob, eb), fma(-b, ob, eb),
                  fma(a2 * b2, u[1], fma(-(a2 + b2), u[3], u[5]))], axis=-1)      # (tiles, cin, xi, nu)
    M = np.zeros((tiles, cout, 6, 6))
    for c in range(cin):
        M = fma(V[:, None, c], U[None, :, c], M)

    def fold(m):                                                     # m: list of 6 arrays -> 4 outputs
        s12, d12, s34, d34 = f32(m[1] + m[2]), f32(m[1] - m[2]), f32(m[3] + m[4]), f32(m[3] - m[4])
        r0 = f32(f32(m[0] + s12) + s34)
        r1 = fma(a, d12, f32(b * d34))
        r2 = fma(a2, s12, f32(b2 * s34))
        r3 = f32(fma(a2 * a, d12, f32(b2 * b * d34)) + m[5])
        return [r0, r1, r2, r3]
    R = np.stack(fold([M[..., j] for j in range(6)]), axis=-1)       # (tiles, cout, xi, 4)
    Y = np.stack(fold([R[:, :, j, :] for j in range(6)]), axis=2)    # (tiles, cout, 4, 4)
    scale = np.sqrt((truth ** 2).mean())
    return float(np.sqrt(((Y - truth) ** 2).mean()) / scale)
