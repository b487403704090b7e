"""Pure-Python-int BN254 G1 arithmetic used ONLY to generate golden vectors.

This is an independent big-integer statement of the group law (affine chord /
tangent with modular inverse) -- it shares no code with `oracle/` (C, 4x64-bit
Montgomery) or with the HIP kernels (8x32-bit Montgomery, XYZZ), so agreement
between the three is meaningful.  Curve: y^2 = x^3 + 3 over Fq, generator
(1, 2), cofactor 1 (reference: SH/constants.metal:121-174 BN254_ONE_{X,Y,Z} =
1,2,1).

Test infrastructure: never imported by the product path.
"""

P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R_ORDER = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
B = 3
G = (1, 2)
INF = None  # point at infinity


def is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - B) % P == 0


def neg(pt):
    if pt is None:
        return None
    return (pt[0], (-pt[1]) % P)


def add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = (3 * x1 * x1) * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


def _jac_dbl(X, Y, Z):
    if Z == 0 or Y == 0:
        return (1, 1, 0)
    A = X * X % P
    Bv = Y * Y % P
    C = Bv * Bv % P
    D = 2 * ((X + Bv) * (X + Bv) - A - C) % P
    E = 3 * A % P
    F = E * E % P
    X3 = (F - 2 * D) % P
    Y3 = (E * (D - X3) - 8 * C) % P
    Z3 = 2 * Y * Z % P
    return (X3, Y3, Z3)


def _jac_add_affine(X1, Y1, Z1, x2, y2):
    if Z1 == 0:
        return (x2, y2, 1)
    Z1Z1 = Z1 * Z1 % P
    U2 = x2 * Z1Z1 % P
    S2 = y2 * Z1 * Z1Z1 % P
    H = (U2 - X1) % P
    r = (S2 - Y1) % P
    if H == 0:
        if r == 0:
            return _jac_dbl(X1, Y1, Z1)
        return (1, 1, 0)
    HH = H * H % P
    HHH = H * HH % P
    V = X1 * HH % P
    X3 = (r * r - HHH - 2 * V) % P
    Y3 = (r * (V - X3) - Y1 * HHH) % P
    Z3 = Z1 * H % P
    return (X3, Y3, Z3)


def _jac_to_affine(X, Y, Z):
    if Z == 0:
        return None
    zi = pow(Z, -1, P)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def mul(k, pt):
    """k * pt by left-to-right double-and-add in Jacobian coordinates."""
    k %= R_ORDER
    if pt is None or k == 0:
        return None
    acc = (1, 1, 0)
    for bit in bin(k)[2:]:
        acc = _jac_dbl(*acc)
        if bit == "1":
            acc = _jac_add_affine(*acc, pt[0], pt[1])
    return _jac_to_affine(*acc)


def msm_naive(points, scalars):
    """sum_i scalars[i] * points[i], one scalar multiplication per term."""
    acc = None
    for pt, s in zip(points, scalars):
        acc = add(acc, mul(s, pt))
    return acc


def to_words(v, n=8):
    """little-endian 32-bit words of a non-negative integer"""
    return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def from_words(ws):
    v = 0
    for i, w in enumerate(ws):
        v |= int(w) << (32 * i)
    return v
