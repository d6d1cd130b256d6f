#!/usr/bin/env python3
"""Derive the fast-path scale factors and the rigorous guard band for the fused kernel.

The kernel computes a fast (AAN, true-cosine, FMA-friendly) DCT and only trusts
round(z) when z is farther than delta_k from every half-integer; otherwise the
coefficient is recomputed in the reference's exact float32 order
(natural_c/src/core/dct.c:63-96, quantization.c:34-36).  This script derives, in double
precision, for every coefficient k = u*8+v:

  G_k      AAN output scale:  aan_k = G_k * sum_xy p * cos * cos   (true cosines)
  K_k      the reference's float scale fl(fl(0.25f*C[u])*C[v])
  E_ref_k  bound on |s_ref - T_lut|   (reference's float32 evaluation error; s units)
  E_lut_k  bound on |T_lut - T_true|  (six-decimal LUT vs true cosines; s units)
  E_aan_k  bound on |aan_float - aan_exact| (fast path float32 error; aan units)

and prints delta_k (z units) for a quantisation table.  The C++ host code
(csrc/quant_consts.cpp) implements the same derivation; tests compare the two.
"""
import math
import numpy as np

U = 2.0 ** -24  # float32 unit roundoff

COS_LUT = np.array([  # [x][u], natural_c/src/core/dct.c:9-18
    [1.000000, 0.980785, 0.923880, 0.831470, 0.707107, 0.555570, 0.382683, 0.195090],
    [1.000000, 0.831470, 0.382683, -0.195090, -0.707107, -0.980785, -0.923880, -0.555570],
    [1.000000, 0.555570, -0.382683, -0.980785, -0.707107, 0.195090, 0.923880, 0.831470],
    [1.000000, 0.195090, -0.923880, -0.555570, 0.707107, 0.831470, -0.382683, -0.980785],
    [1.000000, -0.195090, -0.923880, 0.555570, 0.707107, -0.831470, -0.382684, 0.980785],
    [1.000000, -0.555570, -0.382684, 0.980785, -0.707107, -0.195090, 0.923880, -0.831470],
    [1.000000, -0.831470, 0.382684, 0.195091, -0.707107, 0.980785, -0.923879, 0.555570],
    [1.000000, -0.980785, 0.923880, -0.831470, 0.707107, -0.555570, 0.382684, -0.195090],
], dtype=np.float32).astype(np.float64)   # exact values of the float32 literals

TRUE_COS = np.array([[math.cos((2 * x + 1) * u * math.pi / 16) for u in range(8)] for x in range(8)])
C_REF = np.array([np.float32(0.707107)] + [np.float32(1.0)] * 7, dtype=np.float32)

BASE_Q = [16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
          14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
          49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99]

PMAX = 128.0


# ---- the AAN flow graph, expressed on "tracked" values ---------------------------------
class Tr:
    """A linear functional of the 64 inputs with a rigorous first-order error bound."""
    __slots__ = ("f", "e")

    def __init__(self, f, e=0.0):
        self.f = f   # coefficient vector (64,)
        self.e = e   # abs error bound of the float32 value vs the exact functional

    def bound(self):
        return PMAX * np.abs(self.f).sum()

    def _new(self, f, e_in):
        t = Tr(f, 0.0)
        t.e = e_in + U * (t.bound() + e_in) * 1.0000001
        return t

    def __add__(self, o):
        return self._new(self.f + o.f, self.e + o.e)

    def __sub__(self, o):
        return self._new(self.f - o.f, self.e + o.e)

    def scale(self, c):
        # constant rounded to float32 (rel err U) then product rounded (rel err U)
        t = Tr(self.f * c, 0.0)
        e_in = abs(c) * self.e
        t.e = e_in + 2.0 * U * (t.bound() + e_in) * 1.0000001
        return t


A1 = math.sqrt(0.5)                    # 0.707106781
A2 = math.cos(3 * math.pi / 8) * math.sqrt(2)   # 0.541196100
A4 = math.cos(math.pi / 8) * math.sqrt(2)       # 1.306562965
A5 = math.cos(3 * math.pi / 8)                  # 0.382683433


def aan8(d):
    t0 = d[0] + d[7]; t7 = d[0] - d[7]
    t1 = d[1] + d[6]; t6 = d[1] - d[6]
    t2 = d[2] + d[5]; t5 = d[2] - d[5]
    t3 = d[3] + d[4]; t4 = d[3] - d[4]
    t10 = t0 + t3; t13 = t0 - t3
    t11 = t1 + t2; t12 = t1 - t2
    o0 = t10 + t11; o4 = t10 - t11
    z1 = (t12 + t13).scale(A1)
    o2 = t13 + z1; o6 = t13 - z1
    t10 = t4 + t5; t11 = t5 + t6; t12 = t6 + t7
    z5 = (t10 - t12).scale(A5)
    z2 = t10.scale(A2) + z5
    z4 = t12.scale(A4) + z5
    z3 = t11.scale(A1)
    z11 = t7 + z3; z13 = t7 - z3
    o5 = z13 + z2; o3 = z13 - z2
    o1 = z11 + z4; o7 = z11 - z4
    return [o0, o1, o2, o3, o4, o5, o6, o7]


def aan2d_tracked():
    d = [Tr(np.eye(64)[i]) for i in range(64)]          # d[r*8+c], exact inputs
    for r in range(8):                                   # row pass: over c -> v
        o = aan8([d[r * 8 + c] for c in range(8)])
        for c in range(8):
            d[r * 8 + c] = o[c]
    for c in range(8):                                   # column pass: over r -> u
        o = aan8([d[r * 8 + c] for r in range(8)])
        for r in range(8):
            d[r * 8 + c] = o[r]
    return d                                             # d[u*8+v]


def derive(qtable):
    tracked = aan2d_tracked()
    out = []
    for k in range(64):
        u, v = divmod(k, 8)
        # exact functionals
        t_true = np.array([TRUE_COS[x][u] * TRUE_COS[y][v] for x in range(8) for y in range(8)])
        t_lut = np.array([COS_LUT[x][u] * COS_LUT[y][v] for x in range(8) for y in range(8)])
        f = tracked[k].f
        # G_k: aan functional = G * true functional (check proportionality)
        i0 = np.argmax(np.abs(t_true))
        G = f[i0] / t_true[i0]
        assert np.allclose(f, G * t_true, rtol=0, atol=1e-12), (k, np.abs(f - G * t_true).max())
        K = float(np.float32(np.float32(np.float32(0.25) * C_REF[u]) * C_REF[v]))
        w = np.abs(t_lut)
        # reference evaluation error (products: 2 roundings; adds: sequential)
        prod = 2.0 * U * PMAX * w.sum()
        csum = np.cumsum(w)
        adds = U * PMAX * csum[1:].sum()
        e_ref = (prod + adds) * 1.001
        e_lut = PMAX * np.abs(t_lut - t_true).sum()
        e_aan = tracked[k].e
        q = float(qtable[k])
        zmax = K * PMAX * w.sum() / q
        # z_fast = fl(aan*M + cb) - cb ; M = fl(K/(q*G))
        delta = (K / q) * (e_ref + e_lut) + (K / (q * abs(G))) * e_aan + 4.0 * U * (zmax + 1.0)
        out.append(dict(k=k, G=G, K=K, e_ref=e_ref, e_lut=e_lut, e_aan=e_aan, zmax=zmax, delta=delta,
                        M=K / (q * G)))
    return out


if __name__ == "__main__":
    res = derive(BASE_Q)
    print(" k  u v    G        K         e_ref    e_lut    e_aan(aan)  zmax    delta_z")
    for r in res:
        u, v = divmod(r["k"], 8)
        print(f"{r['k']:2d}  {u} {v} {r['G']:8.4f} {r['K']:.7f} {r['e_ref']:.5f} {r['e_lut']:.5f} "
              f"{r['e_aan']:.5f}   {r['zmax']:7.2f} {r['delta']:.3e}")
    print("max delta (AC):", max(r["delta"] for r in res[1:]))
