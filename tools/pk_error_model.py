#!/usr/bin/env python3
"""Numerical model of the mixed-precision lean step of `rollout_pk_kernel` (fp32-storage mode), run on the CPU
against the fp64 oracle BEFORE the kernel was written (task 1a):

    python tools/pk_error_model.py [c4|c3|fast|big]

Every per-sample quantity is carried as its DEVIATION from the nominal (eps = 0) trajectory; the float32 operations
below are exactly the kernel's (same order, fma where the kernel has one -- emulated as round32(a*b + c) in float64,
which is exact up to a double rounding); the three running sums (heading deviation, position deviation, cost
prefix) stay float64.  Prints max |V - V_oracle| over all (t, k), over the 1000 best samples, and the worst ratio of
the error to the tests' stated fp32 tolerance 3e-7 * max(1, max_t |V - V_nominal|) per sample."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402

f32, f64 = np.float32, np.float64
UMAX, R, WB = 6.35492, 0.033, 0.16


def fma32(a, b, c):
    return (a.astype(f64) * np.asarray(b, f64) + np.asarray(c, f64)).astype(f32)


def nominal(state, goal, u0, T, dt, q=1e3, lam=1e-3, sig=0.9):
    kth, rhalf = R / WB, R / 2
    hk = 0.5 * kth * dt
    ac = np.clip(u0, -UMAX, UMAX)
    pn = hk * ac
    phin = pn[1] - pn[0]
    th = state[2] + np.concatenate([[0.0], np.cumsum(2 * phin)])[:-1]
    f = np.sqrt(0.5 * q)
    rho = f * (dt * rhalf / 6.0) / hk
    c1n, s1n = np.cos(th + phin), np.sin(th + phin)
    Wn = 4 + 2 * np.cos(phin)
    Pn = pn[0] + pn[1]
    Xn = f * (state[0] - goal[0]) + np.cumsum(rho * Pn * Wn * c1n)
    Yn = f * (state[1] - goal[1]) + np.cumsum(rho * Pn * Wn * s1n)
    return dict(hk=hk, ac=ac, a=u0, phin=phin, f=f, rho=rho, c1n=c1n, s1n=s1n, Wn=Wn, Pn=Pn, Xn=Xn, Yn=Yn,
                thT=th[-1] + 2 * phin[-1], w=lam * sig * u0)


def run(state, goal, u0, eps, T, dt, lam=1e-3, sig=0.9, p1=(1e3, 1e3, 1e3)):
    n = nominal(state, goal, u0, T, dt, lam=lam, sig=sig)
    K = eps.shape[2]
    th = np.zeros(K, f64)
    dX = np.zeros(K, f64)
    dY = np.zeros(K, f64)
    pre = np.zeros(K, f64)
    dP = np.zeros((T, K), f64)
    hk = f32(n["hk"])
    almax = 0.0
    for t in range(T):
        dP[t] = pre.astype(f32)
        e0, e1 = eps[t, 0].astype(f32), eps[t, 1].astype(f32)
        d = [f32(n["hk"] * (n["a"][i, t] - n["ac"][i, t])) for i in range(2)]
        lo = [f32(n["hk"] * (-UMAX - n["ac"][i, t])) for i in range(2)]
        hi = [f32(n["hk"] * (UMAX - n["ac"][i, t])) for i in range(2)]
        dp0 = np.clip(fma32(e0, hk, d[0]), lo[0], hi[0])
        dp1 = np.clip(fma32(e1, hk, d[1]), lo[1], hi[1])
        dphi = dp1 - dp0
        dPs = dp0 + dp1
        al = th.astype(f32) + dphi
        almax = max(almax, float(np.abs(al).max()))
        th = 2.0 * dphi.astype(f64) + th
        z = al * al
        S = al * fma32(z, fma32(z, fma32(z, f32(-1 / 5040), f32(1 / 120)), f32(-1 / 6)), f32(1))
        Cm = z * fma32(z, fma32(z, fma32(z, f32(1 / 40320), f32(-1 / 720)), f32(1 / 24)), f32(-0.5))
        rho = n["rho"]
        A1, Cn = f32(-2 * np.sin(n["phin"][t]) * rho), f32(-np.cos(n["phin"][t]) * rho)
        Wn, Pn = f32(n["Wn"][t] * rho), f32(n["Pn"][t])
        dW = dphi * fma32(dphi, Cn, A1)
        P = dPs + Pn
        Aq = P * dW
        dG = fma32(dPs, Wn, Aq)
        G = fma32(P, Wn, Aq)
        a = fma32(G, Cm, dG)
        b = G * S
        c1n, s1n = f32(n["c1n"][t]), f32(n["s1n"][t])
        ix = fma32(a, c1n, -(s1n * b))
        iy = fma32(a, s1n, c1n * b)
        dX = dX + ix.astype(f64)
        dY = dY + iy.astype(f64)
        X2, Y2 = 2 * n["Xn"][t], 2 * n["Yn"][t]
        pre = dX * (X2 + dX) + pre
        pre = dY * (Y2 + dY) + pre
        dcn = fma32(e0, f32(n["w"][0, t]), e1 * f32(n["w"][1, t]))
        pre = pre + dcn.astype(f64)
    thT = n["thT"] + th
    wrap = lambda a: a - (np.ceil((a + np.pi) / (2 * np.pi)) - 1.0) * 2 * np.pi
    thw, thn = wrap(thT), wrap(n["thT"])
    f2 = n["f"] ** 2
    term = p1[0] / f2 * dX * (2 * n["Xn"][-1] + dX) + p1[1] / f2 * dY * (2 * n["Yn"][-1] + dY) \
        + p1[2] * ((thw - goal[2]) ** 2 - (thn - goal[2]) ** 2)
    return dP, (pre + term).astype(f32).astype(f64), almax


CASES = {
    # name: (K, T, state, goal, nominal)
    "c4": (200000, 50, [0.0, 0, 0], [0.0, -1, 0], lambda T: np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)])),
    "c3": (100000, 100, [0.0, 0, 0], [1.0, 0, 0], lambda T: np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)])),
    "fast": (100000, 50, [0.3, -0.2, 0.7], [2.0, 1.0, 0.0], lambda T: np.array([np.full(T, 5.5), np.linspace(6.0, 4.0, T)])),
    "clip": (100000, 50, [0.0, 0, 3.1], [0.0, -1, 0], lambda T: np.array([np.full(T, 7.5), np.full(T, -6.3)])),
}


def main():
    names = sys.argv[1:] or list(CASES)
    for name in names:
        K, T, state, goal, nom = CASES[name]
        state, goal, u0 = np.array(state, f64), np.array(goal, f64), nom(T)
        eps = np.random.RandomState(1).normal(0, 0.9, (T, 2, K)).astype(f32).astype(f64)
        Vo = orc.get_cost2go(state, u0, goal, 1e-3, 0.9, eps)
        Vn = orc.get_cost2go(state, u0, goal, 1e-3, 0.9, np.zeros((T, 2, 1)))
        dP, Stot, almax = run(state, goal, u0, eps, T, 1.0 / T)
        V = Vn + Stot[None, :] - dP
        err = np.abs(V - Vo)
        best = np.argsort(Vo[0])[:1000]
        tol = 3e-7 * np.maximum(1.0, np.abs(Vo - Vn).max(axis=0))
        # what fp32 storage alone costs (the fp64 rollout's V rounded the same way)
        dPo = (Vo[0][None, :] - Vn[0]) - (Vo - Vn)
        Vs = Vn + (Vo[0] - Vn[0]).astype(f32).astype(f64)[None, :] - dPo.astype(f32).astype(f64)
        errs = np.abs(Vs - Vo)
        print("%-5s K=%d T=%d: max|err| %.3g (storage alone %.3g)  best-1000 %.3g  worst err/tol %.3f (storage alone %.3f)  "
              "max|alpha| %.3f" % (name, K, T, err.max(), errs.max(), err[:, best].max(), (err / tol[None, :]).max(),
                                   (errs / tol[None, :]).max(), almax))


if __name__ == "__main__":
    main()
