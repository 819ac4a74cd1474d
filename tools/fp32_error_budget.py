#!/usr/bin/env python3
"""Design study behind DESIGN.md 4 ("why the rollout's state arithmetic stays fp64"): error budget of a rollout written
in DEVIATION form -- every per-sample quantity as its difference from the nominal (eps = 0) trajectory, the form that
would let fp32 / packed-fp32 instructions carry the state (task 4).  Runs on the CPU against the oracle:

    python tools/fp32_error_budget.py

Stages: A heading deviation (clip, d_phi, running sum d_theta), B sin/cos series + rotation of the nominal mid-step
heading, C wheel-sum / Simpson-weight deviations and the position increments, D running sums dX, dY, E stage cost from
dX, dY.  Each line switches the named stages to float32 (numpy rounds every operation, no fma) and reports the largest
error of the cost-to-go V against the float64 oracle, over all samples and over the 1000 best ones -- the ones whose
softmax weights matter (lambda = 1e-3 turns a V error of 1e-5 into a 1 % weight error)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc

f64 = np.float64
UMAX, R, WB = 6.35492, 0.033, 0.16


def nominal(state, goal, u0, T, dt, q=1e3, p1=(1e3, 1e3, 1e3), lam=1e-3, sig=0.9, rr=(1.0, 1.0)):
    kth, rhalf = R / WB, R / 2
    hk = 0.5 * kth * dt
    ac = np.clip(u0, -UMAX, UMAX)
    pn = hk * ac
    phin = pn[1] - pn[0]
    th = state[2] + np.concatenate([[0.0], np.cumsum(2 * phin)])[:-1]   # start-of-step headings (unwrapped)
    f = np.sqrt(0.5 * q)
    rho = f * (dt * rhalf / 6.0) / hk
    c1n, s1n = np.cos(th + phin), np.sin(th + phin)
    Wn = 4 + 2 * np.cos(phin)
    Pn = pn[0] + pn[1]
    incx, incy = rho * Pn * Wn * c1n, rho * Pn * Wn * s1n
    Xn = f * (state[0] - goal[0]) + np.cumsum(incx)
    Yn = f * (state[1] - goal[1]) + np.cumsum(incy)
    thT = th[-1] + 2 * phin[-1]
    return dict(hk=hk, ac=ac, a=u0, pn=pn, phin=phin, th=th, f=f, rho=rho, c1n=c1n, s1n=s1n, Wn=Wn, Pn=Pn,
                Xn=Xn, Yn=Yn, thT=thT, w=lam * sig * u0)




def run(state, goal, u0, eps, T, dt, low, lam=1e-3, sig=0.9):
    n = nominal(state, goal, u0, T, dt, lam=lam, sig=sig)
    K = eps.shape[2]
    def ty(stage):
        return np.float32 if stage in low else np.float64
    A, B, C, D, E = ty("A"), ty("B"), ty("C"), ty("D"), ty("E")
    dth = np.zeros(K, A); dX = np.zeros(K, D); dY = np.zeros(K, D)
    pre = np.zeros(K, f64)
    dP = np.zeros((T, K), f64)
    for t in range(T):
        dP[t] = pre
        e0, e1 = eps[t, 0].astype(A), eps[t, 1].astype(A)
        hk = A(n["hk"])
        d = [A(n["hk"] * (n["a"][i, t] - n["ac"][i, t])) for i in range(2)]
        lo = [A(n["hk"] * (-UMAX - n["ac"][i, t])) for i in range(2)]
        hi = [A(n["hk"] * (UMAX - n["ac"][i, t])) for i in range(2)]
        dp0 = np.clip(e0 * hk + d[0], lo[0], hi[0]); dp1 = np.clip(e1 * hk + d[1], lo[1], hi[1])
        dphi = dp1 - dp0; sp = dp0 + dp1
        al = dth + dphi
        dth = al + dphi
        alB = al.astype(B)
        z = alB * alB
        S = alB * (B(1) + z * (B(-1 / 6) + z * (B(1 / 120) + z * (B(-1 / 5040) + z * B(1 / 362880)))))
        Cm = z * (B(-0.5) + z * (B(1 / 24) + z * (B(-1 / 720) + z * (B(1 / 40320) + z * B(-1 / 3628800)))))
        c1n, s1n = B(n["c1n"][t]), B(n["s1n"][t])
        dc1 = c1n * Cm - s1n * S; ds1 = s1n * Cm + c1n * S
        rho = n["rho"]
        dphiC, spC = dphi.astype(C), sp.astype(C)
        zz = dphiC * dphiC
        A1 = C(-2 * np.sin(n["phin"][t]) * rho); Cn = C(-np.cos(n["phin"][t]) * rho)
        if "W" in low:
            dW = dphiC * (A1 + Cn * dphiC)
        else:
            dW = dphiC * (A1 * (C(1) + zz * C(-1 / 6) + zz * zz * C(1 / 120)) + Cn * dphiC * (C(1) + zz * C(-1 / 12) + zz * zz * C(1 / 360)))
        Wn = C(n["Wn"][t] * rho); Pn = C(n["Pn"][t])
        P = Pn + spC
        Aq = P * dW
        t1 = spC * Wn + Aq
        G = P * Wn + Aq
        ix = t1 * C(n["c1n"][t]) + G * dc1.astype(C)
        iy = t1 * C(n["s1n"][t]) + G * ds1.astype(C)
        dX = dX + ix.astype(D); dY = dY + iy.astype(D)
        dXe, dYe = dX.astype(E), dY.astype(E)
        X2, Y2 = E(2 * n["Xn"][t]), E(2 * n["Yn"][t])
        dc = dXe * (X2 + dXe) + dYe * (Y2 + dYe) + E(n["w"][0, t]) * eps[t, 0].astype(E) + E(n["w"][1, t]) * eps[t, 1].astype(E)
        pre = pre + dc.astype(f64)
    th = n["thT"] + dth.astype(f64)
    wrap = lambda a: a - (np.ceil((a + np.pi) / (2 * np.pi)) - 1.0) * 2 * np.pi
    thw, thn = wrap(th), wrap(n["thT"])
    x = (n["Xn"][-1] + dX.astype(f64)) / n["f"]; y = (n["Yn"][-1] + dY.astype(f64)) / n["f"]
    xn, yn = n["Xn"][-1] / n["f"], n["Yn"][-1] / n["f"]
    term = 1e3 * (x * x + y * y + (thw - goal[2]) ** 2) - 1e3 * (xn * xn + yn * yn + (thn - goal[2]) ** 2)
    return dP, pre + term


K, T = 100000, 50
state, goal = np.array([0.0, 0, 0]), np.array([0.0, -1, 0])
u0 = np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)])
eps = np.random.RandomState(1).normal(0, 0.9, (T, 2, K)).astype(np.float32).astype(np.float64)
Vo = orc.get_cost2go(state, u0, goal, 1e-3, 0.9, eps)
Vn = orc.get_cost2go(state, u0, goal, 1e-3, 0.9, np.zeros((T, 2, 1)))
best = np.argsort(Vo[0])[:1000]
for low in ("", "A", "B", "C", "D", "E", "W", "ABCDE", "ABCD", "ABC", "AB"):
    dP, Stot = run(state, goal, u0, eps, T, 1.0 / T, set(low))
    V = Vn + Stot[None, :] - dP
    err = np.abs(V - Vo)
    print("fp32 stages %-6s max|err| %.3g   on best 1000 %.3g" % (low or "-", err.max(), err[:, best].max()))
