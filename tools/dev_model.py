#!/usr/bin/env python3
"""numpy float32 model of the deviation-form rollout (design study for rollout_pk_kernel).
Per-sample quantities are float32 arrays (each numpy op rounds to fp32, no fma: slightly pessimistic);
nominal per-step constants are computed in float64 and rounded to fp32 where the kernel holds them as fp32."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc

f32 = np.float32
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


def rollout_dev(state, goal, u0, eps32, T, dt, variant="full", lam=1e-3, sig=0.9):
    n = nominal(state, goal, u0, T, dt, lam=lam, sig=sig)
    K = eps32.shape[2]
    hk = f32(n["hk"])
    dth = np.zeros(K, f32); dX = np.zeros(K, f32); dY = np.zeros(K, f32)
    dXl = np.zeros(K, f32); dYl = np.zeros(K, f32)   # low parts (two-float accumulators)
    pre64 = np.zeros(K, np.float64); part = np.zeros(K, f32)
    dP = np.zeros((T, K), np.float64)
    amax = np.zeros(K, f32)
    for t in range(T):
        if t % 6 == 0:
            pre64 += part.astype(np.float64); part[:] = 0
            base32 = pre64.astype(f32)
        dP[t] = (base32 + part).astype(np.float64)
        e0, e1 = eps32[t, 0], eps32[t, 1]
        d = [f32(n["hk"] * (n["a"][i, t] - n["ac"][i, t])) for i in range(2)]
        lo = [f32(n["hk"] * (-UMAX - n["ac"][i, t])) for i in range(2)]
        hi = [f32(n["hk"] * (UMAX - n["ac"][i, t])) for i in range(2)]
        dp0 = np.clip(e0 * hk + d[0], lo[0], hi[0]); dp1 = np.clip(e1 * hk + d[1], lo[1], hi[1])
        dphi = dp1 - dp0; sp = dp0 + dp1
        al = dth + dphi
        dth = al + dphi
        amax = np.maximum(amax, np.abs(al))
        z = al * al
        S = al * (f32(1) + z * (f32(-1 / 6) + z * (f32(1 / 120) + z * f32(-1 / 5040))))
        Cm = z * (f32(-0.5) + z * (f32(1 / 24) + z * (f32(-1 / 720) + z * f32(1 / 40320))))
        c1n, s1n = f32(n["c1n"][t]), f32(n["s1n"][t])
        dc1 = c1n * Cm - s1n * S; ds1 = s1n * Cm + c1n * S
        rho = n["rho"]
        A1 = f32(-2 * np.sin(n["phin"][t]) * rho); Cn = f32(-np.cos(n["phin"][t]) * rho)
        if variant == "full":
            zz = dphi * dphi
            dW = dphi * (A1 * (f32(1) + zz * f32(-1 / 6)) + Cn * dphi * (f32(1) + zz * f32(-1 / 12)))
        else:
            dW = dphi * (A1 + Cn * dphi)
        Wn = f32(n["Wn"][t] * rho); Pn = f32(n["Pn"][t])
        P = Pn + sp
        A = P * dW
        t1 = sp * Wn + A
        G = P * Wn + A
        ix = t1 * c1n + G * dc1
        iy = t1 * s1n + G * ds1
        if "2f" in variant:   # two-float accumulation (branch-free 2Sum)
            def twosum(a, b):
                s_ = a + b; bb = s_ - a; return s_, (a - (s_ - bb)) + (b - bb)
            sx, ex = twosum(dX, ix); dXl = dXl + ex; dX = sx
            sy, ey = twosum(dY, iy); dYl = dYl + ey; dY = sy
            X2h = f32(2 * n["Xn"][t]); X2l = f32(2 * n["Xn"][t] - np.float64(X2h))
            Y2h = f32(2 * n["Yn"][t]); Y2l = f32(2 * n["Yn"][t] - np.float64(Y2h))
            dc = (dX * (X2h + dX) + dY * (Y2h + dY)) + ((dXl * (X2h + dX + dX) + dX * X2l) + (dYl * (Y2h + dY + dY) + dY * Y2l)) \
                + (f32(n["w"][0, t]) * e0 + f32(n["w"][1, t]) * e1)
        else:
            dX = dX + ix
            dY = dY + iy
            X2 = f32(2 * n["Xn"][t]); Y2 = f32(2 * n["Yn"][t])
            dc = dX * (X2 + dX) + dY * (Y2 + dY) + f32(n["w"][0, t]) * e0 + f32(n["w"][1, t]) * e1
        part = part + dc
    pre64 += part.astype(np.float64)
    # terminal (fp64, once per sample)
    th = n["thT"] + dth.astype(np.float64)
    wrap = lambda a: a - (np.ceil((a + np.pi) / (2 * np.pi)) - 1.0) * 2 * np.pi
    thw, thn = wrap(th), wrap(n["thT"])
    x = (n["Xn"][-1] + dX.astype(np.float64) + dXl.astype(np.float64)) / n["f"]; y = (n["Yn"][-1] + dY.astype(np.float64) + dYl.astype(np.float64)) / n["f"]
    xn, yn = n["Xn"][-1] / n["f"], n["Yn"][-1] / n["f"]
    term = 1e3 * (x * x + y * y + (thw - goal[2]) ** 2) - 1e3 * (xn * xn + yn * yn + (thn - goal[2]) ** 2)
    Stot = (pre64 + term).astype(f32).astype(np.float64)
    return dP, Stot, amax


def main():
    K, T = int(sys.argv[1]) if len(sys.argv) > 1 else 200000, int(sys.argv[2]) if len(sys.argv) > 2 else 50
    dt = 1.0 / T
    cases = {"warm_park": ([0, 0, 0], [0, -1, 0], np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)])),
             "zero_park": ([0, 0, 0], [0, -1, 0], np.zeros((2, T))),
             "hot_clip": ([0.1, -0.05, 2.9], [0.4, -1.0, -2.8], np.array([np.linspace(5.5, 6.3, T), np.linspace(-6.3, -5.0, T)])),
             "over_clip": ([0, 0, 3.1], [1, 0, 0], np.array([np.linspace(6.0, 7.5, T), np.linspace(-7.0, -6.0, T)]))}
    for name, (state, goal, u0) in cases.items():
        eps = np.random.RandomState(1).normal(0, 0.9, (T, 2, K)).astype(f32)
        e64 = eps.astype(np.float64)
        Vo = orc.get_cost2go(state, u0, goal, 1e-3, 0.9, e64)
        Vn = orc.get_cost2go(state, u0, goal, 1e-3, 0.9, np.zeros((T, 2, 1)))
        for variant in ("simple", "simple2f"):
            dP, Stot, amax = rollout_dev(np.array(state, float), np.array(goal, float), u0, eps, T, dt, variant)
            V = Vn + Stot[None, :] - dP
            err = np.abs(V - Vo)
            dV = np.abs(Vo - Vn).max(axis=0)
            rel = (err.max(axis=0) / np.maximum(1.0, dV))
            best = np.argsort(Vo[0])[:1000]
            print("%-10s %-6s max|err| %.3g  max err/max(1,|dV|) %.3g  err on 1000 best samples %.3g  |dV|max %.3g  alpha max %.3g" % (
                name, variant, err.max(), rel.max(), err[:, best].max(), dV.max(), amax.max()))


if __name__ == "__main__":
    main()
