"""Would a CORRECTLY ROUNDED sin / cos reproduce the reference's (glibc) environment steps?  Acrobot and cartpole env-step vectors of the
genuine reference (tests/golden/vec_*.jsonl: 300 steps each), replayed in Python three ways: with this container's glibc (math.sin / cos --
must reproduce the vectors bit for bit, which validates the replay), with correctly rounded sin / cos (50-digit decimal Taylor series,
rounded once), and counting the individual trig calls where glibc's result is not the correctly rounded one."""
import json, math, os, sys
from decimal import Decimal, getcontext
getcontext().prec = 60
PI = Decimal("3.14159265358979323846264338327950288419716939937510582097494")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fh = float.fromhex

def d_sin(x):
    x = Decimal(x)
    k = (x / (2 * PI)).to_integral_value()
    x -= 2 * PI * k
    term, s, n = x, x, 1
    while abs(term) > Decimal(10) ** -55:
        term = -term * x * x / ((2 * n) * (2 * n + 1)); s += term; n += 1
    return float(s)
def d_cos(x):
    x = Decimal(x)
    k = (x / (2 * PI)).to_integral_value()
    x -= 2 * PI * k
    term, s, n = Decimal(1), Decimal(1), 1
    while abs(term) > Decimal(10) ** -55:
        term = -term * x * x / ((2 * n - 1) * (2 * n)); s += term; n += 1
    return float(s)

calls = {"n": 0, "glibc_not_cr": 0}
def mk(sin, cos, count=False):
    def S(x):
        r = sin(x)
        if count:
            calls["n"] += 1; calls["glibc_not_cr"] += int(r != d_sin(x))
        return r
    def C(x):
        r = cos(x)
        if count:
            calls["n"] += 1; calls["glibc_not_cr"] += int(r != d_cos(x))
        return r
    return S, C

FPI = 3.14159265358979323846264338327
def acrobot(a, s, S, C):                      # reference examples/acrobot/acrobot.c:31-130 (as csrc/envs.h restates it)
    vmax1, vmax2 = 4 * FPI, 9 * FPI
    m1 = m2 = l1 = 1.0; lc1 = lc2 = 0.5; I1 = I2 = 1.0; g = 9.8; dt = 0.05
    t1, t2, t1d, t2d = s
    c2, s2 = C(t2), S(t2)
    d1 = m1 * lc1 * lc1 + m2 * (l1 * l1 + lc2 * lc2 + 2 * l1 * lc2 * c2) + I1 + I2
    d2 = m2 * (lc2 * lc2 + l1 * lc2 * c2) + I2
    phi2 = m2 * lc2 * g * C(t1 + t2 - FPI / 2)
    phi1 = -m2 * l1 * lc2 * t2d * s2 * (t2d - 2 * t1d) + (m1 * lc1 + m2 * l1) * g * C(t1 - (FPI / 2)) + phi2
    acc2 = (a + phi1 * (d2 / d1) - m2 * l1 * lc2 * t1d * t1d * s2 - phi2)
    acc2 = acc2 / (m2 * lc2 * lc2 + I2 - (d2 * d2 / d1))
    acc1 = -(d2 * acc2 + phi1) / d1
    for _ in range(4):
        t1d = t1d + acc1 * dt; t1d = max(-vmax1, min(vmax1, t1d)); t1 = t1 + t1d * dt
        t2d = t2d + acc2 * dt; t2d = max(-vmax2, min(vmax2, t2d)); t2 = t2 + t2d * dt
    t1 = max(-FPI, min(FPI, t1)); t2 = max(-FPI, min(FPI, t2))
    return [t1, t2, t1d, t2d]
def cartpole(a, s, S, C):                     # reference examples/cartpole/cartpole.c:35-77
    x, xd, th, thd = s
    g, mc, mp, ln, fmag, tau = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    mt, pml = mc + mp, mp * ln
    force = a * fmag
    sn, cs = S(th), C(th)
    temp = (force + pml * thd * thd * sn) / mt
    thacc = (g * sn - cs * temp) / (ln * (4.0 / 3.0 - mp * cs * cs / mt))
    xacc = temp - pml * thacc * cs / mt
    return [x + tau * xd, xd + tau * xacc, th + tau * thd, thd + tau * thacc]

for env, step in (("acrobot", acrobot), ("cartpole", cartpole)):
    vec = [json.loads(l) for l in open(os.path.join(ROOT, "tests", "golden", f"vec_{env}.jsonl"))]
    vec = [r for r in vec if r["k"] == "env"]
    res = {}
    for name, (S, C) in (("glibc (math.sin/cos)", mk(math.sin, math.cos, True)), ("correctly rounded", mk(d_sin, d_cos))):
        diff = 0
        for r in vec:
            ns = step(fh(r["a"]), [fh(v) for v in r["s"]], S, C)
            diff += int(any(x.hex() != float.fromhex(w).hex() for x, w in zip(ns, r["ns"])))
        res[name] = diff
    print(env, "steps", len(vec), {k: f"{v} differ from the reference's vectors" for k, v in res.items()}, "| trig calls", calls["n"], "where glibc != correctly rounded:", calls["glibc_not_cr"])
    calls["n"] = calls["glibc_not_cr"] = 0
