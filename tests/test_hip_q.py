"""GPU parity of the Q-value kernels through the C ABI against the oracle.

five_hip_vag_concl / five_hip_vag_concl_weight / frirl_hip_get_best_action
(reference FIVEVagConcl.c:64-351, FIVEVagConclWeight.c:52-188, frirl_get_best_action.c:31-341).
Bars: hit indices and arg-max actions BIT-EXACT; exact-hit Q values BIT-EXACT (a copy of the rule's
consequent); interpolated Q values and weights within 1e-6 relative (north star) -- the tests assert a
much tighter 1e-11, what the plain-double power + tree reduction actually deliver.
"""
import numpy as np
import pytest

from oracle import binding as ob
from tests.problems import Batch, demo_device_batch

pytestmark = pytest.mark.gpu
RTOL = 1e-11


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rel(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-300)


CASES = [(3, 41, 8, 6, 3), (3, 41, 110, 40, 3), (5, 1001, 182, 24, 21), (5, 41, 367, 33, 3), (5, 41, 4097, 6, 3), (8, 101, 2500, 4, 5),
         (2, 41, 300, 5, 3), (9, 41, 700, 3, 4)]


@pytest.mark.parametrize("nant,U,R,E,A", CASES)
def test_vag_concl_and_weights(nant, U, R, E, A):
    import torch
    b = Batch(nant, U, R, E, A=A, seed=7 + R, ragged=True)
    b.nrules[b.nrules == 0] = 1          # Q of an empty rule base is 0/0 in the reference as well: not a parity case
    x = b.queries(seed=R + 1, hit_fraction=0.4)
    prob = b.to_device()
    conc, hit = prob.vag_concl(dev(x))
    w, hitw = prob.vag_concl_weight(dev(x))
    torch.cuda.synchronize()
    conc, hit, w, hitw = conc.cpu().numpy(), hit.cpu().numpy(), w.cpu().numpy(), hitw.cpu().numpy()
    nh = 0
    for e in range(E):
        f = b.five(e)
        n = f.R
        h, c = f.vag_concl(x[e])
        assert hit[e] == h and hitw[e] == h, e
        if h >= 0:
            nh += 1
            assert c == conc[e], "exact hit returns the rule's consequent bit for bit"
            assert np.isnan(w[e]).all(), "weights row of an exact-hit environment must stay untouched"
        else:
            assert rel(conc[e], c) <= RTOL, (e, conc[e], c)
            assert f.vag_concl_weight(x[e]) == -1
            assert rel(w[e, :n], f.weights[:n]).max() <= RTOL
            assert abs(w[e, :n].sum() - 1.0) < 1e-12
    assert 0 < nh < E or E < 8


@pytest.mark.parametrize("nant,U,R,E,A", [c for c in CASES if 2 <= c[0] <= 9])
def test_get_best_action(nant, U, R, E, A):
    import torch
    b = Batch(nant, U, R, E, A=A, seed=70 + R, ragged=True)
    b.nrules[b.nrules == 0] = 1
    ave, avals = b.action_ve()
    x = b.queries(seed=R + 3, hit_fraction=0.5)
    # on-grid action values in the hitting queries so that exact hits occur for one action only
    prob = b.to_device()
    states = np.ascontiguousarray(x[:, : nant - 1])
    actconc, best = prob.get_best_action(dev(states), dev(ave))
    torch.cuda.synchronize()
    actconc, best = actconc.cpu().numpy(), best.cpu().numpy()
    for e in range(E):
        f = b.five(e)
        bo, ac = f.best_action(states[e], ave)
        assert rel(actconc[e], ac).max() <= RTOL, (e, actconc[e], ac)
        srt = np.sort(ac)
        if len(srt) > 1 and (srt[-1] - srt[-2]) > 1e-9 * max(1.0, abs(srt[-1])):
            assert best[e] == bo, (e, actconc[e], ac)
        else:   # numerically tied maxima: the GPU may only pick among the tied ones
            assert abs(ac[best[e]] - srt[-1]) <= 1e-9 * max(1.0, abs(srt[-1]))


@pytest.mark.parametrize("nant,U,A,compressed", [(5, 1001, 21, True), (5, 1001, 21, False), (3, 41, 9, False), (4, 101, 10, True), (5, 41, 13, False),
                                                 (3, 101, 24, True), (5, 101, 25, False), (2, 41, 32, False)])
def test_get_best_action_many_actions(nant, U, A, compressed):
    """More than 8 actions at >= 256 environments (the 256-thread kernels): 9..24 actions keep every accumulator pair in registers
    (sweep_gba_many: A rounded up to a multiple of three, exact hits recorded in the side path), more use the action-parallel waves
    (sweep_gba_wide); `no_many` forces the latter.  Both against the oracle: arg-max exact, exact-hit conclusions bit-exact, interpolated
    ones within RTOL; odd rule counts, empty-ish rule bases and hits on the last rule included."""
    import torch
    import frirl_amd
    E, R = 259, 2600 if compressed else 700
    b = Batch(nant, U, R, E, A=A, seed=11 + A, ragged=True)
    b.nrules[b.nrules == 0] = 1
    b.nrules[1] = 1
    b.nrules[2] = R - 1 if (R - 1) % 2 else R - 2          # an odd count: the tail pair has no second rule
    ave, _ = b.action_ve()
    x = b.queries(seed=A, hit_fraction=0.6)
    states = np.ascontiguousarray(x[:, : nant - 1])
    prob = b.to_device(compressed=compressed)
    got = {}
    for nm in (0, 1):
        old = frirl_amd.set_option("no_many", nm)
        try:
            actconc, best = prob.get_best_action(dev(states), dev(ave))
            torch.cuda.synchronize()
        finally:
            frirl_amd.set_option("no_many", old)
        got[nm] = (actconc.cpu().numpy(), best.cpu().numpy())
    hits = 0
    for nm in (0, 1):
        actconc, best = got[nm]
        for e in list(range(0, E, 7)) + [1, 2]:
            f = b.five(e)
            bo, ac = f.best_action(states[e], ave)
            assert rel(actconc[e], ac).max() <= RTOL, (nm, e, actconc[e], ac)
            n = int(b.nrules[e])
            exact = np.isin(ac, b.rb[e, nant, :n]) & (actconc[e] == ac)
            hits += int(exact.sum())
            srt = np.sort(ac)
            if (srt[-1] - srt[-2]) > 1e-9 * max(1.0, abs(srt[-1])):
                assert best[e] == bo, (nm, e, actconc[e], ac)
    assert hits > 0, "no exact hit exercised"
    # the two forms sum in different orders: same arg-max wherever the oracle's maximum is clear (checked above), values within RTOL
    assert rel(got[0][0], got[1][0]).max() <= 10 * RTOL


def test_get_best_action_exact_ties_pick_first():
    """The observed state hits one rule per action with equal consequents => exactly tied Q values =>
    the FIRST maximum (index 0) wins (reference src/inl/max.inl:21, strict <)."""
    import torch
    b = Batch(3, 41, 256, 4, A=3, seed=5, ragged=False)
    U = b.U
    aidx = [0, (U - 1) // 2, U - 1]
    for e in range(b.E):
        for a in range(3):          # rules 10, 11, 12: same state, the three actions, same Q
            r = 10 + a
            b.uidx[e, :2, r] = b.uidx[e, :2, 10]
            b.uidx[e, 2, r] = aidx[a]
            b.rb[e, :3, r] = b.ve[np.arange(3), b.uidx[e, :, r]]
            b.rb[e, 3, r] = 7.0 + e
    ave, _ = b.action_ve()
    states = np.ascontiguousarray(b.u[np.arange(2), b.uidx[:, :2, 10]])
    prob = b.to_device()
    actconc, best = prob.get_best_action(dev(states), dev(ave))
    torch.cuda.synchronize()
    assert (best.cpu().numpy() == 0).all()
    assert (actconc.cpu().numpy() == (7.0 + np.arange(b.E))[:, None]).all()
    for e in range(b.E):
        bo, ac = b.five(e).best_action(states[e], ave)
        assert bo == 0 and (ac == 7.0 + e).all()


@pytest.mark.parametrize("env,episodes", [("mountaincar", 6), ("cartpole", 9), ("acrobot", 5)])
def test_demo_rulebases(env, episodes):
    """Real demo tables / rule bases: quantised on-grid states (mostly exact hits) and off-grid states."""
    import torch
    E = 48
    b, fr = demo_device_batch(env, episodes, E)
    f = fr.five
    rng = np.random.default_rng(3)
    nant = f.nant
    x = np.zeros((E, nant))
    for e in range(E):
        for k in range(nant):
            vals = fr.dim(k)["values"]
            x[e, k] = vals[rng.integers(len(vals))]
        if e % 3 == 0:      # an existing rule: exact hit
            x[e] = b.rant[rng.integers(len(b.rant))]
        if e % 3 == 2:      # off-grid state, on-grid action
            for k in range(nant - 1):
                vals = fr.dim(k)["values"]
                x[e, k] = rng.uniform(vals[0], vals[-1])
    prob = b.to_device()
    conc, hit = prob.vag_concl(dev(x))
    ave = np.array(fr.action_vevalues)
    actconc, best = prob.get_best_action(dev(np.ascontiguousarray(x[:, : nant - 1])), dev(ave))
    torch.cuda.synchronize()
    conc, hit, actconc, best = conc.cpu().numpy(), hit.cpu().numpy(), actconc.cpu().numpy(), best.cpu().numpy()
    hits = 0
    for e in range(E):
        h, c = f.vag_concl(x[e])
        assert hit[e] == h
        if h >= 0:
            hits += 1
            assert conc[e] == c
        else:
            assert rel(conc[e], c) <= RTOL
        bo = fr.get_best_action(x[e, : nant - 1])
        ac = np.array(fr.actconc)
        assert rel(actconc[e], ac).max() <= RTOL
        srt = np.sort(ac)
        if (srt[-1] - srt[-2]) > 1e-9 * max(1.0, abs(srt[-1])):
            assert best[e] == bo
    assert hits > 0


@pytest.mark.parametrize("env,episodes", [("mountaincar", 20), ("cartpole", 30), ("acrobot", 40)])
def test_shared_rule_base_evaluation(env, episodes):
    """many observations against ONE trained rule base (lane = observation, sequential sums in the
    reference's order).  Hits / first-max actions exact; Q within tolerance of the oracle AND of the per-environment
    kernels run on copies of the same base."""
    import torch
    Qn = 1000
    b, fr = demo_device_batch(env, episodes, 1)
    f = fr.five
    nant = f.nant
    prob = b.to_device()
    rng = np.random.default_rng(9)
    x = np.zeros((Qn, nant))
    for i in range(Qn):
        for k in range(nant):
            vals = fr.dim(k)["values"]
            x[i, k] = vals[rng.integers(len(vals))] if (i % 2 == 0 or k == nant - 1) else rng.uniform(vals[0], vals[-1])
        if i % 5 == 0:
            x[i] = b.rant[rng.integers(len(b.rant))]
    conc, hit = prob.vag_concl_shared(dev(x))
    ave = np.array(fr.action_vevalues)
    actconc, best = prob.get_best_action_shared(dev(np.ascontiguousarray(x[:, : nant - 1])), dev(ave))
    torch.cuda.synchronize()
    conc, hit, actconc, best = conc.cpu().numpy(), hit.cpu().numpy(), actconc.cpu().numpy(), best.cpu().numpy()
    hits = 0
    for i in range(0, Qn, 3):
        h, c = f.vag_concl(x[i])
        assert hit[i] == h
        if h >= 0:
            hits += 1
            assert conc[i] == c
        else:
            assert rel(conc[i], c) <= 1e-12, (i, conc[i], c)      # same summation order as the reference: tighter than the tree
        bo = fr.get_best_action(x[i, : nant - 1])
        ac = np.array(fr.actconc)
        assert rel(actconc[i], ac).max() <= 1e-12
        srt = np.sort(ac)
        if (srt[-1] - srt[-2]) > 1e-9 * max(1.0, abs(srt[-1])):
            assert best[i] == bo
    assert hits > 10


@pytest.mark.parametrize("nant,U,R,A", [(2, 41, 300, 3), (8, 101, 1500, 5), (9, 33, 700, 21), (4, 1001, 513, 8)])
def test_shared_rule_base_evaluation_synthetic_shapes(nant, U, R, A):
    """The shared-base query kernels for every supported antecedent count (2..9), several tiles of rules, ragged last tile,
    action chunks (A = 21 > 8), 1 in 8 queries an exact hit: hits / first maxima exact, Q <= 1e-10 (sequential sums; consequents of both signs cancel in some sums, which
    amplifies the 1e-16 differences of the weights)."""
    import torch
    from oracle import binding as ob
    f = ob.synth_problem(nant, U, R, A, seed=nant * 1000 + R)
    maxR = f.maxR + (f.maxR & 1)
    rb = np.zeros((1, nant + 1, maxR))
    rb[0, :nant, :R] = f.veval[:, :R]
    rb[0, nant, :R] = f.rconc[:R]
    import frirl_amd
    prob = frirl_amd.Problem(dev(np.array(f.u)), dev(np.array(f.ve)), dev(rb), dev(np.array([R], dtype=np.int32)))
    Qn, st = 260, 99
    x = np.zeros((Qn, nant))
    for i in range(Qn):
        x[i], st = ob.synth_query(f, st, i)
    conc, hit = prob.vag_concl_shared(dev(x))
    # action VE points: A distinct VE values of the last universe
    ave = np.array(f.ve)[nant - 1, np.linspace(0, U - 1, A).astype(int)].copy()
    actconc, best = prob.get_best_action_shared(dev(np.ascontiguousarray(x[:, : nant - 1])), dev(ave))
    torch.cuda.synchronize()
    conc, hit, actconc, best = conc.cpu().numpy(), hit.cpu().numpy(), actconc.cpu().numpy(), best.cpu().numpy()
    hits = 0
    for i in range(Qn):
        h, c = f.vag_concl(x[i])
        assert hit[i] == h, (i, hit[i], h)
        hits += h >= 0
        assert (conc[i] == c) if h >= 0 else (rel(conc[i], c) <= 1e-10)
        bo, ac = f.best_action(x[i, : nant - 1], ave)
        assert rel(actconc[i], ac).max() <= 1e-10
        srt = np.sort(ac)
        if len(srt) < 2 or (srt[-1] - srt[-2]) > 1e-9 * max(1.0, abs(srt[-1])):
            assert best[i] == bo
    assert hits >= Qn // 10


@pytest.mark.parametrize("nant,U,R,A,compressed", [(3, 41, 111, 3, False), (5, 41, 367, 3, True), (5, 1001, 183, 21, True), (5, 41, 4097, 3, True), (4, 101, 301, 10, False)])
def test_rows_beyond_nrules_are_never_read_into_a_result(nant, U, R, A, compressed):
    """The slab beyond nrules[e] is the caller's memory (a C host that hipMallocs without memset): NaN there -- the phantom second
    rule of an odd tail included -- must not reach any conclusion, weight or greedy action.  Odd rule counts everywhere."""
    import torch
    E = 9 if A <= 8 else 259
    b = Batch(nant, U, R, E, A=A, seed=21 + R, ragged=True, maxR=R + 7)
    b.nrules[b.nrules == 0] = 1
    b.nrules[b.nrules % 2 == 0] -= 1
    b.nrules[0] = R
    ave, _ = b.action_ve()
    x = b.queries(seed=R, hit_fraction=0.4)
    clean = b.to_device(compressed=compressed)
    for e in range(E):
        b.rb[e, :, int(b.nrules[e]):] = np.nan
    dirty = b.to_device(compressed=compressed)
    states = np.ascontiguousarray(x[:, : nant - 1])
    for prob in (clean, dirty):
        prob.out = (prob.vag_concl(dev(x)), prob.get_best_action(dev(states), dev(ave)), prob.vag_concl_weight(dev(x)))
    torch.cuda.synchronize()
    (c0, h0), (a0, b0), (w0, _) = clean.out
    (c1, h1), (a1, b1), (w1, _) = dirty.out
    assert torch.isfinite(c1).all() and torch.isfinite(a1).all()
    assert (c0 == c1).all() and (h0 == h1).all() and (a0 == a1).all() and (b0 == b1).all()
    for e in range(E):
        n = int(b.nrules[e])
        if int(h1[e]) < 0 or int(h1[e]) == 0xFFFFFFFF or int(h1[e]) >= n:
            assert (w0[e, :n] == w1[e, :n]).all() and torch.isfinite(w1[e, :n]).all()
