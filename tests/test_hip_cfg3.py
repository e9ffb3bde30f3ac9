"""GPU parity at BASELINE.json's cfg3 shape (cartpole tables: nant 5, U 1001, A 21, 32 768 rules per environment) and
cfg5's full rule-base size (nant 16, U 1001, 262 144 rules), plus the genuine-reference synthetic vectors
(tests/golden/synth_*.jsonl, produced by oracle/_ref's harness) fed to the HIP path directly.

What runs only at these shapes: the 40 KB / 125 KB LDS-table index kernels, the action-parallel greedy sweep
(sweep_gba_wide, A = 21) over tens of thousands of rules, the 256-thread episode step with five antecedents and 21 actions.
Bars as everywhere: distances / hit indices / chosen actions / appended antecedents BIT-EXACT, interpolated Q within the
1e-6 contract (asserted at 1e-11 / 1e-9).
"""
import json
import os

import numpy as np
import pytest

import frirl_amd
from oracle import binding as ob
from tests.problems import Batch

pytestmark = pytest.mark.gpu
NANT, U, R, A = 5, 1001, 32768, 21
RTOL = 1e-11


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def rel(a, b, floor=1e-300):
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


@pytest.fixture(scope="module")
def cartpole():
    return frirl_amd.demo_describe("cartpole")


@pytest.fixture(scope="module")
def batch(cartpole):
    """16 ragged rule bases of up to 32 768 rules on the cartpole universes / vague environments."""
    assert cartpole["nant"] == NANT and cartpole["U"] == U and cartpole["A"] == A
    b = Batch(NANT, U, R, 16, A=A, seed=333, ragged=True, tables=(cartpole["u"], cartpole["ve"]))
    n3 = int(b.nrules[3]) | 1    # an odd rule count (tail lane of a 16-byte pair)
    if n3 > int(b.nrules[3]):
        n3 -= 2
    b.rb[3, :, n3:] = 0.0
    b.uidx[3, :, n3:] = 0
    b.nrules[3] = n3
    b.nrules[2] = 1              # Q of an empty rule base is 0/0 in the reference too: one rule on the first universe points
    b.uidx[2, :, 0] = 0
    b.rb[2, :NANT, 0] = b.ve[:, 0]
    b.rb[2, NANT, 0] = 12.5
    return b


def test_rule_distance_both_layouts(batch):
    """five_hip_rule_distance at cfg3's rule-base size: the f64 columns and the compressed index mirror (40 KB of VE tables
    in LDS) against the oracle, bit for bit, materialised and index-only."""
    import torch
    b = batch
    x = b.queries(seed=91, hit_fraction=0.4)
    d_ref, hit_ref = b.oracle_rule_distance(x)
    assert (hit_ref >= 0).any() and (hit_ref < 0).any()
    for compressed in (False, True):
        prob = b.to_device(compressed=compressed)
        d, hit = prob.rule_distance(dev(x))
        _, hit2 = prob.rule_distance(dev(x), materialise=False)
        torch.cuda.synchronize()
        d, hit, hit2 = d.cpu().numpy(), hit.cpu().numpy(), hit2.cpu().numpy()
        assert (hit.astype(np.int64) == hit_ref.astype(np.int64)).all() and (hit2 == hit).all(), compressed
        for e in range(b.E):
            n = int(b.nrules[e])
            assert (bits(d[e, :n]) == bits(d_ref[e, :n])).all(), (compressed, e)


def test_q_kernels(batch):
    """vag_concl / vag_concl_weight / get_best_action (the A = 21 action-parallel sweep) on 32 768-rule bases vs the oracle."""
    import torch
    b = batch
    ave, _ = b.action_ve()
    x = b.queries(seed=17, hit_fraction=0.5)
    states = np.ascontiguousarray(x[:, : NANT - 1])
    for compressed in (False, True):
        prob = b.to_device(compressed=compressed)
        conc, hit = prob.vag_concl(dev(x))
        w, hitw = prob.vag_concl_weight(dev(x))
        actconc, best = prob.get_best_action(dev(states), dev(ave))
        torch.cuda.synchronize()
        conc, hit, w, hitw, actconc, best = (t.cpu().numpy() for t in (conc, hit, w, hitw, actconc, best))
        hits = 0
        for e in range(b.E):
            f = b.five(e)
            n = f.R
            h, c = f.vag_concl(x[e])
            assert hit[e] == h and hitw[e] == h, (compressed, e)
            if h >= 0:
                hits += 1
                assert conc[e] == c
            else:
                assert rel(conc[e], c) <= RTOL, (compressed, e, conc[e], c)
                assert f.vag_concl_weight(x[e]) == -1
                assert rel(w[e, :n], f.weights[:n]).max() <= RTOL
            bo, ac = f.best_action(states[e], ave)
            assert rel(actconc[e], ac).max() <= RTOL, (compressed, e)
            srt = np.sort(ac)
            if (srt[-1] - srt[-2]) > 1e-9 * max(1.0, abs(srt[-1])):
                assert best[e] == bo, (compressed, e)
            else:
                assert abs(ac[best[e]] - srt[-1]) <= 1e-9 * max(1.0, abs(srt[-1]))
        assert 0 < hits < b.E


def test_update_sarsa_all_branches(batch):
    """frirl_hip_update_sarsa with cartpole's agent on 32 768-rule bases: exact-hit write, weighted spread, insert
    (grid-snapped antecedents, index mirror kept in sync), inactive environments -- vs the oracle's update_sarsa."""
    import torch
    b = batch
    E = b.E
    fr = ob.Frirl("cartpole", trig_mode=1)
    oag = fr.agent()
    from tests.problems import device_agent
    agent = device_agent(fr)
    dims = [fr.dim(k) for k in range(NANT)]
    seen = set()
    for compressed in (False, True):
        maxR = b.maxR + 64
        rb = np.zeros((E, NANT + 1, maxR))
        rb[:, :, : b.maxR] = b.rb
        uidx = np.zeros((E, NANT, maxR), dtype=np.int16)
        uidx[:, :, : b.maxR] = b.uidx
        prob = frirl_amd.Problem(dev(b.u), dev(b.ve), dev(rb), dev(b.nrules), dev(uidx) if compressed else None)
        rant0 = torch.zeros((E, NANT, maxR), dtype=torch.float64, device="cuda")
        envs = frirl_amd.Envs(prob, "cuda", rant_init=rant0)
        fives = []
        for e in range(E):
            n = int(b.nrules[e])
            rant = np.ascontiguousarray(b.u[np.arange(NANT)[:, None], b.uidx[e, :, :n]].T)
            fives.append(ob.Five(b.u.ravel(), b.ve.ravel(), NANT, U, maxR, rant, np.ascontiguousarray(b.rb[e, NANT, :n])))
        fus = np.zeros(E)
        rng = np.random.default_rng(5)
        for rnd in range(4):
            q_ant, cur, reward = np.zeros((E, NANT)), np.zeros((E, NANT)), np.zeros(E)
            for e in range(E):
                for k in range(NANT):
                    v = dims[k]["values"]
                    q_ant[e, k] = v[rng.integers(len(v))]
                    cur[e, k] = v[rng.integers(len(v))]
                if e % 4 == 1:          # an existing rule: exact hit
                    n = fives[e].R
                    q_ant[e] = fives[e].rant[rng.integers(n)]
                if e % 4 == 2:          # off-grid state
                    for k in range(NANT - 1):
                        v = dims[k]["values"]
                        q_ant[e, k] = rng.uniform(v[0], v[-1]) * 0.9
                reward[e] = [-10.0, 1000.0, -3000.0 * rng.random(), 20 * rng.random() - 10][(e + rnd) % 4]
            active = np.ones(E, dtype=np.uint8)
            active[E - 1] = 0
            frirl_amd.update_sarsa(prob, agent, envs, dev(q_ant), dev(reward), dev(cur), active=dev(active))
            torch.cuda.synchronize()
            nr, st, fus_d = prob.nrules.cpu().numpy(), envs.status.cpu().numpy(), envs.fus.cpu().numpy()
            for e in range(E):
                f = fives[e]
                if not active[e]:
                    assert st[e] == frirl_amd.UPD_INACTIVE and nr[e] == f.R
                    continue
                Rb = f.R
                fus[e] = f.update_sarsa(oag, fus[e], q_ant[e], reward[e], cur[e])
                assert nr[e] == f.R and fus_d[e] == int(fus[e]), (compressed, rnd, e)
                if f.R > Rb:
                    assert st[e] == frirl_amd.UPD_INSERTED
                    assert (envs.rant[e, :, Rb].cpu().numpy() == f.rant[Rb]).all()
                    assert (prob.rb[e, :NANT, Rb].cpu().numpy() == f.veval[:, Rb]).all()
                    if compressed:
                        assert (prob.uidx[e, :, Rb].cpu().numpy().astype(np.int64) == f.uidx[:, Rb]).all()
                else:
                    assert st[e] in (frirl_amd.UPD_EXACT, frirl_amd.UPD_SPREAD, frirl_amd.UPD_SKIPPED)
                seen.add(int(st[e]))
                n = f.R
                assert rel(prob.rb[e, NANT, :n].cpu().numpy(), f.rconc[:n], 1e-9).max() <= 1e-9, (compressed, rnd, e, st[e])
    assert {frirl_amd.UPD_EXACT, frirl_amd.UPD_SPREAD, frirl_amd.UPD_INSERTED} <= seen, seen


def test_episode_steps_full_cfg3_batch():
    """BASELINE cfg3 at FULL size -- 32 768 environments x 32 768 rules, cartpole dynamics, 21 actions: K fused episode steps;
    sampled environments are followed state-for-state by the oracle (teacher-forced: the oracle's rule base is a copy of
    the device environment's), and size-independent properties are checked on ALL environments."""
    import ctypes as C
    import torch
    E, K = 32768, 4
    MAXR = R + 256
    dev0 = torch.device("cuda", 0)
    prob, agent, envs = frirl_amd.demo_batch("cartpole", E, R, MAXR, dev0, seed=21)
    sample = [0, 12345, E - 1]
    CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
    traces, oracles = {}, {}
    for e in sample:
        fr = ob.Frirl("cartpole", trig_mode=1, maxR=MAXR)
        f = fr.five
        while f.R:
            assert ob.lib().orc_remove_rule(f.h, 0) == 0
        rant = envs.rant[e, :, :R].T.contiguous().cpu().numpy()
        rconc = prob.rb[e, NANT, :R].cpu().numpy()
        for r in range(R):
            assert f.add_rule(rant[r], rconc[r]) == 0
        assert (f.veval[:, :R] == prob.rb[e, :NANT, :R].cpu().numpy()).all()
        tr = []

        def cb(frp, step, action, cur_states, cur_q, ud, tr=tr, fr=fr):
            tr.append((action, [cur_states[i] for i in range(NANT - 1)], [cur_q[i] for i in range(NANT)], fr.five.R))
        cbo = CB(cb)
        ob.lib().orc_frirl_set_max_steps(fr.h, K)
        ob.lib().orc_frirl_set_trace(fr.h, cbo)
        fr.episode()
        ob.lib().orc_frirl_set_trace(fr.h, None)
        traces[e], oracles[e] = tr, fr
    frirl_amd.episode_begin(prob, agent, envs)
    torch.cuda.synchronize()
    for t in range(K):
        nr_before = prob.nrules.clone()
        frirl_amd.episode_step(prob, agent, envs)
        torch.cuda.synchronize()
        running = envs.done == 0
        assert (envs.ep_steps[running] == t + 1).all()
        assert (prob.nrules >= nr_before).all() and (prob.nrules <= nr_before + 1).all()
        assert torch.isfinite(envs.states).all()
        for e in sample:
            if t >= len(traces[e]):
                continue
            action, cur_states, cur_q, R_before = traces[e][t]
            assert int(nr_before[e]) == R_before, (e, t)
            assert (bits(envs.states[e].cpu().numpy()) == bits(np.array(cur_states))).all(), (e, t)
            assert (bits(envs.q_ant[e].cpu().numpy()) == bits(np.array(cur_q))).all(), (e, t)
    # all environments start from the same state with different rule bases: every one took K steps unless it failed early
    assert int((envs.ep_steps == K).sum()) > E // 2
    for e in sample:
        f = oracles[e].five
        assert int(prob.nrules[e]) == f.R
        got = prob.rb[e, NANT, : f.R].cpu().numpy()
        assert rel(got, f.rconc[: f.R], 1e-9).max() <= 1e-9
        assert (prob.uidx[e, :, : f.R].cpu().numpy().astype(np.int64) == f.uidx[:, : f.R]).all()


def test_cfg5_full_rule_base_size():
    """cfg5: nant 16, U 1001 (125 KB of VE tables), 262 144 rules per rule base -- the real launch shape of the large-LDS
    index kernel (several chunks per rule base, 1024-thread workgroups) and the f64 scan, bit-exact vs the oracle."""
    import torch
    b = Batch(16, 1001, 262144, 3, A=0, seed=77, ragged=False)
    b.nrules[1] = 262144 - 3              # odd count, tail inside the last chunk
    b.rb[1, :, 262144 - 3:] = 0.0
    x = b.queries(seed=3, hit_fraction=0.0)
    x[2] = b.u[np.arange(16), b.uidx[2, :, 200001]]     # exact hit deep in a later chunk
    d_ref, hit_ref = b.oracle_rule_distance(x)
    assert hit_ref[0] == 262143 and 0 <= hit_ref[2] <= 200001 and hit_ref[1] == -1
    for compressed in (False, True):
        prob = b.to_device(compressed=compressed)
        d, hit = prob.rule_distance(dev(x))
        torch.cuda.synchronize()
        d, hit = d.cpu().numpy(), hit.cpu().numpy()
        assert (hit.astype(np.int64) == hit_ref.astype(np.int64)).all(), compressed
        for e in range(b.E):
            n = int(b.nrules[e])
            assert (bits(d[e, :n]) == bits(d_ref[e, :n])).all(), (compressed, e)


SYNTH = [(3, 41, 33, 3, 11), (5, 41, 367, 3, 12), (5, 1001, 4096, 21, 13), (8, 101, 4096, 0, 14), (5, 41, 65536, 3, 15), (3, 41, 8192, 3, 16)]


@pytest.mark.parametrize("nant,U_,R_,A_,seed", SYNTH)
def test_genuine_reference_vectors_through_hip(nant, U_, R_, A_, seed, golden_dir):
    """tests/golden/synth_*.jsonl hold FNV hashes of the distance arrays and the Q values the GENUINE reference (oracle/_ref)
    computed on seeded synthetic rule bases (incl. cfg3's table shape and cfg4's 65 536-rule base).  The HIP path gets the
    same rule base and queries: distance hashes and hit indices must equal the reference's, Q within 1e-11."""
    import torch
    with open(os.path.join(golden_dir, f"synth_n{nant}_u{U_}_r{R_}.jsonl")) as fp:
        recs = [json.loads(l) for l in fp if l.strip()]
    f = ob.synth_problem(nant, U_, R_, A_, seed)
    assert f.R == R_
    rng = seed * 77 + 5
    xs = []
    for r in recs[1:]:
        x, rng = ob.synth_query(f, rng, r["q"])
        xs.append(x)
    xs = np.array(xs)
    Q = len(xs)
    maxR = R_ + (R_ & 1) + 2
    rb = np.zeros((Q, nant + 1, maxR))
    rb[:, :nant, :R_] = f.veval[:, :R_]
    rb[:, nant, :R_] = f.rconc[:R_]
    uidx = np.zeros((Q, nant, maxR), dtype=np.int16)
    uidx[:, :, :R_] = f.uidx[:, :R_]
    nr = np.full(Q, R_, dtype=np.int32)
    for compressed in (False, True):
        prob = frirl_amd.Problem(dev(np.array(f.u)), dev(np.array(f.ve)), dev(rb), dev(nr), dev(uidx) if compressed else None)
        d, hit = prob.rule_distance(dev(xs))
        conc, hq = prob.vag_concl(dev(xs))
        torch.cuda.synchronize()
        d, hit, conc, hq = d.cpu().numpy(), hit.cpu().numpy(), conc.cpu().numpy(), hq.cpu().numpy()
        for e, r in enumerate(recs[1:]):
            assert hit[e] == r["ret"] and hq[e] == r["vc_ret"], (compressed, e)
            c = float.fromhex(r["conc"])
            if r["ret"] == -1:
                assert "%016x" % ob.hash_doubles(d[e, :R_]) == r["d_hash"], (compressed, e)
                assert rel(conc[e], c) <= RTOL
            else:
                assert conc[e] == c
