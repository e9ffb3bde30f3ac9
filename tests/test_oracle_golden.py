"""Pins the oracle (oracle/frirl_oracle.c, the CPU checker) against

 (1) the reference's own known-answer files: tests/golden/orig/*.frirlrb.txt is a data mirror of
     the reference's tests/orig/ (final rule bases of the three demos), and
 (2) vectors produced by the genuine reference compiled in the build container
     (oracle/_ref via oracle/Makefile; generator oracle/ref_harness.c + oracle/make_golden.py).

Bars: indices, rule counts, antecedents, distances, weights, Q values: BIT-EXACT (the oracle keeps
the reference's x87 long-double pow and sequential sums).  vs tests/orig: mountaincar byte-identical;
cartpole antecedents exact + Q <= 1e-12 relative; acrobot antecedents exact only (its golden was
produced with another libm).
"""
import json
import os

import numpy as np
import pytest

from oracle import binding as ob

ENVS = ["mountaincar", "cartpole", "acrobot"]
VEC_EPISODES = {"mountaincar": 6, "cartpole": 9, "acrobot": 5}
EXPECT = {"mountaincar": (15548, 29, 110), "cartpole": (33002, 58, 182), "acrobot": (21207, 110, 367)}


def fh(x):
    return float.fromhex(x)


def fha(xs):
    return np.array([float.fromhex(x) for x in xs], dtype=np.float64)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def load_jsonl(path):
    with open(path) as f:
        return [json.loads(l) for l in f if l.strip()]


def load_rb(path):
    return np.loadtxt(path, dtype=np.float64, ndmin=2)


@pytest.fixture(scope="module")
def demo_runs(tmp_path_factory):
    out = {}
    d = tmp_path_factory.mktemp("demo")
    for e in ENVS:
        fr = ob.Frirl(e)
        conv = fr.run()
        p = str(d / f"{e}.txt")
        fr.save_text(p)
        out[e] = (fr, conv, p)
    return out


@pytest.mark.parametrize("env", ENVS)
def test_demo_matches_reference_run(env, demo_runs, golden_dir):
    fr, conv, path = demo_runs[env]
    assert conv == 1
    steps, episodes, rules = EXPECT[env]          # also the counts in reference frirl_update_sarsa.c:300-302
    assert (fr.total_steps, fr.five.R) == (steps, rules)
    with open(path) as a, open(os.path.join(golden_dir, f"ref_{env}.frirlrb.txt")) as b:
        assert a.read() == b.read(), "final rule base differs from the reference compiled in the build container"
    tr = load_jsonl(os.path.join(golden_dir, f"ref_{env}.trace.jsonl"))
    end = [r for r in tr if r["k"] == "end"][0]
    assert end["total_steps"] == steps and end["episodes"] == episodes and end["R"] == rules
    assert "%016x" % fr.step_hash == end["hash"], "whole-run step hash (every action/state/reward/q-state/rule count)"


@pytest.mark.parametrize("env", ENVS)
def test_demo_per_episode_and_first_steps(env, golden_dir):
    tr = load_jsonl(os.path.join(golden_dir, f"ref_{env}.trace.jsonl"))
    eps = [r for r in tr if r["k"] == "ep"]
    fr = ob.Frirl(env)
    for rec in eps[:25]:
        fr.episode()
        assert fr.ep_steps == rec["steps"]
        assert fr.ep_reward == fh(rec["reward"])
        assert fr.five.R == rec["R"]
        assert "%016x" % fr.step_hash == rec["hash"]
    # teacher-forced env steps of the first 400 steps
    fr2 = ob.Frirl(env)
    ns_prev, ep = None, None
    for rec in [r for r in tr if r["k"] == "step"]:
        if rec["ep"] != ep:
            ep = rec["ep"]
            ns_prev = np.array([fr2.dim(k)["values_def"] for k in range(fr2.nstates)])
        ns, r, f, q = fr2.env_step(fh(rec["a"]), ns_prev)
        assert (bits(ns) == bits(fha(rec["s"]))).all()
        assert r == fh(rec["r"]) and f == rec["f"]
        assert (bits(q) == bits(fha(rec["q"]))).all()
        ns_prev = fha(rec["s"])


@pytest.mark.parametrize("env", ENVS)
def test_demo_vs_reference_shipped_golden(env, demo_runs, golden_dir):
    fr, conv, path = demo_runs[env]
    mine = load_rb(path)
    orig = load_rb(os.path.join(golden_dir, "orig", f"frirl_example_{env}.frirlrb.txt"))
    assert mine.shape == orig.shape
    assert (mine[:, :-1] == orig[:, :-1]).all(), "antecedents / rule order"
    if env == "mountaincar":
        with open(path) as a, open(os.path.join(golden_dir, "orig", f"frirl_example_{env}.frirlrb.txt")) as b:
            assert a.read() == b.read()
    elif env == "cartpole":
        rel = np.abs(mine[:, -1] - orig[:, -1]) / np.maximum(np.abs(orig[:, -1]), 1e-300)
        assert rel.max() <= 1e-12
    # acrobot: Q column of the shipped golden is libm-era dependent (max rel 1.3e-2): not asserted


@pytest.fixture(scope="module", params=ENVS)
def vec(request, golden_dir):
    env = request.param
    recs = load_jsonl(os.path.join(golden_dir, f"vec_{env}.jsonl"))
    fr = ob.Frirl(env)
    fr.run(max_episodes=VEC_EPISODES[env] + 1)
    by = {}
    for r in recs:
        by.setdefault(r["k"], []).append(r)
    return env, fr, by


def test_vec_tables_and_rulebase(vec):
    env, fr, by = vec
    t = by["tables"][0]
    f = fr.five
    assert (f.nant, f.U, fr.nactions, f.c.p, f.R) == (t["nant"], t["U"], t["A"], t["p"], t["R"])
    assert "%016x" % ob.hash_doubles(f.u) == t["u_hash"]
    assert "%016x" % ob.hash_doubles(f.ve) == t["ve_hash"]
    assert (bits(np.array(f.c.udivs[: f.nant])) == bits(fha(t["udivs"]))).all()
    assert (bits(fr.action_vevalues) == bits(fha(t["vevalues"]))).all()
    if "u" in t:
        assert (bits(f.u.ravel()) == bits(fha(t["u"]))).all() and (bits(f.ve.ravel()) == bits(fha(t["ve"]))).all()
    rb = by["rb"][0]
    assert (bits(f.rant[: f.R].ravel()) == bits(fha(rb["rant"]))).all()
    assert (bits(f.rconc[: f.R]) == bits(fha(rb["rconc"]))).all()


def test_vec_snap(vec):
    env, fr, by = vec
    s = by["snap"][0]
    f = fr.five
    n = f.nant
    pts = fha(s["pts"]).reshape(-1, n)
    idx = np.array(s["idx"], dtype=np.int64).reshape(-1, n)
    L = ob.lib()
    u = np.ascontiguousarray(f.u)
    for q in range(pts.shape[0]):
        for k in range(n):
            row = u[k]
            got = L.orc_snap(ob.dp(row), f.U, pts[q, k], f.c.udivs[k])
            assert got == idx[q, k], (q, k, pts[q, k])


def test_vec_check_possible_states(vec):
    env, fr, by = vec
    L = ob.lib()
    for r in by["cps"]:
        obs, out = fha(r["obs"]), fha(r["out"])
        for k in range(fr.nant):
            v = np.ascontiguousarray(fr.dim(k)["values"])
            got = L.orc_check_possible_states(obs[k], ob.dp(v), len(v))
            assert got == out[k]


def test_vec_env_steps(vec):
    env, fr, by = vec
    for r in by["env"]:
        ns, rew, f, q = fr.env_step(fh(r["a"]), fha(r["s"]))
        assert (bits(ns) == bits(fha(r["ns"]))).all(), r
        assert rew == fh(r["r"]) and f == r["f"]
        assert (bits(q) == bits(fha(r["q"]))).all()


def test_vec_five(vec):
    env, fr, by = vec
    f = fr.five
    R = f.R
    nhit = nmiss = 0
    for r in by["five"]:
        x = fha(r["x"])
        ret = f.rule_distance(x)
        assert ret == r["ret"]
        if ret == -1:
            nmiss += 1
            assert "%016x" % ob.hash_doubles(f.ruledists[:R]) == r["d_hash"]
            d = fha(r["d"])
            assert (bits(f.ruledists[: len(d)]) == bits(d)).all()
        else:
            nhit += 1
        h, conc = f.vag_concl(x)
        assert h == r["vc_ret"]
        assert bits(np.array([conc]))[0] == bits(np.array([fh(r["conc"])]))[0]
        hw = f.vag_concl_weight(x)
        assert hw == r["w_ret"]
        if hw == -1:
            assert "%016x" % ob.hash_doubles(f.weights[:R]) == r["w_hash"]
            w = fha(r["w"])
            assert (bits(f.weights[: len(w)]) == bits(w)).all()
    assert nhit > 20 and nmiss > 20


def test_vec_get_best_action(vec):
    env, fr, by = vec
    for r in by["gba"]:
        best = fr.get_best_action(fha(r["s"]))
        assert best == r["best"]
        assert (bits(fr.actconc) == bits(fha(r["actconc"]))).all()


def test_vec_update_sarsa_sequence(vec):
    env, fr, by = vec          # must run after the read-only tests above: it mutates the rule base
    f = fr.five
    inserted = 0
    for i, r in enumerate(by["sarsa"]):
        assert int(fr.fus) == r["fus_before"]
        R0 = f.R
        fr.update_sarsa(fha(r["q_ant"]), fh(r["reward"]), fha(r["cur_q_ant"]))
        assert f.R == r["R"], i
        inserted += f.R - R0
        assert int(fr.fus) == r["fus_after"], i
        assert "%016x" % ob.hash_doubles(f.rconc[: f.R]) == r["rconc_hash"], i
        assert "%016x" % ob.hash_doubles(f.rant[: f.R]) == r["rant_hash"], i
        if "rconc" in r:
            assert (bits(f.rconc[: f.R]) == bits(fha(r["rconc"]))).all()
    assert inserted > 5


SYNTH = [(3, 41, 33, 3, 11), (5, 41, 367, 3, 12), (5, 1001, 4096, 21, 13), (8, 101, 4096, 0, 14), (5, 41, 65536, 3, 15),
         (3, 41, 8192, 3, 16)]


@pytest.mark.parametrize("nant,U,R,A,seed", SYNTH)
def test_synth_bases(nant, U, R, A, seed, golden_dir):
    recs = load_jsonl(os.path.join(golden_dir, f"synth_n{nant}_u{U}_r{R}.jsonl"))
    hdr = recs[0]
    f = ob.synth_problem(nant, U, R, A, seed)
    assert f.R == R, "synthetic rule base must be duplicate-free and fully inserted"
    assert "%016x" % ob.hash_doubles(f.veval[nant - 1, :R]) == hdr["veval_hash"]
    rng = seed * 77 + 5
    hits = 0
    for r in recs[1:]:
        x, rng = ob.synth_query(f, rng, r["q"])
        ret = f.rule_distance(x)
        assert ret == r["ret"]
        if ret == -1:
            assert "%016x" % ob.hash_doubles(f.ruledists[:R]) == r["d_hash"]
        else:
            hits += 1
        h, conc = f.vag_concl(x)
        assert h == r["vc_ret"] and conc == fh(r["conc"])
        if f.vag_concl_weight(x) == -1:
            assert "%016x" % ob.hash_doubles(f.weights[:R]) == r["w_hash"]
    assert hits >= 1


def test_portable_trig_accuracy():
    """orc_sin/orc_cos (the FMA-free trig shared with the HIP env kernels) vs libm: <= 2 ulp on the
    ranges the three environments produce (|x| <= ~30)."""
    L = ob.lib()
    rng = np.random.default_rng(7)
    xs = np.concatenate([rng.uniform(-30, 30, 20000), rng.uniform(-1e-3, 1e-3, 2000), np.linspace(-3.2, 3.2, 5001)])
    worst = 0.0
    for x in xs:
        for mine, ref in ((L.orc_sin(x), np.sin(x)), (L.orc_cos(x), np.cos(x))):
            ulp = np.spacing(abs(ref)) if ref != 0 else 5e-324
            worst = max(worst, abs(mine - ref) / ulp if abs(ref) > 1e-3 else abs(mine - ref) / 2.2e-19)
    assert worst <= 2.0, worst


@pytest.mark.parametrize("env", ENVS)
def test_portable_trig_demo_converges_identically(env):
    """With the portable trig the three demos learn the same rule-base size in the same number of steps."""
    fr = ob.Frirl(env, trig_mode=1)
    assert fr.run() == 1
    assert (fr.total_steps, fr.five.R) == EXPECT[env][0:1] + EXPECT[env][2:3]


@pytest.mark.parametrize("env", ENVS)
@pytest.mark.parametrize("strategy", [1, 2])
def test_reduction_matches_reference(env, strategy, golden_dir, tmp_path):
    """Rule-base reduction (reference frirl_sequential_run.c:170-350) after construction: byte-identical dump."""
    fr = ob.Frirl(env)
    assert fr.run() == 1
    fr.reduce(strategy)
    p = str(tmp_path / "red.txt")
    fr.save_text(p)
    with open(p) as a, open(os.path.join(golden_dir, f"ref_{env}.reduced{strategy}.frirlrb.txt")) as b:
        assert a.read() == b.read()


MERGE_EPISODES = {"mountaincar": (8, 3), "cartpole": (10, 4), "acrobot": (6, 3)}


def merge_records(env, golden_dir):
    recs = {r["k"]: r for r in load_jsonl(os.path.join(golden_dir, f"merge_{env}.jsonl"))}
    assert (recs["hdr"]["master_episodes"], recs["hdr"]["agent_episodes"]) == MERGE_EPISODES[env]
    return recs


def same_rule_base(five, rec, nant):
    R = rec["R"]
    assert five.R == R, (five.R, R)
    assert (bits(five.rant[:R]) == bits(fha(rec["rant"]).reshape(R, nant))).all(), "antecedents / rule order"
    assert (bits(five.rconc[:R]) == bits(fha(rec["rconc"]))).all(), "consequents"


@pytest.mark.parametrize("env", ENVS)
def test_merge_rb_matches_reference(env, golden_dir):
    """merge_rb / gen_def_states / omp_init of the GENUINE reference (frirl_agent.c compiled with BUILD_OPENMP by
    oracle/Makefile, driven by oracle/ref_merge_harness.c): a master after M episodes, agent 1 of a world of 3 (fresh rule
    base, start state moved to the master's first rule) after K episodes, then agent <- master and master <- agent.  The
    oracle's restatement reproduces every rule base bit for bit."""
    recs = merge_records(env, golden_dir)
    m_eps, a_eps = MERGE_EPISODES[env]
    master = ob.Frirl(env)
    for _ in range(m_eps):
        master.episode()
    nant = master.five.nant
    same_rule_base(master.five, recs["master_before"], nant)
    start = master.five.gen_def_states(1, 3, master.nstates)
    assert (bits(start) == bits(fha(recs["agent_start"]["values_def"]))).all()
    assert master.five.gen_def_states(0, 3, master.nstates) is None
    agent = ob.Frirl(env)
    agent.set_start_state(start)
    for _ in range(a_eps):
        agent.episode()
    same_rule_base(agent.five, recs["agent_before"], nant)
    Rm = master.five.R
    agent.five.merge_rb(agent.agent(), np.array(master.five.rant[:Rm]), np.array(master.five.rconc[:Rm]))
    same_rule_base(agent.five, recs["agent_after"], nant)
    Ra = agent.five.R
    master.five.merge_rb(master.agent(), np.array(agent.five.rant[:Ra]), np.array(agent.five.rconc[:Ra]))
    same_rule_base(master.five, recs["master_after"], nant)
    assert recs["agent_after"]["R"] > recs["agent_before"]["R"], "the merge inserted rules"


# ---- the batched (device) environments use the portable trig: what that costs against the genuine reference, stated ----------
# glibc's sin/cos cannot be reproduced on the device (x86-64 glibc dispatches to FMA-compiled variants by CPU) and
# acrobot is chaotic in the last bit of cos / sin, so the throughput path (oracle trig_mode = 1 == csrc/envs.h, bit for bit) learns
# the SAME rules in the SAME order and number of steps as the reference, with drifting consequents on acrobot only.  The drop-in
# path (host callbacks, glibc) stays within 1e-6 on all three demos (tests/test_dropin.py).
PORTABLE_TRIG_Q_BOUND = {          # env: (max abs, max rel, median rel) of Q against tests/golden/ref_<env>.frirlrb.txt
    "mountaincar": (0.0, 0.0, 0.0),                   # identical (measured: 0 of 110 consequents differ)
    "cartpole": (1e-11, 1e-12, 1e-15),                # measured 3.6e-12 abs, 3.4e-15 rel on 29 of 182 consequents
    "acrobot": (0.06, 0.2, 5e-4),                     # measured 0.030 abs, 0.109 rel, median 1.4e-4 on 360 of 367 consequents
}


@pytest.mark.parametrize("env", ENVS)
def test_portable_trig_run_against_the_genuine_reference_rule_base(env, golden_dir):
    """Whole construct run with the device's trig vs the rule base the genuine reference (glibc) wrote: antecedents, rule order,
    rule count and total steps exact for all three demos; consequents within the bound stated above."""
    ref = np.loadtxt(os.path.join(golden_dir, f"ref_{env}.frirlrb.txt"))
    fr = ob.Frirl(env, trig_mode=1)
    assert fr.run() == 1
    R = fr.five.R
    assert (fr.total_steps, R) == EXPECT[env][0:1] + EXPECT[env][2:3] and ref.shape == (R, fr.nant + 1)
    assert (np.array(fr.five.rant[:R]) == ref[:, :-1]).all(), "antecedents / rule order"
    q, qr = np.array(fr.five.rconc[:R]), ref[:, -1]
    d = np.abs(q - qr)
    rel = d / np.maximum(np.abs(qr), 1e-9)
    max_abs, max_rel, med_rel = PORTABLE_TRIG_Q_BOUND[env]
    assert d.max() <= max_abs and rel.max() <= max_rel and np.median(rel) <= med_rel, (d.max(), rel.max(), np.median(rel))


def test_portable_trig_env_steps_that_differ_from_glibc(golden_dir):
    """How often one environment step of the portable trig differs from the reference's (glibc) step on the reference's own
    env-step vectors (measured: 36 of 900 steps, i.e. one step in 25 -- enough for a chaotic system to leave the reference's
    trajectory within an episode), and by how much: <= 1e-13 absolute in any state component (a last-bit effect, amplified only
    where a component cancels to near zero); reward, end flag and the quantised observation never differ."""
    worst, differing, total = 0.0, 0, 0
    for env in ENVS:
        fr = ob.Frirl(env, trig_mode=1)
        for line in open(os.path.join(golden_dir, f"vec_{env}.jsonl")):
            r = json.loads(line)
            if r["k"] != "env":
                continue
            ns, rew, f, q = fr.env_step(fh(r["a"]), fha(r["s"]))
            want = fha(r["ns"])
            total += 1
            if (bits(ns) != bits(want)).any():
                differing += 1
                worst = max(worst, float(np.abs(ns - want).max()))
            assert rew == fh(r["r"]) and f == r["f"] and (bits(q) == bits(fha(r["q"]))).all()
    assert total == 900 and differing <= 60 and worst <= 1e-13, (differing, total, worst)


# ---- the many-agent loop WITH rule-base exchange, as a whole: frirl_omp_run of the genuine reference (5 agents) -----------------
OMPRUN = [("mountaincar", 1000), ("cartpole", 15), ("cartpole", 40), ("acrobot", 15), ("acrobot", 40)]


@pytest.mark.parametrize("env,max_episodes", OMPRUN)
def test_omp_run_loop_matches_reference(env, max_episodes, golden_dir):
    """tests/omp_model.py (chunks of 9 episodes through frirl_sequential_run's loop, `epended` per chunk from the cheap test alone,
    finished agents running again, previous-episode values surviving the merge; frirl_agent.c:294-385) against the GENUINE
    frirl_omp_run compiled with BUILD_OPENMP (oracle/ref_merge_harness `omprun`): the master's final rule base bit for bit, the
    number of episodes all agents ran and the number of exchanges skipped because an agent had "pended"."""
    from tests import omp_model
    recs = {r["k"]: r for r in load_jsonl(os.path.join(golden_dir, f"omprun_{env}_{max_episodes}.jsonl"))}
    assert (recs["hdr"]["world"], recs["hdr"]["max_episodes"]) == (5, max_episodes)
    ag, rounds, pended = omp_model.run_omp(env, 5, max_episodes, trig_mode=0)
    same_rule_base(ag[0].fr.five, recs["master_final"], ag[0].fr.nant)
    assert sum(a.episodes_run for a in ag) == recs["stdout_counts"]["episode_lines"]
    assert pended == recs["stdout_counts"]["pended_lines"]
    assert rounds >= 1
