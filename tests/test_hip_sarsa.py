"""GPU parity of the SARSA update, rule append, env step and fused episode step (through the C ABI)
against the oracle.  Reference: frirl_update_sarsa.c:348-385 (+update_rules :22-143), five_add_rule.c:47-95,
examples/<env>/<env>.c callbacks, frirl_episode.c:28-194.

Bars: rule counts, branch taken, appended antecedents, sticky flag, chosen actions: BIT-EXACT.
Env dynamics (portable trig, same arithmetic on host checker and device): BIT-EXACT.
Consequents after interpolated updates: <= 1e-6 relative by contract; asserted at 1e-9.
"""
import numpy as np
import pytest

import frirl_amd
from oracle import binding as ob
from tests.problems import demo_device_batch, device_agent

pytestmark = pytest.mark.gpu
ENVS = [("mountaincar", 6), ("cartpole", 9), ("acrobot", 5)]


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rel(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-9)


@pytest.mark.parametrize("env,episodes", ENVS)
def test_update_sarsa_rounds(env, episodes):
    import torch
    E, ROUNDS = 96, 6
    b, fr = demo_device_batch(env, episodes, E)
    nant, R0 = b.nant, int(b.nrules[0])
    prob = b.to_device()
    agent = device_agent(fr)
    rant0 = torch.zeros((E, nant, b.maxR), dtype=torch.float64, device="cuda")
    rant0[:, :, :R0] = dev(np.ascontiguousarray(b.rant.T))[None]
    envs = frirl_amd.Envs(prob, "cuda", rant_init=rant0)
    oag = fr.agent()
    fives = []
    for e in range(E):
        f = ob.Five(b.u.ravel(), b.ve.ravel(), nant, b.U, b.maxR, b.rant, np.ascontiguousarray(b.rb[e, nant, :R0]))
        fives.append(f)
    fus = np.zeros(E)
    rng = np.random.default_rng(11)
    dims = [fr.dim(k) for k in range(nant)]
    seen = set()
    for rnd in range(ROUNDS):
        q_ant = np.zeros((E, nant))
        cur = np.zeros((E, nant))
        reward = np.zeros(E)
        for e in range(E):
            for k in range(nant):
                v = dims[k]["values"]
                q_ant[e, k] = v[rng.integers(len(v))]
                cur[e, k] = v[rng.integers(len(v))]
            if e % 5 == 3:      # off-grid state: Shepard spread / snapped insert
                for k in range(nant - 1):
                    v = dims[k]["values"]
                    q_ant[e, k] = rng.uniform(v[0], v[-1]) * 0.9
            reward[e] = [-10.0, 1000.0, -3000.0 * rng.random(), 20 * rng.random() - 10][e % 4]
        active = np.ones(E, dtype=np.uint8)
        active[E - 1] = 0
        frirl_amd.update_sarsa(prob, agent, envs, dev(q_ant), dev(reward), dev(cur), active=dev(active))
        torch.cuda.synchronize()
        nr = prob.nrules.cpu().numpy()
        st = envs.status.cpu().numpy()
        fus_d = envs.fus.cpu().numpy()
        rb = prob.rb.cpu().numpy()
        rant = envs.rant.cpu().numpy()
        for e in range(E):
            f = fives[e]
            if not active[e]:
                assert st[e] == frirl_amd.UPD_INACTIVE and nr[e] == f.R
                continue
            Rb = f.R
            fus[e] = f.update_sarsa(oag, fus[e], q_ant[e], reward[e], cur[e])
            assert nr[e] == f.R, (rnd, e)
            assert fus_d[e] == int(fus[e]), (rnd, e)
            if f.R > Rb:
                assert st[e] == frirl_amd.UPD_INSERTED
                assert (rant[e, :, Rb] == f.rant[Rb]).all(), "appended rule antecedents (grid-snapped)"
                assert (rb[e, :nant, Rb] == f.veval[:, Rb]).all()
            else:
                assert st[e] in (frirl_amd.UPD_EXACT, frirl_amd.UPD_SPREAD, frirl_amd.UPD_SKIPPED)
            seen.add(int(st[e]))
            n = f.R
            assert rel(rb[e, nant, :n], f.rconc[:n]).max() <= 1e-9, (rnd, e, st[e])
    assert {frirl_amd.UPD_EXACT, frirl_amd.UPD_SPREAD, frirl_amd.UPD_INSERTED} <= seen, seen


@pytest.mark.parametrize("env,episodes", ENVS)
def test_add_rule_and_capacity(env, episodes):
    import torch
    E = 8
    b, fr = demo_device_batch(env, episodes, E)
    nant, R0 = b.nant, int(b.nrules[0])
    b.maxR = R0 + 2          # room for exactly two appends
    b.rb = np.ascontiguousarray(b.rb[:, :, : b.maxR])
    prob = b.to_device()
    rng = np.random.default_rng(2)
    f = ob.Five(b.u.ravel(), b.ve.ravel(), nant, b.U, b.maxR, b.rant, np.ascontiguousarray(b.rb[0, nant, :R0]))
    store = torch.zeros((E, nant, b.maxR), dtype=torch.float64, device="cuda")
    for i in range(3):
        rant = np.array([[rng.uniform(b.u[k, 0], b.u[k, -2]) for k in range(nant)] for _ in range(E)])
        rconc = rng.normal(size=E)
        added = prob.add_rule(dev(rant), dev(rconc), rant_store=store)
        torch.cuda.synchronize()
        if i < 2:
            assert (added.cpu().numpy() == 1).all()
            assert f.add_rule(rant[0], rconc[0]) == 0
        else:
            assert (added.cpu().numpy() == 0).all(), "full rule base: append refused (the reference has no check)"
            assert f.add_rule(rant[0], rconc[0]) == -1
    assert (prob.nrules.cpu().numpy() == R0 + 2).all()
    rb = prob.rb.cpu().numpy()
    assert (rb[0, :nant, : f.R] == f.veval[:, : f.R]).all() and (rb[0, nant, : f.R] == f.rconc[: f.R]).all()
    assert (store.cpu().numpy()[0, :, R0: R0 + 2] == f.rant[R0: R0 + 2].T).all()


@pytest.mark.parametrize("env", ["mountaincar", "cartpole", "acrobot"])
def test_env_step_bit_exact(env):
    import torch
    fr = ob.Frirl(env, trig_mode=1)
    agent = device_agent(fr)
    ns = fr.nstates
    rng = np.random.default_rng(5)
    E = 4096
    s = np.zeros((E, ns))
    for k in range(ns):
        v = fr.dim(k)["values"]
        s[:, k] = rng.uniform(v[0] - 0.3 * (v[-1] - v[0]), v[-1] + 0.3 * (v[-1] - v[0]), E)
    av = fr.dim(ns)["values"]
    a = av[rng.integers(len(av), size=E)]
    new_s, rew, succ, q = frirl_amd.env_step(agent, dev(a), dev(s))
    torch.cuda.synchronize()
    new_s, rew, succ, q = new_s.cpu().numpy(), rew.cpu().numpy(), succ.cpu().numpy(), q.cpu().numpy()
    for e in range(E):
        n_o, r_o, f_o, q_o = fr.env_step(a[e], s[e])
        assert (n_o.view(np.uint64) == new_s[e].view(np.uint64)).all(), (e, n_o, new_s[e])
        assert r_o == rew[e] and f_o == succ[e]
        assert (q_o.view(np.uint64) == q[e].view(np.uint64)).all()


@pytest.mark.parametrize("env,n_episodes", [("mountaincar", 4), ("cartpole", 5), ("acrobot", 4)])
def test_fused_episode_steps_follow_oracle(env, n_episodes):
    """Whole episodes from the initial 2^nant corner rule base: every environment of the batch starts
    identically, so each must reproduce the oracle's trajectory (portable trig): per step the continuous
    state bit for bit, the chosen action, the rule count and the update branch; consequents to 1e-9."""
    import torch
    from tests.problems import Batch
    E = 5
    fr = ob.Frirl(env, trig_mode=1)
    f = fr.five
    nant, R0 = f.nant, f.R
    maxR = 1024
    b = Batch.__new__(Batch)
    b.nant, b.U, b.E, b.A, b.maxR = nant, f.U, E, fr.nactions, maxR
    b.u, b.ve = np.array(f.u), np.array(f.ve)
    b.nrules = np.full(E, R0, dtype=np.int32)
    b.rb = np.zeros((E, nant + 1, maxR))
    b.rb[:, :nant, :R0] = f.veval[:, :R0]
    prob = b.to_device()
    agent = device_agent(fr)
    rant0 = torch.zeros((E, nant, maxR), dtype=torch.float64, device="cuda")
    rant0[:, :, :R0] = dev(np.ascontiguousarray(f.rant[:R0].T))[None]
    envs = frirl_amd.Envs(prob, "cuda", rant_init=rant0)

    trace = []

    def run_oracle_episode():
        trace.clear()
        import ctypes as C
        CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)

        def cb(frp, step, action, cur_states, cur_q, ud):
            trace.append((action, [cur_states[i] for i in range(fr.nstates)], [cur_q[i] for i in range(nant)], f.R))
        cbo = CB(cb)
        ob.lib().orc_frirl_set_trace(fr.h, cbo)
        fr.episode()
        ob.lib().orc_frirl_set_trace(fr.h, None)

    for ep in range(n_episodes):
        run_oracle_episode()
        frirl_amd.episode_begin(prob, agent, envs)
        torch.cuda.synchronize()
        assert (envs.q_ant.cpu().numpy()[:, nant - 1] == trace[0][0]).all(), "first action (un-quantised default state)"
        for t, (action, cur_states, cur_q, R_before) in enumerate(trace):
            assert (prob.nrules.cpu().numpy() == R_before).all(), (ep, t)
            frirl_amd.episode_step(prob, agent, envs)
            torch.cuda.synchronize()
            st = envs.states.cpu().numpy()
            qa = envs.q_ant.cpu().numpy()
            assert (st.view(np.uint64) == np.array(cur_states).view(np.uint64)[None]).all(), (ep, t)
            assert (qa.view(np.uint64) == np.array(cur_q).view(np.uint64)[None]).all(), (ep, t, qa[0], cur_q)
        torch.cuda.synchronize()
        assert (envs.done.cpu().numpy() == 1).all()
        assert (envs.ep_steps.cpu().numpy() == fr.ep_steps).all()
        assert (envs.ep_reward.cpu().numpy() == fr.ep_reward).all()
        assert (prob.nrules.cpu().numpy() == f.R).all()
        rb = prob.rb.cpu().numpy()
        assert rel(rb[:, nant, : f.R], f.rconc[None, : f.R]).max() <= 1e-9
        assert (envs.rant.cpu().numpy()[:, :, : f.R] == f.rant[: f.R].T[None]).all()
        assert (envs.fus.cpu().numpy() == int(fr.fus)).all()


@pytest.mark.parametrize("env,E,R", [("acrobot", 300, 6000), ("acrobot", 40, 1500), ("acrobot", 5000, 3000), ("mountaincar", 300, 3000), ("cartpole", 64, 5000)])
def test_spread_candidates_equal_the_second_sweep(env, E, R, hip_option):
    """update_rules' masked write-back (frirl_update_sarsa.c:89-120) from the candidates tracked during the Q(s,a) sweep must
    give the bits of the reference-shaped second sweep (agent.debug_flags bit 0 forces it) -- every step kernel form
    (256 threads / one wave per environment / action-parallel waves), including lanes that hold MORE than two qualifying
    rules (near-duplicate rules planted in one lane's slots: the workgroup must fall back to the sweep)."""
    import torch
    dev0 = torch.device("cuda", 0)
    hip_option("step_track", 1)          # by default only rule bases > 16 K rules use the candidates (5-antecedent, <= 4-action kernels)
    outs = []
    for flags in (0, 1):
        prob, agent, envs = frirl_amd.demo_batch(env, E, R, R + 256, dev0, seed=13)
        agent.desc.debug_flags = flags
        nant = prob.nant
        # environments 0..7: rules 10, 11, 138, 139, 522, 523 become near-duplicates of one point next to the start state, so that
        # all of them carry a large Shepard weight for the first updates (slots of ONE lane in both the 64- and 256-thread kernels)
        d = frirl_amd.demo_describe(env)
        for e in range(8):
            for j, r in enumerate((10, 11, 138, 139, 522, 523)):
                for k in range(nant):
                    base_idx = int(np.argmin(np.abs(d["u"][k] - (d["values_def"][k] if k < nant - 1 else d["grids"][k][0]))))
                    idx = min(max(base_idx + (1 if (k == (j % (nant - 1))) else 0) + (1 if k == 0 else 0), 0), prob.U - 1)
                    prob.uidx[e, k, r] = idx
                    prob.rb[e, k, r] = prob.ve[k, idx]
                    envs.rant[e, k, r] = prob.u[k, idx]
                prob.rb[e, nant, r] = 100.0 + j
        frirl_amd.episode_begin(prob, agent, envs)
        seen = torch.zeros(6, dtype=torch.int64, device=dev0)
        for _ in range(6):
            frirl_amd.episode_steps(prob, agent, envs, 1)
            seen += torch.bincount(envs.status.long(), minlength=6)
        torch.cuda.synchronize()
        outs.append((prob.rb.clone(), prob.nrules.clone(), envs.states.clone(), envs.q_ant.clone(), envs.fus.clone(), seen))
    for i, (a, b) in enumerate(zip(outs[0][:5], outs[1][:5])):
        assert (a == b).all(), i
    assert (outs[0][5] == outs[1][5]).all() and int(outs[0][5][frirl_amd.UPD_SPREAD]) > 0, outs[0][5].tolist()
