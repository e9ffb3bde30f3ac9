"""Parity at BASELINE.json's full size (cfg2: 8192 rules x 8192 environments, mountaincar tables) through
size-independent properties on ALL environments plus the oracle on a sample of them."""
import numpy as np
import pytest

import frirl_amd
from oracle import binding as ob

pytestmark = pytest.mark.gpu
E, R, MAXR = 8192, 8192, 8448


@pytest.fixture(scope="module")
def batch():
    import torch
    return frirl_amd.demo_batch("mountaincar", E, R, MAXR, torch.device("cuda", 0), seed=3)


def oracle_agent_for_env(prob, envs, e, nrules):
    """orc_frirl (portable trig) whose rule base is a copy of device environment e."""
    fr = ob.Frirl("mountaincar", trig_mode=1, maxR=MAXR)
    f = fr.five
    while f.R:
        assert ob.lib().orc_remove_rule(f.h, 0) == 0
    rant = envs.rant[e, :, :nrules].T.contiguous().cpu().numpy()
    rconc = prob.rb[e, prob.nant, :nrules].cpu().numpy()
    for r in range(nrules):
        assert f.add_rule(rant[r], rconc[r]) == 0
    assert (f.veval[:, :nrules] == prob.rb[e, : prob.nant, :nrules].cpu().numpy()).all()
    return fr


def test_q_kernels_properties_full_size(batch):
    import torch
    prob, agent, envs = batch
    nant = prob.nant
    g = torch.Generator(device="cuda").manual_seed(5)
    ar = torch.arange(E, device="cuda")
    # (i) exact hit: each environment queries one of its own rules -> Q is that rule's consequent (lowest duplicate)
    pick = torch.randint(0, R, (E,), generator=g, device="cuda")
    x = envs.rant[ar, :, pick].contiguous()
    conc, hit = prob.vag_concl(x)
    torch.cuda.synchronize()
    hit = hit.long()
    assert (hit >= 0).all() and (hit <= pick).all()
    assert (conc == prob.rb[ar, nant, hit]).all()
    # (ii) partition of unity: with all consequents equal, every interpolated Q equals that constant
    lo, hi = prob.u[:, 0], prob.u[:, prob.U - 2]
    xc = (lo + (hi - lo) * torch.rand((E, nant), generator=g, device="cuda", dtype=torch.float64)).contiguous()
    saved = prob.rb[:, nant, :].clone()
    prob.rb[:, nant, :R] = 3.5
    conc2, hit2 = prob.vag_concl(xc)
    w, hitw = prob.vag_concl_weight(xc)
    torch.cuda.synchronize()
    miss = hit2 < 0
    assert miss.sum() > E // 2
    assert ((conc2[miss] - 3.5).abs() <= 1e-12).all() and (conc2[~miss] == 3.5).all()
    assert (hitw == hit2).all()
    ws = w[miss][:, :R]
    assert (ws >= 0).all() and ((ws.sum(dim=1) - 1.0).abs() <= 1e-12).all()
    prob.rb[:, nant, :] = saved
    # (iii) greedy sweep == per-action Q: actconc[e][a] equals vag_concl on (state, action a)
    states = xc[:, : nant - 1].contiguous()
    actconc, best = prob.get_best_action(states, agent.action_ve)
    avals = agent.grid_values[nant - 1, : agent.A]
    for a in range(agent.A):
        xa = torch.cat([states, avals[a].expand(E, 1)], dim=1).contiguous()
        ca, _ = prob.vag_concl(xa)
        torch.cuda.synchronize()
        rel = (actconc[:, a] - ca).abs() / ca.abs().clamp_min(1e-9)
        assert rel.max() <= 1e-12
    assert (best.long() == actconc.argmax(dim=1)).all() or (actconc.gather(1, best.long()[:, None])[:, 0] == actconc.max(dim=1).values).all()
    # (iv) a sample of environments against the oracle
    xs = xc.cpu().numpy()
    c2 = conc2  # not used further
    conc3, hit3 = prob.vag_concl(xc)
    torch.cuda.synchronize()
    for e in (0, 1, 4095, 8191):
        fr = oracle_agent_for_env(prob, envs, e, R)
        h, c = fr.five.vag_concl(xs[e])
        assert h == int(hit3[e])
        assert abs(float(conc3[e]) - c) <= 1e-11 * max(abs(c), 1e-9) if h < 0 else float(conc3[e]) == c


def test_episode_steps_full_size(batch):
    import torch
    prob, agent, envs = batch
    nant, K = prob.nant, 10
    sample = [0, 7, 8191]
    oracles = {e: oracle_agent_for_env(prob, envs, e, R) for e in sample}
    traces = {}
    import ctypes as C
    CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
    for e, fr in oracles.items():
        tr = []

        def cb(frp, step, action, cur_states, cur_q, ud, tr=tr, fr=fr):
            tr.append((action, [cur_states[i] for i in range(nant - 1)], [cur_q[i] for i in range(nant)], fr.five.R))
        cbo = CB(cb)
        ob.lib().orc_frirl_set_max_steps(fr.h, K)
        ob.lib().orc_frirl_set_trace(fr.h, cbo)
        fr.episode()
        ob.lib().orc_frirl_set_trace(fr.h, None)
        traces[e] = tr
    frirl_amd.episode_begin(prob, agent, envs)
    torch.cuda.synchronize()
    for t in range(K):
        nr_before = prob.nrules.clone()
        frirl_amd.episode_step(prob, agent, envs)
        torch.cuda.synchronize()
        assert (envs.ep_steps == t + 1).all() and (envs.done == 0).all()
        assert (prob.nrules >= nr_before).all() and (prob.nrules <= nr_before + 1).all()
        assert torch.isfinite(envs.states).all() and (envs.status > 0).all()
        for e in sample:
            action, cur_states, cur_q, R_before = traces[e][t]
            assert int(nr_before[e]) == R_before, (e, t)
            assert (envs.states[e].cpu().numpy().view(np.uint64) == np.array(cur_states).view(np.uint64)).all(), (e, t)
            assert (envs.q_ant[e].cpu().numpy().view(np.uint64) == np.array(cur_q).view(np.uint64)).all(), (e, t)
    for e in sample:
        f = oracles[e].five
        assert int(prob.nrules[e]) == f.R
        dev = prob.rb[e, nant, : f.R].cpu().numpy()
        assert (np.abs(dev - f.rconc[: f.R]) <= 1e-9 * np.maximum(np.abs(f.rconc[: f.R]), 1e-9)).all()
    # every environment saw the same dynamics (same start state, mountaincar physics independent of the rule base
    # only through the chosen actions): rewards are -10 per step until success
    assert (envs.ep_reward == -10.0 * K).all()


def test_cfg4_shape_sample_against_oracle():
    """BASELINE cfg4 shape (acrobot tables, nant 5, 65536 rules per environment) on a reduced number of environments:
    distances and hit indices bit-exact vs the oracle, Q values within tolerance, for every environment."""
    import torch
    En, Rn = 12, 65536
    dev = torch.device("cuda", 0)
    prob, agent, envs = frirl_amd.demo_batch("acrobot", En, Rn, Rn + 256, dev, seed=9)
    nant = prob.nant
    g = torch.Generator(device="cuda").manual_seed(2)
    lo, hi = prob.u[:, 0], prob.u[:, prob.U - 2]
    x = (lo + (hi - lo) * torch.rand((En, nant), generator=g, device="cuda", dtype=torch.float64)).contiguous()
    x[0] = envs.rant[0, :, Rn - 1]                       # exact hit on the very last rule (or an earlier duplicate)
    x[1] = envs.rant[1, :, 12345]
    d, hit = prob.rule_distance(x)
    conc, hitq = prob.vag_concl(x)
    torch.cuda.synchronize()
    assert (hit == hitq).all() and int(hit[0]) >= 0 and int(hit[1]) >= 0
    u, ve = prob.u.cpu().numpy(), prob.ve.cpu().numpy()
    xs = x.cpu().numpy()
    for e in range(En):
        rant = envs.rant[e, :, :Rn].T.contiguous().cpu().numpy()
        rconc = prob.rb[e, nant, :Rn].cpu().numpy()
        f = ob.Five(u.ravel(), ve.ravel(), nant, prob.U, Rn + 8, rant, rconc)
        h = f.rule_distance(xs[e])
        assert h == int(hit[e]), e
        assert (d[e, :Rn].cpu().numpy().view(np.uint64) == np.array(f.ruledists[:Rn]).view(np.uint64)).all(), e
        hq, c = f.vag_concl(xs[e])
        assert hq == int(hitq[e])
        assert float(conc[e]) == c if hq >= 0 else abs(float(conc[e]) - c) <= 1e-10 * max(abs(c), 1e-9)


def test_run_to_run_determinism():
    """No float atomics anywhere: two runs of the same fused steps from the same state give bit-identical rule bases."""
    import torch
    dev = torch.device("cuda", 0)
    outs = []
    for _ in range(2):
        prob, agent, envs = frirl_amd.demo_batch("acrobot", 512, 4096, 4352, dev, seed=4)
        frirl_amd.episode_begin(prob, agent, envs)
        frirl_amd.episode_steps(prob, agent, envs, 12)
        torch.cuda.synchronize()
        outs.append((prob.rb.clone(), prob.nrules.clone(), envs.states.clone(), envs.status.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert (a == b).all()
    assert (outs[0][3] == frirl_amd.UPD_SPREAD).any() or (outs[0][3] == frirl_amd.UPD_INSERTED).any()


def test_compressed_index_path_is_bit_identical_to_f64_path(hip_option):
    """The 16-bit index mirror (frirl_hip_rulebases.uidx + LDS tables) feeds the kernels the SAME doubles as the f64
    columns (rb[k][r] == ve[k][uidx[k][r]]): every result must be bit-identical between the two paths, and appends
    must keep the mirror in sync."""
    import torch
    dev = torch.device("cuda", 0)
    En, Rn = 512, 4096

    def run(no_uidx):
        hip_option("no_uidx", 1 if no_uidx else 0)
        prob, agent, envs = frirl_amd.demo_batch("acrobot", En, Rn, Rn + 256, dev, seed=11)
        g = torch.Generator(device="cuda").manual_seed(3)
        lo, hi = prob.u[:, 0], prob.u[:, prob.U - 2]
        x = (lo + (hi - lo) * torch.rand((En, prob.nant), generator=g, device="cuda", dtype=torch.float64)).contiguous()
        x[:64] = envs.rant[:64, :, 77]
        d, hit = prob.rule_distance(x)
        conc, hitq = prob.vag_concl(x)
        w, hitw = prob.vag_concl_weight(x)
        actconc, best = prob.get_best_action(x[:, : prob.nant - 1].contiguous(), agent.action_ve)
        frirl_amd.episode_begin(prob, agent, envs)
        frirl_amd.episode_steps(prob, agent, envs, 8)
        torch.cuda.synchronize()
        return prob, envs, [d[:, :Rn].clone(), hit.clone(), conc.clone(), hitq.clone(), torch.nan_to_num(w[:, :Rn], nan=-1.0), hitw.clone(), actconc.clone(),
                            best.clone(), prob.rb.clone(), prob.nrules.clone(), envs.states.clone(), envs.status.clone()]

    prob_c, envs_c, a = run(no_uidx=False)
    _, _, b = run(no_uidx=True)
    for i, (x, y) in enumerate(zip(a, b)):
        assert (x == y).all(), i
    assert (prob_c.nrules > Rn).any(), "some environments appended rules"
    # mirror consistency after appends: rb[e][k][r] == ve[k][uidx[e][k][r]] for every live rule
    idx = prob_c.uidx.long()
    for k in range(prob_c.nant):
        gathered = prob_c.ve[k][idx[:, k, :]]
        live = torch.arange(prob_c.maxR, device=dev)[None, :] < prob_c.nrules[:, None]
        assert (gathered[live] == prob_c.rb[:, k, :][live]).all(), k
