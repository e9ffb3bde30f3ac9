"""GPU parity of the multi-agent rule-base merge (frirl_hip_merge_rb; reference frirl_agent.c:58-117) against the GENUINE
reference's vectors (tests/golden/merge_<env>.jsonl, produced by oracle/_ref/ref_merge_harness from frirl_agent.c compiled
with BUILD_OPENMP) and against the oracle.  Rule counts, antecedents and rule order exact; consequents within the 1e-6
contract (asserted at 1e-9: the merge multiplies interpolated values, never compares them bit for bit).

The receiver's weights array: merge_rb re-uses FIVERB.weights across sender rules, and an exact-hit sender rule leaves it
untouched (FIVEVagConclWeight.c:67-69) -- so at the START of a merge the reference still holds the weights of the agent's last
interpolated SARSA update.  The C ABI makes that array an explicit caller-owned buffer: given the reference's array state
(taken here from the oracle agent that reproduces the golden `agent_before`) the GPU reproduces the genuine `agent_after`.
The library's own loop (frirl_hip_batch_merge_round) rebuilds that state with frirl_hip_weights_from_spread from the antecedents
and rule count of each agent's last interpolated update, which the learning kernels record (frirl_hip_envs.spread_*): the C-level
merged training below follows the oracle loop with NO adjustment of the oracle's weights arrays."""
import json
import os

import numpy as np
import pytest

import frirl_amd
from oracle import binding as ob

pytestmark = pytest.mark.gpu
ENVS = ["mountaincar", "cartpole", "acrobot"]


def fha(xs):
    return np.array([float.fromhex(x) for x in xs], dtype=np.float64)


def records(env, golden_dir):
    with open(os.path.join(golden_dir, f"merge_{env}.jsonl")) as f:
        return {r["k"]: r for r in (json.loads(l) for l in f if l.strip())}


def batch_from_rules(env, rec, E, maxR, dev):
    """E copies of the golden rule base, built through five_hip_add_rule (snap + VE lookup on the device)."""
    import torch
    d = frirl_amd.demo_describe(env)
    nant, R = d["nant"], rec["R"]
    rant, rconc = fha(rec["rant"]).reshape(R, nant), fha(rec["rconc"])
    prob = frirl_amd.Problem(torch.from_numpy(d["u"]).to(dev), torch.from_numpy(d["ve"]).to(dev), torch.zeros((E, nant + 1, maxR), dtype=torch.float64, device=dev),
                             torch.zeros((E,), dtype=torch.int32, device=dev), torch.zeros((E, nant, maxR), dtype=torch.int16, device=dev))
    agent = frirl_amd.demo_agent(d, dev)
    store = torch.zeros((E, nant, maxR), dtype=torch.float64, device=dev)
    for r in range(R):
        prob.add_rule(torch.from_numpy(rant[r]).to(dev).expand(E, nant).contiguous(), torch.full((E,), rconc[r], dtype=torch.float64, device=dev), rant_store=store)
    return prob, agent, store


def check_against(prob, store, rec, E_check):
    import torch
    torch.cuda.synchronize()
    R, nant = rec["R"], prob.nant
    want_rant, want_rconc = fha(rec["rant"]).reshape(R, nant), fha(rec["rconc"])
    assert (prob.nrules.cpu().numpy()[E_check] == R).all(), (prob.nrules.tolist(), R)
    for e in E_check:
        got = store[e, :, :R].T.cpu().numpy()
        assert (got == want_rant).all(), "antecedents / rule order"
        q = prob.rb[e, nant, :R].cpu().numpy()
        rel = np.abs(q - want_rconc) / np.maximum(np.abs(want_rconc), 1e-9)
        assert rel.max() <= 1e-9, (e, rel.max())
        idx = prob.uidx[e, :, :R].long()
        assert (prob.ve.gather(1, idx) == prob.rb[e, :nant, :R]).all(), "index mirror follows the appended rules"


@pytest.mark.parametrize("env", ENVS)
def test_merge_rb_against_genuine_reference(env, golden_dir):
    import torch
    dev = torch.device("cuda", 0)
    recs = records(env, golden_dir)
    E, maxR = 5, 512
    # agent 1 <- master's rules; receivers 0..3 active, receiver 4 masked out
    prob, agent, store = batch_from_rules(env, recs["agent_before"], E, maxR, dev)
    m = recs["master_before"]
    srant = torch.from_numpy(fha(m["rant"]).reshape(m["R"], prob.nant)).to(dev).contiguous()
    srconc = torch.from_numpy(fha(m["rconc"])).to(dev)
    # the reference's FIVERB.weights at this point: replay the agent with the oracle (== golden agent_before, test_oracle_golden.py)
    m_eps, a_eps = recs["hdr"]["master_episodes"], recs["hdr"]["agent_episodes"]
    master = ob.Frirl(env)
    for _ in range(m_eps):
        master.episode()
    ag_o = ob.Frirl(env)
    ag_o.set_start_state(master.five.gen_def_states(1, 3, master.nstates))
    for _ in range(a_eps):
        ag_o.episode()
    assert ag_o.five.R == recs["agent_before"]["R"]
    weights = torch.zeros((E, maxR), dtype=torch.float64, device=dev)
    weights[:, : ag_o.five.R] = torch.from_numpy(np.array(ag_o.five.weights[: ag_o.five.R])).to(dev)
    active = torch.tensor([1, 1, 1, 1, 0], dtype=torch.uint8, device=dev)
    full = prob.merge_rb(agent, srant, srconc, weights, rant_store=store, active=active)
    check_against(prob, store, recs["agent_after"], [0, 1, 2, 3])
    assert int(prob.nrules[4]) == recs["agent_before"]["R"] and (full == 0).all()
    # master <- the merged agent's rules (one receiver)
    prob2, agent2, store2 = batch_from_rules(env, recs["master_before"], 1, maxR, dev)
    a = recs["agent_after"]
    srant2 = torch.from_numpy(fha(a["rant"]).reshape(a["R"], prob.nant)).to(dev).contiguous()
    w2 = torch.zeros((1, maxR), dtype=torch.float64, device=dev)
    w2[0, : master.five.R] = torch.from_numpy(np.array(master.five.weights[: master.five.R])).to(dev)
    prob2.merge_rb(agent2, srant2, torch.from_numpy(fha(a["rconc"])).to(dev), w2, rant_store=store2)
    check_against(prob2, store2, recs["master_after"], [0])


def test_merge_rb_capacity_and_sender_from_device_rows():
    """A full receiver refuses appends (full[e] = 1, nothing written past maxR); the sender may be a row set of the batch's own
    SoA rant store with its rule count read on the device (the second half of a merge round: master <- agent id)."""
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    d = frirl_amd.demo_describe("acrobot")
    rng = np.random.default_rng(4)
    ss = np.array([[rng.choice(d["grids"][k]) for k in range(4)] for _ in range(6)])
    prob, agent, envs = frirl_amd.demo_fresh_batch("acrobot", 6, 256, dev, start_states=torch.from_numpy(ss).to(dev))
    for _ in range(3):
        frirl_amd.episode_begin(prob, agent, envs)
        frirl_amd.episode_steps(prob, agent, envs, 250)
    torch.cuda.synchronize()
    nr = prob.nrules.cpu().numpy().copy()
    assert nr.min() > 32 and len(set(nr.tolist())) > 1, nr
    # oracle: receiver 0 takes over agent 3's rules
    fr = ob.Frirl("acrobot", trig_mode=1, maxR=256)
    f = fr.five
    while f.R:
        f.remove_rule(0)
    r0 = envs.rant[0, :, : nr[0]].T.contiguous().cpu().numpy()
    for r in range(nr[0]):
        assert f.add_rule(r0[r], float(prob.rb[0, prob.nant, r])) == 0
    s_rant = envs.rant[3, :, : nr[3]].T.contiguous().cpu().numpy()
    s_rconc = prob.rb[3, prob.nant, : nr[3]].cpu().numpy().copy()
    f.merge_rb(fr.agent(), s_rant, s_rconc)
    weights = torch.zeros((6, prob.maxR), dtype=torch.float64, device=dev)
    active = torch.tensor([1, 0, 0, 0, 0, 0], dtype=torch.uint8, device=dev)
    full = torch.zeros((6,), dtype=torch.int32, device=dev)
    snd = frirl_amd.SenderDesc(envs.rant[3].data_ptr(), 1, prob.maxR, prob.rb[3, prob.nant].data_ptr(), 0, 0, prob.nrules[3:4].data_ptr())
    frirl_amd.check(frirl_amd.lib().frirl_hip_merge_rb(C.byref(prob.tables), C.byref(prob.bases), C.byref(agent.desc), envs.rant.data_ptr(), C.byref(snd),
                                                      weights.data_ptr(), active.data_ptr(), full.data_ptr(), None), "frirl_hip_merge_rb")
    torch.cuda.synchronize()
    R = min(f.R, prob.maxR)
    assert int(prob.nrules[0]) == R and (prob.nrules[1:].cpu().numpy() == nr[1:]).all()
    assert (envs.rant[0, :, :R].T.cpu().numpy() == np.array(f.rant[:R])).all()
    q = prob.rb[0, prob.nant, :R].cpu().numpy()
    assert (np.abs(q - f.rconc[:R]) <= 1e-9 * np.maximum(np.abs(f.rconc[:R]), 1e-9)).all()
    # capacity: a receiver with 2 free slots
    small, agent_s, envs_s = frirl_amd.demo_fresh_batch("acrobot", 1, 34, dev)
    w = torch.zeros((1, 34), dtype=torch.float64, device=dev)
    srant = torch.from_numpy(np.ascontiguousarray(s_rant)).to(dev)
    full_s = small.merge_rb(agent_s, srant, torch.from_numpy(s_rconc).to(dev), w, rant_store=envs_s.rant)
    torch.cuda.synchronize()
    assert int(small.nrules[0]) == 34 and int(full_s[0]) == 1


@pytest.mark.parametrize("env,agents,max_episodes", [("mountaincar", 5, 1000), ("mountaincar", 4, 15), ("acrobot", 5, 15), ("acrobot", 5, 40), ("cartpole", 5, 15)])
def test_c_level_merged_training_follows_the_reference_loop(env, agents, max_episodes, tmp_path):
    """`frirl_demo --agents N --merge`: the reference's many-agent loop with rule-base exchange (frirl_omp_run, frirl_agent.c:294-385)
    through the C-level batch object, against tests/omp_model.py -- the same loop written with the oracle's pieces and pinned as a
    WHOLE against the genuine frirl_omp_run (tests/test_oracle_golden.py::test_omp_run_loop_matches_reference): chunks of 9 episodes,
    `epended` per chunk from the cheap test alone, finished agents running again in the next round, the previous episode's steps and
    reward surviving the merge, max_episodes looked at only when a chunk is over.  Master's episodes and rounds equal; its rule base:
    antecedents and order exact, consequents within 1e-6 (portable trig on both sides)."""
    import subprocess
    from tests import omp_model
    demo = os.path.join(frirl_amd.PKG_DIR, "lib", "frirl_demo")
    r = subprocess.run([demo, "--env", env, "--agents", str(agents), "--merge", "--max-episodes", str(max_episodes)], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    ag, rounds, pended = omp_model.run_omp(env, agents, max_episodes, trig_mode=1)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("merged ")][-1].split()
    got = {k: int(line[line.index(k) + 1]) for k in ("episodes", "merge-rounds")}
    assert got == {"episodes": ag[0].episodes_run, "merge-rounds": rounds}, (got, ag[0].episodes_run, rounds, r.stdout[-600:])
    mine = np.loadtxt(tmp_path / f"{env}.merged.frirlrb.txt", dtype=np.float64, ndmin=2)
    m = ag[0].fr.five
    R = m.R
    assert mine.shape == (R, m.nant + 1), (mine.shape, R)
    assert (mine[:, :-1] == np.array(m.rant[:R])).all()
    rel = np.abs(mine[:, -1] - m.rconc[:R]) / np.maximum(np.abs(m.rconc[:R]), 1e-9)
    assert rel.max() <= 1e-6, rel.max()
