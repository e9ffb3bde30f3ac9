"""Persistent construct loop (frirl_hip_learn_run, csrc/learn.hip): many agents, every one running its episodes back to back on
the device until its rule base is "considered complete" (frirl_sequential_run.c:55-165), against the oracle's whole runs --
replicas of the demo AND agents with diversified start states (frirl_agent.c:121-139), which are never in step."""
import numpy as np
import pytest

import frirl_amd
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def check_agent_against_oracle(fr, prob, envs, e):
    f = fr.five
    R = f.R
    assert int(prob.nrules[e]) == R, (e, int(prob.nrules[e]), R)
    rant = envs.rant[e, :, :R].cpu().numpy()
    assert (rant == f.rant[:R].T).all(), f"agent {e}: antecedents / rule order"
    rb = prob.rb[e].cpu().numpy()
    assert (rb[: f.nant, :R] == f.veval[:, :R]).all(), f"agent {e}: VE columns"
    q = rb[prob.nant, :R]
    rel = np.abs(q - f.rconc[:R]) / np.maximum(np.abs(f.rconc[:R]), 1e-9)
    assert rel.max() <= 1e-9, (e, rel.max())
    ui = prob.uidx[e, :, :R].cpu().numpy().astype(np.int64) & 0xFFFF
    assert (ui == f.uidx[:, :R]).all(), f"agent {e}: index mirror"


@pytest.mark.parametrize("slices", [0, 1, 2, 4, 8, 16, 32, 64])
@pytest.mark.parametrize("env,episodes,steps,rules", [("mountaincar", 29, 15548, 110), ("acrobot", 110, 21207, 367), ("cartpole", 58, 33002, 182)])
def test_persistent_training_reaches_the_oracle_rule_base(env, episodes, steps, rules, slices, hip_option):
    """E = 21 replicas of the demo (ragged last wave), launches of 700 steps (every agent is stopped and resumed in the middle of
    episodes many times): episodes, total steps, rule count, antecedents and order exact, Q <= 1e-9 -- for every lane-group size."""
    import torch
    E = 21
    dev = torch.device("cuda", 0)
    if slices:
        hip_option("learn_slices", slices)
    fr = ob.Frirl(env, trig_mode=1)
    assert fr.run() == 1 and fr.five.R == rules and fr.total_steps == steps
    prob, agent, envs = frirl_amd.demo_fresh_batch(env, E, 512, dev)
    assert frirl_amd.learn_supported(prob, agent)
    run = frirl_amd.train_persistent(prob, agent, envs, budget=700)
    torch.cuda.synchronize()
    conv = run.conv
    assert (conv.converged == 1).all()
    assert (conv.episodes == episodes).all(), conv.episodes.tolist()
    assert (run.steps_total == steps).all(), run.steps_total.tolist()
    assert run.launches >= steps // 700
    for e in (0, 7, E - 1):
        check_agent_against_oracle(fr, prob, envs, e)
    # the counted work: one visit per rule and fused sweep (episode starts included), identical for identical agents
    w = run.work.cpu().numpy()
    assert (w == w[0]).all() and w[0, 0] > steps * 8 and w[0, 1] > 0


@pytest.mark.parametrize("env,max_episodes", [("mountaincar", 120), ("acrobot", 60), ("cartpole", 40)])
def test_persistent_training_with_diversified_start_states(env, max_episodes):
    """Per-agent start states on the state grid: agents converge after very different numbers of episodes (or not at all within
    max_episodes), the live list shrinks from launch to launch and the lane-group size changes with it; every sampled agent ends
    exactly where the oracle's run from the same start state ends."""
    import torch
    dev = torch.device("cuda", 0)
    E = 40
    d = frirl_amd.demo_describe(env)
    rng = np.random.default_rng(11)
    start = np.stack([rng.choice(d["grids"][k], E) for k in range(d["nstates"])], 1)
    prob, agent, envs = frirl_amd.demo_fresh_batch(env, E, 512, dev, start_states=torch.from_numpy(np.ascontiguousarray(start)).to(dev))
    run = frirl_amd.train_persistent(prob, agent, envs, max_episodes=max_episodes, budget=1500)
    torch.cuda.synchronize()
    conv = run.conv
    episodes, converged, total = conv.episodes.cpu().numpy(), conv.converged.cpu().numpy(), run.steps_total.cpu().numpy()
    assert len(set(total.tolist())) > 3, "the start states were meant to desynchronise the agents"
    for e in range(0, E, 3):
        fr = ob.Frirl(env, trig_mode=1, maxR=512)
        fr.set_start_state(start[e])
        ok = fr.run(max_episodes=max_episodes)
        assert ok == int(converged[e]), (e, ok, converged[e])
        assert fr.total_steps == total[e], (e, fr.total_steps, total[e])
        check_agent_against_oracle(fr, prob, envs, e)
    assert ((converged == 1) | (episodes == max_episodes - 1)).all()


@pytest.mark.parametrize("env,max_episodes,off_grid", [("mountaincar", 14, False), ("acrobot", 8, True), ("cartpole", 6, True)])
def test_training_loop_on_the_device_with_thousands_of_agents(env, max_episodes, off_grid):
    """3 000 diversified agents through frirl_hip_learn_train (csrc/learn.hip): the queue of live agents, the counting sort by rule
    count and the compaction between launches run on the device, every launch under a work budget.  Sampled agents end exactly where
    the oracle's runs from the same start states end -- also for start states OFF the state grid (the first update of an episode then
    takes the possible-state search, :146-170; later ones use the carried grid point)."""
    import torch
    dev = torch.device("cuda", 0)
    E = 3000
    d = frirl_amd.demo_describe(env)
    rng = np.random.default_rng(5)
    start = np.stack([rng.choice(d["grids"][k], E) for k in range(d["nstates"])], 1)
    if off_grid:
        step = np.array([d["grids"][k][1] - d["grids"][k][0] for k in range(d["nstates"])])
        start = start + rng.uniform(-0.3, 0.3, start.shape) * step
    prob, agent, envs = frirl_amd.demo_fresh_batch(env, E, 512, dev, start_states=torch.from_numpy(np.ascontiguousarray(start)).to(dev))
    seen = []
    run = frirl_amd.train_persistent(prob, agent, envs, max_episodes=max_episodes, budget=400, on_chunk=lambda i, live, conv: seen.append(int(live.numel())))
    torch.cuda.synchronize()
    assert seen[0] == E and run.launches == len(seen) and sorted(seen, reverse=True) == seen
    conv = run.conv
    episodes, converged, total = conv.episodes.cpu().numpy(), conv.converged.cpu().numpy(), run.steps_total.cpu().numpy()
    assert ((converged == 1) | (episodes == max_episodes - 1)).all()
    nr = prob.nrules.cpu().numpy()
    sample = sorted({0, E - 1, int(np.argmax(nr)), int(np.argmin(nr)), 1234, 2047, 2048})      # small and large rule bases
    for e in sample:
        fr = ob.Frirl(env, trig_mode=1, maxR=512)
        fr.set_start_state(start[e])
        ok = fr.run(max_episodes=max_episodes)
        assert ok == int(converged[e]), (e, ok, converged[e])
        assert fr.total_steps == total[e], (e, fr.total_steps, total[e])
        check_agent_against_oracle(fr, prob, envs, e)


@pytest.mark.parametrize("env", ["mountaincar", "acrobot", "cartpole"])
def test_launches_of_a_few_steps_change_nothing(env):
    """A work budget of 5 steps per launch: thousands of launches, every agent stopped and resumed every few steps -- the carried pending
    point, the spread flags' bound and the on-grid shortcut all start from "unknown" each time (the slow paths), and the result is
    still the oracle's run, step for step."""
    import torch
    dev = torch.device("cuda", 0)
    fr = ob.Frirl(env, trig_mode=1)
    assert fr.run() == 1
    prob, agent, envs = frirl_amd.demo_fresh_batch(env, 3, 512, dev)
    run = frirl_amd.train_persistent(prob, agent, envs, budget=5)
    torch.cuda.synchronize()
    assert (run.conv.converged == 1).all() and (run.steps_total == fr.total_steps).all()
    assert run.launches > fr.total_steps // 12
    for e in range(3):
        check_agent_against_oracle(fr, prob, envs, e)
