"""Lane-group learning kernel (frirl_hip_episode_run_lanes: G lanes per environment, sequential per-lane Shepard sums,
transposed rule bases) against the oracle's whole demo runs and against the per-environment step kernel."""
import numpy as np
import pytest

import frirl_amd
from oracle import binding as ob

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["index-store", "f64-store", "index-store/1-slice", "f64-store/2-slices", "index-store/4-slices"])
def store(request, hip_option):
    """Both rule stores of the lane-group kernel: packed 16-bit universe indices + LDS tables (default when the batch keeps
    the index mirror) and plain f64 columns (option "no_uidx"); rule slices per conclusion: the heuristic's choice (8 for
    these small batches), or forced to 1 (pure sequential sums, the reference's order) / 2 / 4."""
    if request.param.startswith("f64-store"):
        hip_option("no_uidx", 1)
    if "/" in request.param:
        hip_option("lanes_slices", int(request.param.split("/")[1][0]))
    return request.param


@pytest.mark.parametrize("env,episodes,steps,rules", [("mountaincar", 29, 15548, 110), ("cartpole", 58, 33002, 182), ("acrobot", 110, 21207, 367)])
def test_lane_group_training_reaches_the_oracle_rule_base(env, episodes, steps, rules, store):
    """E = 21 agents (ragged last wave) learn from the corner rule base through the lane-group kernel; every agent must
    end exactly where the oracle ends: episodes, total steps, rule count, antecedents and order bit-exact, Q <= 1e-9."""
    import torch
    E = 21
    dev = torch.device("cuda", 0)
    fr = ob.Frirl(env, trig_mode=1)
    assert fr.run() == 1 and fr.five.R == rules and fr.total_steps == steps
    prob, agent, envs = frirl_amd.demo_fresh_batch(env, E, 512, dev)
    total = torch.zeros((E,), dtype=torch.int64, device=dev)

    def on_episode(ep, conv):
        total.add_(envs.ep_steps.long() * (conv.episodes == ep).long())

    conv = frirl_amd.train(prob, agent, envs, on_episode=on_episode, lanes=True)
    torch.cuda.synchronize()
    assert (conv.converged == 1).all()
    assert (conv.episodes == episodes).all(), conv.episodes.tolist()
    assert (total == steps).all(), total.tolist()
    assert (prob.nrules == rules).all()
    f = fr.five
    rant = envs.rant[:, :, :rules].cpu().numpy()
    assert (rant == f.rant[:rules].T[None]).all(), "antecedents / rule order"
    rb = prob.rb.cpu().numpy()
    assert (rb[:, : f.nant, :rules] == f.veval[None, :, :rules]).all(), "VE columns"
    q = rb[:, prob.nant, :rules]
    rel = np.abs(q - f.rconc[None, :rules]) / np.maximum(np.abs(f.rconc[None, :rules]), 1e-9)
    assert rel.max() <= 1e-9, rel.max()
    if prob.uidx is not None:
        ui = prob.uidx[:, :, :rules].cpu().numpy().astype(np.int64) & 0xFFFF
        assert (ui == f.uidx[None, :, :rules]).all(), "index mirror"


@pytest.mark.parametrize("env,explore,maxR", [("mountaincar", False, 256), ("acrobot", False, 256), ("cartpole", False, 256), ("acrobot", True, 256),
                                              ("cartpole", True, 256), ("mountaincar", False, 40), ("acrobot", True, 40),
                                              ("mountaincar", "noskip", 256), ("acrobot", "noskip", 256)])
def test_lane_group_steps_equal_step_kernel(env, explore, maxR, store):
    """Same start, per-environment start states (different trajectories, ragged episode ends): chunks of lane-group
    steps vs the same number of frirl_hip_episode_step launches -- states, actions, rule counts, status, step counts
    identical; Q within 1e-10 (up to two near-tie flips, see below).  maxR = 40: the rule bases fill up, further insertions are refused (status FULL) the same
    way by both."""
    import torch
    dev = torch.device("cuda", 0)
    E = 37
    d = frirl_amd.demo_describe(env)
    g = torch.Generator(device=dev).manual_seed(2)
    cols = []
    for k in range(d["nstates"]):
        vals = torch.from_numpy(d["grids"][k]).to(dev)
        cols.append(vals[torch.randint(0, len(vals), (E,), generator=g, device=dev)])
    start = torch.stack(cols, 1).contiguous()
    kw = dict(epsilon=0.2, no_random=0, seed=1234, env_id_base=77) if explore is True else {}      # epsilon-greedy: same counter-based streams
    pa, agent, ea = frirl_amd.demo_fresh_batch(env, E, maxR, dev, start_states=start, max_steps=300, **kw)
    pb, _, eb = frirl_amd.demo_fresh_batch(env, E, maxR, dev, start_states=start, max_steps=300, **kw)
    if explore == "noskip":     # the other update_rules variant (skip_rules = 0: frirl_update_sarsa.c:70-73) and tighter insertion bounds
        agent.desc.skip_rules = 0
        agent.desc.qdiff_pos_boundary *= 0.25
        agent.desc.qdiff_neg_boundary *= 0.25
    # The two kernels add the Shepard sums in different orders (rule slices vs a tree), so a greedy decision may flip where
    # two actions' Q values agree to ~1e-15; such an environment then follows another trajectory.  Environments are compared
    # exactly, at most 2 of the 37 may take a different branch over the whole run, and a diverged one is re-synchronised
    # from the step kernel's copy so that every chunk starts from identical state.
    def differs():
        R = int(max(pa.nrules.max(), pb.nrules.max()))
        bad = (ea.ep_steps != eb.ep_steps) | (ea.done != eb.done) | (pa.nrules != pb.nrules) | (ea.fus != eb.fus) | (ea.status != eb.status)
        bad |= (ea.states != eb.states).any(1) | (ea.q_ant != eb.q_ant).any(1) | (ea.ep_reward != eb.ep_reward)
        qa, qb = pa.rb[:, pa.nant, :R], pb.rb[:, pb.nant, :R]
        bad |= ((qa - qb).abs() > 1e-10 * qb.abs().clamp(min=1.0)).any(1)
        bad |= (pa.rb[:, : pa.nant, :R] != pb.rb[:, : pb.nant, :R]).any(2).any(1) | (ea.rant[:, :, :R] != eb.rant[:, :, :R]).any(2).any(1)
        return bad

    def resync(bad):
        for dst, src in ((pa.rb, pb.rb), (pa.nrules, pb.nrules), (ea.states, eb.states), (ea.q_ant, eb.q_ant), (ea.fus, eb.fus), (ea.done, eb.done),
                         (ea.ep_steps, eb.ep_steps), (ea.ep_reward, eb.ep_reward), (ea.rant, eb.rant), (ea.status, eb.status)):
            dst[bad] = src[bad]
        if pa.uidx is not None:
            pa.uidx[bad] = pb.uidx[bad]

    flips = 0
    for episode in range(3):
        frirl_amd.episode_begin(pa, agent, ea)
        frirl_amd.episode_begin(pb, agent, eb)
        for chunk in (1, 7, 50, 300):
            frirl_amd.episode_run_lanes(pa, agent, ea, chunk)
            frirl_amd.episode_steps(pb, agent, eb, chunk)
            torch.cuda.synchronize()
            bad = differs()
            n = int(bad.sum())
            if n:
                flips += n
                assert flips <= 2, (episode, chunk, bad.nonzero().flatten().tolist())
                resync(bad)
        assert (ea.done == 1).all()
    if maxR < 256:
        assert int(pa.nrules.max()) == maxR, "the small rule bases were meant to fill up"
