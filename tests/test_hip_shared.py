"""One shared, read-only rule base, many environments (SURVEY 8f #3 evaluation mode; 8f #1 try-remove replays):
frirl_hip_rollout_shared against the oracle's frirl_test_run episode (orc_episode_eval, portable trig)."""
import numpy as np
import pytest

import frirl_amd
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def trained(env):
    fr = ob.Frirl(env, trig_mode=1)
    assert fr.run() == 1
    return fr


def shared_problem(fr, dev):
    import torch
    f = fr.five
    R, nant = f.R, f.nant
    maxR = R + 8 + (R & 1)
    rb = np.zeros((1, nant + 1, maxR))
    rb[0, :nant, :R] = f.veval[:, :R]
    rb[0, nant, :R] = f.rconc[:R]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return frirl_amd.Problem(t(np.array(f.u)), t(np.array(f.ve)), t(rb), t(np.array([R], dtype=np.int32))), maxR


def start_states(fr, Q, seed):
    rng = np.random.default_rng(seed)
    ns = fr.nstates
    s = np.zeros((Q, ns))
    for k in range(ns):
        d = fr.dim(k)
        vals = d["values"]
        span = vals[-1] - vals[0]
        s[:, k] = d["values_def"] + rng.uniform(-0.15, 0.15, Q) * span
        s[:, k] = np.clip(s[:, k], vals[0], vals[-1])
        s[0, k] = d["values_def"]                                  # row 0: the reference's own start state
    return s


@pytest.mark.parametrize("env", ["mountaincar", "cartpole", "acrobot"])
def test_rollouts_on_shared_rule_base_follow_oracle(env):
    import torch
    dev = torch.device("cuda", 0)
    Q = 300                                                         # two workgroups, the second one ragged
    fr = trained(env)
    prob, _ = shared_problem(fr, dev)
    agent = frirl_amd.demo_agent(frirl_amd.demo_describe(env), dev)
    s = start_states(fr, Q, 5)
    steps, reward, success, final = prob.rollout_shared(agent, Q, start_states=torch.from_numpy(s).to(dev))
    torch.cuda.synchronize()
    steps, reward, success = steps.cpu().numpy(), reward.cpu().numpy(), success.cpu().numpy()
    ok = 0
    for i in range(0, Q, 3 if env != "mountaincar" else 1):
        fr.set_start_state(s[i])
        fr.episode_eval()
        assert steps[i] == fr.ep_steps, (i, steps[i], fr.ep_steps)
        assert abs(reward[i] - fr.ep_reward) <= 1e-9 * max(1.0, abs(fr.ep_reward)), (i, reward[i], fr.ep_reward)
        ok += 1
    # row 0 = the converged demo's own last episode
    assert success[0] == 1 or env == "cartpole"
    # lanes-per-environment variants (1 = throughput layout, 4 / 8 = actions split over lanes): bit-identical results
    old = frirl_amd.set_option("rollout_group", 1)
    try:
        st1, rw1, _, fin1 = prob.rollout_shared(agent, Q, start_states=torch.from_numpy(s).to(dev))
        torch.cuda.synchronize()
    finally:
        frirl_amd.set_option("rollout_group", old)
    assert (st1.cpu().numpy() == steps).all() and (rw1.cpu().numpy() == reward).all() and (fin1 == final).all()
    # default start (NULL start_states) = row 0 for every lane
    st2, rw2, _, _ = prob.rollout_shared(agent, 70)
    torch.cuda.synchronize()
    assert (st2.cpu().numpy() == steps[0]).all() and (rw2.cpu().numpy() == reward[0]).all()


@pytest.mark.parametrize("env", ["mountaincar", "acrobot"])
def test_try_remove_masks_equal_compacted_rule_base(env):
    """Environment q ignores the candidate rules named by its mask: identical to the oracle's roll-out on the rule base
    after five_remove_rule of those rules (frirl_sequential_run.c:170-350 replays)."""
    import torch
    dev = torch.device("cuda", 0)
    fr = trained(env)
    R = fr.five.R
    prob, maxR = shared_problem(fr, dev)
    agent = frirl_amd.demo_agent(frirl_amd.demo_describe(env), dev)
    order = np.argsort(np.abs(fr.five.rconc[:R]), kind="stable")
    cand = [int(r) for r in order[-16:][::-1]] + [int(r) for r in order[:16]]     # slots 0..15: largest |Q|, 16..31: smallest
    slot = np.full(maxR, 255, dtype=np.uint8)
    for sl, r in enumerate(cand):
        slot[r] = sl
    masks = np.array([0, 1, 0x80000000, 0xffff, 0xffff0000, 0xffffffff, 0x00ff00ff, 0x0f0f0f0f, 0x3f, 0xfff, 0xffffff, 0xa5a5a5a5],
                     dtype=np.uint32).view(np.int32)
    steps, reward, success, _ = prob.rollout_shared(agent, len(masks), exclude_mask=torch.from_numpy(masks).to(dev),
                                                    rule_slot=torch.from_numpy(slot).to(dev))
    torch.cuda.synchronize()
    steps, reward = steps.cpu().numpy(), reward.cpu().numpy()
    changed = 0
    for i, m in enumerate(masks):
        fr2 = trained(env)
        for r in sorted([cand[sl] for sl in range(32) if (int(np.uint32(m)) >> sl) & 1], reverse=True):
            fr2.five.remove_rule(r)
        fr2.episode_eval()
        assert steps[i] == fr2.ep_steps, (i, hex(int(np.uint32(m))), steps[i], fr2.ep_steps)
        assert abs(reward[i] - fr2.ep_reward) <= 1e-9 * max(1.0, abs(fr2.ep_reward))
        changed += (steps[i] != steps[0]) or (reward[i] != reward[0])
    assert changed >= 1, "removals did not change the episode: the masks are not exercised"


@pytest.mark.parametrize("env,strategy", [("mountaincar", 1), ("mountaincar", 2), ("cartpole", 1), ("acrobot", 1), ("acrobot", 2)])
def test_speculative_reduction_equals_sequential_reduction(env, strategy):
    """frirl_hip_reduce_shared (2^depth - 1 speculative replays per launch) must keep exactly the rules the reference's
    one-candidate-per-episode loop keeps (oracle orc_reduce_run = frirl_sequential_run.c:170-350), in the same order,
    with untouched consequents; also for a depth that does not divide the rule count and for depth 1 (= sequential)."""
    import torch
    dev = torch.device("cuda", 0)
    fr = trained(env)
    f = fr.five
    R0, nant = f.R, f.nant
    rant0, rconc0, veval0 = np.array(f.rant[:R0]), np.array(f.rconc[:R0]), np.array(f.veval[:, :R0])
    agent = frirl_amd.demo_agent(frirl_amd.demo_describe(env), dev)
    results = {}
    for depth in ([0, 7, 1] if env == "mountaincar" else [0]):
        prob, maxR = shared_problem(fr, dev)
        rant_d = torch.zeros((nant, maxR), dtype=torch.float64, device=dev)
        rant_d[:, :R0] = torch.from_numpy(np.ascontiguousarray(rant0.T)).to(dev)
        kept, res = prob.reduce_shared(agent, strategy, 0.0, depth, rant=rant_d)
        torch.cuda.synchronize()
        results[depth] = (kept.copy(), res.rules_after, res.rounds)
        assert res.rules_before == R0 and res.rules_after == len(kept) == int(prob.nrules[0].item())
        R1 = res.rules_after
        rb = prob.rb[0].cpu().numpy()
        assert (rb[:nant, :R1] == veval0[:, kept]).all() and (rb[nant, :R1] == rconc0[kept]).all()
        assert (rb[:, R1:R0] == 0).all()
        assert (rant_d[:, :R1].cpu().numpy() == rant0[kept].T).all()
    fr.reduce(strategy, 0.0)                                         # the sequential loop (mutates the oracle's rule base)
    R1 = f.R
    kept0, n0, rounds0 = results[0]
    assert n0 == R1, (n0, R1)
    assert (np.array(f.rant[:R1]) == rant0[kept0]).all(), "surviving rules / order differ from the sequential reduction"
    assert (np.array(f.rconc[:R1]) == rconc0[kept0]).all()
    assert rounds0 == -(-R0 // 10)
    for depth, (k, n, rounds) in results.items():
        assert n == R1 and (k == kept0).all(), depth
