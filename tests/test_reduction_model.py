"""CPU model of the speculative batched try-remove checked against the oracle's sequential reduction
(orc_reduce_run = reference frirl_sequential_run.c:170-350): the two facts the GPU implementation rests on --
(1) the candidate order is known up front (stable sort by |Q|), (2) evaluating the whole accept/reject tree of the next
`depth` candidates and walking it along the actual outcomes makes the same decisions as one candidate per episode."""
import numpy as np
import pytest

from oracle import binding as ob


def trained(env):
    fr = ob.Frirl(env, trig_mode=1)
    assert fr.run() == 1
    return fr


def replay(env, removed):
    """frirl_test_run's episode on the trained rule base with the rules in `removed` (original indices) taken out."""
    fr = trained(env)
    for r in sorted(removed, reverse=True):
        fr.five.remove_rule(r)
    fr.episode_eval()
    return fr.ep_steps, fr.ep_reward


@pytest.mark.parametrize("strategy,depth", [(1, 3), (2, 2)])
def test_speculative_tree_walk_equals_sequential_reduction(strategy, depth):
    env = "mountaincar"
    fr = trained(env)
    f = fr.five
    R0 = f.R
    rconc = np.array(f.rconc[:R0])
    rant0 = np.array(f.rant[:R0])
    good_above = fr.hparams["reward_good_above"]
    order = np.argsort(np.abs(rconc) if strategy == 1 else -np.abs(rconc), kind="stable")
    steps_inc, prev_reward = replay(env, [])
    accepted = []
    cache = {}

    def outcome(removed):
        key = tuple(sorted(removed))
        if key not in cache:
            cache[key] = replay(env, removed)
        return cache[key]

    j = 0
    while j < R0:
        d = min(depth, R0 - j)
        cands = [int(c) for c in order[j:j + d]]
        # every node of the accept/reject tree: (k, bits) -> replay without accepted-so-far, the accepted ones on the path, candidate k
        tree = {}
        for k in range(d):
            for bits in range(1 << k):
                extra = [cands[i] for i in range(k) if (bits >> i) & 1]
                tree[(k, bits)] = outcome(accepted + extra + [cands[k]])
        bits = 0
        for k in range(d):
            st, rw = tree[(k, bits)]
            if rw > good_above and st == steps_inc and abs(prev_reward - rw) <= 0.0:
                bits |= 1 << k
                prev_reward = rw
        accepted += [cands[i] for i in range(d) if (bits >> i) & 1]
        j += d
    kept = [r for r in range(R0) if r not in set(accepted)]
    fr.reduce(strategy, 0.0)                                   # the sequential loop
    assert f.R == len(kept)
    assert (np.array(f.rant[: f.R]) == rant0[kept]).all() and (np.array(f.rconc[: f.R]) == rconc[kept]).all()
