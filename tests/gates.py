"""Parity gates: a handful of sampled environments of a LARGE batch checked against the oracle after a launch -- used by the
full-shape tests (tests/test_full_shapes.py) and by bench.py after each timed leg, outside the timed region ("parity
gates run with every measurement").  The oracle is the checker here, never the thing measured.

  gate_rule_distance  distances of the sampled rule bases bit-exact, exact-hit index exact (reference five_rule_distance.c:63-295)
  gate_env_step       ONE more fused step of the whole batch (untimed); for the sampled environments the oracle replays it from a
                      snapshot with its own primitives (env step, frirl_get_best_action, frirl_update_sarsa): continuous state and
                      chosen action exact, rule count and appended antecedents exact, consequents <= 1e-9
                      (reference frirl_episode.c:86-185, frirl_update_sarsa.c:348-385)
"""
import numpy as np

from oracle import binding as ob


def spread_sample(E, n=8, boundaries=()):
    """First, last, the middle pair and the environments either side of the given boundaries (chunk / batch edges)."""
    s = {0, 1 if E > 1 else 0, E - 1, max(E - 2, 0), E // 2, max(E // 2 - 1, 0)}
    for b in boundaries:
        for e in (b - 1, b):
            if 0 <= e < E:
                s.add(e)
    rng = np.random.default_rng(E)
    while len(s) < min(n, E):
        s.add(int(rng.integers(0, E)))
    return sorted(s)


def _tables(prob):
    return np.ascontiguousarray(prob.u.cpu().numpy()), np.ascontiguousarray(prob.ve.cpu().numpy())


def gate_rule_distance(prob, x, dists, hit, sample):
    """prob.rule_distance(x) -> (dists, hit) already computed on the device; checks the sampled environments.  Returns a record."""
    u, ve = _tables(prob)
    nant, U, maxR = prob.nant, prob.U, prob.maxR
    S = len(sample)
    idx = np.array(sample)
    nr = prob.nrules[idx].cpu().numpy().astype(np.int32)
    rb = np.ascontiguousarray(prob.rb[idx].cpu().numpy())
    xs = np.ascontiguousarray(x[idx].cpu().numpy())
    if prob.uidx is not None:           # the 16-bit mirror the compressed scan streams must name the same doubles
        ui = prob.uidx[idx].cpu().numpy().astype(np.int64) & 0xFFFF
        for i in range(S):
            n = int(nr[i])
            for k in range(nant):
                assert (ve[k][ui[i, k, :n]] == rb[i, k, :n]).all(), f"environment {sample[i]}: index mirror and f64 column {k} disagree"
    d_ref = np.zeros((S, maxR))
    h_ref = np.zeros(S, dtype=np.int32)
    ob.lib().orc_batch_rule_distance(S, nant, U, maxR, ob.dp(u.ravel()), ob.dp(ve.ravel()), ob.dp(rb.reshape(-1)), ob.ip(nr), ob.dp(xs.reshape(-1)),
                                     ob.dp(d_ref.reshape(-1)), ob.ip(h_ref), 0)
    d = dists[idx].cpu().numpy()
    h = hit[idx].cpu().numpy().astype(np.int64)
    h = np.where(h == 0xFFFFFFFF, -1, h)
    hits = 0
    for i in range(S):
        n = int(nr[i])
        assert int(h[i]) == int(h_ref[i]), f"environment {sample[i]}: exact-hit index {int(h[i])} != oracle {int(h_ref[i])}"
        assert (d[i, :n].view(np.uint64) == d_ref[i, :n].view(np.uint64)).all(), f"environment {sample[i]}: distances are not bit-identical"
        hits += int(h_ref[i] >= 0)
    return {"checked": S, "ok": True, "environments": list(map(int, sample)), "exact_hits_in_sample": hits,
            "what": "distances bit-exact + exact-hit index vs the oracle"}


def _oracle_with_rule_base(env_name, maxR, rant, rconc):
    fr = ob.Frirl(env_name, trig_mode=1, maxR=maxR)
    f = fr.five
    while f.R:
        assert ob.lib().orc_remove_rule(f.h, 0) == 0
    for r in range(rant.shape[0]):
        assert f.add_rule(rant[r], rconc[r]) == 0
    return fr


def gate_env_step(prob, agent, envs, env_name, sample, step):
    """`step()` runs ONE more fused step of the whole batch.  The sampled environments are snapshotted before it and replayed by
    the oracle.  Environments whose episode has ended are skipped (nothing happens to them).  Returns a record."""
    import torch
    u, ve = _tables(prob)
    nant, maxR = prob.nant, prob.maxR
    ns = nant - 1
    idx = torch.tensor(sample, device=prob.rb.device)
    snap = dict(nr=prob.nrules[idx].cpu().numpy(), rb=prob.rb[idx].cpu().numpy(), states=envs.states[idx].cpu().numpy(), q_ant=envs.q_ant[idx].cpu().numpy(),
                fus=envs.fus[idx].cpu().numpy(), done=envs.done[idx].cpu().numpy())
    if envs.rant is not None:
        snap["rant"] = envs.rant[idx].cpu().numpy()
    else:                                   # raw antecedents from the index mirror: every stored antecedent is a universe point
        ui = prob.uidx[idx].cpu().numpy().astype(np.int64) & 0xFFFF
        snap["rant"] = np.stack([np.stack([u[k][ui[i, k]] for k in range(nant)]) for i in range(len(sample))])
    step()
    torch.cuda.synchronize()
    after = dict(nr=prob.nrules[idx].cpu().numpy(), rb=prob.rb[idx].cpu().numpy(), states=envs.states[idx].cpu().numpy(), q_ant=envs.q_ant[idx].cpu().numpy(),
                 fus=envs.fus[idx].cpu().numpy())
    if prob.uidx is not None:
        after["uidx"] = prob.uidx[idx].cpu().numpy().astype(np.int64) & 0xFFFF
    checked, inserted = 0, 0
    for i, e in enumerate(sample):
        if snap["done"][i]:
            continue
        n0 = int(snap["nr"][i])
        fr = _oracle_with_rule_base(env_name, maxR, np.ascontiguousarray(snap["rant"][i][:, :n0].T), snap["rb"][i][nant, :n0])
        f = fr.five
        assert (f.veval[:, :n0] == snap["rb"][i][:nant, :n0]).all(), f"environment {e}: VE columns differ from the oracle's for the same antecedents"
        fr.fus = float(snap["fus"][i])
        q_ant = snap["q_ant"][i]
        cur, reward, success, q_obs = fr.env_step(q_ant[ns], snap["states"][i])                      # frirl_episode.c:97-112
        a = fr.get_best_action(q_obs)                                                                  # :148
        action = fr.dim(ns)["values"][a]
        cur_q = np.concatenate([q_obs, [action]])
        fr.update_sarsa(q_ant, reward, cur_q)                                                          # :155 -> frirl_update_sarsa.c:348-385
        assert (after["states"][i].view(np.uint64) == np.asarray(cur).view(np.uint64)).all(), f"environment {e}: continuous state after the step"
        assert (after["q_ant"][i].view(np.uint64) == cur_q.view(np.uint64)).all(), f"environment {e}: observation / chosen action {after['q_ant'][i]} != oracle {cur_q}"
        n1 = f.R
        assert int(after["nr"][i]) == n1, f"environment {e}: rule count {int(after['nr'][i])} != oracle {n1}"
        assert int(after["fus"][i]) == int(fr.fus), f"environment {e}: sticky insert flag"
        assert (after["rb"][i][:nant, :n1] == f.veval[:, :n1]).all(), f"environment {e}: antecedents (appended rule) differ"
        if "uidx" in after:
            assert (after["uidx"][i][:, :n1] == f.uidx[:, :n1]).all(), f"environment {e}: index mirror of the appended rule"
        q, qo = after["rb"][i][nant, :n1], np.array(f.rconc[:n1])
        rel = np.abs(q - qo) / np.maximum(np.abs(qo), 1e-9)
        assert rel.max() <= 1e-9, f"environment {e}: consequents differ by {rel.max():.3g} relative"
        checked += 1
        inserted += int(n1 > n0)
    return {"checked": checked, "ok": True, "environments": list(map(int, sample)), "appended_in_sample": inserted,
            "what": "one more fused step: state, chosen action, rule count, appended antecedents exact; consequents <= 1e-9 vs the oracle's replay"}
