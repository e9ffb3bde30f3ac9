"""How many rules would a lane flag as spread candidates (w > c * running partial sum of its own slice, descending walk)?
CPU experiment on the oracle's acrobot / mountaincar / cartpole training run; prints distribution per lane-group size H."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import binding as ob

env = sys.argv[1] if len(sys.argv) > 1 else 'acrobot'
neps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
fr = ob.Frirl(env, trig_mode=1, maxR=2048)
f = fr.five
ns = fr.nstates
p = fr.nant
sig = fr.hparams['weight_thr']
acts = fr.dim(ns)['values']
start = np.array([fr.dim(k)['values_def'] for k in range(ns)])
Hs = [2, 4, 8, 16]
stat = {H: dict(maxc=[], sumc=[]) for H in Hs}
true_cnt = []
nsteps = 0; nspread = 0
for ep in range(neps):
    states = start.copy()
    a = fr.get_best_action(states)
    q_ant = np.concatenate([states, [acts[a]]])
    for step in range(1000):
        cur, reward, success, q_obs = fr.env_step(q_ant[ns], states)
        a = fr.get_best_action(q_obs)
        cur_q = np.concatenate([q_obs, [acts[a]]])
        R = f.R
        f.rule_distance(q_ant)
        d = np.array(f.ruledists[:R])
        nsteps += 1
        if R > 0 and (d > 0).all():
            w = d ** (-float(p))
            ws = w.sum()
            nt = int((w / ws > sig).sum())
            true_cnt.append(nt)
            for H in Hs:
                mc = 0; sc = 0
                for h in range(H):
                    wl = w[h::H][::-1]            # descending walk over the lane's rules
                    if len(wl) == 0: continue
                    cs = np.cumsum(wl)
                    c = int((wl > sig * (1 - 1e-9) * cs).sum())
                    mc = max(mc, c); sc += c
                stat[H]['maxc'].append(mc); stat[H]['sumc'].append(sc)
        fr.update_sarsa(q_ant, reward, cur_q)
        states, q_ant = cur, cur_q
        if success == 1: break
print(env, 'episodes', neps, 'steps', nsteps, 'rules', f.R)
tc = np.array(true_cnt)
print('true significant: mean %.2f max %d  pct>8: %.3f' % (tc.mean(), tc.max(), (tc > 8).mean()))
for H in Hs:
    m = np.array(stat[H]['maxc']); s = np.array(stat[H]['sumc'])
    print('H=%2d per-lane max flagged: mean %.2f p50 %d p90 %d p99 %d max %d ; group total mean %.1f' % (H, m.mean(), np.percentile(m, 50), np.percentile(m, 90), np.percentile(m, 99), m.max(), s.mean()))
