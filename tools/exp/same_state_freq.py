import sys, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import binding as ob
for env in ('acrobot', 'mountaincar', 'cartpole'):
    fr = ob.Frirl(env, trig_mode=1, maxR=2048)
    ns = fr.nstates
    acts = fr.dim(ns)['values']
    start = np.array([fr.dim(k)['values_def'] for k in range(ns)])
    same = 0; tot = 0
    for ep in range(40):
        states = start.copy()
        a = fr.get_best_action(states)
        q_ant = np.concatenate([states, [acts[a]]])
        for step in range(1000):
            cur, reward, success, q_obs = fr.env_step(q_ant[ns], states)
            a = fr.get_best_action(q_obs)
            cur_q = np.concatenate([q_obs, [acts[a]]])
            tot += 1; same += int((q_obs == q_ant[:ns]).all())
            fr.update_sarsa(q_ant, reward, cur_q)
            states, q_ant = cur, cur_q
            if success == 1: break
    print(env, 'steps', tot, 'same quantized state as previous step: %.3f' % (same / tot))
