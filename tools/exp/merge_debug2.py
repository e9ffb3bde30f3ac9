import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, frirl_amd
from oracle import binding as ob
env, agents = "mountaincar", 4
dev = torch.device("cuda", 0)
ag = [ob.Frirl(env, trig_mode=1) for _ in range(agents)]
ns = ag[0].nstates
starts = np.array([ag[0].five.gen_def_states(i, agents, ns) if i else [ag[0].dim(k)["values_def"] for k in range(ns)] for i in range(agents)])
print("starts", starts.tolist())
for i in range(1, agents): ag[i].set_start_state(starts[i])
prob, agent, envs = frirl_amd.demo_fresh_batch(env, agents, 1024, dev, start_states=torch.from_numpy(starts).to(dev).contiguous())
def cmp(tag):
    torch.cuda.synchronize()
    for i in range(agents):
        f = ag[i].five; R = f.R
        okR = int(prob.nrules[i]) == R
        ra = envs.rant[i, :, :R].T.cpu().numpy()
        okA = okR and (ra == np.array(f.rant[:R])).all()
        q = prob.rb[i, prob.nant, :R].cpu().numpy() if okR else None
        rel = (np.abs(q - f.rconc[:R]) / np.maximum(np.abs(f.rconc[:R]), 1e-9)).max() if okR else -1
        print(tag, "agent", i, "R", int(prob.nrules[i]), R, "ants", okA, "max rel", rel)
for ep in range(9):
    for a in ag: a.episode()
    frirl_amd.episode_begin(prob, agent, envs)
    frirl_amd.episode_run_lanes(prob, agent, envs, agent.desc.max_steps)
cmp("after 9 episodes")
weights = torch.zeros((agents, prob.maxR), dtype=torch.float64, device=dev)
for a in ag: a.five.weights[:] = 0.0
m = ag[0].five
mr, mc = np.array(m.rant[:m.R]), np.array(m.rconc[:m.R])
for i in range(1, agents): ag[i].five.merge_rb(ag[i].agent(), mr, mc)
active = torch.tensor([0] + [1] * (agents - 1), dtype=torch.uint8, device=dev)
full = torch.zeros((agents,), dtype=torch.int32, device=dev)
snd = frirl_amd.SenderDesc(envs.rant[0].data_ptr(), 1, prob.maxR, prob.rb[0, prob.nant].data_ptr(), 0, 0, prob.nrules[0:1].data_ptr())
frirl_amd.check(frirl_amd.lib().frirl_hip_merge_rb(C.byref(prob.tables), C.byref(prob.bases), C.byref(agent.desc), envs.rant.data_ptr(), C.byref(snd), weights.data_ptr(), active.data_ptr(), full.data_ptr(), None), "m")
cmp("after phase 1")
one = frirl_amd.RuleBases(1, prob.maxR, prob.rb.data_ptr(), prob.nrules.data_ptr(), prob.uidx.data_ptr())
for i in range(1, agents):
    f = ag[i].five
    m.merge_rb(ag[0].agent(), np.array(f.rant[:f.R]), np.array(f.rconc[:f.R]))
    snd = frirl_amd.SenderDesc(envs.rant[i].data_ptr(), 1, prob.maxR, prob.rb[i, prob.nant].data_ptr(), 0, 0, prob.nrules[i:i+1].data_ptr())
    frirl_amd.check(frirl_amd.lib().frirl_hip_merge_rb(C.byref(prob.tables), C.byref(one), C.byref(agent.desc), envs.rant.data_ptr(), C.byref(snd), weights.data_ptr(), None, full.data_ptr(), None), "m2")
    cmp(f"after phase 2 sender {i}")
for ep in range(2):
    for a in ag: a.episode()
    frirl_amd.episode_begin(prob, agent, envs)
    frirl_amd.episode_run_lanes(prob, agent, envs, agent.desc.max_steps)
    cmp(f"after episode {10+ep}")
