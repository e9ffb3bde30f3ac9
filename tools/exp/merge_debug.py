import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, frirl_amd
from oracle import binding as ob
sys.path.insert(0, os.path.join(frirl_amd.ROOT, "tests"))
from test_hip_merge import records, batch_from_rules, fha
env = sys.argv[1] if len(sys.argv) > 1 else "mountaincar"
dev = torch.device("cuda", 0)
recs = records(env, os.path.join(frirl_amd.ROOT, "tests", "golden"))
a, m = recs["agent_before"], recs["master_before"]
nant = frirl_amd.demo_describe(env)["nant"]
srant_h, srconc_h = fha(m["rant"]).reshape(m["R"], nant), fha(m["rconc"])
fr = ob.Frirl(env, trig_mode=0)
oag = fr.agent()
def oracle_after(k):
    f = ob.Five(np.array(fr.five.u).ravel(), np.array(fr.five.ve).ravel(), nant, fr.five.U, 512, fha(a["rant"]).reshape(a["R"], nant), fha(a["rconc"]))
    f.merge_rb(oag, srant_h[:k], srconc_h[:k])
    return f
prev_bad = False
for k in range(1, m["R"] + 1):
    prob, agent, store = batch_from_rules(env, a, 1, 512, dev)
    w = torch.zeros((1, 512), dtype=torch.float64, device=dev)
    prob.merge_rb(agent, torch.from_numpy(srant_h[:k].copy()).to(dev), torch.from_numpy(srconc_h[:k].copy()).to(dev), w, rant_store=store)
    torch.cuda.synchronize()
    f = oracle_after(k)
    R = f.R
    q = prob.rb[0, nant, :R].cpu().numpy()
    ok_R = int(prob.nrules[0]) == R
    rel = np.abs(q - f.rconc[:R]) / np.maximum(np.abs(f.rconc[:R]), 1e-9)
    bad = (not ok_R) or rel.max() > 1e-9
    if bad and not prev_bad:
        print("first divergence after sender rule", k - 1, "R gpu/orc", int(prob.nrules[0]), R, "max rel", rel.max(), "at", int(rel.argmax()))
        # classify the k-1 th rule on the oracle side (state before it)
        fb = oracle_after(k - 1)
        x = srant_h[k - 1]
        h, c = fb.vag_concl(x)
        print("   sender rule", x, "Qs", srconc_h[k - 1], "receiver hit", h, "Qr", c, "qdiff", srconc_h[k - 1] - c, "bounds", oag.qdiff_pos, oag.qdiff_neg)
        ww = np.array(fb.weights[:fb.R]); print("   stale weights >thr:", int((ww > oag.weight_thr).sum()), "of", fb.R)
        idx = np.argsort(-rel)[:5]
        for i in idx: print("     rule", i, "gpu", q[i], "orc", f.rconc[i], "rel", rel[i])
        break
    prev_bad = bad
else:
    print("no divergence")
