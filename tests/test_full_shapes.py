"""The HEADLINE launch shapes at their real size through five_hip_rule_distance -- the launches bench.py times:
  cfg4  acrobot tables, 65 536 rules x 8 192 environments (270 336 workgroups, chunk index fastest), compressed and f64 layouts
  cfg3  cartpole tables (40 KB), 32 768 rules x 32 768 environments (persistent workgroups, in-order hand-out of 557 056 items)
Size-independent properties on ALL environments (exact hit found, first zero, nothing non-finite, identical bits between the two
layouts) plus the oracle on environments sampled at the first / last / middle positions and across work-item boundaries
(tests/gates.py).  Reference: five_rule_distance.c:63-295."""
import numpy as np
import pytest

import frirl_amd
from tests import gates

pytestmark = pytest.mark.gpu


def _properties_and_gate(env, E, R, layouts, boundaries):
    import torch
    dev = torch.device("cuda", 0)
    prob, agent, envs = frirl_amd.demo_batch(env, E, R, R + 256, dev, seed=5, keep_rant=False)
    nant = prob.nant
    g = torch.Generator(device=dev).manual_seed(9)
    ar = torch.arange(E, device=dev)
    # every environment queries one of its own rules (universe points of the rule's indices): an exact hit at or before that rule;
    # every 7th environment instead gets a continuous observation (no hit expected in general)
    pick = torch.randint(0, R, (E,), generator=g, device=dev)
    xi = (prob.uidx[ar, :, pick].long() & 0xFFFF)
    x = torch.stack([prob.u[k][xi[:, k]] for k in range(nant)], dim=1)
    lo, hi = prob.u[:, 0], prob.u[:, prob.U - 2]
    cont = lo + (hi - lo) * torch.rand((E, nant), generator=g, device=dev, dtype=torch.float64)
    free = (ar % 7) == 3
    x[free] = cont[free]
    x = x.contiguous()
    sample = gates.spread_sample(E, 10, boundaries)
    results = {}
    for layout in layouts:
        p = prob if layout == "compressed" else frirl_amd.Problem(prob.u, prob.ve, prob.rb, prob.nrules)
        d, hit = p.rule_distance(x)
        torch.cuda.synchronize()
        h = hit.long()
        h = torch.where(h == 0xFFFFFFFF, torch.full_like(h, -1), h)
        hp = h[~free]
        assert (hp >= 0).all() and (hp <= pick[~free]).all(), layout
        # distance exactly 0 at the hit and > 0 before it, checked in slabs of environments (the full matrix is 4 GB)
        for e0 in range(0, E, 1024):
            sl = slice(e0, min(e0 + 1024, E))
            dd = d[sl, :R]
            assert torch.isfinite(dd).all() and (dd >= 0).all(), layout
            zero = dd == 0
            anyz = zero.any(dim=1)
            first = torch.where(anyz, zero.float().argmax(dim=1), torch.full((dd.shape[0],), -1, device=dev))
            assert (first == h[sl]).all(), (layout, e0, "hit is the FIRST zero-distance rule, -1 when there is none")
        rec = gates.gate_rule_distance(p, x, d, hit, sample)
        assert rec["ok"] and rec["checked"] >= 8
        results[layout] = (d, hit)
    if len(layouts) == 2:
        (d0, h0), (d1, h1) = results[layouts[0]], results[layouts[1]]
        assert (h0 == h1).all()
        for e0 in range(0, E, 1024):
            assert (d0[e0:e0 + 1024, :R] == d1[e0:e0 + 1024, :R]).all(), "the compressed and the f64 layout must give identical bits"


def test_cfg4_headline_launch_full_shape():
    # 33 chunks of 2048 rules per environment: sample around environments whose items straddle multiples of the dispatch width too
    _properties_and_gate("acrobot", 8192, 65536, ["compressed", "f64"], boundaries=(256, 4096, 7936))


def test_cfg3_headline_launch_full_shape():
    # persistent form: 17 items per environment handed out in order, 4 per atomic: environments at item-batch edges
    _properties_and_gate("cartpole", 32768, 32768, ["compressed"], boundaries=(4, 1024, 16384, 32764))
