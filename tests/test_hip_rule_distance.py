"""GPU parity of five_hip_rule_distance (through the C ABI) against the oracle:
distances BIT-EXACT, hit indices BIT-EXACT (reference five_rule_distance.c:63-295)."""
import numpy as np
import pytest

from tests.problems import Batch

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def run_case(b, x, materialise=True, compressed=False):
    import torch
    prob = b.to_device(compressed=compressed)
    xd = torch.from_numpy(x).cuda()
    d, hit = prob.rule_distance(xd, materialise=materialise)
    torch.cuda.synchronize()
    return (d.cpu().numpy() if materialise else None), hit.cpu().numpy()


CASES = [  # nant, U, R, E, A
    (3, 41, 8, 5, 3),          # the 2^nant corner-sized base
    (3, 41, 33, 7, 3),         # odd rule count
    (3, 41, 110, 64, 3),       # mountaincar-sized
    (5, 1001, 182, 33, 21),    # cartpole-sized
    (5, 41, 367, 40, 3),       # acrobot-sized
    (5, 41, 4096, 9, 3),
    (8, 101, 4099, 4, 0),      # reference's maximum nant, odd R > one chunk
    (16, 201, 3000, 3, 0),     # cfg5 shape (nant beyond the reference's cap of 8)
    (16, 1001, 70001, 3, 0),   # cfg5 tables (125 KiB: the large-LDS index kernel, 1024 threads), several chunks, odd R
    (8, 1001, 5000, 3, 0),     # 62.6 KiB of tables: large-LDS kernel for nant <= 8 too
    (1, 41, 40, 3, 0),
    (3, 41, 20000, 2, 0),      # several chunks per environment + atomicMin across workgroups
]


@pytest.mark.parametrize("nant,U,R,E,A", CASES)
def test_rule_distance_bit_exact(nant, U, R, E, A):
    b = Batch(nant, U, R, E, A=A, seed=100 + nant + R, ragged=True)
    x = b.queries(seed=R, hit_fraction=0.4)
    d_ref, hit_ref = b.oracle_rule_distance(x)
    d, hit = run_case(b, x)
    assert (hit.astype(np.int64) == hit_ref.astype(np.int64)).all(), "exact-hit index (lowest r < nrules with d == 0, else -1)"
    if E >= 8:
        assert (hit_ref >= 0).any() and (hit_ref < 0).any()
    for e in range(E):
        n = int(b.nrules[e])
        assert (bits(d[e, :n]) == bits(d_ref[e, :n])).all(), f"env {e}: distances must be bit-identical"
    _, hit2 = run_case(b, x, materialise=False)
    assert (hit2 == hit).all(), "index-only form"
    if nant * U * 8 <= 150 * 1024:
        # compressed-antecedent form (16-bit universe indices + LDS tables): the same bits
        dc, hitc = run_case(b, x, compressed=True)
        assert (hitc == hit).all()
        for e in range(E):
            n = int(b.nrules[e])
            assert (bits(dc[e, :n]) == bits(d_ref[e, :n])).all(), f"env {e}: compressed form must be bit-identical too"
        _, hitc2 = run_case(b, x, materialise=False, compressed=True)
        assert (hitc2 == hit).all()


def test_duplicate_zero_distance_rules_lowest_index_wins():
    b = Batch(3, 41, 600, 3, A=3, seed=9, ragged=False)
    # duplicate rule 500 of env 1 into slots 17 and 555
    for r in (17, 555):
        b.rb[1, :3, r] = b.rb[1, :3, 500]
        b.uidx[1, :, r] = b.uidx[1, :, 500]
    x = b.queries(seed=3, hit_fraction=0.0, hit_last=False)
    x[1] = b.u[np.arange(3), b.uidx[1, :, 500]]
    _, hit_ref = b.oracle_rule_distance(x)
    _, hit = run_case(b, x)
    assert hit_ref[1] == 17 and (hit == hit_ref).all()


def test_hit_only_beyond_nrules_is_ignored():
    b = Batch(3, 41, 64, 2, A=3, seed=4, ragged=False)
    b.nrules[:] = 41                                  # rules 41..63 are stale padding with real values
    x = b.queries(seed=8, hit_fraction=0.0, hit_last=False)
    x[0] = b.u[np.arange(3), b.uidx[0, :, 50]]        # would hit rule 50 >= nrules
    _, hit_ref = b.oracle_rule_distance(x)
    _, hit = run_case(b, x)
    assert hit_ref[0] == -1 and (hit == hit_ref).all()


def test_observation_outside_universe_clamps_like_reference():
    b = Batch(3, 41, 200, 4, A=0, seed=12, ragged=False)
    x = b.queries(seed=1, hit_fraction=0.0, hit_last=False)
    x[0, 0] = b.u[0, 0] - 5.0
    x[1, 1] = b.u[1, -1] + 5.0
    x[2, 2] = b.u[2, -1]
    d_ref, hit_ref = b.oracle_rule_distance(x)
    d, hit = run_case(b, x)
    assert (bits(d[:, :200]) == bits(d_ref[:, :200])).all() and (hit == hit_ref).all()


def test_full_size_cfg2_properties():
    """BASELINE cfg2: 8192 rules x 8192 envs (nant 3).  Checked against the oracle on a sample of
    environments and through size-independent properties on all of them."""
    import torch
    nant, U, R, E = 3, 41, 8192, 8192
    small = Batch(nant, U, R, 16, A=0, seed=77, ragged=False)
    g = torch.Generator(device="cuda").manual_seed(1)
    ve = torch.from_numpy(small.ve).cuda()
    u = torch.from_numpy(small.u).cuda()
    uidx = torch.randint(0, U, (E, nant, R), generator=g, device="cuda")
    rb = torch.zeros((E, nant + 1, R), dtype=torch.float64, device="cuda")
    for k in range(nant):
        rb[:, k, :] = ve[k][uidx[:, k, :]]
    rb[:16] = torch.from_numpy(small.rb).cuda()
    uidx[:16] = torch.from_numpy(small.uidx[:, :, :R].astype(np.int64)).cuda()
    nrules = torch.full((E,), R, dtype=torch.int32, device="cuda")
    import frirl_amd
    prob = frirl_amd.Problem(u, ve, rb, nrules)
    # every environment queries the antecedents of one of its own rules: an exact hit must be found at
    # an index <= that rule, with distance exactly 0 there and > 0 before it
    pick = torch.randint(0, R, (E,), generator=g, device="cuda")
    xi = uidx[torch.arange(E, device="cuda"), :, pick]                      # [E][nant] universe indices
    x = torch.stack([u[k][xi[:, k]] for k in range(nant)], dim=1).contiguous()
    d, hit = prob.rule_distance(x)
    torch.cuda.synchronize()
    hit = hit.long()
    assert (hit >= 0).all() and (hit <= pick).all()
    ar = torch.arange(E, device="cuda")
    assert (d[ar, hit] == 0).all()
    first_zero = (d == 0).float().argmax(dim=1)
    assert (first_zero == hit).all(), "hit is the FIRST zero-distance rule"
    assert torch.isfinite(d).all() and (d >= 0).all()
    # oracle on the first 16 environments at full rule count
    xs = x[:16].cpu().numpy()
    d_ref, hit_ref = small.oracle_rule_distance(np.ascontiguousarray(xs))
    assert (hit[:16].cpu().numpy() == hit_ref).all()
    assert (bits(d[:16].cpu().numpy()) == bits(d_ref)).all()


VARIANT_CASES = [(3, 41, 20000, 5, 3), (5, 1001, 4099, 9, 21), (16, 201, 3000, 3, 0), (16, 1001, 70001, 3, 0), (8, 1001, 5000, 3, 0), (5, 41, 367, 40, 3)]


@pytest.mark.parametrize("persist", [0, 1])
@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("nant,U,R,E,A", VARIANT_CASES)
def test_every_launch_form_is_bit_exact(nant, U, R, E, A, persist, order, hip_option):
    """The three launch forms of the scan -- one workgroup per (environment, chunk) item in chunk-fastest (shipped) or
    environment-fastest order, and the persistent form (table filled once, items from an in-order counter, observation VE
    values precomputed) -- forced for every shape through the "rd_persist" / "rd_order" options: same bits, same hit indices,
    also when the item count is not a multiple of the hand-out batch, with ragged / empty / odd rule bases."""
    tab = nant * U * 8
    if persist == 0 and tab > 64 * 1024:
        pytest.skip("one workgroup per item needs the table copy in 64 KiB")
    b = Batch(nant, U, R, E, A=A, seed=500 + nant + R, ragged=True)
    x = b.queries(seed=R + 9, hit_fraction=0.5)
    d_ref, hit_ref = b.oracle_rule_distance(x)
    hip_option("rd_persist", persist)
    hip_option("rd_order", order)
    for compressed in (True, False):
        d, hit = run_case(b, x, compressed=compressed)
        assert (hit.astype(np.int64) == hit_ref.astype(np.int64)).all(), (compressed, hit, hit_ref)
        for e in range(E):
            n = int(b.nrules[e])
            assert (bits(d[e, :n]) == bits(d_ref[e, :n])).all(), (compressed, e)
        _, hit2 = run_case(b, x, materialise=False, compressed=compressed)
        assert (hit2 == hit).all()


def test_persistent_form_many_environments_few_rules(hip_option):
    """Persistent form with more items than resident workgroups and a tiny last batch: 3000 environments x 600 rules."""
    hip_option("rd_persist", 1)
    b = Batch(5, 41, 600, 3000, A=3, seed=8, ragged=True)
    x = b.queries(seed=2, hit_fraction=0.3)
    d_ref, hit_ref = b.oracle_rule_distance(x)
    d, hit = run_case(b, x, compressed=True)
    assert (hit.astype(np.int64) == hit_ref.astype(np.int64)).all()
    for e in range(0, b.E, 7):
        n = int(b.nrules[e])
        assert (bits(d[e, :n]) == bits(d_ref[e, :n])).all(), e
