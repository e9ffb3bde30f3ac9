"""GPU parity of the single-rule-base, host-pointer entry points (five_hip_mirror_*: what the ANSI-C drop-in
library calls) against the oracle: five_rule_distance (ruledists host-visible, bit-exact), FIVE_vag_concl,
FIVE_vag_concl_weight, FIVEVagConcl_FRIRL_BestAct, FIVE_add_rule, five_remove_rule."""
import ctypes as C

import numpy as np
import pytest

import frirl_amd
from oracle import binding as ob
from tests.problems import Batch

pytestmark = pytest.mark.gpu
DP = C.POINTER(C.c_double)


def dp(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(DP)


@pytest.mark.parametrize("nant,U,R,A", [(3, 41, 110, 3), (5, 1001, 183, 21), (5, 41, 2049, 3), (8, 101, 700, 0)])
def test_mirror_matches_oracle(nant, U, R, A):
    L = frirl_amd.lib()
    b = Batch(nant, U, R, 1, A=A, seed=40 + R, ragged=False)
    maxR = b.maxR + 6
    f0 = b.five(0)
    f = ob.Five(b.u.ravel(), b.ve.ravel(), nant, U, maxR, np.ascontiguousarray(f0.rant[:R]), np.ascontiguousarray(f0.rconc[:R]))
    m = L.five_hip_mirror_create(nant, U, dp(np.ascontiguousarray(b.u.ravel())), dp(np.ascontiguousarray(b.ve.ravel())), maxR, 0)
    assert m, L.frirl_hip_last_error()
    try:
        keep = [np.ascontiguousarray(f.veval[k, :R]) for k in range(nant)]
        rows = (DP * nant)(*[dp(a) for a in keep])
        assert L.five_hip_mirror_upload(m, R, rows, dp(np.ascontiguousarray(f.rconc[:R]))) == 0
        assert L.five_hip_mirror_numofrules(m) == R
        rng = np.random.default_rng(1)
        hit = C.c_uint32()
        for q in range(12):
            if q % 3 == 0:
                x = np.ascontiguousarray(f.rant[rng.integers(f.R)])
            else:
                x = np.array([rng.uniform(b.u[k, 0], b.u[k, -2]) for k in range(nant)])
            d = np.full(f.R, np.nan)
            assert L.five_hip_mirror_rule_distance(m, dp(x), dp(d), C.byref(hit)) == 0
            ret = f.rule_distance(x)
            assert (-1 if hit.value == frirl_amd.NO_HIT else hit.value) == ret
            assert (d.view(np.uint64) == np.array(f.ruledists[: f.R]).view(np.uint64)).all(), "host-visible ruledists, bit-exact"
            conc = C.c_double()
            assert L.five_hip_mirror_vag_concl(m, dp(x), C.byref(conc), C.byref(hit)) == 0
            h, c = f.vag_concl(x)
            assert (-1 if hit.value == frirl_amd.NO_HIT else hit.value) == h
            assert conc.value == c if h >= 0 else abs(conc.value - c) <= 1e-11 * abs(c)
            w = np.full(f.R, np.nan)
            assert L.five_hip_mirror_vag_concl_weight(m, dp(x), dp(w), C.byref(hit)) == 0
            hw = f.vag_concl_weight(x)
            if hw == -1:
                assert np.abs(w - f.weights[: f.R]).max() <= 1e-11
                # FIVEVagConcl_FRIRL_BestAct on the distances of this observation
                f.rule_distance(x)
                dd = np.ascontiguousarray(f.ruledists[: f.R])
                assert L.five_hip_mirror_bestact(m, dp(dd), C.byref(conc)) == 0
                ref = ob.lib().orc_bestact(f.h, dp(dd))
                assert abs(conc.value - ref) <= 1e-11 * abs(ref)
            else:
                assert np.isnan(w).all()
            if q % 4 == 1:      # grow
                new = np.array([rng.uniform(b.u[k, 0], b.u[k, -2]) for k in range(nant)])
                ok_o = f.add_rule(new, 3.25 + q)
                rc = L.five_hip_mirror_add_rule(m, dp(new), 3.25 + q)
                assert (rc == 0) == (ok_o == 0)
            if q % 4 == 3:      # shrink
                r = int(rng.integers(f.R))
                assert ob.lib().orc_remove_rule(f.h, r) == 0
                assert L.five_hip_mirror_remove_rule(m, r) == 0
            assert L.five_hip_mirror_numofrules(m) == f.R
        rc_dev = np.zeros(f.R)
        assert L.five_hip_mirror_get_rconc(m, dp(rc_dev), f.R) == 0
        assert (rc_dev == f.rconc[: f.R]).all()
    finally:
        L.five_hip_mirror_destroy(m)
