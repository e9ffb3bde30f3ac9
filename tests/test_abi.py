"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every
symbol include/frirl_hip.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

import frirl_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    frirl_amd.build()
    return frirl_amd.lib()


def declared_functions():
    src = open(os.path.join(ROOT, "include", "frirl_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b((?:five|frirl)_hip_\w+)\s*\(", src)))


def test_header_symbols_exported(lib):
    names = declared_functions()
    assert "five_hip_rule_distance" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/frirl_hip.h but not exported by libfrirl_hip.so"
    assert set(frirl_amd.SIGNATURES) == set(names), "python binding and header disagree on the ABI surface"


def test_code_object_is_gfx950(lib):
    blob = open(frirl_amd.HIP_LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"gfx942" not in blob and b"sm_" not in blob


def test_struct_layouts_match_header():
    assert C.sizeof(frirl_amd.Tables) == 24 and frirl_amd.Tables.u.offset == 8 and frirl_amd.Tables.ve.offset == 16
    assert C.sizeof(frirl_amd.RuleBases) == 32 and frirl_amd.RuleBases.rb.offset == 8 and frirl_amd.RuleBases.nrules.offset == 16 and frirl_amd.RuleBases.uidx.offset == 24


def test_agent_and_envs_struct_layouts():
    A, E = frirl_amd.AgentDesc, frirl_amd.EnvsDesc
    assert A.skip_rules.offset == 40 and A.grid_len.offset == 64 and A.grid_div.offset == 128 and A.values_def.offset == 256
    assert A.grid_values.offset == 384 and A.action_ve.offset == 392 and A.epsilon.offset == 400 and A.seed.offset == 424 and A.evaluate.offset == 432 and A.env_id_base.offset == 440 and C.sizeof(A) == 448
    assert C.sizeof(E) == 96 and E.status.offset == 56 and E.episode.offset == 72 and E.spread_ant.offset == 80 and E.spread_R.offset == 88
    assert C.sizeof(frirl_amd.ConvergenceDesc) == 56 and frirl_amd.ConvergenceDesc.epended.offset == 48


def test_argument_validation_and_no_cpu_fallback(lib):
    import torch
    t = frirl_amd.Tables(3, 41, 0, 0)
    b = frirl_amd.RuleBases(1, 8, 0, 0)
    rc = lib.five_hip_rule_distance(C.byref(t), C.byref(b), None, None, None, None)
    assert rc == -2 and b"NULL" in lib.frirl_hip_last_error()
    if torch.cuda.is_available():
        pytest.skip("GPU present: the ENODEV path is exercised on CPU-only hosts")
    buf = (C.c_double * 4096)()
    addr = (C.addressof(buf) + 15) & ~15
    t = frirl_amd.Tables(3, 41, addr, addr)
    b = frirl_amd.RuleBases(1, 8, addr, addr)
    rc = lib.five_hip_rule_distance(C.byref(t), C.byref(b), addr, addr, addr, None)
    assert rc == -1, "without a GPU the hot path must fail loudly (FRIRL_HIP_ENODEV), never compute on the CPU"
    assert b"no CPU fallback" in lib.frirl_hip_last_error()
    assert lib.frirl_hip_device_count() <= 0


def test_missing_library_raises(monkeypatch, tmp_path):
    monkeypatch.setattr(frirl_amd, "_lib", None)
    monkeypatch.setattr(frirl_amd, "HIP_LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(frirl_amd.FrirlHipError):
        frirl_amd.lib()


def test_rollout_and_reduce_struct_layouts():
    R, Q = frirl_amd.RolloutDesc, frirl_amd.ReduceResult
    assert C.sizeof(R) == 56 and R.exclude_mask.offset == 8 and R.rule_slot.offset == 16 and R.steps.offset == 24 and R.final_states.offset == 48
    assert C.sizeof(Q) == 32 and Q.rules_after.offset == 4 and Q.rollouts.offset == 12 and Q.steps_incremental.offset == 16 and Q.reward.offset == 24


def test_shared_and_lane_entry_points_refuse_without_gpu(lib):
    """The evaluation / reduction / lane-group entry points validate their arguments and, on a host without a GPU, return
    FRIRL_HIP_ENODEV instead of computing anything (no CPU fallback anywhere in the product)."""
    import torch
    buf = (C.c_double * 8192)()
    addr = (C.addressof(buf) + 15) & ~15
    t = frirl_amd.Tables(3, 41, addr, addr)
    b1 = frirl_amd.RuleBases(1, 8, addr, addr)
    b2 = frirl_amd.RuleBases(2, 8, addr, addr)
    # argument validation comes first and does not need a device
    assert lib.five_hip_vag_concl_shared(C.byref(t), C.byref(b2), 0, 4, addr, addr, addr, None) == -2 and b"E == 1" in lib.frirl_hip_last_error()
    ag = frirl_amd.AgentDesc()
    ag.A, ag.env_kind, ag.max_steps, ag.grid_values, ag.action_ve = 3, 0, 10, addr, addr
    for k in range(3):
        ag.grid_len[k] = 3
    res = frirl_amd.ReduceResult()
    assert lib.frirl_hip_lanes_workspace_bytes(3, 100, 64, 3) == 128 * 4 * 64 * 8          # whole 64-environment tiles, f64 store
    assert lib.frirl_hip_lanes_preferred(3, 10, 3) == 1 and lib.frirl_hip_lanes_preferred(5, 96, 3) == 1 and lib.frirl_hip_lanes_preferred(0, 96, 3) == 0
    if torch.cuda.is_available():
        assert lib.frirl_hip_reduce_shared(C.byref(t), C.byref(b1), C.byref(ag), None, 3, 0.0, 0, None, C.byref(res), None) == -2
        assert lib.frirl_hip_reduce_shared(C.byref(t), C.byref(b1), C.byref(ag), None, 1, 0.0, 13, None, C.byref(res), None) == -2
        pytest.skip("GPU present: the ENODEV path is exercised on CPU-only hosts")
    ro = frirl_amd.RolloutDesc()
    ro.steps, ro.reward = addr, addr
    ev = frirl_amd.EnvsDesc()
    ev.states = ev.q_ant = ev.fus = ev.done = ev.ep_steps = ev.ep_reward = addr
    for rc in (lib.five_hip_vag_concl_shared(C.byref(t), C.byref(b1), 0, 4, addr, addr, addr, None),
               lib.frirl_hip_get_best_action_shared(C.byref(t), C.byref(b1), 0, 4, addr, addr, 3, addr, addr, None),
               lib.frirl_hip_rollout_shared(C.byref(t), C.byref(b1), C.byref(ag), 4, C.byref(ro), None),
               lib.frirl_hip_reduce_shared(C.byref(t), C.byref(b1), C.byref(ag), None, 1, 0.0, 0, None, C.byref(res), None),
               lib.frirl_hip_episode_run_lanes(C.byref(t), C.byref(b1), C.byref(ag), C.byref(ev), 5, addr, 1 << 20, None)):
        assert rc == -1, lib.frirl_hip_last_error()
        assert b"no CPU fallback" in lib.frirl_hip_last_error()


def test_learner_launch_plan_and_coverage(lib):
    """Host-side logic of the persistent learner (no GPU needed: a 256-CU chip is assumed when none is visible): which shapes it
    covers, and the launch plan -- lanes per agent 1 ... 64, never more agents than are alive or than fill the chip."""
    MC, CP, AC = 0, 1, 2
    assert lib.frirl_hip_learn_supported(3, 41, 3, 0, MC) == 1
    assert lib.frirl_hip_learn_supported(5, 41, 3, 0, AC) == 1
    assert lib.frirl_hip_learn_supported(5, 1001, 21, 0, CP) == 1          # 21 actions: two walks of 11 conclusions per step
    assert lib.frirl_hip_learn_supported(5, 41, 3, 3, AC) == 0             # a Shepard power other than nant
    assert lib.frirl_hip_learn_supported(5, 2001, 21, 0, CP) == 0          # universes beyond the 10-bit index fields
    assert lib.frirl_hip_learn_supported(4, 41, 3, 0, AC) == 0
    lanes = 256 * 4 * 2 * 64
    last_h = 0
    for n in (1, 63, 2048, 2049, 5000, 16384, 40000, 65536, 65537, 131072, 500000):
        H, take = C.c_int32(), C.c_int32()
        assert lib.frirl_hip_learn_plan(n, 200, C.byref(H), C.byref(take)) == 0
        assert H.value in (1, 2, 4, 8, 16, 32, 64) and 1 <= take.value <= n
        assert take.value * H.value <= lanes, (n, H.value, take.value)
        assert take.value == n or take.value == lanes // H.value              # everybody, or a full chip
        if last_h:
            assert H.value <= last_h                                          # more agents never get more lanes each
        last_h = H.value
    assert lib.frirl_hip_learn_plan(0, 0, C.byref(H), C.byref(take)) != 0
    a = lib.frirl_hip_learn_workspace_bytes(5, 4096, 512, 3)
    assert lib.frirl_hip_learn_train_workspace_bytes(5, 4096, 512, 3) >= a + 3 * 4096 * 4 > 0
