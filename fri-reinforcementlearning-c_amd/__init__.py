"""frirl_amd -- MI355X-native FRIRL / FIVE hot path (package directory: fri-reinforcementlearning-c_amd/).

The product is the C-ABI shared library lib/libfrirl_hip.so (hand-written gfx950 HIP kernels,
declared in include/frirl_hip.h) plus the ANSI-C drop-in host library that exports the
reference's five_* / FIVE_* / frirl_* API on top of it.  This Python module is only plumbing for
tests and bench.py: a ctypes binding that hands raw device pointers of torch tensors to the C ABI.
There is no CPU fallback: a missing library raises, a missing GPU makes every call return
FRIRL_HIP_ENODEV (raised as FrirlHipError).
"""
import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
HIP_LIB_PATH = os.environ.get("FRIRL_HIP_LIB_OVERRIDE") or os.path.join(PKG_DIR, "lib", "libfrirl_hip.so")      # override: A/B of experimental builds (tools/)

NO_HIT = 0xFFFFFFFF
MAX_NANT = 16
MAX_ACTIONS = 32


class FrirlHipError(RuntimeError):
    pass


class Tables(C.Structure):
    """struct frirl_hip_tables (include/frirl_hip.h)."""
    _fields_ = [("nant", C.c_int32), ("U", C.c_int32), ("u", C.c_void_p), ("ve", C.c_void_p)]


class RuleBases(C.Structure):
    """struct frirl_hip_rulebases (include/frirl_hip.h)."""
    _fields_ = [("E", C.c_int32), ("maxR", C.c_int32), ("rb", C.c_void_p), ("nrules", C.c_void_p), ("uidx", C.c_void_p)]


MAX_GRID = 64
ENV_KINDS = {"mountaincar": 0, "cartpole": 1, "acrobot": 2}
UPD_INACTIVE, UPD_EXACT, UPD_SPREAD, UPD_INSERTED, UPD_SKIPPED, UPD_FULL = range(6)


class AgentDesc(C.Structure):
    """struct frirl_hip_agent (include/frirl_hip.h)."""
    _fields_ = [("alpha", C.c_double), ("gamma", C.c_double), ("qdiff_pos_boundary", C.c_double), ("qdiff_neg_boundary", C.c_double),
                ("weight_significant", C.c_double), ("skip_rules", C.c_int32), ("p", C.c_int32), ("A", C.c_int32), ("env_kind", C.c_int32),
                ("max_steps", C.c_int32), ("no_random", C.c_int32), ("grid_len", C.c_int32 * MAX_NANT), ("grid_div", C.c_double * MAX_NANT),
                ("values_def", C.c_double * MAX_NANT), ("grid_values", C.c_void_p), ("action_ve", C.c_void_p), ("epsilon", C.c_double),
                ("reward_good_above", C.c_double), ("qdiff_final_tolerance", C.c_double), ("seed", C.c_uint64), ("evaluate", C.c_int32), ("debug_flags", C.c_int32), ("env_id_base", C.c_uint64)]


class EnvsDesc(C.Structure):
    """struct frirl_hip_envs (include/frirl_hip.h)."""
    _fields_ = [("states", C.c_void_p), ("q_ant", C.c_void_p), ("fus", C.c_void_p), ("done", C.c_void_p), ("ep_steps", C.c_void_p),
                ("ep_reward", C.c_void_p), ("rant", C.c_void_p), ("status", C.c_void_p), ("start_states", C.c_void_p), ("episode", C.c_void_p),
                ("spread_ant", C.c_void_p), ("spread_R", C.c_void_p)]


class RolloutDesc(C.Structure):
    """struct frirl_hip_rollout (include/frirl_hip.h)."""
    _fields_ = [("start_states", C.c_void_p), ("exclude_mask", C.c_void_p), ("rule_slot", C.c_void_p), ("steps", C.c_void_p),
                ("reward", C.c_void_p), ("success", C.c_void_p), ("final_states", C.c_void_p)]


class SenderDesc(C.Structure):
    """struct frirl_hip_sender (include/frirl_hip.h)."""
    _fields_ = [("rant", C.c_void_p), ("rule_stride", C.c_int64), ("dim_stride", C.c_int64), ("rconc", C.c_void_p), ("S", C.c_int32),
                ("reserved", C.c_int32), ("S_dev", C.c_void_p)]


class ReduceResult(C.Structure):
    """struct frirl_hip_reduce_result (include/frirl_hip.h)."""
    _fields_ = [("rules_before", C.c_int32), ("rules_after", C.c_int32), ("rounds", C.c_int32), ("rollouts", C.c_int32),
                ("steps_incremental", C.c_int32), ("reserved", C.c_int32), ("reward", C.c_double)]


class ConvergenceDesc(C.Structure):
    """struct frirl_hip_convergence (include/frirl_hip.h)."""
    _fields_ = [("prev_nrules", C.c_void_p), ("prev_steps", C.c_void_p), ("prev_reward", C.c_void_p), ("prev_rconc", C.c_void_p),
                ("converged", C.c_void_p), ("episodes", C.c_void_p), ("epended", C.c_void_p)]


_lib = None
_DP = C.POINTER(C.c_double)
LEARN_CHUNK_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32)      # frirl_hip_learn_chunk_fn

# name -> (restype, argtypes); every symbol include/frirl_hip.h declares
SIGNATURES = {
    "frirl_hip_version": (C.c_char_p, []),
    "frirl_hip_last_error": (C.c_char_p, []),
    "frirl_hip_device_count": (C.c_int, []),
    "frirl_hip_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "frirl_hip_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "frirl_hip_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "five_hip_rule_distance_uses_uidx": (C.c_int, [C.c_int32, C.c_int32]),
    "frirl_hip_step_uses_uidx": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "five_hip_rule_distance": (C.c_int, [C.POINTER(Tables), C.POINTER(RuleBases), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "five_hip_vag_concl": (C.c_int, [C.POINTER(Tables), C.POINTER(RuleBases), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "five_hip_vag_concl_weight": (C.c_int, [C.POINTER(Tables), C.POINTER(RuleBases), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "frirl_hip_get_best_action": (C.c_int, [C.POINTER(Tables), C.POINTER(RuleBases), C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p]),
    "five_hip_vag_concl_shared": (C.c_int, [C.POINTER(Tables), C.POINTER(RuleBases), C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "frirl_hip_get_best_action_shared": (C.c_int, [C.POINTER(Tables), C.POINTER(RuleBases), C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.c_int,
                                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "frirl_hip_rollout_shared": (C.c_int, [C.POINTER(Tables), C.POINTER(RuleBases), C.POINTER(AgentDesc), C.c_int32, C.POINTER(RolloutDesc), C.c_void_p]),
    "frirl_hip_reduce_shared": (C.c_int, [C.POINTER(Tables), C.POINTER(RuleBases), C.POINTER(AgentDesc), C.c_void_p, C.c_int, C.c_double, C.c_int,
                                          C.POINTER(C.c_int32), C.POINTER(ReduceResult), C.c_void_p]),
    "frirl_hip_lanes_preferred": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "frirl_hip_lanes_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "frirl_hip_episode_run_lanes": (C.c_int, [C.POINTER(Tables), C.POINTER(RuleBases), C.POINTER(AgentDesc), C.POINTER(EnvsDesc), C.c_int32,
                                              C.c_void_p, C.c_size_t, C.c_void_p]),
    "frirl_hip_rollout_resident_rules": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "frirl_hip_learn_supported": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "frirl_hip_learn_plan": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "frirl_hip_learn_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "frirl_hip_learn_run": (C.c_int, [C.POINTER(Tables), C.POINTER(RuleBases), C.POINTER(AgentDesc), C.POINTER(EnvsDesc), C.POINTER(ConvergenceDesc),
                                      C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "frirl_hip_learn_train_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "frirl_hip_learn_train": (C.c_int, [C.POINTER(Tables), C.POINTER(RuleBases), C.POINTER(AgentDesc), C.POINTER(EnvsDesc), C.POINTER(ConvergenceDesc),
                                        C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32),
                                        LEARN_CHUNK_FN, C.c_void_p, C.c_void_p]),
    "five_hip_add_rule": (C.c_int, [C.POINTER(Tables), C.POINTER(RuleBases), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "frirl_hip_update_sarsa": (C.c_int, [C.POINTER(Tables), C.POINTER(RuleBases), C.POINTER(AgentDesc), C.POINTER(EnvsDesc), C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "frirl_hip_env_step": (C.c_int, [C.POINTER(AgentDesc), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "frirl_hip_episode_begin": (C.c_int, [C.POINTER(Tables), C.POINTER(RuleBases), C.POINTER(AgentDesc), C.POINTER(EnvsDesc), C.c_void_p]),
    "frirl_hip_episode_step": (C.c_int, [C.POINTER(Tables), C.POINTER(RuleBases), C.POINTER(AgentDesc), C.POINTER(EnvsDesc), C.c_void_p]),
    "frirl_hip_episode_steps": (C.c_int, [C.POINTER(Tables), C.POINTER(RuleBases), C.POINTER(AgentDesc), C.POINTER(EnvsDesc), C.c_int32, C.c_void_p]),
    "frirl_hip_episode_run": (C.c_int, [C.POINTER(Tables), C.POINTER(RuleBases), C.POINTER(AgentDesc), C.POINTER(EnvsDesc), C.c_int32, C.c_int32, C.c_void_p]),
    "frirl_hip_convergence_init": (C.c_int, [C.POINTER(RuleBases), C.c_int, C.POINTER(ConvergenceDesc), C.c_void_p]),
    "frirl_hip_convergence_update": (C.c_int, [C.POINTER(RuleBases), C.c_int, C.POINTER(AgentDesc), C.POINTER(EnvsDesc), C.POINTER(ConvergenceDesc),
                                               C.c_void_p]),
    "frirl_hip_convergence_refresh": (C.c_int, [C.POINTER(RuleBases), C.c_int, C.POINTER(ConvergenceDesc), C.c_void_p]),
    "five_hip_bestact": (C.c_int, [C.POINTER(RuleBases), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    # library-owned batch with host descriptors (C-level many-agent runner)
    "frirl_hip_batch_create": (C.c_void_p, [C.c_void_p]),
    "frirl_hip_batch_destroy": (None, [C.c_void_p]),
    "frirl_hip_batch_episode": (C.c_int, [C.c_void_p]),
    "frirl_hip_batch_train": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "frirl_hip_batch_stats": (C.c_int, [C.c_void_p, C.c_void_p]),
    "frirl_hip_batch_get_rulebase": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), _DP, _DP]),
    "frirl_hip_batch_save_rulebases": (C.c_int, [C.c_void_p, C.c_char_p]),
    "frirl_hip_batch_load_rulebases": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int32)]),
    "frirl_hip_batch_reduce": (C.c_int, [C.c_void_p, C.c_int32, C.c_int, C.c_double, C.c_int, C.POINTER(ReduceResult)]),
    "frirl_hip_batch_merge_round": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "frirl_hip_batch_train_merged": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "frirl_hip_merge_rb": (C.c_int, [C.POINTER(Tables), C.POINTER(RuleBases), C.POINTER(AgentDesc), C.c_void_p, C.POINTER(SenderDesc), C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "frirl_hip_weights_from_spread": (C.c_int, [C.POINTER(Tables), C.POINTER(RuleBases), C.c_int, C.POINTER(EnvsDesc), C.c_void_p, C.c_void_p]),
    "frirl_hip_gen_def_states": (C.c_int, [_DP, C.c_int32, C.c_int32, C.c_int32, _DP, _DP]),
    # several GPUs from plain C (one batch + host thread per device, RCCL all-reduce of the report)
    "frirl_hip_shard": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "frirl_hip_multi_create": (C.c_void_p, [C.c_void_p, C.c_int64, C.c_int32]),
    "frirl_hip_multi_destroy": (None, [C.c_void_p]),
    "frirl_hip_multi_train": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "frirl_hip_multi_train_merged": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "frirl_hip_multi_stats": (C.c_int, [C.c_void_p, C.c_void_p]),
    "frirl_hip_multi_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "frirl_hip_multi_get_rulebase": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int32), _DP, _DP]),
    # single rule base, host pointers (what the ANSI-C drop-in library calls)
    "five_hip_mirror_create": (C.c_void_p, [C.c_int32, C.c_int32, _DP, _DP, C.c_int32, C.c_int32]),
    "five_hip_mirror_destroy": (None, [C.c_void_p]),
    "five_hip_mirror_upload": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(_DP), _DP]),
    "five_hip_mirror_add_rule": (C.c_int, [C.c_void_p, _DP, C.c_double]),
    "five_hip_mirror_remove_rule": (C.c_int, [C.c_void_p, C.c_uint32]),
    "five_hip_mirror_set_rconc": (C.c_int, [C.c_void_p, _DP, C.c_int32]),
    "five_hip_mirror_get_rconc": (C.c_int, [C.c_void_p, _DP, C.c_int32]),
    "five_hip_mirror_numofrules": (C.c_int32, [C.c_void_p]),
    "five_hip_mirror_rule_distance": (C.c_int, [C.c_void_p, _DP, _DP, C.POINTER(C.c_uint32)]),
    "five_hip_mirror_vag_concl": (C.c_int, [C.c_void_p, _DP, _DP, C.POINTER(C.c_uint32)]),
    "five_hip_mirror_vag_concl_weight": (C.c_int, [C.c_void_p, _DP, _DP, C.POINTER(C.c_uint32)]),
    "five_hip_mirror_bestact": (C.c_int, [C.c_void_p, _DP, _DP]),
    "five_hip_mirror_get_best_action": (C.c_int, [C.c_void_p, _DP, _DP, C.c_int32, _DP, C.POINTER(C.c_uint32)]),
    "five_hip_mirror_greedy_step": (C.c_int, [C.c_void_p, C.POINTER(AgentDesc), _DP, C.c_double, _DP, _DP, _DP, C.c_int32, C.POINTER(C.c_uint32), _DP, _DP,
                                              C.POINTER(C.c_int32), C.POINTER(C.c_int32), _DP, _DP, _DP]),
    "five_hip_mirror_update_sarsa": (C.c_int, [C.c_void_p, C.POINTER(AgentDesc), _DP, C.c_double, _DP, C.POINTER(C.c_int32),
                                               C.POINTER(C.c_int32), _DP, _DP, _DP]),
}


def build(force=False, verbose=False):
    import importlib.util
    spec = importlib.util.spec_from_file_location("frirl_amd_build", os.path.join(PKG_DIR, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build_all(force=force, verbose=verbose)


def lib():
    """The loaded C-ABI library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise FrirlHipError(f"{HIP_LIB_PATH} is missing: run `python __graft_entry__.py` (build()) first; "
                                "the FRIRL hot path has no CPU fallback")
        # One HIP runtime per process: torch bundles its own libamdhip64 (SONAME libamdhip64.so.7).  Loaded
        # first, it satisfies this library's NEEDED entry, so kernels, streams and events all live in the
        # runtime torch uses.  (A plain C host links /opt/rocm's runtime and never sees torch.)
        import torch  # noqa: F401
        L = C.CDLL(HIP_LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)      # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise FrirlHipError(f"{what}: rc={rc}: {lib().frirl_hip_last_error().decode()}")


def set_option(name, value):
    """frirl_hip_set_option: experiment / test switch (e.g. "no_uidx", "lanes_slices"); returns the previous value."""
    old = C.c_int()
    check(lib().frirl_hip_get_option(name.encode(), C.byref(old)), "frirl_hip_get_option")
    check(lib().frirl_hip_set_option(name.encode(), int(value)), "frirl_hip_set_option")
    return old.value


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(stream=None):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


class Problem:
    """Device-resident tables + E rule bases (torch tensors own the HBM; the C ABI gets raw pointers)."""

    def __init__(self, u, ve, rb, nrules, uidx=None):
        """uidx: optional int16 tensor [E, nant, maxR] holding the 16-bit universe indices of the antecedents
        (frirl_hip_rulebases.uidx): the compressed mirror the scans stream instead of the f64 columns."""
        import torch
        assert u.is_cuda and ve.is_cuda and rb.is_cuda and nrules.is_cuda
        assert u.dtype == torch.float64 and ve.dtype == torch.float64 and rb.dtype == torch.float64 and nrules.dtype == torch.int32
        self.nant, self.U = u.shape
        self.E, cols, self.maxR = rb.shape
        assert cols == self.nant + 1 and ve.shape == u.shape and nrules.shape == (self.E,)
        self.u, self.ve, self.rb, self.nrules = u.contiguous(), ve.contiguous(), rb.contiguous(), nrules.contiguous()
        self.uidx = None
        if uidx is not None:
            assert uidx.is_cuda and uidx.dtype == torch.int16 and uidx.shape == (self.E, self.nant, self.maxR) and self.U <= 32767
            self.uidx = uidx.contiguous()
        self.tables = Tables(self.nant, self.U, self.u.data_ptr(), self.ve.data_ptr())
        self.bases = RuleBases(self.E, self.maxR, self.rb.data_ptr(), self.nrules.data_ptr(),
                               self.uidx.data_ptr() if self.uidx is not None else None)

    def rule_distance(self, x, ruledists=None, hit=None, materialise=True, stream=None):
        """five_hip_rule_distance: returns (ruledists [E,maxR] or None, hit [E] int32 with -1 = none)."""
        import torch
        assert x.is_cuda and x.dtype == torch.float64 and x.shape == (self.E, self.nant) and x.is_contiguous()
        if materialise and ruledists is None:
            ruledists = torch.empty((self.E, self.maxR), dtype=torch.float64, device=x.device)
        if hit is None:
            hit = torch.empty((self.E,), dtype=torch.int32, device=x.device)
        rc = lib().five_hip_rule_distance(C.byref(self.tables), C.byref(self.bases), _ptr(x),
                                          _ptr(ruledists) if materialise else None, _ptr(hit), _stream(stream))
        check(rc, "five_hip_rule_distance")
        return (ruledists if materialise else None), hit

    def vag_concl(self, x, p=0, stream=None):
        """five_hip_vag_concl: (conc [E] float64, hit [E] int32 with -1 = interpolated)."""
        import torch
        assert x.is_cuda and x.dtype == torch.float64 and x.shape == (self.E, self.nant) and x.is_contiguous()
        conc = torch.empty((self.E,), dtype=torch.float64, device=x.device)
        hit = torch.empty((self.E,), dtype=torch.int32, device=x.device)
        check(lib().five_hip_vag_concl(C.byref(self.tables), C.byref(self.bases), p, _ptr(x), _ptr(conc), _ptr(hit), _stream(stream)),
              "five_hip_vag_concl")
        return conc, hit

    def vag_concl_weight(self, x, p=0, weights=None, stream=None):
        """five_hip_vag_concl_weight: (weights [E,maxR], hit [E]); rows of exact-hit environments are untouched."""
        import torch
        assert x.is_cuda and x.dtype == torch.float64 and x.shape == (self.E, self.nant) and x.is_contiguous()
        if weights is None:
            weights = torch.full((self.E, self.maxR), float("nan"), dtype=torch.float64, device=x.device)
        hit = torch.empty((self.E,), dtype=torch.int32, device=x.device)
        check(lib().five_hip_vag_concl_weight(C.byref(self.tables), C.byref(self.bases), p, _ptr(x), _ptr(weights), _ptr(hit),
                                              _stream(stream)), "five_hip_vag_concl_weight")
        return weights, hit

    def get_best_action(self, states, action_ve, p=0, stream=None):
        """frirl_hip_get_best_action: (actconc [E,A], best [E] int32)."""
        import torch
        assert states.is_cuda and states.dtype == torch.float64 and states.shape == (self.E, self.nant - 1) and states.is_contiguous()
        A = action_ve.numel()
        actconc = torch.empty((self.E, A), dtype=torch.float64, device=states.device)
        best = torch.empty((self.E,), dtype=torch.int32, device=states.device)
        check(lib().frirl_hip_get_best_action(C.byref(self.tables), C.byref(self.bases), p, _ptr(states), _ptr(action_ve), A,
                                              _ptr(actconc), _ptr(best), _stream(stream)), "frirl_hip_get_best_action")
        return actconc, best

    def vag_concl_shared(self, x, p=0, stream=None):
        """five_hip_vag_concl_shared: Q observations against this ONE rule base (E == 1)."""
        import torch
        assert self.E == 1 and x.is_cuda and x.dtype == torch.float64 and x.shape[1] == self.nant and x.is_contiguous()
        Q = x.shape[0]
        conc = torch.empty((Q,), dtype=torch.float64, device=x.device)
        hit = torch.empty((Q,), dtype=torch.int32, device=x.device)
        check(lib().five_hip_vag_concl_shared(C.byref(self.tables), C.byref(self.bases), p, Q, _ptr(x), _ptr(conc), _ptr(hit), _stream(stream)),
              "five_hip_vag_concl_shared")
        return conc, hit

    def get_best_action_shared(self, states, action_ve, p=0, stream=None):
        """frirl_hip_get_best_action_shared: Q states against this ONE rule base (E == 1)."""
        import torch
        assert self.E == 1 and states.is_cuda and states.dtype == torch.float64 and states.shape[1] == self.nant - 1 and states.is_contiguous()
        Q, A = states.shape[0], action_ve.numel()
        actconc = torch.empty((Q, A), dtype=torch.float64, device=states.device)
        best = torch.empty((Q,), dtype=torch.int32, device=states.device)
        check(lib().frirl_hip_get_best_action_shared(C.byref(self.tables), C.byref(self.bases), p, Q, _ptr(states), _ptr(action_ve), A, _ptr(actconc),
                                                     _ptr(best), _stream(stream)), "frirl_hip_get_best_action_shared")
        return actconc, best

    def rollout_shared(self, agent, Q, start_states=None, exclude_mask=None, rule_slot=None, stream=None):
        """frirl_hip_rollout_shared: Q greedy roll-outs (frirl_test_run's episode) on this ONE rule base (E == 1).
        Returns (steps[Q] i32, reward[Q] f64, success[Q] i32, final_states[Q][nant-1])."""
        import torch
        dev_ = self.rb.device
        assert self.E == 1
        ro = RolloutDesc()
        steps = torch.empty((Q,), dtype=torch.int32, device=dev_)
        reward = torch.empty((Q,), dtype=torch.float64, device=dev_)
        success = torch.empty((Q,), dtype=torch.int32, device=dev_)
        final = torch.empty((Q, self.nant - 1), dtype=torch.float64, device=dev_)
        if start_states is not None:
            assert start_states.shape == (Q, self.nant - 1) and start_states.dtype == torch.float64 and start_states.is_contiguous()
            ro.start_states = _ptr(start_states)
        if exclude_mask is not None:
            assert exclude_mask.shape == (Q,) and exclude_mask.dtype == torch.int32 and rule_slot.dtype == torch.uint8 and rule_slot.numel() == self.maxR
            ro.exclude_mask, ro.rule_slot = _ptr(exclude_mask), _ptr(rule_slot)
        ro.steps, ro.reward, ro.success, ro.final_states = _ptr(steps), _ptr(reward), _ptr(success), _ptr(final)
        check(lib().frirl_hip_rollout_shared(C.byref(self.tables), C.byref(self.bases), C.byref(agent.desc), Q, C.byref(ro), _stream(stream)),
              "frirl_hip_rollout_shared")
        return steps, reward, success, final

    def reduce_shared(self, agent, strategy, reward_tolerance=0.0, depth=0, rant=None, stream=None):
        """frirl_hip_reduce_shared: the reference's rule-base reduction as speculative batched try-remove; compacts this
        ONE rule base in place.  Returns (kept original indices, ReduceResult)."""
        import numpy as np
        assert self.E == 1
        R0 = int(self.nrules[0].item())
        kept = np.zeros(R0, dtype=np.int32)
        res = ReduceResult()
        check(lib().frirl_hip_reduce_shared(C.byref(self.tables), C.byref(self.bases), C.byref(agent.desc), _ptr(rant) if rant is not None else None,
                                            strategy, reward_tolerance, depth, kept.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(res), _stream(stream)),
              "frirl_hip_reduce_shared")
        return kept[: res.rules_after], res

    def merge_rb(self, agent, sndr_rant, sndr_rconc, weights, rant_store=None, active=None, stream=None):
        """frirl_hip_merge_rb: every (active) rule base of this batch takes over the sender rules sndr_rant [S][nant] (AoS),
        sndr_rconc [S]; weights [E][maxR] persists between calls (zeros at first).  Returns full [E] int32."""
        import torch
        S = sndr_rconc.numel()
        assert sndr_rant.shape == (S, self.nant) and sndr_rant.is_contiguous() and weights.shape == (self.E, self.maxR)
        snd = SenderDesc(sndr_rant.data_ptr(), self.nant, 1, sndr_rconc.data_ptr(), S, 0, None)
        full = torch.zeros((self.E,), dtype=torch.int32, device=self.rb.device)
        check(lib().frirl_hip_merge_rb(C.byref(self.tables), C.byref(self.bases), C.byref(agent.desc), _ptr(rant_store), C.byref(snd), _ptr(weights),
                                       _ptr(active), _ptr(full), _stream(stream)), "frirl_hip_merge_rb")
        return full

    def add_rule(self, rant, rconc, active=None, rant_store=None, stream=None):
        """five_hip_add_rule: appends rant[e] -> rconc[e]; returns added [E] int32."""
        import torch
        added = torch.empty((self.E,), dtype=torch.int32, device=rant.device)
        check(lib().five_hip_add_rule(C.byref(self.tables), C.byref(self.bases), _ptr(rant), _ptr(rconc), _ptr(active), _ptr(rant_store),
                                      _ptr(added), _stream(stream)), "five_hip_add_rule")
        return added


class Agent:
    """Host-side frirl_hip_agent: hyper-parameters + grids (device copies of the small tables are owned here)."""

    def __init__(self, device, nant, grids, grid_div, values_def, action_ve, alpha, gamma, qdiff_pos, qdiff_neg, weight_thr=0.05,
                 skip_rules=1, p=0, env_kind=0, max_steps=1000, epsilon=0.0, no_random=1, seed=0, reward_good_above=0.0,
                 qdiff_final_tolerance=250.0, env_id_base=0, evaluate=0, debug_flags=0):
        import numpy as np
        import torch
        assert len(grids) == nant and all(1 <= len(g) <= MAX_GRID for g in grids)
        gv = np.zeros((nant, MAX_GRID))
        for k, g in enumerate(grids):
            gv[k, : len(g)] = g
        self.grid_values = torch.from_numpy(gv).to(device)
        self.action_ve = torch.as_tensor(np.asarray(action_ve, dtype=np.float64)).to(device)
        self.A = len(grids[-1])
        assert self.action_ve.numel() == self.A
        d = AgentDesc()
        d.alpha, d.gamma, d.qdiff_pos_boundary, d.qdiff_neg_boundary = alpha, gamma, qdiff_pos, qdiff_neg
        d.weight_significant, d.skip_rules, d.p, d.A, d.env_kind, d.max_steps = weight_thr, skip_rules, p, self.A, env_kind, max_steps
        for k in range(nant):
            d.grid_len[k] = len(grids[k])
            d.grid_div[k] = grid_div[k] if k < len(grid_div) else 0.0
            d.values_def[k] = values_def[k] if k < len(values_def) else 0.0
        d.grid_values, d.action_ve = self.grid_values.data_ptr(), self.action_ve.data_ptr()
        d.no_random, d.epsilon, d.seed, d.env_id_base, d.evaluate = no_random, epsilon, seed, env_id_base, evaluate
        d.debug_flags = debug_flags
        d.reward_good_above, d.qdiff_final_tolerance = reward_good_above, qdiff_final_tolerance
        self.desc, self.nant = d, nant


class Envs:
    """Device-resident per-environment episode state (struct frirl_hip_envs)."""

    def __init__(self, problem, device, keep_rant=True, rant_init=None, start_states=None):
        import torch
        E, nant, maxR = problem.E, problem.nant, problem.maxR
        self.start_states = start_states
        self.episode = torch.zeros((E,), dtype=torch.int32, device=device)
        self.states = torch.zeros((E, nant - 1), dtype=torch.float64, device=device)
        self.q_ant = torch.zeros((E, nant), dtype=torch.float64, device=device)
        self.fus = torch.zeros((E,), dtype=torch.int32, device=device)
        self.done = torch.zeros((E,), dtype=torch.int32, device=device)
        self.ep_steps = torch.zeros((E,), dtype=torch.int32, device=device)
        self.ep_reward = torch.zeros((E,), dtype=torch.float64, device=device)
        self.status = torch.zeros((E,), dtype=torch.int32, device=device)
        self.rant = None
        if keep_rant:
            self.rant = torch.zeros((E, nant, maxR), dtype=torch.float64, device=device) if rant_init is None else rant_init
        self.spread_ant = torch.zeros((E, nant), dtype=torch.float64, device=device)     # what determines FIVERB.weights after learning
        self.spread_R = torch.zeros((E,), dtype=torch.int32, device=device)
        self.desc = EnvsDesc(self.states.data_ptr(), self.q_ant.data_ptr(), self.fus.data_ptr(), self.done.data_ptr(), self.ep_steps.data_ptr(),
                             self.ep_reward.data_ptr(), self.rant.data_ptr() if self.rant is not None else None, self.status.data_ptr(),
                             self.start_states.data_ptr() if self.start_states is not None else None, self.episode.data_ptr(),
                             self.spread_ant.data_ptr(), self.spread_R.data_ptr())


def update_sarsa(problem, agent, envs, q_ant, reward, cur_q_ant, active=None, stream=None):
    check(lib().frirl_hip_update_sarsa(C.byref(problem.tables), C.byref(problem.bases), C.byref(agent.desc), C.byref(envs.desc), _ptr(q_ant),
                                       _ptr(reward), _ptr(cur_q_ant), _ptr(active), _stream(stream)), "frirl_hip_update_sarsa")


def env_step(agent, action, states, stream=None):
    import torch
    E, ns = states.shape
    new_states, q_states = torch.empty_like(states), torch.empty_like(states)
    reward = torch.empty((E,), dtype=torch.float64, device=states.device)
    success = torch.empty((E,), dtype=torch.int32, device=states.device)
    check(lib().frirl_hip_env_step(C.byref(agent.desc), E, ns, _ptr(action), _ptr(states), _ptr(new_states), _ptr(reward), _ptr(success),
                                   _ptr(q_states), _stream(stream)), "frirl_hip_env_step")
    return new_states, reward, success, q_states


def episode_begin(problem, agent, envs, stream=None):
    check(lib().frirl_hip_episode_begin(C.byref(problem.tables), C.byref(problem.bases), C.byref(agent.desc), C.byref(envs.desc),
                                        _stream(stream)), "frirl_hip_episode_begin")


def episode_step(problem, agent, envs, stream=None):
    check(lib().frirl_hip_episode_step(C.byref(problem.tables), C.byref(problem.bases), C.byref(agent.desc), C.byref(envs.desc),
                                       _stream(stream)), "frirl_hip_episode_step")


def dist():
    """The multi-GPU helper module (fri-reinforcementlearning-c_amd/dist.py)."""
    import importlib.util
    import sys
    if "frirl_amd_dist" in sys.modules:
        return sys.modules["frirl_amd_dist"]
    spec = importlib.util.spec_from_file_location("frirl_amd_dist", os.path.join(PKG_DIR, "dist.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["frirl_amd_dist"] = mod
    spec.loader.exec_module(mod)
    return mod


DROPIN_LIB_PATH = os.path.join(PKG_DIR, "lib", "libfrirl_dropin.so")
_dropin = None


def dropin():
    """ctypes handle of the ANSI-C drop-in library (the reference's five_* / FIVE_* / frirl_* API)."""
    global _dropin
    if _dropin is None:
        lib()                                   # loads torch's HIP runtime + libfrirl_hip.so first
        if not os.path.exists(DROPIN_LIB_PATH):
            raise FrirlHipError(f"{DROPIN_LIB_PATH} is missing: run build() first")
        D = C.CDLL(DROPIN_LIB_PATH)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        D.frirl_demo_describe.restype = C.c_int
        D.frirl_demo_describe.argtypes = [C.c_char_p, ip, ip, ip, dp, dp, dp, ip, dp, dp, dp, dp, ip]
        _dropin = D
    return _dropin


def episode_steps(problem, agent, envs, nsteps, stream=None):
    check(lib().frirl_hip_episode_steps(C.byref(problem.tables), C.byref(problem.bases), C.byref(agent.desc), C.byref(envs.desc), nsteps,
                                        _stream(stream)), "frirl_hip_episode_steps")


def episode_run(problem, agent, envs, nsteps, lds_rules, stream=None):
    """frirl_hip_episode_run: up to nsteps steps per environment in ONE launch, rule bases resident in LDS."""
    check(lib().frirl_hip_episode_run(C.byref(problem.tables), C.byref(problem.bases), C.byref(agent.desc), C.byref(envs.desc), nsteps, lds_rules,
                                      _stream(stream)), "frirl_hip_episode_run")


def episode_run_lanes(problem, agent, envs, nsteps, workspace=None, stream=None):
    """frirl_hip_episode_run_lanes: lane-group episodes for many agents with small rule bases; `workspace` is a
    float64 CUDA tensor of >= lanes_workspace_elems(problem, agent) elements (allocated and cached on the problem if None)."""
    import torch
    need = lib().frirl_hip_lanes_workspace_bytes(problem.nant, problem.E, problem.maxR, agent.A)
    if workspace is None:
        workspace = getattr(problem, "_lanes_ws", None)
        if workspace is None or workspace.numel() * 8 < need:
            workspace = torch.empty((need // 8,), dtype=torch.float64, device=problem.rb.device)
            problem._lanes_ws = workspace
    check(lib().frirl_hip_episode_run_lanes(C.byref(problem.tables), C.byref(problem.bases), C.byref(agent.desc), C.byref(envs.desc), nsteps,
                                            _ptr(workspace), workspace.numel() * 8, _stream(stream)), "frirl_hip_episode_run_lanes")


def can_run_persistent(problem, agent):
    return agent.A <= 8 and 2 * 8 * problem.nant * problem.U <= 16 * 1024


class Convergence:
    """Device-resident construct-loop bookkeeping (struct frirl_hip_convergence)."""

    def __init__(self, problem, device):
        import torch
        E, maxR = problem.E, problem.maxR
        self.prev_nrules = torch.zeros((E,), dtype=torch.int32, device=device)
        self.prev_steps = torch.zeros((E,), dtype=torch.int32, device=device)
        self.prev_reward = torch.zeros((E,), dtype=torch.float64, device=device)
        self.prev_rconc = torch.zeros((E, maxR), dtype=torch.float64, device=device)
        self.converged = torch.zeros((E,), dtype=torch.int32, device=device)
        self.episodes = torch.zeros((E,), dtype=torch.int32, device=device)
        self.full = torch.zeros((E,), dtype=torch.bool, device=device)     # agents whose rule base refused an append (capacity)
        self.epended = torch.zeros((E,), dtype=torch.int32, device=device)  # the cheap "same as the previous episode" test held (frirl_desc.epended)
        self.desc = ConvergenceDesc(self.prev_nrules.data_ptr(), self.prev_steps.data_ptr(), self.prev_reward.data_ptr(),
                                    self.prev_rconc.data_ptr(), self.converged.data_ptr(), self.episodes.data_ptr(), self.epended.data_ptr())
        check(lib().frirl_hip_convergence_init(C.byref(problem.bases), problem.nant, C.byref(self.desc), _stream()), "frirl_hip_convergence_init")

    @property
    def full_envs(self):
        """Number of agents that hit FRIRL_HIP_UPD_FULL at least once: their TD updates were dropped from then on."""
        return int(self.full.sum().item())

    def update(self, problem, agent, envs, stream=None):
        self.full |= (envs.status == UPD_FULL) | (problem.nrules >= problem.maxR)     # an append was (or will be) refused
        check(lib().frirl_hip_convergence_update(C.byref(problem.bases), problem.nant, C.byref(agent.desc), C.byref(envs.desc), C.byref(self.desc),
                                                 _stream(stream)), "frirl_hip_convergence_update")


def train(problem, agent, envs, max_episodes=1000, check_every=50, on_episode=None, persistent=True, persistent_max_rules=256, lanes=None):
    """Batched construct run: frirl_sequential_run's loop (reference frirl_sequential_run.c:55-165) for E agents
    at once.  Episodes run until every environment's rule base is "considered complete" or max_episodes-1 episodes
    have run (:51,59).  Converged environments are masked out of later episodes.  Returns the Convergence object."""
    import torch
    conv = Convergence(problem, problem.rb.device)
    max_steps = agent.desc.max_steps
    if lanes is None:           # lane-group kernel where it is the faster form (many agents / small rule bases)
        lanes = bool(lib().frirl_hip_lanes_preferred(problem.nant, problem.E, agent.A))
    for ep in range(1, max_episodes):
        episode_begin(problem, agent, envs)
        envs.done.copy_(torch.maximum(envs.done, conv.converged))       # converged agents sit this episode out
        steps = 0
        if lanes:
            # many agents, small rule bases: lane-group kernel, the whole episode in one launch (no reductions / barriers)
            episode_run_lanes(problem, agent, envs, max_steps)
            steps = max_steps
        elif persistent and can_run_persistent(problem, agent):
            # small rule bases: the whole episode in one launch out of LDS; an environment whose rule base outgrows
            # the LDS slab comes back not-done (status FULL) and finishes through the step kernel below
            # (measured: it pays only while the LDS slab is small enough for >= ~12 waves per CU, i.e. the 256-rule slab;
            #  larger slabs cut the number of resident environments more than they cut the per-step latency)
            need = int(problem.nrules.max().item()) + 128
            if need <= persistent_max_rules:
                episode_run(problem, agent, envs, max_steps, 256 if need <= 256 else (512 if need <= 512 else 1024))
                if bool((envs.done != 0).all()):
                    steps = max_steps
        while steps < max_steps:
            n = min(check_every, max_steps - steps)
            episode_steps(problem, agent, envs, n)
            steps += n
            if bool((envs.done != 0).all()):
                break
        conv.update(problem, agent, envs)
        if on_episode is not None:
            on_episode(ep, conv)
        if bool((conv.converged != 0).all()):
            break
    if conv.full_envs:
        import warnings
        warnings.warn(f"frirl_amd.train: {conv.full_envs} of {problem.E} rule bases reached their capacity of {problem.maxR} rules: "
                      "appends were refused (FRIRL_HIP_UPD_FULL) and those TD updates dropped", RuntimeWarning)
    return conv


def learn_supported(problem, agent):
    """True when frirl_hip_learn_run (the persistent construct loop) covers this shape."""
    return problem.uidx is not None and bool(lib().frirl_hip_learn_supported(problem.nant, problem.U, agent.A, agent.desc.p, agent.desc.env_kind))


class LearnRun:
    """Result of train_persistent: the Convergence object plus per-agent totals (device tensors)."""

    def __init__(self, conv, steps_total, work, launches):
        self.conv, self.steps_total, self.work, self.launches = conv, steps_total, work, launches


def train_persistent(problem, agent, envs, max_episodes=1000, budget=4096, on_chunk=None, stream=None):
    """The construct loop of frirl_sequential_run for E agents through frirl_hip_learn_train: every agent runs its episodes
    back to back on the device; after each launch (the work of `budget` steps of a mean agent) the agents that are still
    learning are compacted and launched again until none is left -- the launch plan, the queue and the compaction are the
    library's (csrc/learn.hip), the host reads two counters per launch.  on_chunk(launch index, live agent ids, conv) is
    called after every launch (the place for the per-chunk reward statistics).  Returns a LearnRun."""
    import torch
    dev_ = problem.rb.device
    conv = Convergence(problem, dev_)
    envs.done.fill_(1)                                   # fresh agents: every one starts its first episode
    steps_total = torch.zeros((problem.E,), dtype=torch.int64, device=dev_)
    work = torch.zeros((problem.E, 2), dtype=torch.int64, device=dev_)
    need = lib().frirl_hip_learn_train_workspace_bytes(problem.nant, problem.E, problem.maxR, agent.A)
    ws = getattr(problem, "_learn_ws", None)
    if ws is None or ws.numel() * 8 < need:
        ws = torch.empty(((need + 7) // 8,), dtype=torch.float64, device=dev_)
        problem._learn_ws = ws
    problem._learn_progress = (steps_total, work)        # for on_chunk callbacks that report per-launch progress
    ws32 = ws.view(torch.int32)
    failure = []

    def chunk(_user, launch, live_ptr, n):               # `live` is a slice of the workspace
        try:
            off = (live_ptr - ws.data_ptr()) // 4
            on_chunk(launch, ws32[off:off + n], conv)
        except BaseException as exc:                     # an exception must not unwind through the C frame
            failure.append(exc)
    cb = LEARN_CHUNK_FN(chunk) if on_chunk is not None else C.cast(None, LEARN_CHUNK_FN)
    launches = C.c_int32(0)
    refused = conv.full.view(torch.uint8)
    check(lib().frirl_hip_learn_train(C.byref(problem.tables), C.byref(problem.bases), C.byref(agent.desc), C.byref(envs.desc), C.byref(conv.desc),
                                      budget, max_episodes, _ptr(work), _ptr(steps_total), _ptr(refused), _ptr(ws), ws.numel() * 8,
                                      C.byref(launches), cb, None, _stream(stream)), "frirl_hip_learn_train")
    if failure:
        raise failure[0]
    if conv.full_envs:
        import warnings
        warnings.warn(f"frirl_amd.train_persistent: {conv.full_envs} of {problem.E} rule bases reached their capacity of {problem.maxR} rules: "
                      "appends were refused (FRIRL_HIP_UPD_FULL) and those TD updates dropped", RuntimeWarning)
    return LearnRun(conv, steps_total, work, launches.value)


def demo_describe(env):
    """Tables, grids and hyper-parameters of a demo, built by the drop-in library's own host functions
    (frirl_init_ve etc.); no GPU needed."""
    import numpy as np
    D = dropin()
    ns, U, A, ms = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    if D.frirl_demo_describe(env.encode(), C.byref(ns), C.byref(U), C.byref(A), None, None, None, None, None, None, None, None, C.byref(ms)) != 0:
        raise ValueError(f"unknown demo environment {env!r}")
    nant = ns.value + 1
    u, ve = np.zeros((nant, U.value)), np.zeros((nant, U.value))
    grid = np.zeros((nant, MAX_GRID))
    grid_len = np.zeros(nant, dtype=np.int32)
    grid_div, values_def = np.zeros(nant), np.zeros(nant)
    action_ve, hp = np.zeros(A.value), np.zeros(8)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    rc = D.frirl_demo_describe(env.encode(), C.byref(ns), C.byref(U), C.byref(A), u.ctypes.data_as(dp), ve.ctypes.data_as(dp),
                               grid.ctypes.data_as(dp), grid_len.ctypes.data_as(ip), grid_div.ctypes.data_as(dp),
                               values_def.ctypes.data_as(dp), action_ve.ctypes.data_as(dp), hp.ctypes.data_as(dp), C.byref(ms))
    assert rc == 0
    return dict(env=env, kind=ENV_KINDS[env], nstates=ns.value, nant=nant, U=U.value, A=A.value, u=u, ve=ve,
                grids=[grid[k, : grid_len[k]].copy() for k in range(nant)], grid_div=grid_div, values_def=values_def, action_ve=action_ve,
                alpha=hp[0], gamma=hp[1], qdiff_pos=hp[2], qdiff_neg=hp[3], weight_thr=hp[4], skip_rules=int(hp[5]),
                reward_good_above=hp[6], qdiff_final_tolerance=hp[7], max_steps=ms.value)


def demo_batch(env, E, R, maxR, device, seed=0, max_steps=None, compressed=True, keep_rant=True):
    """E environments of a demo on `device`, each with a private synthetic rule base of R rules: the 2^nant
    corner rules first (reference frirl_init_rb.c:99-126, Q = 0), then rules on the universe grid (uniform
    indices; action column on the A action values) with Q ~ U(-1500, 1500).
    Returns (Problem, Agent, Envs)."""
    import numpy as np
    import torch
    d = demo_describe(env)
    nant, U, A = d["nant"], d["U"], d["A"]
    assert maxR >= R and maxR % 2 == 0 and R >= 2 ** nant
    g = torch.Generator(device=device).manual_seed(0x5EED0000 + seed)
    u_d, ve_d = torch.from_numpy(d["u"]).to(device), torch.from_numpy(d["ve"]).to(device)
    rb = torch.zeros((E, nant + 1, maxR), dtype=torch.float64, device=device)
    rant = torch.zeros((E, nant, maxR), dtype=torch.float64, device=device) if keep_rant else None     # raw antecedents (rule-base dumps only)
    uidx = torch.zeros((E, nant, maxR), dtype=torch.int16, device=device) if compressed else None
    ncorner = 2 ** nant
    # action universe index of every action value (nearest universe point, as FIVE_add_rule snaps it)
    ua = d["u"][nant - 1]
    a_idx = torch.tensor([int(np.argmin(np.abs(ua - v))) for v in d["grids"][nant - 1]], device=device)
    for k in range(nant):
        idx = torch.randint(0, U, (E, R), generator=g, device=device)
        if k == nant - 1:
            idx = a_idx[torch.randint(0, A, (E, R), generator=g, device=device)]
        gk = d["grids"][k]
        lo_i, hi_i = int(np.argmin(np.abs(d["u"][k] - gk.min()))), int(np.argmin(np.abs(d["u"][k] - gk.max())))
        divider = ncorner >> (k + 1)
        corner = torch.tensor([lo_i if ((j // divider) % 2) == 0 else hi_i for j in range(ncorner)], device=device)
        idx[:, :ncorner] = corner[None]
        rb[:, k, :R] = ve_d[k][idx]
        if keep_rant:
            rant[:, k, :R] = u_d[k][idx]
        if compressed:
            uidx[:, k, :R] = idx.to(torch.int16)
        del idx
    rb[:, nant, ncorner:R] = torch.rand((E, R - ncorner), generator=g, device=device, dtype=torch.float64) * 3000.0 - 1500.0
    nrules = torch.full((E,), R, dtype=torch.int32, device=device)
    prob = Problem(u_d, ve_d, rb, nrules, uidx)
    agent = demo_agent(d, device, max_steps)
    envs = Envs(prob, device, keep_rant=keep_rant, rant_init=rant)
    return prob, agent, envs


def demo_agent(d, device, max_steps=None, p=0, **kw):
    return Agent(device, d["nant"], d["grids"], d["grid_div"], d["values_def"], d["action_ve"], d["alpha"], d["gamma"], d["qdiff_pos"],
                 d["qdiff_neg"], d["weight_thr"], d["skip_rules"], p, d["kind"], max_steps or d["max_steps"],
                 reward_good_above=d["reward_good_above"], qdiff_final_tolerance=d["qdiff_final_tolerance"], **kw)


def demo_fresh_batch(env, E, maxR, device, start_states=None, **agent_kw):
    """E agents of a demo starting from the reference's initial rule base: the 2^nant corner rules with Q = 0
    (frirl_init_rb.c:99-126), grid min/max as antecedents.  Returns (Problem, Agent, Envs)."""
    import numpy as np
    import torch
    d = demo_describe(env)
    nant, U = d["nant"], d["U"]
    ncorner = 2 ** nant
    assert maxR % 2 == 0 and maxR >= ncorner
    u_d, ve_d = torch.from_numpy(d["u"]).to(device), torch.from_numpy(d["ve"]).to(device)
    rant0 = np.zeros((nant, ncorner))
    for k in range(nant):
        gk = d["grids"][k]
        divider = ncorner >> (k + 1)
        rant0[k] = [gk.min() if ((j // divider) % 2) == 0 else gk.max() for j in range(ncorner)]
    prob = Problem(u_d, ve_d, torch.zeros((E, nant + 1, maxR), dtype=torch.float64, device=device),
                   torch.zeros((E,), dtype=torch.int32, device=device), torch.zeros((E, nant, maxR), dtype=torch.int16, device=device))
    agent = demo_agent(d, device, **agent_kw)
    envs = Envs(prob, device, start_states=start_states)
    for j in range(ncorner):        # FIVE_add_rule per corner, exactly as FIVEInit does for the initial rules
        ra = torch.from_numpy(np.ascontiguousarray(rant0[:, j])).to(device).expand(E, nant).contiguous()
        prob.add_rule(ra, torch.zeros((E,), dtype=torch.float64, device=device), rant_store=envs.rant)
    return prob, agent, envs
