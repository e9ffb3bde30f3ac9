"""ctypes binding of oracle/liboracle.so -- the CPU checker.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py; never by the product package.  See oracle/frirl_oracle.h for what each function
restates (reference file:line).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
MAX_NANT = 16
MAX_ACTIONS = 64

c_double_p = C.POINTER(C.c_double)
c_u32_p = C.POINTER(C.c_uint32)
c_i32_p = C.POINTER(C.c_int32)


class OrcFive(C.Structure):
    _fields_ = [("nant", C.c_int), ("U", C.c_int), ("p", C.c_int), ("R", C.c_int), ("maxR", C.c_int),
                ("u", c_double_p), ("ve", c_double_p), ("udivs", C.c_double * MAX_NANT),
                ("rant", c_double_p), ("veval", c_double_p), ("uidx", c_u32_p), ("rconc", c_double_p),
                ("ruledists", c_double_p), ("weights", c_double_p), ("wi", c_double_p)]


class OrcDim(C.Structure):
    _fields_ = [("values_len", C.c_int), ("values", C.c_double * MAX_ACTIONS), ("values_div", C.c_double),
                ("values_steep", C.c_double), ("values_def", C.c_double), ("universe_div", C.c_double)]


class OrcAgent(C.Structure):
    _fields_ = [("alpha", C.c_double), ("gamma", C.c_double), ("qdiff_pos_boundary", C.c_double), ("qdiff_neg_boundary", C.c_double),
                ("weight_significant", C.c_double), ("skip_rules", C.c_int), ("grid", c_double_p * MAX_NANT), ("grid_len", C.c_int * MAX_NANT)]


def build(force=False):
    src = [os.path.join(HERE, "frirl_oracle.c"), os.path.join(HERE, "frirl_oracle.h")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        subprocess.run(["make", "-C", HERE, "oracle"], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    vp, i, d, u64 = C.c_void_p, C.c_int, C.c_double, C.c_uint64
    F = C.POINTER(OrcFive)
    sig = {
        "orc_gen_fixres_arr": (None, [c_double_p, i, d]),
        "orc_gsc_func": (i, [c_double_p, i, i, c_double_p, i, i, c_double_p]),
        "orc_gvagenv": (None, [c_double_p, i, i, c_double_p, c_double_p]),
        "orc_snap": (C.c_uint, [c_double_p, i, d, d]),
        "orc_fast_pow": (d, [d, i]),
        "orc_five_create": (F, [c_double_p, c_double_p, i, i, i, i, i, c_double_p, c_double_p]),
        "orc_five_destroy": (None, [F]),
        "orc_add_rule": (i, [F, c_double_p, d]),
        "orc_remove_rule": (i, [F, C.c_uint]),
        "orc_rule_distance": (i, [F, c_double_p]),
        "orc_vag_concl": (C.c_uint, [F, c_double_p, c_double_p]),
        "orc_vag_concl_weight": (C.c_uint, [F, c_double_p, c_double_p]),
        "orc_bestact": (d, [F, c_double_p]),
        "orc_five_best_action": (C.c_uint, [F, c_double_p, c_double_p, i, c_double_p]),
        "orc_five_update_sarsa": (None, [F, C.POINTER(OrcAgent), c_double_p, c_double_p, d, c_double_p]),
        "orc_frirl_agent": (None, [vp, C.POINTER(OrcAgent)]),
        "orc_merge_rb": (None, [F, C.POINTER(OrcAgent), c_double_p, c_double_p, i]),
        "orc_gen_def_states": (i, [F, i, i, i, c_double_p]),
        "orc_frirl_new": (vp, [i, i, i]),
        "orc_frirl_delete": (None, [vp]),
        "orc_frirl_frb": (F, [vp]),
        "orc_frirl_actconc": (c_double_p, [vp]),
        "orc_frirl_action_vevalues": (c_double_p, [vp]),
        "orc_frirl_dim": (C.POINTER(OrcDim), [vp, i]),
        "orc_frirl_nstates": (i, [vp]),
        "orc_frirl_nactions": (i, [vp]),
        "orc_frirl_get_fus": (d, [vp]),
        "orc_frirl_set_fus": (None, [vp, d]),
        "orc_frirl_set_max_episodes": (None, [vp, i]),
        "orc_frirl_set_values_def": (None, [vp, i, C.c_double]),
        "orc_frirl_set_max_steps": (None, [vp, i]),
        "orc_frirl_hash": (u64, [vp]),
        "orc_frirl_total_steps": (C.c_long, [vp]),
        "orc_frirl_episode_num": (C.c_uint, [vp]),
        "orc_frirl_ep_steps": (i, [vp]),
        "orc_frirl_ep_reward": (d, [vp]),
        "orc_frirl_hparams": (None, [vp, c_double_p]),
        "orc_frirl_set_trace": (None, [vp, vp]),
        "orc_get_best_action": (C.c_uint, [vp, c_double_p]),
        "orc_check_possible_states": (d, [d, c_double_p, i]),
        "orc_update_sarsa": (None, [vp, c_double_p, d, c_double_p]),
        "orc_episode": (None, [vp]),
        "orc_sequential_run": (i, [vp, i]),
        "orc_save_rb_text": (i, [vp, C.c_char_p]),
        "orc_reduce_run": (i, [vp, i, d]),
        "orc_episode_eval": (None, [vp]),
        "orc_env_do_action": (None, [vp, d, c_double_p, c_double_p]),
        "orc_env_get_reward": (None, [vp, c_double_p, c_double_p, C.POINTER(C.c_int)]),
        "orc_env_quantize": (None, [vp, c_double_p, c_double_p]),
        "orc_sin": (d, [d]),
        "orc_cos": (d, [d]),
        "orc_hash_bytes": (u64, [u64, vp, u64]),
        "orc_splitmix64": (u64, [C.POINTER(u64)]),
        "orc_rand_unit": (d, [C.POINTER(u64)]),
        "orc_synth_tables": (None, [i, i, u64, c_double_p, c_double_p]),
        "orc_synth_rules": (None, [i, i, i, i, u64, c_u32_p, c_double_p]),
        "orc_batch_rule_distance": (None, [i, i, i, i, c_double_p, c_double_p, c_double_p, c_i32_p, c_double_p,
                                           c_double_p, c_i32_p, i]),
        "orc_demo_run": (i, [i, i, C.c_char_p, C.POINTER(u64), C.POINTER(C.c_long), C.POINTER(i), C.POINTER(i)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def dp(a):
    """double* view of a C-contiguous float64 array (or None)."""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_double_p)


def ip(a):
    if a is None:
        return None
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_i32_p)


def up(a):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_u32_p)


def hash_doubles(a, h=0):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return lib().orc_hash_bytes(h, a.ctypes.data, a.nbytes)


ENV_IDS = {"mountaincar": 0, "cartpole": 1, "acrobot": 2}


class Five:
    """Owning wrapper around orc_five."""

    def __init__(self, u, ve, nant, U, maxR, rant=None, rconc=None, p=0):
        u = np.ascontiguousarray(u, dtype=np.float64)
        ve = np.ascontiguousarray(ve, dtype=np.float64)
        R = 0 if rconc is None else len(rconc)
        if R:
            rant = np.ascontiguousarray(rant, dtype=np.float64)
            rconc = np.ascontiguousarray(rconc, dtype=np.float64)
        self.h = lib().orc_five_create(dp(u), dp(ve), p, nant, U, R, maxR, dp(rant) if R else None, dp(rconc) if R else None)
        assert self.h, "orc_five_create failed"
        self.own = True

    @classmethod
    def borrowed(cls, handle):
        o = cls.__new__(cls)
        o.h, o.own = handle, False
        return o

    def __del__(self):
        if getattr(self, "own", False) and self.h:
            lib().orc_five_destroy(self.h)
            self.h = None

    # geometry
    @property
    def c(self):
        return self.h.contents

    @property
    def R(self):
        return self.c.R

    @property
    def nant(self):
        return self.c.nant

    @property
    def U(self):
        return self.c.U

    @property
    def maxR(self):
        return self.c.maxR

    def remove_rule(self, r):
        return lib().orc_remove_rule(self.h, int(r))

    def arr(self, name, n):
        return np.ctypeslib.as_array(getattr(self.c, name), shape=(n,))

    @property
    def u(self):
        return self.arr("u", self.nant * self.U).reshape(self.nant, self.U)

    @property
    def ve(self):
        return self.arr("ve", self.nant * self.U).reshape(self.nant, self.U)

    @property
    def rconc(self):
        return self.arr("rconc", self.maxR)

    @property
    def rant(self):
        return self.arr("rant", self.maxR * self.nant).reshape(self.maxR, self.nant)

    @property
    def veval(self):
        return self.arr("veval", self.maxR * self.nant).reshape(self.nant, self.maxR)

    @property
    def uidx(self):
        return np.ctypeslib.as_array(self.c.uidx, shape=(self.nant * self.maxR,)).reshape(self.nant, self.maxR)

    @property
    def ruledists(self):
        return self.arr("ruledists", self.maxR)

    @property
    def weights(self):
        return self.arr("weights", self.maxR)

    def add_rule(self, rant, rconc):
        rant = np.ascontiguousarray(rant, dtype=np.float64)
        return lib().orc_add_rule(self.h, dp(rant), float(rconc))

    def rule_distance(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        return lib().orc_rule_distance(self.h, dp(x))

    def vag_concl(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = C.c_double()
        h = lib().orc_vag_concl(self.h, dp(x), C.byref(out))
        return (-1 if h == 0xFFFFFFFF else int(h)), out.value

    def vag_concl_weight(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        h = lib().orc_vag_concl_weight(self.h, dp(x), self.c.weights)
        return -1 if h == 0xFFFFFFFF else int(h)

    def best_action(self, states, action_ve):
        st = np.ascontiguousarray(states, dtype=np.float64)
        av = np.ascontiguousarray(action_ve, dtype=np.float64)
        out = np.zeros(len(av))
        b = lib().orc_five_best_action(self.h, dp(st), dp(av), len(av), dp(out))
        return int(b), out

    def update_sarsa(self, agent, fus, q_ant, reward, cur_q_ant):
        """agent: Agent; fus: float flag in; returns the flag after the update."""
        a = np.ascontiguousarray(q_ant, dtype=np.float64)
        b = np.ascontiguousarray(cur_q_ant, dtype=np.float64)
        f = C.c_double(fus)
        lib().orc_five_update_sarsa(self.h, C.byref(agent.c), C.byref(f), dp(a), float(reward), dp(b))
        return f.value

    def merge_rb(self, agent, rant, rconc):
        """merge_rb (reference frirl_agent.c:58-117): this rule base takes over the sender's rules rant [S][nant], rconc [S]."""
        ra = np.ascontiguousarray(rant, dtype=np.float64)
        rc = np.ascontiguousarray(rconc, dtype=np.float64)
        lib().orc_merge_rb(self.h, C.byref(agent.c), dp(ra), dp(rc), len(rc))

    def gen_def_states(self, agent_id, worldsize, nstates):
        """gen_def_states (frirl_agent.c:121-139) with THIS rule base as the master's: start state of agent `agent_id`, or None."""
        out = np.zeros(nstates)
        return out if lib().orc_gen_def_states(self.h, agent_id, worldsize, nstates, dp(out)) else None

    def device_layout(self, maxR=None):
        """rb[nant+1][maxR] float64: antecedent VE values per dimension, then consequents."""
        maxR = maxR or self.maxR
        rb = np.zeros((self.nant + 1, maxR), dtype=np.float64)
        R = self.R
        rb[: self.nant, :R] = self.veval[:, :R]
        rb[self.nant, :R] = self.rconc[:R]
        return rb


class Agent:
    """orc_agent: SARSA hyper-parameters + the grid of possible rule places per antecedent."""

    def __init__(self, alpha, gamma, qdiff_pos, qdiff_neg, weight_thr, skip_rules, grids):
        self.grids = [np.ascontiguousarray(g, dtype=np.float64) for g in grids]     # keep alive
        self.c = OrcAgent()
        self.c.alpha, self.c.gamma = alpha, gamma
        self.c.qdiff_pos_boundary, self.c.qdiff_neg_boundary = qdiff_pos, qdiff_neg
        self.c.weight_significant, self.c.skip_rules = weight_thr, skip_rules
        for k, g in enumerate(self.grids):
            self.c.grid[k] = dp(g)
            self.c.grid_len[k] = len(g)
        self.alpha, self.gamma, self.qdiff_pos, self.qdiff_neg = alpha, gamma, qdiff_pos, qdiff_neg
        self.weight_thr, self.skip_rules = weight_thr, skip_rules


class Frirl:
    """Owning wrapper around orc_frirl (one agent + its environment)."""

    def __init__(self, env, trig_mode=0, maxR=0):
        self.env = ENV_IDS[env] if isinstance(env, str) else int(env)
        self.h = lib().orc_frirl_new(self.env, trig_mode, maxR)
        assert self.h
        self.five = Five.borrowed(lib().orc_frirl_frb(self.h))
        self.nstates = lib().orc_frirl_nstates(self.h)
        self.nactions = lib().orc_frirl_nactions(self.h)
        self.nant = self.nstates + 1

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_frirl_delete(self.h)
            self.h = None

    def dim(self, k):
        d = lib().orc_frirl_dim(self.h, k).contents
        return dict(values=np.array(d.values[: d.values_len]), values_div=d.values_div, values_steep=d.values_steep,
                    values_def=d.values_def, universe_div=d.universe_div)

    @property
    def actconc(self):
        return np.ctypeslib.as_array(lib().orc_frirl_actconc(self.h), shape=(self.nactions,))

    @property
    def action_vevalues(self):
        return np.ctypeslib.as_array(lib().orc_frirl_action_vevalues(self.h), shape=(self.nactions,))

    @property
    def hparams(self):
        out = np.zeros(8)
        lib().orc_frirl_hparams(self.h, dp(out))
        return dict(alpha=out[0], gamma=out[1], qdiff_pos=out[2], qdiff_neg=out[3], weight_thr=out[4],
                    skip_rules=int(out[5]), reward_good_above=out[6], qdiff_final_tolerance=out[7])

    def agent(self):
        hp = self.hparams
        return Agent(hp["alpha"], hp["gamma"], hp["qdiff_pos"], hp["qdiff_neg"], hp["weight_thr"], hp["skip_rules"],
                     [self.dim(k)["values"] for k in range(self.nant)])

    @property
    def fus(self):
        return lib().orc_frirl_get_fus(self.h)

    @fus.setter
    def fus(self, v):
        lib().orc_frirl_set_fus(self.h, float(v))

    def get_best_action(self, states):
        s = np.ascontiguousarray(states, dtype=np.float64)
        return int(lib().orc_get_best_action(self.h, dp(s)))

    def update_sarsa(self, q_ant, reward, cur_q_ant):
        a = np.ascontiguousarray(q_ant, dtype=np.float64)
        b = np.ascontiguousarray(cur_q_ant, dtype=np.float64)
        lib().orc_update_sarsa(self.h, dp(a), float(reward), dp(b))

    def episode(self):
        lib().orc_episode(self.h)

    def run(self, max_episodes=None):
        if max_episodes is not None:
            lib().orc_frirl_set_max_episodes(self.h, max_episodes)
        return lib().orc_sequential_run(self.h, 0)

    def set_start_state(self, states):
        for k, v in enumerate(states):
            lib().orc_frirl_set_values_def(self.h, k, float(v))

    def episode_eval(self):
        lib().orc_episode_eval(self.h)

    def reduce(self, strategy=1, reward_tolerance=0.0):
        return lib().orc_reduce_run(self.h, strategy, reward_tolerance)

    def env_step(self, action, states):
        s = np.ascontiguousarray(states, dtype=np.float64)
        ns = np.zeros(self.nstates)
        q = np.zeros(self.nstates)
        r = C.c_double()
        f = C.c_int()
        lib().orc_env_do_action(self.h, float(action), dp(s), dp(ns))
        lib().orc_env_get_reward(self.h, dp(ns), C.byref(r), C.byref(f))
        lib().orc_env_quantize(self.h, dp(ns), dp(q))
        return ns, r.value, f.value, q

    @property
    def step_hash(self):
        return lib().orc_frirl_hash(self.h)

    @property
    def total_steps(self):
        return lib().orc_frirl_total_steps(self.h)

    @property
    def ep_steps(self):
        return lib().orc_frirl_ep_steps(self.h)

    @property
    def ep_reward(self):
        return lib().orc_frirl_ep_reward(self.h)

    def save_text(self, path):
        return lib().orc_save_rb_text(self.h, path.encode())


def synth_problem(nant, U, R, A, seed, maxR=None):
    """Synthetic tables + duplicate-free on-grid rule base.  Returns a Five."""
    u = np.zeros(nant * U)
    ve = np.zeros(nant * U)
    lib().orc_synth_tables(nant, U, seed, dp(u), dp(ve))
    uidx = np.zeros(nant * R, dtype=np.uint32)
    rc = np.zeros(R)
    lib().orc_synth_rules(nant, U, R, A, seed, up(uidx), dp(rc))
    uidx = uidx.reshape(nant, R)
    rant = np.ascontiguousarray(u.reshape(nant, U)[np.arange(nant)[:, None], uidx].T)
    return Five(u, ve, nant, U, maxR or (R + 8), rant, rc)


def synth_query(five, rng_state, q):
    """Same query stream as oracle/ref_harness.c synth_query()."""
    L = lib()
    n, U = five.nant, five.U
    st = C.c_uint64(rng_state)
    x = np.zeros(n)
    if q % 8 == 7:
        r = L.orc_splitmix64(C.byref(st)) % five.R
        x[:] = five.rant[r]
    else:
        u = five.u
        for k in range(n):
            lo, hi = u[k, 0], u[k, U - 2]
            x[k] = lo + (hi - lo) * L.orc_rand_unit(C.byref(st))
    return x, st.value
