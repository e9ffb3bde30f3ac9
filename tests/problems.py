"""Seeded synthetic FRIRL problems for the parity tests (inputs come from the oracle's generator,
SURVEY 8d: shared universes/VE tables, one private on-grid rule base per environment)."""
import ctypes as C

import numpy as np

from oracle import binding as ob


class Batch:
    """E rule bases in the device layout rb[E][nant+1][maxR] + the tables, as numpy arrays."""

    def __init__(self, nant, U, R, E, A=0, seed=1, maxR=None, ragged=True, tables=None):
        L = ob.lib()
        self.nant, self.U, self.E, self.A = nant, U, E, A
        self.maxR = maxR or (R + (R & 1))
        assert self.maxR % 2 == 0 and self.maxR >= R
        if tables is None:
            u = np.zeros(nant * U)
            ve = np.zeros(nant * U)
            L.orc_synth_tables(nant, U, seed, ob.dp(u), ob.dp(ve))
            self.u, self.ve = u.reshape(nant, U), ve.reshape(nant, U)
        else:
            self.u, self.ve = tables
        rng = np.random.default_rng(seed)
        self.nrules = np.full(E, R, dtype=np.int32)
        if ragged and E > 1:
            # ragged rule counts incl. odd sizes, tiny bases and (when possible) one empty base
            self.nrules = rng.integers(max(1, R // 3), R + 1, size=E).astype(np.int32)
            self.nrules[0] = R
            if E > 2:
                self.nrules[1] = max(1, min(R, 7))
            if E > 3:
                self.nrules[2] = 0
        self.rb = np.zeros((E, nant + 1, self.maxR), dtype=np.float64)
        self.uidx = np.zeros((E, nant, self.maxR), dtype=np.uint32)
        uidx = np.zeros(nant * R, dtype=np.uint32)
        rc = np.zeros(R)
        karange = np.arange(nant)[:, None]
        for e in range(E):
            L.orc_synth_rules(nant, U, R, A, seed * 1000003 + e, ob.up(uidx), ob.dp(rc))
            n = int(self.nrules[e])
            ui = uidx.reshape(nant, R)[:, :n]
            self.uidx[e, :, :n] = ui
            self.rb[e, :nant, :n] = self.ve[karange, ui]
            self.rb[e, nant, :n] = rc[:n]

    def queries(self, seed=5, hit_fraction=0.25, hit_last=True):
        """x[E][nant]: continuous observations (no hit) with a fraction of exact rule hits; environment 0
        hits its LAST rule (tail / odd-lane handling) when hit_last."""
        rng = np.random.default_rng(seed)
        lo, hi = self.u[:, 0], self.u[:, self.U - 2]
        x = lo + (hi - lo) * rng.random((self.E, self.nant))
        hits = rng.random(self.E) < hit_fraction
        for e in np.nonzero(hits)[0]:
            n = int(self.nrules[e])
            if n == 0:
                continue
            r = rng.integers(0, n)
            x[e] = self.u[np.arange(self.nant), self.uidx[e, :, r]]
        if hit_last and self.nrules[0] > 0:
            r = int(self.nrules[0]) - 1
            x[0] = self.u[np.arange(self.nant), self.uidx[0, :, r]]
        return np.ascontiguousarray(x)

    def oracle_rule_distance(self, x, nthreads=0):
        d = np.zeros((self.E, self.maxR))
        hit = np.zeros(self.E, dtype=np.int32)
        ob.lib().orc_batch_rule_distance(self.E, self.nant, self.U, self.maxR, ob.dp(np.ascontiguousarray(self.u.ravel())),
                                         ob.dp(np.ascontiguousarray(self.ve.ravel())), ob.dp(self.rb.reshape(-1)), ob.ip(self.nrules),
                                         ob.dp(x.reshape(-1)), ob.dp(d.reshape(-1)), ob.ip(hit), nthreads)
        return d, hit

    def five(self, e):
        """Oracle rule base of environment e (its own copy)."""
        n = int(self.nrules[e])
        rant = np.ascontiguousarray(self.u[np.arange(self.nant)[:, None], self.uidx[e, :, :n]].T)
        f = ob.Five(self.u.ravel(), self.ve.ravel(), self.nant, self.U, self.maxR, rant if n else None,
                    np.ascontiguousarray(self.rb[e, self.nant, :n]) if n else None)
        assert f.R == n
        return f

    def action_ve(self):
        """VE values of the A actions used by orc_synth_rules' action column."""
        A, U = self.A, self.U
        idx = [0] if A == 1 else [(a * (U - 1)) // (A - 1) for a in range(A)]
        return np.ascontiguousarray(self.ve[self.nant - 1, idx]), np.ascontiguousarray(self.u[self.nant - 1, idx])

    def to_device(self, device="cuda", compressed=False):
        import torch
        import frirl_amd
        uidx = torch.from_numpy(self.uidx.astype(np.int16)).to(device) if compressed else None
        return frirl_amd.Problem(torch.from_numpy(np.ascontiguousarray(self.u)).to(device), torch.from_numpy(np.ascontiguousarray(self.ve)).to(device),
                                 torch.from_numpy(self.rb).to(device), torch.from_numpy(self.nrules).to(device), uidx)


def demo_batch(env, episodes, E=1):
    """A rule base grown by the oracle on a real demo (its tables, grids and hyper-parameters)."""
    fr = ob.Frirl(env)
    fr.run(max_episodes=episodes + 1)
    return fr


def demo_device_batch(env, episodes, E, seed=0):
    """E copies of the rule base a demo has after `episodes` episodes (real tables / grids / hyper-parameters)
    as a Batch-like object + the oracle agent.  Queries differ per environment."""
    fr = demo_batch(env, episodes)
    f = fr.five
    b = Batch.__new__(Batch)
    b.nant, b.U, b.E, b.A = f.nant, f.U, E, fr.nactions
    R = f.R
    b.maxR = R + 64 + ((R + 64) & 1)
    b.u, b.ve = np.array(f.u), np.array(f.ve)
    b.nrules = np.full(E, R, dtype=np.int32)
    b.rb = np.zeros((E, f.nant + 1, b.maxR))
    b.rb[:, : f.nant, :R] = f.veval[:, :R]
    b.rb[:, f.nant, :R] = f.rconc[:R]
    b.uidx = np.zeros((E, f.nant, b.maxR), dtype=np.uint32)
    b.uidx[:, :, :R] = f.uidx[:, :R]
    b.rant = np.array(f.rant[:R])
    return b, fr


def device_agent(fr, device="cuda", max_steps=1000):
    """frirl_amd.Agent mirroring an oracle Frirl (its grids, hyper-parameters, per-action VE values)."""
    import frirl_amd
    hp = fr.hparams
    dims = [fr.dim(k) for k in range(fr.nant)]
    return frirl_amd.Agent(device, fr.nant, [d["values"] for d in dims], [d["values_div"] for d in dims],
                           [d["values_def"] for d in dims], np.array(fr.action_vevalues), hp["alpha"], hp["gamma"], hp["qdiff_pos"],
                           hp["qdiff_neg"], hp["weight_thr"], hp["skip_rules"], 0, fr.env, max_steps)
