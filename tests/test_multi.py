"""Several GPUs from plain C (frirl_hip_multi_*: one batch + host thread per device, environments sharded by global id, the
per-episode report all-reduced with RCCL).  CPU: the shard arithmetic (== the Python sharding used by bench.py / dist.py).
GPU (one device on the test box): `frirl_demo --agents N --gpus 1` goes through ncclCommInitAll / ncclAllReduce for real and
must reproduce the single-batch result."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import frirl_amd


def test_shard_arithmetic_matches_python_sharding():
    lib = frirl_amd.lib()
    D = frirl_amd.dist()
    for total in (0, 1, 7, 8, 65536, 65537, 1000003):
        for world in (1, 2, 3, 8):
            seen = 0
            for rank in range(world):
                s, c = C.c_int64(), C.c_int64()
                assert lib.frirl_hip_shard(total, world, rank, C.byref(s), C.byref(c)) == 0
                assert (s.value, c.value) == D.shard(total, world, rank)
                assert s.value == seen
                seen += c.value
            assert seen == total
    s, c = C.c_int64(), C.c_int64()
    assert lib.frirl_hip_shard(10, 2, 2, C.byref(s), C.byref(c)) == -2          # rank outside the world: EINVAL


def test_multi_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert frirl_amd.lib().frirl_hip_multi_create(None, 8, 1) is None


@pytest.mark.gpu
@pytest.mark.parametrize("env,rules,steps", [("mountaincar", 110, 15548), ("acrobot", 367, 21207)])
def test_c_level_multi_gpu_runner_on_one_device(env, rules, steps, tmp_path):
    from oracle import binding as ob
    frirl_amd.build()
    demo = os.path.join(frirl_amd.PKG_DIR, "lib", "frirl_demo")
    r = subprocess.run([demo, "--env", env, "--agents", "70", "--gpus", "1"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "gpus 1 (RCCL" in r.stdout and "converged 70" in r.stdout and "device 0 runs agents [0, 70)" in r.stdout, r.stdout
    assert f"env-steps {70 * steps}" in r.stdout, r.stdout
    mine = np.loadtxt(tmp_path / f"{env}.multi.frirlrb.txt", dtype=np.float64, ndmin=2)
    fr = ob.Frirl(env, trig_mode=1)
    assert fr.run() == 1
    f = fr.five
    assert mine.shape == (rules, f.nant + 1)
    assert (mine[:, :-1] == np.array(f.rant[:rules])).all()
    rel = np.abs(mine[:, -1] - f.rconc[:rules]) / np.maximum(np.abs(f.rconc[:rules]), 1e-9)
    assert rel.max() <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("env,agents,max_episodes", [("mountaincar", 6, 22), ("acrobot", 5, 12)])
def test_c_level_rule_base_exchange_across_devices_on_one_device(env, agents, max_episodes, tmp_path):
    """`frirl_demo --agents N --gpus 1 --merge`: the many-agent loop WITH the rule-base exchange through frirl_hip_multi_train_merged (master
    broadcast with ncclBroadcast, the staged master list as the sender, the report's "master complete" flag all-reduced) must reproduce
    `frirl_demo --agents N --merge` (frirl_hip_batch_train_merged, itself checked against the oracle loop in test_hip_merge.py) bit for
    bit: same episodes, same rounds, identical master rule base.  (More than one device cannot run on the test box; the peer path --
    ncclSend / ncclRecv of the shards' rule lists -- differs only in where the sender rows lie.)"""
    frirl_amd.build()
    demo = os.path.join(frirl_amd.PKG_DIR, "lib", "frirl_demo")
    a = subprocess.run([demo, "--env", env, "--agents", str(agents), "--merge", "--max-episodes", str(max_episodes)], cwd=tmp_path, capture_output=True, text=True,
                       timeout=900)
    b = subprocess.run([demo, "--env", env, "--agents", str(agents), "--gpus", "1", "--merge", "--max-episodes", str(max_episodes)], cwd=tmp_path,
                       capture_output=True, text=True, timeout=900)
    assert a.returncode == 0, a.stdout[-2000:] + a.stderr[-2000:]
    assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-2000:]
    assert "gpus 1 (RCCL" in b.stdout, b.stdout

    def fields(out):
        line = [ln for ln in out.splitlines() if ln.startswith("merged ")][-1]
        tok = line.split()
        return {k: tok[tok.index(k) + 1] for k in ("agents", "episodes", "merge-rounds", "converged", "env-steps", "mean-rules", "mean-reward")}

    fa, fb = fields(a.stdout), fields(b.stdout)
    assert fa == fb, (fa, fb)
    assert int(fa["merge-rounds"]) >= 1
    one = np.loadtxt(tmp_path / f"{env}.merged.frirlrb.txt", dtype=np.float64, ndmin=2)
    multi = np.loadtxt(tmp_path / f"{env}.multi.merged.frirlrb.txt", dtype=np.float64, ndmin=2)
    assert one.shape == multi.shape and (one.view(np.uint64) == multi.view(np.uint64)).all()


def _demo(args, cwd, loopback=False):
    env = dict(os.environ)
    if loopback:
        env["FRIRL_HIP_MULTI_LOOPBACK"] = "1"        # logical shards on the one device, device-to-device copies instead of RCCL
    demo = os.path.join(frirl_amd.PKG_DIR, "lib", "frirl_demo")
    r = subprocess.run([demo] + args, cwd=cwd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [2, 3])
def test_c_level_runner_with_logical_shards(shards, tmp_path):
    """frirl_hip_multi_train with G > 1: one host thread per shard, the report gathered from every shard and combined in shard order,
    every shard leaving the loop in the same episode -- executed with 2 and 3 LOGICAL shards on the one device (loop-back
    transport); same job report and same master rule base as one shard."""
    frirl_amd.build()
    one = _demo(["--env", "mountaincar", "--agents", "7", "--gpus", "1"], tmp_path)
    a = np.loadtxt(tmp_path / "mountaincar.multi.frirlrb.txt", dtype=np.float64, ndmin=2)
    many = _demo(["--env", "mountaincar", "--agents", "7", "--gpus", str(shards)], tmp_path, loopback=True)
    b = np.loadtxt(tmp_path / "mountaincar.multi.frirlrb.txt", dtype=np.float64, ndmin=2)
    assert f"gpus {shards} (RCCL -1)" in many and f"device {shards - 1} runs agents" in many, many
    tail = lambda out: [ln.split("agents", 1)[1] for ln in out.splitlines() if ln.startswith("multi mountaincar: gpus")][-1]
    assert tail(one) == tail(many), (one, many)
    assert a.shape == b.shape and (a.view(np.uint64) == b.view(np.uint64)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("env,agents,shards,max_episodes", [("mountaincar", 7, 3, 22), ("acrobot", 5, 2, 12), ("mountaincar", 6, 6, 13)])
def test_c_level_rule_base_exchange_with_logical_shards(env, agents, shards, max_episodes, tmp_path):
    """frirl_hip_multi_train_merged with G > 1 -- the g > 0 / p >= 1 branches: the master's list broadcast to the other shards, their
    rule lists packed (strided consequent copy), sent, received and merged into the master in GLOBAL agent order, "master complete"
    riding in the report -- through the loop-back transport with 2, 3 and 6 logical shards (6 = one agent per shard; 7 agents over 3
    shards = ragged shard sizes): episodes, merge rounds, job report and the master's rule base bit-identical to
    frirl_hip_batch_train_merged on the same agents (itself checked against the oracle's loop in test_hip_merge.py)."""
    frirl_amd.build()
    base = ["--env", env, "--agents", str(agents), "--merge", "--max-episodes", str(max_episodes)]
    a = _demo(base, tmp_path)
    b = _demo(base + ["--gpus", str(shards)], tmp_path, loopback=True)
    assert f"gpus {shards} (RCCL -1)" in b, b

    def fields(out):
        line = [ln for ln in out.splitlines() if ln.startswith("merged ")][-1]
        tok = line.split()
        return {k: tok[tok.index(k) + 1] for k in ("agents", "episodes", "merge-rounds", "converged", "env-steps", "mean-rules", "mean-reward")}

    fa, fb = fields(a), fields(b)
    assert fa == fb, (fa, fb)
    assert int(fa["merge-rounds"]) >= 1
    one = np.loadtxt(tmp_path / f"{env}.merged.frirlrb.txt", dtype=np.float64, ndmin=2)
    multi = np.loadtxt(tmp_path / f"{env}.multi.merged.frirlrb.txt", dtype=np.float64, ndmin=2)
    assert one.shape == multi.shape and (one.view(np.uint64) == multi.view(np.uint64)).all()
