"""N>1 path on CPU: world_size-2 gloo processes shard the environments and all-reduce the reward statistics
exactly as bench.py / a multi-GPU run does with RCCL (no data-path collective exists to test)."""
import os
import socket

import pytest

import frirl_amd


def test_shard_partitions_every_env_exactly_once():
    d = frirl_amd.dist()
    for total in (0, 1, 7, 8, 65536, 65537):
        for world in (1, 2, 3, 8):
            got = [d.shard(total, world, r) for r in range(world)]
            assert sum(c for _, c in got) == total
            pos = 0
            for s, c in got:
                assert s == pos
                pos += c
            counts = [c for _, c in got]
            assert max(counts) - min(counts) <= 1


def _worker(rank, world, port, q):
    import torch
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    d = frirl_amd.dist()
    w, r = d.init(backend="gloo")
    assert (w, r) == (world, rank)
    total = 11
    start, count = d.shard(total, world, rank)
    env_id = torch.arange(start, start + count, dtype=torch.float64)
    ep_reward = -10.0 * env_id - 1.0                     # depends on the GLOBAL env id: sharding must not change the totals
    ep_steps = (env_id + 1).to(torch.int32)
    success = (env_id % 2 == 0).to(torch.int32)
    nrules = (8 + env_id).to(torch.int32)
    st = d.allreduce_stats(ep_reward, ep_steps, success, nrules)
    tmax = d.max_over_ranks(1.0 + rank, torch.device("cpu"))
    q.put((rank, st, tmax))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_world2_gloo_stats_allreduce():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ids = list(range(11))
    for rank, st, tmax in res:
        assert st.envs == 11
        assert st.reward_sum == sum(-10.0 * i - 1.0 for i in ids)
        assert st.steps_sum == sum(i + 1 for i in ids)
        assert st.success == sum(1 for i in ids if i % 2 == 0)
        assert st.rules_sum == sum(8 + i for i in ids)
        assert st.reward_min == -101.0 and st.reward_max == -1.0
        assert tmax == 2.0
