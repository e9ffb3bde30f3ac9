"""bench.py's own rank launcher (no GPU needed): `python bench.py --gpus N` with no WORLD_SIZE in the environment must
start N ranks itself (torch.distributed.run as a CHILD process, started before anything touches the GPU), relay rank 0's
JSON line and the child's exit code.  --rank-probe replaces the GPU legs by a gloo all-reduce that counts the ranks."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=300)


def test_bench_launches_its_own_ranks():
    p = run(["--gpus", "2", "--rank-probe"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout           # ONE JSON line (rank 0's)
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] == 2.0 and rec["config"]["world_size"] == 2 and rec["config"]["dist_backend"] == "gloo"


def test_bench_refuses_a_rank_count_mismatch():
    """Under a launcher (WORLD_SIZE set) the rank count must equal --gpus: no silent single-GPU run labelled N."""
    p = run(["--gpus", "4", "--no-cpu-baseline"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert p.returncode != 0 and "--gpus 4 but WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_launcher_parent_does_not_import_torch():
    """The parent must not initialise the GPU: the launch decision is taken before torch is imported."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src.split("def main():")[1].split("import torch")[0]
    assert "launch_ranks(args.gpus" in head
    launcher = src.split("def launch_ranks")[1].split("def rank_probe")[0]
    assert "import torch" not in launcher and "os.exec" not in launcher
