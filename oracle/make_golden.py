#!/usr/bin/env python3
"""Regenerates tests/golden/ from the GENUINE reference (build container only).

TEST INFRASTRUCTURE.  Needs /root/reference: `make -C oracle` compiles the reference's own
sources into oracle/_ref/ (nothing is copied), oracle/_ref/ref_harness then drives the
reference's exported C API and each example's own main().  What is committed is data only:

  tests/golden/orig/*.frirlrb.txt     mirror of the reference's own golden rule bases
                                      (reference tests/orig/, the only known-answer files it ships)
  tests/golden/ref_<env>.frirlrb.txt  final rule base of the reference compiled HERE (construct mode)
  tests/golden/ref_<env>.trace.jsonl  first 400 steps, one record per episode and a running
                                      FNV-1a hash over every step of the whole run
  tests/golden/ref_<env>.reduced<S>.frirlrb.txt  rule base after the reference's reduction phase (strategy S in 1, 2)
  tests/golden/vec_<env>.jsonl        function-level vectors (tables, index snap, rule distance,
                                      vag_concl, weights, best action, SARSA updates, env steps)
                                      on the rule base reached after a few episodes
  tests/golden/omprun_<env>_<N>.jsonl the genuine frirl_omp_run (5 agents, max_episodes N): master's final rule base + episode / "pended" counts
  tests/golden/merge_<env>.jsonl      multi-agent rule-base merge (frirl_agent.c merge_rb / gen_def_states / omp_init, compiled with
                                      BUILD_OPENMP by oracle/_ref/ref_merge_harness): master and agent rule bases before and
                                      after merging in both directions
  tests/golden/synth_*.jsonl          hashes for large synthetic rule bases (inputs are re-created
                                      in the tests by oracle orc_synth_*; nant <= 8 only, the
                                      reference's cap is FIVE_MAX_NUM_OF_UNIVERSES = 8)
"""
import json, os, shutil, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("FRIRL_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")
HARNESS = os.path.join(HERE, "_ref", "ref_harness")
ENVS = ["mountaincar", "cartpole", "acrobot"]
VEC_EPISODES = {"mountaincar": 6, "cartpole": 9, "acrobot": 5}
MERGE_EPISODES = {"mountaincar": (8, 3), "cartpole": (10, 4), "acrobot": (6, 3)}
OMPRUN = [("mountaincar", 1000), ("cartpole", 15), ("cartpole", 40), ("acrobot", 15), ("acrobot", 40)]      # (env, max_episodes), 5 agents
SYNTH = [  # nant, U, R, A, seed, nq
    (3, 41, 33, 3, 11, 64),
    (5, 41, 367, 3, 12, 64),
    (5, 1001, 4096, 21, 13, 48),
    (8, 101, 4096, 0, 14, 32),
    (5, 41, 65536, 3, 15, 16),
    (3, 41, 8192, 3, 16, 24),
]


def run(*cmd, **kw):
    print("+", " ".join(cmd))
    subprocess.run(cmd, check=True, **kw)


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not found: golden vectors can only be regenerated in the build container")
    run("make", "-C", HERE)
    os.makedirs(os.path.join(GOLD, "orig"), exist_ok=True)
    for e in ENVS:
        shutil.copyfile(os.path.join(REF, "tests", "orig", f"frirl_example_{e}.frirlrb.txt"),
                        os.path.join(GOLD, "orig", f"frirl_example_{e}.frirlrb.txt"))
    tmp = "/tmp/frirl_golden"
    os.makedirs(tmp, exist_ok=True)
    for e in ([] if "--merge-only" in sys.argv else ENVS):
        with open(os.path.join(tmp, f"{e}.stdout"), "w") as so:
            run(HARNESS, "demo", e, tmp, stdout=so)
        shutil.copyfile(os.path.join(tmp, f"{e}.frirlrb.txt"), os.path.join(GOLD, f"ref_{e}.frirlrb.txt"))
        shutil.copyfile(os.path.join(tmp, f"{e}.trace.jsonl"), os.path.join(GOLD, f"ref_{e}.trace.jsonl"))
        with open(os.path.join(tmp, f"{e}.vstdout"), "w") as so:
            run(HARNESS, "vectors", e, os.path.join(GOLD, f"vec_{e}.jsonl"), str(VEC_EPISODES[e]), stdout=so)
    for e in ([] if "--merge-only" in sys.argv else ENVS):
        for strategy in (1, 2):
            with open(os.path.join(tmp, f"{e}.rstdout"), "w") as so:
                run(HARNESS, "reduce", e, tmp, str(strategy), stdout=so)
            shutil.copyfile(os.path.join(tmp, f"{e}.reduced{strategy}.frirlrb.txt"), os.path.join(GOLD, f"ref_{e}.reduced{strategy}.frirlrb.txt"))
    for e in ENVS:      # (master episodes, agent episodes): early rule bases, so that the merge inserts, blends and spreads
        m_eps, a_eps = MERGE_EPISODES[e]
        with open(os.path.join(tmp, f"{e}.mstdout"), "w") as so:
            run(os.path.join(HERE, "_ref", "ref_merge_harness"), e, os.path.join(GOLD, f"merge_{e}.jsonl"), str(m_eps), str(a_eps), stdout=so)
    # the GENUINE many-agent loop with rule-base exchange (frirl_omp_run, unmodified) on 5 agents: the master's final rule base,
    # and from its own printouts the episodes run by all agents and the exchanges skipped because an agent had "pended"
    for e, max_eps in OMPRUN:
        out = os.path.join(GOLD, f"omprun_{e}_{max_eps}.jsonl")
        so_path = os.path.join(tmp, f"{e}.ostdout")
        with open(so_path, "w") as so:
            run(os.path.join(HERE, "_ref", "ref_merge_harness"), e, out, "omprun", "5", str(max_eps), stdout=so, stderr=subprocess.DEVNULL)
        txt = open(so_path).read()
        with open(out, "a") as f:
            f.write(json.dumps({"k": "stdout_counts", "episode_lines": txt.count("Episode:"), "pended_lines": txt.count("pended")}) + "\n")
    if "--merge-only" in sys.argv:
        return
    for (nant, U, R, A, seed, nq) in SYNTH:
        run(HARNESS, "synth", str(nant), str(U), str(R), str(A), str(seed), str(nq),
            os.path.join(GOLD, f"synth_n{nant}_u{U}_r{R}.jsonl"))
    sz = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(GOLD) for f in fs)
    print(f"golden fixtures: {sz / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
