"""The ANSI-C drop-in library (libfrirl_dropin.so: the reference's five_* / FIVE_* / frirl_* API on top of
the HIP C ABI).

CPU part: the library builds, exports the reference's full symbol surface (SURVEY 8b `nm -g` list), and --
when the reference tree is present (build container) -- the reference's UNCHANGED example sources compile
against include/*.h and link against it ("examples/ link unchanged", BASELINE north star).
GPU part: tests/test_ALL.sh-equivalent parity: the three demos run through the drop-in API on the MI355X
and their final rule bases are compared with the oracle-validated dumps of the genuine reference
(tests/golden/ref_*.frirlrb.txt) and with the reference's own golden files (tests/golden/orig/).
"""
import os
import subprocess

import numpy as np
import pytest

import frirl_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "fri-reinforcementlearning-c_amd", "lib")
REF = "/root/reference"

FIVE_SYMS = """FIVEInit FIVEGScFunc FIVE_GSc_func FIVEGVagEnv FIVEValVag FIVEVagConcl FIVE_vag_concl FIVEVagConclWeight
FIVE_vag_concl_weight FIVEVagConcl_FRIRL_BestAct FIVEAddRule five_add_rule FIVE_add_rule five_remove_rule five_rule_distance
five_vague_distance five_vague_distance_parallel five_deinit""".split()
FRIRL_SYMS = """frirl_init frirl_deinit frirl_init_ve frirl_init_rb frirl_episode frirl_e_greedy_selection frirl_get_best_action
frirl_check_possible_states frirl_update_sarsa frirl_sequential_run frirl_omp_run frirl_mpi_run frirl_test_run frirl_run
frirl_gen_fixres_arr frirl_visualization_init frirl_visualization_deinit frirl_show_rb frirl_show_hex_rb frirl_save_rb_to_text_file
frirl_save_rb_to_bin_file frirl_load_rb_from_bin_file frirl_print_usage frirl_parse_cmdline getch getActionFromTerminal""".split()


@pytest.fixture(scope="module")
def built():
    frirl_amd.build()
    return os.path.join(LIBDIR, "libfrirl_dropin.so"), os.path.join(LIBDIR, "frirl_demo")


def test_dropin_exports_reference_surface(built):
    lib, demo = built
    out = subprocess.run(["nm", "-D", "--defined-only", lib], check=True, capture_output=True, text=True).stdout
    have = {l.split()[-1] for l in out.splitlines() if l.strip()}
    missing = [s for s in FIVE_SYMS + FRIRL_SYMS if s not in have]
    assert not missing, missing
    assert os.access(demo, os.X_OK)


def test_struct_layouts_match_reference_abi(tmp_path):
    """offsetof/sizeof of the ABI structs (measured against the reference headers)."""
    src = tmp_path / "lay.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "frirl_types_def.h"\n#include "frirl_app_helpers.h"\n'
                   'int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(struct FIVERB), offsetof(struct FIVERB,ruledists),'
                   'offsetof(struct FIVERB,rseqant_veval), offsetof(struct FIVERB,rconc), offsetof(struct FIVERB,udivs), offsetof(struct FIVERB,avx2_rbsize),'
                   'sizeof(struct frirl_dimension_desc), sizeof(struct frirl_values_desc), sizeof(struct frirl_reward_desc), sizeof(struct frirl_desc),'
                   'offsetof(struct frirl_desc,fiverb), offsetof(struct frirl_desc,fus_is_rule_inserted));return 0;}\n')
    exe = tmp_path / "lay"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert [int(x) for x in got] == [280, 96, 152, 176, 208, 276, 64, 32, 24, 472, 264, 400]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "examples")), reason="reference tree only exists in the build container")
@pytest.mark.parametrize("env", ["mountaincar", "cartpole", "acrobot"])
def test_reference_examples_compile_and_link_unchanged(env, built, tmp_path):
    exe = tmp_path / env
    cmd = ["gcc", "-O2", "-w", "-I", os.path.join(ROOT, "include"), os.path.join(REF, "examples", env, env + ".c"), "-o", str(exe),
           "-L", LIBDIR, "-lfrirl_dropin", "-lfrirl_hip", "-Wl,-rpath," + LIBDIR, "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    und = subprocess.run(["nm", "-u", str(exe)], check=True, capture_output=True, text=True).stdout
    assert "frirl_run" in und and "frirl_init" in und


def load_rb(path):
    return np.loadtxt(path, dtype=np.float64, ndmin=2)


@pytest.mark.gpu
@pytest.mark.parametrize("env,steps_rules", [("mountaincar", 110), ("cartpole", 182), ("acrobot", 367)])
def test_demo_parity_through_dropin_api(env, steps_rules, built, tmp_path, golden_dir):
    """test_ALL.sh equivalent (reference tests/_test.sh:1-17 diffs the rule-base text dump), non-interactive.
    Same rule count, same antecedents in the same order (bit-exact), Q within 1e-6 relative of the
    reference compiled in the build container (host callbacks use glibc trig exactly like the reference)."""
    lib, demo = built
    r = subprocess.run([demo, "--env", env, "-q"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "converged 1" in r.stdout
    # every episode's report line (reference frirl_sequential_run.c:77-80) against the genuine reference's trace
    import json
    import re
    eps = [json.loads(l) for l in open(os.path.join(golden_dir, f"ref_{env}.trace.jsonl")) if '"k":"ep"' in l]
    lines = re.findall(r"Episode: (\d+)\tSteps: (\d+)\tReward: (?:\x1b\[[0-9;]*m)?(-?[0-9.]+)(?:\x1b\[[0-9;]*m)?\tRules: (\d+)", r.stdout)
    assert len(lines) == len(eps), (len(lines), len(eps))
    for (n, steps, reward, rules), ref in zip(lines, eps):
        assert int(n) == ref["ep"] and int(steps) == ref["steps"] and int(rules) == ref["R"], (n, steps, rules, ref)
        assert abs(float(reward) - float.fromhex(ref["reward"])) <= 1e-6 * max(1.0, abs(float.fromhex(ref["reward"])))
    mine = load_rb(tmp_path / f"{env}.frirlrb.txt")
    ref = load_rb(os.path.join(golden_dir, f"ref_{env}.frirlrb.txt"))
    assert mine.shape == ref.shape and mine.shape[0] == steps_rules
    assert (mine[:, :-1] == ref[:, :-1]).all(), "antecedents / rule order differ from the reference"
    rel = np.abs(mine[:, -1] - ref[:, -1]) / np.maximum(np.abs(ref[:, -1]), 1e-9)
    assert rel.max() <= 1e-6, rel.max()
    orig = load_rb(os.path.join(golden_dir, "orig", f"frirl_example_{env}.frirlrb.txt"))
    assert (mine[:, :-1] == orig[:, :-1]).all(), "antecedents differ from the reference's shipped golden file"
    if env != "acrobot":        # acrobot's shipped Q column is libm-era dependent
        relo = np.abs(mine[:, -1] - orig[:, -1]) / np.maximum(np.abs(orig[:, -1]), 1e-9)
        assert relo.max() <= 1e-6
    # binary dump: int count + (nant + 1) doubles per rule (reference frirl_utils.c:175-203)
    raw = (tmp_path / f"{env}.frirlrb.bin").read_bytes()
    n = int(np.frombuffer(raw[:4], dtype=np.int32)[0])
    assert n == mine.shape[0] and len(raw) == 4 + n * mine.shape[1] * 8
    body = np.frombuffer(raw[4:], dtype=np.float64).reshape(n, mine.shape[1])
    assert np.allclose(body, mine, rtol=0, atol=1e-15 * np.abs(mine).max() + 1e-18) or (np.abs(body - mine) <= 5e-19 + 1e-15 * np.abs(mine)).all()


@pytest.mark.parametrize("env", ["mountaincar", "cartpole", "acrobot"])
def test_demo_tables_bit_exact_vs_oracle(env, built):
    """Universes, VE tables, per-action VE values, grids and hyper-parameters produced by the drop-in library's
    host functions (frirl_init_ve, FIVE_GSc_func, FIVEGVagEnv, frirl_gen_fixres_arr) == the oracle's, bit for bit
    (the oracle's are pinned against the genuine reference in test_oracle_golden.py).  Host-only, no GPU."""
    from oracle import binding as ob
    d = frirl_amd.demo_describe(env)
    fr = ob.Frirl(env)
    f = fr.five
    assert (d["nant"], d["U"], d["A"]) == (f.nant, f.U, fr.nactions)
    assert (d["u"].view(np.uint64) == np.array(f.u).view(np.uint64)).all()
    assert (d["ve"].view(np.uint64) == np.array(f.ve).view(np.uint64)).all()
    assert (d["action_ve"].view(np.uint64) == np.array(fr.action_vevalues).view(np.uint64)).all()
    hp = fr.hparams
    for k in range(f.nant):
        od = fr.dim(k)
        assert (d["grids"][k] == od["values"]).all() and d["grid_div"][k] == od["values_div"] and d["values_def"][k] == od["values_def"]
    assert (d["alpha"], d["gamma"], d["qdiff_pos"], d["qdiff_neg"], d["weight_thr"], d["skip_rules"]) == \
           (hp["alpha"], hp["gamma"], hp["qdiff_pos"], hp["qdiff_neg"], hp["weight_thr"], hp["skip_rules"])


@pytest.mark.gpu
@pytest.mark.parametrize("env,strategy", [("mountaincar", 1), ("mountaincar", 2), ("cartpole", 1), ("acrobot", 1)])
def test_reduction_parity_through_dropin_api(env, strategy, built, tmp_path, golden_dir):
    """construct, then the reduction phase (what the reference's mountaincar example does as shipped)
    through the drop-in API on the MI355X: same surviving rules in the same order as the reference."""
    lib, demo = built
    r = subprocess.run([demo, "--env", env, "--reduce", str(strategy), "-q"], cwd=tmp_path, capture_output=True, text=True, timeout=1100)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    mine = load_rb(tmp_path / f"{env}.reduced{strategy}.frirlrb.txt")
    ref = load_rb(os.path.join(golden_dir, f"ref_{env}.reduced{strategy}.frirlrb.txt"))
    assert mine.shape == ref.shape, (mine.shape, ref.shape)
    assert (mine[:, :-1] == ref[:, :-1]).all()
    rel = np.abs(mine[:, -1] - ref[:, -1]) / np.maximum(np.abs(ref[:, -1]), 1e-9)
    assert rel.max() <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("env,rules", [("mountaincar", 110), ("acrobot", 367), ("cartpole", 182)])
def test_c_level_batched_agents(env, rules, built, tmp_path):
    """`frirl_demo --agents N`: N independent agents learn on the GPU through the C-level batch object
    (frirl_hip_batch_*), no Python, no torch.  Every agent converges; agent 0's rule base equals the oracle's
    (portable trig): antecedents bit-exact, consequents within 1e-6."""
    from oracle import binding as ob
    lib, demo = built
    r = subprocess.run([demo, "--env", env, "--agents", "96"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "converged 96" in r.stdout, r.stdout
    mine = load_rb(tmp_path / f"{env}.batch.frirlrb.txt")
    fr = ob.Frirl(env, trig_mode=1)
    assert fr.run() == 1
    f = fr.five
    assert mine.shape == (rules, f.nant + 1) and f.R == rules
    assert (mine[:, :-1] == np.array(f.rant[:rules])).all()
    rel = np.abs(mine[:, -1] - f.rconc[:rules]) / np.maximum(np.abs(f.rconc[:rules]), 1e-9)
    assert rel.max() <= 1e-6
    steps = {"mountaincar": 15548, "acrobot": 21207, "cartpole": 33002}[env]
    assert f"env-steps {96 * steps}" in r.stdout, r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("env,strategy", [("mountaincar", 1), ("mountaincar", 2), ("acrobot", 1)])
def test_c_level_batched_reduction(env, strategy, built, tmp_path):
    """`frirl_demo --agents N --reduce S`: learn on the GPU, then reduce agent 0's rule base with the speculative batched
    try-remove (frirl_hip_batch_reduce -> frirl_hip_reduce_shared), all from C.  The surviving rules are the ones the
    oracle's sequential reduction keeps (portable trig), in the same order."""
    from oracle import binding as ob
    lib, demo = built
    r = subprocess.run([demo, "--env", env, "--agents", "8", "--reduce", str(strategy)], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    mine = load_rb(tmp_path / f"{env}.batch.reduced{strategy}.frirlrb.txt")
    fr = ob.Frirl(env, trig_mode=1)
    assert fr.run() == 1
    R0 = fr.five.R
    fr.reduce(strategy, 0.0)
    f = fr.five
    assert f"agent 0 reduced {R0} -> {f.R} rules" in r.stdout, r.stdout[-600:]
    assert mine.shape == (f.R, f.nant + 1)
    assert (mine[:, :-1] == np.array(f.rant[: f.R])).all()
    rel = np.abs(mine[:, -1] - f.rconc[: f.R]) / np.maximum(np.abs(f.rconc[: f.R]), 1e-9)
    assert rel.max() <= 1e-6


def write_bin(path, records):
    """.frirlrb.bin records as frirl_save_rb_to_bin_file writes them: int32 R, then R x (nant antecedents + consequent) doubles."""
    with open(path, "wb") as f:
        for rant, rconc in records:
            f.write(np.int32(len(rconc)).tobytes())
            f.write(np.ascontiguousarray(np.concatenate([rant, rconc[:, None]], axis=1), dtype=np.float64).tobytes())


def read_bin(path, nant):
    raw = open(path, "rb").read()
    out, off = [], 0
    while off < len(raw):
        R = int(np.frombuffer(raw, dtype=np.int32, count=1, offset=off)[0])
        off += 4
        a = np.frombuffer(raw, dtype=np.float64, count=R * (nant + 1), offset=off).reshape(R, nant + 1)
        off += 8 * R * (nant + 1)
        out.append(a)
    return out


@pytest.mark.gpu
def test_batched_rulebase_files(built, tmp_path):
    """rule bases of all agents to / from one file in the reference's .frirlrb.bin record format.
    (1) `--save` after training: every agent's record equals the oracle's rule base; (2) a one-record file (what the
    reference writes) loads into every agent, the greedy replay reproduces the oracle's frirl_test_run episode and the
    reduction its reduced rule base; (3) the multi-record file loads back; (4) the loader is bounds-checked."""
    from oracle import binding as ob
    lib, demo = built
    env, E = "mountaincar", 5
    fr = ob.Frirl(env, trig_mode=1)
    assert fr.run() == 1
    f = fr.five
    R, nant = f.R, f.nant
    rant, rconc = np.array(f.rant[:R]), np.array(f.rconc[:R])
    # (1) save
    r = subprocess.run([demo, "--env", env, "--agents", str(E), "--save", "all.bin"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    recs = read_bin(tmp_path / "all.bin", nant)
    assert len(recs) == E
    for a in recs:
        assert a.shape == (R, nant + 1) and (a[:, :nant] == rant).all()
        assert (np.abs(a[:, nant] - rconc) <= 1e-9 * np.maximum(np.abs(rconc), 1.0)).all()
    # (2) one record -> every agent; replay + reduction
    write_bin(tmp_path / "one.bin", [(rant, rconc)])
    fr.episode_eval()
    steps, reward = fr.ep_steps, fr.ep_reward
    r = subprocess.run([demo, "--env", env, "--agents", str(E), "--load", "one.bin", "--reduce", "1"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "loaded 1 rule base record(s)" in r.stdout
    assert f"env-steps {E * steps} " in r.stdout and f"mean-rules {R:.3f}" in r.stdout and f"mean-reward {reward:.6f}" in r.stdout, r.stdout
    fr.reduce(1, 0.0)
    mine = load_rb(tmp_path / f"{env}.batch.reduced1.frirlrb.txt")
    assert mine.shape == (f.R, nant + 1) and (mine[:, :-1] == np.array(f.rant[: f.R])).all() and (mine[:, -1] == np.array(f.rconc[: f.R])).all()
    # (3) the batch file loads back (E records)
    r = subprocess.run([demo, "--env", env, "--agents", str(E), "--load", "all.bin", "--save", "again.bin"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and f"loaded {E} rule base record(s)" in r.stdout, r.stdout[-800:] + r.stderr[-800:]
    assert open(tmp_path / "again.bin", "rb").read() == open(tmp_path / "all.bin", "rb").read()
    # (4) bounds checks: truncated record, rule count beyond the capacity, NaN
    raw = open(tmp_path / "one.bin", "rb").read()
    open(tmp_path / "trunc.bin", "wb").write(raw[:-12])
    open(tmp_path / "huge.bin", "wb").write(np.int32(100000).tobytes() + raw[4:])
    bad = rconc.copy()
    bad[3] = np.nan
    write_bin(tmp_path / "nan.bin", [(rant, bad)])
    for name, msg in (("trunc.bin", "truncated"), ("huge.bin", "capacity"), ("nan.bin", "non-finite")):
        r = subprocess.run([demo, "--env", env, "--agents", "2", "--load", name], cwd=tmp_path, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and msg in (r.stdout + r.stderr), (name, r.stdout[-500:], r.stderr[-500:])


@pytest.mark.gpu
def test_demo_parity_script(built):
    """tests/demo_parity.sh: the non-interactive counterpart of the reference's tests/test_ALL.sh (three demos through the
    drop-in C API against the golden rule bases of the reference compiled in the build container)."""
    r = subprocess.run(["bash", os.path.join(ROOT, "tests", "demo_parity.sh")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for env in ("mountaincar", "cartpole", "acrobot"):
        assert f"{env}: Valid" in r.stdout, r.stdout
