#!/bin/bash
# Non-interactive counterpart of the reference's tests/test_ALL.sh: runs the three demo applications through the drop-in
# C API on the GPU (frirl_demo, construct mode) and checks each final rule base against the golden dump of the
# reference compiled in the build container (tests/golden/ref_<env>.frirlrb.txt): same number of rules, antecedent
# columns identical, consequents within 1e-6 relative (the north-star contract).  Prints Valid / Invalid per demo and
# exits non-zero if any demo is invalid.   Usage: tests/demo_parity.sh [env ...]
set -u
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
DEMO=$ROOT/fri-reinforcementlearning-c_amd/lib/frirl_demo
[ -x "$DEMO" ] || { echo "frirl_demo is not built (python __graft_entry__.py)"; exit 2; }
ENVS=${*:-mountaincar cartpole acrobot}
WORK=$(mktemp -d)
trap 'rm -rf "$WORK"' EXIT
bad=0
for env in $ENVS; do
    SECONDS=0
    if ! (cd "$WORK" && "$DEMO" --env "$env" -q > "$env.log" 2>&1); then
        echo "$env: Invalid (frirl_demo failed, see below)"; tail -5 "$WORK/$env.log"; bad=1; continue
    fi
    secs=$SECONDS
    if python3 - "$WORK/$env.frirlrb.txt" "$HERE/golden/ref_$env.frirlrb.txt" <<'PY'
import sys
mine = [[float(v) for v in l.split()] for l in open(sys.argv[1]) if l.strip()]
gold = [[float(v) for v in l.split()] for l in open(sys.argv[2]) if l.strip()]
ok = len(mine) == len(gold) and all(len(a) == len(b) for a, b in zip(mine, gold))
worst = 0.0
if ok:
    for a, b in zip(mine, gold):
        ok = ok and a[:-1] == b[:-1]
        worst = max(worst, abs(a[-1] - b[-1]) / max(abs(b[-1]), 1e-9))
    ok = ok and worst <= 1e-6
print("rules %d (golden %d), max relative Q difference %.3g" % (len(mine), len(gold), worst))
sys.exit(0 if ok else 1)
PY
    then echo "$env: Valid (${secs}s)"; else echo "$env: Invalid"; bad=1; fi
done
exit $bad
