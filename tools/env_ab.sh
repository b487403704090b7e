#!/bin/bash
# interleaved A/B of an environment knob that is read once per process: bench.py per size, with and without it.   usage: env_ab.sh VAR=VALUE [log_n ...]
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$GRAFT_REPO_ROOT"
KV="$1"; shift
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'logN', '$2', 'ms', d['value'], 'sort', d['stage_ms_untimed_diagnostic_step'].get('sort_ms'), 'reduce', d['stage_ms_untimed_diagnostic_step'].get('reduce_ms'), 'exact', d['bit_exact'])"; }
for rnd in 1 2 3; do
  for n in ${*:-17 20}; do
    python3 bench.py --log-n $n --no-cpu-baseline --no-host-legs 2>/dev/null | line "default      " $n
    env "$KV" python3 bench.py --log-n $n --no-cpu-baseline --no-host-legs 2>/dev/null | line "$KV" $n
  done
done
