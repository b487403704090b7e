#!/bin/bash
# VERDICT r3 item 4, measured proxy: ONE 2^N MSM against TWO 2^(N-1) MSMs in flight on the same GPU (msm_multi {0,0}: two contexts, two host threads,
# host fold) -- the second MSM's sort runs beside the first one's accumulation and the first one's bucket reduction beside the second one's accumulation.
# Each half carries a FULL bucket reduction (all windows), which a window split would not.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for n in 20 19 17; do
  for r in 1 2 3; do
    a=$(python bench.py --log-n $n --steps 30 --warmup 5 --no-cpu-baseline --no-host-legs 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'])")
    b=$(python bench.py --log-n $n --gpus 2 --in-process --debug-same-device --steps 30 --warmup 5 --no-cpu-baseline --no-host-legs 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['exchange']['shard_ms_max'], j['exchange']['shard_ms_min'])")
    echo "2^$n round $r: one MSM $a ms | two half-size MSMs in flight (ms, shard max, shard min) $b"
  done
done
