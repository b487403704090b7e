#!/bin/bash
# round 4, first GPU call: sustained probe + clock probe, GLV on/off as whole processes alternated on ONE box, SMU clocks beside it
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4a; mkdir -p $O
( for i in $(seq 1 400); do rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n'; echo; sleep 0.2; done ) > $O/smi.jsonl &
SMI=$!
timeout 300 python tools/sustained_probe.py > $O/sustained_probe.txt 2>&1
for i in 1 2 3; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-host-legs --no-cpu-baseline > $O/bench_glv_$i.json 2> $O/bench_glv_$i.err
  timeout 200 python bench.py --steps 20 --warmup 5 --no-host-legs --no-cpu-baseline --no-glv > $O/bench_noglv_$i.json 2> $O/bench_noglv_$i.err
done
kill $SMI 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
timeout 200 python bench.py --gpus 2 --debug-same-device --steps 3 --warmup 1 --no-host-legs --no-cpu-baseline > $O/bench_selflaunch_2x.json 2> $O/bench_selflaunch_2x.err
echo "rc $?" >> $O/bench_selflaunch_2x.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4a/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, j["value"], j["roofline"]["avg_kernel_ms"], j.get("clock"), j["config"]["glv_split"], j["bit_exact"])
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/sustained_probe.txt
