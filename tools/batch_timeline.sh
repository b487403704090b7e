#!/bin/bash
# kernel/copy timeline of the LAST resident batch of tools/batch_trace.py (8 MSMs, two in flight), per hardware queue.
# usage (GPU box): tools/batch_timeline.sh LOG_N table|plain [NAME=VALUE ...]
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/batch_timeline
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o t -- python3 $GRAFT_REPO_ROOT/tools/batch_trace.py "$@" > $O/run.log 2>&1
tail -1 $O/run.log
python3 - "$O" <<'PY'
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s %s" % (r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("msmk::", "")[:34])))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s B" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))))
ev.sort()
ends = [i for i, e in enumerate(ev) if "k_reduce_bits" in e[2]]
lo = ends[-4] + 1 if len(ends) >= 4 else 0   # the last three MSMs
t0 = ev[lo][0]
for s, e, name in ev[lo: ends[-1] + 1]:
    print("%9.1f %9.1f  %7.1f us  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, name))
PY
