#!/bin/bash
# timeline of ONE streamed host call at 2^20 from pinned memory: kernels (per queue) and copies with start/end relative to the call's first
# event -- where the 2.6 ms go (rocprofv3 --kernel-trace --memory-copy-trace).   usage: tools/streamed_timeline.sh   (on the GPU box)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/streamed_timeline
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o t -- python3 $GRAFT_REPO_ROOT/tools/host_call_trace.py pinned > $O/run.log 2>&1
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
# the last call = everything after the last k_reduce_bits_wide but one
ends = [i for i, e in enumerate(ev) if "k_reduce_bits" in e[2]]
lo = ends[-2] + 1 if len(ends) >= 2 else 0
t0 = ev[lo][0]
for s, e, name in ev[lo: ends[-1] + 1]:
    print("%9.1f %9.1f  %7.1f us  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, name))
PY
