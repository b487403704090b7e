: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"  # default: the checkout this script lives in
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc2
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/pmc2/a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc2/a.log 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/pmc2/a/*/*counter_collection.csv')[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'].split('(')[0][-24:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'combine' in k or 'pair_level' in k or 'bits' in k:
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
