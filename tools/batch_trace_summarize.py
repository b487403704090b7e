"""Timeline of tools/batch_trace.py's LAST batch from the rocprofv3 CSVs: per queue/stream busy intervals + copies.  argv: dir"""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "Q" + r["Queue_Id"], r["Kernel_Name"].split("(")[0].split("<")[0][-28:]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY", r["Direction"]))
ev.sort()
end = ev[-1][1]
# the last batch = everything after the last gap > 3 ms ... simply print the last 140 events relative to the end
tail = ev[-int(sys.argv[2]) if len(sys.argv) > 2 else -140:]
t0 = tail[0][0]
for s, e, q, name in tail:
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f}  {q:6s} {name}")
