"""Print the device timeline (kernels + memory copies) of a few steady-state steps from a rocprofv3
--kernel-trace --memory-copy-trace csv output directory."""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", "")))
ev.sort()
mid = len(ev) // 2
base, prev = ev[mid][0], None
for s, e, name in ev[mid:mid + int(sys.argv[2]) if len(sys.argv) > 2 else mid + 24]:
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s-base)/1e3:10.1f} us  dur {(e-s)/1e3:8.1f}  gap {gap:7.1f}  {name}")
    prev = e
