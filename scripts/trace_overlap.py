"""Prints start/end (us, relative) of the hash-grid backward kernels of the last few iterations in a rocprofv3 kernel trace."""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "hashgrid_bwd" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("<")[0].replace("void ", "")))
rows.sort()
rows = rows[-12:]
t0 = rows[0][0]
for s, e, n in rows:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f}  {(e - s) / 1e3:7.1f} us  {n}")
