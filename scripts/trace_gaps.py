"""Idle time between kernels of the steady-state training step, from a rocprofv3 --kernel-trace CSV.
usage: python scripts/trace_gaps.py <dir with *_kernel_trace.csv> [marker kernel substring = hashgrid_fwd] [min marker us] [regime]
`regime` (headline | large_batch_regime | dropin_regime) keeps only the launches between that regime's sentinel launches
(WISP_BENCH_SENTINELS=1; scripts/regime_stats.py) - without it the LAST steps of the command are taken, whatever regime ran last
(that is how the drop-in loop's steps were once filed as the 2^18 regime's).
Steps are delimited by the marker kernel; the last 5 complete steps are summarised: busy time, idle time and the idle
time attributed to the kernel that FOLLOWS each gap (the launch that arrived late).  With a minimum marker duration only
steps whose marker kernel ran at least that long count (bench.py times the 2^21 regime first and the 2^18 regime last:
`hashgrid_fwd 100` selects the former)."""
import csv, glob, os, sys, collections

root = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "hashgrid_fwd"
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
regime = sys.argv[4] if len(sys.argv) > 4 else None
if regime:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import regime_stats
    full = regime_stats.load(root)
    lo, hi = regime_stats.regime_windows(full)[regime]
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi and "FillFunctor<double>" not in r[2]]
    print(f"regime {regime}: {len(rows)} launches between its sentinels")
min_ns = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 0.0
marks = [i for i, r in enumerate(rows) if marker in r[2]]
if len(marks) < 7:
    sys.exit(f"only {len(marks)} marker kernels found")
pairs = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1)
         if rows[marks[i]][1] - rows[marks[i]][0] >= min_ns and rows[marks[i + 1]][1] - rows[marks[i + 1]][0] >= min_ns
         and marks[i + 1] - marks[i] < 64]
if len(pairs) < 6:
    sys.exit(f"only {len(pairs)} qualifying steps found")
steps = pairs[-6:-1]
busy = idle = 0
gap_by = collections.Counter(); time_by = collections.Counter(); n_by = collections.Counter()
# (kernels of two streams overlap - the count read-back's copy runs beside the hash-grid forward: busy = the UNION of the kernel
#  intervals, a gap = time in which NO kernel runs, charged to the kernel that ends it.  Until round 5 a gap was measured against the
#  previous row's end, which counted the whole forward as "idle" behind the short copy: 108 of the 167 us reported then.)
for a, b in steps:
    horizon = rows[a][0]
    for i in range(a, b):
        s, e, name = rows[i]
        short = name.split("(")[0].split("<")[0].replace("void ", "").replace("(anonymous namespace)::", "")[-60:]
        time_by[short] += e - s; n_by[short] += 1
        if s > horizon:
            idle += s - horizon; gap_by[short] += s - horizon
        busy += max(0, e - max(s, horizon))
        horizon = max(horizon, e)
    if rows[b][0] > horizon:
        nshort = rows[b][2].split("(")[0].split("<")[0].replace("void ", "").replace("(anonymous namespace)::", "")[-60:]
        idle += rows[b][0] - horizon; gap_by[nshort] += rows[b][0] - horizon
n = len(steps)
print(f"steps {n}: wall {(busy + idle) / n / 1e3:.1f} us  busy {busy / n / 1e3:.1f} us  idle {idle / n / 1e3:.1f} us  launches/step {sum(n_by.values()) / n:.1f}")
print("kernel".ljust(62), "calls/step   us/step   idle-before us/step   (busy = union of kernel intervals)")
for k, t in time_by.most_common():
    print(k.ljust(62), f"{n_by[k] / n:9.1f} {t / n / 1e3:9.1f} {gap_by[k] / n / 1e3:12.1f}")
# the last summarised step in launch order
a, b = steps[-1]
t0 = rows[a][0]
print("\nlast step in order:  start us   dur us   gap-before us")
horizon = rows[a][0]
for i in range(a, b):
    s0, e0, name = rows[i]
    short = name.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[-70:]
    print(f"  {short.ljust(70)} {(s0 - t0) / 1e3:9.1f} {(e0 - s0) / 1e3:8.1f} {max(0, s0 - horizon) / 1e3:8.1f}")
    horizon = max(horizon, e0)
