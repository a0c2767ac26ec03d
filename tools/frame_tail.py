"""Per-iteration view of the last frame of a rocprofv3 --kernel-trace run: for every extend launch its start, duration and
the idle gap before it on its stream.  Shows where a short frame (a rank's share at 8 GPUs) spends its fixed time.
usage: frame_tail.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys, collections
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        kind = next((k for k in ("k_extend", "k_path_stream", "k_bounce", "k_generate", "k_film_reduce", "k_take", "k_advance", "k_tail") if k in n), None)
        if kind:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, r["Stream_Id"]))
rows.sort()
reduces = [i for i, r in enumerate(rows) if r[2] == "k_film_reduce"]
rows = rows[reduces[-2] + 1:reduces[-1] + 1]
t0 = rows[0][0]
print("frame span %.3f ms" % ((max(r[1] for r in rows) - t0) / 1e6))
busy = collections.Counter()
for s, e, k, st in rows:
    busy[k] += e - s
print("kernel time summed:", {k: round(v / 1e6, 3) for k, v in busy.items()})
# union of all kernel spans = device busy time
spans = sorted((s, e) for s, e, _, _ in rows)
tot = 0; cs, ce = spans[0]
for s, e in spans[1:]:
    if s > ce:
        tot += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
tot += ce - cs
print("device busy (union) %.3f ms" % (tot / 1e6))
by_stream = collections.defaultdict(list)
for r in rows:
    by_stream[r[3]].append(r)
for st, rs in by_stream.items():
    print("stream", st)
    prev_end = None
    for s, e, k, _ in rs:
        if k in ("k_extend", "k_path_stream", "k_film_reduce", "k_generate", "k_bounce", "k_tail"):
            print("  %-14s start %8.3f  dur %8.3f  gap before %7.3f" % (k, (s - t0) / 1e6, (e - s) / 1e6, 0 if prev_end is None else (s - prev_end) / 1e6))
        prev_end = e
