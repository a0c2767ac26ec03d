"""Overlap analysis of a rocprofv3 --kernel-trace run: for the kernels of the last rendered frame, how long 0, 1, 2, ...
extend launches ran at once, how much of bounce/generate time was spent beside an extend launch, and the phase of the
streams.  usage: timeline.py <dir with *_kernel_trace.csv> [frames]"""
import csv, glob, sys, collections
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        kind = "extend" if "k_extend" in n else "bounce" if "k_bounce" in n else "generate" if "k_generate" in n else \
            "reduce" if "k_film_reduce" in n else None
        if kind:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, r["Stream_Id"]))
rows.sort()
# frames end with a film reduce; keep the kernels after the second-to-last reduce
reduces = [i for i, r in enumerate(rows) if r[2] == "reduce"]
if len(reduces) >= 2:
    rows = rows[reduces[-2] + 1:reduces[-1] + 1]
t0, t1 = rows[0][0], max(r[1] for r in rows)
print("frame span %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(rows)))
ev = []
for s, e, k, st in rows:
    ev.append((s, 1, k)); ev.append((e, -1, k))
ev.sort()
active = collections.Counter(); last = t0
hist = collections.Counter(); beside = collections.Counter(); alone = collections.Counter()
for t, dlt, k in ev:
    dt = t - last
    if dt > 0:
        hist[active["extend"]] += dt
        for kk in ("bounce", "generate"):
            if active[kk]:
                (beside if active["extend"] else alone)[kk] += dt
        if not any(active.values()):
            hist["idle"] += dt
    active[k] += dlt; last = t
for k in sorted(hist, key=str):
    print("  extend launches running = %-4s : %7.2f ms" % (k, hist[k] / 1e6))
for kk in ("bounce", "generate"):
    print("  %-8s running beside extend %7.2f ms, without %7.2f ms" % (kk, beside[kk] / 1e6, alone[kk] / 1e6))
per = collections.defaultdict(list)
for s, e, k, st in rows:
    if k == "extend":
        per[st].append(((s - t0) / 1e6, (e - t0) / 1e6))
for st in sorted(per):
    print("  stream %s extend spans (ms): %s" % (st, " ".join("%.1f-%.1f" % p for p in per[st][:8])))
