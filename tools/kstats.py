"""Prints a compact per-kernel table from a rocprofv3 --kernel-trace --stats output directory."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    n = n.split("(")[0]
    print("%-34s calls %5s total %9.3f ms avg %9.1f us %6.2f%%" % (n[:34], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
t = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
seen = set()
for r in csv.DictReader(open(t)):
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    if k in seen or "rocclr" in k or "at::" in k: continue
    seen.add(k)
    print("  %-30s VGPR %3s SGPR %3s LDS %6s scratch %s grid %s" % (k[:30], r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"], r["Grid_Size_X"]))
