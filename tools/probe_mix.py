"""Prints the arguments for tools/gather_probe2 that reproduce a traversal kernel's access mix, from its counter sums
(output of tools/pmc_sum.py): lanes of 64 that fetch (the mean number of distinct 64 B segments a vector load instruction of
the kernel touches: every lane with a ray fetches its own record), percent of record fetches that hit L1, percent that miss
L1 and hit L2."""
import re
import sys

c = {}
for line in open(sys.argv[1]):
    m = re.match(r"(\S+)\s+([0-9.e+]+)\s+\(dispatches", line)
    if m:
        c[m.group(1)] = float(m.group(2))
lanes = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / c["SQ_INSTS_VMEM_RD"]
l1 = 1.0 - c["TCP_TCC_READ_REQ_sum"] / (c["TCP_TOTAL_CACHE_ACCESSES_sum"] / 4.0)
l2 = (1.0 - l1) * c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
print("%d %d %d" % (round(lanes), round(100 * l1), round(100 * l2)))
