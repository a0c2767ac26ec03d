"""Turns the output of tools/profile_round.sh (gpurun_out/<tag>/) into the tracked summaries under profiles/:

    python tools/extend_traffic_json.py <tag> <round-prefix, e.g. r2>

Writes profiles/<prefix>_extend_traffic.json (what bench.py's roofline block reads) and copies the text summaries it is
computed from next to it, so that every figure can be recomputed from tracked files.
"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, prefix = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")


def counters(path):
    out = {}
    for line in open(path):
        m = re.match(r"(\S+)\s+([0-9.e+]+)\s+\(dispatches (\d+)\)", line)
        if m:
            out[m.group(1)] = float(m.group(2))
            out["_dispatches"] = int(m.group(3))
    return out


def kernel_total_ms(path, name):
    for line in open(path):
        if name in line and "total" in line:
            return float(re.search(r"total\s+([0-9.]+) ms", line).group(1)), int(re.search(r"calls\s+(\d+)", line).group(1))
    raise SystemExit("no %s in %s" % (name, path))


def bench_line(path):
    for line in open(path):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit("no bench line in " + path)


def probe(path):
    rates, n = {}, None
    for line in open(path):
        if line.startswith("records"):
            n = int(line.split()[1])
        m = re.match(r"mode 0:\s+[0-9.]+ ms\s+([0-9.]+) G records/s", line)
        if m:
            rates[n] = float(m.group(1))
    return rates


c = counters(os.path.join(src, "pmc_extend.txt"))
b = bench_line(os.path.join(src, "bench_default.log"))
ext_ms_1, calls_1 = kernel_total_ms(os.path.join(src, "kernel_stats_one_stream.txt"), "k_extend_stream")
frames_1 = bench_line(os.path.join(src, "bench_one_stream.log"))
frames_in_trace = frames_1["steps"] + frames_1["warmup"]
pr = probe(os.path.join(src, "gather_probe.txt"))
mixed = None
if os.path.exists(os.path.join(src, "gather_probe_mixed.txt")):
    for line in open(os.path.join(src, "gather_probe_mixed.txt")):
        m = re.match(r"mode 0:\s+[0-9.]+ ms\s+([0-9.]+) G records/s", line)
        if m:
            mixed = float(m.group(1))
r_l1, r_l2, r_fab = pr[400], pr[16000], pr[1300000]

disp = c["_dispatches"]
traffic_frame = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0          # the counters are in KiB
raw_frame = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
ext_ms_frame = ext_ms_1 / frames_in_trace
l2_hit = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
l1_hit = 1.0 - c["TCP_TCC_READ_REQ_sum"] / (c["TCP_TOTAL_CACHE_ACCESSES_sum"] / 4.0)
f_l1, f_l2, f_fab = l1_hit, (1 - l1_hit) * l2_hit, (1 - l1_hit) * (1 - l2_hit)
tm = b["config"]["traversal_means"]
out = {
    "source": "rocprofv3 --pmc passes (one run each, tools/pmc_passes.sh) of `python3 bench.py --no-cpu-baseline --steps 1 "
              "--warmup 0` and rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu-baseline --tune 9=1` on one "
              "MI355X; summaries in profiles/%s_pmc_kernels.txt, %s_kernel_stats_one_stream.txt, %s_gather_probe.txt"
              % (prefix, prefix, prefix),
    "kernel": "k_extend_stream<false,false>",
    "dispatches": disp,
    "fetch_size_kib_sum": c["FETCH_SIZE"],
    "write_size_kib_sum": c["WRITE_SIZE"],
    "fetch_correction": 2.0,
    "note": "FETCH_SIZE on gfx950 reports half of the bytes of wide reads (MI355X_MICROARCH.md, HBM section); doubled here. "
            "The counter sits on the L2's fabric side, so reads served by the Infinity Cache are included: the BVH records "
            "(82 MB) stay resident there, so almost none of these bytes come from HBM itself. The 16 B-per-lane gathers of "
            "this kernel are not a calibrated pattern, so the read side is an upper estimate; raw is half of it.",
    "traffic_bytes_per_frame": round(traffic_frame),
    "traffic_bytes_per_launch": round(traffic_frame / disp),
    "traffic_bytes_per_launch_raw": round(raw_frame / disp),
    "extend_ms_per_frame_one_stream": round(ext_ms_frame, 3),
    "achieved_traffic_gbs": round(traffic_frame / (ext_ms_frame * 1e-3) / 1e9, 1),
    "l2_hit_rate": round(l2_hit, 4),
    "l1_hit_share": round(l1_hit, 4),
    "fabric_share_of_records": round(f_fab, 4),
    "active_lane_fraction": round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), 4),
    "valu_busy": round(4.0 * c["SQ_ACTIVE_INST_VALU"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0), 4),
    "waves_waiting": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4),
    "ta_busy": round(c["TA_TA_BUSY_sum"] / 256.0 / (c["GRBM_GUI_ACTIVE"] / 8.0), 4),
    "valu_insts_per_vmem_rd_inst": round(c["SQ_INSTS_VALU"] / c["SQ_INSTS_VMEM_RD"], 2),
    "probe_g_records_per_s": {"l1_resident": r_l1, "l2_resident": r_l2, "infinity_cache_resident": r_fab},
    # every record passes L1, (1 - l1_hit) of them L2, f_fab of them the fabric: the slowest of the three sets the ceiling
    "record_rate_ceiling": round(min(r_l1, r_l2 / (1 - l1_hit), r_fab / f_fab) * 1e9),
    "record_rate_ceiling_binding": ["l1", "l2", "fabric"][[r_l1, r_l2 / (1 - l1_hit), r_fab / f_fab].index(
        min(r_l1, r_l2 / (1 - l1_hit), r_fab / f_fab))],
    # the same three rates chained one after the other (a ray's next fetch waits for this one)
    "record_rate_serial_model": round(1e9 / (f_l1 / r_l1 + f_l2 / r_l2 + f_fab / r_fab)),
    # a bare dependent-gather loop with extend's own mix and lane occupancy (41 of 64 lanes, 38 % L1 set, 41 % L2 set)
    "probe_mixed_g_records_per_s": mixed,
    # records extend fetches per frame (node visits + triangle tests of the rays that reach it) over its one-stream time
    "extend_alone_records_per_s": round(((tm["nodes"] + tm["tris"]) * b["config"]["rays_per_frame"]
                                         - b["config"]["rays_escaped_at_creation_per_frame"]) / (ext_ms_frame * 1e-3)),
    "algorithmic_bytes_per_ray": tm["bytes_per_ray"],
    "nodes_per_ray": tm["nodes"],
    "tris_per_ray": tm["tris"],
    "algorithmic_bytes_per_ray_source": "120 + 64*nodes + 48*tris + 16*leaves, per-ray means counted by the CPU oracle on "
                                        "dragon.scene 960x540 320 spp (bench line, config.traversal_means)",
}
json.dump(out, open(os.path.join(dst, prefix + "_extend_traffic.json"), "w"), indent=1)
with open(os.path.join(dst, prefix + "_pmc_kernels.txt"), "w") as fh:
    fh.write("# rocprofv3 --pmc passes (one run each, tools/pmc_passes.sh) of: python3 bench.py --no-cpu-baseline --steps 1 "
             "--warmup 0   (dragon stand-in 1920x1080 512 spp, 1 x MI355X, 3 streams)\n")
    for k in ("extend", "bounce", "generate"):
        fh.write("# sums over the dispatches of k_%s\n" % k)
        fh.write(open(os.path.join(src, "pmc_%s.txt" % k)).read())
for a, bname in (("bench_default.log", "bench_default.log"), ("kernel_stats.txt", "bench_default_kernel_stats.txt"),
                 ("kernel_stats_one_stream.txt", "kernel_stats_one_stream.txt"),
                 ("bench_one_stream.log", "bench_one_stream.log"), ("gather_probe.txt", "gather_probe.txt"),
                 ("gather_probe_mixed.txt", "gather_probe_mixed.txt")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, prefix + "_" + bname))
print(json.dumps(out, indent=1))
