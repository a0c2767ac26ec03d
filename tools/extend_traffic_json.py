"""Turns the output of tools/profile_round.sh (gpurun_out/<tag>/) into the tracked summaries under profiles/:

    python tools/extend_traffic_json.py <tag> <round-prefix, e.g. r3> [<suffix, e.g. _dragon_dense>]

Writes profiles/<prefix>_extend_traffic<suffix>.json (what bench.py's roofline block reads: it carries the workload it was
taken on and a hash of the kernel sources, and bench.py uses it only when both match what it runs) and copies the text
summaries it is computed from next to it, so that every figure can be recomputed from tracked files.
"""
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

tag, prefix = sys.argv[1], sys.argv[2]
suffix = sys.argv[3] if len(sys.argv) > 3 else ""
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


c = counters(os.path.join(src, "pmc_kernel.txt"))
b = bench_line(os.path.join(src, "bench_default.log"))
mode = b["config"]["render_mode"]
kernel = "k_path_stream" if mode == 2 else "k_extend_stream"
if mode == 2:
    # one kernel, one stream: its time in the default run is its time alone; the bench line has it per timed frame (HIP
    # events around the launches), the kernel-trace total also holds the untimed frames and the oracle-sized sample
    alg = b["roofline"]["algorithmic"]
    alone_ms = alg["avg_launch_ms"] * alg["launches"]
    frames_in_trace = b["steps"]
    alone_src = "bench_default"
else:
    alone_ms, _ = kernel_total_ms(os.path.join(src, "kernel_stats_one_stream.txt"), kernel)
    one = bench_line(os.path.join(src, "bench_one_stream.log"))
    frames_in_trace = one["steps"] + one["warmup"]
    alone_src = "kernel_stats_one_stream"
pr = probe(os.path.join(src, "gather_probe.txt"))
mixed, mixed_args = None, None
for line in open(os.path.join(src, "gather_probe_mixed.txt")):
    if line.startswith("gather_probe2"):
        mixed_args = line.split()[1:]
    m = re.match(r"mode 0:\s+[0-9.]+ ms\s+([0-9.]+) G records/s", line)
    if m:
        mixed = float(m.group(1))
r_l1, r_l2, r_fab = pr[400], pr[16000], pr[1300000]

disp = c["_dispatches"]
traffic_frame = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0          # the counters are in KiB
raw_frame = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
ext_ms_frame = alone_ms / frames_in_trace
l2_hit = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
l1_hit = 1.0 - c["TCP_TCC_READ_REQ_sum"] / (c["TCP_TOTAL_CACHE_ACCESSES_sum"] / 4.0)
f_l1, f_l2, f_fab = l1_hit, (1 - l1_hit) * l2_hit, (1 - l1_hit) * (1 - l2_hit)
tm = b["config"]["traversal_means"]
settled_out = 0 if mode == 2 else b["config"]["rays_escaped_at_creation_per_frame"]
records_frame = (tm["nodes"] + tm["tris"]) * b["config"]["rays_per_frame"] - settled_out
git = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
out = {
    "source": "rocprofv3 --pmc passes (one run each, tools/pmc_passes.sh) and rocprofv3 --kernel-trace --stats of bench.py on "
              "this workload on one MI355X (tools/profile_round.sh); summaries in profiles/%s_pmc_kernels%s.txt, "
              "%s_%s%s.txt, %s_gather_probe*%s.txt" % (prefix, suffix, prefix, alone_src, suffix, prefix, suffix),
    "workload": b["config"]["workload_key"],
    "render_mode": mode,
    "kernel": kernel,
    # sha256 over csrc/, the Makefile and pt_api.h (__graft_entry__.kernel_source_hash): the kernels the counters belong to
    "kernel_source_hash": G.kernel_source_hash(),
    "git_commit": git,
    "dispatches": disp,
    "fetch_size_kib_sum": c["FETCH_SIZE"],
    "write_size_kib_sum": c["WRITE_SIZE"],
    "fetch_correction": 2.0,
    "note": "FETCH_SIZE on gfx950 reports half of the bytes of wide reads (MI355X_MICROARCH.md, HBM section); doubled here. "
            "The counter sits on the L2's fabric side, so reads served by the Infinity Cache are included: records of a tree "
            "that fits its 256 MB stay resident there, and then almost none of these bytes come from HBM itself. The 16 B-"
            "per-lane gathers of this kernel are not a calibrated pattern, so the read side is an upper estimate; raw is half "
            "of it.",
    "traffic_bytes_per_frame": round(traffic_frame),
    "traffic_bytes_per_launch": round(traffic_frame / disp),
    "traffic_bytes_per_launch_raw": round(raw_frame / disp),
    "extend_ms_per_frame_one_stream": round(ext_ms_frame, 3),
    "achieved_traffic_gbs": round(traffic_frame / (ext_ms_frame * 1e-3) / 1e9, 1),
    "l2_hit_rate": round(l2_hit, 4),
    "l1_hit_share": round(l1_hit, 4),
    "fabric_share_of_records": round(f_fab, 4),
    "fabric_reads_per_record": round(c["TCC_EA0_RDREQ_sum"] / (c["TCP_TOTAL_CACHE_ACCESSES_sum"] / 4.0), 4),
    "active_lane_fraction": round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), 4),
    "waves_waiting": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4),
    "ta_busy": round(c["TA_TA_BUSY_sum"] / 256.0 / (c["GRBM_GUI_ACTIVE"] / 8.0), 4),
    "valu_insts_per_vmem_rd_inst": round(c["SQ_INSTS_VALU"] / c["SQ_INSTS_VMEM_RD"], 2),
    "probe_g_records_per_s": {"l1_resident": r_l1, "l2_resident": r_l2, "infinity_cache_resident": r_fab},
    # the three rates chained one after the other (a ray's next fetch waits for this one)
    "record_rate_serial_model": round(1e9 / (f_l1 / r_l1 + f_l2 / r_l2 + f_fab / r_fab)),
    # a bare dependent-gather loop with the kernel's own mix and fetching lanes (tools/gather_probe2 <lanes> <L1 %> <L2 %>,
    # arguments from the counters by tools/probe_mix.py).  A reference rate, not a bound: the probe's fetches are uniformly
    # random within each set, the kernel's are not.
    "probe_mixed_args": mixed_args,
    "probe_mixed_g_records_per_s": mixed,
    # records the kernel fetches per frame (node visits + triangle tests of the rays it traces) over its time alone
    "extend_alone_records_per_s": round(records_frame / (ext_ms_frame * 1e-3)),
    "algorithmic_bytes_per_ray": tm["bytes_per_ray"],
    "nodes_per_ray": tm["nodes"],
    "tris_per_ray": tm["tris"],
    "algorithmic_bytes_per_ray_source": "120 + 64*nodes + 48*tris + 16*leaves, per-ray means counted by the CPU oracle on the "
                                        "bounded sample of the workload (bench line, config.traversal_means)",
}
json.dump(out, open(os.path.join(dst, prefix + "_extend_traffic" + suffix + ".json"), "w"), indent=1)
with open(os.path.join(dst, prefix + "_pmc_kernels" + suffix + ".txt"), "w") as fh:
    fh.write("# rocprofv3 --pmc passes (one run each, tools/pmc_passes.sh) of bench.py --steps 1 --warmup 0 --no-cpu-baseline on %r, "
             "1 x MI355X, render mode %d\n" % (b["config"]["workload_key"], mode))
    for k, f in ((kernel, "pmc_kernel.txt"), ("k_bounce", "pmc_bounce.txt"), ("k_generate", "pmc_generate.txt")):
        if os.path.exists(os.path.join(src, f)) and os.path.getsize(os.path.join(src, f)):
            fh.write("# sums over the dispatches of %s\n" % k)
            fh.write(open(os.path.join(src, f)).read())
for a, bname in (("bench_default.log", "bench_default"), ("kernel_stats.txt", "bench_default_kernel_stats"),
                 ("kernel_stats_one_stream.txt", "kernel_stats_one_stream"), ("bench_one_stream.log", "bench_one_stream"),
                 ("gather_probe.txt", "gather_probe"), ("gather_probe_mixed.txt", "gather_probe_mixed")):
    if os.path.exists(os.path.join(src, a)):
        ext = ".log" if a.endswith(".log") else ".txt"
        shutil.copy(os.path.join(src, a), os.path.join(dst, prefix + "_" + bname + suffix + ext))
print(json.dumps(out, indent=1))
