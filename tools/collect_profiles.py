#!/usr/bin/env python3
"""Copies what tools/refresh_profiles.sh left under gpurun_out/ into profiles/<round>_* (the tracked evidence) and assembles the SQ counter table.
usage: python tools/collect_profiles.py r03"""
import glob
import hashlib
import os
import re
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
COPIES = [("final_noextras_stats.txt", "kernel_stats_1080p_no_extras.txt"), ("final_noextras_line.json", "bench_line_1080p_no_extras.json"),
          ("final_default_stats.txt", "kernel_stats_1080p_default_cmd.txt"), ("final_default_line.json", "bench_line_1080p_default_cmd.json"),
          ("final_serial_stats.txt", "kernel_stats_1080p_serial_scales.txt"), ("final_unprofiled_line.json", "bench_line_1080p_unprofiled.json"),
          ("final_timeline.txt", "timeline_1080p_one_step.txt"), ("final_4k_stats.txt", "kernel_stats_4k.txt"), ("final_4k_b12_stats.txt", "kernel_stats_4k_b12.txt"),
          ("final_4k_b12_serial_stats.txt", "kernel_stats_4k_b12_serial_scales.txt"),
          ("final_nonuni_stats.txt", "kernel_stats_1080p_nonuniform.txt"), ("final_nonuni_serial_stats.txt", "kernel_stats_1080p_nonuniform_serial_scales.txt"), ("final_band_timeline.txt", "band_timeline_p1_r_p2_gate.txt"),
          ("final_band_host_trace.txt", "band_host_trace.txt"), ("final_fetch_per_kernel.txt", "pmc_fetch_size_per_kernel.txt")]
for src, dst in COPIES:
    shutil.copyfile(os.path.join(G, src), os.path.join(P, "%s_%s" % (tag, dst)))
shutil.copyfile(os.path.join(G, "pmc_traffic.json"), os.path.join(P, "pmc_traffic.json"))

# SQ counters: pmc_final_{pd,eig,est}_s{1,2,3}.txt lines "<kernel name>  <counter>  n=<dispatches x SE>  avg=<value>"
rows = {}
for f in sorted(glob.glob(os.path.join(G, "pmc_final_*_s?.txt"))):
    for line in open(f):
        m = re.match(r"(\S.*?)\s+(SQ_\w+)\s+n=(\d+)\s+avg=(\S+)", line)
        if not m:
            continue
        name = re.sub(r"\(.*", "", m.group(1))
        name = re.sub(r"void |\(anonymous namespace\)::", "", name).strip()
        rows.setdefault(name, {})[m.group(2)] = (int(m.group(3)), float(m.group(4)))
SRC_HASH = hashlib.sha256(open(os.path.join(R, "bcd_amd", "csrc", "k_similarity_fast.hip"), "rb").read()).hexdigest()[:16]
out = ["# k_similarity_fast.hip sha256_16=%s (bench.py quotes the pair-distance kernel's figures only while this is the source that runs)" % SRC_HASH,
       "# SQ counters of the kernels that ship (rocprofv3 --pmc, three passes with --kernel-trace only; tools/refresh_profiles.sh + tools/collect_profiles.py)",
       "# pairdist: tools/exp_similarity.py --quick (1280x720 scale 0, noisy + clean frame); jacobi: tools/exp_eig.py 32768;",
       "# estimate kernels: bench.py --no-extras with BCD_HIP_SERIAL_SCALES=1 (1080p, 3 scales).  avg per dispatch and SE instance (n = dispatches x 32).", ""]
for name in sorted(rows):
    c = rows[name]
    out.append(name)
    for k in sorted(c):
        out.append("   %-30s n=%-5d avg=%.5g" % (k, c[k][0], c[k][1]))
    g = lambda k: c.get(k, (0, 0.0))[1]
    if g("SQ_BUSY_CYCLES") > 0:
        busy = 8.0 * g("SQ_BUSY_CYCLES")   # 8 CUs per SE; ACTIVE_INST_* count quad-cycles of a CU's pipes
        line = "   -> VALU pipe busy %.1f %% (SQ_ACTIVE_INST_VALU / (8 x SQ_BUSY_CYCLES))" % (100.0 * g("SQ_ACTIVE_INST_VALU") / busy)
        if g("SQ_LDS_IDX_ACTIVE") > 0:
            line += ", LDS array busy %.1f %%, bank conflicts %.2f %% of its cycles" % (100.0 * g("SQ_LDS_IDX_ACTIVE") / busy,
                                                                                       100.0 * g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"))
        if g("SQ_VALU_MFMA_BUSY_CYCLES") > 0:
            line += ", matrix core busy %.1f %% (SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES))" % (100.0 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * busy))
        out.append(line)
    out.append("")
open(os.path.join(P, "%s_pmc_sq_counters.txt" % tag), "w").write("\n".join(out))
for extra in ("final_eig.log", "final_host.log", "final_sizes.log"):
    if os.path.exists(os.path.join(G, extra)):
        print("---- " + extra)
        print(open(os.path.join(G, extra)).read())
