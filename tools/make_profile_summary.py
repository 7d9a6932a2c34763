#!/usr/bin/env python3
"""Copy the judged artefacts of a gpurun profiling directory (tools/gpu_pmc.sh output) into profiles/:
    python tools/make_profile_summary.py gpurun_out/prof_r01b r01
writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), profiles/<tag>_pmc_summary.txt
(per-kernel averages of every --pmc pass) and profiles/<tag>_traffic.json (HBM bytes per launch per kernel,
FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md, WRITE_SIZE as reported)."""
import csv
import glob
import json
import os
import re
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
forwards = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] else None      # forwards of the profiled command (tools/bench_configs.py prints it): bench.py divides by it
batch = int(sys.argv[4]) if len(sys.argv) > 4 else None         # images per forward of the profiled command, where bench.py runs the config at another batch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)
# (gpurun merges every call's files into the same local directory under their process ids: the NEWEST trace is this profile's)
stats = sorted(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if stats:
    shutil.copy(stats[-1], os.path.join(out, "%s_kernel_stats.csv" % tag))
shutil.copy(os.path.join(src, "summary.txt"), os.path.join(out, "%s_pmc_summary.txt" % tag))
traffic = {}
cur = None
for line in open(os.path.join(src, "summary.txt")):
    if not line.startswith(" "):
        cur = line.split(" n=")[0].strip()
        m = re.search(r"n=(\d+)\s+avg_us=\s*([\d.]+)", line)
        traffic[cur] = {"launches": int(m.group(1)) if m else None, "avg_us": float(m.group(2)) if m else None}
    else:
        m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\d+\s+avg=\s*([\d.]+)", line)
        if m and cur:
            kib = float(m.group(2))
            if m.group(1) == "FETCH_SIZE":
                traffic[cur]["fetch_bytes_x2_corrected"] = kib * 1024 * 2
            else:
                traffic[cur]["write_bytes"] = kib * 1024
        m = re.match(r"\s+(SQ_VALU_MFMA_BUSY_CYCLES|GRBM_GUI_ACTIVE|SQ_INSTS_MFMA|SQ_LDS_BANK_CONFLICT|SQ_LDS_IDX_ACTIVE)\s+n=\d+\s+avg=\s*([\d.]+)", line)
        if m and cur:
            traffic[cur][m.group(1)] = float(m.group(2))
            mp = re.search(r"pass_us=([\d.]+)", line)
            if mp and m.group(1) == "GRBM_GUI_ACTIVE":
                traffic[cur]["pmc_pass_us"] = float(mp.group(1))      # duration of the kernel in the pass that counted its cycles
for k, v in traffic.items():
    if "fetch_bytes_x2_corrected" in v and "write_bytes" in v:
        v["hbm_bytes_per_launch"] = v["fetch_bytes_x2_corrected"] + v["write_bytes"]
    if v.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (= MFMA instructions x their
        # pipe cycles): busy / (1024 x kernel cycles) = share of the matrix pipes' cycles in use = fraction of the dense peak
        # at the clock the kernel actually ran at
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        # the clock these cycles imply, over the duration the kernel had IN THE COUNTER PASS (the trace pass's duration is the
        # fallback for profiles taken before the passes carried timestamps); a clock this part cannot run at means the counter
        # window and the kernel do not match (short kernels: the window is longer than the kernel) -- such a kernel gets NO
        # mfma_util, and says why (4.4 GHz was silently kept)
        dur_us = v.get("pmc_pass_us") or v.get("avg_us")
        clock = cyc / (dur_us * 1e3) if dur_us else None
        if clock is not None and 1.2 <= clock <= 2.7:
            v["mfma_util"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4)
            v["gfx_clock_ghz"] = round(clock, 3)
        else:
            v["mfma_util_rejected"] = ("implied clock %s GHz over %s us (%s) is outside 1.2-2.7 GHz: GRBM_GUI_ACTIVE does not bracket "
                                       "this kernel" % ("%.2f" % clock if clock else "?", dur_us, "counter pass" if v.get("pmc_pass_us") else "trace pass"))
    if v.get("SQ_LDS_IDX_ACTIVE"):
        v["lds_conflict_share"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"], 4)
# Round 5: a kernel whose own GRBM_GUI_ACTIVE window was rejected (dispatches under ~40 us: the window is longer
# than the kernel) still gets a utilisation -- its matrix-pipe busy cycles over its duration IN THE COUNTER PASS at the REFERENCE CLOCK
# of that pass, the median clock of the kernels whose windows were accepted (they ran interleaved with it, in the same process, on
# the same box) -- and says so.  busy / (1024 SIMDs x duration x clock): the same quotient with the denominator taken from neighbours.
clocks = sorted(v["gfx_clock_ghz"] for v in traffic.values() if "gfx_clock_ghz" in v)
if clocks:
    ref = clocks[len(clocks) // 2]
    for v in traffic.values():
        if "mfma_util_rejected" in v and v.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None and (v.get("pmc_pass_us") or v.get("avg_us")):
            dur_us = v.get("pmc_pass_us") or v.get("avg_us")
            v["mfma_util"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * dur_us * 1e3 * ref), 4)
            v["mfma_util_basis"] = "reference clock %.3f GHz (median of the accepted kernels of this pass); own window: %s" % (ref, v.pop("mfma_util_rejected"))
traffic = {k: v for k, v in traffic.items() if "hbm_bytes_per_launch" in v and not k.startswith("void at::") and not k.startswith("__amd")}
sys.path.insert(0, root)
import bench  # noqa: E402  (csrc_hash: bench.py marks the traffic figures stale when the kernels changed since this profile)
# the hash of the kernel sources AS PROFILED (tools/profile_round.sh writes it on the box, next to the counters); a directory without the
# file predates that and gets the hash of the sources at collect time, as before
hf = os.path.join(src, "csrc_hash.txt")
profiled_hash = open(hf).read().strip() if os.path.exists(hf) else bench.csrc_hash()
json.dump({"csrc_hash": profiled_hash, "forwards": forwards, "batch": batch, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request); "
                     "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)",
           "kernels": traffic}, open(os.path.join(out, "%s_traffic.json" % tag), "w"), indent=1)
print(json.dumps(traffic, indent=1))
