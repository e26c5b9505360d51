"""Summarise tools/prof_round.sh output: kernel stats (avg duration per kernel) and PMC-derived HBM traffic.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of 1 KB; on gfx950 FETCH_SIZE counts 128-byte
requests at 64 bytes, so it is doubled (MI355X_MICROARCH.md, "HBM"); WRITE_SIZE is taken as is (uncalibrated)."""
import collections
import csv
import glob
import json
import os
import sys

O = sys.argv[1]


def short(name):
    name = name.replace("atlas_amd::trans::", "").replace("atlas_amd::fft::", "").replace("void ", "")
    p = name.find("(")
    return name[:p] if p > 0 else name


def stage_of(k):
    if k.startswith("legendre_kernel"):
        return "legendre_kernel"
    if k.startswith("fft_rows"):
        return "fourier_stage"
    return None


# ---- kernel stats from the kernel trace (start/end timestamps) ----
lines = []
for f in glob.glob(os.path.join(O, "stats", "**", "*kernel_trace.csv"), recursive=True):
    dur = collections.defaultdict(list)
    for row in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"])):   # launch order
        dur[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    tot = sum(sum(v) for v in dur.values())
    # "steady_us": average without the first WARM launches of the kernel (the bench's untimed warm-up steps: first touch of the
    # tables, cold L2 / Infinity Cache, clock ramp -- 10.8 / 9.2 / 8.5 ms for the Legendre kernel against 8.2 - 8.3 ms afterwards);
    # this is the figure bench.py's HIP-event average over the timed steps must agree with
    WARM = int(os.environ.get("PROF_WARMUP_LAUNCHES", "3"))
    lines.append("%-70s %8s %12s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "steady_us", "%"))
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        st = v[WARM:] if len(v) > WARM else v
        lines.append("%-70s %8d %12.3f %12.1f %12.1f %7.2f" % (k[:70], len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3,
                                                               sum(st) / len(st) / 1e3, 100.0 * sum(v) / tot))
open(os.path.join(O, "kernel_stats.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:16]))

# ---- PMC ----
per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(os.path.join(O, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        per_kernel[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
stage = collections.defaultdict(lambda: collections.defaultdict(float))
stage_launch = {}
print()
for k, cs in per_kernel.items():
    st = stage_of(k)
    if not st:
        continue
    print(k)
    for c, v in cs.items():
        n = len(disp[(k, c)])
        print("     %-34s per-dispatch %.5g  (%d dispatches)" % (c, v / n, n))
        stage[st][c] += v
        if st == "legendre_kernel":
            stage_launch[(st, c)] = n
# transforms profiled in the PMC passes = number of legendre dispatches
ntr = stage_launch.get(("legendre_kernel", "FETCH_SIZE"), 0) or 1
traffic = {}
detail = {}
for st, cs in stage.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        rd = 2.0 * cs["FETCH_SIZE"] * 1e3 / ntr
        wr = cs["WRITE_SIZE"] * 1e3 / ntr
        traffic[st] = rd + wr
        detail[st] = {"read_bytes(2xFETCH_SIZE)": rd, "write_bytes(WRITE_SIZE)": wr}
# vector-ALU wave-instructions per transform (SQ_INSTS_VALU of the "sq" pass): the issue floor of a stage that is bound by
# its own instruction stream, not by memory (bench.py: roofline.valu_issue_floor_ms)
valu = {}
for st, cs in stage.items():
    n = stage_launch.get(("legendre_kernel", "SQ_INSTS_VALU"), 0) or ntr
    if "SQ_INSTS_VALU" in cs:
        valu[st] = cs["SQ_INSTS_VALU"] / n
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import bench
    digest = bench.kernel_source_digest()
except Exception:  # noqa: BLE001
    digest = None
out = {"traffic_bytes_per_launch": traffic, "detail": detail, "transforms_profiled": ntr,
       "valu_wave_instructions_per_transform": valu,
       "kernel_source_sha256": digest,
       "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes over bench.py --steps 2 --warmup 1; "
               "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B); fourier_stage = sum over the row-class "
               "launches of one transform"}
json.dump(out, open(os.path.join(O, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
