#!/usr/bin/env python3
"""Condenses a tools/profile_bench.sh output directory (gpurun_out/prof_<tag>) into the files
kept under profiles/: <name>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary),
<name>_pmc.json (per-launch counter averages of the count kernel, HBM traffic with the gfx950
FETCH_SIZE correction), and <round>_pmc_traffic_<algo>.json which bench.py reads for roofline.traffic.

usage: tools/summarize_profile.py gpurun_out/prof_<tag> profiles/<name> <kernel substring> bases k algo
"""
import collections, csv, glob, json, os, shutil, sys

src, dst, kname, bases, k, algo = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
os.makedirs(os.path.dirname(dst), exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, dst + "_kernel_stats.csv")
# The first step of a fresh context ramps up through a few small launches; only the
# full-batch launches (duration >= 80 % of the longest) are averaged below.  The raw --stats file
# (all launches) is kept next to this summary.
trace = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))[0]
rows_t = sorted((r for r in csv.DictReader(open(trace)) if kname in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
durs = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in rows_t]
import statistics
thr = 0.75 * statistics.median(durs[-5:])  # the run ends with full-batch launches; the first step's ramp launches are shorter
full = [d for d in durs if d >= thr]  # full-batch launches, in time order
full = full[-5:]  # the 5 timed steps of tools/profile_bench.sh (the warm-up steps before them run at a lower clock)
avg_ns = sum(full) / len(full); calls = len(full)
# the same --stats columns over the timed steps' launches only (the raw file's average also contains the
# few small launches of the first, history-less step)
with open(dst + "_kernel_stats_timed_steps.csv", "w") as f:
    f.write('"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","Note"\n')
    f.write('"%s",%d,%d,%.1f,%d,%d,"last %d full-batch launches of %s (from the kernel trace of the same run)"\n'
            % (kname, calls, int(sum(full)), avg_ns, int(min(full)), int(max(full)), calls, os.path.basename(stats)))
ctr = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(src, "pmc*", "*", "*_counter_collection.csv"))):
    rows = [r for r in csv.DictReader(open(f)) if kname in r["Kernel_Name"]]
    if not rows:
        continue
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dd = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in rows]
    n_ctr = len({r["Counter_Name"] for r in rows})
    thr_p = 0.75 * statistics.median(dd[-5 * n_ctr:])
    for r, d in zip(rows, dd):
        if d >= thr_p:
            ctr[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {c: sum(v) / len(v) for c, v in ctr.items()}
# MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly
# half of the bytes of a wide coalesced streaming read (16 B/lane) -> double it; WRITE_SIZE exact.
fetch = avg.get("FETCH_SIZE", 0) * 1024 * 2
write = avg.get("WRITE_SIZE", 0) * 1024
out = {"kernel": kname, "full_batch_launches_traced": calls, "all_launches_traced": len(durs), "avg_duration_ns": avg_ns, "counters_avg_per_launch": avg,
       "hbm_read_bytes_per_launch": fetch, "hbm_write_bytes_per_launch": write,
       "note": "FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B for 16 B/lane streams); separate --pmc passes"}
json.dump(out, open(dst + "_pmc.json", "w"), indent=1)
json.dump({"workload_bases": bases, "k": k, "algo": algo, "hbm_bytes_per_launch": fetch + write,
           "source": os.path.basename(dst) + "_pmc.json"}, open(os.path.join(os.path.dirname(dst), os.path.basename(dst).split("_")[0] + "_pmc_traffic_" + algo + ".json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
