"""Copy the judged summaries of a final measurement run (tools/final_measure.sh <tag> -> gpurun_out/<tag>_final) into profiles/ under a round tag:
  python tools/collect_profiles.py gpurun_out/r04_final r04"""
import csv, json, shutil, sys, os
src, tag = sys.argv[1], sys.argv[2]
P = "profiles"
line = [l for l in open(src + "/bench.json") if l.startswith("{")][-1]
open(f"{P}/{tag}_bench.json", "w").write(line)
shutil.copy(src + "/stats/k_kernel_stats.csv", f"{P}/{tag}_kernel_stats.csv")
shutil.copy(src + "/traffic.json", f"{P}/{tag}_traffic.json")
shutil.copy(src + "/small_containers.md", f"{P}/{tag}_small_containers.md")
d = json.loads(line)
rows = list(csv.DictReader(open(src + "/stats/k_kernel_stats.csv")))
trace = list(csv.DictReader(open(src + "/stats/k_kernel_trace.csv")))
with open(f"{P}/{tag}_rocprof_summary.txt", "w") as f:
    f.write(f"rocprofv3 --kernel-trace --stats -- python3 bench.py --no-extras   (library build {d.get('library_build_id', '?')})\n")
    f.write("bench line of the default run (with extras): value %.1f GiB/s, ms_per_step %.4f, roofline.kernel_ms %.4f, frac %.4f; encode %.1f GiB/s (%.4f)\n\n"
            % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["encode"]["value"], d["encode"]["roofline"]["frac"]))
    f.write("%-90s %8s %12s %12s %12s %7s\n" % ("kernel", "calls", "avg us", "min us", "max us", "%"))
    for r in rows:
        f.write("%-90s %8s %12.1f %12.1f %12.1f %7s\n" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    f.write("\nper-launch durations of the headline kernels (us, launch order):\n")
    for key in ("k_decode_blocks<", "k_encode8_blocks<", "k_compact("):
        ds = [(int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3 for t in trace if key in t["Kernel_Name"]]
        f.write("  %-20s n=%d  %s\n" % (key, len(ds), " ".join("%.1f" % x for x in ds[:60])))
frows = list(csv.DictReader(open(src + "/frame/f_kernel_stats.csv")))
with open(f"{P}/{tag}_config3_encode_kernel_stats.txt", "w") as f:
    f.write("rocprofv3 --kernel-trace --stats -- python3 tools/frame_prof.py rle64_3symlut_byte   (10 encodes of the 88 MB frame, 4 KiB blocks)\n")
    f.write("%-90s %8s %12s\n" % ("kernel", "calls", "avg us"))
    for r in frows:
        f.write("%-90s %8s %12.1f\n" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
if os.path.exists(src + "/split/s_kernel_stats.csv"):
    srows = list(csv.DictReader(open(src + "/split/s_kernel_stats.csv")))
    with open(f"{P}/{tag}_config3_split_kernel_stats.txt", "w") as f:
        f.write("rocprofv3 --kernel-trace --stats -- python3 tools/split_bench.py   (the 88 MB frame, rle64_3symlut_byte, 4 KiB blocks: plain and split decode; library build %s)\n" % d.get("library_build_id", "?"))
        f.write("%-90s %8s %12s\n" % ("kernel", "calls", "avg us"))
        for r in srows:
            f.write("%-90s %8s %12.1f\n" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
        for l in open(src + "/split.log"):
            if "us" in l and "GiB/s" in l:
                f.write(l)
print("ok")
