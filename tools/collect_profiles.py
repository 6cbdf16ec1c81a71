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
    steady = {}
    for key in ("k_decode_blocks<", "k_encode8_pp<1, 0>", "k_encode8_pp<1, 1>"):
        ds = [(int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3 for t in trace if key in t["Kernel_Name"]]
        f.write("  %-20s n=%d  %s\n" % (key, len(ds), " ".join("%.1f" % x for x in ds[:60])))
        steady[key] = ds
    # the steady state: the timed launches only (bench.py runs `warmup` untimed steps first; the stats table above averages over all of them)
    k = int(d["steps"])
    dec = steady["k_decode_blocks<"][-k:]
    alg = float(d["roofline"]["algorithmic_bytes"])
    if dec:
        avg = sum(dec) / len(dec)
        f.write("\nSTEADY STATE (the last %d launches = the timed steps): k_decode_blocks avg %.1f us -> %.1f GB/s of C + U = %.4f of 8 TB/s   (bench line: kernel_ms %.4f, frac %.4f)\n"
                % (len(dec), avg, alg / (avg * 1e-6) / 1e9, alg / (avg * 1e-6) / 8e12, d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
    e0, e1 = steady["k_encode8_pp<1, 0>"], steady["k_encode8_pp<1, 1>"]
    if e0 and e1:
        n = min(len(e0), len(e1), max(3, k // 4))
        a0, a1 = sum(e0[-n:]) / n, sum(e1[-n:]) / n
        f.write("STEADY STATE encode (the last %d calls): k_encode8_pp<.., 0> (sizes + records) %.1f us + k_encode8_pp<.., 1> (emission) %.1f us (+ the size scan) -> %.4f of 8 TB/s for the two kernels   (bench line: encode.ms %.4f, frac %.4f)\n"
                % (n, a0, a1, alg / ((a0 + a1) * 1e-6) / 8e12, d["encode"]["ms"], d["encode"]["roofline"]["frac"]))
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
