#!/bin/bash
# usage (GPU box, via gpurun): bash tools/traffic.sh [tag]  -> gpurun_out/traffic_<tag>/traffic.json (copy it to profiles/<round>_traffic.json)
# HBM traffic of the headline kernels of THIS build: rocprofv3 PMC passes over `bench.py --steps 3 --warmup 1 --no-cpu --no-extras` (one
# counter group per pass; no side measurements, so every counted launch of a kernel name IS the headline launch), averaged per launch, with
# the launch counts and the library's build id (bench.py refuses the file for any other build).
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/traffic_$tag; mkdir -p $O; cd $R
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-30)
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_$n -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras > $O/pmc_$n.log 2> $O/pmc_$n.err || echo "pass failed: $grp"
done
python3 - "$O" <<'PY'
import csv, glob, collections, json, os, sys
sys.path.insert(0, "hypersonic-rle-kit_amd/python")
import hsrle
O = sys.argv[1]
line = None
for f in sorted(glob.glob(O + "/pmc_*.log")):
    for l in open(f):
        if l.startswith("{"):
            line = json.loads(l)
assert line, "no bench line in the pass logs"
size = int(line["config"]["blocks_per_gpu"]) * int(line["config"]["block_size"])
container = int(line["roofline"]["algorithmic_bytes"]) - size
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        # (round 5: the headline encode is two launches of the position-parallel kernel -- <.., 0> sizes + records, <.., 1> emission -- and no compaction)
        name = "decode" if "k_decode_blocks<" in k else ("encode_sizes" if "k_encode8_pp<1, 0>" in k else ("encode_emit" if "k_encode8_pp<1, 1>" in k else None))
        if name:
            vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
def per(name, counter, launches):
    v = vals[name].get(counter, [])
    return sum(v) / launches if v and launches else None
n_dec = len(vals["decode"].get("FETCH_SIZE", []))
n_enc = len(vals["encode_emit"].get("FETCH_SIZE", []))      # one emission launch per compress call
d = {c: per("decode", c, n_dec) for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum")}
e = {k: {c: per(k, c, n_enc) for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum")} for k in ("encode_sizes", "encode_emit")}
fetch_raw = int(d["FETCH_SIZE"] * 1024); write = int(d["WRITE_SIZE"] * 1024); req = int(d["TCC_EA0_RDREQ_sum"])
lo, hi = container, req * 128
enc_lo = sum(int((e[k]["FETCH_SIZE"] + e[k]["WRITE_SIZE"]) * 1024) for k in e)
enc_hi = sum(int(e[k]["TCC_EA0_RDREQ_sum"] * 128 + e[k]["WRITE_SIZE"] * 1024) for k in e)
out = {"codec": line["config"]["codec"], "size": size, "block": int(line["config"]["block_size"]), "library_build_id": hsrle.build_id(),
       "source": "tools/traffic.sh: rocprofv3 --pmc, one counter group per pass, bench.py --steps 3 --warmup 1 --no-cpu --no-extras; averages per launch",
       "launches": {"decode": n_dec, "compress_calls": n_enc}, "raw_counters_per_launch": d, "container_bytes": container,
       "fetch_size_raw_bytes": fetch_raw, "write_bytes": write, "fabric_read_requests": req,
       "fetch_lower_bound_bytes": lo, "fetch_upper_bound_bytes": hi, "traffic_lower_bound_bytes": lo + write, "traffic_upper_bound_bytes": hi + write,
       "algorithmic_bytes": size + container,
       "encode": {"raw_counters_per_call": e, "payload_bytes": int(line["config"]["blocks_per_gpu"]) and container, "traffic_bounds": [enc_lo, enc_hi],
                  "encode_kernel_write_over_payload": round(e["encode_emit"]["WRITE_SIZE"] * 1024 / container, 3)},
       "note": "FETCH_SIZE counts 64 B per fabric read request whether it asks for 64 or 128 bytes (profiles/r02_fetch_calibration.txt): the read side lies between the container (every byte once) and 128 B x requests; WRITE_SIZE is exact for whole-line streaming stores."}
json.dump(out, open(O + "/traffic.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("library_build_id", "launches", "traffic_lower_bound_bytes", "traffic_upper_bound_bytes", "algorithmic_bytes")}))
print("encode", json.dumps(out["encode"]["traffic_bounds"]), out["encode"]["encode_kernel_write_over_payload"])
PY
