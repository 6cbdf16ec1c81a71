"""Static instruction counts of one kernel per source line:  python tools/isa_lines.py <file.hip> [hipcc flags...]
Compiles the translation unit for gfx950 with line tables (-S -gline-tables-only), attributes every VALU / SALU / LDS / VMEM instruction
to the source line of the nearest .loc directive, and prints the lines with the most VALU instructions (a static count: loops count once).
Used to find what the trip loop of the ring encoders spends its instructions on."""
import collections, os, re, subprocess, sys, tempfile

src = sys.argv[1]
flags = sys.argv[2:]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.gettempdir(), "isa_lines.s")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-gline-tables-only",
       "-I" + os.path.join(repo, "hypersonic-rle-kit_amd", "csrc"), "-I" + os.path.join(repo, "include"), src, "-o", out] + flags
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL, timeout=600)
files = {}
per = collections.defaultdict(lambda: collections.Counter())
cur = None
for line in open(out):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m:
        files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r"\s+(v_|s_|ds_|global_|buffer_|flat_|scratch_)(\w+)", line)
    if m and cur:
        kind = {"v_": "valu", "s_": "salu", "ds_": "lds"}.get(m.group(1), "vmem")
        per[cur][kind] += 1
tot = collections.Counter()
for c in per.values():
    tot.update(c)
print("total", dict(tot))
srcs = {}
for (f, ln), c in sorted(per.items(), key=lambda kv: -kv[1]["valu"])[:int(os.environ.get("TOP", "60"))]:
    path = None
    for d in (os.path.join(repo, "hypersonic-rle-kit_amd", "csrc"), os.path.dirname(os.path.abspath(src))):
        if os.path.exists(os.path.join(d, f)):
            path = os.path.join(d, f)
    text = ""
    if path:
        if path not in srcs:
            srcs[path] = open(path).read().split("\n")
        text = srcs[path][ln - 1].strip()[:110]
    print(f"{f}:{ln:5d} valu {c['valu']:4d} salu {c['salu']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d} | {text}")
