"""Per-kernel register / spill / scratch table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python tools/kernel_resources.py <file.hip> [name filter]"""
import re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-fno-vectorize", "-Iinclude",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_kr.o"] + [a for a in sys.argv[3:]]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None; rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (.*?):(\d+):\d+: +(.*?) \[-Rpass-analysis", line) or re.search(r"(.*?):(\d+):\d+: remark: +(.*?) \[-Rpass-analysis", line)
    if not m:
        continue
    body = m.group(3).strip()
    if body.startswith("Function Name:"):
        cur = body.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in body:
        k, v = body.split(":", 1); rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0]
    if flt and flt not in dem:
        continue
    print(f"{dem[:70]:70s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>4s} spillV {r.get('VGPRs Spill','?'):>3s} spillS {r.get('SGPRs Spill','?'):>3s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} occ {r.get('Occupancy [waves/SIMD]','?')} LDS {r.get('LDS Size [bytes/block]','?')}")
