"""Compile one .hip file for gfx950 and print VGPR / AGPR / scratch / LDS / occupancy per kernel (build-container tool)."""
import re, subprocess, sys, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-I", os.path.join(root, "include"),
                    "-c", src, "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        name = k.replace("_ZN12_GLOBAL__N_1", "")[:70]
        print(f"{name:70s} vgpr {v.get('VGPRs', -1):4d} agpr {v.get('AGPRs', -1):4d} scratch {v.get('ScratchSize', -1):4d} "
              f"lds {v.get('LDS Size', -1):7d} occ {v.get('Occupancy', -1)}")
if r.returncode:
    print(r.stderr[-3000:])
