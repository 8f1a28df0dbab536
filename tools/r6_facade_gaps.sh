#!/bin/bash
# Round 6: where the DEVICE idles in a facade forward (hip_graph): rocprofv3 kernel trace of tools/r6_facade_profile.py, gaps > 20 us between
# consecutive kernels of the last forwards, with the kernels on both sides of each gap.
R=$(cd "$(dirname "$0")/.." && pwd); out=$R/gpurun_out/r6_facade_gaps; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/prof -- python $R/tools/r6_facade_profile.py > $out/run.log 2>&1
echo "rc $?"
python - $out <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for f in glob.glob(out + "/prof/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
rows.sort()
# the last 5 forwards: find decode kernels as forward delimiters
dec = [i for i, r in enumerate(rows) if "decode_kernel" in r[2]]
lo = dec[-6] if len(dec) >= 6 else 0
sel = rows[lo:dec[-1] + 40]
busy_end = sel[0][1]
gaps = []
for s, e, n in sel[1:]:
    if s - busy_end > 20000:
        gaps.append((s - busy_end, prev, n))
    if e > busy_end:
        busy_end, prev = e, n
    prev = n if e >= busy_end else prev
tot = sum(g[0] for g in gaps)
print("forwards covered: 5; total idle in gaps > 20 us: %.3f ms = %.3f ms per forward" % (tot / 1e6, tot / 5e6))
for g in sorted(gaps, key=lambda g: -g[0])[:25]:
    print("  %8.1f us   after %-60s before %s" % (g[0] / 1e3, g[1], g[2]))
PY
rm -rf $out/prof
