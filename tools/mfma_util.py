"""MFMA-pipe utilisation per kernel class from a rocprofv3 PMC pass with SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE:
util = sum(MFMA busy cycles over SIMDs) / (elapsed cycles x 1024 SIMDs), elapsed cycles = GRBM_GUI_ACTIVE / 8 XCDs."""
import collections, csv, sys
busy, act, dur, n = (collections.defaultdict(float) for _ in range(4))
def cls(name):
    for key, c in (("gemm_kernel", "gemm"), ("attn_kernel", "attention"), ("layernorm", "layernorm"), ("rmsnorm", "rmsnorm")):
        if key in name:
            return c
    return "other"
for r in csv.DictReader(open(sys.argv[1])):
    c = cls(r["Kernel_Name"])
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        busy[c] += float(r["Counter_Value"]); dur[c] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); n[c] += 1
    elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        act[c] += float(r["Counter_Value"])
tb = ta = 0.0
for c in sorted(busy, key=lambda k: -dur[k]):
    cyc = act[c] / 8.0
    print(f"{c:10s} launches {int(n[c]):4d}  time {dur[c]/1e6:8.2f} ms  clock {cyc/max(dur[c],1):.2f} GHz  MFMA busy {100*busy[c]/max(cyc*1024,1):5.1f} %")
    tb += busy[c]; ta += cyc
print(f"whole profile: MFMA busy {100*tb/max(ta*1024,1):.1f} % of SIMD cycles")
