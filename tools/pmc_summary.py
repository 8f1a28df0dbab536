import csv, sys, collections
pat = sys.argv[2]
d = collections.defaultdict(list); dur = []
for r in csv.DictReader(open(sys.argv[1])):
    if pat in r["Kernel_Name"]:
        d[r["Counter_Name"]].append(float(r["Counter_Value"])); dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for c, v in sorted(d.items()): print(f"{c:32s} {sum(v)/len(v):.5g}")
if dur: print("dur_ns", sum(dur) / len(dur))
