import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/runc*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("heads_sample")]
a, b = idx[-150], idx[-149]
base = int(rows[a]["End_Timestamp"])
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - base) / 1e3:8.1f} us +{(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:64]}")
print("rollout span ms", (int(rows[idx[-1]]["End_Timestamp"]) - int(rows[idx[-257]]["Start_Timestamp"])) / 1e6)
