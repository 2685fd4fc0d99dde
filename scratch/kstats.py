import csv, glob, sys
d = sys.argv[1]
rows = list(csv.DictReader(open((glob.glob(d + "/runc*kernel_stats.csv") + glob.glob(d + "/r03_run1_bf16_kernel_stats.csv"))[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
print("total kernel ms per iter", tot / n / 1e6)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 14]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.1f} ms {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):5.1f}%  {r['Name'][:88]}")
