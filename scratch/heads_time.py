import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
h = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if r["Kernel_Name"].startswith("heads_sample")]
n = 257
for it in range(len(h) // n):
    seg = h[it * n:(it + 1) * n]
    print(f"iter {it}: t=0 {seg[0]:.1f} us, t>=1 mean {sum(seg[1:]) / (n - 1):.1f} us, min {min(seg[1:]):.1f}")
