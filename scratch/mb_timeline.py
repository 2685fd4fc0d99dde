"""One minibatch update's kernel sequence from a rocprofv3 --kernel-trace csv: everything between two consecutive adam kernels on the main stream
(start offset, duration, gap before), plus the sum of kernels shorter than 40 us."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
a, b = ad[-3], ad[-2]
sid = rows[a]["Stream_Id"] if "Stream_Id" in rows[a] else None
seq = [r for r in rows[a:b + 1] if sid is None or r["Stream_Id"] == sid]
base = int(seq[0]["End_Timestamp"]); prev = base
small = 0.0; nsmall = 0; gaps = 0.0
for r in seq[1:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = (e - s) / 1e3
    if d < 40: small += d; nsmall += 1
    gaps += (s - prev) / 1e3
    print(f"{(s - base) / 1e3:8.1f} us +{d:7.1f} gap {(s - prev) / 1e3:5.1f}  {r['Kernel_Name'][:90]}")
    prev = e
print(f"span {(prev - base) / 1e3:.1f} us; {nsmall} kernels < 40 us: {small:.1f} us; gaps {gaps:.1f} us")
