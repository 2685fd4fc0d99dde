"""One minibatch update's kernels on BOTH streams (main + side) from a rocprofv3 --kernel-trace csv: everything between two consecutive
adam kernels, start offset / duration / stream ('M' = the adam kernel's stream, 'S' = any other), and how much of the side stream's
kernel time lies under main-stream kernels."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
a, b = ad[-3], ad[-2]
key = "Stream_Id" if "Stream_Id" in rows[a] else "Queue_Id"
sid = rows[a][key]
seq = rows[a + 1:b + 1]
base = int(rows[a]["End_Timestamp"])
main = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seq if r[key] == sid]
side_t = under = 0.0
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tag = "M" if r[key] == sid else "S"
    if tag == "S":
        side_t += (e - s) / 1e3
        under += sum(max(0, min(e, me) - max(s, ms)) for ms, me in main) / 1e3
    print(f"{(s - base) / 1e3:8.1f} us +{(e - s) / 1e3:7.1f}  {tag}  {r['Kernel_Name'][:100]}")
span = (int(rows[b]["End_Timestamp"]) - base) / 1e3
print(f"span {span:.1f} us; main-stream kernels {sum(e - s for s, e in main) / 1e3:.1f} us; side-stream kernels {side_t:.1f} us, {under:.1f} us of them under main-stream kernels")
