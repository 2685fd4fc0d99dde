"""Kernel timeline of ONE env group's chain in the 4-group rollout (rocprofv3 --kernel-trace csv): picks the stream with heads_sample launches,
prints start offsets / durations between two consecutive heads_sample kernels of that stream, averaged gaps over many steps."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
key = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
by = collections.defaultdict(list)
for r in rows:
    by[r[key]].append(r)
streams = [(k, v) for k, v in by.items() if sum(1 for r in v if r["Kernel_Name"].startswith("heads_sample")) > 200]
print("streams with a rollout chain:", [(k, len(v)) for k, v in streams], "key", key)
k, v = streams[0]
v.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(v) if r["Kernel_Name"].startswith("heads_sample")]
a, b = idx[-150], idx[-149]
base = int(v[a]["End_Timestamp"])
for r in v[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - base) / 1e3:8.1f} us +{(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:70]}")
# averages over the last 200 steps: per kernel name duration and gap to the previous kernel's end
dur, gap, cnt = collections.Counter(), collections.Counter(), collections.Counter()
for i in range(idx[-201] + 1, idx[-1] + 1):
    r, p = v[i], v[i - 1]
    n = r["Kernel_Name"][:40]
    dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    gap[n] += (int(r["Start_Timestamp"]) - int(p["End_Timestamp"])) / 1e3
    cnt[n] += 1
print("kernel                                     n    avg_us   avg_gap_before_us")
for n in cnt:
    print(f"{n:40s} {cnt[n]:5d} {dur[n] / cnt[n]:8.1f} {gap[n] / cnt[n]:8.1f}")
print("step period us", (int(v[idx[-1]]["End_Timestamp"]) - int(v[idx[-201]]["End_Timestamp"])) / 200 / 1e3)
