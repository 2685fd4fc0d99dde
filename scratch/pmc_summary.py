"""Summarise rocprofv3 --pmc passes (SQ / FETCH_SIZE / WRITE_SIZE) of scratch/pmc_workload.py per kernel:
python scratch/pmc_summary.py <dir_sq> <dir_fetch> <dir_write> <out.json>.  FETCH_SIZE is doubled (gfx950 reports
half the bytes of wide coalesced reads, MI355X_MICROARCH.md section HBM); sizes are KB in the counter."""
import collections, csv, glob, json, sys

def load(d):
    return list(csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])))

sq, fe, wr = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
tr = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
dur, calls = collections.defaultdict(float), collections.Counter()
for r in tr:
    dur[r["Kernel_Name"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    calls[r["Kernel_Name"]] += 1
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for rows in (sq, fe, wr):
    for r in rows:
        agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
ncall = collections.defaultdict(lambda: collections.Counter())
for rows in (sq, fe, wr):
    for r in rows:
        ncall[r["Kernel_Name"]][r["Counter_Name"]] += 1
out = {}
for k, v in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
    n = lambda c: max(ncall[k][c], 1)
    wc = max(v["SQ_WAVE_CYCLES"], 1)
    out[k] = dict(calls=calls[k], us_per_call_under_pmc=dur[k] / max(calls[k], 1),
                  hbm_read_MB_per_call=2 * v["FETCH_SIZE"] / n("FETCH_SIZE") / 1024,
                  hbm_write_MB_per_call=v["WRITE_SIZE"] / n("WRITE_SIZE") / 1024,
                  l2_hit_rate=v["TCC_HIT_sum"] / max(v["TCC_HIT_sum"] + v["TCC_MISS_sum"], 1),
                  lds_bank_conflict_ratio=v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1),
                  wait_any_frac=v["SQ_WAIT_ANY"] / wc, wait_inst_frac=v["SQ_WAIT_INST_ANY"] / wc, active_inst_frac=v["SQ_ACTIVE_INST_ANY"] / wc)
json.dump(out, open(sys.argv[4], "w"), indent=1)
for k, v in list(out.items())[:24]:
    print(f"{k[:64]:64s} {v['calls']:3d} rd {v['hbm_read_MB_per_call']:8.1f} MB wr {v['hbm_write_MB_per_call']:8.1f} MB L2 {100*v['l2_hit_rate']:5.1f}% ldsconf {v['lds_bank_conflict_ratio']:.3f} wait {100*v['wait_any_frac']:4.1f}% act {100*v['active_inst_frac']:4.1f}%")
