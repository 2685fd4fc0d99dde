"""Pipe activity per kernel from two rocprofv3 --pmc passes of scratch/pmc_workload.py (see the gpurun line in DESIGN.md):
python scratch/pmc_pipes.py <dir_pass1> <dir_pass2> <out.json>"""
import collections, csv, glob, json, sys
def load(d):
    return list(csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.Counter())
for d in sys.argv[1:3]:
    for r in load(d):
        agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]][r["Counter_Name"]] += 1
out = {}
for k, v in agg.items():
    per = {c: v[c] / max(cnt[k][c], 1) for c in v}
    busy = max(per.get("SQ_BUSY_CU_CYCLES", 0.0), 1.0)          # summed over CUs
    out[k] = dict(calls=max(cnt[k].values()),
                  valu_active=per.get("SQ_ACTIVE_INST_VALU", 0) / busy, lds_active=per.get("SQ_ACTIVE_INST_LDS", 0) / busy,
                  vmem_active=per.get("SQ_ACTIVE_INST_VMEM", 0) / busy, mfma_busy=per.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / busy,
                  lds_idx_active=per.get("SQ_LDS_IDX_ACTIVE", 0) / busy, lds_bank_conflict=per.get("SQ_LDS_BANK_CONFLICT", 0) / busy,
                  insts_valu=per.get("SQ_INSTS_VALU", 0), insts_mfma=per.get("SQ_INSTS_MFMA", 0), insts_lds=per.get("SQ_INSTS_LDS", 0),
                  insts_vmem=per.get("SQ_INSTS_VMEM", 0), insts_salu=per.get("SQ_INSTS_SALU", 0), wait_inst_lds=per.get("SQ_WAIT_INST_LDS", 0) / max(per.get("SQ_WAVE_CYCLES", 1), 1),
                  busy_cu_cycles=busy, wave_cycles=per.get("SQ_WAVE_CYCLES", 0))
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["busy_cu_cycles"])[:22]:
    print(f"{k[:58]:58s} valu {v['valu_active']:5.2f} lds {v['lds_active']:5.2f} vmem {v['vmem_active']:5.2f} mfma {v['mfma_busy']:5.2f} ldsidx {v['lds_idx_active']:5.2f} conf {v['lds_bank_conflict']:5.2f} | VALU {v['insts_valu']/1e6:7.1f}M MFMA {v['insts_mfma']/1e6:6.2f}M LDS {v['insts_lds']/1e6:6.1f}M")
