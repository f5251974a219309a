"""Per-kernel averages of the SQ counters collected by scripts/dev/sq_counters.sh (rocprofv3 --pmc csv output)."""
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(f"/root/repo/gpurun_out/{tag}/sq*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:58]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen and r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_INSTS_VALU"):
            seen.add(key); calls[(k, r["Counter_Name"])] += 1
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES",
         "SQ_BUSY_CYCLES", "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_ACTIVE_INST_LDS",
         "SQ_ACTIVE_INST_VMEM", "SQ_INST_CYCLES_VMEM", "SQ_WAVES"]
print("kernel".ljust(58), " ".join(n.replace("SQ_", "")[:12].rjust(12) for n in names))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    n1 = max(calls[(k, "SQ_WAVE_CYCLES")], 1); n2 = max(calls[(k, "SQ_INSTS_VALU")], 1)
    print(k.ljust(58), " ".join(f"{v.get(n, 0) / (n1 if i < 8 else n2):12.3g}" for i, n in enumerate(names)))
