"""Turn one `scripts/profile_round.sh <tag>` run (gpurun_out/<tag>/) into the committed artefacts under profiles/:
  <tag>_bench.json, <tag>_bench_under_rocprof.json      the bench lines (plain, and under the kernel trace)
  <tag>_kernel_stats.csv / .json                        rocprofv3 --kernel-trace --stats per-kernel summary
  <tag>_pmc_traffic.json                                HBM bytes per kernel from the FETCH_SIZE / WRITE_SIZE passes
PMC units follow /opt/skills/guides/MI355X_MICROARCH.md: both counters are in KiB-like 1 KiB units on this stack
(value x 1024 = bytes) and FETCH_SIZE under-reports by 2x on gfx950 (doubled here)."""
import csv, glob, json, os, re, shutil, sqlite3, sys
from collections import defaultdict

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
for name in ("bench.json", "bench_under_rocprof.json", "bench_joint_under_rocprof.json", "bench_big.json", "bench_big_under_rocprof.json",
             "bench_cfg3_under_rocprof.json", "bench_joint34800d_under_rocprof.json", "bench_joint34800_under_rocprof.json"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))

def short(n):
    n = re.sub(r"\(.*$", "", n)
    return n.replace("void ", "").strip()

# ---- kernel trace
def kernel_stats(subdir, suffix):
    dbs = glob.glob(os.path.join(src, subdir, "*_results.db"))
    if not dbs:
        return
    db = sqlite3.connect(dbs[0]); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    namecol = "display_name" if "display_name" in cols else "kernel_name"
    rows = cur.execute(f"select s.{namecol}, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                       f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.{namecol} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(os.path.join(dst, f"{tag}{suffix}_kernel_stats.csv"), "w", newline="") as f:
        wr = csv.writer(f); wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows: wr.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", f"{100*r[2]/tot:.2f}", r[4], r[5]])
    js = {short(r[0]) if r[0].startswith("void") or "(" in r[0] else r[0]: {"calls": r[1], "avg_us": round(r[3] / 1e3, 2), "share": round(r[2] / tot, 4)} for r in rows}
    nt = [(r[1], r[2]) for r in rows if "gemm_nt_kernel<" in r[0] or "gemm_nt_kernelI" in r[0]]; tn = [(r[1], r[2]) for r in rows if "gemm_tn" in r[0]]
    js["_summary"] = {"total_kernel_us": round(tot / 1e3, 1),
                      "gemm_nt_avg_us": round(sum(t for _, t in nt) / max(1, sum(c for c, _ in nt)) / 1e3, 2),
                      "gemm_nt_share": round(sum(t for _, t in nt) / tot, 4),
                      "gemm_tn_avg_us": round(sum(t for _, t in tn) / max(1, sum(c for c, _ in tn)) / 1e3, 2),
                      "gemm_tn_share": round(sum(t for _, t in tn) / tot, 4)}
    if suffix.startswith("_joint"):
        # kernels per step of the joint loop: everything launched once per step has the call count of the Adam kernel
        steps = max([r[1] for r in rows if "adam" in r[0] or "joint_finish" in r[0]] or [1])
        per_step = {short(r[0]): round(r[1] / steps, 2) for r in rows if r[1] >= steps // 2}
        js["_summary"].update({"steps_traced": steps, "kernel_us_per_step": round(tot / 1e3 / steps, 2),
                               "launches_per_step": round(sum(r[1] for r in rows if r[1] >= steps // 2) / steps, 2),
                               "per_step_launch_counts": per_step})
    json.dump(js, open(os.path.join(dst, f"{tag}{suffix}_kernel_stats.json"), "w"), indent=1)
    print(f"kernel stats{suffix}:", js["_summary"])

kernel_stats("prof", "")
kernel_stats("prof_joint", "_joint")
kernel_stats("prof_big", "_big")
kernel_stats("prof_cfg3", "_cfg3")
kernel_stats("prof_joint34800d", "_joint34800d")
kernel_stats("prof_joint34800", "_joint34800")

# ---- PMC passes
def pmc(dirname, counter):
    out = defaultdict(lambda: [0, 0.0])
    for fn in glob.glob(os.path.join(src, dirname, "*", "*counter_collection.csv")):
        for row in csv.DictReader(open(fn)):
            if row["Counter_Name"] != counter: continue
            k = short(row["Kernel_Name"])
            out[k][0] += 1; out[k][1] += float(row["Counter_Value"])
    return out
fe, wr_ = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
if fe:
    res = {}
    for k, (n, v) in fe.items():
        w = wr_.get(k, [n, 0.0])
        res[k] = {"launches": n, "fetch_kib_raw": round(v / n, 1), "fetch_bytes_corrected": round(v / n * 1024 * 2),
                  "write_bytes": round(w[1] / max(1, w[0]) * 1024)}
    # the whole step: every kernel launched (about) once per step or more, over the steps run (the step's first and last launches
    # are once-per-step kernels); one-time kernels -- catalogue generators, epoch plans, fills -- stay out of the sum
    steps = max([v["launches"] for k, v in res.items() if "adam" in k or "p2v_prologue" in k or "tn_reduce_group" in k] or [1])
    per_step = [v for v in res.values() if v["launches"] >= steps // 2]
    res["_step_total"] = {"launches": 1, "steps_counted": steps,
                          "fetch_bytes_corrected": round(sum(v["fetch_bytes_corrected"] * v["launches"] for v in per_step) / steps),
                          "write_bytes": round(sum(v["write_bytes"] * v["launches"] for v in per_step) / steps),
                          "note": "the kernels launched at least every other step (loader, step, Adam) summed and divided by the steps run"}
    json.dump(res, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    nt = [v for k, v in res.items() if "gemm_nt_kernel<" in k]
    n = sum(v["launches"] for v in nt)
    print("gemm_nt HBM bytes/launch:", round(sum((v["fetch_bytes_corrected"] + v["write_bytes"]) * v["launches"] for v in nt) / n))

# ---- PMC passes of the joint phase: per kernel, and summed over one step (`_step_total`); the same for the run at the
# reference's NUM_TYPES = 34800 (profiles/<tag>_joint34800_pmc_traffic.json, read by bench.py for that configuration)
for dirs, out_name in ((("pmc_joint_fetch", "pmc_joint_write"), "joint"), (("pmc_joint34800_fetch", "pmc_joint34800_write"), "joint34800"),
                       (("pmc_joint34800d_fetch", "pmc_joint34800d_write"), "joint34800d"),      # d: DROPOUT = 0.1, the reference as shipped
                       (("pmc_jointd_fetch", "pmc_jointd_write"), "jointd"), (("pmc_p2vd_fetch", "pmc_p2vd_write"), "p2vd")):
    fe, wr_ = pmc(dirs[0], "FETCH_SIZE"), pmc(dirs[1], "WRITE_SIZE")
    if not fe:
        continue
    res = {}
    for k, (n, v) in fe.items():
        w = wr_.get(k, [n, 0.0])
        res[k] = {"launches": n, "fetch_kib_raw": round(v / n, 1), "fetch_bytes_corrected": round(v / n * 1024 * 2),
                  "write_bytes": round(w[1] / max(1, w[0]) * 1024)}
    steps = max([v["launches"] for k, v in res.items() if "adam" in k or "joint_finish" in k or "p2v_prologue" in k or "tn_reduce_group" in k] or [1])
    per_step = [v for v in res.values() if v["launches"] >= steps // 2]
    tot_f = sum(v["fetch_bytes_corrected"] * v["launches"] for v in per_step) / steps
    tot_w = sum(v["write_bytes"] * v["launches"] for v in per_step) / steps
    res["_step_total"] = {"launches": 1, "steps_counted": steps, "fetch_bytes_corrected": round(tot_f), "write_bytes": round(tot_w),
                          "note": "the kernels launched at least every other step (batch construction, step, Adam) summed and divided by the steps run"}
    json.dump(res, open(os.path.join(dst, f"{tag}_{out_name}_pmc_traffic.json"), "w"), indent=1)
    print(out_name, "HBM bytes/step:", round(tot_f + tot_w))


# ---- BASELINE configs[4] on one GPU (BIG=1): per-kernel HBM traffic and L2 hit rates, Zipf vs uniform negatives
def pmc_multi(dirname, counters):
    out = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for fn in glob.glob(os.path.join(src, dirname, "*", "*counter_collection.csv")):
        for row in csv.DictReader(open(fn)):
            if row["Counter_Name"] not in counters: continue
            k = short(row["Kernel_Name"])
            out[k][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] == counters[0]: cnt[k] += 1
    return out, cnt
fe, wr_ = pmc("pmc_big_fetch", "FETCH_SIZE"), pmc("pmc_big_write", "WRITE_SIZE")
if fe:
    l2, l2n = pmc_multi("pmc_big_l2", ["TCC_HIT_sum", "TCC_MISS_sum"])
    feu = pmc("pmc_bigu_fetch", "FETCH_SIZE")
    l2u, _ = pmc_multi("pmc_bigu_l2", ["TCC_HIT_sum", "TCC_MISS_sum"])
    res = {}
    for k, (n, v) in fe.items():
        w = wr_.get(k, [n, 0.0])
        hit = lambda d: (round(d[k]["TCC_HIT_sum"] / max(1.0, d[k]["TCC_HIT_sum"] + d[k]["TCC_MISS_sum"]), 4) if k in d else None)
        res[k] = {"launches": n, "fetch_bytes_corrected": round(v / n * 1024 * 2), "write_bytes": round(w[1] / max(1, w[0]) * 1024),
                  "l2_hit_rate": hit(l2),
                  "uniform_negatives": {"fetch_bytes_corrected": round(feu[k][1] / max(1, feu[k][0]) * 1024 * 2) if k in feu else None,
                                        "l2_hit_rate": hit(l2u)}}
    json.dump(res, open(os.path.join(dst, f"{tag}_big_pmc_traffic.json"), "w"), indent=1)
    for k, v in res.items():
        if "gemm_nt_kernel" in k or "gemm_tn8" in k:
            print("big:", k[:60], v)


# ---- BASELINE configs[3] on one GPU (CFG3=1): per-kernel HBM traffic of the 10 M-product sharded chain
fe, wr_ = pmc("pmc_cfg3_fetch", "FETCH_SIZE"), pmc("pmc_cfg3_write", "WRITE_SIZE")
if fe:
    res = {}
    for k, (n, v) in fe.items():
        w = wr_.get(k, [n, 0.0])
        res[k] = {"launches": n, "fetch_bytes_corrected": round(v / n * 1024 * 2), "write_bytes": round(w[1] / max(1, w[0]) * 1024)}
    json.dump(res, open(os.path.join(dst, f"{tag}_cfg3_pmc_traffic.json"), "w"), indent=1)
    nt = [v for k, v in res.items() if "gemm_nt_kernel<" in k]
    n = sum(v["launches"] for v in nt)
    print("cfg3 gemm_nt HBM bytes/launch:", round(sum((v["fetch_bytes_corrected"] + v["write_bytes"]) * v["launches"] for v in nt) / max(n, 1)))
