"""Per-step kernel table of a profile tag (profiles/<tag>_kernel_stats.csv), steps = calls of adam_kernel."""
import csv, sys
tag = sys.argv[1]
rows = list(csv.DictReader(open(f"/root/repo/profiles/{tag}_kernel_stats.csv")))
steps = max(int(r["Calls"]) for r in rows if r["Name"].startswith(("adam_kernel", "adam_at_kernel")))
tot = 0.0
for r in rows:
    c = int(r["Calls"])
    if c < steps // 2:
        continue
    per = int(r["TotalDurationNs"]) / 1e3 / steps
    tot += per
    print(f"{r['Name'][:100]:100s} {c/steps:5.1f} {float(r['AverageNs'])/1e3:8.1f} {per:8.1f}")
print("steps", steps, "sum us/step", round(tot, 1))
