"""Per-step critical-path table from a timestamped rocprofv3 --kernel-trace of bench.py's Product2Vec leg.

    python scripts/step_timeline.py gpurun_out/<tag>/prof/<tag>_results.db profiles/<name>.md [step]

For one steady-state step (delimited by the launch that ends the previous one: Adam, or the slab reduce it rides in) every dispatch with its queue, start and
duration, the gap to its predecessor ON THE SAME QUEUE (end -> start: launch boundary / dependency wait) and, for the queue that
carries the step (the main queue), whether another queue's kernel was running beside it.  Then the sums: kernel time on the
main queue, gaps on the main queue, what ran on side queues, and the median over all traced steps of the step's span."""
import re
import sqlite3
import statistics
import sys


def short(n):
    n = re.sub(r"\.kd$", "", n)
    m = re.match(r"_Z\d+([a-z0-9_]+kernel)", n)
    base = m.group(1) if m else re.sub(r"\(.*$", "", n).replace("void ", "")
    t = re.search(r"kernelI(.*?)E[v]", n)
    if t:
        args = re.findall(r"L[ib](\d+)E", t.group(1))
        base += "<" + ",".join(args) + ">"
    return base[:64]


def main():
    dbp, out = sys.argv[1], sys.argv[2]
    db = sqlite3.connect(dbp)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.queue_id, d.grid_size_x, d.workgroup_size_x from {kd} d "
                       f"join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
    # a step ends with its Adam launch -- or, since the optimizer rides in the slab reduce (round 6), with that launch
    ends = [i for i, r in enumerate(rows) if "adam_at_kernel" in r[0] or "adam_kernel" in r[0]]
    if len(ends) < 8:
        ends = [i for i, r in enumerate(rows) if "tn_reduce_group_kernel" in r[0]]
    if len(ends) < 8:
        raise SystemExit("fewer than 8 step-ending launches in the trace")
    # the loader's builders (uq_* kernels, copies) run on their own queue, several steps ahead: attributed to the step whose
    # span they fall in
    # (only the fused index step's iterations: a trace may also hold the dense drop-in leg, whose autograd path ends in the same
    # reduce kernel but never launches the prologue)
    # (... and, with the loader concatenating the step's rows, no prologue at all: then every step has the FFN's finalize)
    marker = "p2v_prologue_kernel" if any("p2v_prologue_kernel" in r[0] for r in rows) else "bn_finalize_fwd_kernel"
    keep = [(a, b) for a, b in zip(ends[:-1], ends[1:]) if any(marker in r[0] for r in rows[a + 1:b + 1])]
    ends = [keep[0][0]] + [b for _, b in keep] if keep else ends
    steps = []
    for a, b in zip(ends[:-1], ends[1:]):
        steps.append(rows[a + 1:b + 1])
    pick = int(sys.argv[3]) if len(sys.argv) > 3 else len(steps) * 2 // 3
    st = steps[pick]
    step_q = rows[ends[pick + 1]][3]                         # the queue of the Adam launch = the step's main queue
    t0 = min(r[1] for r in st if r[3] == step_q)
    t_prev_end = rows[ends[pick]][2]
    lines = []
    last_end = {step_q: t_prev_end}
    main_k = main_gap = side_k = 0.0
    n_main = n_side = 0
    for r in st:
        name, s_, e_, q = short(r[0]), r[1], r[2], r[3]
        gap = (s_ - last_end[q]) / 1e3 if q in last_end else float("nan")
        last_end[q] = max(last_end.get(q, 0), e_)
        beside = [short(o[0]) for o in st if o[3] != q and o[1] < e_ and o[2] > s_]
        tag = "main" if q == step_q else f"side{q}"
        if q == step_q:
            main_k += (e_ - s_) / 1e3
            main_gap += max(gap, 0.0)
            n_main += 1
        else:
            side_k += (e_ - s_) / 1e3
            n_side += 1
        lines.append(f"| {tag} | {name} | {r[4] // max(r[5], 1)} x {r[5]} | {(s_ - t0) / 1e3:8.1f} | {(e_ - s_) / 1e3:7.1f} | "
                     f"{gap:6.1f} | {', '.join(sorted(set(beside)))[:80]} |")
    span = (rows[ends[pick + 1]][2] - t_prev_end) / 1e3
    spans = [(rows[b][2] - rows[a][2]) / 1e3 for a, b in zip(ends[:-1], ends[1:])]
    spans_steady = spans[len(spans) // 4:]
    per_step_counts = [len([r for r in s if r[3] == rows[ends[0]][3]]) for s in steps]
    with open(out, "w") as f:
        f.write(f"# Product2Vec step timeline ({dbp.split('/')[-1]}, traced step {pick} of {len(steps)})\n\n")
        f.write("`rocprofv3 --kernel-trace` timestamps (the profiler serialises nothing but adds ~1 us per dispatch).  "
                "gap = start minus the end of the previous dispatch on the SAME queue.\n\n")
        f.write("| queue | kernel | grid | start us | dur us | gap us | running beside it (other queues) |\n|---|---|---|---|---|---|---|\n")
        f.write("\n".join(lines) + "\n\n")
        f.write(f"* step span (previous Adam end -> this Adam end): **{span:.1f} us**; median over the steady steps of the trace: "
                f"**{statistics.median(spans_steady):.1f} us** (min {min(spans_steady):.1f}, max {max(spans_steady):.1f})\n")
        f.write(f"* main queue: {n_main} launches, {main_k:.1f} us of kernels + {main_gap:.1f} us of gaps "
                f"(= {main_gap / max(n_main, 1):.2f} us per launch boundary)\n")
        f.write(f"* side queues: {n_side} launches, {side_k:.1f} us of kernels (overlapping the main queue's)\n")
        f.write(f"* launches per step on the main queue over the trace: median {statistics.median(per_step_counts)}\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
