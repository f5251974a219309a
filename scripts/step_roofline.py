"""Per-kernel roofline table of one Product2Vec step from a rocprofv3 kernel trace (profiles/<tag>_kernel_stats.json)
and the shapes of the benchmark (profiles/<tag>_bench.json): algorithmic FLOPs and HBM bytes per launch, the achieved
rates and their fractions of the two roofs (dense bf16 MFMA / 6 products = 416.7 TFLOP/s fp32-equivalent; 8 TB/s).
    python scripts/step_roofline.py r03c  ->  profiles/archive/r03c_step_roofline.md"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
ks = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.json")))
bench = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench.json")))
cfgtxt = bench["config"]["neighbour_rows"]
nbc = int(cfgtxt.split("avg ")[1].split(" ")[0]) + 1
B = bench["config"]["global_batch"]
R = 7 * B + nbc
D, H = 128, 256
MF, HBM = 416.7e12, 8e12
# kernel-name fragment -> (what, rows, flops/row, bytes/row)
rows = [
    ("gemm_nt_kernel<2, 2, 16, 3, false, 0, 1>", "Linear0 + BN sums (gather x -> H0)", R, 2 * D * H, 4 * (D + H)),
    ("gemm_nt_kernel<2, 4, 16, 2, true, 1, 0>", "BN+tanh -> Linear3 -> tanh (H0 -> A2, A1 saved)", R, 2 * H * H, 4 * (H + H + H)),
    ("gemm_nt_kernel<2, 2, 16, 3, false, 0, 0>", "Linear5 / K|V projection / dKeys (3 launches, mean)", (R + 2 * nbc) / 3, 2 * D * H, 4 * (D + H)),
    ("gemm_nt_kernel<2, 2, 16, 3, false, 3, 0>", "dZ2 = dY W5 * tanh' (aux A2)", R, 2 * D * H, 4 * (D + H + H)),
    ("gemm_nt_kernel<2, 2, 16, 3, false, 4, 2>", "dZ1 = dZ2 W3 * tanh'(BN) + BN-bwd sums (aux H0)", R, 2 * H * H, 4 * (H + H + H)),
    ("gemm_tn8_kernel<2, 4, 4, 2, 32, 2, false, false>", "dW3 = dZ2^T A1 (whole 256 x 256 tile: builds before the split into halves)", R, 2 * H * H, 4 * (H + H)),
    ("gemm_tn8_kernel<4, 2, 2, 2, 16, 3, false, true>", "dW0 = BNbwd(dZ1)^T x (aux H0, gathered x)", R, 2 * D * H, 4 * (H + H + D)),
    ("gemm_tn8_kernel<2, 4, 2, 2, 32, 2, false, false>", "dW5 = dY^T A2 and the two halves of dW3 = dZ2^T A1 (mean of 3)", R, 2 * D * H, 4 * (D + H + (H - D) * 2 / 3)),
    ("gemm_tn8_kernel<4, 2, 2, 2, 32, 2, false, false>", "dW_kv = dKV^T keys", nbc, 2 * D * H, 4 * (D + H)),
]
steps = None
for k, v in ks.items():
    if k.startswith("adam_kernel"):
        steps = v["calls"]
out = [f"# Product2Vec step, per-kernel roofline ({tag}: B = {B}, FFN rows R = {R}, distinct neighbour rows = {nbc - 1})", "",
       "Algorithmic work per launch (operands once, weights L2-resident); peaks: 416.7 TFLOP/s fp32-equivalent (2.5 PFLOP/s dense bf16 / 6 products), 8 TB/s.", "",
       "| kernel | what | launches/step | avg µs | GFLOP | MB | TFLOP/s-eq | frac MFMA | TB/s | frac HBM |", "|---|---|---|---|---|---|---|---|---|---|"]
tot = 0.0
for name, what, r, fpr, bpr in rows:
    v = next((ks[k] for k in ks if k.startswith(name)), None)
    if v is None:
        continue
    us = v["avg_us"]
    n = v["calls"] / steps
    fl, by = r * fpr, r * bpr
    tot += us * n
    out.append(f"| `{name}` | {what} | {n:.0f} | {us:.1f} | {fl / 1e9:.2f} | {by / 1e6:.0f} | {fl / us / 1e6:.0f} | {fl / (us * 1e-6) / MF:.2f} | {by / us / 1e6:.2f} | {by / (us * 1e-6) / HBM:.2f} |")
other = [(k, v) for k, v in ks.items() if k != "_summary" and not any(k.startswith(n) for n, *_ in rows)]
other.sort(key=lambda kv: -kv[1]["avg_us"] * kv[1]["calls"])
out += ["", f"The nine large products above: {tot:.0f} µs of the step.  Everything else, by time per step:", "",
        "| kernel | launches/step | avg µs | µs/step |", "|---|---|---|---|"]
for k, v in other[:22]:
    out.append(f"| `{k[:70]}` | {v['calls'] / steps:.1f} | {v['avg_us']:.1f} | {v['avg_us'] * v['calls'] / steps:.1f} |")
out.append("")
out.append(f"(loader kernels -- `build_pairs_negatives`, `uq_*`, copies -- run on a side stream one batch ahead; bench line: {bench['ms_per_step']} ms/step)")
open(os.path.join(ROOT, "profiles", f"{tag}_step_roofline.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:20]))
