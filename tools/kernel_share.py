"""Share of GPU time per kernel family from a rocprofv3 --kernel-trace [--stats] output directory:
ours (namespace tvmi::) vs library kernels (MIOpen / Tensile / CK) vs ATen, and the idle share of the steady-state window.

    python tools/kernel_share.py <dir> [out.json]

Families: tvmi = our HIP kernels; conv_gemm = MIOpen / Tensile (Cijk_*) / composable_kernel / rocBLAS; aten = at::native
elementwise / reduction / indexing / sort / copy kernels; other = the rest.  Idle = 1 - (union of kernel intervals / span) over the
LAST HALF of the trace's dispatches (steady state: model build, first-use compilation and warm-up steps lie in the first half)."""
import csv
import glob
import json
import os
import sys


def family(name):
    n = name.lower()
    if "tvmi::" in name:
        return "tvmi"
    if any(t in n for t in ("miopen", "cijk_", "igemm", "gridwise", "ck::", "ck_tile", "naive_conv", "rocblas", "winograd", "conv_", "gemm", "batched_transpose", "sp3asm", "gfx9_")):
        return "conv_gemm"
    if any(t in n for t in ("at::native", "elementwise", "vectorized", "reduce_kernel", "index", "cunn", "rocprim", "hipcub", "at::cuda", "cat_", "copy")):
        return "aten"
    return "other"


def main(d, out_path=None):
    traces = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not traces:
        sys.exit(f"no *kernel_trace.csv under {d}")
    rows = []
    for r in csv.DictReader(open(traces[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    steady = rows[len(rows) // 2:]
    span = max(e for _, e, _ in steady) - steady[0][0]
    busy, cur_s, cur_e = 0, steady[0][0], steady[0][1]
    for s, e, _ in steady[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    fam, per_kernel = {}, {}
    for s, e, n in steady:
        f = family(n)
        fam[f] = fam.get(f, 0) + (e - s)
        k = per_kernel.setdefault(n, [0, 0])
        k[0] += e - s
        k[1] += 1
    tot = sum(fam.values())
    out = {"window": "last half of the dispatches of the trace (steady state)", "dispatches": len(steady), "span_ms": round(span / 1e6, 3),
           "kernel_time_ms": round(tot / 1e6, 3), "idle_frac_of_span": round(1 - busy / span, 4),
           "share_of_kernel_time": {f: round(v / tot, 4) for f, v in sorted(fam.items(), key=lambda kv: -kv[1])},
           "top_kernels": [{"share": round(v[0] / tot, 4), "calls": v[1], "avg_us": round(v[0] / v[1] / 1e3, 1), "family": family(n), "name": n[:100]}
                           for n, v in sorted(per_kernel.items(), key=lambda kv: -kv[1][0])[:14]],
           "tvmi_kernels": [{"share": round(v[0] / tot, 4), "calls": v[1], "avg_us": round(v[0] / v[1] / 1e3, 1), "name": n[:100]}
                            for n, v in sorted(per_kernel.items(), key=lambda kv: -kv[1][0]) if family(n) == "tvmi"][:16]}
    print(json.dumps(out, indent=1))
    if out_path:
        json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
