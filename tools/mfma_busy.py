#!/usr/bin/env python3
"""tools/mfma_busy.sh's summary: per deform_conv2d call shape, the fraction of the chip's MFMA issue capacity in use.

  mfma_busy_frac = sum over the call's kernels of SQ_VALU_MFMA_BUSY_CYCLES            (cycles a SIMD's matrix pipe was busy, summed over SIMDs)
                   / (1024 SIMDs x sum over the same kernels of GRBM_GUI_ACTIVE / 8)  (GRBM_GUI_ACTIVE is summed over the 8 XCDs)

i.e. busy SIMD-cycles over available SIMD-cycles while the call's kernels run (the guide: SQ_VALU_MFMA_BUSY_CYCLES counts
cycles, MI355X_MICROARCH.md "Per-instruction cycle constants").  Kernels of other namespaces (torch fills etc.) are ignored."""
import collections
import csv
import glob
import json
import os
import sys

ROWS = {"dcn": "deform_conv2d_g1_fp32", "dcn_bf16": "deform_conv2d_g1_bf16", "dcn_bwd": "deform_conv2d_backward_g1_fp32",
        "dcn_bwd_bf16": "deform_conv2d_backward_g1_bf16"}
SIMDS, XCDS = 1024, 8


def main(root, out_path):
    rows = {}
    for which, key in ROWS.items():
        files = glob.glob(os.path.join(root, which, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        calls = collections.Counter()
        for f in files:
            seen = set()
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "tvmi::" not in k:
                    continue
                per[k][r["Counter_Name"]] += float(r["Counter_Value"])
                did = r.get("Dispatch_Id")
                if (k, did) not in seen:
                    seen.add((k, did))
                    calls[k] += 1
        busy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for v in per.values())
        active = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for v in per.values())
        if active <= 0:
            continue
        rows[key] = {
            "mfma_busy_frac": round(busy / (SIMDS * active / XCDS), 4),
            "source": f"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (tools/mfma_busy.sh, run_kernel.py {which})",
            "kernels": {k[:90]: {"launches": calls[k], "mfma_busy_frac": round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (SIMDS * v["GRBM_GUI_ACTIVE"] / XCDS), 4),
                                 "share_of_gpu_cycles": round(v["GRBM_GUI_ACTIVE"] / active, 3),
                                 "waves_waiting_frac": (round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 3) if v.get("SQ_WAVE_CYCLES") else None)}
                        for k, v in per.items() if v.get("GRBM_GUI_ACTIVE", 0) > 0},
        }
    json.dump({"how": __doc__.split("\n\n")[1], "rows": rows}, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
