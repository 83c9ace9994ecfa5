#!/usr/bin/env python3
"""Where a unit of the RoIAlign forward spends its time (diagnostic build with profiles/r06_roi_fwd_bound/stamps.patch applied,
tools/build_variant.sh stamps roi_align.hip, swapped in as vision_amd/_lib/libtvmi_kernels.so): mean per DMA-path unit of
start -> first DMA pass issued, start -> first pass landed, start -> end (s_memrealtime, 100 MHz), config 2, 7x7 fp32."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, vision_amd, bench
lib = vision_amd._loader.kernels()
assert hasattr(lib, "tvmi_debug_roi_stamps"), "not the stamps build"
lib.tvmi_debug_roi_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
feats, boxes, scores = bench.make_inputs(dev, 1000)
shapes = [(800, 1344)] * 4
pool = vision_amd.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
buf = np.zeros(4096, dtype=np.uint64)
for fold in (1, 0):
    torch.ops.tvmi.set_option("roi_align.fold_order", fold)
    with torch.no_grad():
        for _ in range(3):
            pool(feats, boxes, shapes)
        torch.cuda.synchronize()
        lib.tvmi_debug_roi_stamps(None, 1)
        n = 10
        for _ in range(n):
            pool(feats, boxes, shapes)
        torch.cuda.synchronize()
    assert lib.tvmi_debug_roi_stamps(buf.ctypes.data, 1) == 0
    s = buf.reshape(1024, 4).astype(np.float64)
    units = s[:, 3].sum()
    iss, land, end = (s[:, i].sum() / units * 0.01 for i in range(3))
    per_slot = s[:, 2] * 0.01 / n   # busy us per wave slot and launch
    print(f"fold_order={fold}: {units / n:.0f} DMA-path units per launch; per unit (us): start->first pass issued {iss:.2f}, "
          f"->landed {land:.2f} (first-pass latency {land - iss:.2f}), ->end {end:.2f}; no DMA in flight for the unit's own work: "
          f"{land / end * 100:.1f} % of its time; busy time per wave slot {per_slot.mean():.1f} us (min {per_slot.min():.1f}, max {per_slot.max():.1f})")
