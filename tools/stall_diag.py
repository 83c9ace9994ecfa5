#!/usr/bin/env python3
"""tools/stall_diag.py — root cause of the one-off 38-47 ms step of the round-4 contract line.

Replays the ROUND-4 timed loop of bench.py at the driver's flags (5 warm-up steps whose result is dropped, then 20 steps
that hold `out` across the next step and record an event) with three probes per step:
  * host wall time of the step call itself (perf_counter around step()),
  * CPython garbage collections (gc.callbacks: generation + duration),
  * caching-allocator growth (torch.cuda.memory_stats()["num_device_alloc"]: a hipMalloc of a new segment),
and prints the UNSORTED per-step event times with the index of the maximum.  Then the same loop in the three
candidate-fix configurations: gc frozen + disabled; warm-up identical to the timed loop; both.

    python tools/stall_diag.py [--steps 20 --warmup 5]
"""
import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (make_inputs + constants only)


def run(label, steps, warmup, freeze_gc, same_warmup, device, state):
    import vision_amd
    from vision_amd import sharding

    sets, pool, img_idx = state
    image_shapes = [(bench.IMG_H, bench.IMG_W)] * bench.BATCH
    counter = {"i": 0}

    def step():
        with torch.no_grad():
            d = sets[counter["i"] % bench.N_SETS]
            counter["i"] += 1
            keep, num = vision_amd.boxes.batched_nms_padded(d["all_boxes"], d["all_scores"], img_idx, bench.NMS_THR, bench.BATCH)
            payload = sharding.pack_kept_payload(d["all_boxes"], d["all_scores"], img_idx, keep, num, bench.BATCH, bench.MAX_DETS)
            pooled = pool(d["feats"], d["boxes"], image_shapes)
            gd, gcnt = sharding.all_gather_payload(payload, bench.MAX_DETS)
        return pooled, num, gd, gcnt

    gc_log = []
    t_gc = {}

    def cb(phase, info):
        if phase == "start":
            t_gc["t"] = time.perf_counter()
        else:
            gc_log.append((info["generation"], (time.perf_counter() - t_gc["t"]) * 1e3, counter["i"]))

    torch.cuda.synchronize()
    torch.cuda.empty_cache()          # every variant starts from the same allocator state as a fresh process would
    gc.callbacks.append(cb)
    if freeze_gc:
        gc.collect()
        gc.freeze()
        gc.disable()
    out = None
    wmarks = [torch.cuda.Event(enable_timing=True) for _ in range(warmup + 1)]
    if same_warmup:
        wmarks[0].record()
        for i in range(warmup):
            out = step()
            wmarks[i + 1].record()
    else:
        for _ in range(warmup):
            step()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    host, segs = [], []
    seg0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        h0 = time.perf_counter()
        out = step()
        marks[i + 1].record()
        host.append((time.perf_counter() - h0) * 1e3)
        segs.append(torch.cuda.memory_stats().get("num_device_alloc", 0) - seg0)
    torch.cuda.synchronize()
    elapsed = (time.perf_counter() - t0) * 1e3
    if freeze_gc:
        gc.enable()
        gc.unfreeze()
    gc.callbacks.remove(cb)
    per = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    am = max(range(steps), key=lambda i: per[i])
    srt = sorted(per)
    rec = {"variant": label, "ms_per_step": round(elapsed / steps, 4), "median": round(srt[steps // 2], 4), "max": round(per[am], 4),
           "argmax_step": am, "events_ms": [round(x, 3) for x in per], "host_ms": [round(x, 3) for x in host],
           "new_device_segments_after_step": segs,
           "gc_collections": [{"gen": g, "ms": round(ms, 3), "during_step": s} for g, ms, s in gc_log if g >= 1 or ms > 0.5]}
    del out
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    import vision_amd

    sets = []
    for i in range(bench.N_SETS):
        feats, boxes, scores = bench.make_inputs(device, seed=1000 + 97 * i)
        sets.append(dict(feats=feats, boxes=boxes, scores=scores, all_boxes=torch.cat(boxes), all_scores=torch.cat(scores)))
    pool = vision_amd.MultiScaleRoIAlign([str(i) for i in range(len(bench.STRIDES))], bench.POOL, bench.SAMPLING)
    img_idx = torch.cat([torch.full((bench.PROPOSALS,), i, device=device, dtype=torch.int64) for i in range(bench.BATCH)])
    state = (sets, pool, img_idx)
    for label, fz, sw in (("r04_loop", False, False), ("gc_frozen", True, False), ("same_warmup", False, True), ("both", True, True),
                          ("r04_loop_again", False, False)):
        print(json.dumps(run(label, args.steps, args.warmup, fz, sw, device, state)), flush=True)


if __name__ == "__main__":
    main()
