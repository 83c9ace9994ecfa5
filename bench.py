#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

metric   : boxes/sec RoIAlign+NMS (1000 proposals per image, 256-channel FPN maps)   [BASELINE.json]
workload : BASELINE config 2 — a batch of 4 padded 800x1344 images per GPU, 4 FPN levels x 256 channels
           (strides 4/8/16/32, fp32), 1000 proposals per image.  One "step" = one pass of the hot path
           over that batch: MultiScaleRoIAlign 7x7 (sampling_ratio 2) of the 4000 proposals, then per-image
           NMS (batched_nms over the image index, IoU 0.5), packing of the padded top-100 detections of every
           image (fixed shape) and — when N > 1 — their one RCCL all-gather.  The per-rank part has no host
           synchronisation, so the launches queue back to back (--graph replays them from captured hipGraphs, one
           per input set).  The NMS + packing does not depend on the RoIAlign output: since round 6 its workgroups ride in
           front of the RoIAlign grid — the step is ONE launch on one stream (`--one-launch 0`: two launches, the NMS one on a
           second HIP stream forked from and joined back into the step's stream; `--one-stream` keeps both on one stream; the
           line carries those times beside `value`: config.two_launches_two_streams_ms_per_step, one_stream_ms_per_step).
inputs   : synthetic (seeded), resident in HBM before the timed region.
scaling  : weak — every rank owns its own batch of images (the path shards over images; no data-path
           collective besides the detection all-gather).  value = boxes processed by ALL ranks / time.

Besides the contract line it reports
  roofline     : dominant kernel (multi-scale RoIAlign forward): algorithmic bytes per launch
                 (= all feature maps + output + rois) / its mean launch time measured
                 with events on the launch stream, against the 8 TB/s HBM peak; `traffic` = HBM bytes
                 per launch from a separate rocprofv3 --pmc pass (profiles/roofline_traffic.json) or null.
  cpu_baseline : the reference's own CPU kernels (oracle/_ref, kind "reference") — or the C port when
                 that library is absent — timed on this host on the same workload, rank 0 at N=1 only.
  parity       : outside the timed region, the pooled output and the per-image keep lists of the timed
                 configuration are compared with what cpu_baseline computed on the same inputs (max abs
                 error against the 1e-4 bar; index lists equal).  A failed parity check exits non-zero.
  config       : also the time of the same step through the reference-schema ops alone (per-level
                 torchvision::roi_align + torchvision::nms), the channels_last kernel, the dense NMS variant
                 — reported beside `value`, never part of it.  Inputs rotate over N_SETS sets so the 256 MiB
                 Infinity Cache cannot carry one step's feature maps into the next.
  config5      : BASELINE config 5 inside the SAME line — Mask R-CNN R50-FPN inference img/s (random init, synthetic
                 3x800x1333 images, box_score_thresh 0.0 so that the post-processing is busy) through the unchanged
                 reference python laid over this library and with the fused vision_amd pieces swapped in, plus the
                 check that both give the same detections; tools/e2e_maskrcnn.py --variant both, ONE fresh process per
                 rank (under N > 1 the ranks form their own RCCL group and all-gather the detections).  --no-e2e skips it.
`--gpus N` without a torchrun environment re-executes itself under `python -m torch.distributed.run` with N ranks; a
rank count that differs from --gpus, or fewer visible GPUs than ranks, is an error (never a silent 1-GPU run).
--e2e prints a SECOND JSON line with all four (variant x score threshold) runs of config 5 in fresh processes.
--dry-run (CPU, gloo) exercises only the launcher / rank plumbing / all-gather and prints a line with "dry_run": true —
used by tests/test_dist.py; it measures nothing.
"""
import argparse
import gc
import json
import math
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
IMG_H, IMG_W, CHANNELS, BATCH, PROPOSALS = 800, 1344, 256, 4, 1000
N_SETS = 4  # rotated input sets
STRIDES = (4, 8, 16, 32)
POOL, SAMPLING, NMS_THR, MAX_DETS = 7, 2, 0.5, 100


def make_inputs(device, seed):
    g = torch.Generator().manual_seed(seed)
    feats = {}
    for i, s in enumerate(STRIDES):
        feats[str(i)] = torch.randn(BATCH, CHANNELS, IMG_H // s, IMG_W // s, generator=g).to(device)
    boxes, scores = [], []
    for _ in range(BATCH):
        # proposal-like boxes: sqrt(area) log-uniform in [16, 640] px (all four FPN levels receive
        # proposals) and aspect ratio log-uniform in [1/3, 3] (RPN anchors use 1:2..2:1)
        xy = torch.rand(PROPOSALS, 2, generator=g) * torch.tensor([IMG_W - 64.0, IMG_H - 64.0])
        side = torch.exp(torch.rand(PROPOSALS, generator=g) * (math.log(640.0) - math.log(16.0)) + math.log(16.0))
        aspect = torch.exp((torch.rand(PROPOSALS, generator=g) * 2 - 1) * math.log(3.0))
        wh = torch.stack([side * aspect.sqrt(), side / aspect.sqrt()], 1)
        x2y2 = torch.minimum(xy + wh, torch.tensor([float(IMG_W), float(IMG_H)]))
        boxes.append(torch.cat([xy, x2y2], 1).to(device))
        scores.append(torch.rand(PROPOSALS, generator=g).to(device))
    return feats, boxes, scores


def algorithmic_bytes(feats, n_boxes):
    inp = sum(f.numel() * f.element_size() for f in feats.values())
    out = n_boxes * CHANNELS * POOL * POOL * 4
    return inp + out + n_boxes * 5 * 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--one-stream", dest="overlap", action="store_false", help="keep the NMS + packing chain of a step on the "
                    "RoIAlign launch's stream; the default runs it on a second HIP stream under that launch (-3..-6 %% per step in "
                    "eight of eight round-4/5 visits once the collector stall was out of the way)")
    ap.set_defaults(overlap=True)
    ap.add_argument("--one-launch", type=int, default=1, help="1 (default): the step is ONE launch — the NMS workgroups ride in front of the "
                    "RoIAlign grid (tvmi::roi_align_boxes_nms_step, round 6); 0: two launches, on two streams (or one with --one-stream)")
    ap.add_argument("--side-priority", type=int, default=-1, help="HIP priority of the second stream (NMS + packing chain): -1 = "
                    "high (default: its short kernels get the wave slots the RoIAlign workgroups free), 0 = normal")
    ap.add_argument("--event-scope", type=int, default=1, help="fork / join events of the two-stream step: 1 = device-scope release "
                    "(default), 0 = system scope, 2 = no fence from the event, -1 = torch's own wait_stream")
    ap.add_argument("--diag-skip", choices=["fork", "join", "both"], default=None, help="DIAGNOSIS ONLY (the line is marked invalid): "
                    "leave out the fork and / or the join of the two streams, to price the hops")
    ap.add_argument("--nms-step-fused", type=int, default=1, help="1 (default): the step's batched NMS + payload as ONE launch "
                    "(tvmi::nms_step); 0: round 5's chain of five launches")
    ap.add_argument("--roi-inline-mop", type=int, default=None, help="roi_align.inline_mop: 1 = the units the LDS-DMA path declines take "
                    "the wave path inside the same launch (no mop-up launch); default: the library's default")
    ap.add_argument("--roi-fold-order", type=int, default=None, help="roi_align.fold_order: 1 = the order pre-pass runs as a workgroup "
                    "of the step's launch (no pre-pass launch on the critical path); default: the library's default")
    ap.add_argument("--e2e", action="store_true", help="after the contract line, also measure BASELINE config 5 (Mask R-CNN R50-FPN "
                    "inference img/s, unchanged reference python on this library, then with the fused vision_amd pieces) and print it "
                    "as a SECOND JSON object; never mixed into `value`")
    ap.add_argument("--no-e2e", action="store_true", help="skip the config-5 block (Mask R-CNN img/s) of the contract line")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (BASELINE configs 3 and 4: NMS 100k, deform_conv2d)")
    ap.add_argument("--dry-run", action="store_true", help="CPU / gloo: launcher, rank plumbing and the all-gather only; measures nothing")
    ap.add_argument("--force-collective", action="store_true", help="world of one: form a 1-rank RCCL group anyway and run the step's "
                    "all-gather as a real collective (exercises the N > 1 code path — RCCL next to the two streams — on a "
                    "1-GPU box; the line says so in config.parallelism)")
    ap.add_argument("--graph", action="store_true", help="replay the per-rank chain from a captured hipGraph (measured: no gain "
                    "over eager sync-free launches on this stack, so off by default)")
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))      # one process per GPU: re-execute under torch.distributed.run
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE); refusing to report a "
                 f"{world}-GPU number as a {args.gpus}-GPU one")
    if args.dry_run:
        return dry_run(rank, world, args)
    if world > 1 or args.force_collective:
        # RCCL prints a version banner through C stdio on fd 1 (flushed at exit, i.e. BEHIND the JSON line): the contract is
        # ONE JSON line on stdout, so everything written to fd 1 from here on goes to stderr, and python's own stdout — the
        # JSON line of rank 0 — keeps the original descriptor
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        sys.stdout = os.fdopen(real_stdout, "w")
    assert torch.cuda.is_available(), "bench.py measures the HIP path; no GPU visible"
    if torch.cuda.device_count() < world:
        sys.exit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible (one process per GPU)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    elif args.force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    device = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(device)

    import vision_amd  # fails loudly if the HIP extension is missing
    from vision_amd import sharding

    # N_SETS independent input sets (1.46 GB of feature maps) are rotated step by step, so that the 256 MiB
    # Infinity Cache cannot serve one step's maps to the next — the traffic is HBM traffic
    sets = []
    for i in range(N_SETS):
        feats, boxes, scores = make_inputs(device, seed=1000 + rank + 97 * i)
        sets.append(dict(feats=feats, boxes=boxes, scores=scores, all_boxes=torch.cat(boxes), all_scores=torch.cat(scores)))
    feats, boxes, scores = sets[0]["feats"], sets[0]["boxes"], sets[0]["scores"]
    pool = vision_amd.MultiScaleRoIAlign([str(i) for i in range(len(STRIDES))], POOL, SAMPLING)
    image_shapes = [(IMG_H, IMG_W)] * BATCH
    img_idx = torch.cat([torch.full((PROPOSALS,), i, device=device, dtype=torch.int64) for i in range(BATCH)])
    counter = {"i": 0}

    nms_stream = torch.cuda.Stream(device=device, priority=args.side_priority)
    overlap = {"on": args.overlap}
    mode = {"one_launch": bool(args.one_launch) and bool(args.nms_step_fused) and not args.diag_skip}
    # The RoIAlign launch owns nearly all the LDS of every CU it runs on (4 workgroups x 40 KB of 160 KB): a side-stream launch that
    # needs more LDS than is left waits in the dispatcher until it drains.  The step's NMS + payload launch (tvmi::nms_step) is
    # enqueued FIRST (the fork precedes the RoIAlign launches) and is resident before the RoIAlign grid arrives.
    if args.roi_inline_mop is not None:
        torch.ops.tvmi.set_option("roi_align.inline_mop", int(args.roi_inline_mop))
    if args.roi_fold_order is not None:
        torch.ops.tvmi.set_option("roi_align.fold_order", int(args.roi_fold_order))
    for kv in filter(None, os.environ.get("TVMI_SET_OPTIONS", "").split(",")):   # diagnosis: TVMI_SET_OPTIONS=name=value,...
        torch.ops.tvmi.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    torch.ops.tvmi.set_option("nms.step_fused", int(args.nms_step_fused))
    if not args.nms_step_fused:
        def _chain(boxes_, scores_, idxs_, thr_, nseg_, img_, nimg_, maxd_):
            k_, n_ = vision_amd.boxes.batched_nms_padded(boxes_, scores_, idxs_, thr_, nseg_)
            return k_, n_, sharding.pack_kept_payload(boxes_, scores_, img_, k_, n_, nimg_, maxd_)
        sharding.nms_pack_payload = _chain
    # fork / join of the two streams: device-scope events (vision_amd.streams.wait_stream) unless --torch-events
    if args.event_scope >= 0:
        vision_amd.streams.set_event_scope(args.event_scope)
        fork_join = vision_amd.streams.wait_stream
    else:
        fork_join = lambda waiter, signaler: waiter.wait_stream(signaler)   # noqa: E731

    def device_step(which=None):
        # the whole per-rank hot path, no host synchronisation anywhere (-> hipGraph-capturable).  The two halves of the
        # step do not depend on each other (RoIAlign reads the maps and the boxes, NMS the boxes and the scores), and the
        # NMS chain is ~8 short latency-bound launches: it runs on a second HIP stream UNDER the RoIAlign launch, forked
        # from and joined back into the step's stream (0.295-0.300 vs 0.306-0.313 ms per step, gpurun_out r05b; a +25 %
        # outlier seen once in round 3 has not reproduced in eight visits since).  `--one-stream` keeps one stream.
        d = sets[counter["i"] % N_SETS if which is None else which]
        counter["i"] += 1
        if mode["one_launch"]:
            # round 6: ONE launch for the step — the NMS workgroups (tiles, rank counting, sweeps, keep list, padded top-k payload)
            # are the first workgroups of the RoIAlign grid: no second stream, no fork, no join
            pooled, keep, num, payload = pool.forward_with_nms_step(d["feats"], d["boxes"], image_shapes, d["all_boxes"], d["all_scores"],
                                                                    img_idx, NMS_THR, BATCH, img_idx, BATCH, MAX_DETS)
            return pooled, num, payload, keep
        cur = torch.cuda.current_stream()
        side = nms_stream if overlap["on"] else cur
        if side is not cur and args.diag_skip not in ("fork", "both"):
            fork_join(side, cur)
        with torch.cuda.stream(side):
            # per-image NMS + the padded top-MAX_DETS detections per image with the count of every image, written as the
            # collective payload itself (fixed shape, keep length read on the device): ONE launch (tvmi::nms_step, round 6;
            # --nms-step-fused 0 = round 5's chain: score sort, collect, tiles, sweep, pack)
            keep, num, payload = sharding.nms_pack_payload(d["all_boxes"], d["all_scores"], img_idx, NMS_THR, BATCH, img_idx, BATCH, MAX_DETS)
        pooled = pool(d["feats"], d["boxes"], image_shapes)                                 # [4000, 256, 7, 7], 1 launch
        if side is not cur:
            if args.diag_skip not in ("join", "both"):
                fork_join(cur, side)
            for t in (keep, num, payload):
                t.record_stream(cur)     # produced on the side stream, consumed (all-gather, parity check) on this one
        return pooled, num, payload, keep

    graph, static_out = None, None
    if args.graph:
        # launch-bound chain (~12 short kernels next to the RoIAlign launch): capture it once PER INPUT SET (the rotation of
        # the sets is what keeps the Infinity Cache from serving one step's maps to the next), replay round-robin
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for i in range(N_SETS):
                    device_step(i)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph, static_out = [], []
            for i in range(N_SETS):
                gph = torch.cuda.CUDAGraph()
                with torch.no_grad(), torch.cuda.graph(gph):
                    o = device_step(i)
                graph.append(gph)
                static_out.append(o)
        except Exception as exc:  # pragma: no cover - depends on the runtime
            print(f"[bench] hipGraph capture unavailable ({type(exc).__name__}: {exc}); running eagerly", file=sys.stderr)
            graph, static_out = None, None
            torch.cuda.synchronize()

    def step():
        with torch.no_grad():
            if graph is not None:
                i = counter["i"] % N_SETS
                counter["i"] += 1
                graph[i].replay()
                pooled, num, payload, _ = static_out[i]
            else:
                pooled, num, payload, _ = device_step()
            gd, gc = sharding.all_gather_payload(payload, MAX_DETS, always_collective=args.force_collective)   # the one collective (views only at world 1)
        return pooled, num, gd, gc

    def sync():
        if world > 1 or args.force_collective:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the contract region.  Round 4's driver line lost 89 % of its timed region to ONE step of 38-47 ms: a CPython
    # generation-2 garbage collection (36-39 ms in a process with torch imported) that the object counts of 5 warm-up + 5 timed
    # steps happened to trigger inside step 4 (profiles/r05_stall_root_cause_r04_loop.log: 3 of 3 runs, host time of that step
    # = the collection).  So, like `timeit`: collect, freeze and disable the collector around the region; and the warm-up is the
    # timed loop byte for byte (holds `out` across the next step, records an event per step), followed by an internal pre-roll
    # until two consecutive steps agree within 10 %, so that no first-use cost (second 200 MB output block, event creation)
    # is left for the timed steps.  Nothing is trimmed out of `value`: it is K steps between two (barrier + synchronize).
    gc_seen = []

    def gc_probe(phase, info):
        if phase == "stop":
            gc_seen.append(info["generation"])

    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    wm = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for m in marks + wm:
        m.record()                        # creates the HIP event now, not inside a timed region
    held = {"out": None}
    host = {"enqueue_ms": None}

    def contract_region():
        """warm-up (= the timed loop) + pre-roll + EXACTLY K steps between two (barrier + synchronize); returns
        (seconds of the K steps, pre-roll steps run, per-step event times in step order)"""
        for _ in range(args.warmup):
            held["out"] = step()
            marks[1].record()
        preroll = 0
        while preroll < 64:
            wm[0].record()
            held["out"] = step()
            wm[1].record()
            held["out"] = step()
            wm[2].record()
            sync()
            preroll += 2
            a, b = wm[0].elapsed_time(wm[1]), wm[1].elapsed_time(wm[2])
            stable = torch.tensor([1 if abs(a - b) <= 0.1 * min(a, b) else 0], device=device, dtype=torch.int32)
            if world > 1:
                dist.all_reduce(stable, op=dist.ReduceOp.MIN)     # every rank leaves the pre-roll after the same step
            if int(stable.item()):
                break
        sync()
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(args.steps):
            held["out"] = step()
            marks[i + 1].record()
        host["enqueue_ms"] = (time.perf_counter() - t0) / max(args.steps, 1) * 1e3   # host time per step before the final sync
        sync()
        dt = time.perf_counter() - t0
        return dt, preroll, [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]

    gc.collect()
    gc.freeze()
    gc.disable()
    gc.callbacks.append(gc_probe)
    elapsed, preroll, per_step_seq = contract_region()
    host_enqueue_ms = host["enqueue_ms"]
    # the same K steps in the OTHER stream mode (reported next to `value`, never `value` itself)
    other_stream_ms, other_launch_ms = None, None
    if graph is None:
        was = mode["one_launch"]
        mode["one_launch"] = False
        if was:   # the step as two launches on two streams
            other_launch_ms = contract_region()[0] / max(args.steps, 1) * 1e3
        overlap["on"] = not args.overlap
        other_stream_ms = contract_region()[0] / max(args.steps, 1) * 1e3
        overlap["on"] = args.overlap
        mode["one_launch"] = was
    gc.callbacks.remove(gc_probe)
    gc.enable()
    gc.unfreeze()
    out = held["out"]
    argmax_step = max(range(args.steps), key=lambda i: per_step_seq[i]) if args.steps else None
    per_step = sorted(per_step_seq)
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
    ms_per_step = elapsed / max(args.steps, 1) * 1e3
    boxes_per_step = BATCH * PROPOSALS * world
    value = boxes_per_step / (ms_per_step / 1e3)

    # ---- roofline of the dominant kernel, measured live with events on the launch stream (input sets rotated)
    from vision_amd.poolers import LevelMapper, _convert_to_roi_format

    scales = [1.0 / s for s in STRIDES]
    ms_args = (POOL, POOL, SAMPLING, False, 2, 5, 224.0, 4.0, 1e-6)
    for d in sets:
        d["rois5"] = _convert_to_roi_format(d["boxes"]).float()
        d["flist"] = [d["feats"][str(i)] for i in range(len(STRIDES))]

    def timed(fn, n_k=32, warm=4):
        for i in range(warm):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n_k):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n_k

    with torch.no_grad():
        # exactly what pool() launches, without its python-side roi formatting
        k_ms = timed(lambda i: torch.ops.tvmi.multiscale_roi_align(sets[i % N_SETS]["flist"], sets[i % N_SETS]["rois5"], scales, *ms_args))
        roi_alone_ms = k_ms
        if mode["one_launch"]:   # the launch the step makes: the same grid with the NMS workgroups in front (and the order pre-pass)
            k_ms = timed(lambda i: pool.forward_with_nms_step(sets[i % N_SETS]["feats"], sets[i % N_SETS]["boxes"], image_shapes,
                                                              sets[i % N_SETS]["all_boxes"], sets[i % N_SETS]["all_scores"], img_idx, NMS_THR,
                                                              BATCH, img_idx, BATCH, MAX_DETS))
        # NMS + payload alone (what the step launches), for the breakdown in `config`
        nms_ms = timed(lambda i: sharding.nms_pack_payload(sets[i % N_SETS]["all_boxes"], sets[i % N_SETS]["all_scores"], img_idx, NMS_THR,
                                                           BATCH, img_idx, BATCH, MAX_DETS))
        # dense variant: proposals clustered around 40 objects per image, most boxes are suppressed
        dg = torch.Generator().manual_seed(5)
        centers = torch.rand(BATCH, 40, 2, generator=dg) * torch.tensor([IMG_W - 200.0, IMG_H - 200.0])
        pick = torch.randint(0, 40, (BATCH, PROPOSALS), generator=dg)
        xy = torch.gather(centers, 1, pick[..., None].expand(-1, -1, 2)) + torch.randn(BATCH, PROPOSALS, 2, generator=dg) * 12
        wh = 120 + torch.randn(BATCH, PROPOSALS, 2, generator=dg).abs() * 30
        dense_boxes = torch.cat([xy, xy + wh], 2).reshape(-1, 4).to(device)
        dense_scores = torch.rand(BATCH * PROPOSALS, generator=dg).to(device)
        nms_dense_ms = timed(lambda i: vision_amd.boxes.batched_nms_padded(dense_boxes, dense_scores, img_idx, NMS_THR, BATCH))
        dense_kept = int(vision_amd.boxes.batched_nms_padded(dense_boxes, dense_scores, img_idx, NMS_THR, BATCH)[1])
        # the same step through the REFERENCE-SCHEMA ops only — what the unchanged torchvision python
        # (ops/poolers.py:199-222 per-level roi_align + index_put, ops/boxes.py nms per image) launches on this library
        tv = torch.ops.torchvision
        mapper = LevelMapper(2, 5)
        for d in sets:
            lv = mapper(d["boxes"])
            d["lvl_idx"] = [torch.where(lv == l)[0] for l in range(len(STRIDES))]

        def schema_step(i):
            d = sets[i % N_SETS]
            result = torch.zeros(BATCH * PROPOSALS, CHANNELS, POOL, POOL, device=device)
            for l in range(len(STRIDES)):
                idx = d["lvl_idx"][l]
                result[idx] = tv.roi_align(d["flist"][l], d["rois5"][idx], scales[l], POOL, POOL, SAMPLING, False)
            return result, [tv.nms(b, sc, NMS_THR) for b, sc in zip(d["boxes"], d["scores"])]
        schema_ms = timed(schema_step, n_k=16)
        # the same op on channels_last maps (native NHWC kernel) — reported, never part of `value`
        flist_cl = [f.contiguous(memory_format=torch.channels_last) for f in sets[0]["flist"]]
        cl_ms = timed(lambda i: torch.ops.tvmi.multiscale_roi_align(flist_cl, sets[0]["rois5"], scales, *ms_args))
        del flist_cl
    alg_bytes = algorithmic_bytes(feats, BATCH * PROPOSALS)
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    # `traffic` is a committed PMC measurement (separate rocprofv3 --pmc passes, tools/gpu_round.sh + tools/pmc_traffic.py);
    # the line says whether the kernel sources it was taken on are the ones shipped now
    traffic, traffic_current = None, None
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.exists(tpath):
        try:
            import hashlib

            tj = json.load(open(tpath))
            traffic = tj.get("roi_align_fwd_ms_dma_inl_step" if mode["one_launch"] else "roi_align_fwd_ms_dma", {}).get("hbm_bytes_per_launch")
            h = hashlib.sha256()
            for name in ("roi_align.hip", "roi_common.h"):
                h.update(open(os.path.join(ROOT, "vision_amd", "csrc", name), "rb").read())
            traffic_current = tj.get("_measured_on", {}).get("roi_align_sources_sha16") == h.hexdigest()[:16]
        except Exception:
            traffic = None

    result = {
        "metric": "boxes/sec RoIAlign+NMS (1000 prop, 256ch FPN)",
        "value": round(value, 1),
        "unit": "boxes/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "ms_per_step_stats": ({"min": round(per_step[0], 4), "median": round(per_step[len(per_step) // 2], 4),
                               "p90": round(per_step[int(len(per_step) * 0.9)], 4), "max": round(per_step[-1], 4),
                               "argmax_step": argmax_step, "max_over_median": round(per_step[-1] / per_step[len(per_step) // 2], 3),
                               "steps_ms": [round(x, 4) for x in per_step_seq[:64]],
                               "preroll_steps": preroll, "gc_collections_in_timed_region": len(gc_seen),
                               "host_enqueue_ms_per_step": None if host_enqueue_ms is None else round(host_enqueue_ms, 4),
                               "how": "HIP events between consecutive steps of the timed region (rank 0), in step order; the "
                                      "collector is frozen + disabled around warm-up and region, the warm-up loop is the timed loop, "
                                      "then pairs of pre-roll steps until two consecutive steps agree within 10 %"} if per_step else None),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE config 2: per GPU 4x(800x1344) images, FPN P2-P5 x256ch fp32, 1000 proposals/image; "
                        "step = MultiScaleRoIAlign 7x7 sr2 (1 launch) + per-image NMS@0.5 + all-gather of padded top-100 dets",
            "boxes_per_step_per_gpu": BATCH * PROPOSALS,
            "roi_align_ms": round(roi_alone_ms, 4),
            "nms_ms": round(nms_ms, 4),
            "roi_align_channels_last_ms": round(cl_ms, 4),
            "kept_boxes": int(out[1].item()),
            "nms_dense_ms": round(nms_dense_ms, 4),
            "nms_dense_kept_boxes": dense_kept,
            "schema_ops_ms_per_step": round(schema_ms, 4),
            "schema_ops_boxes_per_s": round(BATCH * PROPOSALS / (schema_ms / 1e3), 1),
            "rotated_input_sets": N_SETS,
            "streams": ("one stream, ONE launch: the NMS workgroups ride in front of the RoIAlign grid" if mode["one_launch"] else
                        "NMS + payload launch on a second HIP stream under the RoIAlign launch" if args.overlap else "one stream, two launches"),
            **({"two_launches_two_streams_ms_per_step": round(other_launch_ms, 4)} if other_launch_ms is not None else {}),
            **({"INVALID_diagnosis_run": f"--diag-skip {args.diag_skip}: stream hops left out, not a measurement of the step"} if args.diag_skip else {}),
            "fork_join_events": {1: "device-scope release", 0: "system-scope release", 2: "no event fence", -1: "torch wait_stream"}[args.event_scope],
            ("one_stream_ms_per_step" if args.overlap else "two_stream_ms_per_step"): None if other_stream_ms is None else round(other_stream_ms, 4),
            "hip_graph": graph is not None,
            "nms_step_one_launch": bool(args.nms_step_fused),
            "parallelism": f"images sharded over {world} GPU(s), one process per GPU"
                           + (" (1-rank RCCL group, the all-gather run as a real collective: --force-collective)" if args.force_collective else ""),
        },
        "roofline": {
            "kernel": "roi_align_fwd_ms_dma_inl_step<7,7,2>" if mode["one_launch"] else "roi_align_fwd_ms_dma_inl<7,7,2>",
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "algorithmic_bytes_per_launch": alg_bytes,
            "launch_ms": round(k_ms, 4),
            "traffic": traffic,
            "traffic_measured_on_shipped_sources": traffic_current,
        },
    }

    parity_ok = True
    if rank == 0 and not args.no_cpu_baseline:      # also under N > 1: the line of every run carries its own baseline and parity check
        base, ref_pooled, ref_keeps = cpu_baseline(feats, boxes, scores)
        result["cpu_baseline"] = base
        # parity of THIS run's outputs against what the CPU baseline just computed on the same inputs (input set 0)
        with torch.no_grad():
            pooled, num, _, keep = device_step(0)
        torch.cuda.synchronize()
        err = float((pooled.cpu() - ref_pooled).abs().max())
        keep = keep[: int(num)].cpu()
        same = True
        for i in range(BATCH):
            mine = keep[(keep >= i * PROPOSALS) & (keep < (i + 1) * PROPOSALS)] - i * PROPOSALS
            same = same and torch.equal(mine, ref_keeps[i])
        parity_ok = err <= 1e-4 and same
        result["parity"] = {"roi_align_max_abs_err": err, "roi_align_tolerance": 1e-4, "nms_index_sets_equal": bool(same),
                            "checked_against": base["kind"], "ok": bool(parity_ok)}
    if rank == 0 and not args.no_configs:
        for d in sets[1:]:
            d.clear()                           # the rotated config-2 sets (1.1 GB) are not needed any more
        torch.cuda.empty_cache()
        try:
            result["configs"] = other_configs(device)
        except Exception as exc:  # pragma: no cover - depends on the box
            result["configs"] = {"error": f"{type(exc).__name__}: {exc}"}
    if world > 1 or args.force_collective:
        dist.barrier()
        dist.destroy_process_group()          # the config-5 processes form their own group
    if not args.no_e2e:
        c5 = config5_block(rank, local_rank, world)
        if rank == 0:
            result["config5"] = c5
    if rank == 0:
        print(json.dumps(result), flush=True)
    if not parity_ok:
        sys.exit("bench.py: outputs differ from the CPU reference (see the parity block)")
    if args.e2e and rank == 0 and world == 1:
        print(json.dumps(e2e_config5()), flush=True)


FP32_PEAK_TF = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: fp32 MFMA = fp32 vector peak
BF16_PEAK_TF = 2500.0    # dense bf16 / fp16 MFMA peak


def other_configs(device):
    """BASELINE configs 3 and 4 (and the legs of config 2 that the step does not contain) inside the contract line (VERDICT r03
    item 2): ms = MEDIAN of per-call HIP-event times over
    >= 20 calls, inputs rotated over independent sets; every entry carries the fraction of the SURVEY.md section-8d bound.
      nms / batched_nms : 100,000 boxes (80 classes), IoU 0.5, sparse (1000 px canvas) and dense (200 px) variants; wall
                          time of torchvision::nms incl. the score sort and the output-size sync; bound =
                          max(pairs * 20 flop / 157.3 TF, mask bytes / 8 TB/s)
      deform_conv2d     : 2x256x100x136, k3, 256 -> 256: groups=1 fp32 / bf16 against the dense MFMA peak of the dtype,
                          groups=256 (depthwise) against HBM on its compulsory bytes; the backward (five gradients) of the
                          groups=1 problem against the fp32 MFMA peak on its two contractions"""
    import vision_amd

    tv = torch.ops.torchvision

    def med(fn, n=24, warm=3):
        for i in range(warm):
            fn(i)
        torch.cuda.synchronize()
        ts = []
        for i in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn(i)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ts.sort()
        return ts[len(ts) // 2], ts[0]

    out = {"timing": "median (and min) of 24 per-call HIP-event times, 3 rotated input sets"}
    # config 2, the legs the headline step does not contain: the 14x14 forward and the fused multi-scale BACKWARD (7x7 / 14x14,
    # fp32 / bf16) against HBM on their algorithmic bytes (grads read once + every gradient map written once)
    from vision_amd.poolers import _convert_to_roi_format
    c2 = []
    for i in range(3):
        f, b, _ = make_inputs(device, 300 + i)
        c2.append(([f[str(l)] for l in range(4)], _convert_to_roi_format(b).float()))
    hs, ws = [t.shape[2] for t in c2[0][0]], [t.shape[3] for t in c2[0][0]]
    scales, ms_args = [1.0 / st for st in STRIDES], (2, 5, 224.0, 4.0, 1e-6)
    map_bytes = sum(t.numel() for t in c2[0][0])
    for P in (7, 14):
        for dt, tag in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
            esz = 4 if dt == torch.float32 else 2
            sets = [([t.to(dt) for t in fl], r, torch.randn(BATCH * PROPOSALS, CHANNELS, P, P, device=device).to(dt)) for fl, r in c2]
            bytes_ = (BATCH * PROPOSALS * CHANNELS * P * P + map_bytes) * esz
            if P == 14:
                ms, mn = med(lambda i: torch.ops.tvmi.multiscale_roi_align(sets[i % 3][0], sets[i % 3][1], scales, P, P, 2, False, *ms_args))
                out[f"roi_align_fwd_{P}x{P}_{tag}"] = {"ms": round(ms, 4), "min_ms": round(mn, 4), "GBs": round(bytes_ / ms / 1e6, 1),
                                                       "frac_of_hbm_peak": round(bytes_ / ms / 1e6 / HBM_PEAK_GBS, 4)}
            ms, mn = med(lambda i: torch.ops.tvmi.multiscale_roi_align_backward(sets[i % 3][2], sets[i % 3][1], hs, ws, scales, BATCH, P, P, 2,
                                                                                False, *ms_args))
            out[f"roi_align_bwd_{P}x{P}_{tag}"] = {"ms": round(ms, 4), "min_ms": round(mn, 4), "GBs": round(bytes_ / ms / 1e6, 1),
                                                   "frac_of_hbm_peak": round(bytes_ / ms / 1e6 / HBM_PEAK_GBS, 4),
                                                   "note": "fused multi-scale tile-owner backward, deterministic, no atomics, no zero-fill"}
            del sets
    del c2
    n, nsets = 100_000, 3
    pairs = n * (n - 1) / 2
    for canvas, tag in ((1000, "sparse"), (200, "dense")):
        data = []
        for i in range(nsets):
            g = torch.Generator().manual_seed(7 + 13 * i)
            xy = torch.rand(n, 2, generator=g) * torch.tensor([max(canvas - 64.0, 1.0)] * 2)
            wh = 1 + torch.rand(n, 2, generator=g) * 100
            b = torch.cat([xy, torch.minimum(xy + wh, torch.tensor([float(canvas)] * 2))], 1).to(device)
            data.append((b, torch.rand(n, generator=g).to(device), torch.randint(0, 80, (n,), generator=g).to(device)))
        kept = int(tv.nms(data[0][0], data[0][1], NMS_THR).numel())
        ms, mn = med(lambda i: tv.nms(data[i % nsets][0], data[i % nsets][1], NMS_THR))
        # bound: every pair tested once (20 flop) or the upper-triangular 64-bit mask written and read once
        mask_bytes = 2 * (n * ((n + 63) // 64) * 8) / 2
        bound_ms = max(pairs * 20 / (FP32_PEAK_TF * 1e12), mask_bytes / (HBM_PEAK_GBS * 1e9)) * 1e3
        out[f"nms_100k_{tag}"] = {"ms": round(ms, 4), "min_ms": round(mn, 4), "kept": kept, "Gpairs_per_s": round(pairs / ms / 1e6, 1),
                                  "bound_ms": round(bound_ms, 4), "frac_of_bound": round(bound_ms / ms, 4),
                                  "bound": "all N(N-1)/2 pair tests at 20 flop on the 157.3 TF fp32 vector peak (rows of boxes that are "
                                           "already suppressed are skipped, so the dense variant tests fewer pairs than the bound assumes)"}
        keptb = int(vision_amd.batched_nms(*data[0], NMS_THR).numel())
        ms, mn = med(lambda i: vision_amd.batched_nms(*data[i % nsets], NMS_THR))
        pairs_b = pairs / 80          # same-class pairs only (80 uniform classes)
        bound_b = max(pairs_b * 20 / (FP32_PEAK_TF * 1e12), (2 * n * 4 + n * 16) / (HBM_PEAK_GBS * 1e9)) * 1e3
        out[f"batched_nms_100k_x80_{tag}"] = {"ms": round(ms, 4), "min_ms": round(mn, 4), "kept": keptb, "bound_ms": round(bound_b, 5),
                                              "frac_of_bound": round(bound_b / ms, 4)}
        del data
    B, C, H, W = 2, 256, 100, 136
    cfg4 = []
    for i in range(nsets):
        g = torch.Generator().manual_seed(100 + i)
        cfg4.append(dict(x=torch.randn(B, C, H, W, generator=g).to(device), off=torch.randn(B, 18, H, W, generator=g).to(device),
                         w1=(torch.randn(256, C, 3, 3, generator=g) * 0.01).to(device),
                         wd=(torch.randn(256, 1, 3, 3, generator=g) * 0.01).to(device), bias=torch.randn(256, generator=g).to(device)))
    flops = 2.0 * B * 256 * C * 9 * H * W
    ms, mn = med(lambda i: vision_amd.deform_conv2d(cfg4[i % nsets]["x"], cfg4[i % nsets]["off"], cfg4[i % nsets]["w1"], cfg4[i % nsets]["bias"], padding=1))
    out["deform_conv2d_g1_fp32"] = {"ms": round(ms, 4), "min_ms": round(mn, 4), "TFLOPs": round(flops / ms / 1e9, 1),
                                    "frac_of_mfma_peak": round(flops / ms / 1e9 / FP32_PEAK_TF, 4), "peak_TFLOPs": FP32_PEAK_TF}
    h = [{k: v.to(torch.bfloat16) for k, v in d.items()} for d in cfg4]
    ms, mn = med(lambda i: vision_amd.deform_conv2d(h[i % nsets]["x"], h[i % nsets]["off"], h[i % nsets]["w1"], h[i % nsets]["bias"], padding=1))
    out["deform_conv2d_g1_bf16"] = {"ms": round(ms, 4), "min_ms": round(mn, 4), "TFLOPs": round(flops / ms / 1e9, 1),
                                    "frac_of_mfma_peak": round(flops / ms / 1e9 / BF16_PEAK_TF, 4), "peak_TFLOPs": BF16_PEAK_TF,
                                    "note": "incl. the channels-last re-layout pre-passes of the call"}
    ms, mn = med(lambda i: vision_amd.deform_conv2d(cfg4[i % nsets]["x"], cfg4[i % nsets]["off"], cfg4[i % nsets]["wd"], cfg4[i % nsets]["bias"], padding=1))
    dw_bytes = (2 * B * C * H * W + B * 18 * H * W + 256 * 9 + 256) * 4      # in + out + offsets + weights + bias
    out["deform_conv2d_g256_fp32"] = {"ms": round(ms, 4), "min_ms": round(mn, 4), "compulsory_MB": round(dw_bytes / 1e6, 1),
                                      "GBs": round(dw_bytes / ms / 1e6, 1), "frac_of_hbm_peak": round(dw_bytes / ms / 1e6 / HBM_PEAK_GBS, 4)}
    # backward of the same problem (all five gradients of torchvision::_deform_conv2d_backward, mask in use): two contractions of the
    # forward's size (grad_input / grad_offset / grad_mask from W^T x grad_out, grad_weight from grad_out x columns)
    for tag, sets in (("fp32", cfg4), ("bf16", h)):
        gsets = [dict(d, go=torch.randn(B, 256, H, W, device=device).to(d["x"].dtype), m=torch.rand(B, 9, H, W, device=device).to(d["x"].dtype))
                 for d in sets]
        ms, mn = med(lambda i: tv._deform_conv2d_backward(gsets[i % nsets]["go"], gsets[i % nsets]["x"], gsets[i % nsets]["w1"],
                                                          gsets[i % nsets]["off"], gsets[i % nsets]["m"], gsets[i % nsets]["bias"],
                                                          1, 1, 1, 1, 1, 1, 1, 1, True), n=12)
        peak = FP32_PEAK_TF if tag == "fp32" else BF16_PEAK_TF     # the matrix-core peak of the type the contraction runs in
        out[f"deform_conv2d_backward_g1_{tag}"] = {"ms": round(ms, 4), "min_ms": round(mn, 4), "TFLOPs": round(2 * flops / ms / 1e9, 1),
                                                   "frac_of_mfma_peak": round(2 * flops / ms / 1e9 / peak, 4), "peak_TFLOPs": peak,
                                                   "note": "fp32 tensors on v_mfma_f32_32x32x2_f32, 16-bit tensors on v_mfma_f32_32x32x16_{bf16,f16} (round 5), fp32 accumulation; 12 calls"}
    del cfg4, h
    torch.cuda.empty_cache()
    # ---- round 6: the rows north_star names that lived only in the builder's matrix
    # BASELINE config 1 on the GPU (its CPU form is the reference's own plumbing case): 1x256x200x272 fp32 map, 1000 RoIs 7x7 + nms(1000)
    g = torch.Generator().manual_seed(0)
    c1 = []
    for i in range(nsets):
        x = torch.randn(1, 256, 200, 272, generator=g).to(device)
        xy = torch.rand(1000, 2, generator=g) * torch.tensor([1088 - 64.0, 800 - 64.0])
        wh = 16 + torch.rand(1000, 2, generator=g) * 284
        r1 = torch.cat([torch.zeros(1000, 1), xy, torch.minimum(xy + wh, torch.tensor([1088.0, 800.0]))], 1).to(device)
        c1.append((x, r1, r1[:, 1:].contiguous(), torch.rand(1000, generator=g).to(device)))
    ms_r, mn_r = med(lambda i: tv.roi_align(c1[i % nsets][0], c1[i % nsets][1], 0.25, 7, 7, 2, False))
    ms_n, mn_n = med(lambda i: tv.nms(c1[i % nsets][2], c1[i % nsets][3], NMS_THR))
    c1_bytes = (256 * 200 * 272 + 1000 * 256 * 49 + 5000) * 4
    out["config1_roi_align_1x256x200x272_1000rois_7x7"] = {"ms": round(ms_r, 4), "min_ms": round(mn_r, 4), "GBs": round(c1_bytes / ms_r / 1e6, 1),
                                                           "frac_of_hbm_peak": round(c1_bytes / ms_r / 1e6 / HBM_PEAK_GBS, 4)}
    out["config1_nms_1000"] = {"ms": round(ms_n, 4), "min_ms": round(mn_n, 4), "boxes_per_s_roi_align_plus_nms": round(1000 / (ms_r + ms_n) * 1e3, 1),
                               "note": "torchvision::nms incl. the score sort and the output-size read"}
    del c1
    # resize (transforms/v2/functional/_geometry.py:283-362 -> F.interpolate): 8x3x1080x1920 fp32 -> 800x1422, bytes = input + output
    big = [torch.rand(8, 3, 1080, 1920, generator=g).to(device) for _ in range(nsets)]
    rz_bytes = (8 * 3 * 1080 * 1920 + 8 * 3 * 800 * 1422) * 4
    for mode, aa in (("bilinear", False), ("bilinear", True), ("bicubic", False), ("bicubic", True), ("nearest", False)):
        ms, mn = med(lambda i: vision_amd.interpolate(big[i % nsets], size=(800, 1422), mode=mode, **({} if mode == "nearest" else {"antialias": aa})), n=12)
        out[f"resize_8x3x1080x1920_to_800x1422_{mode}{'_aa' if aa else ''}"] = {
            "ms": round(ms, 4), "min_ms": round(mn, 4), "GBs": round(rz_bytes / ms / 1e6, 1), "frac_of_hbm_peak": round(rz_bytes / ms / 1e6 / HBM_PEAK_GBS, 4)}
    go = torch.randn(8, 3, 800, 1422, device=device)
    ms, mn = med(lambda i: torch.ops.tvmi.interpolate2d_backward(go, 1080, 1920, 2, False, False, -1.0, -1.0), n=12)
    out["resize_bwd_8x3x800x1422_to_1080x1920_bilinear"] = {"ms": round(ms, 4), "min_ms": round(mn, 4), "GBs": round(rz_bytes / ms / 1e6, 1),
                                                            "frac_of_hbm_peak": round(rz_bytes / ms / 1e6 / HBM_PEAK_GBS, 4),
                                                            "note": "gather form: a lane owns an input pixel, deterministic"}
    del big, go
    # RoIPool (cuda/roi_pool_kernel.cu:15-78) on a config-2-shaped single-level workload: 4x256x100x168 map (stride 8), 4000 RoIs, 7x7;
    # bytes = map + values + argmax (forward), grads + argmax read + gradient map written (backward)
    rp = []
    for i in range(nsets):
        x = torch.randn(4, 256, 100, 168, generator=g).to(device)
        xy = torch.rand(4000, 2, generator=g) * torch.tensor([1344 - 64.0, 800 - 64.0])
        wh = 32 + torch.rand(4000, 2, generator=g) * 368
        img = torch.arange(4).repeat_interleave(1000).float()[:, None]
        rp.append((x, torch.cat([img, xy, torch.minimum(xy + wh, torch.tensor([1344.0, 800.0]))], 1).to(device)))
    in_b, out_b = 4 * 256 * 100 * 168 * 4, 4000 * 256 * 49 * 4
    ms, mn = med(lambda i: tv.roi_pool(rp[i % nsets][0], rp[i % nsets][1], 0.125, 7, 7))
    out["roi_pool_fwd_4x256x100x168_4000rois_7x7"] = {"ms": round(ms, 4), "min_ms": round(mn, 4), "GBs": round((in_b + 2 * out_b) / ms / 1e6, 1),
                                                      "frac_of_hbm_peak": round((in_b + 2 * out_b) / ms / 1e6 / HBM_PEAK_GBS, 4)}
    y, am = tv.roi_pool(rp[0][0], rp[0][1], 0.125, 7, 7)
    gr = torch.randn_like(y)
    ms, mn = med(lambda i: tv._roi_pool_backward(gr, rp[0][1], am, 0.125, 7, 7, 4, 256, 100, 168), n=12)
    out["roi_pool_bwd_4x256x100x168_4000rois_7x7"] = {"ms": round(ms, 4), "min_ms": round(mn, 4), "GBs": round((in_b + 2 * out_b) / ms / 1e6, 1),
                                                      "frac_of_hbm_peak": round((in_b + 2 * out_b) / ms / 1e6 / HBM_PEAK_GBS, 4)}
    del rp, y, am, gr
    # rotated IoU (csrc/ops/box_iou_rotated_utils.h:67-383): 2000 x 2000 boxes (cx, cy, w, h, angle), fp32 matrix out
    ra, rb_ = [torch.cat([torch.rand(2000, 2, generator=g) * 800, torch.rand(2000, 2, generator=g) * 200 + 4,
                          (torch.rand(2000, 1, generator=g) - 0.5) * 180], 1).to(device) for _ in range(2)]
    ms, mn = med(lambda i: tv.box_iou_rotated(ra, rb_), n=12)
    out["box_iou_rotated_2000x2000"] = {"ms": round(ms, 4), "min_ms": round(mn, 4), "Mpairs_per_s": round(4e6 / ms / 1e3, 1),
                                        "note": "VALU-bound polygon clipping (no MFMA shape, 16 MB out): pairs/s is the figure of merit"}
    # MFMA utilisation of the deform_conv2d kernels from the committed SQ-counter passes (like roofline.traffic: a separate
    # rocprofv3 --pmc run, tools/prof_pmc_sq.sh + tools/pmc_summary.py -> profiles/dcn_mfma_busy.json)
    mpath = os.path.join(ROOT, "profiles", "dcn_mfma_busy.json")
    if os.path.exists(mpath):
        try:
            mj = json.load(open(mpath))
            for key, row in mj.get("rows", {}).items():
                if key in out:
                    out[key]["mfma_busy_frac"] = row.get("mfma_busy_frac")
                    out[key]["mfma_busy_source"] = row.get("source")
        except Exception:
            pass
    return out


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` outside a launcher: start N ranks (one per GPU) of this very command line."""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n}: launching {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd)


def dry_run(rank, world, args):
    """CPU / gloo: what the launcher, the rank plumbing and the one collective do — no kernel, no measurement."""
    from vision_amd import sharding

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    # the payload form of the timed path (sharding.pack_kept_payload writes it on the GPU): detections + the count in-row
    payload = torch.full((BATCH, MAX_DETS * sharding.DET_FIELDS + 1), float(rank))
    payload[:, -1] = float(rank + 1)
    gd, gc = sharding.all_gather_payload(payload, MAX_DETS)
    ok = gd.shape[0] == BATCH * world and gc.tolist() == [float(r + 1) for r in range(world) for _ in range(BATCH)]
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "boxes/sec RoIAlign+NMS (1000 prop, 256ch FPN)", "value": None, "dry_run": True, "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "gathered_images": int(gd.shape[0]), "gather_ok": bool(ok)}), flush=True)
    if not ok:
        sys.exit("bench.py --dry-run: all-gather payload mismatch")


def config5_block(rank, local_rank, world):
    """BASELINE config 5 for the contract line: ONE fresh process per rank runs tools/e2e_maskrcnn.py --variant both (the
    overlay of the reference python needs its own interpreter state).  Under N > 1 the child processes form their own
    RCCL group on a port next to the launcher's.  Never raises: a failure is reported inside the block."""
    import subprocess

    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env.update(RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + 17))
    cmd = [sys.executable, os.path.join(ROOT, "tools", "e2e_maskrcnn.py"), "--variant", "both", "--score-thresh", "0.0",
           "--steps", "12", "--warmup", "4"]
    t0 = time.perf_counter()
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        out = json.loads(line[-1]) if line else {"error": (p.stderr or p.stdout)[-600:]}
    except Exception as exc:  # pragma: no cover - depends on the box
        out = {"error": f"{type(exc).__name__}: {exc}"}
    # RetinaNet-R50-FPN of the reference on this library (north_star names it): unchanged python (per-level top-k +
    # batched_nms, models/detection/retinanet.py:509-571, landing in our NMS kernels), then vision_amd.fuse_detection_model
    try:
        cmd_r = [sys.executable, os.path.join(ROOT, "tools", "e2e_maskrcnn.py"), "--model", "retinanet", "--variant", "both",
                 "--score-thresh", "0.0", "--steps", "8", "--warmup", "3"]
        env["MASTER_PORT"] = str(int(env["MASTER_PORT"]) + 1)
        p = subprocess.run(cmd_r, capture_output=True, text=True, timeout=300, env=env)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if line:
            r = json.loads(line[-1])
            out["retinanet_reference_python_img_s"] = r.get("reference_python_img_s", r.get("value"))
            out["retinanet_fused_img_s"] = r.get("fused_img_s")          # vision_amd.fuse_detection_model: fused post-processing + transform
            out["retinanet_reference_python_autofuse_img_s"] = r.get("reference_python_autofuse_img_s")   # TVMI_AUTOFUSE=1 class-level swaps
            out["retinanet_same_detections_autofuse"] = r.get("same_detections_autofuse")
            out["retinanet_same_detections_both_ways"] = r.get("same_detections_both_ways")
            out["retinanet_detections_per_image"] = r.get("detections_per_image")
        elif rank == 0:
            out["retinanet_error"] = (p.stderr or p.stdout)[-400:]
    except Exception as exc:  # pragma: no cover - depends on the box
        out["retinanet_error"] = f"{type(exc).__name__}: {exc}"
    out["wall_s"] = round(time.perf_counter() - t0, 1)
    # where the GPU time of the fused step goes: a committed rocprofv3 --kernel-trace of tools/e2e_maskrcnn.py
    # --variant fused, summarised by tools/kernel_share.py (a separate profiler run, like roofline.traffic)
    kpath = os.path.join(ROOT, "profiles", "config5_kernel_share.json")
    if os.path.exists(kpath):
        try:
            ks = json.load(open(kpath))
            out["kernel_share"] = {"share_of_kernel_time": ks.get("share_of_kernel_time"), "idle_frac_of_span": ks.get("idle_frac_of_span"),
                                   "top_kernel": (ks.get("top_kernels") or [{}])[0], "source": ks.get("source", "profiles/config5_kernel_share.json")}
        except Exception:
            pass
    return out


def e2e_config5():
    """BASELINE config 5 in fresh processes (tools/e2e_maskrcnn.py): the overlay of the reference python needs its own
    interpreter state.  For N > 1 launch that script under torchrun directly (it shards the images over the ranks)."""
    import subprocess

    out = {"metric": "MaskRCNN-R50 img/s (BASELINE config 5)", "n_gpus": 1, "runs": []}
    for variant in ("reference", "fused"):
        for thresh in ("0.0", "0.05"):
            cmd = [sys.executable, os.path.join(ROOT, "tools", "e2e_maskrcnn.py"), "--variant", variant, "--score-thresh", thresh]
            try:
                p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
                line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
                out["runs"].append(json.loads(line[-1]) if line else {"variant": variant, "error": p.stderr[-800:]})
            except Exception as exc:  # pragma: no cover
                out["runs"].append({"variant": variant, "error": f"{type(exc).__name__}: {exc}"})
    return out


def cpu_baseline(feats, boxes, scores):
    """The reference's CPU path on the SAME workload, on this host (checker only — never the product)."""
    from oracle import oracle as O

    cpu_feats = {k: v.cpu() for k, v in feats.items()}
    cpu_boxes = [b.cpu() for b in boxes]
    cpu_scores = [s.cpu() for s in scores]
    kind = "reference" if O.load_reference() else "port"
    import vision_amd
    from vision_amd.poolers import LevelMapper, _convert_to_roi_format

    rois = _convert_to_roi_format(cpu_boxes)
    levels = LevelMapper(2, 5)(cpu_boxes)
    scales = [1.0 / s for s in STRIDES]
    torch.set_num_threads(1)

    ref_pooled = torch.zeros(BATCH * PROPOSALS, CHANNELS, POOL, POOL)
    ref_keeps = [None] * BATCH

    def run_once():
        t0 = time.perf_counter()
        outs = []
        for lvl in range(len(STRIDES)):
            sel = torch.nonzero(levels == lvl)[:, 0]
            if kind == "reference":
                outs.append((sel, torch.ops.torchvision.roi_align(cpu_feats[str(lvl)], rois[sel], scales[lvl], POOL, POOL, SAMPLING, False)))
            else:
                outs.append((sel, torch.from_numpy(O.roi_align(cpu_feats[str(lvl)].numpy(), rois[sel].numpy(), scales[lvl], POOL, POOL,
                                                               SAMPLING, False))))
        keeps = []
        for b, s in zip(cpu_boxes, cpu_scores):
            if kind == "reference":
                keeps.append(torch.ops.torchvision.nms(b, s, NMS_THR))
            else:
                keeps.append(torch.from_numpy(O.nms(b.numpy(), s.numpy(), NMS_THR)))
        dt = time.perf_counter() - t0
        for sel, o in outs:   # kept for the parity block (outside the timed part)
            ref_pooled[sel] = o
        ref_keeps[:] = keeps
        return dt

    times = [run_once()]  # first run doubles as warm-up and is kept if the budget is tight
    budget = 20.0
    while sum(times) < budget and len(times) < 9:
        times.append(run_once())
    times.sort()
    med = times[len(times) // 2]
    return {
        "value": round(BATCH * PROPOSALS / med, 1),
        "unit": "boxes/s",
        "cores": 1,
        "kind": kind,
        "sample": f"full step workload (4 images x 1000 proposals, 4 FPN levels, RoIAlign 7x7 + per-image NMS), "
                  f"median of {len(times)} runs, {sum(times):.1f} s of CPU work, single-threaded reference kernels; "
                  f"host has {os.cpu_count()} logical CPUs",
        "seconds_per_step": round(med, 4),
    }, ref_pooled, ref_keeps


if __name__ == "__main__":
    main()
