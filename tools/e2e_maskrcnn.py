#!/usr/bin/env python3
"""BASELINE config 5 at N GPUs of one node: Mask R-CNN ResNet50-FPN inference, random-init weights, synthetic
3x800x1333 images, the batch sharded over the ranks (images are independent), ONE fixed-shape all-gather of the
padded detections per step (vision_amd.sharding).  Runs the UNCHANGED reference python package
(models/detection/generalized_rcnn.py:53-133, mask_rcnn.py) laid over our operator library
(vision_amd.integration.make_overlay); the backbone / heads are MIOpen + hipBLASLt through torch, outside our kernels.

    python tools/e2e_maskrcnn.py --variant reference|fused|both|check [--model maskrcnn|fasterrcnn|retinanet] [--batch 2]
                                 [--steps 8] [--warmup 3] [--score-thresh 0.0]

variant reference : nothing swapped — every torchvision.ops call of the reference python lands in the
                    `torchvision::` schema kernels of this library (and, with the opt-in aten override, every
                    F.interpolate in resize.hip)
variant fused     : vision_amd.{MultiScaleRoIAlign, postprocess_detections, filter_proposals, paste_masks_in_image,
                    transform_images} swapped in
variant both      : ONE process, one model: the reference variant is timed, then the same unchanged model with the opt-in
                    class-level swaps of TVMI_AUTOFUSE=1 (vision_amd/autofuse.py), then the fused pieces bound to the model
                    (vision_amd.fuse_detection_model); the detections of all three on the same images are compared (what
                    bench.py puts into its `config5` block)
Prints one JSON object.  Launched by `bench.py --e2e` in a fresh process (the overlay needs TVMI_NO_PY_REGISTRATIONS=1
before vision_amd is imported: the reference package brings its own fake / autograd registrations)."""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TVMI_NO_PY_REGISTRATIONS"] = "1"

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="reference", choices=["reference", "fused", "both", "check"],
                    help="check: run the SAME model both ways on the same images and compare the detections; both: time both "
                         "ways in one process and compare")
    ap.add_argument("--model", default="maskrcnn", choices=["maskrcnn", "fasterrcnn", "retinanet"],
                    help="detection model of the reference (all ResNet50-FPN, random init); the fused variant swaps what the "
                         "model has: RetinaNet gets vision_amd.retinanet_postprocess_detections (retinanet.py:509-571: per-level top-k + batched_nms, "
                         "which lands in our NMS kernels unchanged) and only gets the fused image transform")
    ap.add_argument("--batch", type=int, default=2, help="images per GPU and step")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--score-thresh", type=float, default=0.0,
                    help="box_score_thresh; random-init class scores are ~1/91, so 0.0 keeps the post-processing busy "
                         "(100 detections per image) and the default 0.05 of the reference keeps none")
    ap.add_argument("--no-aten-override", action="store_true")
    ap.add_argument("--channels-last", action="store_true", help="run the model (backbone / FPN / heads) in torch.channels_last: the FPN "
                    "maps then reach MultiScaleRoIAlign as NHWC tensors and the 7x7 pooling takes the native channels_last kernel")
    args = ap.parse_args()

    rank, local_rank, world = (int(os.environ.get(k, "0")) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"))
    world = max(world, 1)
    assert torch.cuda.is_available(), "config 5 is measured on the GPU"
    device = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import vision_amd
    from vision_amd import integration, sharding
    from tools.stage_reference_python import reference_package

    scratch = tempfile.mkdtemp(prefix="tvmi_e2e_")
    pkg = reference_package(scratch)
    if pkg is None:
        print(json.dumps({"e2e": None, "reason": "reference python package neither present nor staged"}))
        return
    sys.path.insert(0, integration.make_overlay(os.path.join(scratch, "overlay"), pkg))
    import torchvision  # the reference package, unmodified
    from torchvision import extension
    from torchvision.models import detection as D

    assert extension._has_ops(), "the reference loader did not find our library as _C / _C_stable"
    if not args.no_aten_override:
        vision_amd.override_aten_upsample(True)

    torch.manual_seed(0)
    if args.model == "retinanet":
        model = D.retinanet_resnet50_fpn(weights=None, weights_backbone=None, score_thresh=args.score_thresh)
    else:
        ctor = D.maskrcnn_resnet50_fpn if args.model == "maskrcnn" else D.fasterrcnn_resnet50_fpn
        model = ctor(weights=None, weights_backbone=None, box_score_thresh=args.score_thresh)
    model = model.eval().to(device)
    if args.channels_last:
        model = model.to(memory_format=torch.channels_last)
    has_masks = args.model == "maskrcnn"

    def apply_fused():
        # the product API (vision_amd/integration.py): pools, post-processing, proposal filtering, transform, mask pasting
        integration.fuse_detection_model(model)

    if args.variant == "fused":
        apply_fused()

    g = torch.Generator().manual_seed(100 + rank)
    batches = [[torch.rand(3, 800, 1333, generator=g).to(device) for _ in range(args.batch)] for _ in range(2)]

    def compare(ref, fus):
        """same detections in the same order; values differ by fp32 rounding carried through a random-init network.  Where the
        scores TIE (a random-init RetinaNet: sigmoid collapses many logits onto one float) the order among equal scores is
        torch.topk's choice and differs between the per-image 1-D calls of the reference and the batched call — there the
        detections are compared as a multiset, leaving out the ties that straddle the detections_per_img cut."""
        rep, ok = [], True
        for a, b in zip(ref, fus):
            n = min(len(a["scores"]), len(b["scores"]))
            same_n = len(a["scores"]) == len(b["scores"])
            lab = bool(torch.equal(a["labels"][:n], b["labels"][:n]))
            ds = float((a["scores"][:n] - b["scores"][:n]).abs().max()) if n else 0.0
            db = float((a["boxes"][:n] - b["boxes"][:n]).abs().max()) if n else 0.0
            dm = float((a["masks"][:n] - b["masks"][:n]).abs().max()) if (n and has_masks) else 0.0
            entry = {"detections": [len(a["scores"]), len(b["scores"])], "labels_equal": lab, "max_score_diff": ds,
                     "max_box_diff_px": db, "max_mask_diff": dm}
            good = same_n and lab and ds < 2e-4 and db < 0.25 and dm < 5e-3
            if not good and same_n and n:
                # Same detections in another ORDER: two scores closer than the fp32 noise the two pipelines carry (5e-5: a
                # random-init network produces many such pairs, a RetinaNet's sigmoid even exact ties) swap ranks, and every
                # rank-wise difference above explodes.  One-to-one matching instead: same label, score within 2e-4, every
                # coordinate within 0.25 px (and the mask within 5e-3); a detection may stay unmatched only if its score is
                # within 2e-4 of the lowest one (a tie that straddles the detections_per_img cut).
                # (An earlier form rounded the coordinates to 0.1 px and compared sorted lists: it flipped on boxes that sit on a
                # rounding boundary — the same run came out True on one GPU-box visit and False on the next.)
                d = (a["boxes"][:, None, :] - b["boxes"][None, :, :]).abs().amax(-1)
                ok_pair = (a["labels"][:, None] == b["labels"][None, :]) & ((a["scores"][:, None] - b["scores"][None, :]).abs() < 2e-4)
                d = torch.where(ok_pair, d, torch.full_like(d, 1e9))
                near, idx = d.min(1)
                matched = near < 0.25
                cut = float(torch.minimum(a["scores"].min(), b["scores"].min())) + 2e-4
                good = bool((matched | (a["scores"] <= cut)).all())
                mi = idx[matched]
                good = good and int(torch.unique(mi).numel()) == int(mi.numel())
                unmatched_b = torch.ones(len(b["scores"]), dtype=torch.bool, device=mi.device)
                unmatched_b[mi] = False
                good = good and bool((b["scores"][unmatched_b] <= cut).all())
                if good and has_masks and int(matched.sum()):
                    dm2 = float((a["masks"][matched] - b["masks"][mi]).abs().max())
                    entry["max_mask_diff_matched"] = dm2
                    good = dm2 < 5e-3
                entry["same_detections_in_another_order"] = good
                entry["matched"] = int(matched.sum())
            rep.append(entry)
            # the two pipelines differ by the rounding of their first op (fused normalise + resize vs F.interpolate) carried through
            # a random-init 50-layer network: 5e-5 .. 6e-5 in the scores and 0.003 .. 0.06 px in the boxes across GPU-box visits
            ok = ok and good
        return rep, ok

    if args.variant == "check":
        # same weights, same images: unchanged reference python vs the fused pieces
        with torch.no_grad():
            ref = model(batches[0])
            apply_fused()
            fus = model(batches[0])
        images, ok = compare(ref, fus)
        print(json.dumps({"e2e_check": "reference python vs fused vision_amd pieces, same model and images", "images": images,
                          "ok": ok}), flush=True)
        sys.exit(0 if ok else 1)

    def step(i):
        with torch.no_grad():
            out = model(batches[i % 2])
        # fixed-shape detection payload of this rank's images (shapes are host-known: no device read) and its one all-gather
        dets, counts = sharding.pack_detection_dicts(out, 100)
        sharding.all_gather_detections(dets, counts)
        return out

    def timed_run(variant):
        # like bench.py's contract region (and `timeit`): the collector is frozen + disabled around warm-up and timed steps — a
        # generation-2 collection of a process with torch imported is a 40-50 ms stop (one landed in 1 of 6 steps of the
        # round-5 first visit: 18.3 ms steps, one of 70 ms), which is not what a steps-per-second figure over 6 steps is about
        import gc

        gc.collect()
        gc.freeze()
        gc.disable()
        try:
            for i in range(args.warmup):
                out = step(i)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            calls0 = int(torch.ops.tvmi.aten_upsample_calls())
            t0 = time.perf_counter()
            for i in range(args.steps):
                out = step(i)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            # the spread of single steps (each one synchronised; not part of `value`)
            per = []
            for i in range(min(args.steps, 6)):
                t1 = time.perf_counter()
                out = step(i)
                torch.cuda.synchronize()
                per.append(round((time.perf_counter() - t1) * 1e3, 2))
        finally:
            gc.enable()
            gc.unfreeze()
        if world > 1:
            tmax = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = tmax.item()
        return {
            "e2e": f"{args.model}_resnet50_fpn inference" + (" (BASELINE config 5)" if has_masks else ""),
            "variant": variant,
            "value": round(args.batch * world * args.steps / dt, 3),
            "unit": "img/s",
            "n_gpus": world,
            "images_per_gpu_per_step": args.batch,
            "ms_per_step": round(dt / args.steps * 1e3, 2),
            "synced_single_steps_ms": per,
            "steps": args.steps,
            "warmup": args.warmup,
            "box_score_thresh": args.score_thresh,
            "detections_per_image": [int(o["boxes"].shape[0]) for o in out],
            "mask_shape": list(out[0]["masks"].shape) if has_masks else None,
            "aten_upsample_override": not args.no_aten_override,
            "channels_last": bool(args.channels_last),
            "aten_upsample_calls_per_step": (int(torch.ops.tvmi.aten_upsample_calls()) - calls0) / max(args.steps, 1),
            "reference_python": torchvision.__file__,
            "data": "synthetic", "weights": "random init (seed 0)", "dtype": "f32",
        }

    if args.variant == "both":
        ref_run = timed_run("reference")
        with torch.no_grad():
            ref_out = model(batches[0])
        # the same UNCHANGED model object with the opt-in class-level swaps of TVMI_AUTOFUSE=1 (vision_amd/autofuse.py) ...
        from vision_amd import autofuse
        autofuse.install()
        auto_run = timed_run("autofuse")
        with torch.no_grad():
            auto_out = model(batches[0])
        auto_images, auto_ok = compare(ref_out, auto_out)
        autofuse.uninstall()
        # ... and with the fused pieces bound to this one model (vision_amd.fuse_detection_model)
        apply_fused()
        fus_run = timed_run("fused")
        with torch.no_grad():
            fus_out = model(batches[0])
        images, ok = compare(ref_out, fus_out)
        res = {"e2e": ref_run["e2e"], "unit": "img/s", "n_gpus": world, "images_per_gpu_per_step": args.batch,
               "box_score_thresh": args.score_thresh, "steps": args.steps, "warmup": args.warmup,
               "reference_python_img_s": ref_run["value"], "reference_python_ms_per_step": ref_run["ms_per_step"],
               "fused_img_s": fus_run["value"], "fused_ms_per_step": fus_run["ms_per_step"],
               "reference_python_autofuse_img_s": auto_run["value"], "reference_python_autofuse_ms_per_step": auto_run["ms_per_step"],
               "synced_single_steps_ms": {"reference": ref_run["synced_single_steps_ms"], "autofuse": auto_run["synced_single_steps_ms"],
                                          "fused": fus_run["synced_single_steps_ms"]},
               "same_detections_autofuse": auto_ok,
               "detections_per_image": fus_run["detections_per_image"],
               "aten_upsample_calls_per_step": [ref_run["aten_upsample_calls_per_step"], fus_run["aten_upsample_calls_per_step"]],
               "same_detections_both_ways": ok, "check": images, "data": "synthetic", "weights": "random init (seed 0)",
               "dtype": "f32"}
    else:
        res = timed_run(args.variant)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
