"""RCCL on hardware: a 1-rank `nccl` process group (RCCL rejects two ranks on one device, and the GPU box has one GPU)
pushes the REAL detection payload of the hot path through `all_gather_into_tensor` on device buffers — the world-1
short-circuit of vision_amd.sharding is bypassed with always_collective=True.  Reference pattern being replaced:
references/detection/utils.py:70-83 (pickled all_gather_object) and :260-282 (init_process_group("nccl"))."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from helpers import gen, random_boxes

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_one_rank_rccl_group_all_gathers_the_detection_payload():
    import vision_amd
    from vision_amd import sharding

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        g = gen(3)
        B, P = 4, 1000
        boxes = torch.cat([random_boxes(P, 1344, 800, 16, 300, g) for _ in range(B)]).to(dev)
        scores = torch.rand(B * P, generator=g).to(dev)
        img = torch.arange(B, device=dev).repeat_interleave(P)
        keep, num = vision_amd.boxes.batched_nms_padded(boxes, scores, img, 0.5, B)
        dets, counts = sharding.pack_kept_detections(boxes, scores, img, keep, B, 100, num_keep=num)
        gd, gc = sharding.all_gather_detections(dets, counts, always_collective=True)   # RCCL all_gather_into_tensor
        torch.cuda.synchronize()
        assert gd.data_ptr() != dets.data_ptr(), "the collective was short-circuited"
        assert torch.equal(gd, dets) and torch.equal(gc, counts) and gd.shape == (B, 100, 6)
        # the lean form bench.py uses: the packing launch writes the payload, the collective returns views of the gathered buffer
        payload = sharding.pack_kept_payload(boxes, scores, img, keep, num, B, 100)
        pd, pc = sharding.all_gather_payload(payload, 100, always_collective=True)
        torch.cuda.synchronize()
        assert pd.data_ptr() != payload.data_ptr() and torch.equal(pd, dets) and torch.equal(pc.round().to(torch.int32), counts)
        # and a plain all_reduce, so that a second RCCL kernel has run on this communicator
        t = torch.ones(8, device=dev)
        dist.all_reduce(t)
        assert float(t.sum()) == 8.0
        # the result dict form used by tools/e2e_maskrcnn.py
        outs = sharding.unpack_detections(gd, gc)
        d2, c2 = sharding.pack_detection_dicts(outs, 100)
        assert torch.equal(d2, dets) and torch.equal(c2, counts)
    finally:
        dist.destroy_process_group()
