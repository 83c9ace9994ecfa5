"""Device-scope cross-stream waits (vision_amd/streams.py, tvmi_stream_wait_stream of include/tvmi.h): the fork / join bench.py
runs the two halves of its step with."""
import pytest
import torch

import vision_amd
from helpers import gen, rois_for

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_fork_join_order_and_results():
    main, side = torch.cuda.Stream(), torch.cuda.Stream()
    # fork / join through tvmi_stream_wait_stream: every hand-over is ordered (a race would leave another number)
    x = torch.zeros(1 << 20, device=DEV)
    torch.cuda.synchronize()
    with torch.cuda.stream(main):
        for _ in range(20):
            vision_amd.streams.wait_stream(side, main)
            with torch.cuda.stream(side):
                x.add_(1)
            vision_amd.streams.wait_stream(main, side)
            x.mul_(2)
    torch.cuda.synchronize()
    want = 0
    for _ in range(20):
        want = (want + 1) * 2
    assert float(x.min()) == float(x.max()) == float(want)
    # the two halves of a step on the pair give what they give on one stream
    g = gen(5)
    feat = torch.randn(2, 64, 50, 84, generator=g).to(DEV)
    rois = rois_for(2, 300, 672, 400, 16, 200, g).to(DEV)
    boxes = rois[:, 1:].contiguous()
    scores = torch.rand(300, generator=g).to(DEV)
    ref_pool = torch.ops.torchvision.roi_align(feat, rois, 0.125, 7, 7, 2, False)
    ref_keep = torch.ops.torchvision.nms(boxes, scores, 0.5)
    torch.cuda.synchronize()
    with torch.cuda.stream(main):
        pool = torch.ops.torchvision.roi_align(feat, rois, 0.125, 7, 7, 2, False)
        vision_amd.streams.wait_stream(side, main)
        with torch.cuda.stream(side):
            keep = torch.ops.torchvision.nms(boxes, scores, 0.5)
        vision_amd.streams.wait_stream(main, side)
    torch.cuda.synchronize()
    assert torch.equal(pool, ref_pool) and torch.equal(keep, ref_keep)
    for scope in (0, 2, 1):
        vision_amd.streams.set_event_scope(scope)
        with torch.cuda.stream(main):
            vision_amd.streams.wait_stream(side, main)
            with torch.cuda.stream(side):
                y = x + 1
            vision_amd.streams.wait_stream(main, side)
            z = y * 2
        torch.cuda.synchronize()
        assert float(z[0]) == (want + 1) * 2
    with pytest.raises(ValueError):
        vision_amd.streams.set_event_scope(7)
