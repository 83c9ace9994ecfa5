"""N>1 path on CPU: world_size-2 gloo run of the image sharding + the one fixed-shape
all-gather of padded detections (vision_amd/sharding.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vision_amd import sharding


def test_shard_range_partitions_everything():
    for n in (0, 1, 7, 8, 9, 64):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(4, 2, 2)


def test_pack_unpack_roundtrip():
    boxes = [torch.rand(3, 4), torch.rand(0, 4), torch.rand(7, 4)]
    scores = [torch.rand(3), torch.rand(0), torch.rand(7)]
    labels = [torch.randint(0, 90, (3,)), torch.zeros(0, dtype=torch.int64), torch.randint(0, 90, (7,))]
    dets, counts = sharding.pack_detections(boxes, scores, labels, max_dets=5)
    assert dets.shape == (3, 5, 6) and counts.tolist() == [3, 0, 5]
    out = sharding.unpack_detections(dets, counts)
    torch.testing.assert_close(out[0]["boxes"], boxes[0])
    torch.testing.assert_close(out[2]["scores"], scores[2][:5])
    assert out[2]["labels"].tolist() == labels[2][:5].tolist() and out[1]["boxes"].shape == (0, 4)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total_images, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = sharding.shard_range(total_images, rank, world)
        g = torch.Generator().manual_seed(1234)  # same stream on all ranks -> comparable "dataset"
        all_boxes = [torch.rand(4 + i, 4, generator=g) for i in range(total_images)]
        all_scores = [torch.rand(4 + i, generator=g) for i in range(total_images)]
        all_labels = [torch.randint(0, 80, (4 + i,), generator=g) for i in range(total_images)]
        dets, counts = sharding.pack_detections(all_boxes[lo:hi], all_scores[lo:hi], all_labels[lo:hi], max_dets=6)
        gd, gc = sharding.all_gather_detections(dets, counts)
        full_d, full_c = sharding.pack_detections(all_boxes, all_scores, all_labels, max_dets=6)
        ok = torch.equal(gd, full_d) and torch.equal(gc, full_c)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_two_rank_all_gather_of_detections():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, 6, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_single_process_gather_is_identity():
    d, c = torch.rand(2, 3, 6), torch.tensor([3, 1], dtype=torch.int32)
    gd, gc = sharding.all_gather_detections(d, c)
    assert gd is d and gc is c


def _bench(*argv, env=None, timeout=300):
    import subprocess
    import sys

    from helpers import ROOT

    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=e, timeout=timeout)


def test_bench_gpus_n_launches_n_ranks_itself():
    """`python bench.py --gpus 2` outside torchrun must start two ranks, here on CPU through --dry-run / gloo:
    launcher, RANK / WORLD_SIZE plumbing and the one all-gather; the line reports n_gpus = 2."""
    import json

    p = _bench("--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["dry_run"] is True and line["gather_ok"] is True and line["gathered_images"] == 8


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    p = _bench("--gpus", "2", "--dry-run", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "refusing" in p.stderr


def test_bench_dry_run_with_four_ranks():
    """the launcher / rank plumbing / collective beyond two ranks (gloo, CPU)."""
    import json

    p = _bench("--gpus", "4", "--dry-run", "--steps", "2", "--warmup", "1", timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 4 and line["gather_ok"] is True and line["gathered_images"] == 16


def _payload_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, D = 3, 5
        payload = torch.full((B, D * sharding.DET_FIELDS + 1), float(rank))
        payload[:, -1] = torch.arange(B, dtype=torch.float32) + rank      # counts ride in the last column
        gd, gc = sharding.all_gather_payload(payload, D)
        ok = gd.shape == (world * B, D, sharding.DET_FIELDS) and gc.tolist() == [float(i + r) for r in range(world) for i in range(B)]
        ok = ok and all(float(gd[r * B:(r + 1) * B].min()) == float(r) == float(gd[r * B:(r + 1) * B].max()) for r in range(world))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_three_rank_all_gather_of_the_in_place_payload():
    world, port = 3, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_payload_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True, 2: True}
