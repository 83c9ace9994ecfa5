"""Cross-stream waits for the two halves of a detection step on one MI355X.

The NMS + payload launch of a step does not depend on its RoIAlign launch: `bench.py` runs it on a second HIP stream, forked
from and joined back into the step's stream.  `wait_stream(waiter, signaler)` is torch's `waiter.wait_stream(signaler)` with an
event that releases to DEVICE scope (tvmi_stream_wait_stream): both streams live on one GPU, and torch's events release to
system scope — a write-back + invalidate meant for host readers — on every record.

    side = torch.cuda.Stream()
    vision_amd.streams.wait_stream(side, torch.cuda.current_stream())        # fork
    with torch.cuda.stream(side):
        keep, num, payload = vision_amd.sharding.nms_pack_payload(...)
    pooled = pool(features, proposals, image_shapes)
    vision_amd.streams.wait_stream(torch.cuda.current_stream(), side)        # join

(Rounds 3-5 also shipped a CU-masked stream pair — the RoIAlign stream leaving one CU per XCD to the side stream,
hipExtStreamCreateWithCUMask — for side launches that need more LDS than the RoIAlign kernel leaves on a CU.  The step never
needed it once its NMS launches were slimmed down, nothing else used it, and it was removed in round 6; the measurements are in
DESIGN.md / HISTORY.md.)
"""
import ctypes

import torch

from . import _loader


def wait_stream(waiter: "torch.cuda.Stream", signaler: "torch.cuda.Stream") -> None:
    """`waiter.wait_stream(signaler)` with a device-scope event (tvmi_stream_wait_stream): both streams are on one GPU, and
    torch's events release to system scope on every record."""
    if waiter == signaler:
        return
    lib = _loader.kernels()
    with torch.cuda.device(waiter.device):
        st = lib.tvmi_stream_wait_stream(ctypes.c_void_p(waiter.cuda_stream), ctypes.c_void_p(signaler.cuda_stream))
    if st != 0:
        lib.tvmi_last_error.restype = ctypes.c_char_p
        raise RuntimeError(f"tvmi_stream_wait_stream failed: {lib.tvmi_last_error().decode()}")


def set_event_scope(scope: int) -> None:
    """0: system-scope release (torch's events), 1: device scope (default), 2: no fence from the event itself."""
    if _loader.kernels().tvmi_stream_event_scope(ctypes.c_int(scope)) != 0:
        raise ValueError("event scope must be 0, 1 or 2")
