"""Stream pair for the two halves of a detection step on one MI355X.

The RoIAlign launch of a step fills the chip: 8,000 workgroups of 39 KB of LDS each, four per CU = all 160 KB of every CU.
The NMS / packing chain next to it is ~6 short launches whose workgroups need 0.5-32 KB of LDS; on a second HIP stream they
sit in the dispatcher until the RoIAlign kernel drains (kernel-trace of the step: `nms_small_seg_tiles` 16 us alone, 100-195 us
under the RoIAlign launch, sweep + packing after its end), and a high-priority queue does not change that (measured).
Two cures exist.  This module is the general one; the detection step of this library took the other (every launch of its NMS
chain was slimmed down to the 4 KB of LDS the RoIAlign kernel leaves on a CU — rank-counting score sort, split small-segment
kernels — so `bench.py` runs on two ordinary streams by default and `--reserve-cus 8` selects this module's pair).
`partitioned_streams(reserve)` returns (main, side): `main` may use every CU except `reserve` of them — one per XCD for
reserve = 8, the driver deals the mask bits round-robin over the XCDs — and `side` may use all; the short launches find
the reserved CUs empty and the chain finishes under the RoIAlign launch instead of behind it.

    main, side = vision_amd.streams.partitioned_streams(8)
    with torch.cuda.stream(main):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            keep, num = vision_amd.boxes.batched_nms_padded(...)
        pooled = pool(features, proposals, image_shapes)
        main.wait_stream(side)
"""
import atexit
import ctypes
from typing import Optional, Tuple

import torch

from . import _loader

_KEEP = []   # hipStream_t handles: torch's ExternalStream does not own its handle; destroyed by destroy_all() / at exit


def destroy_all() -> None:
    """Destroy the CU-masked streams created so far (idempotent; also registered with atexit, before the HIP runtime and any
    profiler tear down their queues).  The torch stream objects that wrapped them must not be used afterwards."""
    lib = _loader.kernels()
    while _KEEP:
        h = _KEEP.pop()
        try:
            torch.cuda.synchronize()
        except Exception:   # pragma: no cover - interpreter shutdown
            pass
        lib.tvmi_stream_destroy(ctypes.c_void_p(h))


atexit.register(destroy_all)


def cu_masked_stream(enabled_cus, total_cus: int, device: Optional[torch.device] = None) -> "torch.cuda.ExternalStream":
    """A torch stream that may only run on the CUs in `enabled_cus` (iterable of CU indices in [0, total_cus))."""
    lib = _loader.kernels()
    words = (total_cus + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for cu in enabled_cus:
        if not 0 <= cu < total_cus:
            raise ValueError(f"CU index {cu} outside [0, {total_cus})")
        mask[cu >> 5] |= 1 << (cu & 31)
    handle = ctypes.c_void_p()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev):
        lib.tvmi_stream_create_cu_mask.restype = ctypes.c_int
        st = lib.tvmi_stream_create_cu_mask(mask, ctypes.c_uint32(words), ctypes.byref(handle))
    if st != 0 or not handle.value:
        lib.tvmi_last_error.restype = ctypes.c_char_p
        raise RuntimeError(f"tvmi_stream_create_cu_mask failed: {lib.tvmi_last_error().decode()}")
    _KEEP.append(handle.value)
    return torch.cuda.ExternalStream(handle.value, device=dev)


def partitioned_streams(reserve: int = 8, device: Optional[torch.device] = None) -> Tuple["torch.cuda.Stream", "torch.cuda.Stream"]:
    """(main, side): main runs on all CUs but the first `reserve` mask bits (one CU per XCD for each 8), side on all CUs.
    reserve = 0 gives two ordinary streams."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if reserve <= 0:
        return torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    total = torch.cuda.get_device_properties(dev).multi_processor_count
    if reserve >= total:
        raise ValueError(f"cannot reserve {reserve} of {total} CUs")
    main = cu_masked_stream(range(reserve, total), total, dev)
    side = cu_masked_stream(range(total), total, dev)
    return main, side


def wait_stream(waiter: "torch.cuda.Stream", signaler: "torch.cuda.Stream") -> None:
    """`waiter.wait_stream(signaler)` with a device-scope event (tvmi_stream_wait_stream): both streams are on one GPU, and
    torch's events release to system scope on every record.  Two hops per step were ~25 us of the 0.28 ms step."""
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
