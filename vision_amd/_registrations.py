"""Python-level dispatcher registrations for the `torchvision::` ops that our library owns:
fake (meta) kernels, autograd formulas and autocast wrappers.

Role of torchvision/_meta_registrations.py and torchvision/_autograd_registrations.py in
the reference (which are reused UNCHANGED when the reference python package is laid over
this library, see INTEGRATION.md — call `vision_amd.load(register_python=False)` then).
Everything here is table-driven; shapes/dtypes follow the reference fake kernels
(_meta_registrations.py:25-231) and the backward op argument orders follow
_autograd_registrations.py:14-205.
"""
import torch
import torch.library

_done = {"v": False}


def _same_type(a, b, an, bn):
    torch._check(
        a.dtype == b.dtype,
        lambda: f"Expected tensor for {an} to have the same type as tensor for {bn}; "
        f"but type {a.dtype} does not equal {b.dtype}",
    )


def _rois_ok(rois):
    torch._check(rois.size(1) == 5, lambda: "rois must have shape as Tensor[K, 5]")


# --------------------------------------------------------------------- fake kernels
def _fake_nms(dets, scores, iou_threshold):
    torch._check(dets.dim() == 2, lambda: f"boxes should be a 2d tensor, got {dets.dim()}D")
    torch._check(dets.size(1) == 4, lambda: f"boxes should have 4 elements in dimension 1, got {dets.size(1)}")
    torch._check(scores.dim() == 1, lambda: f"scores should be a 1d tensor, got {scores.dim()}")
    torch._check(
        dets.size(0) == scores.size(0),
        lambda: "boxes and scores should have same number of elements in dimension 0, "
        f"got {dets.size(0)} and {scores.size(0)}",
    )
    n = torch.library.get_ctx().new_dynamic_size()
    return dets.new_empty(n, dtype=torch.long)


def _fake_nms_segmented(dets, scores, idxs, iou_threshold, num_segments=-1):
    return _fake_nms(dets, scores, iou_threshold)


def _fake_roi_align(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio, aligned):
    _rois_ok(rois)
    _same_type(input, rois, "input", "rois")
    return input.new_empty((rois.size(0), input.size(1), pooled_height, pooled_width))


def _fake_roi_align_bwd(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels, height,
                        width, sampling_ratio, aligned):
    _same_type(grad, rois, "grad", "rois")
    return grad.new_empty((batch_size, channels, height, width))


def _fake_roi_pool(input, rois, spatial_scale, pooled_height, pooled_width):
    _rois_ok(rois)
    _same_type(input, rois, "input", "rois")
    size = (rois.size(0), input.size(1), pooled_height, pooled_width)
    return input.new_empty(size), torch.empty(size, device=input.device, dtype=torch.int32)


def _fake_roi_pool_bwd(grad, rois, argmax, spatial_scale, pooled_height, pooled_width, batch_size, channels,
                       height, width):
    _same_type(grad, rois, "grad", "rois")
    return grad.new_empty((batch_size, channels, height, width))


def _fake_ps_roi_align(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio):
    _rois_ok(rois)
    _same_type(input, rois, "input", "rois")
    c = input.size(1)
    torch._check(c % (pooled_height * pooled_width) == 0,
                 lambda: "input channels must be a multiple of pooling height * pooling width")
    size = (rois.size(0), c // (pooled_height * pooled_width), pooled_height, pooled_width)
    return input.new_empty(size), torch.empty(size, device=input.device, dtype=torch.int32)


def _fake_ps_roi_align_bwd(grad, rois, channel_mapping, spatial_scale, pooled_height, pooled_width,
                           sampling_ratio, batch_size, channels, height, width):
    _same_type(grad, rois, "grad", "rois")
    return grad.new_empty((batch_size, channels, height, width))


def _fake_ps_roi_pool(input, rois, spatial_scale, pooled_height, pooled_width):
    return _fake_ps_roi_align(input, rois, spatial_scale, pooled_height, pooled_width, 0)


def _fake_ps_roi_pool_bwd(grad, rois, channel_mapping, spatial_scale, pooled_height, pooled_width, batch_size,
                          channels, height, width):
    _same_type(grad, rois, "grad", "rois")
    return grad.new_empty((batch_size, channels, height, width))


def _conv_out(size, pad, dil, k, stride):
    return (size + 2 * pad - (dil * (k - 1) + 1)) // stride + 1


def _fake_deform_conv2d(input, weight, offset, mask, bias, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w,
                        n_weight_grps, n_offset_grps, use_mask):
    out_h = _conv_out(input.size(2), pad_h, dil_h, weight.size(2), stride_h)
    out_w = _conv_out(input.size(3), pad_w, dil_w, weight.size(3), stride_w)
    return input.new_empty((input.size(0), weight.size(0), out_h, out_w))


def _fake_deform_conv2d_bwd(grad, input, weight, offset, mask, bias, stride_h, stride_w, pad_h, pad_w, dil_h,
                            dil_w, n_weight_grps, n_offset_grps, use_mask):
    return (input.new_empty(input.shape), weight.new_empty(weight.shape), offset.new_empty(offset.shape),
            mask.new_empty(mask.shape), bias.new_empty(bias.shape))


def _fake_box_iou_rotated(boxes1, boxes2):
    return boxes1.new_empty((boxes1.size(0), boxes2.size(0)), dtype=torch.float32)


def _fake_multiscale(features, rois, scales, pooled_height, pooled_width, sampling_ratio, aligned, k_min, k_max,
                     canonical_scale, canonical_level, eps):
    return features[0].new_empty((rois.size(0), features[0].size(1), pooled_height, pooled_width))


def _fake_multiscale_bwd(grad, rois, heights, widths, scales, batch_size, pooled_height, pooled_width, sampling_ratio,
                         aligned, k_min, k_max, canonical_scale, canonical_level, eps):
    return [grad.new_empty((batch_size, grad.size(1), h, w)) for h, w in zip(heights, widths)]


def _fake_multiscale_boxes(features, boxes, scales, pooled_height, pooled_width, sampling_ratio, aligned, k_min, k_max,
                           canonical_scale, canonical_level, eps):
    K = sum(b.shape[0] for b in boxes)
    f0 = features[0]
    return f0.new_empty((K, f0.shape[1], pooled_height, pooled_width)), f0.new_empty((K, 5), dtype=torch.float32)


def _multiscale_boxes_setup(ctx, inputs, output):
    features = inputs[0]
    ctx.save_for_backward(output[1])          # the [K,5] rows the pre-pass wrote
    ctx.set_materialize_grads(False)
    ctx.shapes = [tuple(f.shape) for f in features]
    ctx.n_boxes = len(inputs[1])
    ctx.params = inputs[2:]


def _multiscale_boxes_backward(ctx, grad, _grad_rois):
    (rois,) = ctx.saved_tensors
    scales, ph, pw, sr, aligned, k_min, k_max, s0, lvl0, eps = ctx.params
    if grad is None:
        return (None, None) + (None,) * len(ctx.params)
    grads = torch.ops.tvmi.multiscale_roi_align_backward(
        grad, rois, [s[2] for s in ctx.shapes], [s[3] for s in ctx.shapes], scales, ctx.shapes[0][0], ph, pw, sr, aligned,
        k_min, k_max, s0, lvl0, eps)
    return (list(grads), [None] * ctx.n_boxes) + (None,) * len(ctx.params)


def _multiscale_setup(ctx, inputs, output):
    features, rois = inputs[0], inputs[1]
    ctx.save_for_backward(rois)
    ctx.shapes = [tuple(f.shape) for f in features]
    ctx.params = inputs[2:]


def _multiscale_backward(ctx, grad):
    (rois,) = ctx.saved_tensors
    scales, ph, pw, sr, aligned, k_min, k_max, s0, lvl0, eps = ctx.params
    grads = torch.ops.tvmi.multiscale_roi_align_backward(
        grad, rois, [s[2] for s in ctx.shapes], [s[3] for s in ctx.shapes], scales, ctx.shapes[0][0], ph, pw, sr, aligned,
        k_min, k_max, s0, lvl0, eps)
    return (list(grads), None) + (None,) * len(ctx.params)


def _fake_interpolate2d(input, out_h, out_w, mode, align_corners, antialias, scale_h, scale_w):
    # the real op: channels_last in -> channels_last out for every dtype and size (torch_shim.cpp interpolate2d)
    cl = (not input.is_contiguous() and input.is_contiguous(memory_format=torch.channels_last) and input.numel() > 0
          and out_h > 0 and out_w > 0)
    return torch.empty((input.size(0), input.size(1), out_h, out_w), dtype=input.dtype, device=input.device,
                       memory_format=torch.channels_last if cl else torch.contiguous_format)


def _fake_interpolate2d_backward(grad_output, in_h, in_w, mode, align_corners, antialias, scale_h, scale_w):
    return grad_output.new_empty((grad_output.size(0), grad_output.size(1), in_h, in_w))


def _fake_pack_detections(boxes, scores, labels, image_idx, keep, num_images, max_dets):
    return (boxes.new_empty((num_images, max_dets, 6), dtype=torch.float32),
            boxes.new_empty((num_images,), dtype=torch.int32))


def _fake_paste_masks(masks, boxes, im_h, im_w, padding):
    return masks.new_empty((masks.shape[0], 1, im_h, im_w))


def _fake_detection_candidates(class_logits, box_regression, proposals, row_image, image_hw, weights, clip, score_thresh, min_size):
    R, C = class_logits.shape
    f = class_logits.new_empty
    return f((R, C - 1, 4), dtype=torch.float32), f((R, C - 1), dtype=torch.float32), f((R, C - 1), dtype=torch.uint8)


def _fake_rpn_candidates(objectness, boxes, deltas, top_idx, level_offsets, image_hw, clip, score_thresh, min_size):
    B, T = top_idx.shape
    f = objectness.new_empty
    return (f((B, T, 4), dtype=torch.float32), f((B, T), dtype=torch.float32), f((B, T), dtype=torch.int64),
            f((B, T), dtype=torch.uint8))


def _fake_nms_padded(dets, scores, idxs, iou_threshold, num_segments=-1):
    return dets.new_empty((dets.shape[0],), dtype=torch.int64), dets.new_empty((1,), dtype=torch.int64)


def _fake_nms_step(dets, scores, idxs, iou_threshold, num_segments, image_idx, labels, num_images, max_dets):
    return (dets.new_empty((dets.shape[0],), dtype=torch.int64), dets.new_empty((1,), dtype=torch.int64),
            dets.new_empty((num_images, max_dets * 6 + 1), dtype=torch.float32))


def _fake_roi_boxes_nms_step(features, boxes, scales, pooled_height, pooled_width, sampling_ratio, aligned, k_min, k_max, canonical_scale,
                             canonical_level, eps, dets, scores, idxs, iou_threshold, num_segments, image_idx, labels, num_images, max_dets):
    return (_fake_multiscale_boxes(features, boxes, scales, pooled_height, pooled_width, sampling_ratio, aligned, k_min, k_max,
                                   canonical_scale, canonical_level, eps)
            + _fake_nms_step(dets, scores, idxs, iou_threshold, num_segments, image_idx, labels, num_images, max_dets))


def _fake_nms_masked(dets, scores, idxs, valid, iou_threshold, num_segments=-1, max_segment_size=-1):
    return dets.new_empty((dets.shape[0],), dtype=torch.int64), dets.new_empty((1,), dtype=torch.int64)


def _fake_pack_payload(boxes, scores, labels, image_idx, keep, num_keep, num_images, max_dets):
    return boxes.new_empty((num_images, max_dets * 6 + 1), dtype=torch.float32)


def _fake_pack_devcount(boxes, scores, labels, image_idx, keep, num_keep, num_images, max_dets):
    return boxes.new_empty((num_images, max_dets, 6), dtype=torch.float32), boxes.new_empty((num_images,), dtype=torch.int32)


def _fake_box_iou_pairwise(boxes1, boxes2, mode, eps=1e-7):
    dt = torch.float64 if torch.float64 in (boxes1.dtype, boxes2.dtype) else torch.float32
    return boxes1.new_empty((boxes1.shape[0], boxes2.shape[0]), dtype=dt)


def _fake_normalize_resize_batch(images, out_heights, out_widths, mean, std, padded_h, padded_w):
    return images[0].new_empty((len(images), images[0].shape[0], padded_h, padded_w))


def _fake_boxes_to_rois(boxes):
    return boxes[0].new_empty((sum(b.shape[0] for b in boxes), 5))


def _fake_qnms(dets, scores, iou_threshold):
    # torchvision/_meta_registrations.py:177-188
    torch._check(dets.dim() == 2, lambda: f"boxes should be a 2d tensor, got {dets.dim()}D")
    torch._check(dets.size(1) == 4, lambda: f"boxes should have 4 elements in dimension 1, got {dets.size(1)}")
    torch._check(scores.dim() == 1, lambda: f"scores should be a 1d tensor, got {scores.dim()}")
    torch._check(dets.size(0) == scores.size(0),
                 lambda: f"boxes and scores should have same number of elements in dimension 0, got {dets.size(0)} and {scores.size(0)}")
    ctx = torch.library.get_ctx()
    return dets.new_empty(ctx.new_dynamic_size(), dtype=torch.long)


def _fake_qroi_align(input, rois, input_scale, input_zero_point, rois_scale, rois_zero_point, spatial_scale, pooled_height,
                     pooled_width, sampling_ratio, aligned):
    # torchvision/_meta_registrations.py:191-215
    torch._check(rois.size(1) == 5, lambda: "rois must have shape as Tensor[K, 5]")
    torch._check(input.dtype == rois.dtype,
                 lambda: f"Expected tensor for input to have the same type as tensor for rois; but type {input.dtype} does not equal {rois.dtype}")
    return input.new_empty((rois.size(0), input.size(1), pooled_height, pooled_width))


_FAKES = {
    "tvmi::boxes_to_rois": _fake_boxes_to_rois,
    "tvmi::sort_scores_desc": lambda scores: scores.new_empty(scores.shape, dtype=torch.int64),
    "tvmi::normalize_resize_batch": _fake_normalize_resize_batch,
    "tvmi::box_iou_pairwise": _fake_box_iou_pairwise,
    "tvmi::nms_segmented_padded": _fake_nms_padded,
    "tvmi::nms_segmented_masked": _fake_nms_masked,
    "tvmi::nms_step": _fake_nms_step,
    "tvmi::pack_detections_payload": _fake_pack_payload,
    "tvmi::pack_detections_devcount": _fake_pack_devcount,
    "tvmi::paste_masks": _fake_paste_masks,
    "tvmi::detection_candidates": _fake_detection_candidates,
    "tvmi::rpn_candidates": _fake_rpn_candidates,
    "tvmi::pack_detections": _fake_pack_detections,
    "tvmi::multiscale_roi_align": _fake_multiscale,
    "tvmi::multiscale_roi_align_boxes": _fake_multiscale_boxes,
    "tvmi::roi_align_boxes_nms_step": _fake_roi_boxes_nms_step,
    "tvmi::multiscale_roi_align_backward": _fake_multiscale_bwd,
    "tvmi::interpolate2d": _fake_interpolate2d,
    "tvmi::interpolate2d_backward": _fake_interpolate2d_backward,
    "torchvision::nms": _fake_nms,
    "tvmi::nms_segmented": _fake_nms_segmented,
    "torchvision::roi_align": _fake_roi_align,
    "torchvision::_roi_align_backward": _fake_roi_align_bwd,
    "torchvision::roi_pool": _fake_roi_pool,
    "torchvision::_roi_pool_backward": _fake_roi_pool_bwd,
    "torchvision::ps_roi_align": _fake_ps_roi_align,
    "torchvision::_ps_roi_align_backward": _fake_ps_roi_align_bwd,
    "torchvision::ps_roi_pool": _fake_ps_roi_pool,
    "torchvision::_ps_roi_pool_backward": _fake_ps_roi_pool_bwd,
    "torchvision::deform_conv2d": _fake_deform_conv2d,
    "torchvision::_deform_conv2d_backward": _fake_deform_conv2d_bwd,
    "torchvision::box_iou_rotated": _fake_box_iou_rotated,
    "torchvision::qnms": _fake_qnms,
    "torchvision::qroi_align": _fake_qroi_align,
}


# --------------------------------------------------------------------- autograd formulas
# name -> (backward op, indices of scalar forward args appended to the backward call in the
#          order the backward schema wants them, saved-output index or None)
def _make_roi_autograd(fwd_name, bwd_name, n_scalar_before_shape, tail_idx, save_aux):
    """Builds (setup_context, backward) for the RoI pooling family.

    Forward args are (input, rois, *scalars); the backward op takes
    (grad, rois, [aux], *scalars[:n_scalar_before_shape], N, C, H, W, *scalars[tail_idx]).
    """
    bwd_op = getattr(torch.ops.torchvision, bwd_name)

    def setup_context(ctx, inputs, output):
        inp, rois = inputs[0], inputs[1]
        ctx.scalars = tuple(inputs[2:])
        ctx.in_shape = tuple(inp.shape)
        if save_aux:
            ctx.save_for_backward(rois, output[1])
        else:
            ctx.save_for_backward(rois)

    def backward(ctx, grad, *unused):
        saved = ctx.saved_tensors
        sc = ctx.scalars
        args = [grad, saved[0]]
        if save_aux:
            args.append(saved[1])
        args += list(sc[:n_scalar_before_shape])
        args += list(ctx.in_shape)
        args += [sc[i] for i in tail_idx]
        return (bwd_op(*args),) + (None,) * (1 + len(sc))

    return setup_context, backward


def _deform_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs[:5])
    ctx.params = tuple(inputs[5:])


def _deform_backward(ctx, grad):
    inp, weight, offset, mask, bias = ctx.saved_tensors
    grads = torch.ops.torchvision._deform_conv2d_backward(grad, inp, weight, offset, mask, bias, *ctx.params)
    return tuple(grads) + (None,) * len(ctx.params)


def _no_double_backward(name):
    def backward(ctx, *grads):
        raise RuntimeError(f"double backwards on {name} not supported")

    return backward


# --------------------------------------------------------------------- autocast
_AUTOCAST_KEYS = None


def _fp32(t):
    # at::autocast::cached_cast(kFloat, ...): only lower-precision floating tensors are cast
    if isinstance(t, torch.Tensor) and t.is_floating_point() and t.dtype is not torch.float64:
        return t.float()
    return t


def _make_autocast(op_name, n_tensor_args, restore):
    op = getattr(torch.ops.torchvision, op_name)

    def wrapper(*args):
        orig = args[0].dtype
        with torch._C._ExcludeDispatchKeyGuard(_AUTOCAST_KEYS):
            out = op(*[_fp32(a) for a in args[:n_tensor_args]], *args[n_tensor_args:])
        if not restore:
            return out
        if isinstance(out, tuple):
            return tuple(o.to(orig) for o in out)
        return out.to(orig)

    return wrapper


def register_all(torchvision_schemas: bool = True):
    """Idempotent.  `torchvision_schemas=False` registers only what belongs to the `tvmi::` namespace (fake kernels,
    the autograd formula of the fused multi-scale op): the mode for processes in which the reference's python package
    is imported over this library and brings its own registrations for the `torchvision::` schemas."""
    global _AUTOCAST_KEYS
    if _done["v"]:
        return
    _done["v"] = True
    for name, fn in _FAKES.items():
        if torchvision_schemas or name.startswith("tvmi::"):
            torch.library.register_fake(name, fn)
    torch.library.register_autograd("tvmi::multiscale_roi_align", _multiscale_backward, setup_context=_multiscale_setup)
    torch.library.register_autograd("tvmi::multiscale_roi_align_boxes", _multiscale_boxes_backward, setup_context=_multiscale_boxes_setup)
    torch.library.register_autograd("tvmi::multiscale_roi_align_backward", _no_double_backward("multiscale_roi_align"))
    if not torchvision_schemas:
        return

    tv = "torchvision::"
    for fwd, bwd, nb, tail, aux in (
        ("roi_align", "_roi_align_backward", 3, (3, 4), False),
        ("roi_pool", "_roi_pool_backward", 3, (), True),
        ("ps_roi_align", "_ps_roi_align_backward", 4, (), True),
        ("ps_roi_pool", "_ps_roi_pool_backward", 3, (), True),
    ):
        setup, backward = _make_roi_autograd(fwd, bwd, nb, tail, aux)
        torch.library.register_autograd(tv + fwd, backward, setup_context=setup)
        torch.library.register_autograd(tv + bwd, _no_double_backward(fwd))
    torch.library.register_autograd(tv + "deform_conv2d", _deform_backward, setup_context=_deform_setup)
    torch.library.register_autograd(tv + "_deform_conv2d_backward", _no_double_backward("deform_conv2d"))

    _AUTOCAST_KEYS = torch._C.DispatchKeySet(torch._C.DispatchKey.AutocastCUDA) | torch._C.DispatchKeySet(
        torch._C.DispatchKey.AutocastCPU
    )
    lib = torch.library.Library("torchvision", "IMPL")
    _done["lib"] = lib  # keep alive
    for key in ("AutocastCUDA", "AutocastCPU"):
        lib.impl("nms", _make_autocast("nms", 2, restore=False), key)
        lib.impl("roi_align", _make_autocast("roi_align", 2, restore=True), key)
    lib.impl("roi_pool", _make_autocast("roi_pool", 2, restore=True), "AutocastCUDA")
    lib.impl("ps_roi_align", _make_autocast("ps_roi_align", 2, restore=True), "AutocastCUDA")
    lib.impl("ps_roi_pool", _make_autocast("ps_roi_pool", 2, restore=True), "AutocastCUDA")
    lib.impl("deform_conv2d", _make_autocast("deform_conv2d", 5, restore=True), "AutocastCUDA")
