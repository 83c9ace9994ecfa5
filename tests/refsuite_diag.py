#!/usr/bin/env python3
"""Diagnosis for the two reference-suite failures seen on the MI355X (profiles/r06_reference_test_models.log):
test_models.py::test_detection_model[cuda-fasterrcnn_resnet50_fpn{,_v2}] — eager vs TorchScript boxes differ in 1 of 400
elements by 2e-4 relative (bar 1e-4).  Runs inside the overlay (PYTHONPATH set by tests/run_reference_tests.py --diag):
the model of the test, eager twice, scripted twice, and stage by stage (backbone features, RPN proposals, box head) to
find the first stage at which the two executions differ.  TEST INFRASTRUCTURE."""
import sys

import torch
import torchvision
from torchvision import models


def run(name):
    sys.path.insert(0, ".")
    import test_models as T  # the reference's own file: _model_params, _get_image

    torch.manual_seed(0)
    kwargs = {"num_classes": 50, "weights_backbone": None, **T._model_params.get(name, {})}
    shape, real = kwargs.pop("input_shape"), kwargs.pop("real_image", False)
    model = models.get_model_builder(name)(**kwargs).eval().cuda()
    x = T._get_image(input_shape=shape, real_image=real, device="cuda", dtype=torch.float32)
    sm = torch.jit.script(model).eval()

    def d(a, b):
        if a.shape != b.shape:
            return f"shape {tuple(a.shape)} vs {tuple(b.shape)}"
        return f"{float((a.float() - b.float()).abs().max()):.3e}"

    with torch.no_grad():
        e1, e2 = model([x])[0], model([x])[0]
        s1, s2 = sm([x])[1][0], sm([x])[1][0]
        print(name, "eager1 vs eager2 boxes", d(e1["boxes"], e2["boxes"]), "| script1 vs script2", d(s1["boxes"], s2["boxes"]),
              "| eager vs script1", d(e1["boxes"], s1["boxes"]), "| eager vs script2", d(e1["boxes"], s2["boxes"]),
              "| scores", d(e1["scores"], s1["scores"]))
        # stage by stage
        il_e, _ = model.transform([x])
        il_s, _ = sm.transform([x], None)
        print("  transform", d(il_e.tensors, il_s.tensors))
        f_e, f_s = model.backbone(il_e.tensors), sm.backbone(il_e.tensors)
        print("  backbone ", {k: d(f_e[k], f_s[k]) for k in f_e})
        f_e2 = model.backbone(il_e.tensors)
        print("  backbone eager again", {k: d(f_e[k], f_e2[k]) for k in f_e})
        p_e, _ = model.rpn(il_e, f_e)
        p_s, _ = sm.rpn(il_e, f_e, None)
        print("  rpn proposals (same features)", d(p_e[0], p_s[0]))
        r_e, _ = model.roi_heads(f_e, p_e, il_e.image_sizes)
        r_s, _ = sm.roi_heads(f_e, p_e, il_e.image_sizes, None)
        print("  roi_heads (same features + proposals) boxes", d(r_e[0]["boxes"], r_s[0]["boxes"]), "scores", d(r_e[0]["scores"], r_s[0]["scores"]))
        bf_e = model.roi_heads.box_roi_pool(f_e, p_e, il_e.image_sizes)
        bf_s = sm.roi_heads.box_roi_pool(f_e, p_e, il_e.image_sizes)
        print("  box_roi_pool", d(bf_e, bf_s))
        h_e = model.roi_heads.box_predictor(model.roi_heads.box_head(bf_e))
        h_s = sm.roi_heads.box_predictor(sm.roi_heads.box_head(bf_e))
        print("  box head logits", d(h_e[0], h_s[0]), "regression", d(h_e[1], h_s[1]))
        pd_e = model.roi_heads.postprocess_detections(h_e[0], h_e[1], p_e, il_e.image_sizes)
        pd_s = sm.roi_heads.postprocess_detections(h_e[0], h_e[1], p_e, il_e.image_sizes)
        print("  postprocess_detections (same head outputs) boxes", d(pd_e[0][0], pd_s[0][0]))
        dec_e = model.roi_heads.box_coder.decode(h_e[1], p_e)
        dec_s = sm.roi_heads.box_coder.decode(h_e[1], p_e)
        print("  box_coder.decode", d(dec_e, dec_s), " max |box|", float(dec_e.abs().max()))


if __name__ == "__main__":
    for n in sys.argv[1:] or ["fasterrcnn_resnet50_fpn", "fasterrcnn_resnet50_fpn_v2"]:
        run(n)
