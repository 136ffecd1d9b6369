"""Known-answer vectors held by the REFERENCE'S OWN TESTS for the post-processing / training-side rows of the path
(SURVEY.md 8 f-3, f-4), extracted as data:

  packages/mmdetection3d/tests/test_utils/test_nms.py        test_aligned_3d_nms, test_nms_bev
  packages/mmdetection3d/tests/test_metrics/test_losses.py   test_axis_aligned_iou_loss, test_rotated_iou_3d_loss
  packages/mmdetection3d/tests/test_metrics/test_indoor_eval.py   test_indoor_eval, test_indoor_eval_less_classes, test_average_precision

Those tests need mmdet3d / mmcv / CUDA, none of which exists here, so they cannot be run; what they pin is their literal
inputs and expected outputs.  This script (build container only: it reads /root/reference) parses the test files with
`ast`, evaluates the literal tensor / array expressions and the expected values of the assertions, and writes
tests/golden/ref_kats.npz -- numbers only, no source text.  Box conventions are recorded as the tests state them
(`DepthInstance3DBoxes(t, origin=(0.5, 0.5, 0))` = bottom-centre rows; `gt_boxes_upright_depth` = gravity-centre rows,
core/evaluation/indoor_eval.py:248-256).
"""
import ast
import os
import sys

import numpy as np
import torch

REF = "/root/reference/packages/mmdetection3d/tests"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_kats.npz")


class _Box:                                   # stands in for DepthInstance3DBoxes while the literals are evaluated
    def __init__(self, tensor, origin=(0.5, 0.5, 0), **kwargs):
        self.tensor, self.origin = torch.as_tensor(tensor, dtype=torch.float32), tuple(origin)


class _Cuda(ast.NodeTransformer):             # x.cuda() -> x
    def visit_Call(self, node):
        self.generic_visit(node)
        if isinstance(node.func, ast.Attribute) and node.func.attr == "cuda" and not node.args:
            return node.func.value
        return node


def _functions(path):
    tree = ast.parse(open(path).read())
    return {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}


def _eval(node, env):
    node = _Cuda().visit(node)
    ast.fix_missing_locations(node)
    return eval(compile(ast.Expression(node), "<kat>", "eval"), env)


def _assignments(fn, env, names):
    """evaluate `name = <literal expression>` statements of a test function, in order"""
    out = {}
    for st in fn.body:
        if isinstance(st, ast.Assign) and len(st.targets) == 1 and isinstance(st.targets[0], ast.Name) and st.targets[0].id in names:
            out[st.targets[0].id] = _eval(st.value, dict(env, **out))
    return out


def _isclose_asserts(fn):
    """assert np.isclose(ret_value['key'], value) -> {key: value}"""
    exp = {}
    for st in ast.walk(fn):
        if isinstance(st, ast.Assert) and isinstance(st.test, ast.Call) and getattr(st.test.func, "attr", "") == "isclose":
            a, b = st.test.args[:2]
            if isinstance(a, ast.Subscript):
                exp[ast.literal_eval(a.slice)] = float(ast.literal_eval(b))
    return exp


def _np(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def main():
    env = dict(torch=torch, np=np, DepthInstance3DBoxes=_Box)
    out = {}
    f = _functions(os.path.join(REF, "test_utils/test_nms.py"))
    a = _assignments(f["test_aligned_3d_nms"], env, {"boxes", "scores", "cls", "expected_pick"})
    out.update(nms3d_boxes=_np(a["boxes"]), nms3d_scores=_np(a["scores"]), nms3d_cls=_np(a["cls"]), nms3d_thr=np.float32(0.25),
               nms3d_expected=_np(a["expected_pick"]))
    a = _assignments(f["test_nms_bev"], env, {"np_boxes", "np_scores", "np_inds"})
    out.update(nmsbev_boxes=a["np_boxes"], nmsbev_scores=a["np_scores"], nmsbev_thr=np.float32(0.3), nmsbev_expected=a["np_inds"])

    f = _functions(os.path.join(REF, "test_metrics/test_losses.py"))
    a = _assignments(f["test_axis_aligned_iou_loss"], env, {"boxes1", "boxes2", "expect_ious"})
    out.update(aaloss_boxes1=_np(a["boxes1"]), aaloss_boxes2=_np(a["boxes2"]), aaloss_expected=_np(a["expect_ious"]))
    a = _assignments(f["test_rotated_iou_3d_loss"], env, {"boxes1", "boxes2", "expect_ious"})
    out.update(rotloss_boxes1=_np(a["boxes1"]), rotloss_boxes2=_np(a["boxes2"]), rotloss_expected=_np(a["expect_ious"]))

    f = _functions(os.path.join(REF, "test_metrics/test_indoor_eval.py"))
    for tag, name in (("ev1", "test_indoor_eval"), ("ev2", "test_indoor_eval_less_classes")):
        a = _assignments(f[name], env, {"det_infos", "label2cat", "gt_annos"})
        exp = _isclose_asserts(f[name])
        out[f"{tag}_n_scenes"] = np.int64(len(a["det_infos"]))
        for i, (det, gt) in enumerate(zip(a["det_infos"], a["gt_annos"])):
            box = det["boxes_3d"]
            out[f"{tag}_det{i}_boxes_bottom_center"] = _np(box.tensor)
            out[f"{tag}_det{i}_origin"] = np.asarray(box.origin, dtype=np.float32)
            out[f"{tag}_det{i}_labels"] = _np(det["labels_3d"])
            out[f"{tag}_det{i}_scores"] = _np(det["scores_3d"])
            out[f"{tag}_gt{i}_boxes_gravity_center"] = np.asarray(gt["gt_boxes_upright_depth"], dtype=np.float32)
            out[f"{tag}_gt{i}_class"] = np.asarray(gt["class"])
        out[f"{tag}_label_ids"] = np.asarray(sorted(a["label2cat"]))
        out[f"{tag}_label_names"] = np.asarray([a["label2cat"][k] for k in sorted(a["label2cat"])])
        out[f"{tag}_expected_keys"] = np.asarray(list(exp))
        out[f"{tag}_expected_values"] = np.asarray([exp[k] for k in exp], dtype=np.float64)
    # test_average_precision: recalls / precisions -> AP with the '11points' mode, expected 0.06611571 (tolerance 0.001)
    call = next(n for n in ast.walk(f["test_average_precision"]) if isinstance(n, ast.Call) and getattr(n.func, "id", "") == "average_precision")
    out.update(ap11_recalls=_eval(call.args[0], env), ap11_precisions=_eval(call.args[1], env), ap11_expected=np.float64(0.06611571))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: getattr(v, "shape", None) for k, v in out.items() if k.endswith("expected") or k.endswith("values")})


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("the reference tree only exists in the build container")
    main()
