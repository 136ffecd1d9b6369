"""Known-answer vectors of the REFERENCE'S OWN TESTS for the post-processing / training-side rows of the path
(packages/mmdetection3d/tests/test_utils/test_nms.py, tests/test_metrics/test_losses.py, tests/test_metrics/
test_indoor_eval.py; extracted as data by tests/golden/make_golden_ref_kats.py): the oracle and the host-side functions on
the CPU, the HIP kernels on the GPU."""
import os

import numpy as np
import pytest
import torch

KATS = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_kats.npz"))


def _t(name):
    return torch.from_numpy(np.asarray(KATS[name]))


def _check_aligned_nms(ops, device):
    pick = ops.aligned_nms3d(_t("nms3d_boxes").to(device), _t("nms3d_scores").to(device), _t("nms3d_cls").to(device),
                             float(KATS["nms3d_thr"]))
    assert pick.cpu().tolist() == KATS["nms3d_expected"].tolist()          # test_aligned_3d_nms: same picks, same order


def _check_nms_bev(ops, device):
    """test_nms_bev: nms_bev(boxes xyxyr, scores, thresh=0.3) -> [1, 0, 3] -- the one-class case of the multi-class kernel"""
    keep, n_keep = ops.nms_rotated_bev(_t("nmsbev_boxes").to(device), _t("nmsbev_scores").to(device)[:, None].contiguous(), 0.0,
                                       float(KATS["nmsbev_thr"]))
    assert keep[0, :int(n_keep[0])].cpu().tolist() == KATS["nmsbev_expected"].tolist()


def test_oracle_nms_matches_the_reference_tests_known_answers(oracle_ops):
    _check_aligned_nms(oracle_ops, "cpu")
    _check_nms_bev(oracle_ops, "cpu")


@pytest.mark.gpu
def test_hip_nms_matches_the_reference_tests_known_answers(gpu_ops):
    _check_aligned_nms(gpu_ops, "cuda")
    _check_nms_bev(gpu_ops, "cuda")


def test_iou_losses_match_the_reference_tests_known_answers():
    """test_axis_aligned_iou_loss (1 - IoU = 0, 14/15, 1) and test_rotated_iou_3d_loss (1 - [1, .5, .7071, 1/15, 0])"""
    from sgcdet_amd.plugin import losses
    aa = 1 - losses.axis_aligned_iou(_t("aaloss_boxes1"), _t("aaloss_boxes2"))
    assert torch.allclose(aa, _t("aaloss_expected")[0], atol=1e-4)
    rot = 1 - losses.rotated_iou_3d(_t("rotloss_boxes1"), _t("rotloss_boxes2"))
    assert torch.allclose(rot, _t("rotloss_expected")[0], atol=1e-4)
    # the reduced form the heads call (mean over boxes)
    assert abs(float(losses.axis_aligned_iou_loss(_t("aaloss_boxes1"), _t("aaloss_boxes2"))) - float(_t("aaloss_expected").mean())) < 1e-4
    assert abs(float(losses.rotated_iou_3d_loss(_t("rotloss_boxes1"), _t("rotloss_boxes2"))) - float(_t("rotloss_expected").mean())) < 1e-4
    # as the reference's tests call them: the registered classes with reduction='none' (test_losses.py:188, :210)
    from sgcdet_amd.mmcv_lite import LOSSES
    got = LOSSES.build(dict(type="AxisAlignedIoULoss", reduction="none"))(_t("aaloss_boxes1"), _t("aaloss_boxes2"))
    assert got.shape == (3,) and torch.allclose(got, _t("aaloss_expected").reshape(-1), atol=1e-4)
    got = LOSSES.build(dict(type="RotatedIoU3DLoss", reduction="none"))(_t("rotloss_boxes1"), _t("rotloss_boxes2"))
    assert got.shape == (5,) and torch.allclose(got, _t("rotloss_expected").reshape(-1), atol=1e-4)


def _eval_inputs(tag):
    """detections are bottom-centre rows of a DepthInstance3DBoxes(origin=(.5,.5,0)); ground truth is gravity-centre rows
    (core/evaluation/indoor_eval.py:248-256): sgcdet_amd.evaluation takes gravity-centre rows for both"""
    dets, gts = [], []
    for i in range(int(KATS[f"{tag}_n_scenes"])):
        b = _t(f"{tag}_det{i}_boxes_bottom_center").clone()
        assert KATS[f"{tag}_det{i}_origin"].tolist() == [0.5, 0.5, 0.0]
        b[:, 2] += b[:, 5] / 2
        dets.append(dict(boxes_3d=b, labels_3d=_t(f"{tag}_det{i}_labels"), scores_3d=_t(f"{tag}_det{i}_scores")))
        g = KATS[f"{tag}_gt{i}_boxes_gravity_center"]
        gts.append(dict(gt_num=len(g), gt_boxes_upright_depth=g, **{"class": KATS[f"{tag}_gt{i}_class"]}))
    label2cat = {int(k): str(v) for k, v in zip(KATS[f"{tag}_label_ids"], KATS[f"{tag}_label_names"])}
    expected = {str(k): float(v) for k, v in zip(KATS[f"{tag}_expected_keys"], KATS[f"{tag}_expected_values"])}
    return gts, dets, label2cat, expected


def _check_indoor_eval(tag, device):
    from sgcdet_amd.evaluation import indoor_eval
    gts, dets, label2cat, expected = _eval_inputs(tag)
    for d in dets:
        d["boxes_3d"] = d["boxes_3d"].to(device)
    ret = indoor_eval(gts, dets, [0.25, 0.5], label2cat)
    for k, v in expected.items():
        assert np.isclose(ret[k], v, atol=1e-5), (k, ret[k], v)


def test_indoor_eval_matches_the_reference_tests_known_answers():
    """test_indoor_eval: cabinet / bed / chair AP, mAP, mAR at 0.25 (upright boxes)"""
    _check_indoor_eval("ev1", "cpu")


@pytest.mark.gpu
def test_indoor_eval_less_classes_known_answers_with_the_hip_rotated_iou(gpu_ops):
    """test_indoor_eval_less_classes: boxes with yaw = 1 rad -> the rotated IoU runs on sgc_box_iou_rotated"""
    _check_indoor_eval("ev2", "cuda")
    _check_indoor_eval("ev1", "cuda")


def test_average_precision_11_points_known_answer():
    from sgcdet_amd.evaluation import average_precision
    ap = average_precision(KATS["ap11_recalls"], KATS["ap11_precisions"], "11points")
    assert abs(ap[0] - float(KATS["ap11_expected"])) < 0.001
