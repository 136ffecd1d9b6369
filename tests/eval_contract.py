"""indoor_eval against the fixture made by the reference's own indoor_eval (tests/golden/make_golden_eval.py)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "indoor_eval.npz")


def check_case(ci):
    from sgcdet_amd.evaluation import indoor_eval
    d = np.load(GOLDEN)
    k = f"case{ci}_"
    n_scenes, n_cls = int(d[k + "n_scenes"]), int(d[k + "n_cls"])
    gts, dts = [], []
    for s in range(n_scenes):
        gb = d[k + f"gt_boxes{s}"]
        gts.append({"gt_num": len(gb), "gt_boxes_upright_depth": gb, "class": d[k + f"gt_cls{s}"]})
        dts.append(dict(boxes_3d=torch.from_numpy(d[k + f"dt_boxes{s}"]), scores_3d=torch.from_numpy(d[k + f"dt_scores{s}"]),
                        labels_3d=torch.from_numpy(d[k + f"dt_labels{s}"])))
    res = indoor_eval(gts, dts, [float(t) for t in d["metric"]], {i: f"c{i}" for i in range(n_cls)})
    keys = [str(x) for x in d[k + "keys"]]
    assert sorted(res.keys()) == sorted(keys), set(keys) ^ set(res.keys())
    for key, want in zip(keys, d[k + "values"]):
        assert abs(res[key] - want) < 1e-6, (key, res[key], want)
    assert 0.05 < res["mAP_0.25"] < 0.95 and res["mAP_0.50"] < res["mAP_0.25"]
