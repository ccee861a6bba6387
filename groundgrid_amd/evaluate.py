"""Ground-segmentation evaluator: the metric definitions of the reference's evaluation node
(/root/reference/scripts/eval_groundpoint_classifier.py:62-78, :95-132, :135-195), without ROS.

The reference's evaluator subscribes to the segmented cloud, in which `intensity` is 49 (predicted ground) or 99
(predicted non-ground) and `ring` carries the SemanticKITTI label the player put there
(scripts/kitti_data_publisher.py:117-153).  Here the same counting runs on arrays.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

# cfg/semantic-kitti-all.yaml `labels:` (the only section the evaluator reads, eval...py:66,106)
LABELS: Dict[int, str] = {
    0: "unlabeled", 1: "outlier", 10: "car", 11: "bicycle", 13: "bus", 15: "motorcycle", 16: "on-rails", 18: "truck",
    20: "other-vehicle", 30: "person", 31: "bicyclist", 32: "motorcyclist", 40: "road", 44: "parking", 48: "sidewalk",
    49: "other-ground", 50: "building", 51: "fence", 52: "other-structure", 60: "lane-marking", 70: "vegetation", 71: "trunk",
    72: "terrain", 80: "pole", 81: "traffic-sign", 99: "other-object", 252: "moving-car", 253: "moving-bicyclist",
    254: "moving-person", 255: "moving-motorcyclist", 256: "moving-on-rails", 257: "moving-bus", 258: "moving-truck",
    259: "moving-other-vehicle",
}
# eval_groundpoint_classifier.py:74-78 (vegetation, unlabeled, outlier are in none of the lists -> excluded from P/R/F1)
GROUND_LABELS = ["road", "sidewalk", "parking", "lane-marking"]
ADDITIONAL_GROUND_LABELS = ["other-ground", "terrain"]
NON_GROUND_LABELS = ["bicycle", "moving-bicyclist", "motorcycle", "moving-motorcyclist", "person", "moving-person", "traffic-sign",
                     "car", "moving-car", "motorcyclist", "bicyclist", "truck", "moving-truck", "building", "fence", "trunk", "pole",
                     "bus", "on-rails", "other-vehicle", "other-structure", "other-object", "moving-on-rails", "moving-bus",
                     "moving-other-vehicle"]

GROUND, NONGROUND = 49, 99


class GroundEvaluator:
    def __init__(self):
        names = list(LABELS.values())
        self.non_ground = {n: 0 for n in names}       # nonGroundPointLabelCount
        self.total = {n: 0 for n in names}            # semanticCloudLabelCount
        self.true_positive = {n: 0 for n in names}    # truePositiveCloudLabelCount
        self.false_positive = {n: 0 for n in names}   # falsePositiveCloudLabelCount
        self.cloud_count = 0

    def add_cloud(self, predicted: np.ndarray, semantic: np.ndarray):
        """predicted: 49 / 99 per point of the RETURNED cloud; semantic: SemanticKITTI label id (the `ring` field)."""
        predicted = np.asarray(predicted)
        semantic = np.asarray(semantic).astype(np.int64)
        for lid in np.unique(semantic):
            name = LABELS[int(lid)]  # KeyError for ids outside the yaml, like the reference
            sel = semantic == lid
            ng = int(np.count_nonzero(predicted[sel] == NONGROUND))
            gr = int(np.count_nonzero(predicted[sel] == GROUND))
            self.non_ground[name] += ng                      # :108-109
            if name in GROUND_LABELS or name in ADDITIONAL_GROUND_LABELS:
                self.true_positive[name] += gr               # :110-114
            else:
                self.false_positive[name] += gr              # :115-116
            self.total[name] += int(np.count_nonzero(sel))   # :118
        self.cloud_count += 1

    @classmethod
    def from_counts(cls, non_ground: Dict[str, int], total: Dict[str, int]) -> "GroundEvaluator":
        """Rebuild the counters from a printed table (non-ground and total per label)."""
        ev = cls()
        for name, tot in total.items():
            ng = non_ground[name]
            ev.non_ground[name], ev.total[name] = ng, tot
            if name in GROUND_LABELS or name in ADDITIONAL_GROUND_LABELS:
                ev.true_positive[name] = tot - ng
            else:
                ev.false_positive[name] = tot - ng
        return ev

    def summary(self) -> dict:
        """eval_groundpoint_classifier.py:153-195."""
        tp = sum(self.true_positive[n] for n in GROUND_LABELS + ADDITIONAL_GROUND_LABELS)
        gt_ground = sum(self.total[n] for n in GROUND_LABELS + ADDITIONAL_GROUND_LABELS)
        fn = sum(self.non_ground[n] for n in GROUND_LABELS + ADDITIONAL_GROUND_LABELS)
        fp = sum(self.false_positive[n] for n in NON_GROUND_LABELS)
        tn = sum(self.non_ground[n] for n in NON_GROUND_LABELS)
        div = lambda a, b: a / b if b else float("nan")  # noqa: E731
        return {
            "clouds": self.cloud_count,
            "TP": tp, "FP": fp, "FN": fn, "TN": tn,
            "precision": div(tp, fp + tp),
            "recall": div(tp, fn + tp),
            "f1": div(2 * tp, 2 * tp + fp + fn),
            "accuracy": div(tp + tn, tp + tn + fp + fn),
            "iou_ground": div(tp, fp + gt_ground),
        }

    def rows(self) -> dict:
        """The table as data: per label (non-ground %, ground %, non-ground, total) for labels that occurred, + the summary."""
        out = {}
        for name in LABELS.values():
            tot = self.total[name]
            if tot:
                ng = self.non_ground[name]
                out[name] = {"nonground_pct": round(100.0 * ng / tot, 2), "ground_pct": round(100.0 * (1.0 - ng / tot), 2), "nonground": ng, "total": tot}
        return {"clouds": self.cloud_count, "labels": out, "summary": {k: (round(100.0 * v, 2) if isinstance(v, float) else v) for k, v in self.summary().items()}}

    def compare_with(self, published: dict, tolerance_pct: float = 1.0) -> dict:
        """Difference to a published table of the form tests/golden/readme_seq00_table.json (the reference's README.md:57-94):
        per label the non-ground percentage (absolute difference in percentage points) and the point total (relative), and the
        five summary percentages.  `within_tolerance` = every percentage within tolerance_pct points and every total within
        tolerance_pct percent -- a close-match criterion: the live ROS pipeline's tf timing is not part of the data."""
        mine = self.rows()
        diff, ok = {}, self.cloud_count == published.get("clouds", self.cloud_count)
        for name, row in published["labels"].items():
            got = mine["labels"].get(name)
            if got is None:
                diff[name] = {"missing": True}
                ok = False
                continue
            d_pct = got["nonground_pct"] - row["nonground_pct"]
            d_tot = 100.0 * (got["total"] - row["total"]) / row["total"]
            diff[name] = {"nonground_pct_delta": round(d_pct, 3), "total_rel_pct": round(d_tot, 3)}
            ok &= abs(d_pct) <= tolerance_pct and abs(d_tot) <= tolerance_pct
        summary = {}
        for key, name in (("precision", "Precision"), ("recall", "Recall"), ("f1", "F1"), ("accuracy", "Accuracy"), ("iou_ground", "IoUg")):
            d = mine["summary"][key] - published["summary"][name][0]
            summary[name] = {"mine": mine["summary"][key], "published": published["summary"][name][0], "delta": round(d, 3)}
            ok &= abs(d) <= tolerance_pct
        return {"clouds": self.cloud_count, "published_clouds": published.get("clouds"), "tolerance_pct": tolerance_pct,
                "within_tolerance": bool(ok), "labels": diff, "summary": summary}

    def table(self) -> str:
        """The text eval_groundpoint_classifier.py:135-195 prints."""
        lines = ["Stats", f"Received {self.cloud_count} point clouds.", "label\t\t\tnonground %\tground %\tnonground\ttotal"]
        for name in LABELS.values():
            tot = self.total[name]
            if tot == 0:
                continue
            ng = self.non_ground[name]
            lab = name + ("\t" if len(name) < 8 else "")
            lab += "\t" if len(lab) < 16 else ""
            lines.append(lab + "\t{:2.2%}\t\t{:2.2%}\t\t".format(ng / tot, 1.0 - ng / tot) + str(ng) + "\t\t" + str(tot))
        s = self.summary()
        lines.append("Precision\t\t{:2.2%}\t\t{:0}\t{:0}".format(s["precision"], s["TP"], s["FP"]))
        lines.append("Recall\t\t\t{:2.2%}\t\t{:0}\t{:0}".format(s["recall"], s["TP"], s["FN"]))
        lines.append("F1\t\t\t{:2.2%}\t\t{:0}\t\t{:0}".format(s["f1"], s["FP"], s["FN"]))
        lines.append("Accuracy\t\t{:2.2%}\t\t{:0}\t{:0}".format(s["accuracy"], s["TP"] + s["TN"], s["TP"] + s["TN"] + s["FP"] + s["FN"]))
        lines.append("IoUg\t\t\t{:2.2%}".format(s["iou_ground"]))
        return "\n".join(lines)
