#!/usr/bin/env python
"""BASELINE.json configs[4]: replay a SemanticKITTI sequence through the MI355X path and print the evaluator table
(compare with the reference's README.md:57-94 for sequence 00).

    python tools/kitti_replay.py /data/semantickitti/sequences/00 [--max-frames N]
"""
import argparse, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groundgrid_amd import kitti, replay  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("sequence_dir")
    ap.add_argument("--max-frames", type=int, default=0)
    args = ap.parse_args()
    seq = kitti.KittiSequence(args.sequence_dir)
    n = len(seq) if not args.max_frames else min(len(seq), args.max_frames)
    ev, spent = replay.replay((seq.frame(i) for i in range(n)), replay.DeviceBackend())
    print(ev.table())
    print(f"{n} clouds in {spent:.2f} s inside the device path (host staging + PCIe included): {n / spent:.1f} clouds/s")
