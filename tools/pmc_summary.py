"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/<rev>/pmc_raw_per_launch.json and
profiles/pmc_summary.json (what bench.py reports as roofline.traffic).

usage: python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <profiles/rev dir> <batch> <points bytes per launch>

The cold first launch of each kernel is dropped.  Correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE
on gfx950 counts 16 B/lane coalesced streaming reads at half their bytes, so half of K1's point stream is added back;
4 B/lane reads and WRITE_SIZE calibrate 1:1 in this code (see profiles/README.md)."""
import collections
import csv
import json
import os
import sys

GROUPS = {"K1_classify": ["gg::k_classify"], "K2_sort_reduce": ["gg::k_scan", "gg::k_scatter", "gg::k_reduce"],
          "K3_patch": ["gg::k_patch"], "K4_sweep": ["gg::k_sweep"], "K5_label": ["gg::k_label"]}


def per_kernel(path):
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("gg::k_") and "fill" not in k:
            vals[k].append(float(r["Counter_Value"]))
    return {k: (sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0]) for k, v in vals.items()}


def main():
    fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
    out_dir, batch, point_bytes = sys.argv[3], int(sys.argv[4]), float(sys.argv[5])
    raw = {k: {"FETCH_SIZE_KB": round(fetch.get(k, 0.0), 1), "WRITE_SIZE_KB": round(write.get(k, 0.0), 1)} for k in sorted(set(fetch) | set(write))}
    os.makedirs(out_dir, exist_ok=True)
    json.dump(raw, open(os.path.join(out_dir, "pmc_raw_per_launch.json"), "w"), indent=1)
    summary = {"batch": batch, "source": os.path.join(out_dir, "pmc_raw_per_launch.json"),
               "note": "HBM bytes per launch = (FETCH_SIZE + WRITE_SIZE) KB x 1024, separate --pmc passes, cold first launch dropped; "
                       "FETCH of k_classify corrected by + half of the 16-B/lane point stream (gfx950 counts wide coalesced reads at 1/2); "
                       "includes Infinity-Cache hits"}
    for g, prefixes in GROUPS.items():
        f = sum(v["FETCH_SIZE_KB"] for k, v in raw.items() if any(k.startswith(p) for p in prefixes)) * 1024.0
        w = sum(v["WRITE_SIZE_KB"] for k, v in raw.items() if any(k.startswith(p) for p in prefixes)) * 1024.0
        if g == "K1_classify":
            f += 0.5 * point_bytes
        summary[g] = {"hbm_bytes_per_launch": int(f + w), "fetch_MB": round(f / 1e6, 1), "write_MB": round(w / 1e6, 1)}
    json.dump(summary, open(os.path.join(os.path.dirname(out_dir.rstrip("/")), "pmc_summary.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
