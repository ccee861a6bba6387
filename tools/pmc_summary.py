"""Turn rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/<rev>/pmc_raw_per_launch.json and profiles/pmc_summary.json
(what bench.py reports as roofline.traffic for the dominant kernel).

usage: python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <profiles/rev dir> <clouds per launch> [section]
       section "kernels" (default: the headline workload) or "config4_kernels" (configs[3]; merged into an existing pmc_summary.json)

The cold first launch of each kernel is dropped.  Correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE on gfx950
counts wide coalesced streaming reads (16 B / lane, and the 512-byte-per-wavefront streams of k_sweep) at half their bytes:
kernels listed in HALVED get their FETCH doubled.  WRITE_SIZE is taken 1:1."""
import collections
import csv
import json
import os
import sys

KERNELS = ["k_classify", "k_scan", "k_scatter", "k_reduce", "k_patch", "k_sweep", "k_label"]
HALVED = {"k_classify": "16-B / lane point stream", "k_sweep": "512 B contiguous per wavefront and access (sheared layer)"}


def per_kernel(path, biggest_only=False):
    """average counter value per kernel over its launches (the cold first one dropped).  biggest_only: the run also launched the
    same kernels on much smaller workloads (bench.py --only-config4: a token headline batch, the single-cloud leg): keep the
    launches within a factor of two of the kernel's largest."""
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].replace("gg::", "")
        k = {"k_scan_parts": "k_scan"}.get(k, k)  # (the scan of one cloud by several work-groups: the same row of the tables)
        if k in KERNELS:
            vals[k].append(float(r["Counter_Value"]))
    if biggest_only:
        vals = {k: [x for x in v if x >= 0.5 * max(v)] for k, v in vals.items()}
    return {k: (sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0]) for k, v in vals.items()}


def main():
    section = sys.argv[5] if len(sys.argv) > 5 else "kernels"
    fetch, write = per_kernel(sys.argv[1], section != "kernels"), per_kernel(sys.argv[2], section != "kernels")
    out_dir, batch = sys.argv[3], int(sys.argv[4])
    raw = {k: {"FETCH_SIZE_KB": round(fetch.get(k, 0.0), 1), "WRITE_SIZE_KB": round(write.get(k, 0.0), 1)} for k in KERNELS if k in fetch or k in write}
    os.makedirs(out_dir, exist_ok=True)
    raw_name = "pmc_raw_per_launch.json" if section == "kernels" else f"pmc_raw_per_launch_{section}.json"
    json.dump(raw, open(os.path.join(out_dir, raw_name), "w"), indent=1)
    summary_path = os.path.join(os.path.dirname(out_dir.rstrip("/")), "pmc_summary.json")
    if section != "kernels":  # a second workload: keep what the file already says about the headline
        summary = json.load(open(summary_path)) if os.path.exists(summary_path) else {}
        summary[section + "_clouds_per_launch"] = batch
        summary[section] = {}
        for k, v in raw.items():
            f = v["FETCH_SIZE_KB"] * 1024.0 * (2.0 if k in HALVED else 1.0)
            w = v["WRITE_SIZE_KB"] * 1024.0
            summary[section][k] = {"hbm_bytes_per_launch": int(f + w), "hbm_bytes_per_cloud": int((f + w) / batch), "fetch_MB": round(f / 1e6, 1),
                                   "write_MB": round(w / 1e6, 1)}
        json.dump(summary, open(summary_path, "w"), indent=1)
        print(json.dumps(summary[section], indent=1))
        return
    summary = {"clouds_per_launch": batch, "source": os.path.join(out_dir, raw_name),
               "note": "HBM bytes per launch = (FETCH_SIZE x correction + WRITE_SIZE) KB x 1024, separate --pmc passes, cold first launch dropped; "
                       "FETCH doubled for the kernels in `halved` (gfx950 counts wide coalesced reads at 1/2); includes Infinity-Cache hits",
               "halved": HALVED, "kernels": {}}
    for k, v in raw.items():
        f = v["FETCH_SIZE_KB"] * 1024.0 * (2.0 if k in HALVED else 1.0)
        w = v["WRITE_SIZE_KB"] * 1024.0
        summary["kernels"][k] = {"hbm_bytes_per_launch": int(f + w), "hbm_bytes_per_cloud": int((f + w) / batch), "fetch_MB": round(f / 1e6, 1),
                                 "write_MB": round(w / 1e6, 1)}
    json.dump(summary, open(summary_path, "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
