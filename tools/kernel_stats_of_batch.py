"""Kernel statistics of the launches of ONE batch size out of a rocprofv3 --kernel-trace CSV.

    python tools/kernel_stats_of_batch.py <..._kernel_trace.csv> <n_clouds> > kernel_stats_<n_clouds>.csv

A bench run launches the same kernels for batches of different sizes (bench.py --only-config4: a token headline of 8 clouds, the
128-cloud steps, then one cloud per launch); rocprofv3's own --stats averages them together.  Every step of bench.py's pipelines starts
with the re-initialisation of its maps -- one launch whose grid has one row per map (k_reset_fresh, or k_fill2_strided when fresh maps are
switched off) -- so the dispatches from a re-initialisation of exactly <n_clouds> maps up to the next re-initialisation are the kernels of
a step of that batch size."""
import csv
import sys
from collections import OrderedDict


def main():
    path, n_clouds = sys.argv[1], int(sys.argv[2])
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    is_reset = lambda r: r["Kernel_Name"].startswith("gg::k_reset_fresh") or r["Kernel_Name"].startswith("gg::k_fill2_strided")
    stats, inside, steps = OrderedDict(), False, 0
    for r in rows:
        if is_reset(r):
            inside = int(r["Grid_Size_Y"]) == n_clouds
            steps += inside
        if not inside:
            continue
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        s = stats.setdefault(r["Kernel_Name"], [0, 0, None, 0])
        s[0] += 1
        s[1] += d
        s[2] = d if s[2] is None else min(s[2], d)
        s[3] = max(s[3], d)
    total = sum(s[1] for s in stats.values()) or 1
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", f"(steps of {n_clouds} clouds: {steps})"])
    for name, s in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        w.writerow([name, s[0], s[1], round(s[1] / s[0], 1), round(100.0 * s[1] / total, 2), s[2], s[3]])


if __name__ == "__main__":
    main()
