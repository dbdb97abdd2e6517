#!/usr/bin/env python
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes -> profiles/<tag>_traffic.json: HBM bytes per launch for the kernels
bench.py names.  FETCH_SIZE / WRITE_SIZE are reported in KB; per MI355X_MICROARCH.md (HBM section)
gfx950's FETCH_SIZE counts 128-byte read requests at 64 B, so reads are doubled ("corrected"); WRITE_SIZE is taken
as reported (uncalibrated).  usage: make_traffic_json.py <dir with pmc*_counter_collection.csv> <out.json> [kernel_stats.csv]
With a `rocprofv3 --kernel-trace --stats` summary as third argument its average launch durations go into the file too (`rocprofv3_avg_us`):
bench.py prints `roofline.frac_rocprofv3` from them next to its own HIP-event figure when the kernel source hash matches."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    """the hash bench.py keys a traffic profile on (device code + the Makefile's code-generation flags)"""
    sys.path.insert(0, ROOT)
    import bench

    return bench.kernel_source_hash()


NAMES = [("warp_fast_kernel<3, true, true", "warp_img_mask"), ("warp_fast_kernel<2, true, true", "warp_img_mask"),
         ("warp_fast_kernel<0, true, true", "warp_img_mask"),
         ("warp_fast_kernel<3, true, true, false, 0, true", "warp_img_mask_gain"),
         ("seam_resize4", "seam_mask_resize"), ("dilate3x3", "seam_mask_dilate"), ("mb_level0_deferred", "mb_level0_deferred"),
         ("gain_rows_kernel", "block_gain_rows"), ("mb_level0_pk_kernel", "mb_level0"),
         ("mb_down0_lds_kernel", "mb_down0"), ("mb_down_lds_kernel", "mb_down"), ("warp_tables_kernel", "warp_tables"),
         ("mb_level_pk_kernel", "mb_level"), ("mb_coarse_kernel", "mb_coarse"), ("roi_kernel", "warp_roi")]


def main(d, out, stats=None):
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(f"{d}/*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            k = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"]).replace("void ", "")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    res = {}
    for k, c in acc.items():
        for pat, name in NAMES:
            if k.startswith(pat) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                fetch_kb = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
                write_kb = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
                res[name] = {"hip_kernel": k, "fetch_bytes_raw": fetch_kb * 1024, "write_bytes_raw": write_kb * 1024,
                             "traffic_bytes": 2 * fetch_kb * 1024 + write_kb * 1024,
                             "note": "2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE, mean per launch"}
    items = dict(res)
    if stats and os.path.exists(stats):
        avg = {}
        for row in csv.DictReader(open(stats)):
            k = re.sub(r"\(anonymous namespace\)::", "", row["Name"]).replace("void ", "")
            for pat, name in NAMES:
                if k.startswith(pat):
                    tot, calls = avg.get(name, (0.0, 0))
                    avg[name] = (tot + float(row["TotalDurationNs"]), calls + int(row["Calls"]))
        res["rocprofv3_avg_us"] = {name: round(tot / calls / 1e3, 3) for name, (tot, calls) in avg.items() if calls}
        res["rocprofv3_stats_file"] = os.path.basename(stats)
    res["kernel_source_hash"] = kernel_source_hash()
    # which workload the passes ran: 2 (the bench default) unless the caller names another leg (tools/prof_cmd.sh)
    wl = os.environ.get("STX_TRAFFIC_WORKLOAD", "")
    res["workload_cfg"] = 2 if not wl else wl
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in items.items():
        print(f"{k:16s} fetch {v['fetch_bytes_raw']/1e6:9.1f} MB (x2 = {2*v['fetch_bytes_raw']/1e6:9.1f})  write {v['write_bytes_raw']/1e6:9.1f} MB")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
