#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB per dispatch) into profiles/traffic.json
(HBM bytes per launch and kernel) and compact per-kernel CSV summaries.

usage: make_traffic.py fetch_counter_collection.csv write_counter_collection.csv traffic.json fetch_summary.csv write_summary.csv
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    tot = collections.defaultdict(float)
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        tot[k] += float(r["Counter_Value"]) * 1024.0          # counter unit: KB
        disp[k].add(r["Dispatch_Id"])
    return {k: (tot[k] / max(len(disp[k]), 1), len(disp[k])) for k in tot}


def main():
    fpath, wpath, out, fsum, wsum = sys.argv[1:6]
    tile = int(sys.argv[6]) if len(sys.argv) > 6 else 1 << 21          # pairs per launch of the profiled command (bench.py tile_pairs)
    f = per_kernel(fpath, "FETCH_SIZE")
    w = per_kernel(wpath, "WRITE_SIZE")
    for path, d, name in ((fsum, f, "FETCH_SIZE"), (wsum, w, "WRITE_SIZE")):
        with open(path, "w") as fh:
            fh.write(f"kernel,launches,{name}_bytes_per_launch\n")
            for k in sorted(d, key=lambda k: -d[k][0] * d[k][1]):
                fh.write(f"{k},{d[k][1]},{d[k][0]:.0f}\n")
    detail = {k: {"fetch_bytes_per_launch": f.get(k, (0, 0))[0], "write_bytes_per_launch": w.get(k, (0, 0))[0],
                  "launches": f.get(k, (0, 0))[1]} for k in sorted(set(f) | set(w)) if k.startswith("k_")}
    tot = lambda k: detail.get(k, {}).get("fetch_bytes_per_launch", 0) + detail.get(k, {}).get("write_bytes_per_launch", 0)
    # the pair stage of an item = one k_pair launch + the heavy pairs' kernels beside it: k_pair_heavy (round 3), or the pipeline
    # k_hp_* (several launches of some of them per stage) with k_pair_heavy as its fall-back -- all their bytes, per k_pair launch
    n_stage = max(detail.get("k_pair", {}).get("launches", 0), 1)
    beside = [k for k in detail if k == "k_pair_heavy" or k.startswith("k_hp_")]
    pair_stage = tot("k_pair") + sum(tot(k) * detail[k]["launches"] for k in beside) / n_stage
    js = {"workload": "hg38like", "preset": "dense (r03: ~60 000 genes, tiered repeat families)", "pairs": tile,
          "note": "FETCH_SIZE/WRITE_SIZE (KB) x 1024 from two separate rocprofv3 --pmc passes of `bench.py --steps 2 --warmup 1`, averaged per launch (one launch = one mapping round of a tile of `pairs` pairs against one packed contig of the hg38-like genome; `pairs` = pairs per launch).  MI355X_MICROARCH.md: FETCH_SIZE under-counts wide coalesced streams by 2x (128-B requests tallied at 64 B) and other access widths are uncalibrated; for this path's pattern (4-16 B random gathers + scratch rows) the r01 calibration against TCC_MISS x 64 B agreed with the counters at face value, so no correction is applied.  Stage entries sum the kernels of a stage (light + heavy; pair stage: k_pair + k_pair_heavy + every k_hp_* launch of the stage).",
          "bytes_per_launch": {"k_seed": tot("k_seed"), "k_chain": tot("k_chain") + tot("k_chain_heavy"),
                               "k_pair": pair_stage},
          "detail": detail}
    json.dump(js, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
