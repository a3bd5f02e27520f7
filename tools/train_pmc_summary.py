#!/usr/bin/env python
"""HBM traffic of the native training kernels from two rocprofv3 --pmc passes of tools/train_step_profile.py
(FETCH_SIZE, WRITE_SIZE in KiB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950):
    python tools/train_pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <rays per step>"""
import collections
import csv
import glob
import sys

KERNELS = ("trunk_fwd_train", "trunk_bwd", "trunk_wgrad", "wgrad_operands", "bend_fwd_train", "bend_bwd", "bend_div_fwd", "bend_div_bwd", "bend_wgrad", "composite_bwd", "composite_kernel")


def collect(d):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            for k in KERNELS:
                if k in r["Kernel_Name"]:
                    per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    per[k]["_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
                    break
    return per


def main():
    fetch, write, rays = collect(sys.argv[1]), collect(sys.argv[2]), int(sys.argv[3])
    print(f"# HBM traffic of the training kernels, bf16 mode, {rays} rays x (64 + 64) per step, shipped recipe; per launch = mean over the launches of 23 steps")
    print(f"# (two launches per step of each trunk / bender kernel: coarse pass 64 samples per ray, fine pass 128; bend_wgrad a third time for the divergence term)")
    print(f"{'kernel':18s} {'launches':>8s} {'avg us':>10s} {'read MB':>10s} {'write MB':>10s} {'TB/s':>8s}")
    for k in KERNELS:
        if k not in fetch and k not in write:
            continue
        us = fetch[k]["_us"] or write[k]["_us"]
        n = len(us)
        rd = 2 * sum(fetch[k].get("FETCH_SIZE", [0])) / max(len(fetch[k].get("FETCH_SIZE", [1])), 1) * 1024 / 1e6
        wr = sum(write[k].get("WRITE_SIZE", [0])) / max(len(write[k].get("WRITE_SIZE", [1])), 1) * 1024 / 1e6
        avg = sum(us) / n
        print(f"{k:18s} {n:8d} {avg:10.1f} {rd:10.1f} {wr:10.1f} {(rd + wr) / avg:8.2f}")


if __name__ == "__main__":
    main()
