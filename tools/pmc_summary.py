#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter CSVs (one directory per pass) for net_kernel into the text + json committed
under profiles/.  FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for wide coalesced reads
on gfx950; FETCH_SIZE / WRITE_SIZE are in KiB.

    python tools/pmc_summary.py gpurun_out/pmcA gpurun_out/pmcB ... > profiles/r02_pmc_summary.txt

Also writes profiles/<NRNERF_PROFILE_TAG, default r03>_pmc_fine.json (per-pass medians + the hash of the kernel sources that were profiled + the bench
scene): bench.py reports roofline.traffic from it only while that hash matches the sources of the checkout it runs from.
"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(dirs):
    per = collections.defaultdict(dict)       # (pass, dispatch) -> counters
    bend = collections.defaultdict(dict)      # the stand-alone bender kernel (split-bender path)
    for d in dirs:
        for f in glob.glob(d + "/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "bend_kernel" in r["Kernel_Name"]:
                    kb = (d, int(r["Dispatch_Id"]))
                    bend[kb][r["Counter_Name"]] = float(r["Counter_Value"])
                    bend[kb]["_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                if "net_kernel" not in r["Kernel_Name"]:
                    continue
                k = (d, int(r["Dispatch_Id"]))
                per[k][r["Counter_Name"]] = float(r["Counter_Value"])
                per[k]["_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    # launches alternate coarse / fine within a pass; take the median of each counter over the fine launches
    out = {}
    for label, parity in (("coarse", 0), ("fine", 1)):
        agg = collections.defaultdict(list)
        for d in dirs:
            ids = sorted(i for (dd, i) in per if dd == d)
            for n, i in enumerate(ids):
                if n % 2 == parity:
                    for c, v in per[(d, i)].items():
                        agg[c].append(v)
        out[label] = {c: sorted(v)[len(v) // 2] for c, v in agg.items()}
    # the stand-alone bender: one launch per frame (importance samples) up to round 4; two since round 5 (coarse samples, then the
    # importance samples) -- told apart by their order within a pass when the launch count is even and the durations differ by ~2x
    for d in dirs:
        ids = sorted(i for (dd, i) in bend if dd == d)
        durs = [bend[(d, i)]["_us"] for i in ids]
        two = len(ids) >= 2 and len(ids) % 2 == 0 and sum(durs[1::2]) > 1.4 * sum(durs[0::2])
        for n, i in enumerate(ids):
            bend[(d, i)]["_label"] = ("bend_coarse" if n % 2 == 0 else "bend_fine") if two else "bend"
    for label in ("bend", "bend_coarse", "bend_fine"):
        agg = collections.defaultdict(list)
        for kb, cs in bend.items():
            if cs.get("_label") != label:
                continue
            for c, v in cs.items():
                if c != "_label":
                    agg[c].append(v)
        if agg:
            out[label] = {c: sorted(v)[len(v) // 2] for c, v in agg.items()}
    print("# rocprofv3 --pmc summary, net_kernel / bend_kernel (median over launches), bench.py workload (196608 rays, 64+128)")
    for label in [l for l in ("coarse", "fine", "bend", "bend_coarse", "bend_fine") if l in out]:
        c = out[label]
        print(f"\n[{label} pass]  duration {c.get('_us', 0):.1f} us")
        for k in sorted(c):
            if k != "_us":
                print(f"  {k:32s} {c[k]:.6g}")
        if "GRBM_GUI_ACTIVE" in c:
            clk = c["GRBM_GUI_ACTIVE"] / 8 / c["_us"] * 1e-3     # summed over the 8 XCDs
            print(f"  -> effective shader clock            {clk:.2f} GHz")
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                print(f"  -> MFMA pipe busy                    {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * c['GRBM_GUI_ACTIVE'] / 8):.3f} of SIMD-cycles")
        if "SQ_WAVE_CYCLES" in c:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if k in c:
                    print(f"  -> {k:20s} / SQ_WAVE_CYCLES = {c[k] / c['SQ_WAVE_CYCLES']:.3f}")
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rd, wr = 2 * c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
            print(f"  -> HBM traffic per launch: read {rd / 1e6:.1f} MB (2 x FETCH_SIZE), write {wr / 1e6:.1f} MB, total {(rd + wr) / 1e9:.3f} GB")
            out[label]["hbm_bytes_per_launch"] = rd + wr
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            tot = c["TCC_HIT_sum"] + c["TCC_MISS_sum"]
            print(f"  -> L2 hit rate                       {c['TCC_HIT_sum'] / max(tot, 1):.4f}  ({tot:.4g} requests: the weight stream is "
                  f"re-read from L2 by every workgroup pass, the rays / depths / outputs stream through once)")
    import bench
    out["kernel_source_sha16"] = bench.kernel_source_sha16()
    out["scene"] = os.environ.get("NRNERF_PROFILE_SCENE", "fitted")
    tag = os.environ.get("NRNERF_PROFILE_TAG", "r03")
    json.dump(out, open(os.path.join(REPO, "profiles", f"{tag}_pmc_fine.json"), "w"), indent=1)


if __name__ == "__main__":
    sys.path.insert(0, REPO)
    main(sys.argv[1:])
