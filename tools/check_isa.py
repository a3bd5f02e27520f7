#!/usr/bin/env python
"""Static checks on the compiled network kernels (gfx950 ISA of nonrigid_nerf_amd/csrc/build*/net_*.o).

  python tools/check_isa.py [build_dir]          -> one line per kernel, exit 1 on a violated invariant

Invariants the hand-placed waits of nrnerf_net_impl.h rely on (16-bit kernels, WRing::frag / ready):
  * no scalar memory load (s_load / s_buffer_load) inside the MFMA section of a pass (first to last MFMA): a counted
    `s_waitcnt lgkmcnt(N)` stays correct with SMEM in flight (at most N operations outstanding still means at most N LDS
    reads, which retire in order), but the compiler waits for a scalar result with lgkmcnt(0), which drains the fragment
    prefetch queue.  Outside the section (tile prologue, the fused compositing epilogue) scalar loads are free;
  * no scratch traffic: every scratch reload is followed by `s_waitcnt vmcnt(0)`, which drains the LDS-DMA queue;
  * at most 256 VGPRs (two waves per SIMD) for the 16-bit kernels of nrnerf_net_impl.h; the two-blocks-per-wave kernels
    of nrnerf_net_mb.h and the 16x16x32 trunk-only kernel of nrnerf_net_x16.h run one wave per SIMD: at most 512 registers, and at most one v_accvgpr copy per 3 MFMAs (more
    means the accumulators went to AccVGPRs: the build lost -mllvm -amdgpu-mfma-vgpr-form);
  * nrnerf_net_x16.h: no draining `s_waitcnt lgkmcnt(0)` between the first and the last MFMA except at the layer ends (the biases
    travel in the counted fragment queue).
Also reported: MFMA count, VALU count, counted vs draining LDS waits.
"""
from __future__ import annotations

import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_code_object(obj: str, tmp: str) -> str | None:
    local = os.path.join(tmp, os.path.basename(obj))
    shutil.copy(obj, local)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, capture_output=True)
    hits = glob.glob(local + ".*gfx950*")
    return hits[0] if hits else None


def analyse(co: str) -> dict:
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    # per kernel; the rules apply to the kernels that stream weights through the LDS ring, i.e. not to trunk_wgrad (a
    # ring-less one-wave-per-SIMD kernel with 256 accumulator registers that shares an object with the training kernels)
    meta = {}
    for blk in re.split(r"\n\s*- \.agpr_count", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if name and "trunk_wgrad" in name.group(1):
            continue
        for k, v in re.findall(r"\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\d+)", blk):
            meta[k] = max(meta.get(k, 0), int(v))
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
    dis = re.sub(r"(?ms)^[0-9a-f]{16} <[^>]*trunk_wgrad[^>]*>:.*?(?=^[0-9a-f]{16} <|\Z)", "", dis)
    ins = [l.split()[0] for l in dis.splitlines() if re.match(r"^\s+[a-z]+_", l)]
    mfma = [i for i, x in enumerate(ins) if "mfma" in x]
    first = mfma[0] if mfma else len(ins)
    # scalar memory loads must all sit before the persistent loop
    base = None
    addr_of, smem_addrs, back_targets = {}, [], []
    for l in dis.splitlines():
        m = re.match(r"^([0-9a-f]{16}) <", l)
        if m:
            base = int(m.group(1), 16)
            continue
        m = re.search(r"//\s*([0-9A-F]{12}):", l)
        if not m or base is None:
            continue
        addr = int(m.group(1), 16)
        op = l.split()[0]
        if op.startswith(("s_load", "s_buffer_load")):
            smem_addrs.append(addr)
        t = re.search(r"<[^>]*\+0x([0-9a-f]+)>", l)
        if op.startswith(("s_cbranch", "s_branch")) and t and base + int(t.group(1), 16) <= addr:
            back_targets.append((addr - (base + int(t.group(1), 16)), base + int(t.group(1), 16)))
    # the persistent loop is the backward branch with the largest span (small loops before it, e.g. the bias-table
    # copy, end in an lgkmcnt(0) long before the first counted wait)
    loop_start = max(back_targets)[1] if back_targets else None
    smem_in_loop = sum(a >= loop_start for a in smem_addrs) if loop_start is not None else 0
    waits = [l.split("//")[0].strip() for l in dis.splitlines() if "s_waitcnt" in l]
    lg = [int(m.group(1)) for w in waits for m in [re.search(r"lgkmcnt\((\d+)\)", w)] if m]
    # the MFMA section of a pass: first to last MFMA in program order.  Since round 4 the kernels without a fused bender carry
    # the compositing epilogue (nrnerf_composite_ray.h) in their persistent loop, outside that section: it re-reads a few
    # kernel arguments with scalar loads (the compiler's cure for SGPR pressure) and parks values in AccVGPRs -- harmless
    # there (no fragment prefetch queue to drain, once per group of rays); inside the section the rules stay as they were
    # (per kernel: an object may hold several -- nrnerf_net_x16.o has one per 16-bit type)
    span, drains_in_section, kernels = [], 0, 0
    for fn in re.split(r"(?m)^[0-9a-f]{16} <[^>]*>:", dis)[1:]:
        f_lines = [l for l in fn.splitlines() if re.match(r"^\s+[a-z]+_", l)]
        f_ins = [l.split()[0] for l in f_lines]
        f_mfma = [i for i, x in enumerate(f_ins) if "mfma" in x]
        if f_mfma:
            kernels += 1
            span += f_ins[f_mfma[0]:f_mfma[-1] + 1]
            drains_in_section += sum("s_waitcnt" in l and "lgkmcnt(0)" in l for l in f_lines[f_mfma[0]:f_mfma[-1] + 1])
    return dict(meta, mfma=len(mfma), valu=sum(x.startswith("v_") and "mfma" not in x for x in ins),
                mb="net_kernel_mb" in dis or "net_kernel_x16" in dis, accvgpr=sum("accvgpr" in x for x in span), accvgpr_total=sum("accvgpr" in x for x in ins),
                smem_in_mfma_section=sum(x.startswith(("s_load", "s_buffer_load")) for x in span),
                smem_after_first_mfma=sum(x.startswith(("s_load", "s_buffer_load")) for x in ins[first:]) + smem_in_loop,
                scratch=sum(x.startswith("scratch_") for x in ins),
                lgkm_counted=sum(n > 0 for n in lg), lgkm_drain=sum(n == 0 for n in lg),
                lgkm_drain_in_section=drains_in_section, kernels=kernels, x16="net_kernel_x16" in dis)


def check(build_dir: str) -> list[str]:
    errors = []
    objs = sorted(glob.glob(os.path.join(build_dir, "net_*.o")) + glob.glob(os.path.join(build_dir, "bend_*.o")) +
                  glob.glob(os.path.join(build_dir, "train_*.o")) + glob.glob(os.path.join(build_dir, "nrnerf_net_x16_e*.o")))
    if not objs:
        raise FileNotFoundError(f"no net_*.o under {build_dir} (run make first)")
    with tempfile.TemporaryDirectory() as tmp:
        for obj in objs:
            co = device_code_object(obj, tmp)
            base = os.path.basename(obj)[:-2]
            name = base[4:] if base.startswith("net_") else base.replace("nrnerf_net_", "")
            if co is None:
                errors.append(f"{name}: no gfx950 code object")
                continue
            r = analyse(co)
            print(f"{name:24s} vgpr {r.get('vgpr_count', -1):3d} spill {r.get('vgpr_spill_count', -1):3d} scratch {r['scratch']:3d} "
                  f"mfma {r['mfma']:5d} valu {r['valu']:5d} lgkm counted/drain {r['lgkm_counted']:4d}/{r['lgkm_drain']:3d} "
                  f"smem in-section/in-loop {r['smem_in_mfma_section']}/{r['smem_after_first_mfma']}" + (f"  [1 wave/SIMD, accvgpr in-section {r['accvgpr']} of {r['accvgpr_total']}]" if r["mb"] else ""))
            sixteen = "_f32_" not in "_" + name + "_" and not name.startswith("train_bend_")      # the training bender is fp32 only
            if sixteen:
                # (reported, not enforced, for the stand-alone bender -- which reads its per-block inputs with scalar loads on
                #  purpose -- and the training kernels: a counted LDS wait is conservative with SMEM in flight, at most N
                #  operations outstanding still means at most N LDS reads; the cost is an occasional lgkmcnt(0) the compiler
                #  adds for the scalar result, which drains the fragment prefetch queue.  Enforced for the inference kernels.)
                if r["smem_in_mfma_section"] and not base.startswith(("bend_", "train_")):
                    errors.append(f"{name}: {r['smem_in_mfma_section']} scalar memory load(s) between the first and the last MFMA")
                # (a non-zero vgpr_spill_count with no scratch segment is a VGPR parked in a free AccVGPR: no memory traffic)
                # tolerated: ONE spilled register (8 B per lane) in x16_e0 -- the 128-wide trunk's raw-to-memory instantiation at two
                # waves per SIMD (256 registers); its spill-free alternatives were measured slower (profiles/r05_w128_x16_ab.txt)
                tolerated = name == "x16_e0" and r.get("private_segment_fixed_size", 0) <= 8 and r["scratch"] <= 16
                if (r["scratch"] or r.get("private_segment_fixed_size", 0)) and not tolerated:
                    errors.append(f"{name}: scratch traffic ({r['scratch']} instructions, {r.get('private_segment_fixed_size', 0)} B per lane)")
                limit = 512 if r["mb"] else 256
                if r.get("vgpr_count", 0) > limit:
                    errors.append(f"{name}: {r['vgpr_count']} VGPRs > {limit}")
                # the 16x16x32 kernel keeps EVERYTHING it reads from LDS between its first and last MFMA in the counted queue (fragments
                # and biases): the only lgkmcnt(0) in there are the layer ends, where the queue runs empty (9 layers + head per kernel).
                # A plain LDS load in the section brings hipcc's own lgkmcnt(0) back (87 per pass with the biases: -2.5 %)
                if r["x16"] and r["lgkm_drain_in_section"] > 12 * r["kernels"]:
                    errors.append(f"{name}: {r['lgkm_drain_in_section']} draining LDS waits inside the MFMA sections of {r['kernels']} kernels")
                if r["mb"] and 3 * r["accvgpr"] > r["mfma"]:
                    errors.append(f"{name}: {r['accvgpr']} AccVGPR copies for {r['mfma']} MFMAs (accumulators not in VGPRs?)")
        # round 5: the width-class trunk kernel (nrnerf_gx16.h) and the 16x16x32 bender (nrnerf_bend_x16.h) -- same counted LDS queue
        # (dense_x16), so: no scratch (a reload drains the LDS-DMA queue of the ring), no scalar load between the first and last MFMA
        for obj in sorted(glob.glob(os.path.join(build_dir, "nrnerf_gx16_w*.o")) + glob.glob(os.path.join(build_dir, "nrnerf_bend_x16.o"))):
            co = device_code_object(obj, tmp)
            name = os.path.basename(obj)[:-2].replace("nrnerf_", "")
            if co is None:
                errors.append(f"{name}: no gfx950 code object")
                continue
            r = analyse(co)
            print(f"{name:24s} vgpr {r.get('vgpr_count', -1):3d} spill {r.get('vgpr_spill_count', -1):3d} scratch {r['scratch']:3d} "
                  f"mfma {r['mfma']:5d} valu {r['valu']:5d} lgkm counted/drain {r['lgkm_counted']:4d}/{r['lgkm_drain']:3d} "
                  f"smem in-section/in-loop {r['smem_in_mfma_section']}/{r['smem_after_first_mfma']}")
            if r["scratch"] or r.get("private_segment_fixed_size", 0):
                errors.append(f"{name}: scratch traffic ({r['scratch']} instructions, {r.get('private_segment_fixed_size', 0)} B per lane)")
            if r["smem_in_mfma_section"] and name.startswith("gx16"):
                errors.append(f"{name}: {r['smem_in_mfma_section']} scalar memory load(s) between the first and the last MFMA")
    return errors


if __name__ == "__main__":
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "nonrigid_nerf_amd", "csrc", "build")
    errs = check(d)
    for e in errs:
        print("VIOLATION:", e)
    sys.exit(1 if errs else 0)
