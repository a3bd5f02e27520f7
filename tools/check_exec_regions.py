#!/usr/bin/env python
"""No matrix instruction under a SAVED exec mask in the 16x16x32 kernels (csrc/build/nrnerf_net_x16*.o, nrnerf_gx16_w*.o, nrnerf_bend_x16.o).

The layers of these kernels read their weight fragments with `asm volatile` LDS reads that do not name the exec mask, so a lane-divergent
region (`s_and_saveexec_b64 … s_or_b64 exec, exec, …`) that the compiler stretches over a layer would run it for some lanes only -- round 6
met exactly that: a lane-0-only `if` in gx16_kernel's loop made the f16 view-dependent instantiation lose 10 x in accuracy
(tools/experiments/README.md).  A linear scan of the disassembly: every saveexec / restore pair, the v_mfma instructions between them.
    python tools/check_exec_regions.py [build_dir]      -> the offending regions, exit 1 if any"""
import glob
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_isa  # noqa: E402


def offending_regions(build_dir):
    out = []
    objs = sorted(glob.glob(os.path.join(build_dir, "nrnerf_net_x16*.o")) + glob.glob(os.path.join(build_dir, "nrnerf_gx16_w*.o")) +
                  glob.glob(os.path.join(build_dir, "nrnerf_bend_x16.o")))
    for obj in objs:
        with tempfile.TemporaryDirectory() as tmp:
            co = check_isa.device_code_object(obj, tmp)
            if co is None:
                continue
            dis = subprocess.run([f"{check_isa.LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
        fn, open_regs = None, {}
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]{16} <(.*)>:", line)
            if m:
                fn, open_regs = m.group(1), {}
                continue
            t = line.split()
            if not t or not re.match(r"^[a-z]+_", t[0]):
                continue
            op = t[0]
            if op.startswith(("s_and_saveexec_b64", "s_andn2_saveexec_b64", "s_or_saveexec_b64")):
                open_regs[t[1].rstrip(",")] = 0
            elif op == "s_or_b64" and len(t) > 3 and t[1].startswith("exec"):
                n = open_regs.pop(t[3].rstrip(","), None)
                if n:
                    out.append(f"{os.path.basename(obj)}: {n} v_mfma under a saved exec mask in {fn[:90]}")
            elif "mfma" in op:
                for r in open_regs:
                    open_regs[r] += 1
    return out, len(objs)


if __name__ == "__main__":
    build = sys.argv[1] if len(sys.argv) > 1 else os.path.join(check_isa.REPO, "nonrigid_nerf_amd", "csrc", "build")
    bad, n = offending_regions(build)
    print("\n".join(bad) if bad else f"{n} objects: no matrix instruction under a saved exec mask")
    sys.exit(1 if bad else 0)
