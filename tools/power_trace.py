#!/usr/bin/env python
"""GPU box: socket power, power cap and shader clock (hwmon / sysfs of GPU 0, sampled every 50 ms) while bench.py renders
frames back to back -- the direct evidence for DESIGN.md's "power-bound" reading of the network kernel.
    python tools/power_trace.py [precision] [steps]     -> summary on stdout, samples in gpurun_out/power_trace_<precision>.csv"""
import glob
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def find_hwmons():
    """every GPU of the node is visible in sysfs, whichever one this container was given: sample them all, report the busy one"""
    out = []
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        if "-" in os.path.basename(card):
            continue
        for hw in glob.glob(os.path.join(card, "device/hwmon/hwmon*")):
            for name in ("power1_average", "power1_input"):
                if read(os.path.join(hw, name)):
                    out.append((card, hw, os.path.join(hw, name)))
                    break
    return out


def sclk_mhz(card):
    txt = read(os.path.join(card, "device/pp_dpm_sclk")) or ""
    for line in txt.splitlines():
        if line.rstrip().endswith("*"):
            return float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
    return float("nan")


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    steps = sys.argv[2] if len(sys.argv) > 2 else "250"
    sensors = find_hwmons()
    if not sensors:
        print("no hwmon power sensor visible in this container")
        smi = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True)
        print(smi.stdout[-1500:])
        return
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--precision", prec, "--steps", steps, "--warmup", "5", "--no-cpu-baseline", "--no-psnr",
           "--no-train-step", "--min-gpu-seconds", "0"]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    samples, t0 = [], time.time()
    while p.poll() is None:
        samples.append((time.time() - t0, [(float(read(pf) or 0) / 1e6, sclk_mhz(card)) for card, hw, pf in sensors]))
        time.sleep(0.05)
    out = p.stdout.read()
    swing = [max(s[1][i][0] for s in samples) - min(s[1][i][0] for s in samples) for i in range(len(sensors))]
    k = swing.index(max(swing))
    card, hw, _ = sensors[k]
    cap = read(os.path.join(hw, "power1_cap"))
    print(f"{len(sensors)} GPUs with a power sensor; the one this run drove: {os.path.basename(card)} (power swing {swing[k]:.0f} W; others <= "
          f"{max([x for i, x in enumerate(swing) if i != k] or [0]):.0f} W)")
    rows = [(t, v[k][0], v[k][1], read(os.path.join(hw, "freq1_input"))) for t, v in samples]
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", f"power_trace_{prec}.csv"), "w") as f:
        f.write("t_s,power_w,sclk_mhz_dpm,freq1_input_hz\n")
        for r in rows:
            f.write(",".join(str(x) for x in r) + "\n")
    pw = sorted(r[1] for r in rows)
    busy = [r for r in rows if r[1] > 0.8 * pw[-1]]
    line = [l for l in out.splitlines() if l.startswith("{")]
    print(f"[{prec}] power cap {float(cap) / 1e6 if cap else float('nan'):.0f} W; {len(rows)} samples, idle {pw[0]:.0f} W, max {pw[-1]:.0f} W; "
          f"while rendering ({len(busy)} samples > 80 % of max): mean {sum(r[1] for r in busy) / max(len(busy), 1):.0f} W, "
          f"dpm sclk mean {sum(r[2] for r in busy) / max(len(busy), 1):.0f} MHz")
    if line:
        import json
        d = json.loads(line[-1])
        print(f"[{prec}] bench: {d['value'] / 1e6:.3f} M rays/s, {d['ms_per_step']} ms/step, dominant kernel {d['roofline']['achieved']} TFLOP/s")


if __name__ == "__main__":
    main()
