"""Export a down-sampled copy of the reference's example sequence as a small fixture.  TEST INFRASTRUCTURE ONLY.

Run in the build container only (needs /root/reference/data; the GPU box does not have it):

    python oracle/export_example_sequence.py            # -> tests/golden/example_sequence_96x72.npz

The fixture is what ``oracle/fit_checkpoint.py`` fits NR-NeRF to, to obtain a checkpoint with trained-like weight
statistics (the reference ships none), and what the PSNR-vs-ground-truth tests compare
renders with (PSNR definition of free_viewpoint_rendering.py:821-828).

What is stored (the loader's outputs of load_llff.py:5-33 and train.py:1345-1372, at 1/SCALE resolution):
  images  uint8 [F, H, W, 3]   every STEP-th frame, area-averaged from the 512x384 PNGs
  poses   f32   [F, 3, 4]      camera-to-world, unchanged (precomputed.json "poses"[:, :, :4])
  hwf     f32   [3]            height, width, focal scaled by 1/SCALE (poses[0, :, 4])
  bds     f32   [F, 2]         depth bounds (near = min * 0.9, far = max, train.py:1418-1419)
  frame_ids int [F]            index of each kept frame in the original sequence
  i_test  int                  position (in the kept frames) of the held-out frame: the original i_test = 20
  render_poses f32 [R, 3, 4]   the first few poses of the reference's render path
"""
from __future__ import annotations

import json
import os

import numpy as np
from PIL import Image

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("NRNERF_REFERENCE", "/root/reference")
SCALE, STEP = 16 / 3, 2          # 512x384 -> 96x72, every second frame (43 of 86)
OUT_W, OUT_H = 96, 72


def main():
    seq = os.path.join(REF, "data", "example_sequence")
    with open(os.path.join(seq, "precomputed.json")) as f:
        pre = json.load(f)
    poses = np.asarray(pre["poses"], dtype=np.float64)           # [86, 3, 5]
    bds = np.asarray(pre["bds"], dtype=np.float64)
    names = sorted(os.listdir(os.path.join(seq, "images")))
    keep = list(range(0, len(names), STEP))
    assert int(pre["i_test"]) in keep
    imgs = []
    for i in keep:
        im = Image.open(os.path.join(seq, "images", names[i])).convert("RGB")
        assert im.size == (512, 384)
        imgs.append(np.asarray(im.resize((OUT_W, OUT_H), Image.BOX), dtype=np.uint8))
    h, w, focal = poses[0, :, 4]
    out = dict(images=np.stack(imgs, 0), poses=poses[keep, :, :4].astype(np.float32),
               hwf=np.asarray([h / SCALE, w / SCALE, focal / SCALE], dtype=np.float32),
               bds=bds[keep].astype(np.float32), frame_ids=np.asarray(keep, dtype=np.int32),
               i_test=np.asarray(keep.index(int(pre["i_test"])), dtype=np.int32),
               render_poses=np.asarray(pre["render_poses"], dtype=np.float32)[:8, :, :4])
    path = os.path.join(REPO, "tests", "golden", f"example_sequence_{OUT_W}x{OUT_H}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {len(keep)} frames {OUT_W}x{OUT_H}, hwf {out['hwf']}, near {bds.min() * 0.9:.6f} far {bds.max():.6f}, "
          f"{os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
