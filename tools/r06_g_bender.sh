set -x
python - <<'PY' > gpurun_out/r06_bender_regs_identity.txt 2>&1
import os, sys, torch
sys.path.insert(0, os.getcwd())
from nonrigid_nerf_amd import _lib, render as R
from nonrigid_nerf_amd.synthetic import SceneConfig, make_scene, make_rays
from nonrigid_nerf_amd.synthetic import build_modules
lib = _lib.load()
dev = torch.device("cuda:0")
for prec in ("bf16", "f16"):
    for kw in (dict(), dict(N_samples=48, N_importance=37)):
        cfg = SceneConfig(**kw)
        scene = make_scene(cfg, 1)
        rays, lat = make_rays(5000, 3, cfg)
        rb, coarse, fine = build_modules(scene, device=dev)
        R.set_precision(prec)
        outs = []
        for reg in (0, 1):
            lib.nrnerf_experimental_bender_registers(reg)
            with torch.no_grad():
                o = R.render_rays(rays.to(dev), coarse, None, cfg.N_samples, N_importance=cfg.N_importance, network_fine=fine,
                                  additional_pixel_information={"ray_bending_latents": lat.to(dev)})
            outs.append({k: v.clone() for k, v in o.items()})
        same = all(torch.equal(outs[0][k], outs[1][k]) for k in outs[0])
        print(prec, kw, "bit-identical:", same, {k: float((outs[0][k] - outs[1][k]).abs().max()) for k in outs[0]})
PY
cat gpurun_out/r06_bender_regs_identity.txt | tail -6
for i in 1 2 3; do for r in 0 1; do echo "registers=$r run $i"; NRNERF_BENDER_REGISTERS=$r python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], d["ms_per_step"], r["kernels_ms_per_step"])'; done; done > gpurun_out/r06_bender_registers_ab.txt 2>&1
cat gpurun_out/r06_bender_registers_ab.txt
