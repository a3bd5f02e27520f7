"""Which outputs / rays differ between the fused compositing epilogue and the composite kernel (debug aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nonrigid_nerf_amd import render as R
from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_rays, make_scene

DEV = "cuda:0"
for prec in ("bf16", "f16", "f32"):
    cfg = SceneConfig()
    scene = make_scene(cfg, 4)
    n = 3001
    rays, latents = make_rays(n, 29, cfg)
    rb, coarse, fine = build_modules(scene, device=DEV)
    R.set_precision(prec)
    model = R.get_model(coarse, fine)
    outs = []
    for unfused in ("0", "1", "0"):
        os.environ["NRNERF_UNFUSED_COMPOSITE"] = unfused
        with torch.no_grad():
            outs.append(model.render(rays.to(DEV), latents.to(DEV), 64, 128, retraw=True, want_z_vals=True, surface=True))
        torch.cuda.synchronize()
    f, u, f2 = outs
    for k in f:
        a, b, c = torch.nan_to_num(f[k].float()), torch.nan_to_num(u[k].float()), torch.nan_to_num(f2[k].float())
        d = (a - b).abs()
        bad = (d.reshape(n, -1) > 0).any(1)
        print(prec, k, "fused!=unfused rays:", int(bad.sum()), "max", float(d.max()), "fused run1!=run2:", int(((a - c).abs().reshape(n, -1) > 0).any(1).sum()),
              "first bad rays", bad.nonzero().flatten()[:12].tolist())
