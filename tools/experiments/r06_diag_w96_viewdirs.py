import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from nonrigid_nerf_amd.synthetic import SceneConfig, make_scene, make_rays
from test_gpu_parity import hip_render, _setenv
cfg = SceneConfig(N_importance=64, netdepth=5, netwidth=96, skips=(1,), use_viewdirs=True, multires_views=3)
scene = make_scene(cfg, 4)
rays, latents = make_rays(1777, 23, cfg)
ref = hip_render(scene, rays, latents, "f32", retraw=True)
for prec in ("f16", "bf16"):
    for env in ({}, {"NRNERF_X16_BENDER": "0"}, {"NRNERF_X16": "0"}, {"NRNERF_FIXED_SHARES": "1"}, {"NRNERF_X16_BENDER": "0", "NRNERF_FIXED_SHARES": "1"}):
        import contextlib
        with contextlib.ExitStack() as st:
            for k, v in env.items(): st.enter_context(_setenv(k, v))
            o = hip_render(scene, rays, latents, prec, retraw=True)
        e = (o["rgb0"].float() - ref["rgb0"].float()).abs().max(1).values
        bad = (e > 5e-4).nonzero().flatten()
        print(prec, env, "mean err", float(e.mean()), "rays > 5e-4:", bad.numel(), bad[:12].tolist())
