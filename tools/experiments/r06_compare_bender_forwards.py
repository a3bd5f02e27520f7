import os, sys, torch
sys.path.insert(0, os.getcwd())
from nonrigid_nerf_amd import _lib, render as R, training
from nonrigid_nerf_amd.synthetic import SceneConfig, make_scene, make_rays, build_modules
import ctypes as C
dev = torch.device("cuda:0")
cfg = SceneConfig(N_importance=64)
scene = make_scene(cfg, 1)
rays, lat = make_rays(300, 3, cfg)
rays = rays.to(dev); lat = lat.to(dev).requires_grad_(True)
rb, coarse, fine = build_modules(scene, device=dev)
for prec in ("f32", "bf16"):
    R.set_precision(prec)
    model = R.get_model(coarse, fine, precision=prec, device=dev, flags=training._training_handle_flags([coarse, fine], rb))
    N, S = 300, 64
    u = torch.rand(N, S, device=dev)
    z = torch.empty(N, S, device=dev); pts = torch.empty(N, S, 3, device=dev)
    _lib.check(model.lib.nrnerf_sample_depths_points(rays.data_ptr(), int(rays.shape[1]), u.data_ptr(), N, S, 0, z.data_ptr(), pts.data_ptr(), torch.cuda.current_stream().cuda_stream), "x")
    tok = training._param_token(rb, training._bender_params(rb))
    o0 = training._Bender.apply(lat, model, rb, rays, z, tok, None)
    share = dict(e=torch.randn(N * S, 3, device=dev), pts=pts)
    o1 = training._Bender.apply(lat, model, rb, rays, z, tok, share)
    pts_fma = torch.addcmul(rays[:, None, 0:3], rays[:, None, 3:6], z[:, :, None])
    share2 = dict(e=share["e"], pts=pts_fma)
    o2 = training._Bender.apply(lat, model, rb, rays, z, tok, share2)
    for name, o in (("points with two roundings", o1), ("points by fma", o2)):
        print(prec, name, "bent", float((o0[0] - o[0]).abs().max()), "unmasked", float((o0[1] - o[1]).abs().max()), "mask", float((o0[2] - o[2]).abs().max()))
