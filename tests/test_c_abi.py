"""The boundary from plain C: tests/c_abi/render_from_c.c includes include/nrnerf.h, links libnrnerf_hip.so and renders
without Python or torch in the process; its pixels must equal the ctypes path's (same library, same weights).  Catches a
ctypes mirror (nonrigid_nerf_amd/_lib.py) that drifted from the header, and a header that is not valid C."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "tests", "c_abi", "render_from_c.c")
LIBDIR = os.path.join(REPO, "nonrigid_nerf_amd", "lib")


def test_header_compiles_as_c99_and_matches_the_ctypes_mirror(tmp_path):
    """No GPU needed: sizeof / offsetof of the ABI structures as the C compiler sees them == the ctypes mirror."""
    import ctypes as C
    from nonrigid_nerf_amd import _lib
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    probe = tmp_path / "probe.c"
    probe.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "nrnerf.h"\nint main(void) {\n'
                     'printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(nrnerf_render_args), sizeof(nrnerf_model_desc), sizeof(nrnerf_mlp_desc),\n'
                     '       offsetof(nrnerf_render_args, workspace), offsetof(nrnerf_render_args, u_coarse), offsetof(nrnerf_render_args, coarse));\n'
                     'printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(nrnerf_trunk_args), sizeof(nrnerf_composite_args), sizeof(nrnerf_profile),\n'
                     '       offsetof(nrnerf_trunk_args, d_pts4), offsetof(nrnerf_trunk_args, raw_ch), offsetof(nrnerf_composite_args, d_raw4),\n'
                     '       offsetof(nrnerf_composite_args, z_merged));\n'
                     'printf("%zu %zu %zu %zu\\n", sizeof(nrnerf_bender_args), offsetof(nrnerf_bender_args, z), offsetof(nrnerf_bender_args, bent4),\n'
                     '       offsetof(nrnerf_bender_args, d_latents));\n'
                     'printf("%zu %zu %zu %d %d %d\\n", sizeof(nrnerf_wgrad_args), sizeof(nrnerf_bender_wgrad_args), offsetof(nrnerf_wgrad_args, partials),\n'
                     '       (int)NRNERF_WGRAD_STRIDE(8, 256), (int)NRNERF_WGRAD_STRIDE(8, 128), (int)NRNERF_BENDER_WGRAD_SLOT);\n'
                     'printf("%d %d %d\\n", (int)NRNERF_WGRAD_SHORT_PARTIALS(28, 256), (int)NRNERF_WGRAD_SHORT_PARTIALS(1, 256), (int)NRNERF_WGRAD_SHORT_PARTIALS(30, 128));\n'
                     'printf("%zu %zu %zu %zu %zu\\n", sizeof(nrnerf_divergence_args), offsetof(nrnerf_divergence_args, probe),\n'
                     '       offsetof(nrnerf_divergence_args, divergence), offsetof(nrnerf_divergence_args, g_divergence), offsetof(nrnerf_divergence_args, partials));\n'
                     'printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(nrnerf_loss_args), offsetof(nrnerf_loss_args, schedule), sizeof(nrnerf_generic_trunk_args),\n'
                     '       offsetof(nrnerf_generic_trunk_args, raw_ch), offsetof(nrnerf_generic_trunk_args, d_raw4), offsetof(nrnerf_generic_trunk_args, d_enc1));\n'
                     'return 0; }\n')
    exe = tmp_path / "probe"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(REPO, "include"), str(probe), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [C.sizeof(_lib.RenderArgs), C.sizeof(_lib.ModelDesc), C.sizeof(_lib.MlpDesc), _lib.RenderArgs.workspace.offset,
            _lib.RenderArgs.u_coarse.offset, _lib.RenderArgs.coarse.offset,
            C.sizeof(_lib.TrunkArgs), C.sizeof(_lib.CompositeArgs), C.sizeof(_lib.Profile), _lib.TrunkArgs.d_pts4.offset,
            _lib.TrunkArgs.raw_ch.offset, _lib.CompositeArgs.d_raw4.offset, _lib.CompositeArgs.z_merged.offset,
            C.sizeof(_lib.BenderArgs), _lib.BenderArgs.z.offset, _lib.BenderArgs.bent4.offset, _lib.BenderArgs.d_latents.offset,
            C.sizeof(_lib.WgradArgs), C.sizeof(_lib.BenderWgradArgs), _lib.WgradArgs.partials.offset,
            _lib.wgrad_stride(8, 256), _lib.wgrad_stride(8, 128), _lib.BENDER_WGRAD_SLOT,
            _lib.wgrad_short_partials(28, 256), _lib.wgrad_short_partials(1, 256), _lib.wgrad_short_partials(30, 128),
            C.sizeof(_lib.DivergenceArgs), _lib.DivergenceArgs.probe.offset, _lib.DivergenceArgs.divergence.offset,
            _lib.DivergenceArgs.g_divergence.offset, _lib.DivergenceArgs.partials.offset,
            C.sizeof(_lib.LossArgs), _lib.LossArgs.schedule.offset, C.sizeof(_lib.GenericTrunkArgs), _lib.GenericTrunkArgs.raw_ch.offset,
            _lib.GenericTrunkArgs.d_raw4.offset, _lib.GenericTrunkArgs.d_enc1.offset]
    assert got == want, (got, want)


@pytest.mark.gpu
def test_render_from_plain_c_equals_the_python_boundary(tmp_path):
    from nonrigid_nerf_amd import render as R
    from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_rays, make_scene
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no gcc / ROCm headers on this box")
    exe = str(tmp_path / "render_from_c")
    subprocess.run([gcc, "-std=gnu99", "-D__HIP_PLATFORM_AMD__", SRC, "-I", os.path.join(REPO, "include"), "-I", "/opt/rocm/include",
                    "-L", LIBDIR, "-L", "/opt/rocm/lib", "-lnrnerf_hip", "-lamdhip64", "-Wl,-rpath," + LIBDIR,
                    "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    cfg = SceneConfig(N_importance=64)
    scene = make_scene(cfg, 0)
    order = [f"network.{i}" for i in range(5)] + [f"rigidity_network.{i}" for i in range(3)]
    nets = [(scene.bender, order), (scene.coarse, [f"pts_linears.{i}" for i in range(8)] + ["output_linear"]),
            (scene.fine, [f"pts_linears.{i}" for i in range(8)] + ["output_linear"])]
    with open(tmp_path / "weights.bin", "wb") as f:
        f.write(struct.pack("<i", 26))
        for sd, names in nets:
            for name in names:
                w = sd[name + ".weight"].numpy().astype(np.float32)
                b = sd.get(name + ".bias")
                f.write(struct.pack("<iii", w.shape[0], w.shape[1], 0 if b is None else 1))
                f.write(w.tobytes())
                if b is not None:
                    f.write(b.numpy().astype(np.float32).tobytes())
    n = 777
    rays, latents = make_rays(n, 3, cfg)
    with open(tmp_path / "rays.bin", "wb") as f:
        f.write(struct.pack("<i", n))
        f.write(rays.numpy().astype(np.float32).tobytes())
        f.write(latents.numpy().astype(np.float32).tobytes())
    res = subprocess.run([exe, str(tmp_path / "weights.bin"), str(tmp_path / "rays.bin"), str(tmp_path / "out.bin")],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr + res.stdout
    out = np.fromfile(tmp_path / "out.bin", dtype=np.float32)
    rgb_c, disp_c, acc_c = out[:3 * n].reshape(n, 3), out[3 * n:4 * n], out[4 * n:5 * n]
    rb, coarse, fine = build_modules(scene, device="cuda:0")
    R.set_precision("f32")
    with torch.no_grad():
        py = R.batchify_rays(rays.to("cuda:0"), {"ray_bending_latents": latents.to("cuda:0")}, network_fn=coarse,
                             network_fine=fine, N_samples=64, N_importance=64)
    assert np.array_equal(rgb_c, py["rgb_map"].cpu().numpy())
    assert np.array_equal(acc_c, py["acc_map"].cpu().numpy())
    assert np.array_equal(np.nan_to_num(disp_c), np.nan_to_num(py["disp_map"].cpu().numpy()))
