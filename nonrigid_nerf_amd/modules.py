"""Parameter containers with the attribute surface the drop-in boundary reads.

The boundary (`nonrigid_nerf_amd.render`) never calls ``forward`` on the
networks it is handed: it only reads weights and a handful of attributes
(reference: ``NeRF`` run_nerf_helpers.py:172-238, ``ray_bending``
run_nerf_helpers.py:388-505; the attribute list is SURVEY.md section 8b).  The
reference modules satisfy that surface; on a machine where the reference is
not importable (the GPU box, the bench) these two holders provide the same
surface so tests and ``bench.py`` can drive the boundary exactly as
``train.py`` does.

They deliberately have NO ``forward``: they are weight holders, not a second
implementation of the networks.  The arithmetic lives in the HIP library
(product) and in ``oracle/`` (checker).
"""
from __future__ import annotations

import torch
from torch import nn


class RayBenderWeights(nn.Module):
    """Weights + editing knobs of the ray-bending network.

    Mirrors the attributes of reference ``ray_bending``
    (run_nerf_helpers.py:388-505): ``network`` (offset MLP, last layer has no
    bias), ``rigidity_network``, and the test-time knobs
    ``rigidity_test_time_cutoff`` / ``test_time_scaling`` that
    free_viewpoint_rendering.py:264-283 mutates between calls.
    """

    def __init__(self, latent_size: int = 32, hidden: int = 64, depth: int = 5,
                 rigidity_hidden: int = 32, rigidity_depth: int = 3):
        super().__init__()
        self.use_positionally_encoded_input = False
        self.input_ch = 3
        self.output_ch = 3
        self.ray_bending_latent_size = latent_size
        self.ray_bending_mode = "simple_neural"
        self.use_rigidity_network = True
        self.rigidity_test_time_cutoff = None
        self.test_time_scaling = None
        self.hidden_dimensions = hidden
        self.network_depth = depth
        self.skips = []
        self.rigidity_hidden_dimensions = rigidity_hidden
        self.rigidity_network_depth = rigidity_depth
        self.rigidity_skips = []
        dims = [3 + latent_size] + [hidden] * (depth - 1) + [3]
        self.network = nn.ModuleList(
            [nn.Linear(dims[i], dims[i + 1], bias=(i != depth - 1)) for i in range(depth)]
        )
        rdims = [3] + [rigidity_hidden] * (rigidity_depth - 1) + [1]
        self.rigidity_network = nn.ModuleList(
            [nn.Linear(rdims[i], rdims[i + 1]) for i in range(rigidity_depth)]
        )


class NeRFWeights(nn.Module):
    """Weights + attributes of one canonical NeRF MLP (coarse or fine).

    Mirrors reference ``NeRF.__init__`` (run_nerf_helpers.py:172-238): the
    trunk ``pts_linears`` with the skip-concatenation after layer ``skips``,
    ``views_linears`` (always allocated, as in the reference), and either
    ``output_linear`` or the view-dependent head.  The bender is kept in a
    1-tuple exactly like the reference (run_nerf_helpers.py:213-215) so it
    does not show up in ``parameters()``.
    """

    def __init__(self, D=8, W=256, input_ch=63, input_ch_views=0, output_ch=4,
                 skips=(4,), use_viewdirs=False, ray_bender=None,
                 ray_bending_latent_size=32, num_ray_samples=64,
                 approx_nonrigid_viewdirs=True, time_conditioned_baseline=False):
        super().__init__()
        self.D, self.W = D, W
        self.input_ch = input_ch
        self.input_ch_views = input_ch_views
        self.skips = list(skips)
        self.use_viewdirs = use_viewdirs
        self.approx_nonrigid_viewdirs = approx_nonrigid_viewdirs
        self.num_ray_samples = num_ray_samples
        self.test_time_nonrigid_object_removal_threshold = None
        self.time_conditioned_baseline = time_conditioned_baseline
        self.ray_bending_latent_size = ray_bending_latent_size
        self.ray_bender = (ray_bender,)
        net_in = input_ch + (ray_bending_latent_size if time_conditioned_baseline else 0)
        self.pts_linears = nn.ModuleList(
            [nn.Linear(net_in, W)]
            + [nn.Linear(W + net_in, W) if i in self.skips else nn.Linear(W, W)
               for i in range(D - 1)]
        )
        self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            self.rgb_linear = nn.Linear(W // 2, 3)
        else:
            self.output_linear = nn.Linear(W, output_ch)


def load_named_arrays(module: nn.Module, arrays: dict) -> nn.Module:
    """Copy a ``{state_dict key: array}`` mapping into ``module`` (strict)."""
    sd = {k: torch.as_tensor(v).clone() for k, v in arrays.items()}
    module.load_state_dict(sd, strict=True)
    return module
