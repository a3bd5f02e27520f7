"""The block schedule of the 16x16x32 trunk-only kernel (csrc/nrnerf_net_x16.h: `locate`, RW / TG, the grid-stride over
groups or over tiles of blocks), restated in Python: every sample of every ray is visited exactly once, by one wave, in
both modes -- and in the fused mode (a wave owns whole rays) a ray's blocks arrive in order within one group of one wave,
which is what lets the wave composite the ray from its own LDS stage.  A model of the arithmetic, kept next to the kernel's
comments; the kernel itself is held to the oracle by tests/test_gpu_parity.py."""
import itertools

import pytest

WAVES, NB = 4, 4


def schedule(n_rays, S, grid, fuse):
    bpr = (S + 15) // 16
    RW = 1 if bpr % NB == 0 else (2 if (2 * bpr) % NB == 0 else NB)
    TG = RW * bpr // NB
    visits = []                                         # (workgroup, wave, iteration, ray, first sample, samples in block)
    if fuse:
        ngroups = (n_rays + WAVES * RW - 1) // (WAVES * RW)
        for wg in range(min(grid, ngroups)):
            it, grp = 0, wg
            while grp < ngroups:
                for tg, wave, b in itertools.product(range(TG), range(WAVES), range(NB)):
                    q = tg * NB + b
                    ray = (grp * WAVES + wave) * RW + q // bpr
                    if ray < n_rays:
                        s0 = (q % bpr) * 16
                        visits.append((wg, wave, it + tg, ray, s0, min(16, S - s0)))
                it += TG
                grp += grid
    else:
        nblocks = n_rays * bpr
        per_wg = WAVES * NB
        want = (nblocks + per_wg - 1) // per_wg
        for wg in range(min(grid, want)):
            it, b0 = 0, wg * per_wg
            while b0 < nblocks:
                for wave, b in itertools.product(range(WAVES), range(NB)):
                    blk = b0 + wave * NB + b
                    if blk < nblocks:
                        s0 = (blk % bpr) * 16
                        visits.append((wg, wave, it, blk // bpr, s0, min(16, S - s0)))
                it += 1
                b0 += grid * per_wg
    return visits, RW, TG, bpr


@pytest.mark.parametrize("fuse", [False, True])
@pytest.mark.parametrize("n_rays,S,grid", [(1, 16, 256), (3001, 85, 256), (1000, 192, 7), (513, 256, 16), (77, 97, 3), (4096, 64, 256), (9, 33, 2)])
def test_every_sample_is_visited_once(n_rays, S, grid, fuse):
    visits, RW, TG, bpr = schedule(n_rays, S, grid, fuse)
    assert RW * bpr == TG * NB                                              # a group of RW rays fills whole iterations
    seen = {}
    for wg, wave, it, ray, s0, cnt in visits:
        for s in range(s0, s0 + cnt):
            assert (ray, s) not in seen
            seen[(ray, s)] = (wg, wave, it)
    assert len(seen) == n_rays * S
    if fuse:                                                                # whole rays per wave, blocks in order, inside one group
        for ray in range(n_rays):
            owners = {seen[(ray, s)][:2] for s in range(S)}
            assert len(owners) == 1
            its = [seen[(ray, s)][2] for s in range(0, S, 16)]
            assert its == sorted(its) and its[-1] - its[0] < TG and its[0] // TG == its[-1] // TG
