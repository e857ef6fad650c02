"""The peer-to-peer halo plan (csrc/smm_dist.hip, p2pMakePlan; csrc/smm_p2p.h) replayed on the CPU for worlds the one-GPU boxes cannot
run: every rank derives ITS push / forward / land jobs from the globally known column ranges and row bounds alone; here all ranks' jobs are
executed on numpy arrays -- direct parts into the destinations' landing areas, staged shares into the relays' staging areas and on from
there, landing areas into the halos -- and every halo element of every rank must hold its owner's value, exactly once, for 2-8 ranks,
0 to world - 2 relays, nearest-neighbour halos and halos that reach several ranks."""
import ctypes

import numpy as np
import pytest

from sparse_matrix_math_amd import _lib


def _plan(lib, world, rank, needs, bounds, relays, share):
    n = ctypes.c_int()
    nd = (ctypes.c_longlong * (2 * world))(*[int(v) for v in needs])
    bd = (ctypes.c_int * (world + 1))(*[int(v) for v in bounds])
    _lib.check(lib.smm_hip_dist_p2p_plan(world, rank, nd, bd, relays, share, None, 0, ctypes.byref(n)))
    out = (ctypes.c_longlong * (8 * max(1, n.value)))()
    _lib.check(lib.smm_hip_dist_p2p_plan(world, rank, nd, bd, relays, share, out, 8 * n.value, ctypes.byref(n)))
    return np.array(out[:8 * n.value], dtype=np.int64).reshape(-1, 8)


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_every_halo_element_arrives_exactly_once(world):
    lib = _lib.load()
    rng = np.random.default_rng(world)
    for trial in range(6):
        rows = int(rng.integers(40 * world, 4000 * world))
        cuts = np.sort(rng.choice(np.arange(1, rows), size=world - 1, replace=False))
        bounds = np.concatenate(([0], cuts, [rows]))
        reach = int(rng.integers(1, rows if trial % 3 == 0 else max(2, rows // world)))  # every third trial: halos that span several ranks
        lo_reach, hi_reach = (reach, 0) if trial == 4 else (reach, int(rng.integers(0, reach + 1)))  # trial 4: one-sided (upwind) halos
        needs = np.empty(2 * world, dtype=np.int64)
        for q in range(world):
            needs[2 * q] = max(0, bounds[q] - lo_reach)
            needs[2 * q + 1] = min(rows, bounds[q + 1] + hi_reach)
        if trial == 5 and world >= 3:
            # a decoupled last block: the last rank needs nothing but its own rows and nobody needs them -- yet it relays
            needs[2 * (world - 1)], needs[2 * (world - 1) + 1] = bounds[world - 1], bounds[world]
            for q in range(world - 1):
                needs[2 * q + 1] = min(needs[2 * q + 1], bounds[world - 1])
        for relays in sorted({0, 1, max(0, world - 4), max(0, world - 2)}):
            relays = min(relays, max(0, world - 2))
            share = 4.0 / (relays + 4) if relays else 1.0
            plans = [_plan(lib, world, r, needs, bounds, relays, share) for r in range(world)]
            value = np.arange(rows, dtype=np.float64) * 3.0 + 1.0  # the owner's value of global column c
            land = [np.full(int((p[p[:, 0] == 2][:, 4]).sum()) + 1, np.nan) for p in plans]
            stage = [np.full(int(max([0] + [r[3] + ((r[5] + 3) & ~3) for p2 in plans for r in p2 if r[0] == 0 and r[2] == q])) + 1, np.nan) for q in range(world)]
            writes_land = [np.zeros(len(a), dtype=np.int64) for a in land]
            staged = {}
            for r, p in enumerate(plans):  # stage 1: pushes
                for kind, dst, relay, col, pos, count, path, job in p[p[:, 0] == 0]:
                    assert bounds[r] <= col and col + count <= bounds[r + 1]  # a rank only sends what it owns
                    if relay < 0:
                        land[dst][pos:pos + count] = value[col:col + count]
                        writes_land[dst][pos:pos + count] += 1
                    else:
                        assert relay not in (r, dst)
                        assert np.isnan(stage[relay][pos:pos + count]).all()  # staging areas of different jobs never overlap
                        stage[relay][pos:pos + count] = value[col:col + count]
                        staged[(relay, int(job))] = (r, int(dst), int(pos), int(count))
            for r, p in enumerate(plans):  # stage 2: forwards
                for kind, src, dst, spos, lpos, count, path, job in p[p[:, 0] == 1]:
                    assert staged.pop((r, int(job))) == (int(src), int(dst), int(spos), int(count))  # the source staged exactly this job here
                    land[dst][lpos:lpos + count] = stage[r][spos:spos + count]
                    writes_land[dst][lpos:lpos + count] += 1
            assert not staged  # every staged share was forwarded
            if trial == 5 and world >= 3 and relays >= world - 2 and reach > 0:
                last = plans[world - 1]
                assert not len(last[last[:, 0] == 0]) and not len(last[last[:, 0] == 2])  # it sends and receives nothing ...
                long_segments = any(rec[0] == 2 and rec[4] >= 64 * (relays + 1) for p2 in plans for rec in p2)
                assert (len(last[last[:, 0] == 1]) > 0) == long_segments  # ... and relays every segment long enough to be split: a relay-only rank
            for r, p in enumerate(plans):  # stage 3: landing areas into the halos
                cmin = needs[2 * r]
                seen = 0
                for kind, src, ext_off, land_off, count, paths, _, _ in p[p[:, 0] == 2]:
                    assert 1 <= paths <= relays + 1
                    # the land kernel waits for the FIRST `paths` parts: none of them may be empty (a short segment goes whole on the direct path)
                    parts = sorted(int(rec[6]) for q2, p2 in enumerate(plans) for rec in p2
                                   if (rec[0] == 0 and q2 == src and rec[1] == r and cmin + ext_off <= rec[3] < cmin + ext_off + count))
                    assert parts == list(range(int(paths))), (parts, paths)
                    got = land[r][land_off:land_off + count]
                    cols = cmin + ext_off + np.arange(count)
                    assert ((cols >= bounds[src]) & (cols < bounds[src + 1])).all()
                    np.testing.assert_array_equal(got, value[cols])
                    assert (writes_land[r][land_off:land_off + count] == 1).all()  # every element written exactly once
                    seen += count
                own = bounds[r + 1] - bounds[r]
                assert seen == (needs[2 * r + 1] - needs[2 * r]) - own  # the whole halo is covered


def test_plan_is_refused_for_bad_arguments():
    lib = _lib.load()
    n = ctypes.c_int()
    assert lib.smm_hip_dist_p2p_plan(0, 0, None, None, 0, 1.0, None, 0, ctypes.byref(n)) != 0
    assert lib.smm_hip_dist_p2p_plan(2, 5, None, None, 0, 1.0, None, 0, ctypes.byref(n)) != 0
