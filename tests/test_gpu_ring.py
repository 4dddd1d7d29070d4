"""GPU tests of the global ring step (ring.RingLUM on ring.HipBackend): equals
the oracle's registrationLUM pass on the golden 12-view fixture, and a 1-GPU
"fake world" that runs the N-rank partitioning serially reproduces the
unsharded edge table (the reduction the RCCL all-reduce performs)."""
import importlib

import numpy as np
import pytest

from conftest import PKG, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ring(mvr):
    return importlib.import_module(PKG + ".ring")


def test_ring_step_matches_oracle_lum_pass(mvr, ring):
    g = load_golden("ring_12x2048.npz")
    scans, poses0 = list(g["scans"]), list(g["poses0"])
    sp = mvr.synth_params(12, 3)
    be = ring.HipBackend(scans, device=0)
    try:
        r = ring.RingLUM(be, 12, [len(s) for s in scans], 8.0, np.array(sp.pivot))
        new = r.step([p.copy() for p in poses0])
        assert [int(n) for n in r.last["pair_n"]] == list(g["lum_ncorr"])
        assert r.last["lum_iterations"] == int(g["lum_its"][0])
        assert np.abs(r.last["lum_pose"] - g["lum_P"]).max() < 1e-6
        for v in range(12):
            assert np.abs(new[v][:3, :3] - g["lum_poses"][v][:3, :3]).max() < 1e-5
            assert np.abs(new[v][:3, 3] - g["lum_poses"][v][:3, 3]).max() < 1e-4
        # per-pair rigid solve: T_e from the moments brings each source onto its target
        for T, n in zip(r.last["pair_T"], r.last["pair_n"]):
            assert n > 500 and abs(np.linalg.det(T[:3, :3].astype(np.float64)) - 1) < 1e-5
    finally:
        be.close()


def test_golden_ring_36_views_three_passes(mvr):
    """the committed 36-view fixture (tests/golden/ring_36x768.npz: BASELINE configs[4]'s shape in small, made by the oracle's
    registrationLUM pass, registrator.cpp:625-664): three outer passes from the prior through mvr_ring_step -- per-edge
    correspondence counts equal, the LUM poses to 1e-6, the views' poses within 1e-5 / 1e-4 mm after every pass.  A fixture
    rather than a fresh oracle run: an oracle that drifted together with the product would still be caught."""
    g = load_golden("ring_36x768.npz")
    scans, poses = list(g["scans"]), [p.copy() for p in g["poses0"]]
    V = len(scans)
    edges = [(i, (i + 1) % V) for i in range(V)]
    with mvr.Context(0) as ctx:
        for v in range(V):
            ctx.upload(V + v, scans[v])
        for k in range(g["lum_poses"].shape[0]):
            poses, info = ctx.ring_step(list(range(V)), [V + v for v in range(V)], edges, poses, 8.0, g["origin"])
            assert [int(n) for n in info["pair_n"]] == list(g["lum_ncorr"][k]), k
            assert info["lum_iterations"] == int(g["lum_its"][k])
            assert np.abs(np.asarray(info["lum_pose"]) - g["lum_P"][k]).max() < 1e-6
            for v in range(V):
                assert np.abs(poses[v][:3, :3] - g["lum_poses"][k][v][:3, :3]).max() < 1e-5
                assert np.abs(poses[v][:3, 3] - g["lum_poses"][k][v][:3, 3]).max() < 1e-4


def test_golden_sequential_sweeps(mvr):
    """the committed 12-view fixture's SEQUENTIAL record (two sweeps of registrationICP, registrator.cpp:526-588, by the oracle
    driver): mvr_seq_run, the native loop, align by align -- views in the reference's order, correspondence counts equal, mean
    squared distances to 1e-9, every align's transformation and the final poses within 1e-5 / 1e-4 mm."""
    g = load_golden("ring_12x2048.npz")
    scans, poses0 = list(g["scans"]), [p.copy() for p in g["poses0"]]
    V = len(scans)
    RAW, TARGET, SOURCE, OUT = V, 2 * V, 2 * V + 1, 2 * V + 2
    with mvr.Context(0) as ctx:
        for v in range(V):
            ctx.upload(RAW + v, scans[v])
        poses, log = ctx.seq_run([RAW + v for v in range(V)], TARGET, SOURCE, OUT, mvr.icp_params(max_dist=8.0, max_iter=1000), poses0, repeat=2)
    assert [e["view"] for e in log] == list(g["seq_view"])
    assert [e["n_corr"] for e in log] == list(g["seq_ncorr"])
    for e, T, mse in zip(log, g["seq_T"], g["seq_mse"]):
        assert abs(e["mse"] - mse) < 1e-9
        assert np.abs(e["T"][:3, :3] - T[:3, :3]).max() < 1e-5 and np.abs(e["T"][:3, 3] - T[:3, 3]).max() < 1e-4
    for v in range(V):
        assert np.abs(poses[v][:3, :3] - g["seq_poses"][v][:3, :3]).max() < 1e-5
        assert np.abs(poses[v][:3, 3] - g["seq_poses"][v][:3, 3]).max() < 1e-4


def test_batched_orderings_are_the_one_by_one_orderings(mvr):
    """A registration's first pass builds the orderings of all its scans in ONE composite sort ((scan << 30) | Hilbert code,
    stable: round 4) -- the permutations must be the ones the one-scan-at-a-time build gives (the sums of a pass are taken in
    this order: any difference shows in the last bits of a pose), built from the POSED copies either way."""
    V, N = 12, 9000
    sp = mvr.synth_params(V, 3)
    scans = [mvr.synth_view(sp, v, N + 17 * v) for v in range(V)]          # ragged sizes
    piv, ax = mvr.synth_prior(sp)
    poses0 = [np.eye(4)] + [mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V)) for v in range(1, V)]
    edges = [(i, (i + 1) % V) for i in range(V)]
    perms = []
    for kn in (dict(order_batch=1), dict(order_batch=0)):
        with mvr.Context(0) as ctx:
            ctx.tune(**kn)
            for v in range(V):
                ctx.upload(V + v, scans[v])
            ctx.ring_step(list(range(V)), [V + v for v in range(V)], edges, [p.copy() for p in poses0], 4.0, np.array(sp.pivot))
            perms.append([ctx.debug_order(v) for v in range(V)])
            raw = [ctx.debug_order(V + v) for v in range(V)]
            for v in range(V):                      # a scan and its posed copy share one ordering
                assert np.array_equal(raw[v], perms[-1][v])
    for v in range(V):
        assert perms[0][v] is not None and len(perms[0][v]) == len(scans[v])
        assert np.array_equal(np.sort(perms[0][v]), np.arange(len(scans[v]), dtype=np.uint32))
        assert np.array_equal(perms[0][v], perms[1][v]), v


def test_one_call_step_equals_three_call_step(mvr, ring):
    """mvr_ring_step (posing, searches, reductions, table copy and host solve in one native call) walks exactly the
    poses of the step driven from Python through mvr_cloud_transform_batch / mvr_pair_moments2_batch /
    mvr_ring_host_step, over several outer passes."""
    g = load_golden("ring_12x2048.npz")
    scans, poses0 = list(g["scans"]), list(g["poses0"])
    origin = np.array(mvr.synth_params(12, 3).pivot)
    runs = []
    for fused in (True, False):
        be = ring.HipBackend(scans, device=0)
        try:
            r = ring.RingLUM(be, 12, [len(s) for s in scans], 8.0, origin, fused=fused)
            poses, log = [p.copy() for p in poses0], []
            for _ in range(3):
                poses = r.step(poses)
                log.append((np.stack(poses).tobytes(), [int(n) for n in r.last["pair_n"]], r.last["lum_iterations"],
                            np.asarray(r.last["lum_pose"]).tobytes(), float(r.last["mse"])))
            if fused:
                assert r.last["rows"].shape == (12, 32) and [int(x) for x in r.last["rows"][:, 0]] == log[-1][1]
                assert all(t >= 0 for t in r.last["timing_ms"])
            runs.append(log)
            if fused:                     # the native loop (mvr_ring_run) walks the same poses as step-by-step calls
                r2 = ring.RingLUM(be, 12, [len(s) for s in scans], 8.0, origin)
                p3 = r2.run([p.copy() for p in poses0], 3)
                assert np.asarray(p3).tobytes() == log[-1][0] and [int(n) for n in r2.last["pair_n"]] == log[-1][1]
                assert r2.last["ms_drain"] > 0
        finally:
            be.close()
    assert runs[0] == runs[1]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_fake_world_partition_sums_to_unsharded(mvr, ring, world):
    g = load_golden("ring_12x2048.npz")
    scans, poses0 = list(g["scans"]), list(g["poses0"])
    sp = mvr.synth_params(12, 3)
    origin = np.array(sp.pivot)
    be = ring.HipBackend(scans, device=0)
    try:
        be.pose_clouds(poses0)
        edges = ring.ring_edges(12)
        sizes = [len(scans[s]) for s, _ in edges]
        full = be.to_host(be.edge_rows(ring.split_queries(sizes, 1, 0), edges, 8.0, origin)).copy()
        total = np.zeros_like(full)
        for rank in range(world):
            total += be.to_host(be.edge_rows(ring.split_queries(sizes, world, rank), edges, 8.0, origin))
        assert np.array_equal(total[:, 0], full[:, 0])
        assert np.allclose(total[:, 4:], full[:, 4:], rtol=1e-12, atol=1e-7)
    finally:
        be.close()


@pytest.mark.parametrize("streams", [1, 3, 4, 6])
def test_batched_pairs_equal_single_calls(gpu, mvr, streams):
    """mvr_pair_moments2_batch (pairs on concurrent worker streams) returns exactly the sums of
    one mvr_pair_moments2 call per pair, with and without query sub-ranges, whatever the stream count."""
    g = load_golden("ring_12x2048.npz")
    scans, poses0 = list(g["scans"]), list(g["poses0"])
    sp = mvr.synth_params(12, 3)
    origin = np.array(sp.pivot)
    V = 7
    for v in range(V):
        gpu.upload(V + v, scans[v])
        gpu.transform(v, V + v, poses0[v])
    # 16 pairs: enough for the fused pass to split them into 2 and into 4 groups (it wants four pairs per group)
    pairs = [(v, (v + 1) % V) for v in range(V)] + [(2, 5), (5, 2)] + [(v, (v + 2) % V) for v in range(V)]
    ranges = [(0, None), (100, 700), (0, 0), (2000, None), (5, 1), (0, None), (1024, 1024), (3, 2040), (0, None)] + \
             [(0, None), (7, 1000), (0, None), (1500, 548), (0, None), (0, 1), (64, 64)]
    # streams == 3: the worker-stream path also in culled mode; the fused pass in 1, 2 (default) or 4 groups of pairs
    gpu.tune(pair_streams=streams, pair_fused=int(streams != 3), pair_groups={1: 1, 3: 1, 6: 2}.get(streams, 4))
    try:
        for rg in (None, ranges):
            single = [gpu.pair_moments2(s, t, 8.0, origin, q_begin=0 if rg is None else rg[k][0],
                                        q_count=None if rg is None else rg[k][1]) for k, (s, t) in enumerate(pairs)]
            batch = gpu.pair_moments2_batch(pairs, 8.0, origin, ranges=rg)
            assert len(batch) == len(pairs)
            for k, (a, b) in enumerate(zip(single, batch)):
                assert bytes(a) == bytes(b), (k, pairs[k], a.n, b.n)
            assert sum(m.n for m in batch) > 1000
        # an empty batch is a no-op
        assert gpu.pair_moments2_batch([], 8.0, origin) == []
    finally:
        gpu.tune(pair_streams=6, pair_fused=1, pair_groups=2)


def test_grid_search_equals_culled_search_over_passes(mvr, orc):
    """The fused pass answers its BOUNDED queries (forward searches seeded by the previous pass's matches, all reverse
    searches) with the thread-per-query grid search (mvr_grid.hip) and the rest with the culled kernel: same edge
    tables, same poses, bit for bit, as with the culled kernel alone -- over several passes from the prior (so that
    unseeded, freshly seeded and well seeded passes all occur), at 12 views x 20k points and 36 x 3k, and the table of
    the last pass equals the oracle's correspondences."""
    for V, N, max_d in ((12, 20000, 4.0), (36, 3000, 8.0)):
        sp = mvr.synth_params(V, 3)
        scans = [mvr.synth_view(sp, v, N) for v in range(V)]
        piv, ax = mvr.synth_prior(sp)
        poses0 = [np.eye(4)] + [mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V)) for v in range(1, V)]
        origin = np.array(sp.pivot)
        edges = [(i, (i + 1) % V) for i in range(V)]
        runs = []
        for mode in (0, 1):
            with mvr.Context(0) as ctx:
                ctx.tune(ring_search=mode)
                for v in range(V):
                    ctx.upload(V + v, scans[v])
                poses, log = [p.copy() for p in poses0], []
                for _ in range(5):
                    poses, info = ctx.ring_step(list(range(V)), [V + v for v in range(V)], edges, poses, max_d, origin)
                    log.append((np.asarray(poses).tobytes(), info["rows"].tobytes()))
                runs.append(log)
                if mode == 1:
                    last_in = np.frombuffer(log[-2][0]).reshape(V, 4, 4)          # poses that went INTO the last pass
                    clouds = [orc.transform_f64(last_in[v], scans[v]) for v in range(V)]
                    for e in (0, V // 2, V - 1):
                        cc = orc.correspondences(clouds[edges[e][0]], clouds[edges[e][1]], max_d, kdtree=True)
                        assert int(info["rows"][e, 0]) == len(cc)
                        assert abs(info["rows"][e, 31] - float(cc["dist2"].astype(np.float64).sum())) < 1e-9 * max(1.0, info["rows"][e, 31])
        assert runs[0] == runs[1], (V, N)


@pytest.mark.parametrize("knobs", [
    dict(grid_lanes=2), dict(grid_lanes=4), dict(grid_lanes=8, grid_cell_points=20),
    dict(grid_wide=0), dict(grid_wide=0, grid_light_rows=1), dict(grid_light_rows=1), dict(grid_light_rows=64),
    dict(cull_list=0), dict(cull_list_w=1), dict(cull_list_w=4, grid_cluster=1), dict(grid_cluster=65),
    dict(grid_cell_points=1), dict(grid_cell_points=40, grid_light_rows=3), dict(fused_mark=0), dict(fused_mark=2), dict(grid_sets=0), dict(grid_tail=0), dict(grid_sets=2, grid_light_rows=1, grid_cluster=1), dict(grid_sets=2, grid_cell_points=1, grid_light_rows=2, grid_cluster=2), dict(fused_mark=0, grid_lanes=4, pair_groups=1),
    dict(grid_wide_waves=1), dict(pair_groups=3, cull_slices=8),
    dict(grid_probe_rows=1), dict(grid_probe_rows=4, grid_cell_points=1), dict(grid_probe_rows=3, grid_light_rows=64, grid_cell_points=40),
    dict(grid_probe=0), dict(grid_probe=0, grid_light_rows=2), dict(grid_probe=1, grid_light_rows=1, grid_cluster=1), dict(setup_first=0), dict(cull_w=4), dict(cull_w=2),
    dict(lazy_super=0), dict(lazy_super=0, grid_sets=0), dict(lazy_super=1, grid_tail=0, cull_list=0),
    # the staged walk (round 4): off, for the reverse launches too, with tiny cells (more rows in a wave's box than its table holds),
    # with huge cells and wide balls walked in-thread (more points than a wave's LDS holds: the rows behind are walked from global memory)
    dict(grid_stage=0), dict(grid_stage=2), dict(grid_stage=2, grid_cell_points=1), dict(grid_stage=2, grid_cell_points=40, grid_light_rows=64),
    dict(order_batch=0), dict(order_batch=0, grid_stage=0),
    # rim certificates (round 4): off, with a margin of 1 mm, with the compact index and the staged walk everywhere
    dict(rim_cert_um=50), dict(rim_cert_um=1000), dict(rim_cert_um=1000, seed_delta_um=1000, grid_stage=2, grid_index=1), dict(rim_cert_um=200, pipeline=0),
    # seed_delta (round 4): start bounds from the previous distance + the clouds' motion instead of the old match's coordinates: off, always on
    dict(seed_delta_um=50), dict(seed_delta_um=100000), dict(seed_delta_um=100000, grid_stage=0), dict(seed_delta_um=100000, pipeline=0, grid_index=1),
    # the compact cell-start tables (round 4): forced for these small grids, with the staged walk everywhere, with tiny / huge cells, dense forced
    dict(grid_index=1), dict(grid_index=1, grid_stage=2), dict(grid_index=1, grid_cell_points=1), dict(grid_index=1, grid_cell_points=40, grid_light_rows=64, grid_probe=0),
    dict(grid_index=1, grid_stage=0, grid_lanes=4), dict(grid_index=1, grid_sets=2, grid_light_rows=1, grid_cluster=1), dict(grid_index=0),
    dict(grid_stage=2, grid_light_rows=64, grid_probe=0), dict(grid_stage=1, grid_cell_points=2, grid_light_rows=64, grid_probe=0), dict(grid_stage=2, grid_cell_points=12, grid_wide=0),
], ids=lambda k: ",".join("%s=%s" % kv for kv in k.items()))
def test_grid_search_knobs_never_show_in_a_result(mvr, knobs):
    """Every routing knob of the grid search (lanes per query, cell size, what counts as a wide ball, where the wide ones
    go, set lists, marking) only moves queries between three exact searches: six passes of the 12 x 20k ring from the
    prior give the same poses and edge tables, bit for bit, as the default settings (so do the probe for wide balls, the order of
    the set-up of a plain pass, the waves per query set of the culled kernel, and who refreshes the super boxes)."""
    V, N, max_d = 12, 20000, 4.0
    sp = mvr.synth_params(V, 3)
    scans = [mvr.synth_view(sp, v, N) for v in range(V)]
    piv, ax = mvr.synth_prior(sp)
    poses0 = [np.eye(4)] + [mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V)) for v in range(1, V)]
    origin = np.array(sp.pivot)
    edges = [(i, (i + 1) % V) for i in range(V)]
    runs = []
    for kn in ({}, knobs):
        with mvr.Context(0) as ctx:
            ctx.tune(**kn)
            for v in range(V):
                ctx.upload(V + v, scans[v])
            poses, log = [p.copy() for p in poses0], []
            for _ in range(6):
                poses, info = ctx.ring_step(list(range(V)), [V + v for v in range(V)], edges, poses, max_d, origin)
                log.append((np.asarray(poses).tobytes(), info["rows"].tobytes()))
            runs.append(log)
    assert runs[0] == runs[1], knobs


@pytest.mark.parametrize("n_points", [20000, 3000])
def test_unseeded_first_pass_through_the_grid(mvr, n_points):
    """unseeded_grid = 1: the forward searches of a registration's FIRST pass -- no seeds yet, the scans' grids being built on the
    side stream -- walk the grid as well (every query probes the cells next to it; what finds nothing there goes to the listed
    sets or the culled kernel) instead of the culled kernel: five passes in one call from a fresh context give the same poses
    and edge tables, bit for bit, with the probe off, with the sets on the culled kernel and the plain walk too."""
    V, max_d = 12, 4.0
    sp = mvr.synth_params(V, 3)
    scans = [mvr.synth_view(sp, v, n_points) for v in range(V)]
    piv, ax = mvr.synth_prior(sp)
    poses0 = [np.eye(4)] + [mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V)) for v in range(1, V)]
    origin = np.array(sp.pivot)
    edges = [(i, (i + 1) % V) for i in range(V)]
    runs = []
    for kn in ({}, dict(unseeded_grid=1), dict(unseeded_grid=1, grid_probe=0), dict(unseeded_grid=1, grid_sets=0, grid_stage=0), dict(unseeded_grid=1, grid_index=1, grid_light_rows=2)):
        with mvr.Context(0) as ctx:
            ctx.tune(**kn)
            for v in range(V):
                ctx.upload(V + v, scans[v])
            poses, info = ctx.ring_step(list(range(V)), [V + v for v in range(V)], edges, [p.copy() for p in poses0], max_d, origin, steps=5)
            one, info1 = ctx.ring_step(list(range(V)), [V + v for v in range(V)], edges, [p.copy() for p in poses0], max_d, origin, steps=1)      # (a restart: seeds 2 mm off)
            runs.append((np.asarray(poses).tobytes(), info["rows"].tobytes(), np.asarray(one).tobytes(), info1["rows"].tobytes()))
    for r in runs[1:]:
        assert r == runs[0]


def test_a_culled_pass_after_grid_passes_finds_its_super_boxes(mvr):
    """The posing launch of a pass on the grid kernels leaves the views' super boxes (read by the culled kernel alone) stale
    (lazy_super); a pass that then takes the culled kernel -- ring_search switched off in mid-registration, a one-pair search
    against a posed view -- must bring them up to date first: same bits as a context that never used the grid."""
    V, N, max_d = 12, 15000, 4.0
    sp = mvr.synth_params(V, 3)
    scans = [mvr.synth_view(sp, v, N) for v in range(V)]
    piv, ax = mvr.synth_prior(sp)
    poses0 = [np.eye(4)] + [mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V)) for v in range(1, V)]
    origin = np.array(sp.pivot)
    edges = [(i, (i + 1) % V) for i in range(V)]
    runs = []
    for switch in (False, True):
        with mvr.Context(0) as ctx:
            ctx.tune(ring_search=1 if switch else 0)
            for v in range(V):
                ctx.upload(V + v, scans[v])
            poses, log = [p.copy() for p in poses0], []
            for k in range(7):
                if switch and k in (4, 6):
                    ctx.tune(ring_search=0)          # this pass: culled kernel, over boxes the grid passes did not keep up
                elif switch:
                    ctx.tune(ring_search=1)
                poses, info = ctx.ring_step(list(range(V)), [V + v for v in range(V)], edges, poses, max_d, origin)
                log.append((np.asarray(poses).tobytes(), info["rows"].tobytes()))
            if switch:
                ctx.tune(ring_search=1)
            poses, info = ctx.ring_step(list(range(V)), [V + v for v in range(V)], edges, poses, max_d, origin)      # (a grid pass last when switching)
            log.append((np.asarray(poses).tobytes(), info["rows"].tobytes()))
            # a one-pair culled search against a posed view right after a grid pass
            q, m, d = ctx.correspondences(3, 4, max_d)        # (posed slots 3 and 4: mvr_correspondences runs the culled kernel)
            runs.append((log, q.tobytes(), m.tobytes(), d.tobytes()))
    assert runs[0] == runs[1]


@pytest.mark.parametrize("case", ["fma", "sheared_pose", "slightly_sheared_pose", "sub_ranges"])
def test_grid_search_equals_culled_search_off_the_main_road(mvr, case):
    """ring_search 1 == ring_search 0, bit for bit, also (a) with the fused multiply-add form of the distance, (b) when a
    view's pose is not rigid (a sheared matrix: the grid of that scan cannot be mapped into by an inverse pose, the
    pass must notice and keep the culled kernel) and (c) when a rank only owns sub-ranges of the queries."""
    V, N, max_d = 12, 15000, 4.0
    sp = mvr.synth_params(V, 3)
    scans = [mvr.synth_view(sp, v, N) for v in range(V)]
    piv, ax = mvr.synth_prior(sp)
    poses0 = [np.eye(4)] + [mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V)) for v in range(1, V)]
    if case == "sheared_pose":
        poses0[5] = poses0[5].copy(); poses0[5][0, 1] += 3e-3
    if case == "slightly_sheared_pose":          # within the bar: the grid is used, with a ball widened by the pose's stretch
        poses0[5] = poses0[5].copy(); poses0[5][0, 1] += 2e-4; poses0[7] = poses0[7].copy(); poses0[7][:3, :3] *= 1.0003
    origin = np.array(sp.pivot)
    edges = [(i, (i + 1) % V) for i in range(V)]
    runs = []
    for mode in (0, 1):
        with mvr.Context(0) as ctx:
            ctx.tune(ring_search=mode)
            for v in range(V):
                ctx.upload(V + v, scans[v])
            log = []
            if case == "sub_ranges":
                ranges = [(0, None), (100, 7000), (0, 0), (14000, None), (5, 1), (0, None), (1024, 1024), (3, 14040), (0, None), (7, 1000), (0, None), (64, 64)]
                for it in range(4):
                    P = [p.copy() for p in poses0]
                    for v in range(1, V):
                        P[v] = mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V) + 1e-4 * it * v)      # the clouds move a little between the passes
                    ctx.transform_batch(list(range(V)), [V + v for v in range(V)], P)
                    rows = ctx.pair_moments2_batch(edges, max_d, origin, ranges=ranges)
                    log.append(b"".join(bytes(r) for r in rows))
            else:
                poses = [p.copy() for p in poses0]
                for _ in range(5):
                    poses, info = ctx.ring_step(list(range(V)), [V + v for v in range(V)], edges, poses, max_d, origin, fma=(case == "fma"))
                    log.append((np.asarray(poses).tobytes(), info["rows"].tobytes()))
            runs.append(log)
    assert runs[0] == runs[1], case


def test_grid_search_survives_the_drift_of_a_long_registration(mvr):
    """The poses of pass k are products of k float-rounded LUM matrices: after a few hundred passes they are off
    orthonormal by 1e-5 and more.  The grid search must neither give up (a 2e-6 rigidity bar once sent every pass after
    the ~200th back to the culled kernel, unnoticed: same results, twice the time) nor lose a neighbour: 600 passes of
    the 12 x 20k ring end in the same poses and table, bit for bit, with and without it, and the last pass still runs
    on the grid kernels."""
    V, N, max_d = 12, 20000, 4.0
    sp = mvr.synth_params(V, 3)
    scans = [mvr.synth_view(sp, v, N) for v in range(V)]
    piv, ax = mvr.synth_prior(sp)
    poses0 = [np.eye(4)] + [mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V)) for v in range(1, V)]
    origin = np.array(sp.pivot)
    edges = [(i, (i + 1) % V) for i in range(V)]
    runs = []
    for mode in (0, 1):
        with mvr.Context(0) as ctx:
            ctx.tune(ring_search=mode)
            for v in range(V):
                ctx.upload(V + v, scans[v])
            poses, info = ctx.ring_step(list(range(V)), [V + v for v in range(V)], edges, [p.copy() for p in poses0], max_d, origin, steps=600)
            ctx.prof_reset(); ctx.prof_enable(1)
            poses, info = ctx.ring_step(list(range(V)), [V + v for v in range(V)], edges, poses, max_d, origin, steps=1)
            ctx.prof_enable(False)
            grid_launches = ctx.prof_get(mvr.K_NN_GRID)[0]
            assert (grid_launches > 0) == (mode == 1), (mode, grid_launches)
            runs.append((np.asarray(poses).tobytes(), info["rows"].tobytes()))
            drift = max(np.abs(np.asarray(p)[:3, :3].T @ np.asarray(p)[:3, :3] - np.eye(3)).max() for p in poses)
            assert drift > 2e-6 or mode == 0, drift          # (the case this test is about did occur)
    assert runs[0] == runs[1]


def test_grid_search_through_the_events_of_a_session(mvr):
    """ring_search 1 == ring_search 0, pass by pass, through what a session does between passes: scans of different
    sizes, a scan uploaded again (its grid goes with the old point set), a different search radius (the distance map
    was built for the first one), a different set of pairs (no seeds for that pass), a view posed by hand."""
    V, max_d = 8, 4.0
    sp = mvr.synth_params(V, 3)
    sizes = [15000 - 1700 * v for v in range(V)]
    scans = [mvr.synth_view(sp, v, sizes[v]) for v in range(V)]
    piv, ax = mvr.synth_prior(sp)
    poses0 = [np.eye(4)] + [mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V)) for v in range(1, V)]
    origin = np.array(sp.pivot)
    ring = [(i, (i + 1) % V) for i in range(V)]
    runs = []
    for mode in (0, 1):
        with mvr.Context(0) as ctx:
            ctx.tune(ring_search=mode)
            for v in range(V):
                ctx.upload(V + v, scans[v])
            poses, log = [p.copy() for p in poses0], []
            for k in range(12):
                edges, md = ring, max_d
                if k == 3:
                    ctx.upload(V + 2, scans[2])                                   # the same scan again: a new point set
                if k == 5:
                    ctx.upload(V + 4, scans[4][: sizes[4] - 999])                 # ... and a shorter one
                if k in (6, 7):
                    md = 7.5                                                      # wider than the distance map was built for
                if k == 8:
                    edges = [(b, a) for a, b in ring]                             # other pairs: nothing to seed from
                if k == 10:
                    poses[3] = mvr.axis_rotation(piv, ax, mvr.turntable_angle(3, V) + 2e-3)
                poses, info = ctx.ring_step(list(range(V)), [V + v for v in range(V)], edges, poses, md, origin)
                log.append((np.asarray(poses).tobytes(), info["rows"].tobytes()))
            runs.append(log)
    for k, (a, b) in enumerate(zip(*runs)):
        assert a == b, k


@pytest.mark.parametrize("seed", range(8))
def test_grid_search_equals_culled_search_on_odd_rings(mvr, seed):
    """ring_search 1 == ring_search 0 over six passes on rings nobody would scan: empty and one-point views, views of
    63 / 64 / 65 / 257 points, duplicated points (ties: the lowest index must win in both searches), a view that is a
    line, views that barely overlap, radii from a tenth of the point spacing to the whole object."""
    rng = np.random.default_rng(1000 + seed)
    V = int(rng.integers(3, 7))
    sp = mvr.synth_params(V, 3)
    piv, ax = mvr.synth_prior(sp)
    sizes = [int(rng.choice([0, 1, 2, 63, 64, 65, 257, 1500, 4000, 9000])) for _ in range(V)]
    sizes[int(rng.integers(V))] = 6000                                    # at least one real view
    scans = []
    for v in range(V):
        c = mvr.synth_view(sp, v, max(sizes[v], 1))[: sizes[v]].copy()
        if len(c) > 10 and rng.random() < 0.5:                               # duplicates, some of them many times over
            k = int(rng.integers(1, len(c) // 3 + 1))
            c[rng.integers(0, len(c), k)] = c[rng.integers(0, len(c), 1)]
            c[-k:] = c[:k]
        if len(c) > 100 and rng.random() < 0.2:                              # a line
            c[:, 1] = c[0, 1]; c[:, 2] = c[0, 2]
        scans.append(c)
    poses0 = [np.eye(4)] + [mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V) + float(rng.normal(0, 2e-3))) for v in range(1, V)]
    max_d = float(rng.choice([0.05, 0.5, 4.0, 25.0, 300.0]))
    origin = np.array(sp.pivot)
    edges = [(i, (i + 1) % V) for i in range(V)] + ([(0, 2)] if rng.random() < 0.5 else [])
    runs = []
    for mode in (0, 1):
        with mvr.Context(0) as ctx:
            ctx.tune(ring_search=mode)
            for v in range(V):
                ctx.upload(V + v, scans[v])
            poses, log = [p.copy() for p in poses0], []
            for _ in range(6):
                ctx.transform_batch(list(range(V)), [V + v for v in range(V)], poses)
                rows = ctx.pair_moments2_batch(edges, max_d, origin)
                log.append(b"".join(bytes(r) for r in rows))
                poses = [p @ mvr.axis_rotation(piv, ax, float(rng.normal(0, 1e-4))) if i else p for i, p in enumerate(poses)] if mode < 0 else \
                        [poses[i] if i == 0 else mvr.axis_rotation(piv, ax, mvr.turntable_angle(i, V) + 1e-4 * (len(log) + i)) for i in range(V)]
            runs.append(log)
    assert runs[0] == runs[1], (seed, sizes, max_d)


def test_transform_batch_equals_single_transforms(gpu, mvr):
    """mvr_cloud_transform_batch poses many clouds in one launch, bit for bit like mvr_cloud_transform."""
    g = load_golden("ring_12x2048.npz")
    scans, poses0 = list(g["scans"]), list(g["poses0"])
    V = 5
    for v in range(V):
        gpu.upload(20 + v, scans[v][: 2048 - 37 * v])          # ragged sizes
    for v in range(V):
        gpu.transform(v, 20 + v, poses0[v])
    single = [gpu.download(v) for v in range(V)]
    gpu.transform_batch([10 + v for v in range(V)], [20 + v for v in range(V)], poses0[:V])
    for v in range(V):
        assert np.array_equal(gpu.download(10 + v).view(np.uint32), single[v].view(np.uint32))
    # in place, and searchable afterwards (the index follows the new coordinates)
    gpu.transform_batch([20, 21], [20, 21], poses0[:2])
    assert np.array_equal(gpu.download(20).view(np.uint32), single[0].view(np.uint32))
    a = gpu.pair_moments2(20, 21, 8.0, np.zeros(3))
    b = gpu.pair_moments2(0, 1, 8.0, np.zeros(3))
    assert bytes(a) == bytes(b) and a.n > 100


def test_transform_batch_refuses_aliased_slots(gpu, mvr):
    """The entries of mvr_cloud_transform_batch run in one launch: a destination that appears twice, or that is another
    entry's source, is an argument error -- also through mvr_ring_step, which forwards its slot lists;
    nothing is written in that case."""
    g = load_golden("ring_12x2048.npz")
    scans, poses0 = list(g["scans"]), list(g["poses0"])
    for v in range(3):
        gpu.upload(30 + v, scans[v])
    gpu.transform_batch([33, 34], [30, 31], poses0[:2])
    before = gpu.download(33).copy()
    for dst, src in (([33, 33], [30, 31]), ([33, 30], [30, 31]), ([31, 34], [30, 31]), ([33, 34, 33], [30, 31, 32])):
        with pytest.raises(mvr.MvrError) as e:
            gpu.transform_batch(dst, src, poses0[:len(dst)])
        assert e.value.status == mvr.E_ARG
    assert np.array_equal(gpu.download(33), before)
    with pytest.raises(mvr.MvrError) as e:      # posed slot of view 1 == raw slot of view 0
        gpu.ring_step([33, 30, 34], [30, 31, 32], [(0, 1), (1, 2), (2, 0)], poses0[:3], 8.0, np.zeros(3))
    assert e.value.status == mvr.E_ARG
    gpu.transform_batch([30, 34], [30, 31], poses0[:2])          # in place + a plain entry: fine
    assert np.array_equal(gpu.download(30).view(np.uint32), before.view(np.uint32))


def test_posed_index_refresh_equals_lazy_refresh(gpu, mvr):
    """In culled mode mvr_cloud_transform_batch also brings the posed copies' index up to date, straight from the
    sources' sorted copies (tune key posed_refresh).  Same searches, bit for bit, as the lazy gather refresh --
    over repeated re-posing, in place, and for a posed copy of a posed copy."""
    g = load_golden("ring_12x2048.npz")
    scans, poses0 = list(g["scans"]), list(g["poses0"])
    V = 6
    origin = np.array(mvr.synth_params(12, 3).pivot)
    pairs = [(v, (v + 1) % V) for v in range(V)]
    nudge = np.eye(4); nudge[:3, 3] = (0.3, -0.2, 0.1); nudge[0, 1], nudge[1, 0] = -0.01, 0.01
    outs = []
    try:
        for flag in (0, 1):
            gpu.tune(posed_refresh=flag)
            rec = []
            for v in range(V):
                gpu.upload(20 + v, scans[v][: 2048 - 11 * v])            # fresh point sets: the ordering is rebuilt
            gpu.transform_batch(list(range(V)), [20 + v for v in range(V)], poses0[:V])
            rec += [bytes(m) for m in gpu.pair_moments2_batch(pairs, 8.0, origin)]
            gpu.transform_batch(list(range(V)), [20 + v for v in range(V)], [nudge @ p for p in poses0[:V]])     # re-posed: orders and source copies reused
            rec += [bytes(m) for m in gpu.pair_moments2_batch(pairs, 8.0, origin)]
            gpu.transform_batch([9, 8], [20, 21], [poses0[0], poses0[1]])
            gpu.transform_batch([9], [9], [nudge])                       # in place: the lazy refresh
            rec.append(bytes(gpu.pair_moments2(9, 8, 8.0, origin)))
            rec.append(gpu.download(9).tobytes())
            gpu.transform_batch([7], [0], [nudge])                       # a posed copy of a posed copy
            rec.append(bytes(gpu.pair_moments2(7, 1, 8.0, origin)))
            idx, d2 = gpu.nn(7, 1)
            rec += [idx.tobytes(), d2.tobytes()]
            outs.append(rec)
    finally:
        gpu.tune(posed_refresh=1)
    assert outs[0] == outs[1]


@pytest.mark.parametrize("case", ["ring12", "ring36", "slightly_sheared", "sheared", "two_runs", "uploads_between_runs"])
def test_pipelined_run_equals_pass_by_pass(mvr, case):
    """mvr_ring_run enqueues pass k+1's launch chain while pass k runs (behind a gate the host's solve opens; the poses reach
    the kernels through a device table, filled by the chain's posing launch on the way or by a launch of its own): the SAME
    poses and edge tables, bit for bit, as one mvr_ring_step per pass and as the run with the pipeline switched off -- on the 12- and 36-view rings, with a pose that is rigid only to 2e-4 (wider
    balls), with one that is not nearly rigid at all (those passes must not be queued ahead), over two calls (the second
    starts pipelined) and with a scan uploaded again between two calls (the queued chain would need a new ordering: the
    run must notice and enqueue that pass the ordinary way)."""
    V, N, max_d, K = (36, 3000, 8.0, 9) if case == "ring36" else (12, 12000, 4.0, 11)
    sp = mvr.synth_params(V, 3)
    scans = [mvr.synth_view(sp, v, N) for v in range(V)]
    piv, ax = mvr.synth_prior(sp)
    poses0 = [np.eye(4)] + [mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V)) for v in range(1, V)]
    if case == "slightly_sheared":
        poses0[5] = poses0[5].copy(); poses0[5][0, 1] += 2e-4; poses0[7] = poses0[7].copy(); poses0[7][:3, :3] *= 1.0002      # (e = 2e-4 and 6.9e-4: inside note_pose's 1e-3)
    if case == "sheared":
        poses0[5] = poses0[5].copy(); poses0[5][0, 1] += 3e-3
    origin = np.array(sp.pivot)
    edges = [(i, (i + 1) % V) for i in range(V)]
    posed, raw = list(range(V)), [V + v for v in range(V)]
    runs, piped = [], []
    for mode in ("piped", "off", "stepwise", "piped_prep_launch"):
        with mvr.Context(0) as ctx:
            ctx.tune(pipeline=int(mode != "off"))
            if mode == "piped_prep_launch":      # the device pose records filled by the launch made for that (what a chain that does not pose every view gets)
                ctx.tune(pose_prep_launch=1)
            for v in range(V):
                ctx.upload(V + v, scans[v])
            P, log = [p.copy() for p in poses0], []
            chunks = [K] if case not in ("two_runs", "uploads_between_runs") else [K - 4, 4]
            for ci, steps in enumerate(chunks):
                if ci == 1 and case == "uploads_between_runs":
                    ctx.upload(V + 3, scans[3])                         # the same points, a NEW point set: no ordering, no grid
                if mode == "stepwise":
                    for _ in range(steps):
                        P, info = ctx.ring_step(posed, raw, edges, P, max_d, origin)
                else:
                    P, info = ctx.ring_step(posed, raw, edges, P, max_d, origin, steps=steps)
                log.append((np.asarray(P).tobytes(), info["rows"].tobytes(), tuple(info["pair_n"])))
            # the posed clouds are what the LAST pass searched, and the slots' bookkeeping fits them: one more batch over them
            rows = ctx.pair_moments2_batch([(a, b) for a, b in edges], max_d, origin)
            log.append(b"".join(bytes(r) for r in rows))
            log.append(ctx.download(3).tobytes() + ctx.download(V - 1).tobytes())      # (the queued passes leave the posed points in original order out: written when the run ends)
            runs.append(log)
            piped.append(ctx.stat("piped_passes"))
    assert runs[0] == runs[1] == runs[2] == runs[3], case
    assert piped[1] == 0 and piped[2] == 0 and piped[3] == piped[0]
    if case == "sheared":
        assert piped[0] == 0, piped                 # a pose outside the nearly-rigid range: never queued ahead
    else:
        assert piped[0] >= K - 7, (case, piped)     # (the first passes build orderings and grids)
