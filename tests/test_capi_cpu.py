"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports
every symbol include/tdoa_mi355x.h declares, carries the reference's constants, refuses to run
without a GPU (no CPU fallback), and its host-side geodesy/solver agree with the oracle."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    import tdoa_amd
    tdoa_amd.build.build()
    return tdoa_amd.capi


def _header_functions():
    text = open(os.path.join(ROOT, "include", "tdoa_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tdoa_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(capi):
    L = capi.load()
    declared = _header_functions()
    assert len(declared) >= 30
    missing = [f for f in declared if not hasattr(L, f)]
    assert not missing, missing
    assert sorted(capi.SYMBOLS) == declared          # the ctypes binding covers the whole header
    assert L.tdoa_abi_version() == 4          # 4: tdoa_params grew lag_mode (round 3); 3: k1_smooth, k1_gate (round 2)


def test_default_params_are_the_reference_constants(capi):
    p = capi.default_params()
    assert p.sample_rate == 2e6            # processor.go:440
    assert p.max_lag == 20000              # processor.go:633
    assert p.corr_block == 1000            # processor.go:682
    assert p.weak_threshold == 0.001       # processor.go:476
    assert p.window_len == 2_000_000       # processor.go:772
    assert p.lag_mode == capi.LAGS_SIGNED and p.k1_smooth == 0 and p.k1_gate == 0 and p.reserved == 0
    import ctypes as C
    assert C.sizeof(capi.Params) == 56     # the C struct: 8 + 4 + 4 + 8 + 8 + 6 x 4, in the header's order


def test_no_cpu_fallback(capi):
    import tdoa_amd
    L = capi.load()
    if L.tdoa_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(tdoa_amd.TdoaError) as e:
        tdoa_amd.Context()
    assert e.value.status == 2             # TDOA_ERR_NO_DEVICE
    assert b"no CPU fallback" in L.tdoa_strerror(2)


def test_invalid_params_rejected(capi):
    import ctypes as C
    L = capi.load()
    p = capi.default_params()
    p.max_lag = 0
    h = C.c_void_p()
    assert L.tdoa_create(C.byref(p), C.byref(h)) == 1 and not h.value
    p = capi.default_params()
    p.lag_mode = 2                          # neither TDOA_LAGS_SIGNED nor TDOA_LAGS_GO
    assert L.tdoa_create(C.byref(p), C.byref(h)) == 1 and not h.value
    assert L.tdoa_create(None, None) == 1
    L.tdoa_destroy(None)                    # must be a no-op


def test_host_geodesy_and_solver_match_oracle(capi, oracle):
    for lle in oracle.STATIONS.values():
        assert np.allclose(capi.latlon_to_ecef(*lle), oracle.latlon_to_ecef(*lle), rtol=0, atol=1e-6)
        xyz = oracle.latlon_to_ecef(*lle)
        assert np.allclose(capi.ecef_to_latlon(*xyz), oracle.ecef_to_latlon(*xyz), rtol=0, atol=1e-9)
    st = [oracle.STATIONS[k] for k in oracle.COLLECTORS]
    for rd in ([0.0, 0.0, 0.0], [1500.0, -2500.0, 0.0], [-800.0, 300.0, 0.0]):
        rc, lle, it = capi.solve_3station(st, rd)
        orc, olle, oit = oracle.solve_tdoa(st, rd)
        assert rc == 0 and orc == 0 and it == oit
        assert np.allclose(lle, olle, rtol=0, atol=1e-7)
    # baselines of PROJECT_NOTES.md:25-27 through the product's own conversion
    e = [capi.latlon_to_ecef(*s) for s in st]
    d = lambda a, b: float(np.linalg.norm(e[a] - e[b])) / 1000
    assert (round(d(0, 1), 2), round(d(0, 2), 2), round(d(1, 2), 2)) == (12.29, 17.02, 10.02)


def test_sharding_covers_every_window_once():
    from tdoa_amd import sharding
    for world in (1, 2, 3, 8):
        for n in (1, 7, 99, 300):
            seen = sorted(w for r in range(world) for w in sharding.owned_windows(r, world, n))
            assert seen == list(range(n))
    with pytest.raises(ValueError):
        sharding.owned_windows(2, 2, 5)


def test_nstation_solver_reduces_to_reference_and_uses_all_pairs(capi, oracle):
    st = [oracle.STATIONS[k] for k in oracle.COLLECTORS]
    # n = 3 with weights {1,1,0}: the reference's own 2x2 system (processor.go:967-1003)
    for rd in ([0.0, 0.0, 0.0], [1500.0, -2500.0, 0.0]):
        rc, lle, it = capi.solve_nstation(st, rd, weights=[1, 1, 0])
        orc, olle, oit = oracle.solve_tdoa(st, rd)
        assert rc == 0 and it == oit and np.allclose(lle, olle, rtol=0, atol=1e-6)
    # 8 stations on a ring, transmitter inside, exact range differences, X/Y/Z unknowns
    rng = np.random.default_rng(4)
    c = np.mean(np.array(st), axis=0)
    ring = [(c[0] + 0.11 * np.cos(a), c[1] + 0.14 * np.sin(a), 300 + 100 * rng.random())
            for a in np.linspace(0, 2 * np.pi, 8, endpoint=False)]
    tx = (c[0] + 0.02, c[1] - 0.03, 420.0)
    e = [capi.latlon_to_ecef(*s) for s in ring]
    te = capi.latlon_to_ecef(*tx)
    r = [float(np.linalg.norm(x - te)) for x in e]
    rd = [r[j] - r[i] for i in range(8) for j in range(i + 1, 8)]
    rc, lle, it = capi.solve_nstation(ring, rd, solve_z=False)
    assert rc == 0 and abs(lle[0] - tx[0]) < 2e-3 and abs(lle[1] - tx[1]) < 2e-3
    assert capi.solve_nstation(st[:2], [0.0])[0] != 0       # needs >= 3 stations


def test_surface_solver_finds_a_ground_transmitter_where_the_frozen_z_plane_cannot(capi, oracle):
    """tdoa_solve_surface (position held on the ellipsoid): exact range differences of simulator.go's example transmitter
    (simulator.go:229) give it back to a metre for 3, 8 and 16 stations; whole-sample delays (one sample = 150 m of range,
    PROJECT_NOTES.md:29-32) stay inside 150 m; the reference's own solver (ECEF Z frozen at the centroid,
    processor.go:1004) is kilometres off on the same input -- which is why bench.py checks its fix with this one"""
    import math
    import bench
    for n, fs in ((3, 2e6), (8, 2e6), (16, 4e6)):
        st = bench.station_table(n)
        tx = capi.latlon_to_ecef(*bench.TX)
        dist = [math.dist(tx, capi.latlon_to_ecef(*s)) for s in st]
        h0 = sum(s[2] for s in st) / n
        pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]

        def err(lle):
            return math.dist(capi.latlon_to_ecef(float(lle[0]), float(lle[1]), bench.TX[2]), tx)
        rc, lle, it = capi.solve_surface(st, [dist[j] - dist[i] for i, j in pairs], None, h0)
        assert rc == 0 and it <= 6 and err(lle) < 1.0
        delay = [round(d / 299792458.0 * fs) for d in dist]
        assert delay == bench.propagation_delays(st, fs)
        rd = [(delay[j] - delay[i]) / fs * 299792458.0 for i, j in pairs]
        rc, lle, it = capi.solve_surface(st, rd, None, h0)
        assert rc == 0 and err(lle) < 150.0
        ref = capi.solve_3station(st, rd) if n == 3 else capi.solve_nstation(st, rd)
        assert ref[0] == 0 and err(ref[1]) > 1000.0
    assert capi.solve_surface(bench.station_table(3)[:2], [0.0])[0] != 0        # needs >= 3 stations
    assert capi.solve_surface(bench.station_table(3), [0.0, 0.0, 0.0], [1.0, -1.0, 1.0])[0] != 0      # negative weight


def test_unit_ownership_window_major_and_pair_major():
    from tdoa_amd import sharding
    # enough windows: a window's pairs all live on one rank
    for world, W, P in ((2, 9, 3), (8, 99, 28), (3, 3, 3)):
        for wid in range(W):
            assert {sharding.unit_owner(wid, p, world, W, P) for p in range(P)} == {wid % world}
    # fewer windows than ranks: units are dealt round-robin and every rank gets work
    world, W, P = 8, 3, 3
    owners = [sharding.unit_owner(w, p, world, W, P) for w in range(W) for p in range(P)]
    assert owners == [u % world for u in range(W * P)]
    assert set(owners) == set(range(8))



def test_nstation_solver_ignores_a_zero_weight_outlier(capi):
    """the plausibility gate hands a pair weight 0: a wildly wrong range difference on that pair must not move the fix"""
    rng = np.random.default_rng(12)
    ring = [(41.25 + 0.10 * np.cos(a), -96.0 + 0.13 * np.sin(a), 300.0 + 50 * rng.random())
            for a in np.linspace(0.3, 2 * np.pi + 0.3, 6, endpoint=False)]
    for _ in range(5):
        tx = (41.25 + 0.04 * (rng.random() - 0.5), -96.0 + 0.05 * (rng.random() - 0.5), 330.0)
        e = [capi.latlon_to_ecef(*s) for s in ring]
        te = capi.latlon_to_ecef(*tx)
        r = [float(np.linalg.norm(x - te)) for x in e]
        rd = [r[j] - r[i] for i in range(6) for j in range(i + 1, 6)]
        w = [1.0] * len(rd)
        rc, clean, _ = capi.solve_nstation(ring, rd, weights=w)
        assert rc == 0
        err = np.linalg.norm(capi.latlon_to_ecef(clean[0], clean[1], tx[2]) - te)
        assert err < 150.0                                     # Z is frozen at the centroid height, as in the reference
        bad = list(rd)
        bad[4] += 9000.0                                       # 30 us of error on one pair
        w[4] = 0.0
        rc, gated, _ = capi.solve_nstation(ring, bad, weights=w)
        assert rc == 0 and np.allclose(gated[:2], clean[:2], rtol=0, atol=2e-4)     # ~20 m: one pair fewer, same fix
        w[4] = 1.0
        rc, pulled, _ = capi.solve_nstation(ring, bad, weights=w)
        assert rc == 0 and np.linalg.norm(np.array(pulled[:2]) - np.array(clean[:2])) > 1e-3   # with weight 1 it would


def test_nstation_solver_rejects_empty_and_invalid_weight_sets(capi, oracle):
    """with every pair at weight 0 the iteration used to stop after 0 steps and report the station
    centroid as the fix; NaN weights produced a NaN position with status 0."""
    st = [oracle.STATIONS[k] for k in oracle.COLLECTORS]
    rd = [1500.0, -2500.0, 0.0]
    assert capi.solve_nstation(st, rd, weights=[0, 0, 0])[0] != 0                 # nothing to fit
    assert capi.solve_nstation(st, rd, weights=[1, 0, 0])[0] != 0                 # 1 pair < 2 unknowns
    assert capi.solve_nstation(st, rd, weights=[1, 1, 0], solve_z=True)[0] != 0   # 2 pairs < 3 unknowns
    assert capi.solve_nstation(st, rd, weights=[1, 1, 0])[0] == 0
    assert capi.solve_nstation(st, rd, weights=[1, -1, 1])[0] != 0                # negative weight
    assert capi.solve_nstation(st, rd, weights=[1, float("nan"), 1])[0] != 0
    assert capi.solve_nstation(st, rd, weights=[1, float("inf"), 1])[0] != 0
    assert capi.solve_nstation(st, [1500.0, float("nan"), 0.0], weights=[1, 1, 1])[0] != 0
    rc, lle, _ = capi.solve_nstation(st, [1500.0, float("nan"), 0.0], weights=[1, 0, 1])   # NaN on an unused pair is fine
    assert rc == 0 and np.isfinite(lle).all()


def test_cgo_shim_only_uses_declared_entry_points():
    """go/tdoa_cgo.go cannot be compiled here (no Go toolchain): at least every C.tdoa_* it calls must be declared in the
    header and exported by the library, and the INTEGRATION.md copy must be the same text"""
    import os
    import re
    from tdoa_amd import capi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "go", "tdoa_cgo.go")).read()
    hdr = open(os.path.join(root, "include", "tdoa_mi355x.h")).read()
    used = set(re.findall(r"C\.(tdoa_[a-z0-9_]+)\(", src))
    assert len(used) >= 8
    lib = capi.load()
    for name in used:
        assert re.search(r"\b%s\s*\(" % name, hdr), name
        assert hasattr(lib, name), name
    for t in set(re.findall(r"C\.(tdoa_[a-z_]+)\b(?!\()", src)) - used:     # types: tdoa_ctx, tdoa_params, tdoa_peak ...
        assert re.search(r"\b%s\b" % t, hdr), t
    body = src[src.index("package main"):]
    assert body in open(os.path.join(root, "INTEGRATION.md")).read()
    assert "len(b) < 2" in body                                      # the empty-slice guard


@pytest.mark.parametrize("n_stations", [2, 3, 4, 5, 8, 16, 24])
def test_segment_quads_cover_every_pair_once(capi, n_stations):
    """the quad cover of the segment form (two packed station transforms per segment serve up to four pairs): every
    pair exactly once, in the caller's orientation, in the slot its stations say; about P/2 quads for all pairs"""
    rng = np.random.default_rng(n_stations)
    all_pairs = [(i, j) for i in range(n_stations) for j in range(i + 1, n_stations)]
    subset = [p for p in all_pairs if rng.random() < 0.4] or all_pairs[:1]
    flipped = [(j, i) if rng.random() < 0.5 else (i, j) for i, j in all_pairs]      # any orientation is the caller's
    for pairs in (all_pairs, subset, flipped, all_pairs[:1]):
        quads = capi.segment_quads(n_stations, pairs)
        seen = []
        for a, b, c, d, *slots in quads:
            assert a >= 0 and c >= 0
            for (t, s), idx in zip(((a, c), (a, d), (b, c), (b, d)), slots):
                if idx >= 0:
                    assert tuple(pairs[idx]) == (t, s)
                    seen.append(idx)
            assert slots[0] >= 0 or slots[1] >= 0 or slots[2] >= 0 or slots[3] >= 0
            assert (b >= 0) == (slots[2] >= 0 or slots[3] >= 0) and (d >= 0) == (slots[1] >= 0 or slots[3] >= 0)
        assert sorted(seen) == list(range(len(pairs)))
        assert len(quads) <= max(1, (len(pairs) + 1) // 2 + n_stations // 2)
    if n_stations == 3:
        assert capi.segment_quads(3, all_pairs).tolist() == [[0, 1, 1, 2, 0, 1, -1, 2]]
    if n_stations >= 8:
        assert 2 * len(capi.segment_quads(n_stations, all_pairs)) <= 0.6 * len(all_pairs)
    with pytest.raises(ValueError):
        capi.segment_quads(3, [(0, 3)])
    with pytest.raises(ValueError):
        capi.segment_quads(3, [(1, 1)])


def test_the_documented_measurement_variant_still_builds(tmp_path):
    """fft_radix8.hpp / build.py advertise `build.py --variant dec14 TDOA_DEC_STEPS=14` (the 140 dB decimation filter of rounds
    2-3) for same-box A/B runs; the column walks are written for 8 or 12 steps per phase and are left out of such a build
    (TDOA_HAVE_DEC_COLS = 0: the library then runs the tile form and the full inverse) -- ADVICE r04: it no longer compiled"""
    import os
    from tdoa_amd import build
    out = build.build_variant("dec14_test", ["TDOA_DEC_STEPS=14"])
    try:
        assert os.path.getsize(out) > 100_000
    finally:
        os.remove(out)


@pytest.mark.parametrize("max_pairs", [16, 15, 14, 12, 5])
def test_staged_walk_groups_place_every_pair_once(capi, max_pairs):
    """the LDS-staged column walk's share-out of a window's pairs to workgroups (build_stg_groups, tdoa_mi355x.hip): for every
    station count 2..16 every pair is in exactly one group, no group carries more pairs than a workgroup has walks, a group's
    mask is exactly the stations its pairs touch, and above eight stations every group stays within eight -- the LDS ring then
    holds eight rows per phase for any number of stations"""
    for S in range(2, 17):
        pairs = [(i, j) for i in range(S) for j in range(i + 1, S)]
        groups = capi.staged_groups(S, max_pairs)
        seen = []
        for mask, members in groups:
            assert 1 <= len(members) <= max_pairs
            touched = 0
            for p in members:
                i, j = pairs[p]
                touched |= (1 << i) | (1 << j)
            assert touched == mask
            if S > 8:
                assert bin(mask).count("1") <= 8
            seen += members
        assert sorted(seen) == list(range(len(pairs)))
        need = -(-len(pairs) // max_pairs)
        assert need <= len(groups) <= need + 2 + (S > 8) * (len(pairs) // 40)
    # the timed geometries: 8 stations two equal runs of 14, 16 stations nine groups whose first three are the 15 pairs of six stations
    g8 = capi.staged_groups(8, 15)
    assert [len(m) for _, m in g8] == [14, 14] and g8[0][1] == list(range(14)) and [mask for mask, _ in g8] == [0xFF, 0xFC]
    g16 = capi.staged_groups(16, 15)
    assert len(g16) == 9 and [len(m) for _, m in g16[:3]] == [15, 15, 15] and all(bin(mask).count("1") == 6 for mask, _ in g16[:3])
    assert sum(bin(mask).count("1") for mask, _ in g16) <= 64                 # station-rows staged per row of the window (one group of all: 8 x 16 = 128)
    # the form without a loader wave: sixteen walks (four per SIMD) first, the remainder after them
    assert [len(m) for _, m in capi.staged_groups(8, 16)] == [16, 12]
    g16f = capi.staged_groups(16, 16)
    assert len(g16f) <= 9 and max(len(m) for _, m in g16f) == 16
    with pytest.raises(ValueError):
        capi.staged_groups(17)
    with pytest.raises(ValueError):
        capi.staged_groups(8, 17)
