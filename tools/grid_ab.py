"""Old against new grid finder on random candidate sets (host only): python tools/grid_ab.py old.so new.so [nsets]
Each library exports grid_find(xy, n, gridn, out, ring_seed, last_match) (a wrapper around mrg::find_grid_from_points).
Sets: projected lattices with sub-pixel noise, outliers, missing and duplicated points, near-ambiguous spacing, perfect
(cocircular) lattices, and perturbed visiting orders; boards and refusals must agree one for one."""
import ctypes, sys, time
import numpy as np

old, new = (ctypes.CDLL(p) for p in sys.argv[1:3])
nsets = int(sys.argv[3]) if len(sys.argv) > 3 else 16000
for L in (old, new):
    L.grid_find.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_int]
rng = np.random.RandomState(12345)
bad = found = 0
t_old = t_new = 0.0
for it in range(nsets):
    gridn = int(rng.choice([10, 10, 10, 14, 7]))
    kind = it % 8
    s = rng.uniform(20, 120)
    ang = rng.uniform(-0.6, 0.6)
    ii, jj = np.meshgrid(np.arange(gridn), np.arange(gridn), indexing="xy")
    P = np.stack([ii.ravel(), jj.ravel()], 1).astype(np.float64)
    if kind in (1, 5):   # perspective
        k = rng.uniform(0, 0.03)
        wgt = 1.0 + k * P[:, 0] + rng.uniform(0, 0.02) * P[:, 1]
        P = P / wgt[:, None]
    R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
    pts = (P * s) @ R.T + rng.uniform(200, 900, size=2)
    if kind != 7:
        pts = pts + rng.normal(0, rng.choice([0.0, 0.05, 0.3, 1.0]), size=pts.shape)
    extra = []
    if kind in (2, 3, 5, 6):   # outliers: random, and on the lattice just outside the board (the double-size frame corners)
        extra.append(rng.uniform(pts.min(0) - 3 * s, pts.max(0) + 3 * s, size=(int(rng.randint(1, 12)), 2)))
    if kind in (3, 6):
        m = int(rng.randint(1, 6))
        oi = rng.choice([-1, gridn], size=m)
        oj = rng.randint(-1, gridn + 1, size=m)
        extra.append((np.stack([oi, oj], 1) * s) @ R.T + pts[0] + rng.normal(0, 0.3, size=(m, 2)))
    if kind == 4 and rng.rand() < 0.5:   # a missing point
        pts = np.delete(pts, rng.randint(len(pts)), 0)
    if kind == 4 and rng.rand() < 0.5:   # a duplicated point
        pts = np.vstack([pts, pts[rng.randint(len(pts))][None]])
    allp = np.vstack([pts] + extra) if extra else pts
    allp = allp[rng.permutation(len(allp))]
    xy = np.ascontiguousarray(np.round(allp * 1000).astype(np.int32))
    seed, lastm = (0, 0) if it % 3 else (int(rng.randint(1, 1 << 30)), int(rng.randint(2)))
    oa, ob = np.zeros((gridn * gridn, 2)), np.zeros((gridn * gridn, 2))
    t0 = time.perf_counter()
    ra = old.grid_find(xy.ctypes.data, len(xy), gridn, oa.ctypes.data, seed, lastm)
    t1 = time.perf_counter()
    rb = new.grid_find(xy.ctypes.data, len(xy), gridn, ob.ctypes.data, seed, lastm)
    t2 = time.perf_counter()
    t_old += t1 - t0
    t_new += t2 - t1
    found += ra
    if ra != rb or (ra and not np.array_equal(oa, ob)):
        bad += 1
        if bad < 5:
            print("MISMATCH set", it, "kind", kind, "gridn", gridn, ra, rb)
print(f"{nsets} sets, {found} boards found by the old build, {bad} mismatches; old {t_old / nsets * 1e6:.1f} us / set, new {t_new / nsets * 1e6:.1f} us / set")
sys.exit(1 if bad else 0)
