"""Hand-built clamped responses that exercise the rules of the reference's component search
(find_chessboard_corners.cc:159-267, :284-397), with hand-derived expectations.  Shared by the CPU
tests of the oracle (tests/test_oracle.py) and the GPU tests that drive the HIP kernels with the very
same buffers through mrgingham_amd_cc_on_response_batch (tests/test_gpu_cc_rules.py)."""
import numpy as np

H, W = 48, 64


def flat_img(h=H, w=W, var=True):
    img = np.zeros((h, w), np.uint8)
    if var:
        img[:, ::2] = 255          # every 21x21 window has variance ~ 127^2 > 400
    return img


def _z():
    return np.zeros((H, W), np.int16)


def _pt(sx, sy, sw):
    """(x, y) * 1000 the way the reference rounds it (:262-263, :350-351)."""
    return [int(0.5 + sx / sw * 1000), int(0.5 + sy / sw * 1000)]


def detect_cases():
    """-> list of (name, response int16 [H,W], level image u8 [H,W], expected [[x1000, y1000], ...])"""
    img = flat_img()
    cases = []

    def add(name, d, expected, image=img):
        cases.append((name, d, image, expected))

    d = _z(); d[20, 20] = 500
    add("single pixel: N < 2 rejected (:205)", d, [])
    d = _z(); d[20, 20] = 300; d[20, 21] = 100
    add("two pixels: weighted centroid (:262-263), *1000 rounding (:350-351)", d,
        [[int(0.5 + (300 * 20 + 100 * 21) / 400 * 1000), 20000]])
    d = _z(); d[20, 20] = 120; d[20, 21] = 100
    add("peak 120 is not > 120 (:206)", d, [])
    d = _z(); d[20, 20] = 121; d[20, 21] = 100
    add("peak 121 passes (:206)", d, [[int(0.5 + (121 * 20 + 100 * 21) / 221 * 1000), 20000]])
    d = _z(); d[20, 20] = 300; d[20, 21] = 15; d[20, 22] = 300
    add("15 neither seeds nor extends (:169): two isolated single pixels", d, [])
    d = _z(); d[20, 20] = 300; d[20, 21] = 16; d[20, 22] = 300
    add("16 > 15 but not > 300>>4 = 18 (:27): consumed, not accumulated, not expanded", d, [])
    d = _z(); d[20, 20] = 300; d[20, 21] = 18; d[20, 22] = 300
    add("18 is not > 18 (:170, strict)", d, [])
    d = _z(); d[20, 20] = 300; d[20, 21] = 19; d[20, 22] = 300
    add("19 > 18 joins the two", d, [_pt(300 * 20 + 19 * 21 + 300 * 22, 20 * 619, 619)])
    d = _z(); d[20, 20] = 300; d[20, 21] = 100
    add("low-variance window rejects (:207, :50-88)", d, [], flat_img(var=False))
    d = _z(); d[20, 9] = 300; d[20, 10] = 100
    add("variance window leaves the image at x_peak = 9 (:52-57)", d, [])
    d = _z(); d[20, 10] = 300; d[20, 11] = 100
    add("variance window fits at x_peak = 10", d, [_pt(300 * 10 + 100 * 11, 20 * 400, 400)])
    d = _z(); d[20, W - 11] = 300; d[20, W - 12] = 100
    add("variance window fits at x_peak = w-11", d, [_pt(300 * (W - 11) + 100 * (W - 12), 20 * 400, 400)])
    d = _z(); d[20, W - 10] = 300; d[20, W - 11] = 100
    add("variance window leaves the image at x_peak = w-10", d, [])
    d = _z(); d[9, 30] = 300; d[10, 30] = 100
    add("variance window leaves the image at y_peak = 9", d, [])
    # Margin (:216-221): a pushed neighbour outside [7, w-7) x [7, h-7) sets touched_margin, i.e. any
    # ACCUMULATED pixel in column 7 / w-8 or row 7 / h-8 invalidates the component (which is still
    # consumed).  Peaks sit >= 10 px inside so that the variance window is not what rejects.
    blob = [50, 50, 50, 300, 50]
    d = _z(); d[20, 7:12] = blob
    add("blob reaching column 7: touched", d, [])
    d = _z(); d[20, 8:13] = blob
    add("the same blob one column to the right: accepted", d,
        [_pt(sum(v * x for v, x in zip(blob, range(8, 13))), 20 * 500, 500)])
    d = _z(); d[20, W - 12:W - 7] = blob[::-1]
    add("blob reaching column w-8: touched", d, [])
    d = _z(); d[20, W - 13:W - 8] = blob[::-1]
    add("the same blob one column to the left: accepted", d,
        [_pt(sum(v * x for v, x in zip(blob[::-1], range(W - 13, W - 8))), 20 * 500, 500)])
    d = _z(); d[7:12, 30] = blob
    add("blob reaching row 7: touched", d, [])
    d = _z(); d[8:13, 30] = blob
    add("the same blob one row down: accepted", d,
        [_pt(30 * 500, sum(v * y for v, y in zip(blob, range(8, 13))), 500)])
    d = _z(); d[H - 12:H - 7, 30] = blob[::-1]
    add("blob reaching row h-8: touched", d, [])
    d = _z(); d[H - 13:H - 8, 30] = blob[::-1]
    add("the same blob one row up: accepted", d,
        [_pt(30 * 500, sum(v * y for v, y in zip(blob[::-1], range(H - 13, H - 8))), 500)])
    # hot pixels outside the seed range [8, w-8) never seed (:332-333): a blob entirely in column 7
    d = _z(); d[20, 7] = 300; d[21, 7] = 300
    add("column 7 alone is never a seed", d, [])
    # ratio-of-max rule is order dependent (:27, :170).  Seed (x=20,y=20)=100; LIFO of +x,-x,+y,-y pops
    # -y,+y,-x,+x: (x=20,y=21)=100 [+y] is accumulated at max=100 before (x=21,y=20)=2000 [+x] raises the bar
    d = _z(); d[20, 20] = 100; d[20, 21] = 2000; d[21, 20] = 100
    add("running max: small pixel accumulated BEFORE the big one raises the bar", d,
        [_pt(100 * 20 + 100 * 20 + 2000 * 21, 100 * 20 + 100 * 21 + 2000 * 20, 2200)])
    # the big one pops first when it sits at +y of the seed: then the 100 at +x is below 2000>>4 = 125
    d = _z(); d[20, 20] = 130; d[21, 20] = 2000; d[20, 21] = 100
    add("running max: big pixel first, the 100 is zeroed without being accumulated", d,
        [[20000, int(0.5 + (130 * 20 + 2000 * 21) / 2130 * 1000)]])
    # first maximum wins (:176-181, strict >): two equal peaks, the variance test looks at the first one
    d = _z(); d[20, 9] = 300; d[20, 10] = 300
    add("equal peaks: the first (seed, x=9) stays the peak -> window leaves the image", d, [])
    # duplicate pushes: a 2x2 block pushes its last pixel twice; it is accumulated once (:243-250)
    d = _z(); d[20, 20] = 200; d[20, 21] = 200; d[21, 20] = 200; d[21, 21] = 200
    add("2x2 block: duplicate pushes dedup at pop", d, [[20500, 20500]])
    # a rejected first fill consumes pixels a later seed would have needed (raster order of seeds)
    d = _z(); d[20, 20] = 100; d[20, 21] = 1900; d[20, 22] = 100; d[21, 22] = 100
    add("seed order: chain 100-1900-100-100 from the left", d, None)
    # output order = raster order of the seeds (:332-353), not of the centroids
    d = _z(); d[30, 40] = 300; d[30, 41] = 300; d[12, 50] = 300; d[13, 50] = 300; d[12, 20] = 300; d[12, 21] = 200
    add("three blobs: output in seed raster order", d,
        [_pt(300 * 20 + 200 * 21, 12 * 500, 500), [50000, 12500], [40500, 30000]])
    return cases


def refine_cases():
    """-> list of (name, points f64 [N,2], levels i8 [N], response, image, level,
                   expected (points', levels', nrefined) or None)"""
    img = flat_img()
    d = _z(); d[20, 20] = 300; d[20, 21] = 100
    cx = (300 * 20 + 100 * 21) / 400
    cases = [
        ("only points tagged level+1 are touched (:362)", np.array([[20.2, 19.8], [100.0, 60.0]]),
         np.array([1, 2], np.int8), d, img, 0, (np.array([[cx, 20.0], [100.0, 60.0]]), [0, 2], 1)),
        ("3x3 seed neighbourhood (:371-382): 2 px away finds nothing", np.array([[23.0, 20.0]]),
         np.array([1], np.int8), d, img, 0, (np.array([[23.0, 20.0]]), [1], 0)),
        ("the response mutates between points (:358-396): the second identical point finds nothing",
         np.array([[20.0, 20.0], [20.0, 20.0]]), np.array([1, 1], np.int8), d, img, 0,
         (np.array([[cx, 20.0], [20.0, 20.0]]), [0, 1], 1)),
        ("level 1: point given in full-resolution coordinates (:367-372), result rescaled (:390)",
         np.array([[40.7, 40.3]]), np.array([2], np.int8), d, img, 1,
         (np.array([[(cx + 0.5) * 2 - 0.5, (20.0 + 0.5) * 2 - 0.5]]), [1], 1)),
    ]
    # two points whose 3x3 seeds reach the same blob from different sides: index order decides
    d2 = _z(); d2[20, 20:25] = [300, 200, 150, 200, 300]
    cases.append(("two points share one blob: the first takes it", np.array([[19.6, 20.0], [25.4, 20.0]]),
                  np.array([1, 1], np.int8), d2, img, 0, None))
    # a point on the image border: seeds outside the image are skipped (:159-163)
    cases.append(("point at the image corner", np.array([[0.0, 0.0], [63.0, 47.0]]), np.array([1, 1], np.int8),
                  d, img, 0, (np.array([[0.0, 0.0], [63.0, 47.0]]), [1, 1], 0)))
    return cases


def random_sparse_response(rng, h, w, nblobs, noise=0.0):
    """Blobs with values clustered around the thresholds (15/16, 120/121, max>>4), some touching the
    margin columns / rows, plus optional salt noise of hot single pixels."""
    d = np.zeros((h, w), np.int32)
    vals = np.array([14, 15, 16, 17, 30, 100, 119, 120, 121, 122, 200, 255, 256, 300, 1000, 1900, 2040])
    for _ in range(nblobs):
        kind = rng.randint(0, 6)
        if kind == 0:      # near the left / right margin
            cx = int(rng.choice([6, 7, 8, 9, 10, 11, w - 12, w - 11, w - 10, w - 9, w - 8, w - 7]))
            cy = rng.randint(0, h)
        elif kind == 1:    # near the top / bottom margin
            cx = rng.randint(0, w)
            cy = int(rng.choice([6, 7, 8, 9, 10, 11, h - 12, h - 11, h - 10, h - 9, h - 8, h - 7]))
        else:
            cx, cy = rng.randint(0, w), rng.randint(0, h)
        n = rng.randint(1, 14)
        x, y = cx, cy
        for _ in range(n):  # random walk blob
            if 0 <= x < w and 0 <= y < h:
                d[y, x] = int(rng.choice(vals)) if rng.rand() < 0.7 else rng.randint(1, 2041)
            if rng.rand() < 0.5:
                x += rng.randint(-1, 2)
            else:
                y += rng.randint(-1, 2)
    if noise > 0:
        m = rng.rand(h, w) < noise
        d[m] = rng.randint(-50, 400, size=int(m.sum()))
    return d.astype(np.int16)
