"""Deterministic synthetic chessboard frames (integer-only, torch).

Inputs for tests/ and bench.py: the same bytes come out on the CPU here and on
the GPU box because every step is int64 arithmetic (no libm, no floats).

Board geometry follows the reference's own generator
(generate-chessboard-fig.py:103-134): gridn+3 cells per side, the outer ring
of cells merged into double-size squares, so exactly gridn x gridn interior
X-corners.  Rendering: cell = 0.8*H/(gridn+3) px,
centred, in-plane rotation 0.1 rad, dark/light 30/220 on background 200, 4x4
supersampling, per-pixel pseudo-noise (sigma ~2.3), then the 3x3 box blur the
reference CLI applies before the detector (mrgingham-from-image.cc:106-111,
(sum+4)/9, reflect-101 border).
"""
import torch

_COS_Q16 = 65209  # round(cos(0.1) * 2^16)
_SIN_Q16 = 6543   # round(sin(0.1) * 2^16)


def _mix(h):
    """32-bit integer hash on int64 tensors (xorshift-multiply)."""
    m = 0xFFFFFFFF
    h = h & m
    h = ((h ^ (h >> 16)) * 0x45D9F3B) & m
    h = ((h ^ (h >> 16)) * 0x45D9F3B) & m
    return h ^ (h >> 16)


def box_blur3(img):
    """(sum of 3x3 + 4) // 9 with reflect-101 borders; img int64 [H,W]."""
    p = torch.cat([img[1:2], img, img[-2:-1]], dim=0)
    p = torch.cat([p[:, 1:2], p, p[:, -2:-1]], dim=1)
    H, W = img.shape
    s = torch.zeros_like(img)
    for dy in range(3):
        for dx in range(3):
            s += p[dy:dy + H, dx:dx + W]
    return (s + 4) // 9


def board_frame(W, H, gridn=10, seed=0, noise=True, blur=True, device="cpu"):
    """One uint8 [H,W] frame.  `seed` moves the board by a sub-pixel offset and
    reseeds the noise, so a batch is `seed = frame index`."""
    dev = torch.device(device)
    ncell = gridn + 3
    cell = (8 * 65536 * 8 * H) // (10 * ncell)      # cell edge, units of 2^-16 * (1/8 px)
    half = (ncell * cell) // 2
    s = int(_mix(torch.tensor(seed * 7919 + 13, dtype=torch.int64)).item())
    ox, oy = s & 63, (s >> 6) & 63                      # offset, 1/8 px units (< 8 px)

    ys = torch.arange(H, dtype=torch.int64, device=dev).view(H, 1)
    xs = torch.arange(W, dtype=torch.int64, device=dev).view(1, W)
    acc = torch.zeros((H, W), dtype=torch.int64, device=dev)
    for j in range(4):
        dy = 8 * ys + (2 * j + 1) - 4 * H - oy
        for i in range(4):
            dx = 8 * xs + (2 * i + 1) - 4 * W - ox
            bu = _COS_Q16 * dx + _SIN_Q16 * dy + half
            bv = -_SIN_Q16 * dx + _COS_Q16 * dy + half
            cx = torch.div(bu, cell, rounding_mode="floor")
            cy = torch.div(bv, cell, rounding_mode="floor")
            inside = (cx >= 0) & (cx < ncell) & (cy >= 0) & (cy < ncell)
            par = (cx.clamp(1, ncell - 2) + cy.clamp(1, ncell - 2)) & 1
            val = torch.where(par == 1, 30, 220)
            acc += torch.where(inside, val, 200)
    img = (acc + 8) >> 4
    if noise:
        h = _mix((ys * W + xs) * 2654435761 + (seed + 1) * 40503)
        n = (h & 7) + ((h >> 3) & 7) + ((h >> 6) & 7) + ((h >> 9) & 7) - 14
        img = (img + (n >> 1)).clamp(0, 255)
    if blur:
        img = box_blur3(img)
    return img.to(torch.uint8)


def board_batch(B, W, H, gridn=10, seed0=0, noise=True, blur=True, device="cpu"):
    out = torch.empty((B, H, W), dtype=torch.uint8, device=device)
    for b in range(B):
        out[b] = board_frame(W, H, gridn, seed0 + b, noise, blur, device)
    return out


def noise_frame(W, H, seed=0, smooth=0, device="cpu"):
    """Pseudo-random uint8 frame (optionally box-blurred `smooth` times): the
    adversarial input for the connected-component rules."""
    dev = torch.device(device)
    ys = torch.arange(H, dtype=torch.int64, device=dev).view(H, 1)
    xs = torch.arange(W, dtype=torch.int64, device=dev).view(1, W)
    img = _mix((ys * W + xs) * 2246822519 + (seed + 1) * 3266489917) & 255
    for _ in range(smooth):
        img = box_blur3(img)
    return img.to(torch.uint8)


def cluttered_board_frame(W, H, gridn=10, seed=0, smooth=2, amp=128, device="cpu"):
    """board_frame over a textured background: smoothed pseudo-noise (box-blurred `smooth` times, contrast `amp` of
    255 around mid-grey) everywhere except a rectangle around the board (its bounding box + 40 px).  The texture
    gives the detector what a real calibration scene gives it -- tens of thousands of pixels with a ChESS
    response above 15 that belong to no corner (4096x3072, smooth 2, amp 128: ~8e4 at level 0, ~1e4 at level 1,
    against ~1.3e3 on the flat background) -- so the component search cannot run out of its LDS tables."""
    clean = board_frame(W, H, gridn, seed, device=device).to(torch.int64)
    bg = noise_frame(W, H, seed + 1000, smooth=smooth, device=device).to(torch.int64)
    bg = torch.div((bg - 128) * amp, 255, rounding_mode="floor") + 128
    lat = board_lattice(W, H, gridn, seed)                       # corner positions -> the board's extent
    pitch = float(((lat[0, 1] - lat[0, 0]) ** 2).sum() ** 0.5)
    pad = int(2.5 * pitch) + 40                                  # the outer ring of double cells + margin
    x0 = max(int(lat[..., 0].min()) - pad, 0); x1 = min(int(lat[..., 0].max()) + pad, W)
    y0 = max(int(lat[..., 1].min()) - pad, 0); y1 = min(int(lat[..., 1].max()) + pad, H)
    img = bg.clone()
    img[y0:y1, x0:x1] = clean[y0:y1, x0:x1]
    return img.clamp(0, 255).to(torch.uint8)


def cluttered_board_batch(B, W, H, gridn=10, seed0=0, device="cpu", **kw):
    out = torch.empty((B, H, W), dtype=torch.uint8, device=device)
    for b in range(B):
        out[b] = cluttered_board_frame(W, H, gridn, seed0 + b, device=device, **kw)
    return out


def board_lattice(W, H, gridn=10, seed=0):
    """Analytic positions of the gridn x gridn interior X-corners of board_frame(W, H, gridn, seed), from the
    renderer's own geometry (no detector involved): float64 numpy [gridn, gridn, 2] of (x, y) in the
    detector's convention (integer coordinates at pixel centres), rows in the order of the board's v axis
    (top to bottom for the +0.1 rad rotation), columns along u (left to right)."""
    import numpy as np
    ncell = gridn + 3
    cell = (8 * 65536 * 8 * H) // (10 * ncell)
    half = (ncell * cell) // 2
    s = int(_mix(torch.tensor(seed * 7919 + 13, dtype=torch.int64)).item())
    ox, oy = s & 63, (s >> 6) & 63
    out = np.empty((gridn, gridn, 2), dtype=np.float64)
    det = float(_COS_Q16) ** 2 + float(_SIN_Q16) ** 2
    for b in range(gridn):
        for a in range(gridn):
            bu, bv = (a + 2) * cell - half, (b + 2) * cell - half       # X-corners sit on cell boundaries 2 .. gridn+1
            dx = (_COS_Q16 * bu - _SIN_Q16 * bv) / det                   # 1/8-pixel units from the image centre
            dy = (_SIN_Q16 * bu + _COS_Q16 * bv) / det
            out[b, a, 0] = (dx + 4 * W + ox) / 8.0 - 0.5                 # continuous position -> pixel-centre convention
            out[b, a, 1] = (dy + 4 * H + oy) / 8.0 - 0.5
    return out


def dots_frame(W, H, gridn=10, seed=0, noise=True, device="cpu"):
    """A circle-grid target (what the reference's blob path is for): gridn x gridn dark discs on a light
    board, same placement, rotation and noise model as board_frame.  uint8 [H, W]."""
    dev = torch.device(device)
    ncell = gridn + 3
    cell = (8 * 65536 * 8 * H) // (10 * ncell)
    half = (ncell * cell) // 2
    s = int(_mix(torch.tensor(seed * 7919 + 13, dtype=torch.int64)).item())
    ox, oy = s & 63, (s >> 6) & 63
    ys = torch.arange(H, dtype=torch.int64, device=dev).view(H, 1)
    xs = torch.arange(W, dtype=torch.int64, device=dev).view(1, W)
    acc = torch.zeros((H, W), dtype=torch.int64, device=dev)
    r2 = (cell * 3 // 10) ** 2                                  # disc radius = 0.3 cell
    for j in range(4):
        dy = 8 * ys + (2 * j + 1) - 4 * H - oy
        for i in range(4):
            dx = 8 * xs + (2 * i + 1) - 4 * W - ox
            bu = _COS_Q16 * dx + _SIN_Q16 * dy + half
            bv = -_SIN_Q16 * dx + _COS_Q16 * dy + half
            inside = (bu >= 0) & (bu < ncell * cell) & (bv >= 0) & (bv < ncell * cell)
            # nearest lattice node a*cell, a in 2 .. gridn+1
            a = torch.div(bu + cell // 2, cell, rounding_mode="floor").clamp(2, gridn + 1)
            b = torch.div(bv + cell // 2, cell, rounding_mode="floor").clamp(2, gridn + 1)
            du, dv = bu - a * cell, bv - b * cell
            disc = (du * du + dv * dv) <= r2
            acc += torch.where(inside, torch.where(disc, 25, 215), 120)
    img = (acc + 8) >> 4
    if noise:
        hsh = _mix((ys * W + xs) * 2654435761 + (seed + 1) * 40503)
        n = (hsh & 7) + ((hsh >> 3) & 7) + ((hsh >> 6) & 7) + ((hsh >> 9) & 7) - 14
        img = (img + (n >> 1)).clamp(0, 255)
    return box_blur3(img).to(torch.uint8)
