"""Python face of the C-ABI.

Single-image functions mirror the reference module (names, arguments, error
behaviour: mrgingham_pywrap.c:40-112 ChESS_response_5, :128-212 find_points).
`Detector` is the batch interface over device-resident torch tensors.
"""
import ctypes
import os

import numpy as np

from . import _lib


def level_dims(width, height, level):
    """Size (w, h) of pyramid level `level` (find_chessboard_corners.cc:449-450)."""
    w, h = ctypes.c_int(), ctypes.c_int()
    if _lib.lib().mrgingham_amd_level_dims(width, height, level, ctypes.byref(w), ctypes.byref(h)) != 0:
        raise RuntimeError(f"Got an unreasonable image_pyramid_level = {level}")
    return w.value, h.value


def _require_device():
    """The C symbols can only report a missing device as "nothing found"; the Python face fails loudly."""
    if _lib.lib().mrgingham_amd_device_count() <= 0:
        raise RuntimeError("mrgingham_amd: no usable HIP device, and there is no CPU fallback")


def _check_image(image, exact_2d):
    image = np.asarray(image) if not isinstance(image, np.ndarray) else image
    # same checks, same messages as mrgingham_pywrap.c:53-68 / :163-178
    if exact_2d and image.ndim != 2:
        raise RuntimeError("The input image array must have exactly 2 dims (broadcasting not supported here); "
                           f"got {image.ndim}")
    if not exact_2d and image.ndim < 2:
        raise RuntimeError("The input image array must have at least 2 dims (extra ones will be broadcasted); "
                           f"got {image.ndim}")
    if image.dtype != np.uint8:
        raise RuntimeError("The input image array must contain 8-bit unsigned data")
    if image.shape[-1] > 1 and image.strides[-1] != 1:
        raise RuntimeError("Image rows must live in contiguous memory")
    return image


def ChESS_response_5(image):
    """int16 ChESS response, broadcasting over leading dims (mrgingham_pywrap.c:40-112).

    Like the reference, only the interior [7,W-7) x [7,H-7) of each slice is
    computed; the reference leaves the 7-pixel frame uninitialised, here it is 0.
    """
    image = _check_image(image, exact_2d=False)
    _require_device()
    L = _lib.lib()
    out = np.zeros(image.shape, dtype=np.int16)
    H, W = image.shape[-2:]
    lead = image.shape[:-2]
    for idx in np.ndindex(*lead):
        src = image[idx]
        dst = out[idx]
        stride = src.strides[0] if H > 1 else W
        L.mrgingham_ChESS_response_5(dst.ctypes.data, src.ctypes.data, W, H, stride)
    return out


def find_points(image, image_pyramid_level=0, blobs=False, debug=False):
    """Unordered corner candidates, float64 (N,2); (0,2) when none (mrgingham_pywrap.c:128-212)."""
    if blobs and image_pyramid_level != 0:
        raise RuntimeError("blob detector requires that image_pyramid_level == 0")
    image = _check_image(image, exact_2d=True)
    _require_device()
    result = []

    @_lib.ADD_POINTS_INT
    def add_points(xy, n, scale, cookie):  # add_points__find_points, mrgingham_pywrap.c:115-127
        a = np.ctypeslib.as_array(xy, shape=(2 * n,)).astype(np.float64)
        result.append((a * scale).reshape(n, 2))
        return True

    H, W = image.shape
    stride = image.strides[0] if H > 1 else W
    ok = _lib.lib().find_chessboard_corners_from_image_array_C(H, W, stride, image.ctypes.data,
                                                              int(image_pyramid_level), bool(blobs), bool(debug),
                                                              add_points, None)
    if not ok:
        if not result:
            return np.zeros((0, 2), dtype=np.float64)
        raise RuntimeError("find_chessboard_corners_from_image_array_C() failed")
    return result[0]


find_chessboard_corners = find_points  # compatibility alias, mrgingham_pywrap.c:365


def refine_points(points, levels, image, image_pyramid_level):
    """refine_chessboard_corners_from_image_array (find_chessboard_corners.hh:51-72):
    returns (points', levels', Nrefined); inputs are not modified."""
    image = _check_image(image, exact_2d=True)
    _require_device()
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 2).copy()
    lv = np.ascontiguousarray(levels, dtype=np.int8).copy()
    assert len(lv) == len(pts)
    H, W = image.shape
    stride = image.strides[0] if H > 1 else W
    n = _lib.lib().refine_chessboard_corners_from_image_array_C(H, W, stride, image.ctypes.data, pts.ctypes.data,
                                                                lv.ctypes.data, len(lv), int(image_pyramid_level),
                                                                False)
    return pts, lv, n


def read_image(filename, cli_scaling=False):
    """Decode a binary PGM or non-interlaced PNG to uint8 [H, W] with the library's own decoder (host only).
    16-bit files: the high byte (cv::imread(IMREAD_GRAYSCALE)) or, with cli_scaling, the CLI's
    convertTo(255/65535).  None when the file is unreadable, unsupported or malformed."""
    L = _lib.lib()
    w, h, d = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    name = os.fsencode(filename)
    if L.mrgingham_amd_read_image(name, int(bool(cli_scaling)), None, 0, ctypes.byref(w), ctypes.byref(h),
                                  ctypes.byref(d)) != 0:
        return None
    out = np.empty((h.value, w.value), dtype=np.uint8)
    if L.mrgingham_amd_read_image(name, int(bool(cli_scaling)), out.ctypes.data, out.size, ctypes.byref(w),
                                  ctypes.byref(h), ctypes.byref(d)) != 0:
        return None
    return out


def set_wait_policy(policy):
    """How this process's threads wait for the device: 0 runtime default, 1 spin, 2 yield, 3 block
    (mrgingham_amd_set_wait_policy; before the first context -- inside a PyTorch process the runtime is usually
    running already and refuses)."""
    return _lib.lib().mrgingham_amd_set_wait_policy(int(policy))


def device_for_thread(thread_index, ndevices, env_value=None):
    """The library's policy for the device of the k-th thread that calls a reference symbol (include/mrgingham_amd.h,
    "several GPUs"): MRGINGHAM_AMD_DEVICE if set, else k modulo the number of devices.  Host only."""
    env = None if env_value is None else str(env_value).encode()
    return _lib.lib().mrgingham_amd_device_for_thread(int(thread_index), int(ndevices), env)


def shard_range(total, k, n):
    """(first, count) of shard k of n over `total` frames (mrgingham_amd_shard_range).  Host only."""
    a, b = ctypes.c_int(), ctypes.c_int()
    if _lib.lib().mrgingham_amd_shard_range(int(total), int(k), int(n), ctypes.byref(a), ctypes.byref(b)) != 0:
        raise ValueError("bad shard arguments")
    return a.value, b.value


def set_thread_device(device):
    """The calling thread's single-image calls (ChESS_response_5, find_points, find_board ...) run on this device."""
    if _lib.lib().mrgingham_amd_set_thread_device(int(device)) != 0:
        raise ValueError(f"no such device: {device}")


def thread_device():
    """Device of the calling thread's context (created on first use: see device_for_thread); -1 without a device."""
    return _lib.lib().mrgingham_amd_thread_device()


class PinnedArray:
    """A numpy array over page-locked host memory (mrgingham_amd_host_alloc): a frame handed to find_points /
    find_board / ChESS_response_5 out of it is uploaded at the speed of the link, with no pinning on the fly.
    Keep the object alive as long as `.array` is in use."""

    def __init__(self, shape, dtype=np.uint8):
        self._L = _lib.lib()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = self._L.mrgingham_amd_host_alloc(max(n, 1))
        if not self._p:
            raise RuntimeError("mrgingham_amd_host_alloc failed (no device?)")
        buf = (ctypes.c_char * max(n, 1)).from_address(self._p)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        if getattr(self, "_p", None):
            self.array = None
            self._L.mrgingham_amd_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def chain_multi(detectors, shards, start_level=3, max_points=1024, sync=True):
    """mrgingham_amd_chain_multi: Detector k (its own device, or several on one) takes shards[k], a uint8 tensor
    [B_k, H, W] on that device; -> (points f64 [sum B, P, 2], levels int8 [sum B, P], npoints int32 [sum B]) on the
    FIRST detector's device, shard after shard -- one call, one gather."""
    import torch
    assert len(detectors) == len(shards) and len(detectors) > 0
    L = _lib.lib()
    root = detectors[0]
    frs = (_lib.Frames * len(shards))()
    total = 0
    for k, (d, fr) in enumerate(zip(detectors, shards)):
        assert fr.device == d.device, "every shard must live on its detector's device"
        f, B, H, W = d._frames(fr)
        frs[k] = f
        total += B
        torch.cuda.current_stream(fr.device).synchronize()
    P = int(max_points)
    pts = torch.empty((total, P, 2), dtype=torch.float64, device=root.device)
    lv = torch.empty((total, P), dtype=torch.int8, device=root.device)
    npts = torch.empty((total,), dtype=torch.int32, device=root.device)
    torch.cuda.current_stream(root.device).synchronize()
    ctxs = (ctypes.c_void_p * len(detectors))(*[d.ctx for d in detectors])
    for attempt in range(4):
        root._check(L.mrgingham_amd_chain_multi(ctxs, len(detectors), frs, int(start_level), pts.data_ptr(), lv.data_ptr(),
                                                npts.data_ptr(), P))
        if not sync:
            break
        rc = L.mrgingham_amd_sync_multi(ctxs, len(detectors))
        if rc == 0:
            break
        if rc != Detector.ERR_CAPACITY or attempt == 3:     # (the tables have grown: the same call again)
            root._check(rc)
    return pts, lv, npts


def find_grid_from_points(points_scaled, gridn=10):
    """mrgingham::find_grid_from_points (find_grid.cc:1216-1445), host only: int (N,2) candidates
    (pixel coordinates * 1000) -> float64 (gridn*gridn, 2) corners in board order, or None."""
    pts = np.ascontiguousarray(points_scaled, dtype=np.int32).reshape(-1, 2)
    out = np.empty((gridn * gridn, 2), dtype=np.float64)
    ok = _lib.lib().mrgingham_amd_find_grid_from_points(pts.ctypes.data, len(pts), int(gridn), out.ctypes.data)
    return out if ok else None


def find_grid_from_points_traced(points_scaled, gridn=10, debug_sequence=(-1, -1), debug=False):
    """find_grid_from_points with the reference's debug arguments: `debug` writes the /tmp/mrgingham-[2-6]-* vnlog
    dumps and reports progress on stderr; debug_sequence = (x, y) >= 0 traces the sequences from the candidate
    nearest to that pixel on stderr."""
    pts = np.ascontiguousarray(points_scaled, dtype=np.int32).reshape(-1, 2)
    out = np.empty((gridn * gridn, 2), dtype=np.float64)
    ok = _lib.lib().mrgingham_amd_find_grid_from_points_traced(pts.ctypes.data, len(pts), int(gridn), out.ctypes.data,
                                                              int(bool(debug)), int(debug_sequence[0]),
                                                              int(debug_sequence[1]))
    return out if ok else None


def find_grid_from_points_perturbed(points_scaled, gridn=10, ring_seed=0, last_match=False):
    """Test hook: find_grid_from_points with the neighbour-ring start of every site randomised (ring_seed != 0)
    and / or the last instead of the first matching neighbour taken along a sequence."""
    pts = np.ascontiguousarray(points_scaled, dtype=np.int32).reshape(-1, 2)
    out = np.empty((gridn * gridn, 2), dtype=np.float64)
    ok = _lib.lib().mrgingham_amd_find_grid_from_points_perturbed(pts.ctypes.data, len(pts), int(gridn),
                                                                 out.ctypes.data, int(ring_seed), int(bool(last_match)))
    return out if ok else None


def preprocess(image, clahe=True, blur_radius=1):
    """The CLI's preprocessing of an 8-bit image on the GPU (mrgingham-from-image.cc:71-111; the cv2
    recipe of find_board.docstring:8-10): normalize + CLAHE(8) when `clahe`, then a box blur of
    `blur_radius` (0 = none).  -> uint8 [H, W]."""
    _require_device()
    image = _check_image(image, exact_2d=True)
    H, W = image.shape
    out = np.empty((H, W), dtype=np.uint8)
    L = _lib.lib()
    L.mrgingham_amd_preprocess_image.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_void_p]
    rc = L.mrgingham_amd_preprocess_image(image.ctypes.data, W, H, image.strides[0], int(bool(clahe)), int(blur_radius),
                                          out.ctypes.data)
    if rc != 0:
        raise RuntimeError("mrgingham_amd: preprocessing failed (bad arguments or no device)")
    return out


def preprocess16(image16, clahe=True, blur_radius=1):
    """The CLI's preprocessing of a 16-bit image on the GPU (mrgingham-from-image.cc:85-111): normalize to
    0..65535 and CLAHE(8) on 16 bits when `clahe`, convertTo 8 bit with 255/65535, box blur.  -> uint8 [H, W]."""
    _require_device()
    image16 = np.ascontiguousarray(image16)
    if image16.dtype != np.uint16 or image16.ndim != 2:
        raise RuntimeError("preprocess16 takes a 2-D uint16 array")
    H, W = image16.shape
    out = np.empty((H, W), dtype=np.uint8)
    L = _lib.lib()
    L.mrgingham_amd_preprocess_image16.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_int, ctypes.c_void_p]
    if L.mrgingham_amd_preprocess_image16(image16.ctypes.data, W, H, W, int(bool(clahe)), int(blur_radius),
                                          out.ctypes.data) != 0:
        raise RuntimeError("mrgingham_amd: 16-bit preprocessing failed (bad arguments or no device)")
    return out


def find_board(image, image_pyramid_level=-1, gridn=10, blobs=False, debug=False, debug_sequence=None):
    """The full detector: float64 (gridn*gridn, 2) board corners, or None (mrgingham_pywrap.c:227-337).

    image_pyramid_level < 0 (default): try levels 3, 2, 1, 0 until a grid is found, then refine the
    corners down to level 0 (mrgingham.cc:116-139, :81-99)."""
    if blobs and image_pyramid_level != 0:
        raise RuntimeError("blob detector requires that image_pyramid_level == 0")
    dsx = dsy = -1
    if debug_sequence is not None:
        try:
            dsx, dsy = (int(t) for t in str(debug_sequence).split(","))
        except ValueError:
            raise RuntimeError("Couldn't parse debug_sequence as an 'INTEGER,INTEGER' string") from None
    image = _check_image(image, exact_2d=True)
    if gridn < 2:
        raise RuntimeError("gridn value must be >= 2")
    _require_device()
    result = []

    @_lib.ADD_POINTS_F64
    def add_points(xy, n, cookie):  # add_points__find_board, mrgingham_pywrap.c:214-226
        result.append(np.ctypeslib.as_array(xy, shape=(2 * n,)).copy().reshape(n, 2))
        return True

    H, W = image.shape
    stride = image.strides[0] if H > 1 else W
    ok = _lib.lib().find_chessboard_from_image_array_C(H, W, stride, image.ctypes.data, int(gridn),
                                                       int(image_pyramid_level), bool(blobs), bool(debug), dsx, dsy,
                                                       add_points, None)
    return result[0] if ok and result else None  # "possibly found no chessboard": None (:322-330)


find_chessboard = find_board  # compatibility alias, mrgingham_pywrap.c:366


class Detector:
    """Batch interface: frames are a uint8 torch tensor [B,H,W] already on the GPU.

    Wraps one mrgingham_amd_ctx (streams + scratch).  All results stay on the
    device until the caller moves them.
    """

    def __init__(self, device=None):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise RuntimeError("mrgingham_amd.Detector needs a HIP device; there is no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.L = _lib.lib()
        self.ctx = self.L.mrgingham_amd_create(self.device.index)
        if not self.ctx:
            raise RuntimeError("mrgingham_amd_create failed")
        self._options = {}
        self._fb_live = {}       # find_boards jobs in flight: ticket -> (boards, found, frames), see find_boards_submit

    def close(self):
        if getattr(self, "ctx", None):
            self.L.mrgingham_amd_destroy(self.ctx)   # (joins the host threads of the jobs in flight: nothing writes after this)
            self.ctx = None
        self._fb_live = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value):
        if self.L.mrgingham_amd_set_option(self.ctx, name.encode(), int(value)) != 0:
            raise ValueError(f"bad option {name}={value}")
        self._options[name] = int(value)

    def _check(self, rc):
        if rc != 0:
            e = RuntimeError(f"mrgingham_amd error {rc}: {self.L.mrgingham_amd_last_error(self.ctx).decode()}")
            e.code = rc
            raise e

    ERR_CAPACITY = -3

    def _sync_retrying(self, issue, retry, restore=None):
        """issue() + sync(); a frame that overflowed the component tables of its level makes the sync fail with
        ERR_CAPACITY *after the tables have grown to what it asked for* (include/mrgingham_amd.h,
        "hot_capacity_shift"), so the same call is simply made again -- like the reference-symbol wrappers of the
        library do it.  `restore` puts in/out arguments back first."""
        full = False
        try:
            for attempt in range(4):
                issue()
                try:
                    self.sync()
                    return
                except RuntimeError as e:
                    if not retry or getattr(e, "code", 0) != self.ERR_CAPACITY or attempt == 3:
                        raise
                    if restore:
                        restore()
                    if attempt == 2:      # the last try takes a table entry for every pixel, like the C wrappers
                        self.L.mrgingham_amd_set_option(self.ctx, b"hot_capacity_shift_temporary", 0)
                        full = True
        finally:
            if full:                      # (back to the caller's choice; what the tables have grown to is KEPT)
                self.L.mrgingham_amd_set_option(self.ctx, b"hot_capacity_shift_temporary", self._options.get("hot_capacity_shift", 7))

    def _frames(self, frames):
        t = self.torch
        assert frames.dtype == t.uint8 and frames.is_cuda and frames.dim() == 3 and frames.stride(2) == 1
        B, H, W = frames.shape
        fr = _lib.Frames(frames.data_ptr(), frames.stride(0) if B > 1 else H * frames.stride(1), B, W, H,
                         frames.stride(1) if H > 1 else W)
        return fr, B, H, W

    def sync(self):
        self._check(self.L.mrgingham_amd_sync(self.ctx))

    def stream_wait(self, stream=None):
        """Make a torch stream (default: the current one) wait for the last queued call, on the device."""
        t = self.torch
        st = t.cuda.current_stream(self.device) if stream is None else stream
        self._check(self.L.mrgingham_amd_stream_wait(self.ctx, st.cuda_stream))

    def after_stream(self, stream=None):
        """The next queued call starts after what is queued so far on a torch stream (default: the
        current one), e.g. the upload of its frames; on the device, the host is not blocked."""
        t = self.torch
        st = t.cuda.current_stream(self.device) if stream is None else stream
        self._check(self.L.mrgingham_amd_after_stream(self.ctx, st.cuda_stream))

    def chess_response(self, frames, level=0, clamp=False, out=None):
        """Dense int16 response [B,h,w] (border zero).  Runs on torch's current stream."""
        t = self.torch
        fr, B, H, W = self._frames(frames)
        w, h = level_dims(W, H, level)
        if out is None:
            out = t.empty((B, h, w), dtype=t.int16, device=frames.device)
        stream = t.cuda.current_stream(frames.device).cuda_stream
        self._check(self.L.mrgingham_amd_chess_response_batch(self.ctx, ctypes.byref(fr), level, int(clamp),
                                                              out.data_ptr(), stream))
        return out

    def decimate(self, frames, level):
        t = self.torch
        fr, B, H, W = self._frames(frames)
        w, h = level_dims(W, H, level)
        out = t.empty((B, h, w), dtype=t.uint8, device=frames.device)
        stream = t.cuda.current_stream(frames.device).cuda_stream
        self._check(self.L.mrgingham_amd_decimate_batch(self.ctx, ctypes.byref(fr), level, out.data_ptr(), stream))
        return out

    def box_blur(self, frames, radius=1):
        t = self.torch
        fr, B, H, W = self._frames(frames)
        out = t.empty((B, H, W), dtype=t.uint8, device=frames.device)
        stream = t.cuda.current_stream(frames.device).cuda_stream
        self._check(self.L.mrgingham_amd_box_blur_batch(self.ctx, ctypes.byref(fr), radius, out.data_ptr(), stream))
        return out

    def preprocess(self, frames, clahe=True, blur_radius=1):
        """The reference CLI's preprocessing (mrgingham-from-image.cc:71-111): normalize + CLAHE(8),
        then a box blur; on torch's current stream."""
        t = self.torch
        fr, B, H, W = self._frames(frames)
        out = t.empty((B, H, W), dtype=t.uint8, device=frames.device)
        stream = t.cuda.current_stream(frames.device).cuda_stream
        self._check(self.L.mrgingham_amd_preprocess_batch(self.ctx, ctypes.byref(fr), int(bool(clahe)),
                                                          int(blur_radius), out.data_ptr(), stream))
        return out

    def detect(self, frames, level, capacity=4096, sync=True, retry=True, out=None):
        """-> (xy int32 [B,capacity,2], counts int32 [B]) on the device.  sync=True waits for the result and, with
        retry, repeats the call when a frame overflowed the (self-growing) component tables.  `out` = (xy, counts) of an
        earlier call to write into (a pipelined caller rotates a few: no allocation, no stream synchronisation per call)."""
        t = self.torch
        fr, B, H, W = self._frames(frames)
        if out is None:
            xy = t.empty((B, capacity, 2), dtype=t.int32, device=frames.device)
            counts = t.empty((B,), dtype=t.int32, device=frames.device)
            t.cuda.current_stream(frames.device).synchronize()  # inputs/outputs ready before the ctx streams run
        else:
            xy, counts = out
            assert xy.dtype == t.int32 and counts.dtype == t.int32 and xy.is_contiguous() and xy.shape[0] == B and counts.shape[0] == B
            capacity = xy.shape[1]

        def issue():
            self._check(self.L.mrgingham_amd_detect_batch(self.ctx, ctypes.byref(fr), level, xy.data_ptr(), capacity,
                                                          counts.data_ptr()))
        if sync:
            self._sync_retrying(issue, retry)
        else:
            issue()
        return xy, counts

    def refine(self, frames, level, points, levels, npoints, sync=True, retry=True):
        """In-place refine of points f64 [B,P,2], levels int8 [B,P], npoints int32 [B]; -> nrefined int32 [B]."""
        t = self.torch
        fr, B, H, W = self._frames(frames)
        assert points.dtype == t.float64 and points.is_contiguous() and levels.dtype == t.int8
        P = points.shape[1]
        nref = t.empty((B,), dtype=t.int32, device=frames.device)
        keep = (points.clone(), levels.clone()) if (sync and retry) else None
        t.cuda.current_stream(frames.device).synchronize()

        def issue():
            self._check(self.L.mrgingham_amd_refine_batch(self.ctx, ctypes.byref(fr), level, points.data_ptr(),
                                                          levels.data_ptr(), npoints.data_ptr(), P, nref.data_ptr()))

        def restore():
            points.copy_(keep[0]); levels.copy_(keep[1])
            t.cuda.current_stream(frames.device).synchronize()
        if sync:
            self._sync_retrying(issue, retry, restore)
        else:
            issue()
        return nref

    def chain(self, frames, start_level=3, max_points=1024, out=None, sync=True, retry=True):
        """detect at start_level, refine down to 0 -> (points f64 [B,P,2], levels int8 [B,P], npoints int32 [B])."""
        t = self.torch
        fr, B, H, W = self._frames(frames)
        if out is None:
            out = (t.empty((B, max_points, 2), dtype=t.float64, device=frames.device),
                   t.empty((B, max_points), dtype=t.int8, device=frames.device),
                   t.empty((B,), dtype=t.int32, device=frames.device))
            t.cuda.current_stream(frames.device).synchronize()
        pts, lv, npts = out

        def issue():
            self._check(self.L.mrgingham_amd_chain_batch(self.ctx, ctypes.byref(fr), start_level, pts.data_ptr(),
                                                         lv.data_ptr(), npts.data_ptr(), pts.shape[1]))
        if sync:
            self._sync_retrying(issue, retry)    # (the chain writes all of its outputs: nothing to restore)
        else:
            issue()
        return pts, lv, npts

    def cc_detect_on_response(self, resp, level_images, level=0, capacity=4096, sync=True, retry=True):
        """The component search alone on caller-built responses (int16 [B,h,w], device) and level
        images (uint8 [B,h,w]): -> (xy int32 [B,capacity,2], counts int32 [B]).  For the rule tests."""
        t = self.torch
        assert resp.dtype == t.int16 and resp.is_cuda and resp.is_contiguous() and resp.dim() == 3
        assert level_images.dtype == t.uint8 and level_images.is_contiguous() and level_images.shape == resp.shape
        B, h, w = resp.shape
        xy = t.empty((B, capacity, 2), dtype=t.int32, device=resp.device)
        counts = t.empty((B,), dtype=t.int32, device=resp.device)
        t.cuda.current_stream(resp.device).synchronize()

        def issue():
            self._check(self.L.mrgingham_amd_cc_on_response_batch(self.ctx, resp.data_ptr(), level_images.data_ptr(), B,
                                                                  w, h, level, xy.data_ptr(), capacity,
                                                                  counts.data_ptr(), None, None, None, 0, None))
        if sync:
            self._sync_retrying(issue, retry)
        else:
            issue()
        return xy, counts

    def cc_refine_on_response(self, resp, level_images, level, points, levels, npoints, sync=True, retry=True):
        """In-place refinement of points f64 [B,P,2] / levels int8 [B,P] / npoints int32 [B] against
        caller-built responses; -> nrefined int32 [B]."""
        t = self.torch
        assert resp.dtype == t.int16 and resp.is_cuda and resp.is_contiguous() and resp.dim() == 3
        assert level_images.dtype == t.uint8 and level_images.is_contiguous() and level_images.shape == resp.shape
        assert points.dtype == t.float64 and points.is_contiguous() and levels.dtype == t.int8
        B, h, w = resp.shape
        nref = t.empty((B,), dtype=t.int32, device=resp.device)
        keep = (points.clone(), levels.clone()) if (sync and retry) else None   # (what a capacity retry restores)
        t.cuda.current_stream(resp.device).synchronize()

        def issue():
            self._check(self.L.mrgingham_amd_cc_on_response_batch(self.ctx, resp.data_ptr(), level_images.data_ptr(), B,
                                                                  w, h, level, None, 0, None, points.data_ptr(),
                                                                  levels.data_ptr(), npoints.data_ptr(),
                                                                  points.shape[1], nref.data_ptr()))

        def restore():
            points.copy_(keep[0]); levels.copy_(keep[1])
            t.cuda.current_stream(resp.device).synchronize()
        if sync:
            self._sync_retrying(issue, retry, restore if retry else None)
        else:
            issue()
        return nref

    def find_boards(self, frames, gridn=10, image_pyramid_level=-1, nthreads=0):
        """Full detector over a batch: -> (boards float64 [B, gridn*gridn, 2] (numpy, host),
        found_level int8 [B], -1 where no board was found).  Synchronous."""
        t = self.torch
        fr, B, H, W = self._frames(frames)
        boards = np.full((B, gridn * gridn, 2), np.nan, dtype=np.float64)
        found = np.full((B,), -1, dtype=np.int8)
        t.cuda.current_stream(frames.device).synchronize()
        self._check(self.L.mrgingham_amd_find_boards_batch(self.ctx, ctypes.byref(fr), int(gridn),
                                                           int(image_pyramid_level), boards.ctypes.data,
                                                           found.ctypes.data, int(nthreads)))
        return boards, found

    def find_boards_submit(self, frames, gridn=10, image_pyramid_level=-1, nthreads=0):
        """First half of find_boards: queues the device passes of the batch and returns a job to hand to
        find_boards_collect.  Submit the next batch(es) before collecting this one and the device passes, the host's
        grid finder and the refinement of consecutive batches overlap (include/mrgingham_amd.h)."""
        t = self.torch
        fr, B, H, W = self._frames(frames)
        boards = np.full((B, gridn * gridn, 2), np.nan, dtype=np.float64)
        found = np.full((B,), -1, dtype=np.int8)
        t.cuda.current_stream(frames.device).synchronize()
        ticket = self.L.mrgingham_amd_find_boards_submit(self.ctx, ctypes.byref(fr), int(gridn), int(image_pyramid_level),
                                                         boards.ctypes.data, found.ctypes.data, int(nthreads))
        if ticket < 0:
            self._check(ticket)
        # The library writes into `boards` / `found` (and reads `frames`) until the job is complete -- from inside ANY later
        # call on this context.  The Detector holds them until then, so a caller that drops the job tuple without
        # collecting it cannot make the library write into freed memory.
        self._fb_live[ticket] = (boards, found, frames)
        if len(self._fb_live) > 64:            # jobs nobody collected: complete what is in flight, then let the old ones go
            self.find_boards_stats(reset=False)   # (completes every batch in flight: their outputs are final)
            for tk in sorted(self._fb_live)[:-8]:
                del self._fb_live[tk]
        return (ticket, boards, found, frames)

    def find_boards_collect(self, job):
        """Second half: waits for the job -> (boards float64 [B, gridn*gridn, 2], found_level int8 [B])."""
        try:
            self._check(self.L.mrgingham_amd_find_boards_collect(self.ctx, job[0]))
        finally:
            self._fb_live.pop(job[0], None)
        return job[1], job[2]

    FB_STATS = ("batches", "host_threads", "ms_submit_checks", "ms_submit_prev_host_begin", "ms_submit_device_queued",
                "ms_grid_finder_joined", "ms_refinement_queued", "ms_collect_wait_refinement", "ms_collect_boards_copied",
                "grid_calls", "grid_found", "grid_us_graph", "grid_us_adjacency", "grid_us_sequences", "grid_us_cycles_rows",
                "device_ms_first_pass", "device_ms_refinement")

    def find_boards_stats(self, reset=True):
        """mrgingham_amd_find_boards_stats as a dict (totals since the last reset; completes the batches in flight)."""
        out = np.zeros(len(self.FB_STATS), dtype=np.float64)
        n = self.L.mrgingham_amd_find_boards_stats(self.ctx, out.ctypes.data, len(out), int(bool(reset)))
        if n < 0:
            self._check(n)
        return dict(zip(self.FB_STATS, out.tolist()))

    def sparse_fallbacks(self):
        """Frames the sparse refinement (option "sparse_refine") handed back to the dense kernels since the last
        call of this method; the library repeats them inside the call that met them.  Synchronises."""
        n = self.L.mrgingham_amd_sparse_fallbacks(self.ctx)
        if n < 0:
            self._check(n)
        return n

    def debug_paths(self, level, nframes):
        """Test hook: per frame of the most recent call at `level`, 1 = component search out of LDS, 0 = the
        global-memory kernels."""
        out = np.zeros((nframes,), dtype=np.int32)
        self._check(self.L.mrgingham_amd_debug_paths(self.ctx, int(level), int(nframes), out.ctypes.data))
        return out

    def debug_refine_clock(self):
        """Phase clock of the most recent refinement (option cc_lds = 1 | 512): 12 int64, see the header."""
        out = np.zeros(12, dtype=np.int64)
        self._check(self.L.mrgingham_amd_debug_refine_clock(self.ctx, out.ctypes.data))
        return out

    def chain_info(self):
        """(fused_pyramid, merged_levels) of the most recent chain() call: whether the level-0 response kernel
        also wrote the level images, and how many levels shared one response launch."""
        f, m = ctypes.c_int(0), ctypes.c_int(0)
        self._check(self.L.mrgingham_amd_chain_info(self.ctx, ctypes.byref(f), ctypes.byref(m)))
        return bool(f.value), int(m.value)

    def scratch_bytes(self):
        """Device memory the context holds right now."""
        return int(self.L.mrgingham_amd_scratch_bytes(self.ctx))

    def set_kernel_timing(self, enable):
        """True / 1: hipEvents around the level-0 response launches + the engine-clock probe; 2: the probe alone; False: off."""
        self.L.mrgingham_amd_set_kernel_timing(self.ctx, int(enable))

    def sclk_mhz(self):
        """Engine clock (MHz) the level-0 response launches since the last call ran at (kernel timing on); 0.0 = none probed."""
        return float(self.L.mrgingham_amd_sclk_mhz(self.ctx))

    def chess_kernel_ms(self):
        n = ctypes.c_int()
        ms = self.L.mrgingham_amd_chess_kernel_ms(self.ctx, ctypes.byref(n))
        return ms, n.value
