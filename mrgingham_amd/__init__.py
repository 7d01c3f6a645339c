"""mrgingham_amd: MI355X-native chessboard-corner candidate path of mrgingham.

Host-side mirror of the reference's Python module `mrgingham`
(mrgingham_pywrap.c:357-368) for the path this library covers, plus a batch
interface over torch tensors that already live in HBM.  Everything computes in
libmrgingham_amd.so (hand-written HIP, gfx950); PyTorch only supplies device
memory and streams.

    import mrgingham_amd as mrgingham
    r   = mrgingham.ChESS_response_5(image)            # int16[..., H, W]
    pts = mrgingham.find_points(image, image_pyramid_level=0)   # float64[N, 2]
    board = mrgingham.find_board(image, gridn=10)               # float64[100, 2] or None
"""
from .api import (ChESS_response_5, find_points, find_chessboard_corners, refine_points, find_board, find_chessboard,
                  find_grid_from_points, preprocess, read_image, Detector, level_dims)

__all__ = ["ChESS_response_5", "find_points", "find_chessboard_corners", "refine_points", "find_board",
           "find_chessboard", "find_grid_from_points", "preprocess", "read_image", "Detector", "level_dims"]
