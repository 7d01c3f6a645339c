"""`import mrgingham` -- the reference's Python module name (mrgingham_pywrap.c:357-368), served by mrgingham_amd.

A script written against the reference's wrapper runs unchanged with this repository on its path: the five
functions of the reference module (ChESS_response_5, find_points, find_board and the aliases
find_chessboard_corners, find_chessboard) with the same arguments, defaults, checks and return types, computed by
libmrgingham_amd.so on the GPU (there is no CPU path: with no HIP device they raise)."""
from mrgingham_amd import ChESS_response_5, find_points, find_board, find_chessboard_corners, find_chessboard

__all__ = ["ChESS_response_5", "find_points", "find_board", "find_chessboard_corners", "find_chessboard"]
