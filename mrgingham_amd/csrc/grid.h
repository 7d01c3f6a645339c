// Host-side grid finder (see grid.cpp).
#pragma once
#include <vector>

namespace mrg {

struct PointI { int x, y; };        // candidate, pixel coordinates * 1000 (point.hh:5-9)
struct PointD { double x, y; };     // corner, pixel coordinates (point.hh:11-15)

// mrgingham::find_grid_from_points (mrgingham.hh:83-87, find_grid.cc:1216-1445): appends the
// gridn*gridn corners in board order (rows top to bottom, each left to right) to `out`.
bool find_grid_from_points(std::vector<PointD>& out, const std::vector<PointI>& pts, int gridn);

// The reference's --debug-sequence trace (find_grid.cc:216, :247-306, :515-553): with `on`, find_grid_from_points
// reports on stderr, for the candidate nearest to (x, y) pixels, every neighbour it starts a sequence towards and
// every connection it considers / rejects / accepts along that sequence.  Thread-local: set around the call.
struct GridDebugSequence { bool on; int x, y; };
extern thread_local GridDebugSequence g_grid_debug_sequence;

// The reference's `debug` argument of find_grid_from_points (find_grid.cc:1223, :1229-1442): with it the finder
// writes its self-plotting vnlog dumps -- /tmp/mrgingham-2-voronoi.vnl (the neighbour graph), -3-candidates(.vnl,
// -detailed.vnl), -4-outer-edges(.vnl, -detailed.vnl), -5-outer-edge-cycles, -6-identified-outer-edge-cycle -- and says
// on stderr what it found or why it gave up.  Thread-local: set around the call.
extern thread_local bool g_grid_debug;

// Where the calling thread's find_grid_from_points calls spent their time (a handful of clock reads per call): the
// neighbour graph (sort + Delaunay sweep + site rings), the adjacency lists in the reference's visiting order, the
// sequence-candidate search, and everything after it (outer edges, 4-cycles, rows).  Thread-local; the find_boards
// calls add their workers' clocks up (mrgingham_amd_find_boards_stats).  The four phase fields count TICKS of the phase clock
// (the time-stamp counter on x86-64, nanoseconds of the steady clock elsewhere); grid_clock_tick_us() is what one is worth,
// applied where the totals are read out (no calibration loop on the path of a caller who never asks).
struct GridPhaseClock { double graph_t, adjacency_t, sequences_t, cycles_t; long calls, found; };
extern thread_local GridPhaseClock g_grid_clock;
double grid_clock_tick_us();

// visiting-order perturbations for the insensitivity tests (see grid.cpp); thread-local, default off
struct GridPerturbation { unsigned ring_seed; bool last_match; };
extern thread_local GridPerturbation g_grid_perturbation;

}  // namespace mrg
