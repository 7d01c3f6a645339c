"""Frame sharding across GPUs and the one collective of the path.

Frames are independent (the reference already parallelises per image,
mrgingham-from-image.cc:50, :374-379), so each rank runs the whole path on its
own contiguous shard with no data-path communication.  The only exchange is the
gather of the (tiny) corner lists to rank 0: per-frame counts plus fixed-pitch
point blocks, one torch.distributed gather each (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).  Output order on rank 0 is
frame-major, and within a frame the reference's order.
"""
import torch
import torch.distributed as dist


def shard_range(nframes, rank, world):
    """Contiguous block [lo, hi) of frames owned by `rank`."""
    per, extra = divmod(nframes, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


def gather_corner_lists(points, levels, npoints, dst=0, group=None):
    """points f64 [B,P,2], levels int8 [B,P], npoints int32 [B] (same B,P on every
    rank) -> on `dst`: the three tensors concatenated over ranks along dim 0
    ([world*B, ...]); None elsewhere."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return points, levels, npoints
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    outs = []
    for t in (points, levels, npoints):
        t = t.contiguous()
        # int8 is not a NCCL reduction type but gather only moves bytes; view as uint8 for safety
        payload = t.view(torch.uint8) if t.dtype == torch.int8 else t
        bufs = [torch.empty_like(payload) for _ in range(world)] if rank == dst else None
        dist.gather(payload, bufs, dst=dst, group=group)
        if rank == dst:
            cat = torch.cat(bufs, dim=0)
            outs.append(cat.view(torch.int8) if t.dtype == torch.int8 else cat)
    return tuple(outs) if rank == dst else None


def packed_outputs(nframes, max_points, device):
    """One byte buffer per step that holds all three outputs of `Detector.chain`, so that the
    gather below is a single collective: -> (pack uint8 [nbytes], points f64 [B,P,2] view,
    levels int8 [B,P] view, npoints int32 [B] view)."""
    B, P = int(nframes), int(max_points)
    o_lv = B * P * 16
    o_np = (o_lv + B * P + 7) // 8 * 8
    pack = torch.zeros((o_np + 4 * B + 7) // 8 * 8, dtype=torch.uint8, device=device)   # (a multiple of 8: stacks of packs stay viewable as doubles)
    points = pack[:o_lv].view(torch.float64).view(B, P, 2)
    levels = pack[o_lv:o_lv + B * P].view(torch.int8).view(B, P)
    npoints = pack[o_np:o_np + 4 * B].view(torch.int32)
    return pack, points, levels, npoints


def unpack_outputs(pack, nframes, max_points):
    """Views into a buffer (or a [world, nbytes] stack of buffers) laid out by packed_outputs."""
    B, P = int(nframes), int(max_points)
    o_lv = B * P * 16
    o_np = (o_lv + B * P + 7) // 8 * 8
    lead = pack.shape[:-1]
    points = pack[..., :o_lv].contiguous().view(torch.float64).view(*lead, B, P, 2)
    levels = pack[..., o_lv:o_lv + B * P].contiguous().view(torch.int8).view(*lead, B, P)
    npoints = pack[..., o_np:o_np + 4 * B].contiguous().view(torch.int32).view(*lead, B)
    return points, levels, npoints


def gather_packed(pack, dst=0, group=None, out=None, force=False):
    """THE collective of the path: one gather of every rank's packed corner lists to `dst`
    (-> uint8 [world, nbytes] there, None elsewhere).  `out` may be a preallocated [world, nbytes]
    buffer on `dst`.  With one rank there is nothing to exchange and no collective is issued, unless
    `force` (used to run the collective's code path on a one-GPU box).

    The buffers are fixed-pitch (64 frames x 256 points x 17 B + counts = 0.28 MB per rank): the call
    needs no sizes on the host, so it is queued behind the step on the device and the pipeline never
    stalls.  `gather_exact` below moves only the live records but has to read the counts on the host
    first -- a round trip that costs more than the 0.2 MB it saves."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return pack.unsqueeze(0)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bufs = None
    if rank == dst:
        if out is None:
            out = torch.empty((world, pack.numel()), dtype=torch.uint8, device=pack.device)
        bufs = list(out.unbind(0))
    dist.gather(pack, bufs, dst=dst, group=group)
    return out if rank == dst else None


def _global_rank(group, group_rank):
    return group_rank if group is None else dist.get_global_rank(group, group_rank)


def gather_exact(points, levels, npoints, dst=0, group=None):
    """Exact-size form of the exchange (counts first, then grouped send / recv -- there is
    no gatherv): every rank compacts its live corners into records (frame, x, y, level), the record
    counts are all-gathered (one int64 per rank), then every rank but `dst` sends exactly its records
    and `dst` posts the matching receives in one batch (ncclGroupStart .. ncclSend / ncclRecv ..
    ncclGroupEnd under RCCL).  Returns on `dst` a list over ranks of (frame int32 [n], xy f64 [n,2],
    level int8 [n]) with frame numbered within the rank's shard, in frame-major then reference order;
    None elsewhere.  Synchronises the host with the device (the counts)."""
    B, P = levels.shape
    live = torch.arange(P, device=npoints.device).unsqueeze(0) < npoints.clamp(max=P).unsqueeze(1)   # [B,P]
    fidx = torch.arange(B, device=npoints.device, dtype=torch.float64).unsqueeze(1).expand(B, P)[live]
    rec = torch.stack([fidx, points[..., 0][live], points[..., 1][live], levels[live].to(torch.float64)], dim=1)
    rec = rec.contiguous()                                                                  # f64 [n, 4]

    def unpack(r):
        return r[:, 0].to(torch.int32), r[:, 1:3].contiguous(), r[:, 3].to(torch.int8)

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [unpack(rec)]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n_here = torch.tensor([rec.shape[0]], dtype=torch.int64, device=rec.device)
    counts = [torch.zeros_like(n_here) for _ in range(world)]
    dist.all_gather(counts, n_here, group=group)
    counts = [int(c.item()) for c in counts]
    if rank == dst:
        bufs = [rec if r == dst else torch.empty((counts[r], 4), dtype=torch.float64, device=rec.device)
                for r in range(world)]
        # (P2POp's peer is a GLOBAL rank: translate the group-local ones for a sub-group)
        ops = [dist.P2POp(dist.irecv, bufs[r], _global_rank(group, r), group) for r in range(world)
               if r != dst and counts[r] > 0]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return [unpack(b) for b in bufs]
    if counts[rank] > 0:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, rec, _global_rank(group, dst), group)]):
            w.wait()
    return None


# ---------------------------------------------------------------------------
# Mixed-resolution streams (BASELINE config 5)
# ---------------------------------------------------------------------------

def frame_cost(width, height, start_level=3):
    """Relative cost of one frame through the chain: ChESS pixels summed over the levels
    start_level .. 0 = W*H*(1 + 1/4 + ... ) (1.328*W*H for start_level 3)."""
    return width * height * sum(0.25 ** L for L in range(start_level + 1))


def lpt_assign(costs, world):
    """Longest-processing-time-first assignment of items to `world` ranks: returns one list of
    item indices per rank (greedy: heaviest remaining item to the currently lightest rank;
    ties broken by rank index, so every rank computes the same plan without communicating)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world
    plan = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        plan[r].append(i)
        loads[r] += costs[i]
    return plan


def plan_mixed_stream(sizes, world, rank, start_level=3):
    """sizes: list of (width, height) per frame of the stream.  Returns this rank's work as
    {(width, height): [frame indices]} -- the batch API takes equally-sized frames, so a rank
    runs one batch per distinct resolution it was assigned."""
    plan = lpt_assign([frame_cost(w, h, start_level) for (w, h) in sizes], world)[rank]
    groups = {}
    for i in sorted(plan):
        groups.setdefault(tuple(sizes[i]), []).append(i)
    return groups


# ---------------------------------------------------------------------------
# Dynamic balance: a shared work counter (SURVEY.md 8e's alternative to the static plan)
# ---------------------------------------------------------------------------

def stream_units(sizes, unit_frames=16, start_level=3):
    """The work units of a mixed-resolution stream: frames of one resolution in sub-batches of at most
    `unit_frames` (the batch API takes equally-sized frames), heaviest unit first by the cost MODEL
    1.328*W*H per frame.  Deterministic, so every rank builds the same list without communicating.
    -> list of ((width, height), [frame indices])."""
    by_size = {}
    for i, wh in enumerate(sizes):
        by_size.setdefault(tuple(wh), []).append(i)
    units = []
    for wh, idx in sorted(by_size.items()):
        for lo in range(0, len(idx), unit_frames):
            units.append((wh, idx[lo:lo + unit_frames]))
    units.sort(key=lambda u: (-frame_cost(*u[0], start_level) * len(u[1]), u[1][0]))
    return units


class WorkQueue:
    """Ranks pull work units off one shared counter until it runs out: the rank that finishes a unit early takes
    the next one, whatever the units really cost.  That matters here because the cost of a frame is NOT known in
    advance -- the reference's level search stops at the first pyramid level where a grid is found
    (mrgingham.cc:127-138), so a frame costs anything between the level-3 pass alone (1/64 of its pixels) and all
    four levels -- and a static plan on the model 1.328*W*H (lpt_assign) is wrong by up to that factor.

    The counter lives in the process group's key-value store (`store.add` is an atomic fetch-and-add served by
    rank 0's TCPStore: one small round trip per UNIT, not per frame), so it needs no GPU, no collective and no
    symmetric participation: a rank that never gets a unit never blocks the others.

    Every queue counts under a key of its OWN: the name plus the number of queues of that name this process has built
    so far.  Ranks build their queues in the same order (one per stream / pass, like every collective call), so they
    agree on the key without talking -- and a second queue never starts from the exhausted counter of the first
    (which would hand every rank None at once and skip the whole stream without an error)."""

    _generation = {}                         # name -> queues of that name built so far in this process

    def __init__(self, nunits, name="mrgingham_amd/wq", store=None):
        self.n = int(nunits)
        gen = WorkQueue._generation.get(name, 0)
        WorkQueue._generation[name] = gen + 1
        self.key = f"{name}#{gen}"
        if store is None and dist.is_available() and dist.is_initialized():
            store = dist.distributed_c10d._get_default_store()
        self.store = store
        self._local = 0                      # single process: a plain counter

    def next(self):
        """Index of the next unit, or None when the queue is empty."""
        if self.store is None:
            i = self._local
            self._local += 1
        else:
            i = self.store.add(self.key, 1) - 1
        return i if i < self.n else None

    def __iter__(self):
        while True:
            i = self.next()
            if i is None:
                return
            yield i
