"""Probe lengths of the LDS hash map of the component search (cc.hip, lds_hash / lds_find) on the hot lists of
synthetic boards: python tools/hash_probe.py   (CPU only; uses the oracle for the responses).
Prints, per board / level: entries, mean/max probes per hit and per miss of the four-neighbour lookups, for the
single-multiplier hash round 2 started with and for the per-coordinate one in use (both with one slot per probe),
and what a wave pays -- the longest probe sequence among 64 lanes, averaged over the lookups -- with one slot
per probe and with the buckets of two in use."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from mrgingham_amd import synth
M = 0xffffffff
def h_packed(e): return ((e * 0x9E3779B1) & M) >> 16
def h_xy(e): return (((e & 0xffff) * 0x9E3779B1 + (e >> 16) * 0x85EBCA77) & M) >> 20
def stats(es, hf, LHASH=4096):
    tab = -np.ones(LHASH, np.int64)
    for e in es:
        s = hf(int(e)) & (LHASH - 1)
        while tab[s] >= 0: s = (s + 1) & (LHASH - 1)
        tab[s] = e
    sset = set(int(e) for e in es)
    def probes(e):
        s = hf(e) & (LHASH - 1); n = 1
        while tab[s] >= 0 and tab[s] != e: s = (s + 1) & (LHASH - 1); n += 1
        return n
    hit, miss = [], []
    for e in es:
        for d in (1, -1, 65536, -65536):
            q = int(e) + d
            (hit if q in sset else miss).append(probes(q))
    return "hit %.2f/%d miss %.2f/%d" % (np.mean(hit), np.max(hit), np.mean(miss), np.max(miss))
def wave_max(es, bucketed):
    if bucketed:
        NB = 2048
        tab = -np.ones((NB, 2), np.int64)
        hb = lambda e: (((e & 0xffff) * 0x9E3779B1 + (e >> 16) * 0x85EBCA77) & M) >> 21
        for e in es:
            b = hb(int(e))
            while True:
                if tab[b, 0] < 0: tab[b, 0] = e; break
                if tab[b, 1] < 0: tab[b, 1] = e; break
                b = (b + 1) & (NB - 1)
        def probes(q):
            b = hb(q); n = 1
            while True:
                if tab[b, 0] == q or tab[b, 1] == q or tab[b, 0] < 0 or tab[b, 1] < 0: return n
                b = (b + 1) & (NB - 1); n += 1
    else:
        LH = 4096
        tab = -np.ones(LH, np.int64)
        for e in es:
            s = h_xy(int(e)) & (LH - 1)
            while tab[s] >= 0: s = (s + 1) & (LH - 1)
            tab[s] = e
        def probes(q):
            s = h_xy(q) & (LH - 1); n = 1
            while tab[s] >= 0 and tab[s] != q: s = (s + 1) & (LH - 1); n += 1
            return n
    tot = waves = 0
    for w0 in range(0, len(es), 64):
        for d in (1, -1, 65536, -65536):
            tot += max(probes(int(e) + d) for e in es[w0:w0 + 64])
        waves += 1
    return tot / waves / 4
for gridn, seed, (W, H) in ((14, 0, (4096, 3072)), (10, 1, (4096, 3072)), (10, 2, (1920, 1080)), (14, 3, (2560, 1920))):
    f = synth.board_frame(W, H, gridn, seed).numpy()
    for level in (0, 1, 2, 3):
        r = oracle.clamped_response(f, level)[0]
        ys, xs = np.nonzero(r > 15)
        es = ((ys.astype(np.int64) << 16) | xs)[:2040]
        sh = es[np.random.default_rng(0).permutation(len(es))]   # list order on the device is arbitrary
        print(f"{gridn}x{gridn} {W}x{H} level {level}: {len(es)} entries | packed: {stats(es, h_packed)} | per coordinate: {stats(es, h_xy)}"
              f" | longest of 64: one slot per probe {wave_max(sh, False):.2f}, buckets of two {wave_max(sh, True):.2f}")
