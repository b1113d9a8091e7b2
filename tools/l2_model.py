"""A model of the L2 traffic of a planned mean-shift launch: 8 XCDs with
a private LRU L2 each (4 MiB, unit = one 24 KiB tile image), 32 workgroups per XCD that consume
one list entry per time step each.  Compares the committed flat schedule (the concatenated lists
cut into 256 equal ranges, an XCD gets 32 neighbouring ranges) with the phased schedule the
document proposes (the tile range in W windows; in every phase the entries of that window, of
the lists of the XCD, are cut into 32 equal ranges; all workgroups of the XCD enter a phase
together) and with whole lists per workgroup.  Lists are synthetic: every (shape, resident block)
keeps all tiles but a few contiguous runs (the locality order makes the skipped tiles runs),
`keep` of them on average.  Prints image bytes that miss the L2 and the bytes of partial sums
(one 128 KiB partial per list fragment, written once and read once) per launch.

    python tools/l2_model.py [keep=0.78] [images_per_tile=2] [windows=4]
"""
import sys
from collections import OrderedDict

import numpy as np

keep = float(sys.argv[1]) if len(sys.argv) > 1 else 0.78
nimg = int(sys.argv[2]) if len(sys.argv) > 2 else 2          # column pass: q and gu images
W = int(sys.argv[3]) if len(sys.argv) > 3 else 4
B, NT, NBLK, XCD, WG = 4, 314, 40, 8, 32                     # cfg5: 4 shapes, 314 tiles, 40 resident blocks of 256 rows
IMG = 24 * 1024 * nimg                                       # bytes streamed per list entry
L2 = 4 * 1024 * 1024 // IMG                                  # images (or image pairs) an L2 holds
PARTIAL = 256 * 128 * 4
rng = np.random.RandomState(0)


def make_list(k):
    on = np.ones(NT, bool)
    drop = int(round((1 - k) * NT))
    while drop > 0:
        run = min(drop, rng.randint(8, 60))
        s = rng.randint(0, NT - run)
        on[s:s + run] = False
        drop = int(round((1 - k) * NT)) - int((~on).sum())
    return np.nonzero(on)[0]


lists = []                                                   # (shape, tiles)
for b in range(B):
    kb = np.clip(keep + 0.1 * (b - 1.5) / 1.5, 0.3, 1.0)     # shapes of a batch differ (0.7 ... 0.95 measured)
    for r in range(NBLK):
        lists.append((b, make_list(kb)))
total = sum(len(t) for _, t in lists)


def simulate(per_wg):
    """per_wg[x][k] = list of (shape, tile) in the order workgroup k of XCD x consumes them;
    returns image bytes missed."""
    miss = 0
    for x in range(XCD):
        cache = OrderedDict()
        seqs = per_wg[x]
        for t in range(max(len(s) for s in seqs)):
            for s in seqs:
                if t < len(s):
                    key = s[t]
                    if key in cache:
                        cache.move_to_end(key)
                    else:
                        miss += 1
                        cache[key] = None
                        if len(cache) > L2:
                            cache.popitem(last=False)
    return miss * IMG


def flat():
    seq = [(b, int(t), li) for li, (b, tl) in enumerate(lists) for t in tl]
    c = -(-len(seq) // (XCD * WG))
    per, frags = [], 0
    for x in range(XCD):
        row = []
        for k in range(WG):
            part = seq[(x * WG + k) * c:(x * WG + k + 1) * c]
            frags += len({p[2] for p in part})
            row.append([(p[0], p[1]) for p in part])
        per.append(row)
    return per, frags


def xcd_lists():
    """lists -> XCDs in order, balanced by entries"""
    out, acc, x = [[] for _ in range(XCD)], 0, 0
    for li, (b, tl) in enumerate(lists):
        if acc >= (x + 1) * total / XCD and x < XCD - 1:
            x += 1
        out[x].append(li)
        acc += len(tl)
    return out


def phased():
    edges = [round(w * NT / W) for w in range(W + 1)]
    per, frags = [], 0
    for own in xcd_lists():
        row = [[] for _ in range(WG)]
        for w in range(W):
            seq = [(lists[li][0], int(t), li) for li in own for t in lists[li][1] if edges[w] <= t < edges[w + 1]]
            c = -(-len(seq) // WG)
            for k in range(WG):
                part = seq[k * c:(k + 1) * c]
                frags += len({p[2] for p in part})
                row[k] += [(p[0], p[1]) for p in part]
        per.append(row)
    return per, frags


def whole_lists():
    per = [[[] for _ in range(WG)] for _ in range(XCD)]
    for x, own in enumerate(xcd_lists()):
        for i, li in enumerate(own):
            per[x][i % WG] += [(lists[li][0], int(t)) for t in lists[li][1]]
    return per, len(lists)


alg = B * NT * IMG
print("cfg5 model: %d lists, %.2f of the tile pairs kept, %d entries, %d image(s) per entry; one pass over the "
      "images = %.0f MB; an L2 holds %d entries" % (len(lists), total / (len(lists) * NT), total, nimg, alg / 1e6, L2))
for name, fn in (("flat (committed)", flat), ("phased, %d windows" % W, phased), ("whole lists per workgroup", whole_lists)):
    per, frags = fn()
    m = simulate(per)
    steps = max(max(len(s) for s in row) for row in per)
    print("  %-28s images missing the L2 %7.0f MB (%4.1f x)   partial sums %5.0f MB (%4d fragments)   "
          "longest workgroup %4d entries (ideal %d)" % (name, m / 1e6, m / alg, 2 * frags * PARTIAL / 1e6, frags, steps,
                                                       -(-total // (XCD * WG))))
