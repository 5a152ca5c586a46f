"""Array ("level-synchronous") formulation of DistributeOctTree -- the algorithm the HIP kernel
`octree_select` implements -- written in plain Python so it can be checked against the literal
std::list restatement in oracle/orb_oracle.cpp on the CPU (reference src/ORBextractor.cc:539-763,
SURVEY.md Appendix C2).

Equivalences used (proved in DESIGN.md "octree"):
  * each step's candidates are ALL nodes holding >1 key;
  * a sweep splits them in list order; a "final phase" pass splits them in (count desc, list
    position asc) order -- list position asc == creation desc, the canonical F9 tie-break --
    and stops at the first split that brings |L| >= N;
  * new list = reverse(children in creation order) ++ (old list minus the split nodes);
  * a node's best key = max response, earliest candidate index on ties.
"""
import math
import numpy as np

f32 = np.float32


def _round_half_away(v):
    return int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)


def octree_twin(cands, minX, maxX, minY, maxY, N):
    cands = np.asarray(cands, np.int64).reshape(-1, 3)
    n = len(cands)
    W, H = maxX - minX, maxY - minY
    nIni = max(1, _round_half_away(float(f32(W) / f32(H))))
    hX = f32(W) / f32(nIni)
    # node arrays in list order: rect = (ULx, ULy, URx, BRy)
    rects = []
    for i in range(nIni):
        rects.append((int(hX * f32(i)), 0, int(hX * f32(i + 1)), H))
    key_node = np.zeros(n, np.int64)
    for k in range(n):
        key_node[k] = min(int(f32(cands[k, 0]) / hX), nIni - 1)
    cnt = np.bincount(key_node, minlength=nIni) if n else np.zeros(nIni, np.int64)
    # drop empty initial nodes (list order preserved)
    keep = [i for i in range(nIni) if cnt[i] > 0]
    remap = {o: p for p, o in enumerate(keep)}
    rects = [rects[i] for i in keep]
    count = [int(cnt[i]) for i in keep]
    key_node = np.array([remap[int(v)] for v in key_node], np.int64)
    final_phase = False
    while True:
        prev = len(rects)
        cand = [p for p in range(len(rects)) if count[p] > 1]
        # children occupancy of every candidate
        quad = np.full(n, -1, np.int64)
        ccount = {p: [0, 0, 0, 0] for p in cand}
        cset = set(cand)
        half = {}
        for p in cand:
            ULx, ULy, URx, BRy = rects[p]
            half[p] = (ULx + int(math.ceil(float(f32(URx - ULx) / f32(2)))), ULy + int(math.ceil(float(f32(BRy - ULy) / f32(2)))))
        for k in range(n):
            p = int(key_node[k])
            if p in cset:
                mx, my = half[p]
                x, y = cands[k, 0], cands[k, 1]
                q = (0 if x < mx else 1) + (0 if y < my else 2)      # n1=UL(0) n2=UR(1) n3=BL(2) n4=BR(3)
                quad[k] = q
                ccount[p][q] += 1
        if not final_phase:
            order = cand
            m = len(order)
        else:
            order = sorted(cand, key=lambda p: (-count[p], p))
            m = len(order)
            size = len(rects)
            for j, p in enumerate(order):
                size += sum(1 for c in ccount[p] if c > 0) - 1
                if size >= N:
                    m = j + 1
                    break
        split = order[:m]
        # children in creation order
        created = []          # (rect, count, parent, quadrant)
        for p in split:
            ULx, ULy, URx, BRy = rects[p]
            mx, my = half[p]
            crect = [(ULx, ULy, mx, my), (mx, ULy, URx, my), (ULx, my, mx, BRy), (mx, my, URx, BRy)]
            for q in range(4):
                if ccount[p][q] > 0:
                    created.append((crect[q], ccount[p][q], p, q))
        nToExpand = sum(1 for c in created if c[1] > 1)
        nc = len(created)
        split_set = set(split)
        newpos_child = {}
        for e, (_, _, p, q) in enumerate(created):
            newpos_child[(p, q)] = nc - 1 - e
        new_rects = [None] * nc
        new_count = [0] * nc
        for e, (r, c, p, q) in enumerate(created):
            new_rects[nc - 1 - e] = r
            new_count[nc - 1 - e] = c
        old_newpos = {}
        for p in range(len(rects)):
            if p in split_set:
                continue
            old_newpos[p] = len(new_rects)
            new_rects.append(rects[p]); new_count.append(count[p])
        for k in range(n):
            p = int(key_node[k])
            key_node[k] = newpos_child[(p, int(quad[k]))] if p in split_set else old_newpos[p]
        rects, count = new_rects, new_count
        if len(rects) >= N or len(rects) == prev:
            break
        if not final_phase and len(rects) + 3 * nToExpand > N:
            final_phase = True
    # best key per node
    best = [-1] * len(rects)
    for k in range(n):
        p = int(key_node[k])
        if best[p] < 0 or cands[k, 2] > cands[best[p], 2]:
            best[p] = k
    return cands[best].astype(np.int32).reshape(-1, 3) if best else np.zeros((0, 3), np.int32)
