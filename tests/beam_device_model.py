"""Sequential Python model of the DEVICE beam-search kernel's restructuring (csrc/beam.hip).
TEST INFRASTRUCTURE ONLY: it exists so the CPU suite can check, without a GPU, that the
kernel's phase structure is the same function as the host algorithm (decode_host.cpp /
oracle/decode.py), ties and evictions included.

What the kernel changes with respect to the host loop, and what is modelled here:

* node ids are allocated in blocks of C-1 when a prefix first expands, so ``order`` (the
  tie-break of TensorFlow's BeamComparer as the host code defines it) IS the node id;
* a frame's branches are the previous frame's beam in its sorted order (best first), so no
  sort of the branches; the beam is a sorted array (total descending, id ascending) whose last
  element is the heap bottom;
* only the state of the <= W branches of the frame is kept (ob, ol, ot, nb, nl, nt); nodes
  outside the beam keep only their child block; ``turn[node]`` maps a node in the beam to its
  branch index;
* a branch's turn evaluates all C-1 children at once: rejected children are decided
  immediately (the bottom only rises), the lowest-label candidate is inserted, children of
  lower label are thereby decided, higher ones are re-evaluated (an insertion can evict a
  sibling that was active);
* the turn loop stops at the first branch whose ORIGINAL old total does not beat the bottom of
  a full beam.
"""
import numpy as np

NEG = -np.inf


def _lse(a, b):
    if a == NEG:
        return b
    if b == NEG:
        return a
    m = max(a, b)
    return m + np.log(np.exp(a - m) + np.exp(b - m))


def beam_device_model(logits, W, merge_repeated=True, stats=None):
    x = np.asarray(logits, np.float32)
    T, C = x.shape
    K, blank = C - 1, C - 1
    children = {0: -1}                  # node -> child block (rec[].children)
    turn = {0: 0}                       # node -> branch index (rec[].turn), -1 / absent = not in beam
    block_parent = []
    # branch table
    b_node, b_par = [0], [-1]
    b_ob, b_ol, b_ot = [0.0], [NEG], [0.0]
    for t in range(T):
        inp = x[t].astype(np.float64) - np.float64(x[t].max())
        nb = len(b_node)
        b_ot0 = list(b_ot)
        b_nb, b_nl, b_nt = [NEG] * nb, [NEG] * nb, [NEG] * nb
        b_pt, b_nkids, b_evicted = [-1] * nb, [0] * nb, [False] * nb
        # phase B: every branch in parallel
        for q in range(nb):
            node, par = b_node[q], b_par[q]
            nl = b_ol[q]
            if par >= 0:
                label = (node - 1) % K
                pt = turn.get(par, -1)
                b_pt[q] = pt
                if pt >= 0:
                    plabel = -1 if par == 0 else (par - 1) % K
                    nl = _lse(nl, b_ob[pt] if label == plabel else b_ot[pt])
                    b_nkids[pt] += 1
                nl = nl + inp[label]
            b_nb[q] = b_ot[q] + inp[blank]
            b_nl[q] = nl
            b_nt[q] = _lse(b_nb[q], nl)
        # phase C: rank sort into the beam array
        hn = nb
        h = [None] * nb
        for q in range(nb):
            p = sum(1 for j in range(nb)
                    if b_nt[j] > b_nt[q] or (b_nt[j] == b_nt[q] and b_node[j] < b_node[q]))
            h[p] = [b_nt[q], b_node[q], b_par[q], q]
        # phase D: branch turns
        for r in range(nb):
            full = hn == W
            theta = h[W - 1][0] if full else NEG
            if full and not (b_ot0[r] > theta):
                break
            ot = b_ot[r]
            if not (ot > NEG and (not full or ot > theta)):
                continue
            if stats is not None:
                stats['turns'] = stats.get('turns', 0) + 1
            ob, node = b_ob[r], b_node[r]
            blk = children[node]
            if blk < 0:
                blk = len(block_parent)
                block_parent.append(node)
                children[node] = blk
                for c in range(K):
                    children[1 + blk * K + c] = -1
            base = 1 + blk * K
            blabel = -1 if node == 0 else (node - 1) % K
            kid = [-1] * K
            if b_nkids[r] > 0:
                for j in range(nb):
                    if b_pt[j] == r:
                        kid[(b_node[j] - 1) % K] = j
            v = []
            for c in range(K):
                prev = ob if c == blabel else ot
                v.append(NEG if prev == NEG else inp[c] + prev)
            undecided = set(range(K))
            while undecided:
                fullk = hn == W
                th = h[W - 1][0] if fullk else NEG
                cand = []
                for c in sorted(undecided):
                    active = kid[c] >= 0 and not b_evicted[kid[c]]
                    if active:
                        continue
                    if v[c] > NEG and (not fullk or v[c] > th):
                        cand.append(c)
                    else:
                        if kid[c] >= 0:
                            b_ob[kid[c]] = b_ol[kid[c]] = b_ot[kid[c]] = NEG
                        undecided.discard(c)
                if not cand:
                    break
                cs = cand[0]
                undecided = {c for c in undecided if c > cs}
                if fullk:
                    rb = h[W - 1][3]
                    if rb >= 0:
                        b_evicted[rb] = True
                    hn = W - 1
                    h.pop()
                vs, nid = v[cs], base + cs
                p = sum(1 for j in range(hn) if h[j][0] > vs or (h[j][0] == vs and h[j][1] < nid))
                h.insert(p, [vs, nid, node, -1])
                hn += 1
                if stats is not None:
                    stats['inserts'] = stats.get('inserts', 0) + 1
        # phase E: the beam becomes the next frame's branch table
        for q in range(nb):
            turn[b_node[q]] = -1
        nb2 = hn
        n_node, n_par, n_ob, n_ol, n_ot = [], [], [], [], []
        for j in range(nb2):
            nt, node, par, ref = h[j]
            n_node.append(node)
            n_par.append(par)
            n_ob.append(b_nb[ref] if ref >= 0 else NEG)
            n_ol.append(b_nl[ref] if ref >= 0 else nt)
            n_ot.append(nt)
            turn[node] = j
        b_node, b_par, b_ob, b_ol, b_ot = n_node, n_par, n_ob, n_ol, n_ot
    best, score = b_node[0], b_ot[0]
    out, prev, c = [], -1, best
    while c != 0:
        label = (c - 1) % K
        if not merge_repeated or label != prev:
            out.append(label)
        prev = label
        c = block_parent[(c - 1) // K]
    return out[::-1], float(score)
