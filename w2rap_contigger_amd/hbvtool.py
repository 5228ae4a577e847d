"""Canonicaliser and diff for <prefix>.hbv / <prefix>.paths pairs (parity tooling for users).

The reference numbers the edges of a graph in an order that depends on thread timing (BuildReadQGraph.cc:275-306, BigKPather.cc:275-292), so
two runs on the same reads write different files.  `canonicalise` renumbers a graph the way this package's canonical mode does: the
unipaths (edge objects that are not REV-canonical, dna/CanonicalForm.h:34-46) in lexicographic order of their sequences, each followed by
its reverse complement unless it is a palindrome (HBVFromEdges.cc:137-151); vertices keep their ids (they do not depend on the edge order);
the adjacency lists are rebuilt by digraphE::AddEdge's rule (sorted by target vertex, ties in insertion order, DigraphTemplate.h:1829-1839);
paths are rewritten through the map.  Two canonicalised graphs of the same reads are the same bytes; their paths may differ in the few reads
where an extension chose between parallel edges, which `diff` counts separately.

Host-side inspection tool (numpy); nothing here is on the GPU path.

    python -m w2rap_contigger_amd.hbvtool canon <in_prefix> <out_prefix>      # reads <in_prefix>.hbv/.paths, writes <out_prefix>.hbv/.paths
    python -m w2rap_contigger_amd.hbvtool diff  <a_prefix> <b_prefix>         # exit 0: same graph and paths up to numbering (ties reported)
"""
from __future__ import annotations

import sys

import numpy as np

from . import formats as F


def _form(s: np.ndarray) -> int:
    """bvec::getCanonicalForm: 0 FWD, 1 REV, 2 PALINDROME"""
    n = len(s)
    if n & 1:
        return 1 if (int(s[n // 2]) & 2) else 0
    r = 3 - s[::-1]
    d = np.nonzero(s != r)[0]
    if len(d) == 0:
        return 2
    return 0 if s[d[0]] < r[d[0]] else 1


def _seqs(h: F.HBV):
    codes, off = h.edge_codes()
    off = off.astype(np.int64)
    return [codes[off[i]:off[i + 1]] for i in range(h.n_edges)]


def _left_right(h: F.HBV):
    left = np.full(h.n_edges, -1, np.int64); right = np.full(h.n_edges, -1, np.int64)
    fo = h.from_off.astype(np.int64); to = h.to_off.astype(np.int64)
    for v in range(h.n_vertices):
        left[h.from_e[fo[v]:fo[v + 1]]] = v
        right[h.to_e[to[v]:to[v + 1]]] = v
    return left, right


def canonicalise(h: F.HBV, paths=None):
    """-> (canonical HBV, relabelled paths or None, new id of every old edge object)"""
    seqs = _seqs(h)
    by_seq = {s.tobytes(): i for i, s in enumerate(seqs)}
    if len(by_seq) != len(seqs):
        raise ValueError("two edge objects with the same sequence: not a unipath graph")
    forms = [_form(s) for s in seqs]
    fwd = sorted((i for i in range(len(seqs)) if forms[i] != 1), key=lambda i: seqs[i].tobytes())
    new_id = np.full(len(seqs), -1, np.int64)
    order = []
    for i in fwd:
        new_id[i] = len(order); order.append(i)
        if forms[i] != 2:
            rc = (3 - seqs[i][::-1]).astype(np.uint8).tobytes()
            if rc not in by_seq:
                raise ValueError(f"edge object {i} has no reverse complement in the graph")
            j = by_seq[rc]
            new_id[j] = len(order); order.append(j)
    if (new_id < 0).any():
        raise ValueError("an edge object is the reverse complement of no canonical object")
    left, right = _left_right(h)
    nv = h.n_vertices
    # AddEdge in new id order: from_[v] sorted by target vertex, ties after the existing entries; to_[w] likewise by source vertex
    ids = np.arange(len(order), dtype=np.int64)
    l_new, r_new = left[order], right[order]
    kf = np.lexsort((ids, r_new, l_new))                  # by (left, right, id)
    kt = np.lexsort((ids, l_new, r_new))                  # by (right, left, id)
    from_off = np.zeros(nv + 1, np.uint64); np.cumsum(np.bincount(l_new, minlength=nv), out=from_off[1:])
    to_off = np.zeros(nv + 1, np.uint64); np.cumsum(np.bincount(r_new, minlength=nv), out=to_off[1:])
    codes = np.concatenate([seqs[i] for i in order]) if order else np.zeros(0, np.uint8)
    off = np.zeros(len(order) + 1, np.uint64); np.cumsum([len(seqs[i]) for i in order], out=off[1:])
    pk, bo, ln = F.pack_bases(codes, off)
    out = F.HBV(h.K, from_off, r_new[kf].astype(np.int32), kf.astype(np.int32), to_off, kt.astype(np.int32), pk, bo, ln)
    new_paths = None
    if paths is not None:
        o, p, e = paths
        new_paths = (np.asarray(o, np.int32), np.asarray(p, np.uint64), new_id[np.asarray(e, np.int64)].astype(np.int32) if len(e) else np.zeros(0, np.int32))
    return out, new_paths, new_id


def diff(a_prefix: str, b_prefix: str, out=sys.stdout) -> int:
    """-> 0 when the two graphs and path sets are equal up to edge numbering (parallel-edge extension ties are reported, not counted as
    differences), 1 otherwise"""
    ha, hb = F.read_hbv(a_prefix + ".hbv"), F.read_hbv(b_prefix + ".hbv")
    pa, pb = F.read_paths(a_prefix + ".paths"), F.read_paths(b_prefix + ".paths")
    if ha.K != hb.K:
        print(f"K differs: {ha.K} vs {hb.K}", file=out); return 1
    sa, sb = {s.tobytes() for s in _seqs(ha)}, {s.tobytes() for s in _seqs(hb)}
    if sa != sb:
        print(f"edge sequence sets differ: {len(sa - sb)} only in A, {len(sb - sa)} only in B", file=out); return 1
    ca, qa, _ = canonicalise(ha, pa)
    cb, qb, _ = canonicalise(hb, pb)
    same_graph = F.hbv_to_bytes(ca, zero_padding=True) == F.hbv_to_bytes(cb, zero_padding=True)
    print(f"graphs: {ha.n_edges} edge objects, {ha.n_vertices} vertices: " + ("identical after canonicalisation" if same_graph else "DIFFERENT after canonicalisation"), file=out)
    if not same_graph:
        return 1
    if len(qa[0]) != len(qb[0]):
        print(f"path counts differ: {len(qa[0])} vs {len(qb[0])}", file=out); return 1
    left, right = _left_right(ca)
    pa_off, pb_off = qa[1].astype(np.int64), qb[1].astype(np.int64)
    ties = bad = 0
    for r in range(len(qa[0])):
        x, y = qa[2][pa_off[r]:pa_off[r + 1]], qb[2][pb_off[r]:pb_off[r + 1]]
        if qa[0][r] == qb[0][r] and len(x) == len(y) and np.array_equal(x, y):
            continue
        if len(x) == len(y) and all(u == v or (left[u] == left[v] and right[u] == right[v]) for u, v in zip(x, y)):
            ties += 1
        else:
            bad += 1
            if bad <= 10:
                print(f"read {r}: offset {qa[0][r]} path {list(x)}  vs  offset {qb[0][r]} path {list(y)}", file=out)
    print(f"paths: {len(qa[0])} reads, {ties} differ by a choice between parallel edges (extension tie-break), {bad} differ otherwise", file=out)
    return 1 if bad else 0


def main(argv=None) -> int:
    a = sys.argv[1:] if argv is None else argv
    if len(a) == 3 and a[0] == "canon":
        h, p, _ = canonicalise(F.read_hbv(a[1] + ".hbv"), F.read_paths(a[1] + ".paths"))
        F.write_hbv(a[2] + ".hbv", h); F.write_paths(a[2] + ".paths", *p)
        return 0
    if len(a) == 3 and a[0] == "diff":
        return diff(a[1], a[2])
    print(__doc__, file=sys.stderr)
    return 2


if __name__ == "__main__":
    sys.exit(main())
