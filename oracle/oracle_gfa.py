"""CPU restatement of the reference's hbv2gfa tool without line finding (src/modules/hbv2gfa.cc:50-99, src/GFADump.cc:228-286):
the graph statistics it prints and the <out_prefix>_raw.gfa it writes -- TEST INFRASTRUCTURE ONLY.  Pure Python loops: small graphs."""
from __future__ import annotations

import os
import subprocess

import numpy as np

from . import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
REF_GFA_BIN = os.path.join(HERE, "_ref", "ref_hbv2gfa")


def _adjacency(hbv):
    """to_left / to_right of every edge object and the per-vertex edge lists (digraphE::ToLeft/ToRight, Digraph.h)"""
    ne = hbv.n_edges
    to_left = np.full(ne, -1, np.int64); to_right = np.full(ne, -1, np.int64)
    out_e, in_e = [], []
    for v in range(hbv.n_vertices):
        a, b = int(hbv.from_off[v]), int(hbv.from_off[v + 1])
        out_e.append([int(x) for x in hbv.from_e[a:b]])
        for e in out_e[-1]:
            to_left[e] = v
        a, b = int(hbv.to_off[v]), int(hbv.to_off[v + 1])
        in_e.append([int(x) for x in hbv.to_e[a:b]])
        for e in in_e[-1]:
            to_right[e] = v
    return to_left, to_right, out_e, in_e


def involution(hbv):
    """HyperBasevector::Involution (paths/HyperBasevector.cc:648-660): the i-th edge object in sequence order pairs with the i-th in
    the order of the reverse complements"""
    codes, off = hbv.edge_codes()
    seq = [codes[int(off[e]):int(off[e + 1])] for e in range(hbv.n_edges)]
    x1 = sorted(range(len(seq)), key=lambda e: seq[e].tobytes())
    x2 = sorted(range(len(seq)), key=lambda e: (3 - seq[e][::-1]).astype(np.uint8).tobytes())
    inv = np.zeros(len(seq), np.int32)
    for i in range(len(seq)):
        inv[x1[i]] = x2[i]
    return inv


def stats_text(hbv, genome_size=0) -> str:
    """hbv2gfa.cc:57-92: what follows "=== Graph stats === " on stdout"""
    codes, off = hbv.edge_codes()
    sizes, canonical = [], 0
    for e in range(hbv.n_edges):
        s = codes[int(off[e]):int(off[e + 1])]
        if O.eform(s) != 1:
            canonical += len(s); sizes.append(len(s))
    sizes.sort(reverse=True)
    out = [f"Canonical graph sequences size: {canonical}"]
    k, cs = 0, 0
    for i in range(10, 100, 10):
        while cs * 100.0 / canonical < i:
            cs += sizes[k]; k += 1
        out.append(f"N{i}: {sizes[k - 1]}")
    if genome_size:
        k, cs = 0, 0
        out += ["", f"User provided size: {genome_size}"]
        for i in range(10, 100, 10):
            while cs * 100.0 / genome_size < i and k < len(sizes):
                cs += sizes[k]; k += 1
            out.append(f"NG{i}: n/a" if k == len(sizes) else f"NG{i}: {sizes[k - 1]}")
    return "\n".join(out) + "\n"


def raw_gfa(hbv) -> bytes:
    """GFADump.cc:228-286 with find_lines = false: S lines of the edge objects that are not REV-canonical (colour "black"), then per such
    object its links: followers (its own, and the inverses of its inverse's predecessors) in ascending id, each named by its canonical
    object, kept when that id is not smaller; then predecessors likewise; overlap always written as 0M"""
    codes, off = hbv.edge_codes()
    ne = hbv.n_edges
    form = [O.eform(codes[int(off[e]):int(off[e + 1])]) for e in range(ne)]
    inv = involution(hbv)
    to_left, to_right, out_e, in_e = _adjacency(hbv)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    out = []
    for e in range(ne):
        if form[e] == 1:
            continue
        out.append(b"S\tedge%d\t" % e + acgt[codes[int(off[e]):int(off[e + 1])]].tobytes() + b"\tCL:z:black\n")
    nxt = [out_e[to_right[e]] for e in range(ne)]
    prv = [in_e[to_left[e]] for e in range(ne)]
    for e in range(ne):
        if form[e] == 1:
            continue
        all_next = set(nxt[e]) | {int(inv[p]) for p in prv[int(inv[e])]}
        for n in sorted(all_next):
            cn = n if form[n] != 1 else int(inv[n])
            if cn < e:
                continue
            out.append(b"L\tedge%d\t+\tedge%d\t%s\t0M\n" % (e, cn, b"+" if cn == n else b"-"))
        all_prev = set(prv[e]) | {int(inv[n]) for n in nxt[int(inv[e])]}
        for p in sorted(all_prev):
            cp = p if form[p] != 1 else int(inv[p])
            if cp < e:
                continue
            out.append(b"L\tedge%d\t-\tedge%d\t%s\t0M\n" % (e, cp, b"-" if cp == p else b"+"))
    return b"".join(out)


def run_reference_gfa(workdir: str, in_prefix: str, out_prefix: str, genome_kb=0):
    """the real reference tool (oracle/_ref/ref_hbv2gfa) on workdir/<in_prefix>.hbv/.paths -> (stdout, bytes of <out_prefix>_raw.gfa)"""
    cmd = [REF_GFA_BIN, "-i", in_prefix, "-o", out_prefix]
    if genome_kb:
        cmd += ["-g", str(genome_kb)]
    txt = subprocess.run(cmd, check=True, capture_output=True, text=True, cwd=workdir).stdout
    return txt, open(os.path.join(workdir, out_prefix + "_raw.gfa"), "rb").read()
