"""On-disk formats either side of Step 2 (little-endian, numpy only).

Step-1 -> Step-2 inputs (reference writer: feudal/FeudalFileWriter.cc:83-95,
feudal/FeudalControlBlock.h:157-166):

* ``frag_reads_orig.fastb``  Feudal file of 2-bit reads.  Header 24 B
  ``{u32 n; u8 flags=1; u8 sizeofFixed=4; u8 sizeofX=16; u8 sizeofA=1;
  u64 offTableOff; u64 fixedOff}``, variable data = ceil(len/4) bytes per read
  with base i at bits 2*(i%4) of byte i/4 (feudal/FieldVec.h:768), (n+1)
  absolute u64 file offsets, n u32 lengths (FieldVec.h:585-587).
* ``frag_reads_orig.qualp``  same container, sizeofFixed=0, sizeofX=8; each
  element a PQVec byte string (feudal/PQVec.cc:87-127): blocks
  ``{u8 nQs; 3 bit nBits; 6 bit minQ; nQs*nBits bits of q-minQ}`` padded to a
  byte, terminated by a 0 byte.

Step-2 -> Step-3 outputs:

* ``<prefix>.small_K.hbv``  ``"BINWRITE"``, i32 K, from_, from_edge_obj_,
  to_edge_obj_ (each ``u64 n; n x {u64 deg; i32[deg]}``), edges_ ``u64 E;
  E x {u32 nbases; u8[ceil(nbases/4)]}`` (paths/HyperBasevector.cc:121-125,
  graph/DigraphTemplate.h:2226-2231, graph/Digraph.h:350-351,
  feudal/FieldVec.h:595-597).
* ``<prefix>.small_K.paths``  ``u64 n; n x {i32 offset; u16 len; i32[len]}``
  (paths/long/ReadPath.cc:6-20).
* ``small_K.freqs``  text, lines ``"i, hist[i]\\n"`` for i=1..100
  (paths/long/BuildReadQGraph.cc:1108-1112).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass

import numpy as np

FEUDAL_HDR = struct.Struct("<IBBBBQQ")
assert FEUDAL_HDR.size == 24


# --------------------------------------------------------------------------- bases
def pack_bases(codes: np.ndarray, off: np.ndarray):
    """codes: u8 base codes concatenated; off: u64[n+1] -> (packed u8, byte_off u64[n+1], len u32[n]).

    Each read starts on a byte boundary (the .fastb variable-data layout)."""
    off = np.asarray(off, dtype=np.uint64)
    lens = np.diff(off).astype(np.uint32)
    nbytes = (lens.astype(np.uint64) + 3) // 4
    byte_off = np.zeros(len(lens) + 1, dtype=np.uint64)
    np.cumsum(nbytes, out=byte_off[1:])
    total = int(byte_off[-1])
    if total == 0:
        return np.zeros(0, np.uint8), byte_off, lens
    # position of every base inside the padded base stream
    read_id = np.repeat(np.arange(len(lens), dtype=np.int64), lens.astype(np.int64))
    within = np.arange(len(codes), dtype=np.int64) - off[:-1].astype(np.int64)[read_id]
    slot = byte_off[:-1].astype(np.int64)[read_id] * 4 + within
    padded = np.zeros(total * 4, dtype=np.uint8)
    padded[slot] = codes
    p = padded.reshape(-1, 4)
    packed = (p[:, 0] | (p[:, 1] << 2) | (p[:, 2] << 4) | (p[:, 3] << 6)).astype(np.uint8)
    return packed, byte_off, lens


def unpack_bases(packed: np.ndarray, byte_off: np.ndarray, lens: np.ndarray):
    """inverse of pack_bases -> (codes u8, off u64[n+1])"""
    lens = np.asarray(lens, dtype=np.int64)
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    if len(packed) == 0:
        return np.zeros(0, np.uint8), off
    allb = np.empty((len(packed), 4), dtype=np.uint8)
    for j in range(4):
        allb[:, j] = (packed >> (2 * j)) & 3
    allb = allb.reshape(-1)
    read_id = np.repeat(np.arange(len(lens), dtype=np.int64), lens)
    within = np.arange(int(off[-1]), dtype=np.int64) - off[:-1].astype(np.int64)[read_id]
    slot = np.asarray(byte_off, dtype=np.int64)[:-1][read_id] * 4 + within
    return allb[slot], off


# --------------------------------------------------------------------------- feudal container
def _write_feudal(path, var: bytes, elem_off: np.ndarray, fixed: bytes, sizeof_fixed: int, sizeof_x: int, sizeof_a: int):
    n = len(elem_off) - 1
    var_off = 24 + len(var)
    fixed_off = var_off + (n + 1) * 8
    with open(path, "wb") as f:
        f.write(FEUDAL_HDR.pack(n & 0xFFFFFFFF, 1, sizeof_fixed, sizeof_x, sizeof_a, var_off, fixed_off))
        f.write(var)
        f.write((np.asarray(elem_off, dtype=np.uint64) + np.uint64(24)).astype("<u8").tobytes())
        f.write(fixed)


def _read_feudal(path):
    buf = np.fromfile(path, dtype=np.uint8)
    n32, flags, szf, szx, sza, var_off, fixed_off = FEUDAL_HDR.unpack(buf[:24].tobytes())
    if (flags & 3) != 1:
        raise ValueError(f"{path}: not a single-file feudal file")
    n = (fixed_off - var_off) // 8 - 1
    if (n & 0xFFFFFFFF) != n32:
        raise ValueError(f"{path}: element count mismatch")
    offs = buf[var_off:fixed_off].view("<u8").astype(np.uint64) - np.uint64(24)
    var = buf[24:var_off]
    fixed = buf[fixed_off:]
    return n, var, offs, fixed


def write_fastb(path, packed: np.ndarray, byte_off: np.ndarray, lens: np.ndarray):
    _write_feudal(path, np.asarray(packed, np.uint8).tobytes(), byte_off, np.asarray(lens, "<u4").tobytes(), 4, 16, 1)


def read_fastb(path):
    """-> (packed u8, byte_off u64[n+1], len u32[n])"""
    n, var, offs, fixed = _read_feudal(path)
    lens = fixed.view("<u4").astype(np.uint32)
    if len(lens) != n:
        raise ValueError(f"{path}: fixed data is not one u32 per read")
    return np.ascontiguousarray(var), offs, lens


# --------------------------------------------------------------------------- PQVec
def _ceil_lg2(x: int) -> int:
    return int(x - 1).bit_length() if x > 1 else 0


def pq_encode(q: np.ndarray) -> bytes:
    """Encode one quality vector as a PQVec byte string: one block per <=255
    values (a valid, not cost-optimal, split; any valid split decodes the same)."""
    q = np.asarray(q, dtype=np.uint8)
    if q.size and int(q.max()) > 63:
        raise ValueError("quality > 63 (feudal/PQVec.cc:30-35 is fatal on these)")
    out = bytearray()
    for i in range(0, len(q), 255):
        blk = q[i:i + 255]
        mn, mx = int(blk.min()), int(blk.max())
        nbits = _ceil_lg2(mx + 1 - mn)
        out.append(len(blk))
        bits = nbits | (mn << 3)
        nb = 9
        if nbits:
            for v in blk:
                bits |= (int(v) - mn) << nb
                nb += nbits
        out += int(bits).to_bytes((nb + 7) // 8, "little")
    out.append(0)
    return bytes(out)


def _pq_encode_rows(q2d: np.ndarray):
    """Vectorised pq_encode for equal-length rows (len <= 255): -> list of (row_idx, u8[n_sub, nbytes])"""
    n, L = q2d.shape
    mn = q2d.min(axis=1)
    rng = q2d.max(axis=1).astype(np.int32) - mn + 1
    nbits = np.zeros(n, dtype=np.int64)
    for b in range(1, 7):
        nbits[rng > (1 << (b - 1))] = b
    out = []
    for b in np.unique(nbits):
        b = int(b)
        idx = np.nonzero(nbits == b)[0]
        sub = q2d[idx]
        hdr = (b | (mn[idx].astype(np.uint16) << 3)).astype("<u2")
        hbits = np.unpackbits(hdr.view(np.uint8).reshape(-1, 2), axis=1, bitorder="little")[:, :9]
        if b:
            v = (sub - mn[idx][:, None]).astype(np.uint8)
            vbits = np.unpackbits(v[:, :, None], axis=2, bitorder="little")[:, :, :b].reshape(len(idx), L * b)
            bits = np.concatenate([hbits, vbits], axis=1)
        else:
            bits = hbits
        body = np.packbits(bits, axis=1, bitorder="little")
        rec = np.concatenate([np.full((len(idx), 1), L, np.uint8), body, np.zeros((len(idx), 1), np.uint8)], axis=1)
        out.append((idx, rec))
    return out


def pq_decode(blob) -> np.ndarray:
    """Decode one PQVec byte string (feudal/PQVec.cc:129-188 semantics)."""
    b = bytes(blob)
    out = []
    p = 0
    while True:
        nqs = b[p]
        p += 1
        if nqs == 0:
            break
        nbits = b[p] & 7
        nbytes = (nqs * nbits + 9 + 7) // 8
        bits = int.from_bytes(b[p:p + nbytes], "little")
        p += nbytes
        mn = (bits >> 3) & 63
        bits >>= 9
        mask = (1 << nbits) - 1
        for _ in range(nqs):
            out.append(mn + (bits & mask))
            bits >>= nbits
    return np.array(out, dtype=np.uint8)


def write_qualp(path, quals: np.ndarray, off: np.ndarray):
    off = np.asarray(off, dtype=np.int64)
    n = len(off) - 1
    lens = np.diff(off)
    blobs = [None] * n
    quals = np.asarray(quals, dtype=np.uint8)
    if quals.size and int(quals.max()) > 63:
        raise ValueError("quality > 63 (feudal/PQVec.cc:30-35 is fatal on these)")
    sizes = np.zeros(n, dtype=np.int64)
    groups = []
    for L in np.unique(lens):
        L = int(L)
        rows = np.nonzero(lens == L)[0]
        if L == 0 or L > 255:
            for r in rows:
                blobs[r] = pq_encode(quals[off[r]:off[r + 1]])
                sizes[r] = len(blobs[r])
            continue
        for c0 in range(0, len(rows), 1 << 19):                 # in pieces: the index arrays below are 8 B per quality value
            rws = rows[c0:c0 + (1 << 19)]
            q2d = quals[(off[rws][:, None] + np.arange(L)[None, :])]
            for idx, rec in _pq_encode_rows(q2d):
                groups.append((rws[idx], rec))
                sizes[rws[idx]] = rec.shape[1]
    eoff = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(sizes, out=eoff[1:])
    var = np.zeros(int(eoff[-1]), dtype=np.uint8)
    for rows, rec in groups:
        dst = eoff[rows].astype(np.int64)[:, None] + np.arange(rec.shape[1])[None, :]
        var[dst] = rec
    for r in range(n):
        if blobs[r] is not None:
            var[int(eoff[r]):int(eoff[r + 1])] = np.frombuffer(blobs[r], np.uint8)
    _write_feudal(path, var.tobytes(), eoff, b"", 0, 8, 1)


def write_qualp_blobs(path, pq: np.ndarray, pq_off: np.ndarray):
    """a .qualp from already encoded PQVec byte strings (element r = pq[pq_off[r]:pq_off[r+1]])"""
    _write_feudal(path, np.ascontiguousarray(pq, np.uint8).tobytes(), np.asarray(pq_off, np.uint64), b"", 0, 8, 1)


def read_qualp(path):
    """-> (pq bytes u8, pq_off u64[n+1])"""
    n, var, offs, fixed = _read_feudal(path)
    return np.ascontiguousarray(var), offs


def qualp_to_raw(pq: np.ndarray, pq_off: np.ndarray):
    """decode every PQVec -> (quals u8 concatenated, off u64[n+1]); host-side, test/IO helper"""
    parts = [pq_decode(pq[int(pq_off[i]):int(pq_off[i + 1])]) for i in range(len(pq_off) - 1)]
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    np.cumsum([len(x) for x in parts], out=off[1:])
    return (np.concatenate(parts) if parts else np.zeros(0, np.uint8)), off


# --------------------------------------------------------------------------- HBV
@dataclass
class HBV:
    K: int
    from_off: np.ndarray      # u64[nv+1]
    from_v: np.ndarray        # i32  targets, per vertex ascending (ties: insertion order)
    from_e: np.ndarray        # i32  edge-object ids, parallel to from_v
    to_off: np.ndarray        # u64[nv+1]
    to_e: np.ndarray          # i32  edge-object ids entering each vertex
    edge_packed: np.ndarray   # u8   each object ceil(len/4) bytes
    edge_byte_off: np.ndarray # u64[ne+1]
    edge_len: np.ndarray      # u32[ne]

    @property
    def n_vertices(self):
        return len(self.from_off) - 1

    @property
    def n_edges(self):
        return len(self.edge_len)

    def edge_codes(self):
        return unpack_bases(self.edge_packed, self.edge_byte_off, self.edge_len)

    def to_left_right(self):
        """(to_left i32[ne], to_right i32[ne]): the vertices an edge object leaves and enters (digraphE::ToLeft / ToRight)"""
        deg = np.diff(np.asarray(self.from_off, dtype=np.int64))
        src = np.repeat(np.arange(self.n_vertices, dtype=np.int32), deg)
        tl = np.full(self.n_edges, -1, np.int32); tr = np.full(self.n_edges, -1, np.int32)
        fe = np.asarray(self.from_e, dtype=np.int64)
        tl[fe] = src; tr[fe] = np.asarray(self.from_v, dtype=np.int32)
        return tl, tr


def _csr_bytes(off: np.ndarray, vals: np.ndarray) -> bytes:
    n = len(off) - 1
    out = bytearray(struct.pack("<Q", n))
    off = np.asarray(off, dtype=np.int64)
    vals = np.asarray(vals, dtype="<i4")
    for v in range(n):
        a, b = int(off[v]), int(off[v + 1])
        out += struct.pack("<Q", b - a)
        out += vals[a:b].tobytes()
    return bytes(out)


def hbv_to_bytes(h: HBV, zero_padding: bool = False) -> bytes:
    """the .hbv file image.  zero_padding: clear the unused 2-bit groups in the last byte of every edge object -- the
    reference's large-K writer leaves there what its edge builder pushed and popped last (BigKPather.cc:241-251 push_back /
    pop_back on a reused bvec), bits no reader looks at; files are compared "modulo padding" with this set on both sides"""
    out = bytearray(b"BINWRITE")
    out += struct.pack("<i", h.K)
    out += _csr_bytes(h.from_off, h.from_v)
    out += _csr_bytes(h.from_off, h.from_e)
    out += _csr_bytes(h.to_off, h.to_e)
    out += struct.pack("<Q", h.n_edges)
    bo = np.asarray(h.edge_byte_off, dtype=np.int64)
    for e in range(h.n_edges):
        n = int(h.edge_len[e])
        out += struct.pack("<I", n)
        b = np.asarray(h.edge_packed[bo[e]:bo[e + 1]], np.uint8)
        if zero_padding and n & 3:
            b = b.copy(); b[-1] &= (1 << (2 * (n & 3))) - 1
        out += b.tobytes()
    return bytes(out)


def write_hbv(path, h: HBV):
    with open(path, "wb") as f:
        f.write(hbv_to_bytes(h))


def _parse_csr(buf: memoryview, p: int):
    (n,) = struct.unpack_from("<Q", buf, p)
    p += 8
    off = np.zeros(n + 1, dtype=np.uint64)
    chunks = []
    for v in range(n):
        (d,) = struct.unpack_from("<Q", buf, p)
        p += 8
        chunks.append(np.frombuffer(buf, dtype="<i4", count=d, offset=p))
        p += 4 * d
        off[v + 1] = off[v] + np.uint64(d)
    vals = np.concatenate(chunks).astype(np.int32) if chunks else np.zeros(0, np.int32)
    return off, vals, p


def read_hbv(path) -> HBV:
    with open(path, "rb") as f:
        raw = f.read()
    buf = memoryview(raw)
    if raw[:8] != b"BINWRITE":
        raise ValueError(f"{path}: missing BINWRITE magic")
    (K,) = struct.unpack_from("<i", buf, 8)
    p = 12
    from_off, from_v, p = _parse_csr(buf, p)
    from_off2, from_e, p = _parse_csr(buf, p)
    to_off, to_e, p = _parse_csr(buf, p)
    (E,) = struct.unpack_from("<Q", buf, p)
    p += 8
    lens = np.zeros(E, dtype=np.uint32)
    boff = np.zeros(E + 1, dtype=np.uint64)
    chunks = []
    for e in range(E):
        (nb,) = struct.unpack_from("<I", buf, p)
        p += 4
        nby = (nb + 3) // 4
        chunks.append(np.frombuffer(buf, dtype=np.uint8, count=nby, offset=p))
        p += nby
        lens[e] = nb
        boff[e + 1] = boff[e] + np.uint64(nby)
    if p != len(raw):
        raise ValueError(f"{path}: {len(raw) - p} trailing bytes")
    if not np.array_equal(from_off, from_off2):
        raise ValueError(f"{path}: from_ / from_edge_obj_ shape mismatch")
    packed = np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)
    return HBV(K, from_off, from_v, from_e, to_off, to_e, packed, boff, lens)


# --------------------------------------------------------------------------- paths
def paths_to_bytes(offset: np.ndarray, path_off: np.ndarray, edges: np.ndarray) -> bytes:
    n = len(offset)
    po = np.asarray(path_off, dtype=np.int64)
    lens = np.diff(po)
    if n and int(lens.max(initial=0)) > 0xFFFF:
        raise ValueError("path longer than 65535 edges (ReadPath.cc:10 stores u16)")
    # record i occupies 6 + 4*len bytes
    rec_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(6 + 4 * lens, out=rec_off[1:])
    out = np.zeros(8 + int(rec_off[-1]), dtype=np.uint8)
    out[:8] = np.frombuffer(struct.pack("<Q", n), dtype=np.uint8)
    base = 8 + rec_off[:-1]
    o4 = np.asarray(offset, dtype="<i4").view(np.uint8).reshape(n, 4) if n else np.zeros((0, 4), np.uint8)
    l2 = lens.astype("<u2").view(np.uint8).reshape(n, 2) if n else np.zeros((0, 2), np.uint8)
    for j in range(4):
        out[base + j] = o4[:, j]
    for j in range(2):
        out[base + 4 + j] = l2[:, j]
    tot = int(po[-1]) if n else 0
    if tot:
        rid = np.repeat(np.arange(n, dtype=np.int64), lens)
        within = np.arange(tot, dtype=np.int64) - po[:-1][rid]
        dst = base[rid] + 6 + 4 * within
        e4 = np.asarray(edges, dtype="<i4").view(np.uint8).reshape(tot, 4)
        for j in range(4):
            out[dst + j] = e4[:, j]
    return out.tobytes()


def write_paths(path, offset, path_off, edges):
    with open(path, "wb") as f:
        f.write(paths_to_bytes(offset, path_off, edges))


def read_paths(path):
    """-> (offset i32[n], path_off u64[n+1], edges i32[])"""
    raw = np.fromfile(path, dtype=np.uint8)
    (n,) = struct.unpack("<Q", raw[:8].tobytes())
    offset = np.zeros(n, dtype=np.int32)
    po = np.zeros(n + 1, dtype=np.uint64)
    chunks = []
    p = 8
    buf = memoryview(raw.tobytes())
    for i in range(n):
        o, l = struct.unpack_from("<iH", buf, p)
        p += 6
        offset[i] = o
        chunks.append(np.frombuffer(buf, dtype="<i4", count=l, offset=p))
        p += 4 * l
        po[i + 1] = po[i] + np.uint64(l)
    if p != len(raw):
        raise ValueError(f"{path}: {len(raw) - p} trailing bytes")
    edges = np.concatenate(chunks).astype(np.int32) if chunks else np.zeros(0, np.int32)
    return offset, po, edges


# --------------------------------------------------------------------------- freqs
def freqs_text(hist) -> str:
    """hist: 101 counts -> small_K.freqs text (BuildReadQGraph.cc:1108-1112)"""
    return "".join(f"{i}, {int(hist[i])}\n" for i in range(1, 101))
