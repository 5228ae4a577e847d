"""The C-ABI library loads and exports every symbol include/w2rap_step2.h declares; without a
GPU the product path fails loudly (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, load_fixture
from w2rap_contigger_amd import step2


def declared_functions(header="w2rap_step2.h", stem="w2rap_step2_"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + stem + r"[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = step2.lib()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libw2rap_step2.so does not export {n}"
    assert lib.w2rap_step2_abi_version() == 3
    names3 = declared_functions("w2rap_step3.h", "w2rap_step3_")
    assert set(names3) == {"w2rap_step3_run", "w2rap_step3_run_after_step2", "w2rap_step3_free", "w2rap_step3_profile"}
    for n in names3:
        assert hasattr(lib, n), f"libw2rap_step2.so does not export {n}"
    namesg = declared_functions("w2rap_gfa.h", "w2rap_gfa_")
    assert set(namesg) == {"w2rap_gfa_dump", "w2rap_gfa_free", "w2rap_gfa_profile"}
    for n in namesg:
        assert hasattr(lib, n), f"libw2rap_step2.so does not export {n}"
    names1 = declared_functions("w2rap_step1.h", "w2rap_step1_")
    assert set(names1) == {"w2rap_step1_run", "w2rap_step1_free", "w2rap_step1_run_into_step2", "w2rap_step1_profile"}
    for n in names1:
        assert hasattr(lib, n), f"libw2rap_step2.so does not export {n}"


def test_struct_layouts_match_header():
    # sizes the C compiler gives the structs (x86-64 SysV): guards the ctypes mirror
    assert C.sizeof(step2.Reads) == 72
    assert C.sizeof(step2.EdgeHint) == 32
    assert C.sizeof(step2.Params) == 56          # + n_gpus, n_passes, devices (ABI version 2), flags (version 3)
    assert C.sizeof(step2.Xchg) == 16 + 64 * 8   # w2rap_xchg of the sharded graph phase
    assert C.sizeof(step2.Out) == 8 + 8 * 2 + 8 * 11 + 8 + 8 * 2 + 8 + 8 * 3 + 101 * 8 + 8 * 5 + 4 * 3 + 4


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert step2.lib().w2rap_step2_device_count() == 0
    with pytest.raises(step2.Step2Error) as e:
        step2.Step2Context(0)
    assert e.value.code == 2
    fx = load_fixture("random20k")
    with pytest.raises(step2.Step2Error) as e:
        step2.build_read_qgraph(fx["packed"], fx["byte_off"], fx["read_len"], pq=fx["pq"], pq_off=fx["pq_off"])
    assert e.value.code == 2 and "no CPU fallback" in str(e.value)
    # Step 3: argument errors are reported before the device is touched, a valid call fails for want of a GPU
    from w2rap_contigger_amd import formats as F, step3
    from conftest import GOLDEN
    h = F.read_hbv(os.path.join(GOLDEN, "random20k.ref.hbv")); p = F.read_paths(os.path.join(GOLDEN, "random20k.ref.paths"))
    with pytest.raises(step2.Step2Error) as e:
        step3.repath_in_memory(h, p, 201)
    assert e.value.code == 1
    with pytest.raises(step2.Step2Error) as e:
        step3.repath_in_memory(h, p, 200)
    assert e.value.code == 2 and "no CPU fallback" in str(e.value)
    assert C.sizeof(step3.Step3In) == 96 and C.sizeof(step3.Step3Params) == 56       # Step3In + n_vertices, vleft, vright (--extend_paths)
    # Step 1
    from w2rap_contigger_amd import step1
    with pytest.raises(step2.Step2Error) as e:
        step1.extract_reads(b"@a\nACGT\n+\nIIII\n", b"@a\nACGT\n+\nIIII\n")
    assert e.value.code == 2 and "no CPU fallback" in str(e.value)
    from w2rap_contigger_amd import gfa
    with pytest.raises(step2.Step2Error) as e:
        gfa.gfa_dump(h)
    assert e.value.code == 2 and "no CPU fallback" in str(e.value)
    assert C.sizeof(gfa.GfaIn) == 80 and C.sizeof(gfa.GfaParams) == 16 and C.sizeof(gfa.GfaOut) == 216
    assert C.sizeof(step1.Step1In) == 40 and C.sizeof(step1.Step1Params) == 8 and C.sizeof(step1.Step1Out) == 104


def test_bad_k_is_rejected():
    import numpy as np
    L = step2.lib()
    r = step2.Reads(0, None, None, None, None, None, None, None, 0)
    p = step2.Params(61, 7, 4, 0, None, None)
    o = step2.Out()
    err = C.create_string_buffer(256)
    assert L.w2rap_step2_run(C.byref(r), C.byref(p), C.byref(o), err, 256) == 1
    assert b"K must be 60" in err.value


def _tool(out_dir, *extra):
    import subprocess
    exe = os.path.join(ROOT, "w2rap_contigger_amd", "w2rap-step2")
    return subprocess.run([exe, "-o", out_dir, "-p", "t", *extra], capture_output=True, text=True, timeout=120)


def test_tool_rejects_truncated_inputs(tmp_path):
    """the standalone tool checks every count in its input files against the bytes that are there (no GPU needed: these
    checks come before the first library call)"""
    import shutil
    from conftest import GOLDEN
    d = str(tmp_path)
    fb, qp = os.path.join(d, "frag_reads_orig.fastb"), os.path.join(d, "frag_reads_orig.qualp")
    shutil.copy(os.path.join(GOLDEN, "palindrome_circle.qualp"), qp)
    raw = open(os.path.join(GOLDEN, "palindrome_circle.fastb"), "rb").read()
    # (1) a .fastb cut in the middle of its offsets table, (2) one whose first offset points outside the file
    open(fb, "wb").write(raw[: len(raw) // 2])
    r = _tool(d)
    assert r.returncode == 1 and "feudal" in r.stderr
    bad = bytearray(raw)
    var_off = int.from_bytes(raw[8:16], "little")
    bad[var_off:var_off + 8] = (1 << 40).to_bytes(8, "little")
    open(fb, "wb").write(bytes(bad))
    r = _tool(d)
    assert r.returncode == 1 and "feudal" in r.stderr
    # (3) a truncated --edge_order_from .hbv, (4) one with an absurd degree count
    open(fb, "wb").write(raw)
    hbv = open(os.path.join(GOLDEN, "palindrome_circle.ref.hbv"), "rb").read()
    hp = os.path.join(d, "hint.hbv")
    for blob in (hbv[: len(hbv) // 3], hbv[:20] + (1 << 60).to_bytes(8, "little") + hbv[28:]):
        open(hp, "wb").write(blob)
        r = _tool(d, "--edge_order_from", hp)
        assert r.returncode == 1 and "truncated" in r.stderr, r.stderr


def test_step3_tool_rejects_truncated_inputs(tmp_path):
    import shutil, subprocess
    from conftest import GOLDEN
    exe = os.path.join(ROOT, "w2rap_contigger_amd", "w2rap-step3")
    d = str(tmp_path)
    hbv = open(os.path.join(GOLDEN, "palindrome_circle.ref.hbv"), "rb").read()
    paths = open(os.path.join(GOLDEN, "palindrome_circle.ref.paths"), "rb").read()
    for hb, pb, what in ((hbv[: len(hbv) // 2], paths, "truncated"), (hbv, paths[: len(paths) // 2], "truncated"), (b"NOTANHBV" + hbv[8:], paths, "BINWRITE")):
        open(os.path.join(d, "t.small_K.hbv"), "wb").write(hb)
        open(os.path.join(d, "t.small_K.paths"), "wb").write(pb)
        r = subprocess.run([exe, "-o", d, "-p", "t"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 1 and what in r.stderr, r.stderr
