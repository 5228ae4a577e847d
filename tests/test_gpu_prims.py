"""The library's own device-wide primitives (csrc/step2_prims.hip, hand-written since round 5: single-pass decoupled-look-back scans, a
maximum, a stable LSD radix sort of (u64, u32) pairs) against host-side references, at sizes around every tile boundary."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 255, 2047, 2048, 2049, 4097, 100_003, 1_000_000, 5_000_001])
@pytest.mark.parametrize("key_bits", [8, 40, 60, 64, -13, -33])       # negative: 64 random key bits, sorted by the low |key_bits| only (ties keep their order)
def test_scans_maximum_and_sort_match_the_host(n, key_bits):
    import torch
    assert torch.cuda.is_available()
    from w2rap_contigger_amd import step2
    L = step2.lib()
    L.w2rap_step2_selftest_prims.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int]
    L.w2rap_step2_selftest_prims.restype = C.c_int
    with step2.Step2Context(0) as ctx:
        assert L.w2rap_step2_selftest_prims(ctx.h, n, 17 + n, key_bits) == 0
