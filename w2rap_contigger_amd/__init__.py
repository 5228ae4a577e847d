"""MI355X-native Step 2 (k=60 graph build + read pathing) of w2rap-contigger.

Only the hot path lives here: `csrc/` (HIP kernels + the C ABI of
libw2rap_step2.so), `step2` (ctypes binding mirroring buildReadQGraph/FixPaths),
`formats` (the Step-1/2/3 on-disk formats) and `synth` (seeded synthetic reads).
"""
__all__ = ["formats", "synth", "step2"]
