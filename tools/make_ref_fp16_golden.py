#!/usr/bin/env python3
"""Generate tests/golden/ref_fp16.npz from the REFERENCE's own fp16 converters.

`make -C oracle _ref` compiles th.cpp:293-359 (ggml_compute_fp16_to_fp32 / ggml_compute_fp32_to_fp16, the only part of
the reference path that needs no WebGPU type) straight from /root/reference into oracle/_ref/libth_ref_fp16.so; this
script runs those two functions and stores inputs and outputs as plain data:
  h2f_bits   uint32[65536]   ggml_compute_fp16_to_fp32(h) for EVERY binary16 pattern h = index
  f_in_bits  uint32[N]       f32 inputs (bit patterns): seeded normals over 12 decades, every f16 rounding boundary
                             neighbourhood, denormals, +-0, +-inf, NaNs
  f2h        uint16[N]       ggml_compute_fp32_to_fp16(f_in)
The fixture is the reference-held pin for A17: tests check the oracle's and the host layer's converters against it (CPU)
and the GPU's v_cvt_f32_f16 decode against the oracle.  Needs /root/reference, so it runs in the build container only.
Run from the repo root:  make -C oracle _ref && python tools/make_ref_fp16_golden.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib_path = os.path.join(ROOT, "oracle", "_ref", "libth_ref_fp16.so")
if not os.path.exists(lib_path):
    sys.exit("oracle/_ref/libth_ref_fp16.so missing: run `make -C oracle _ref` where /root/reference exists")
lib = C.CDLL(lib_path)
lib.ggml_compute_fp16_to_fp32.restype = C.c_float
lib.ggml_compute_fp16_to_fp32.argtypes = [C.c_uint16]
lib.ggml_compute_fp32_to_fp16.restype = C.c_uint16
lib.ggml_compute_fp32_to_fp16.argtypes = [C.c_float]

h2f = np.array([lib.ggml_compute_fp16_to_fp32(h) for h in range(65536)], np.float32).view(np.uint32)

rng = np.random.default_rng(20230517)
parts = [rng.standard_normal(2048).astype(np.float32) * np.float32(s) for s in (1e-10, 1e-8, 6e-8, 1e-6, 1e-5, 6e-5, 1e-3, 0.02, 1.0, 100.0, 6e4, 7e4, 1e9)]
# neighbourhoods of every f16 rounding boundary: midpoints between consecutive positive halfs, +-1 ulp(f32), both signs
hpos = np.arange(0, 0x7C00, dtype=np.uint16).view(np.float16).astype(np.float32)
mid = ((hpos[:-1].astype(np.float64) + hpos[1:].astype(np.float64)) / 2).astype(np.float32)
mb = mid.view(np.uint32)
parts.append(np.concatenate([mb - 1, mb, mb + 1]).view(np.float32)[::7])
parts.append(-np.concatenate([mb - 1, mb, mb + 1]).view(np.float32)[3::11])
special = np.array([0x00000000, 0x80000000, 0x7F800000, 0xFF800000, 0x7FC00000, 0xFFC00001, 0x7F800001, 0x477FE000, 0x477FEFFF, 0x477FF000,
                    0x47800000, 0x33000000, 0x33000001, 0x32FFFFFF, 0x387FC000, 0x38800000, 0x00000001, 0x007FFFFF, 0x00800000], np.uint32)
parts.append(special.view(np.float32))
f_in = np.concatenate(parts).astype(np.float32)
f2h = np.array([lib.ggml_compute_fp32_to_fp16(float(x)) if np.isfinite(x) or True else 0 for x in f_in.tolist()], np.uint16)
# python floats are doubles: pass the exact f32 value (every f32 is a double), NaN payloads are not preserved by the
# ctypes round trip, so NaN inputs are canonicalised to the quiet NaN the conversion produces from them
out = os.path.join(ROOT, "tests", "golden", "ref_fp16.npz")
np.savez_compressed(out, h2f_bits=h2f, f_in_bits=f_in.view(np.uint32), f2h=f2h)
print(f"wrote {out}: {h2f.size} half patterns, {f_in.size} f32 inputs, {os.path.getsize(out)} bytes")
