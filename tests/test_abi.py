"""CPU tests of the drop-in boundary: libbcd_hip.so loads without a GPU and exports every symbol that
include/bcd_hip.h declares; host-only entry points behave; device entry points fail loudly without a device."""
import ctypes as C
import os
import re

import numpy as np

import bcd_amd.hip as bh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "bcd_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bcd_hip_[a-z_0-9]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported():
    lib = bh.lib()
    syms = header_symbols()
    assert len(syms) >= 25
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(bh.SYMBOLS) == syms


def test_no_torch_types_in_the_abi():
    txt = open(os.path.join(ROOT, "include", "bcd_hip.h")).read()
    assert "torch" not in txt.lower() and "at::" not in txt and "#include <hip" not in txt


def test_default_params_mirror_reference_defaults():
    p = bh.default_params()
    assert (p.hist_dist_threshold, p.patch_radius, p.search_radius) == (1.0, 1, 6)
    assert abs(p.min_eigen_value - 1e-8) < 1e-12 and p.use_random_pixel_order == 1 and p.marked_skip_probability == 1.0


def test_visit_order_is_a_permutation_of_main_pixels():
    W, H, w = 37, 23, 1
    main = np.array([l * W + c for l in range(w, H - w) for c in range(w, W - w)], np.int32)
    scan = bh.visit_order(W, H, w, 0, 5)
    assert np.array_equal(scan, main)                      # reference 1-thread -r 0 order (Denoiser.cpp:136-146)
    r1, r2, r3 = bh.visit_order(W, H, w, 1, 5), bh.visit_order(W, H, w, 1, 5), bh.visit_order(W, H, w, 1, 6)
    assert np.array_equal(r1, r2) and not np.array_equal(r1, r3) and not np.array_equal(r1, scan)
    assert np.array_equal(np.sort(r1), main)
    assert bh.scale_seed(10, 0) == 10 and bh.scale_seed(10, 2) != bh.scale_seed(10, 1)


def test_device_entry_points_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    assert bh.lib().bcd_hip_device_count() == 0
    h = C.c_void_p()
    rc = bh.lib().bcd_hip_ctx_create(C.byref(h), 0, None)
    assert rc == -2 and not h.value                          # BCD_HIP_EDEVICE: no CPU fallback
    import bcd_amd.core as core
    col, ns, hist, cov = core.synthetic_scene(16, 12, 2)
    ok, out, _ = core.denoise(col, ns, hist, cov, 1)
    assert not ok                                            # bcd::Denoiser::denoise() returns false, like bad inputs


def test_sparse_upload_packer_every_simd_form_the_host_has():
    """the host half of the sparse histogram upload (bcd_sparse_upload.hip): one 32-value group -> mask bits + the values whose bit pattern is not
    zero, in order; scalar, AVX2 and AVX-512 forms must agree with the definition (incl. -0.0f = 0x80000000, which is a VALUE).  Each form in a
    child process (the choice is made once per process)"""
    import subprocess
    import sys
    code = """
import ctypes as C, numpy as np, sys
sys.path.insert(0, %r)
import bcd_amd.hip as bh
lib = bh.lib()
rng = np.random.default_rng(1)
kind = -1
for trial in range(1500):
    dens = rng.choice([0.0, 0.05, 0.3, 0.7, 1.0])
    v = ((rng.random(32) < dens) * rng.integers(1, 2**32, 32, dtype=np.uint64)).astype(np.uint32)
    if trial %% 7 == 0: v[rng.integers(32)] = 0x80000000
    out = np.zeros(64, np.uint32); bits = C.c_uint32(0); cnt = C.c_int(0)
    kind = lib.bcd_hip_selftest_pack32(v.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.byref(bits), C.byref(cnt))
    want = v[v != 0]
    assert cnt.value == len(want) and bits.value == sum(1 << i for i in range(32) if v[i] != 0) and np.array_equal(out[:len(want)], want)
print("KIND", kind)
""" % ROOT
    kinds = set()
    for simd in ("scalar", "avx2", "avx512"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BCD_HIP_UPLOAD_SIMD=simd), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-800:]
        kinds.add(int(r.stdout.split("KIND")[1]))
    assert 0 in kinds                                        # the scalar form always exists; the others when the host has them
