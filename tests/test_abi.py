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
