"""The drop-in boundary without a GPU: the in-tree HIP library loads, exports every symbol include/lightloam_hip.h
declares, validates parameters, and refuses to run without a gfx950 device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lightloam_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ll_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_reference_seams():
    syms = declared_symbols()
    for s in ("ll_extract_batch", "ll_set_target", "ll_associate_batch", "ll_vote_batch", "ll_residual_jacobian",
              "ll_normal_equations_batch", "ll_gn_step_batch", "ll_hot_path_batch"):
        assert s in syms


def test_library_exports_every_declared_symbol(api):
    lib = api.load_library()
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.ll_abi_version() == 3          # 3: ll_params.voxel_sort_ranks, .input_stride_floats
    assert sorted(api.EXPORTS) == sorted(s for s in declared_symbols())


def test_default_params_are_the_reference_constants(api):
    p = api.default_params(64)
    assert (p.n_scans, p.minimum_range, round(p.lower_bound, 4), p.up_bound) == (64, 5.0, -24.9, 2.0)
    assert (round(p.curv_threshold, 6), round(p.gap_sq_threshold, 6), round(p.leaf_size, 6)) == (0.1, 0.05, 0.2)
    assert (p.nn_dist_sq_max, p.nearby_scan, round(p.huber_delta, 6)) == (25.0, 2.5, 0.1)
    assert api.default_params(16).minimum_range == pytest.approx(0.3)


def test_create_validates_before_touching_a_device(api):
    with pytest.raises(api.LightLoamError) as e:
        api.Context(api.default_params(48))                      # scan_line must be 16/32/64 (scanRegistration.cpp:447-451)
    assert e.value.code == -3
    with pytest.raises(api.LightLoamError) as e:
        api.Context(api.default_params(64, max_points=500000))   # the reference arrays hold 400000 points (:34-40)
    assert e.value.code == -2
    with pytest.raises(api.LightLoamError) as e:
        api.Context(api.default_params(64, batch=0))
    assert e.value.code == -2
    for bad in (dict(voxel_sort_ranks=2), dict(input_stride_floats=5), dict(input_stride_floats=2)):
        with pytest.raises(api.LightLoamError) as e:
            api.Context(api.default_params(64, **bad))
        assert e.value.code == -2, bad
    p = api.default_params(64)
    assert (p.voxel_sort_ranks, p.input_stride_floats) == (0, 4)     # auto ranking; KITTI .bin / PointXYZ stride


def test_no_cpu_fallback(api):
    """Without a HIP device ll_create must fail with LL_ERR_DEVICE; with one this test is not applicable."""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    with pytest.raises(api.LightLoamError) as e:
        api.Context(api.default_params(64))
    assert e.value.code == -1 and "no CPU fallback" in str(e.value)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "light-loam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".c", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("import orc", "from oracle", "oracle.orc", "ll_oracle", "orc_", "libll_oracle"):
                    assert needle not in src, f"{f} references the oracle ({needle})"
