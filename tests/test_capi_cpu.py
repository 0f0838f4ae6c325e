"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/beluga_mcl.h declares, and refuses to run without a GPU (no CPU fallback). No compute calls."""
import ctypes as C
import os
import re

import pytest

from beluga_amd import build as mcl_build
from beluga_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    mcl_build.build()
    return capi.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "beluga_mcl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mcl_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in beluga_mcl.h but not exported"
    assert sorted(capi.exported_names()) == declared, "capi.py signatures out of sync with beluga_mcl.h"


def test_default_config_matches_reference_defaults(lib):
    cfg = capi.Config()
    lib.mcl_default_config(C.byref(cfg))
    a = cfg.amcl  # amcl_core.hpp:34-55
    assert (a.update_min_d, a.update_min_a, a.resample_interval, a.selective_resampling) == (0.25, 0.2, 1, 0)
    assert (a.min_particles, a.max_particles, a.alpha_slow, a.alpha_fast, a.kld_epsilon, a.kld_z) == (500, 2000, 0.001, 0.1, 0.05, 3.0)
    assert cfg.motion.distance_threshold == 0.01  # differential_drive_model.hpp:67
    lf = cfg.lf  # likelihood_field_model_base.hpp:42-64
    assert (lf.max_obstacle_distance, lf.max_laser_distance, lf.z_hit, lf.z_random, lf.sigma_hit) == (100.0, 2.0, 0.5, 0.5, 0.2)
    b = cfg.beam  # beam_model.hpp:43-58
    assert (b.z_hit, b.z_short, b.z_max, b.z_rand, b.sigma_hit, b.lambda_short, b.beam_max_range) == (0.5, 0.5, 0.05, 0.05, 0.2, 0.1, 60.0)


def test_struct_sizes_match_c_layout(lib, tmp_path):
    # Ask a C compiler for the layout of every struct in the header and compare with the ctypes mirrors.
    import subprocess
    names = ["mcl_amcl_params", "mcl_diffdrive_params", "mcl_lf_params", "mcl_beam_params", "mcl_config", "mcl_estimate",
             "mcl_update_info", "mcl_weight_stats", "mcl_device_view"]
    mirrors = [capi.AmclParams, capi.DiffDriveParams, capi.LfParams, capi.BeamParams, capi.Config, capi.Estimate,
               capi.UpdateInfo, capi.WeightStats, capi.DeviceView]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "beluga_mcl.h"\nint main(void){' +
                   "".join(f'printf("%zu\\n", sizeof({n}));' for n in names) +
                   'printf("%zu\\n", offsetof(mcl_config, shard_offset));return 0;}')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert out[:-1] == [C.sizeof(m) for m in mirrors]
    assert out[-1] == capi.Config.shard_offset.offset


def test_estimate_from_sums_is_pure_host_math(lib):
    # two particles at (1,2,0) and (0,0,0), weights 1: test_estimation.cpp:129-139 (PureTranslation)
    import numpy as np
    from beluga_amd.amcl import estimate_from_sums
    sums = np.array([2.0, 2.0, 2.0, 0.0, 1.0, 2.0, 1.0, 2.0, 4.0, 0.0, 0.0, 0.0])
    pose, cov = estimate_from_sums(sums)
    assert pose[2] == pytest.approx(0.5) and pose[3] == pytest.approx(1.0) and pose[0] == pytest.approx(1.0)
    assert cov[0, 0] == pytest.approx(0.5) and cov[0, 1] == pytest.approx(1.0) and cov[1, 1] == pytest.approx(2.0)
    assert cov[2, 2] == pytest.approx(0.0, abs=1e-12)


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    cfg = capi.Config()
    lib.mcl_default_config(C.byref(cfg))
    ctx = capi._ctx()
    st = lib.mcl_create(C.byref(cfg), C.byref(ctx))
    assert st == capi.MCL_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.mcl_last_error(None)
    assert not ctx.value


def test_product_never_imports_oracle():
    # The oracle is test infrastructure: nothing under beluga_amd/ may import, link or dlopen it.
    pat = re.compile(r"(import\s+oracle|from\s+oracle|oracle[/.]binding|beluga_oracle|liboracle|orc_[a-z_]+\s*\()")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "beluga_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), f"{f} references the oracle"


def test_laser_scan_preparation_matches_oracle(lib):
    """mcl_prepare_laser_scan is host arithmetic: bit-identical to the oracle's restatement."""
    import numpy as np
    from beluga_amd.amcl import make_laser_scan, prepare_laser_scan
    from oracle import binding as orc
    rng = np.random.Generator(np.random.MT19937(3))
    ranges = rng.uniform(0.05, 40.0, 1081).astype(np.float32)
    ranges[::17] = np.nan
    ranges[5::29] = np.inf
    q = (0.01, -0.02, 0.38, 0.9247, 0.2, 0.0, 0.4)
    q = tuple(np.array(q[:4]) / np.linalg.norm(q[:4])) + q[4:]
    for max_beams in (2 ** 64 - 1, 1081, 1080, 360, 100, 2, 1):
        scan = make_laser_scan(ranges, -2.356, 0.004363, 0.1, 30.0, origin_se3=q, max_beams=max_beams, min_range=0.2, max_range=25.0)
        got = prepare_laser_scan(scan)
        want = orc.prepare_laser_scan(ranges, -2.356, 0.004363, 0.1, 30.0, origin_se3=q, max_beams=min(max_beams, 2 ** 63), min_range=0.2,
                                      max_range=25.0)
        assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("bits", [4, 5, 6])
def test_ordering_curve_is_a_hilbert_curve(lib, bits):
    """The heading-major ordering key follows a 3-D Hilbert curve through the (2^bits)^3 (heading, y, x) bins (kernels.h
    hilbert_index_3; pure host arithmetic through the C ABI): a bijection onto [0, 2^(3 bits)), consecutive positions are face
    neighbours (so any run of the spatial order is a connected set of bins - the property the LDS-patch kernel's workgroups
    rely on), it enters at (0, 0, 0) and leaves at (max, 0, 0) so that the slabs of the key's top heading bits chain."""
    import numpy as np
    n = 1 << bits
    index = np.empty((n, n, n), dtype=np.int64)
    for t in range(n):
        for y in range(n):
            for x in range(n):
                index[t, y, x] = lib.mcl_debug_curve_index(t, y, x, bits)
    flat = index.ravel()
    assert flat.min() == 0 and flat.max() == n ** 3 - 1 and len(np.unique(flat)) == n ** 3
    cell_of = np.empty((n ** 3, 3), dtype=np.int64)
    t, y, x = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    cell_of[flat] = np.stack([t.ravel(), y.ravel(), x.ravel()], axis=1)
    steps = np.abs(np.diff(cell_of, axis=0))
    assert np.all(steps.sum(axis=1) == 1), "consecutive positions of the curve must be face neighbours"
    assert tuple(cell_of[0]) == (0, 0, 0) and tuple(cell_of[-1]) == (n - 1, 0, 0)
    # locality of runs: the bounding box of ANY 512 consecutive positions stays within 16 bins per axis (a Morton run that
    # crosses the middle of the cube spans all of it)
    worst = 0
    for start in range(0, n ** 3 - 512, 97):
        run = cell_of[start:start + 512]
        worst = max(worst, int((run.max(axis=0) - run.min(axis=0)).max()) + 1)
    assert worst <= 16, worst
