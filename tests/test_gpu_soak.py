"""Reduced soaks of the batch path (tools/soak_extract.py, tools/soak_extract_s64.py: the full runs' logs are kept under
profiles/): many random irregular scans and synthetic 64-ring scans of three generator settings through k_organize + the ring
kernel (more than 64 slots: the one-workgroup-per-scan organise path) against the oracle, bit for bit, plus association index
tuples on every fifth scan; and tools/soak_hot_path.py: the stages downstream of the feature clouds (association, vote, normal
equations, Gauss-Newton step) on consecutive scans of four data shapes with a random pose guess per slot."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, *args, **extra_env):
    env = dict(os.environ, GRAFT_REPO_ROOT=ROOT, **extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + [str(a) for a in args], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    return out.stdout


def test_irregular_scans_soak_800():
    out = _run("soak_extract.py", 2)                # 2 seeds x 400 scans (the committed log: 8 seeds)
    assert "soak passed:" in out and int(out.strip().split("soak passed:")[1].split()[0]) > 700


def test_s64_generator_settings_soak_36():
    # 3 settings x 12 scans (the committed log: 3 x 96 through the default dispatch); LIGHTLOAM_ORG_SMALL=0 sends these
    # 12-scan calls through k_organize, the path batches of more than 64 scans take
    out = _run("soak_extract_s64.py", 12, LIGHTLOAM_ORG_SMALL="0")
    assert "soak passed: 36 scans" in out


def test_hot_path_soak_4_shapes_x_9_pairs():
    # the committed log: 4 shapes x 255 pairs
    out = _run("soak_hot_path.py", 10)
    assert "hot-path soak passed: 36 scan pairs" in out


def test_frame_loops_soak_2_drives():
    # odometry frame loop + free-running cube map on a synthetic and an HDL-64E drive (the committed log: 2 x 399 / 2 x 200 frames)
    out = _run("soak_frames.py", 14, 8)
    assert "frame-loop soak passed: 2 x 13 odometry frames, 2 x 8 mapping frames" in out


def test_hot_path_soak_all_8_shapes_x_5_pairs():
    # also 16 / 32 / 128 rings and the HDL-64E table in firing order (the committed log: 8 shapes x 383 pairs)
    out = _run("soak_hot_path.py", 6, LL_SOAK_ALL_SHAPES="1")
    assert "hot-path soak passed: 40 scan pairs" in out
