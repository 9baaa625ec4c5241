"""bench.py started as a plain command line with --gpus N > 1 launches its own ranks: the parent makes no HIP call and never
imports torch (replacing / re-launching a process that has initialised the GPU is forbidden on the pool), the children get
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* like under torch.distributed.run, rank 0's JSON line is the parent's last line and
the children's exit code is the parent's."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("extra", [[], ["--mode", "map", "--row-parallel"], ["--stream-input", "--rings", "128"]])
def test_dry_launch_prints_one_command_per_rank_and_imports_no_torch(extra):
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--dry-launch"] + extra,
                         capture_output=True, text=True, env=_clean_env(), timeout=120)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["dry_launch"] and d["n_ranks"] == 4 and d["parent_imported_torch"] is False
    assert d["command"][1] == BENCH and "--dry-launch" not in d["command"] and d["command"][2:4] == ["--gpus", "4"]
    for x in extra:
        assert x in d["command"]
    assert [e["RANK"] for e in d["rank_env"]] == ["0", "1", "2", "3"] == [e["LOCAL_RANK"] for e in d["rank_env"]]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" for e in d["rank_env"])
    assert len({e["MASTER_PORT"] for e in d["rank_env"]}) == 1


def test_under_a_launcher_the_process_is_a_rank_not_a_parent():
    """WORLD_SIZE in the environment (torch.distributed.run): no second level of children -- the process goes on as rank 0 and
    needs a GPU, which this container lacks.  (On a GPU box it would wait for a rank 1 that nobody starts: skipped there.)"""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("a GPU is visible: rank 0 would wait for the rendezvous of a world nobody launched")
    env = _clean_env(); env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2", "--no-cpu-baseline", "--rings", "16"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode != 0 and "no HIP device" in (out.stderr + out.stdout)
    assert "rank exit codes" not in out.stderr                     # it did not act as a launching parent


def test_children_exit_code_is_relayed():
    """no GPU here: every rank stops with "no HIP device visible" and the parent must say so with a non-zero code (on a GPU box
    with fewer than 2 devices the second rank fails instead; with 2 or more this is simply a tiny successful run)."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2", "--no-cpu-baseline", "--rings", "16"],
                         capture_output=True, text=True, env=_clean_env(), timeout=600)
    try:
        import torch
        n_dev = torch.cuda.device_count()
    except Exception:
        n_dev = 0
    if n_dev >= 2:
        assert out.returncode == 0, out.stderr[-2000:]
        assert json.loads(out.stdout.strip().splitlines()[-1])["n_gpus"] == 2
    else:
        assert out.returncode != 0
        assert "rank exit codes" in out.stderr


@pytest.mark.gpu
def test_self_launched_two_ranks_share_the_gpu_over_gloo():
    """The whole N = 2 path of `python bench.py --gpus 2` on a one-GPU box: parent launches two ranks, both on device 0, gloo for
    the barrier and the max-over-ranks reduction (RCCL refuses two ranks on one device); rank 0's line reports both ranks' scans."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "2", "--warmup", "1", "--batch", "256",
                          "--no-cpu-baseline"], capture_output=True, text=True, env=_clean_env(), timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["scans_per_gpu_per_step"] == 256 and abs(d["value"] - 2 * 256 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert d["metric"].endswith("64-ring cloud")


def test_traffic_is_taken_from_the_profile_only_when_its_stamp_matches(tmp_path):
    """roofline.traffic comes from profiles/pmc_traffic.json ONLY when that file was measured on this code (source digest), ring
    count, workload and batch -- a stale file must not dress a new kernel in an old number."""
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    args = argparse.Namespace(rings=64, batch=16384, workload="synthetic")
    good = {"source_digest": bench.source_digest(), "rings": 64, "batch": 16384, "workload": "synthetic",
            "kernels": {"k_ring_features": {"hbm_bytes_per_scan": 6.5e6}}}
    p = tmp_path / "t.json"
    p.write_text(json.dumps(good))
    t, why = bench.traffic_from_profile(args, "k_ring_features", 1, str(p))
    assert t == 6.5e6 * 16384 and "source digest" in why
    for key, val in (("source_digest", "0" * 16), ("rings", 128), ("batch", 8192), ("workload", "hdl64")):
        p.write_text(json.dumps(dict(good, **{key: val})))
        t, why = bench.traffic_from_profile(args, "k_ring_features", 1, str(p))
        assert t is None and key in why
    t, why = bench.traffic_from_profile(args, "k_ring_features", 1, str(tmp_path / "absent.json"))
    assert t is None and "absent" in why
    # the committed file is the one tools/profile_round.sh stamped; whether it matches this tree is reported, never assumed
    t, why = bench.traffic_from_profile(args, "k_ring_features", 1)
    assert (t is None) == ("not used" in why or "absent" in why or "no entry" in why)
    assert len(bench.source_digest()) == 16
