#!/usr/bin/env python3
"""Where k_ring_features spends its time, without instrumenting it: copies of the library whose kernel returns after
phase i (-DLL_PHASE_STOP=i) are timed on the same batch under full occupancy (HIP events around the launch, the
library's own profiler); the differences between consecutive stops are the phases' shares of the launch.

    python tools/phase_stop_time.py build          # CPU box: builds _phase/liblightloam_hip_stop<i>.so (travels with gpurun)
    python tools/phase_stop_time.py [batch]        # GPU box: one child process per build
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STOPS = [(3, "lists + bitmap + segment loads + bounds + voxel keys"), (4, "voxel sort"), (12, "gather + run heads + scan"),
         (13, "centroid sums (own range)"), (5, "runs continued into later lanes"),
         (7, "centroid stores + list gather + small-cloud offsets"), (9, "small-list stores (full kernel)")]


def lib(stop):
    return os.path.join(ROOT, "_phase", f"liblightloam_hip_stop{stop}.so")


def child(stop, batch):
    os.environ["LIGHTLOAM_HIP_LIB"] = lib(stop)
    sys.path.insert(0, ROOT)
    import lightloam_amd  # noqa: F401
    from lightloam_amd import api, synth
    cfg = synth.default_cfg(64)
    scans = [synth.scan(cfg, k) for k in range(8)]
    ctx = api.Context(api.default_params(64, batch=batch, max_points=max(map(len, scans))))
    for i in range(batch):
        ctx.upload_scan(i, scans[i % 8])
    ctx.extract(0, batch)
    ctx.synchronize()
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    for _ in range(5):
        ctx.extract(0, batch)
    ctx.synchronize()
    prof = ctx.profile_read(reset=True)
    ms, n = prof["k_ring_features"]
    print(f"{ms / n:.4f}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        sys.path.insert(0, ROOT)
        import lightloam_amd  # noqa: F401
        from lightloam_amd import build
        os.makedirs(os.path.join(ROOT, "_phase"), exist_ok=True)
        procs = []
        for stop, _ in STOPS:
            os.environ["LIGHTLOAM_HIP_LIB"] = lib(stop)
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            cmd = [hipcc] + build.HIPCC_FLAGS + ([] if stop == 9 else [f"-DLL_PHASE_STOP={stop}"]) + \
                  ["-I", os.path.join(ROOT, "include"), "-I", build._CSRC, "-o", lib(stop)] + [os.path.join(build._CSRC, s) for s in build.HIP_SOURCES]
            procs.append(subprocess.Popen(cmd))
            if len(procs) == 4:
                for p in procs:
                    assert p.wait() == 0
                procs = []
        for p in procs:
            assert p.wait() == 0
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(int(sys.argv[2]), int(sys.argv[3]))
        sys.exit(0)
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    prev = 0.0
    rows = []
    for stop, name in STOPS:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(stop), str(batch)], capture_output=True, text=True)
        if out.returncode != 0:
            print(f"stop {stop}: failed\n{out.stderr[-400:]}")
            continue
        ms = float(out.stdout.strip().splitlines()[-1])
        rows.append((name, ms, ms - prev))
        prev = ms
    total = rows[-1][1]
    print(f"k_ring_features, {batch} S64 scans per launch: time of the launch when the kernel returns after each phase")
    for name, ms, d in rows:
        print(f"  {name:44s} {ms:8.3f} ms  +{d:7.3f} ms  {100.0 * d / total:5.1f}%")
