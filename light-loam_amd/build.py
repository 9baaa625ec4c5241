"""In-tree builds: the HIP C-ABI library (hipcc, gfx950) and the host-side synthetic-scan generator (gcc)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_HOST = os.path.join(_HERE, "host")
_ROOT = os.path.dirname(_HERE)

HIP_SOURCES = ["ll_api.hip", "ll_organize.hip", "ll_features.hip", "ll_pick.hip", "ll_associate.hip", "ll_vote.hip", "ll_factors.hip", "ll_mapping.hip", "ll_voxel.hip", "ll_cubemap.hip", "ll_functors.hip"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               # parity: the reference build has no FMA (baseline x86-64); contraction would change f32 results
               "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
               "-Wall", "-Wno-unused-function"]


def lib_path():
    # LIGHTLOAM_HIP_LIB lets a profiling tool load an instrumented build of the same sources
    return os.environ.get("LIGHTLOAM_HIP_LIB") or os.path.join(_HERE, "liblightloam_hip.so")


def synth_lib_path():
    return os.path.join(_HERE, "libll_synth.so")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _all_files(d, exts):
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(exts)]


def build_synth(force=False):
    src = os.path.join(_HOST, "ll_synth.c")
    out = synth_lib_path()
    if force or _newer(out, [src]):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", "-std=c99", "-o", out, src, "-lm"])
    return out


def build_hip(force=False, extra_flags=()):
    """One object per .hip file (compiled in parallel, rebuilt only when it or a header changed), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    out = lib_path()
    hdrs = _all_files(_CSRC, (".h", ".hpp")) + [os.path.join(_ROOT, "include", "lightloam_hip.h")]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    import hashlib                  # a stable name per flag set (hash() is salted per process: objects would never be reused)
    objdir = os.path.join(_CSRC, "_obj" + ("_" + hashlib.sha256(" ".join(extra_flags).encode()).hexdigest()[:8] if extra_flags else ""))
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra_flags) + ["-I", os.path.join(_ROOT, "include"), "-I", _CSRC]

    def compile_one(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        path = os.path.join(_CSRC, src)
        if force or _newer(obj, [path] + hdrs):
            subprocess.check_call([hipcc] + cflags + ["-c", "-o", obj, path])
            return obj, True
        return obj, False

    with ThreadPoolExecutor(max_workers=min(6, len(HIP_SOURCES))) as ex:
        res = list(ex.map(compile_one, HIP_SOURCES))
    objs = [o for o, _ in res]
    if force or any(ch for _, ch in res) or _newer(out, objs):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_all(force=False):
    build_synth(force)
    build_hip(force)
