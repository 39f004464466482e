"""Build libmvin_hip.so in-tree with hipcc for gfx950 (no JIT cache, no pip install).

    python -m mvin_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container too.
The .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import argparse
import hashlib
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), "include")
LIB_PATH = os.path.join(PKG_DIR, "libmvin_hip.so")
STAMP_PATH = LIB_PATH + ".stamp"
OBJ_DIR = os.path.join(PKG_DIR, "_build")
SOURCES = ["mvin_kernels.hip", "mvin_fused.hip", "mvin_fused_split.hip", "mvin_fused_packed.hip", "mvin_fused_wpp.hip", "mvin_fused_agg.hip", "mvin_fused_agg32.hip", "mvin_fused_wpp_fold.hip", "mvin_fused_d16.hip", "mvin_fused_d32.hip", "mvin_tail.hip", "mvin_tail_flash.hip", "mvin_score_small.hip", "mvin_keyaddr.hip", "mvin_keyaddr_stream.hip", "mvin_keyaddr_grouped.hip", "mvin_keyaddr_dense.hip", "mvin_keyaddr_static.hip", "mvin_keyaddr_flash.hip", "mvin_keyaddr_wave.hip", "mvin_hoist.hip", "mvin_probe.hip", "mvin_group.hip", "mvin_order.hip", "mvin_prep.hip", "mvin_linear_mfma.hip", "mvin_bwd.hip", "mvin_abi.hip"]
HEADERS = ["mvin_common.h", "mvin_kernels.h", "mvin_fused_agg.h"]
VERSION_SCRIPT = os.path.join(CSRC, "libmvin_hip.map")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-shared", f"--offload-arch={ARCH}"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def _existing(names, base):
    return [os.path.join(base, n) for n in names if os.path.exists(os.path.join(base, n))]


def _digest():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for path in _existing(SOURCES + HEADERS, CSRC) + [os.path.join(INCLUDE, "mvin_hip.h"), VERSION_SCRIPT]:
        with open(path, "rb") as f:
            h.update(path.encode())
            h.update(f.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB_PATH) or not os.path.exists(STAMP_PATH):
        return True
    with open(STAMP_PATH) as f:
        return f.read().strip() != _digest()


def _obj_digest(src):
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for path in [src] + _existing(HEADERS, CSRC) + [os.path.join(INCLUDE, "mvin_hip.h")]:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:20]


def build(force=False, verbose=False, jobs=None):
    """Compile every HIP source (one object per source, in parallel, cached under mvin_amd/_build by the
    hash of the source, the shared headers and the flags) and link them into one shared library.
    Returns the library path."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    # several processes may import the package on a fresh checkout at once (torchrun ranks): one builds, the others wait
    # on the lock and then find the stamp up to date
    import fcntl
    with open(os.path.join(OBJ_DIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB_PATH
            return _build_locked(force, verbose, jobs)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose, jobs):
    from concurrent.futures import ThreadPoolExecutor
    tmp_suffix = f".{os.getpid()}.tmp"
    hipcc = _hipcc()
    cflags = [f for f in FLAGS if f != "-shared"]
    todo, objs = [], []
    for src in _existing(SOURCES, CSRC):
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + "." + _obj_digest(src) + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj):
            todo.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + cflags + ["-c", f"-I{INCLUDE}", f"-I{CSRC}", src, "-o", obj + tmp_suffix]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed ({res.returncode}) on {src}:\n{res.stdout}\n{res.stderr}")
        if verbose and res.stderr:
            print(res.stderr, file=sys.stderr)
        os.replace(obj + tmp_suffix, obj)

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(6, os.cpu_count() or 1)) as ex:
            list(ex.map(compile_one, todo))
    keep = set(objs)
    for f in os.listdir(OBJ_DIR):        # objects of older source revisions
        if os.path.join(OBJ_DIR, f) not in keep and not f.endswith(".tmp") and f != ".lock":
            os.remove(os.path.join(OBJ_DIR, f))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", f"-Wl,--version-script={VERSION_SCRIPT}"] + objs + ["-o", LIB_PATH + tmp_suffix]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed ({res.returncode}):\n{res.stdout}\n{res.stderr}")
    os.replace(LIB_PATH + tmp_suffix, LIB_PATH)
    with open(STAMP_PATH, "w") as f:
        f.write(_digest())
    return LIB_PATH


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    print(build(force=a.force, verbose=a.verbose))
