"""Build libmichigan_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

The library is pure HIP + a C ABI (include/michigan_hip.h): no torch headers, no
pybind -- one object per translation unit, compiled in parallel, linked into
``michigan_amd/lib/libmichigan_hip.so``.  hipcc cross-compiles for gfx950 without
a GPU, so this also runs in the CPU-only build container.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(REPO_ROOT, "include")
LIB_DIR = os.path.join(PKG_DIR, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libmichigan_hip.so")

ARCH = "gfx950"
SOURCES = ["mg_api.hip", "mg_conv.hip", "mg_wgrad.hip", "mg_norm.hip", "mg_pointwise.hip", "mg_pack.hip", "mg_conv_halo.hip", "mg_conv_halo64.hip", "mg_gabor.hip", "mg_wgrad3x3.hip", "mg_spectral.hip", "mg_conv_thin.hip", "mg_inputs.hip", "mg_conv_dot.hip", "mg_loss.hip", "mg_grad.hip", "mg_weights.hip", "mg_glue.hip", "mg_attention.hip", "mg_comm.hip"]
CXXFLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
            "-fno-gpu-rdc", f"-I{INCLUDE}", f"-I{CSRC}"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def _fingerprint(src: str = "") -> str:
    """Hash of the flags, every header, and either one translation unit (`src`) or all of them."""
    h = hashlib.sha256()
    h.update(" ".join(CXXFLAGS).encode())
    paths = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h") or (not src and f.endswith(".hip"))]
    paths += [os.path.join(INCLUDE, f) for f in sorted(os.listdir(INCLUDE))]
    if src:
        paths.append(os.path.join(CSRC, src))
    for p in paths:
        if os.path.isfile(p):
            h.update(os.path.basename(p).encode())
            with open(p, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def source_hash() -> str:
    """Hash of the kernel sources and the compile flags WITHOUT the checkout's absolute include paths: identical on every box for the
    same sources (profiles/*_conv_traffic.json is stamped with it; bench.py compares)."""
    h = hashlib.sha256()
    h.update(" ".join(f for f in CXXFLAGS if not f.startswith("-I")).encode())
    for d in (CSRC, INCLUDE):
        for f in sorted(os.listdir(d)):
            if f.endswith((".h", ".hip")):
                h.update(f.encode())
                with open(os.path.join(d, f), "rb") as fh:
                    h.update(fh.read())
    return h.hexdigest()[:16]


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile (if stale) and return the path of libmichigan_hip.so."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    stamp = os.path.join(LIB_DIR, "build.stamp")
    fp = _fingerprint()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(stamp):
        with open(stamp) as fh:
            if fh.read().strip() == fp:
                return LIB_PATH
    hipcc = _hipcc()

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
        ostamp, ofp = obj + ".stamp", _fingerprint(src)
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read().strip() == ofp:
            return obj                                   # this translation unit is up to date
        cmd = [hipcc, *CXXFLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[michigan_amd.build]", " ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{res.stdout}\n{res.stderr}")
        with open(ostamp, "w") as fh:
            fh.write(ofp)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp_lib = LIB_PATH + ".tmp.%d" % os.getpid()             # link aside and rename: a concurrent dlopen never sees a partial file
    link = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp_lib, *objs, "-ldl"]
    if verbose:
        print("[michigan_amd.build]", " ".join(link), flush=True)
    res = subprocess.run(link, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    os.replace(tmp_lib, LIB_PATH)
    with open(stamp, "w") as fh:
        fh.write(fp)
    return LIB_PATH


def build_locked(force: bool = False, verbose: bool = True) -> str:
    """build() under an exclusive file lock on lib/: when every rank of a multi-process launch finds the library missing,
    one compiles and the others block here and then find the stamp up to date."""
    import fcntl
    os.makedirs(LIB_DIR, exist_ok=True)
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return build(force=force, verbose=verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
