"""TEST INFRASTRUCTURE ONLY -- stage the reference's Python packages for the GPU box.

`/root/reference` exists in the builder container only.  The GPU box receives a snapshot of this repository, so the one parity leg
that needs the reference's OWN code next to a GPU -- `tests/test_dropin.py[hip]`: the unmodified `trainers/pix2pix_trainer.py` /
`models/pix2pix_model.py` driving the HIP classes through `michigan_amd.dropin.install()` on the real kernels -- and `bench.py`'s
`cpu_baseline.kind = "reference"` could never run there.  This script packs the five packages the harness imports
(`models`, `trainers`, `options`, `util`, `data`; `.py` files only, 0.3 MB) into ONE archive, `oracle/_ref/reference_py.zip`:

  * `oracle/_ref/` is git-ignored (the history stays free of reference sources) but NOT gpurun-ignored, so the archive travels with
    the snapshot exactly like the built `libmichigan_hip.so`;
  * `oracle/ref_harness.py` puts the archive on `sys.path` (zipimport) when `/root/reference` is absent -- the reference has no
    `__file__`-relative code, so it imports from the archive unchanged;
  * `__graft_entry__.build()` calls `stage()` wherever `/root/reference` is present, next to compiling the HIP library
    ("building the checker is not using it"); nothing on the product path reads the archive.

    python oracle/stage_reference.py        # -> oracle/_ref/reference_py.zip
"""
from __future__ import annotations

import hashlib
import os
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
ARCHIVE = os.path.join(REF_DIR, "reference_py.zip")
PACKAGES = ("models", "trainers", "options", "util", "data")


def stage(reference_root: str = "/root/reference", verbose: bool = True):
    """Write the archive (deterministic: sorted members, fixed timestamps) and return its path, or None without a checkout."""
    if not os.path.isdir(os.path.join(reference_root, "models", "networks")):
        return None
    members = []
    for pkg in PACKAGES:
        for d, _, files in os.walk(os.path.join(reference_root, pkg)):
            for f in files:
                if f.endswith(".py"):
                    p = os.path.join(d, f)
                    members.append((os.path.relpath(p, reference_root), p))
    members.sort()
    os.makedirs(REF_DIR, exist_ok=True)
    tmp = ARCHIVE + ".tmp.%d" % os.getpid()
    h = hashlib.sha256()
    with zipfile.ZipFile(tmp, "w", zipfile.ZIP_DEFLATED) as z:
        for rel, p in members:
            with open(p, "rb") as fh:
                data = fh.read()
            h.update(rel.encode()); h.update(data)
            zi = zipfile.ZipInfo(rel, date_time=(2020, 1, 1, 0, 0, 0))
            zi.compress_type = zipfile.ZIP_DEFLATED
            z.writestr(zi, data)
    os.replace(tmp, ARCHIVE)
    if verbose:
        print(f"[stage_reference] {len(members)} files -> {ARCHIVE} ({os.path.getsize(ARCHIVE)} bytes, sha256 of contents {h.hexdigest()[:16]})")
    return ARCHIVE


if __name__ == "__main__":
    stage()
