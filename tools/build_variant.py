"""Build an A/B variant of libmichigan_hip.so in the CPU container: one or more translation units recompiled with extra -D flags, linked
with the other (up-to-date) objects into michigan_amd/lib/variants/lib_<name>.so.  The GPU-side scripts swap it in, or a measurement tool
loads it with MG_LIB=<path> (michigan_amd/_cabi.py).
    python tools/build_variant.py epi_scalar mg_conv_halo.hip -DMG_EPI_SCALAR=1
    python tools/build_variant.py probes mg_conv.hip mg_conv_halo.hip mg_wgrad3x3.hip -DMG_PROBES=1      # tools/probe_halo.py, tools/probe_wgrad3x3.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from michigan_amd import build as B   # noqa: E402

name = sys.argv[1]
srcs = [a for a in sys.argv[2:] if not a.startswith("-")]
flags = [a for a in sys.argv[2:] if a.startswith("-")]
B.build(verbose=False)
hipcc = B._hipcc()
objs = [os.path.join(B.OBJ_DIR, os.path.splitext(s)[0] + ".o") for s in B.SOURCES if s not in srcs]
for src in srcs:
    obj = os.path.join("/tmp", "variant_%s_%s.o" % (name, os.path.splitext(src)[0]))
    subprocess.run([hipcc, *B.CXXFLAGS, *flags, "-c", os.path.join(B.CSRC, src), "-o", obj], check=True, capture_output=True)
    objs.append(obj)
out = os.path.join(B.LIB_DIR, "variants", "lib_%s.so" % name)
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.run([hipcc, "--offload-arch=%s" % B.ARCH, "-shared", "-fPIC", "-o", out, *objs], check=True, capture_output=True)
print(out)
