"""Build an A/B variant of libmichigan_hip.so in the CPU container: one translation unit recompiled with extra -D flags, linked with the
other (up-to-date) objects into michigan_amd/lib/variants/lib_<name>.so.  The GPU-side scripts (tools/ab_epi_scalar.sh, tools/ab_halo_sched.sh)
swap it in.     python tools/build_variant.py epi_scalar mg_conv_halo.hip -DMG_EPI_SCALAR=1"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from michigan_amd import build as B   # noqa: E402

name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build(verbose=False)
hipcc = B._hipcc()
obj = os.path.join("/tmp", "variant_%s_%s.o" % (name, os.path.splitext(src)[0]))
subprocess.run([hipcc, *B.CXXFLAGS, *flags, "-c", os.path.join(B.CSRC, src), "-o", obj], check=True, capture_output=True)
objs = [os.path.join(B.OBJ_DIR, os.path.splitext(s)[0] + ".o") for s in B.SOURCES if s != src] + [obj]
out = os.path.join(B.LIB_DIR, "variants", "lib_%s.so" % name)
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.run([hipcc, "--offload-arch=%s" % B.ARCH, "-shared", "-fPIC", "-o", out, *objs], check=True, capture_output=True)
print(out)
