"""Build a VARIANT of libavsr_hip.so for A/B timing: python tools/build_variant.py <tag> [extra hipcc flags ...]
-> avsr-tf1_amd/csrc/_probe/libavsr_hip_<tag>.so (git-ignored, travels with gpurun); load it with AVSR_LIB=<path>."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avsr_tf1_amd import build as B                                  # noqa: E402

tag, flags = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(B.CSRC, "_probe", "obj_" + tag)
os.makedirs(out_dir, exist_ok=True)
procs, objs = [], []
for s in B.SOURCES:
    o = os.path.join(out_dir, s[:-4] + ".o")
    cmd = [B._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-result", "-Wno-pass-failed",
           "-I", os.path.join(ROOT, "include"), "-I", B.CSRC] + flags + ["-c", os.path.join(B.CSRC, s), "-o", o]
    procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs.append(o)
for cmd, p in procs:
    out, _ = p.communicate()
    if p.returncode:
        raise SystemExit("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode()))
lib = os.path.join(B.CSRC, "_probe", "libavsr_hip_%s.so" % tag)
subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
print(lib)
