"""Compile the HIP engine (libavsr_hip.so) for gfx950 in-tree.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(CSRC, "libavsr_hip.so")
SOURCES = ["capi.hip", "gemm.hip", "step.hip", "rnn.hip", "rnn_persist.hip", "rnn_persist_bwd.hip", "attention.hip", "attn_rnn.hip", "beam_gemm.hip", "dec_persist.hip", "dec_persist_bwd.hip", "elementwise.hip", "conv.hip", "conv_direct.hip", "conv_mfma.hip", "conv_wgrad.hip", "conv_bn.hip"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found; the HIP engine cannot be built")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps.append(os.path.join(ROOT, "include", "avsr_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    objs = []
    procs = []
    for s in srcs:
        o = s[:-4] + ".o"
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-result",
               "-Wno-pass-failed", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", s, "-o", o]
        cmd[1:1] = os.environ.get("AVSR_HIPCC_FLAGS", "").split()     # e.g. -DPERSIST_TIMING for the probes
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode()))
        if verbose and out:
            print(out.decode())
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stdout.decode()))
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
