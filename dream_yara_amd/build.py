"""Builds the gfx950 shared library and the host tools in-tree (dream_yara_amd/csrc/).

    python -m dream_yara_amd.build            # build what is out of date
    python -m dream_yara_amd.build --force

hipcc cross-compiles for gfx950 without a GPU.  Outputs: libdyhip.so (C ABI, include/dyhip.h) and the
command-line tools dream_yara_mapper / dream_yara_build_filter / dream_yara_indexer.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(CSRC, "libdyhip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _sources():
    out = [os.path.join(ROOT, "include", "dyhip.h")]
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".h", ".hpp", ".hip", ".cpp")):
            out.append(os.path.join(CSRC, f))
    return out


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force=False):
    srcs = _sources()
    if force or not _newer(LIB, srcs):
        _run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", LIB,
              os.path.join(CSRC, "dyhip.hip")])
    for tool in ("dream_yara_mapper", "dream_yara_build_filter", "dream_yara_indexer"):
        src = os.path.join(CSRC, tool + ".cpp")
        exe = os.path.join(CSRC, tool)
        if not os.path.exists(src):
            continue
        if force or not _newer(exe, srcs + [LIB]):
            _run(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, src, "-L" + CSRC, "-ldyhip", "-Wl,-rpath,$ORIGIN"])
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
