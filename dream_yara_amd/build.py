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
    out = [os.path.join(ROOT, "include", "dyhip.h"), os.path.abspath(__file__)]
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".h", ".hpp", ".hip", ".cpp")):
            out.append(os.path.join(CSRC, f))
    return out


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force=False):
    srcs = _sources()
    lib_srcs = [s for s in srcs if not s.endswith(".cpp")]   # headers, .hip and this build script
    extra = os.environ.get("DY_HIPCC_FLAGS", "").split()           # tuning experiments only (e.g. -DDY_K3_WAVES=6)
    stamp = LIB + ".flags"                                         # a library built with other flags is stale whatever its mtime says
    built_with = open(stamp).read() if os.path.exists(stamp) else ""
    if force or not _newer(LIB, lib_srcs) or built_with != " ".join(extra):
        _run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC"] + extra + ["-o", LIB,
              os.path.join(CSRC, "dyhip.hip")])
        open(stamp, "w").write(" ".join(extra))
    for tool in ("dream_yara_mapper", "dream_yara_build_filter", "dream_yara_update_filter", "dream_yara_indexer"):
        src = os.path.join(CSRC, tool + ".cpp")
        exe = os.path.join(CSRC, tool)
        if not os.path.exists(src):
            continue
        if force or not _newer(exe, [src, LIB] + [s for s in srcs if s.endswith((".h", ".hpp"))]):
            link = ["-L" + CSRC, "-ldyhip", "-Wl,-rpath,$ORIGIN"]
            if tool == "dream_yara_mapper":
                # BAM output, gzip input; the HIP runtime API for the device buffers of the multi-GPU exchange (multi_gpu.hpp; RCCL itself is
                # loaded on demand).  -D__HIP_PLATFORM_AMD__ only tells the runtime header which of its two API sets to declare.
                link = ["-DDY_HAVE_ZLIB", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"] + link + ["-lz", "-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-Wl,-rpath,/opt/rocm/lib"]
            _run(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, src] + link)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
