"""Build the in-tree native library: HIP kernels + C ABI -> himg_amd/lib/libhimg_hip.so.

hipcc cross-compiles gfx950 code objects without a GPU, so this runs in the
build container as well as on the MI355X box.  The .so is git-ignored but
travels with the repo snapshot to the GPU box.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "himg_amd", "csrc")
LIBDIR = os.path.join(ROOT, "himg_amd", "lib")
LIB = os.path.join(LIBDIR, "libhimg_hip.so")

HIP_SOURCES = ["kernels_enc.hip", "kernels_dec.hip", "himg_hip.hip", "himg_multi.hip"]
CXX_SOURCES = ["encoder.cpp", "decoder.cpp"]
C_SOURCES = ["himg_tables.c", "himg_synth.c"]
HEADERS = ["himg_dev.h", "loop_counts.h", "himg_tables.h", "ctx_pool.h", "../../include/himg_hip.h",
           "../../include/encoder.h", "../../include/decoder.h"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; the HIMG engine has no CPU fallback")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    """Compile every HIP/C++/C source for gfx950 into one shared library."""
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES + CXX_SOURCES + C_SOURCES
            if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS]
    # Experimental -D flags are part of the rebuild decision: a library built with
    # them is not silently reused by a run without them (and vice versa).
    extra = os.environ.get("HIMG_EXTRA_HIPCC_FLAGS", "").strip()
    stamp = os.path.join(LIBDIR, "hipcc_flags.stamp")
    old_extra = open(stamp).read() if os.path.exists(stamp) else ""
    if extra != old_extra:
        force = True
    if not force and not _stale(LIB, deps):
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    for s in srcs:
        o = os.path.join(LIBDIR, os.path.basename(s) + ".o")
        if force or _stale(o, [s] + deps[len(srcs):]):
            if s.endswith(".c"):
                cmd = ["gcc", "-O2", "-fPIC", "-std=c99", "-c", s, "-o", o] + inc
            elif s.endswith(".cpp"):
                cmd = ["g++", "-O2", "-fPIC", "-std=c++11", "-c", s, "-o", o] + inc
            else:
                cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17",
                       "-Wno-unused-value", "-c", s, "-o", o] + inc
                cmd += os.environ.get("HIMG_EXTRA_HIPCC_FLAGS", "").split()   # experiments (-D...)
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
        objs.append(o)
    tmp = LIB + ".tmp"
    subprocess.run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", tmp] + objs
                   + ["-lpthread"], check=True)
    os.replace(tmp, LIB)
    with open(stamp, "w") as fh:
        fh.write(extra)
    return LIB


CLIDIR = os.path.join(ROOT, "himg_amd", "cli")
BINDIR = os.path.join(ROOT, "himg_amd", "bin")


def build_cli(verbose=False):
    """chimg / dhimg / benchmark (the reference's three callers) against the in-tree library."""
    lib = build_lib(verbose=verbose)
    os.makedirs(BINDIR, exist_ok=True)
    out = []
    for name in ("chimg", "dhimg", "benchmark"):
        src = os.path.join(CLIDIR, name + ".cpp")
        exe = os.path.join(BINDIR, name)
        if _stale(exe, [src, os.path.join(CLIDIR, "pnm_io.h"), lib]):
            cmd = ["g++", "-std=c++11", "-O2", "-I" + os.path.join(ROOT, "include"), "-I" + CLIDIR, src,
                   "-L" + LIBDIR, "-lhimg_hip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,$ORIGIN/../lib",
                   "-o", exe]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
        out.append(exe)
    return out


REF_SRC = "/root/reference/src"


def build_ref_benchmark(verbose=False):
    """The reference's OWN benchmark main (src/benchmark.cpp), compiled where it lies and
    unchanged, linked against this library -- FreeImage, which that main only needs for
    non-HIMG files, is a test-only link stub whose loaders fail (tests/cpp/fi_stub).  Built
    in the container that has /root/reference; the binary (git-ignored, like the .so)
    travels to the GPU box, where tests/test_reference_benchmark.py runs it.  Returns the
    path, or None when neither the sources nor a prebuilt binary are there."""
    exe = os.path.join(BINDIR, "ref_benchmark")
    src = os.path.join(REF_SRC, "benchmark.cpp")
    if not os.path.exists(src):
        return exe if os.path.exists(exe) else None
    lib = build_lib(verbose=verbose)
    os.makedirs(BINDIR, exist_ok=True)
    stub_dir = os.path.join(ROOT, "tests", "cpp", "fi_stub")
    stub = os.path.join(stub_dir, "freeimage_link_stub.cpp")
    if _stale(exe, [src, stub, os.path.join(stub_dir, "FreeImage.h"), lib]):
        cmd = ["g++", "-std=c++11", "-O2", "-I" + os.path.join(ROOT, "include"), "-I" + stub_dir, src, stub,
               "-L" + LIBDIR, "-lhimg_hip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,$ORIGIN/../lib", "-o", exe]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return exe


def build_unit_checks(verbose=False):
    """Device-side unit checks (tests/ run them on the GPU box): tile_plane_check
    compares the packed inverse transform with a scalar host model."""
    os.makedirs(BINDIR, exist_ok=True)
    # (the HBM calibration kernels bench.py's extras and tools/calibrate_pmc.sh run)
    cal_src = os.path.join(ROOT, "tools", "micro", "hbm_calib.hip")
    cal_exe = os.path.join(BINDIR, "hbm_calib")   # (built here, never taken from the tree: the binary matches the source and the box's ROCm)
    if _stale(cal_exe, [cal_src]):
        subprocess.run([_hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", cal_src, "-o", cal_exe], check=True)
    src = os.path.join(ROOT, "tools", "micro", "tile_plane_check.hip")
    exe = os.path.join(BINDIR, "tile_plane_check")
    deps = [src, os.path.join(CSRC, "kernels_dec.hip")] + [os.path.join(CSRC, h) for h in HEADERS]
    if _stale(exe, deps):
        cmd = [_hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-Wno-unused-value",
               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, src, "-o", exe]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return [exe]


def build_oracle(verbose=False):
    """Build the CPU checker (tests / smoke / cpu_baseline only)."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")] + ([] if verbose else ["-s"]),
                   check=True)


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
    build_oracle(verbose=True)
