"""The drop-in gate of SURVEY.md section 7 step 3: the reference's OWN callers
(src/benchmark.cpp, src/chimg.cpp, src/dhimg.cpp) compile unchanged against
include/encoder.h + include/decoder.h, and every himg:: symbol they reference is
exported by the engine library.

The reference files are compiled where they lie under /root/reference (never
copied); FreeImage, which only the three mains use for file I/O, is absent from
this image, so a declarations-only stub header (tests/cpp/fi_stub/FreeImage.h,
test infrastructure) stands in for its header.  Skipped where /root/reference
does not exist (the GPU box)."""
import os
import subprocess

import pytest

import himg_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src"
MAINS = ["benchmark", "chimg", "dhimg"]

pytestmark = pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="reference sources not present")

FLAGS = ["-std=c++11", "-I" + os.path.join(ROOT, "include"),
         "-I" + os.path.join(ROOT, "tests", "cpp", "fi_stub")]


@pytest.mark.parametrize("name", MAINS)
def test_reference_main_compiles_unchanged(name):
    """-fsyntax-only with -Wall: the same text the reference ships, our headers."""
    r = subprocess.run(["g++", "-fsyntax-only", "-Wall"] + FLAGS + [os.path.join(REF_SRC, name + ".cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def _undefined(obj):
    out = subprocess.run(["nm", "-C", "--undefined-only", obj], capture_output=True, text=True, check=True).stdout
    return sorted(line.split(None, 1)[1].strip() for line in out.splitlines() if " U " in line or line.startswith("U "))


def _exported(lib):
    out = subprocess.run(["nm", "-C", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    return set(line.split(None, 2)[2].strip() for line in out.splitlines() if len(line.split(None, 2)) == 3)


@pytest.mark.parametrize("name", MAINS)
def test_reference_main_links_against_engine_symbols(name, tmp_path):
    """Compile the reference main to an object and check that every himg:: symbol
    it needs is defined by libhimg_hip.so (same mangled names = same signatures)."""
    himg_amd.lib()
    obj = str(tmp_path / (name + ".o"))
    subprocess.run(["g++", "-c", "-O1"] + FLAGS + [os.path.join(REF_SRC, name + ".cpp"), "-o", obj], check=True)
    need = [s for s in _undefined(obj) if s.startswith("himg::")]
    assert need, "the caller does not reference the codec API at all?"
    have = _exported(os.path.join(ROOT, "himg_amd", "lib", "libhimg_hip.so"))
    missing = [s for s in need if s not in have]
    assert not missing, missing
