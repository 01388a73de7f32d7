"""The host C++ that touches untrusted bytes -- container readers, page table and seek search (host_ogg.cpp), header and packet
parsers (host_setup.cpp, host_parse.cpp) -- built with AddressSanitizer + UndefinedBehaviorSanitizer and driven with mutated files
(tools/fuzz/*.cpp): valid-checksum page mutations, raw byte damage, truncated and bit-flipped setup headers and audio packets.
Passing = no sanitizer report; what the code RETURNS for such inputs is the parity tests' business.  No GPU, no HIP."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nvorbis_amd", "csrc")
FILES = [os.path.join(ROOT, "tests", "golden", n + ".ogg") for n in ("1test", "2test", "3test", "issue6test")]


def _build(tmp_path, harness, sources):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path / harness)
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", CSRC,
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "fuzz", harness + ".cpp")] + \
          [os.path.join(CSRC, s) for s in sources] + ["-o", exe]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    if r.returncode != 0 and ("asan" in r.stdout.lower() or "sanitize" in r.stdout.lower()):
        pytest.skip("sanitizer runtime not available: " + r.stdout[-300:])
    assert r.returncode == 0, r.stdout[-3000:]
    return exe


@pytest.mark.parametrize("harness,sources", [
    ("asan_ogg_fuzz", ["host_ogg.cpp"]),
    ("asan_parse_fuzz", ["host_ogg.cpp", "host_setup.cpp", "host_parse.cpp"]),
])
def test_host_code_is_clean_under_asan_and_ubsan(tmp_path, harness, sources):
    exe = _build(tmp_path, harness, sources)
    r = subprocess.run([exe] + FILES, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "no sanitizer report" in r.stdout, r.stdout[-4000:]
