"""Freeze the oracle: SHA-256 of the PCM oracle/ produces for the four shipped TestFiles (clipping on and off) into
tests/golden/oracle_pcm_digests.json, which tests/test_oracle_kat.py::test_oracle_pcm_is_frozen checks -- an edit to
oracle/ that changes a single sample is caught, whatever the product does.  This does NOT pin the oracle to the reference
(nothing in this image can: no .NET runtime); it pins it to itself as of the commit that wrote the file.

  python tools/freeze_oracle.py          (rewrites the file; say why in the commit message)"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import oracle_py  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "oracle_pcm_digests.json")
FILES = ("1test", "2test", "3test", "issue6test")


def digests():
    orc = oracle_py.load()
    out = {}
    for name in FILES:
        data = open(os.path.join(ROOT, "tests", "golden", name + ".ogg"), "rb").read()
        for clip in (True, False):
            pcm, info = orc.decode_ogg(data, clip=clip)
            out["%s clip=%d" % (name, clip)] = {"floats": int(pcm.size), "channels": int(info["channels"]),
                                               "sha256": hashlib.sha256(pcm.tobytes()).hexdigest()}
    return out


if __name__ == "__main__":
    d = digests()
    json.dump({"what": "sha256 of the float32 PCM oracle/ decodes from tests/golden/*.ogg (tools/freeze_oracle.py)", "digests": d},
              open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT)
