"""Replays of test files in child pytest processes under an environment (kernel-variant toggles, parser forms): the children of ONE
replay run side by side -- they are independent processes checking bit-exactness, nothing in them is timed -- and every one of them
must pass.  The parent test waits; no other test of the parent process runs meanwhile."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_children(children, timeout=1200):
    """children: [(files, env, extra pytest args)]; asserts that every child's pytest exits 0 (the tail of its output otherwise)."""
    from tests import oracle_py
    oracle_py.build()  # (no child finds the checker stale and rebuilds it beside another)
    procs = []
    for files, env, extra in children:
        log = tempfile.TemporaryFile(mode="w+")
        cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + list(extra)
        cmd += [os.path.join(ROOT, "tests", f) for f in files]
        procs.append((subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=log, stderr=subprocess.STDOUT, text=True), log, files))
    failed = []
    for p, log, files in procs:
        try:
            rc = p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
            rc = -9
        log.seek(0)
        out = log.read()
        log.close()
        if rc != 0:
            failed.append("%s: exit %d\n%s" % (" ".join(files), rc, out[-3000:]))
    assert not failed, "\n\n".join(failed)
