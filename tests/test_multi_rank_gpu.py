"""The N > 1 code paths on the one GPU a test box has: bench.py's rank launch, barriers, MAX-over-ranks timing and
rank-0 reporting, and the corpus transcoder's LPT shard + gather of device-resident PCM -- two ranks on device 0 over gloo
(NVH_BENCH_SHARE_GPU=1).  Not a scaling measurement (the 8-GPU run is the driver's): it checks that the line is well formed,
that the PCM digests hold on every rank, and that the gathered corpus PCM is byte-identical to the single-rank run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_bench_two_ranks_share_one_gpu():
    env = dict(os.environ)
    env["NVH_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--min-timed-ms", "60",
                        "--working-set-mib", "64", "--no-configs", "--no-cpu-baseline", "--no-unfused", "--c5-scale", "0.1"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["pcm_digest_ok"] is True and d["pcm_digests_checked"] >= 3
    # the one collective of the north star is in the line: the corpus shard + gather block, and what the ranks talked through
    assert d["collective"]["world"] == 2 and d["collective"]["backend"] == "gloo"
    c5 = d["c5"]
    assert c5["files"] == 1004 and c5["pcm_sha256_ok"] is True and c5["files_checked"] == 1004
    assert c5["decode_s"] > 0 and c5["gather_s"] > 0 and 0 < c5["gathered_bytes_over_links"] < c5["pcm_bytes"]
    assert d["value"] > 1e6 and "NVH_BENCH_SHARE_GPU" in d["data"]
    # value = frames of all ranks / the slowest rank's time
    assert abs(d["value"] - 2 * 4096 * d["config"]["passes_per_step"] * d["steps"] / d["config"]["timed_region_s"]) < 1e-6 * d["value"]


def test_bench_eight_ranks_share_one_gpu():
    """The driver's 8-GPU launch shape on the one GPU of a test box: eight ranks through bench.py's own launcher, 8 x 3 decoder
    instances, eight worker pools, the eight-way LPT shard (every rank builds ONLY its shard from the committed sizes), the
    per-rank SHA-256 checks, the gather into the root and its device-side word sums.  A dry run of the code path, not a rate."""
    env = dict(os.environ)
    env["NVH_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--min-timed-ms", "60",
                        "--working-set-mib", "64", "--no-configs", "--no-cpu-baseline", "--no-unfused", "--c5-scale", "0.1", "--c5-workers", "4"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["pcm_digest_ok"] is True
    assert d["collective"]["world"] == 8 and d["collective"]["backend"] == "gloo"
    c5 = d["c5"]
    assert c5["files"] == 1004 and c5["pcm_sha256_ok"] is True and c5["files_checked"] == 1004
    assert c5["check"] == dict(c5["check"], per_rank_sha256_ok=True, inputs_ok=True, gathered_word_sums_ok=True)
    assert 100 <= c5["untimed_rank0"]["files_built"] <= 140  # an eighth of the corpus, not all of it
    assert 0 < c5["decode_s_min_rank"] <= c5["decode_s"] and c5["gather_s"] > 0
    assert 0.8 * c5["pcm_bytes"] < c5["gathered_bytes_over_links"] < c5["pcm_bytes"]  # seven eighths crossed to the root


def test_corpus_transcode_two_ranks_equals_one_rank():
    def run(world):
        env = dict(os.environ)
        env["NVH_BENCH_SHARE_GPU"] = "1"
        cmd = [sys.executable]
        if world > 1:
            cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                    "--master-port", "29731"]
        cmd += [os.path.join(ROOT, "tools", "corpus_transcode.py"), "--files", "13", "--workers", "4"]
        r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:]
        return _last_json(r.stdout)
    one, two = run(1), run(2)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert one["pcm_floats"] == two["pcm_floats"] > 0
    assert one["pcm_sha256"] == two["pcm_sha256"]


def test_corpus_transcode_one_rank_through_rccl():
    """RCCL itself on the one GPU a test box has: a process group of one rank over backend "nccl" (two ranks cannot share a device
    under RCCL), so that the library initialises under this code and the counts' all_gather and the barriers of the gather go
    through it with device tensors.  The point-to-point payloads need a peer: they stay with the driver's 8-GPU run."""
    def run(rccl):
        env = dict(os.environ)
        env.pop("NVH_BENCH_SHARE_GPU", None)
        cmd = [sys.executable]
        if rccl:
            env["NVH_RCCL_WORLD1"] = "1"
            cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29733"]
        cmd += [os.path.join(ROOT, "tools", "corpus_transcode.py"), "--files", "9", "--workers", "4"]
        r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:]
        return _last_json(r.stdout)
    plain, rccl = run(False), run(True)
    assert plain["backend"] is None and rccl["backend"] == "nccl" and rccl["n_gpus"] == 1
    assert plain["pcm_sha256"] == rccl["pcm_sha256"] and plain["pcm_floats"] == rccl["pcm_floats"] > 0


def test_native_rccl_gather_entry_points(tmp_path):
    """include/nvorbis_hip.h's nvh_comm_* (the gather for a host without torch.distributed: RCCL's C API inside the library) on
    the one GPU of a test box: a communicator of one rank, the counts' all-gather, and the payload through ncclSend / ncclRecv
    (NVH_GATHER_SELF_P2P: the root's own part takes the point-to-point path too) -- byte-identical to the plain run, and to the
    device-copy form."""
    import numpy as np
    import torch
    import nvorbis_amd as nv
    ctx = nv.Context(0)
    comm = nv.Comm(ctx, nv.Comm.unique_id(), 0, 1)
    try:
        assert comm.allgather_i64([5, 0, 7]) == [[5, 0, 7]]
        src = torch.arange(1 << 20, dtype=torch.float32, device="cuda:0") * 0.5
        for flags in (0, nv.Comm.SELF_P2P):
            dst = torch.full((src.numel(),), -1.0, dtype=torch.float32, device="cuda:0")
            torch.cuda.synchronize()
            comm.gather_pcm(src.data_ptr(), src.numel(), dst.data_ptr(), [src.numel()], 0, flags)
            assert torch.equal(src, dst)
        with pytest.raises(nv.NvhError):  # counts must say what is sent
            comm.gather_pcm(src.data_ptr(), src.numel(), src.data_ptr(), [src.numel() - 1], 0, 0)
    finally:
        comm.close()
        ctx.close()

    def run(extra):
        env = dict(os.environ)
        env.pop("NVH_BENCH_SHARE_GPU", None)
        cmd = [sys.executable, os.path.join(ROOT, "tools", "corpus_transcode.py"), "--files", "9", "--workers", "4"] + extra
        r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:]
        return _last_json(r.stdout)
    plain = run([])
    native = run(["--native-gather", str(tmp_path / "comm_id"), "--self-p2p"])
    assert native["backend"].startswith("rccl") and plain["backend"] is None
    assert plain["pcm_sha256"] == native["pcm_sha256"] and plain["pcm_floats"] == native["pcm_floats"] > 0
