"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle, bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


@pytest.mark.parametrize("n", [64, 128, 256, 512, 1024, 2048, 4096, 8192])
def test_mdct_reverse_bit_exact(oracle, gpu_ctx, n):
    """IMdct.Reverse (Mdct.cs:13-21): HIP k_mdct_reverse == oracle restatement, every bit."""
    torch = _torch()
    rng = np.random.default_rng(n)
    batch = 37
    x = rng.uniform(-1, 1, (batch, n)).astype(np.float32)
    x[0, : n // 2] = 0.0
    x[1, : n // 2] = 1e-41  # denormals must survive
    ref = np.stack([oracle.mdct_reverse(x[b], n) for b in range(batch)])
    d = torch.from_numpy(x.copy()).cuda()
    gpu_ctx.mdct_reverse(n, batch, d.data_ptr(), n)
    gpu_ctx.synchronize()
    got = d.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), np.abs(got - ref).max()


@pytest.mark.parametrize("name", ["1test", "2test", "3test", "issue6test"])
@pytest.mark.parametrize("batch_frames", [7, 1024])
def test_ogg_files_bit_exact(oracle, gpu_ctx, ogg_bytes, name, batch_frames):
    """VorbisReader.ReadSamples over the shipped TestFiles: GPU PCM == oracle PCM, every bit."""
    import nvorbis_amd as nv
    ref, info = oracle.decode_ogg(ogg_bytes[name])
    rd = nv.VorbisReader(ogg_bytes[name], ctx=gpu_ctx, batch_frames=batch_frames)
    got = rd.read_all()
    assert rd.Channels == info["channels"]
    assert got.size == ref.size
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), np.abs(got - ref).max()
    assert rd.HasClipped == info["has_clipped"]
    rd.close()
