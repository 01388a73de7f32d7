"""Where does nvh_floor0_apply differ from the oracle's Floor0.Apply?  (tools: diagnostic for DESIGN.md section 8)"""
import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import nvorbis_amd as nv
from tests import synth_stream as ss, oracle_py
orc = oracle_py.load()
ctx = nv.Context(0)
rng = np.random.default_rng(12)
pk, _, _ = ss.filtered_stream(orc, "floor0_stereo", 4, 21)
d = orc.open_headers(pk[:3])
st = nv.Stream(ctx, pk[0], pk[1], pk[2])
t, order, _ = st.floor_info(0)
n = st.block1
half, batch = n // 2, 64
amps = rng.uniform(0.25, 6.0, batch).astype(np.float32)
coeffs = np.sort(rng.uniform(0.05, 3.1, (batch, order + 3)), axis=1).astype(np.float32)
res = np.ones((batch, half), np.float32)
got = torch.from_numpy(res.copy()).cuda()
st.floor0_apply(0, n, amps, coeffs, got.data_ptr(), half)
got = got.cpu().numpy()
bad_frames = 0
for b in range(batch):
    ref = np.zeros(st.block1, np.float32); ref[:half] = res[b]
    cf = np.ascontiguousarray(coeffs[b])
    orc.L.orc_floor0_apply_coeffs(d, 0, n, float(amps[b]), cf.ctypes.data, ref.ctypes.data, st.block1)
    r = ref[:half]
    neq = got[b].view(np.uint32) != r.view(np.uint32)
    if neq.any():
        bad_frames += 1
        idx = np.nonzero(neq)[0]
        ul = (got[b].view(np.int32)[idx].astype(np.int64) - r.view(np.int32)[idx].astype(np.int64))
        if bad_frames <= 6:
            print("frame %d: %d of %d bins differ; first bins %s; ulp diffs min %d max %d; got %s ref %s" % (
                b, idx.size, half, idx[:8].tolist(), ul.min(), ul.max(), got[b][idx[:3]], r[idx[:3]]))
print("frames with any difference: %d of %d" % (bad_frames, batch))
