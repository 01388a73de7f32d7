import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, nvorbis_amd as nv
from tests import synth_stream as ss, oracle_py
orc = oracle_py.load(); ctx = nv.Context(0)
for name in ["floor0_stereo", "two_submaps", "stereo_res1_coupled"]:
    pk, gr, fl = ss.filtered_stream(orc, name, 300, 3, True)
    for clip in (True, False):
        st = nv.Stream(ctx, pk[0], pk[1], pk[2]); st.set_clip(clip)
        audio = pk[3:]
        st.push_packet(audio[0], -1, 0); st.synth_host()
        k = 0
        while st.pending()[0] < 4096:
            st.push_packet(audio[1 + k % (len(audio) - 1)], -1, 0); k += 1
        b = st.upload_batch()
        pcm = torch.empty(max(b.samples * st.channels, 1), dtype=torch.float32, device="cuda")
        tot, km = b.time(pcm.data_ptr(), pcm.numel(), 30)
        print(name, "clip", clip, {n: round(v * 1e3, 1) for n, v in zip(b.kernels(), km) if n != "-"})
        b.free(); st.close()
