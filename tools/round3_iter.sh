#!/bin/bash
# round3_iter.sh TAG -- development iteration: GPU suite, k_synth phase stamps (profiling build), a short bench line
cd $GRAFT_REPO_ROOT
TAG=${1:-it}
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/${TAG}_pytest.log 2>&1
tail -4 gpurun_out/${TAG}_pytest.log
NVH_LIB=nvorbis_amd/libnvorbis_hip_dbg.so python tools/dbg_phase_synth.py > gpurun_out/${TAG}_phase.txt 2>&1
NVH_LIB=nvorbis_amd/libnvorbis_hip_dbg.so python tools/dbg_phase_synth.py grand >> gpurun_out/${TAG}_phase.txt 2>&1
grep -v amdgpu.ids gpurun_out/${TAG}_phase.txt
python bench.py --no-cpu-baseline --min-timed-ms 600 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench.json"))
print("value %.1f M frames/s  ms_per_pass %.4f  kernels %s  frac %.3f  l3 %s" % (d["value"] / 1e6, d["config"]["ms_per_pass"], d["kernels_ms"], d["roofline"]["frac"], {k: d["roofline"]["l3_resident"][k] for k in ("frames_per_s", "kernels_ms")}))
for k, v in d.get("configs", {}).items():
    print(k, "%.1f M frames/s" % (v["frames_per_s_kernel_only"] / 1e6), v["kernels_us"])
PY
