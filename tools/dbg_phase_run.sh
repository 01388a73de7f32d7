cd $GRAFT_REPO_ROOT
python -m nvorbis_amd.build --debug > /dev/null 2>&1
for w in 6 4; do echo "== waves $w"; NVH_LIB=nvorbis_amd/libnvorbis_hip_dbg.so python tools/dbg_phase_run.py $w 2>&1 | grep -v amdgpu.ids; done
