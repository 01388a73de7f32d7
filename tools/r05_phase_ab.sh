# r05_phase_ab.sh -- phase stamps of k_synth (profiling builds): the tree's own against build_ab/lib_r4_dbg.so
cd $GRAFT_REPO_ROOT
for w in "" grand; do
for L in nvorbis_amd/libnvorbis_hip_dbg.so build_ab/lib_r4_dbg.so; do
  echo "== $L $w"
  NVH_LIB=$GRAFT_REPO_ROOT/$L NVH_ALLOW_STALE=1 python tools/dbg_phase_synth.py $w 2>&1 | grep -v amdgpu.ids | tail -9
done
done
