export TPNET_DEV_LIB=$GRAFT_REPO_ROOT/tpnet_amd/libtpnet_hip_dev.so
for f in 3 2 1 0; do echo "== full-step level $f"; TPNET_DEV_WIN_MED_FULL=$f python tools/short_sweep.py "158,400" windowed 2>&1 | grep -v amdgpu.ids; done
for pp in 2 1; do echo "== partial-step level $pp (full 1)"; TPNET_DEV_WIN_MED_FULL=1 TPNET_DEV_WIN_MED_PART=$pp python tools/short_sweep.py "20,158" windowed 2>&1 | grep -v amdgpu.ids; done
