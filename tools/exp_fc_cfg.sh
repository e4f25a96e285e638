cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export FRCNN_LIB_PATH=build/libfrcnn_knobs.so
for cfg in 0 1; do for ns in 1 2; do echo "== FRCNN_HX_CFG=$cfg NSUB=$ns"; FRCNN_HX_CFG=$cfg FRCNN_HX_NSUB=$ns python tools/x3t_bench.py --only fc 2>&1 | grep -v amdgpu | cut -c1-200; done; done
