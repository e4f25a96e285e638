cd $GRAFT_REPO_ROOT
python tools/exp_fc1_ablate.py 2>&1 | grep "us per call"
for f in 1 2 4 16 3; do FRCNN_LIB_PATH=build/libfrcnn_hxa$f.so python tools/exp_fc1_ablate.py 2>&1 | grep "us per call"; done
