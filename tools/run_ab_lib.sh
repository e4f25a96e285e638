cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gemm_x3t_gpu.py tests/test_model_gpu.py -m gpu -q -x 2>&1 | tail -2
for i in 1 2; do for l in build/libfrcnn_knobs.so fasterrcnn_amd/csrc/libfrcnn_hip.so; do FRCNN_LIB_PATH=$l python bench.py --steps 20 --warmup 5 --no-secondary --no-extra-legs --no-cpu-baseline 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(\"$l\", d[\"value\"],d[\"parity\"][\"golden_600x1000\"][\"forward_rows_within_gate\"], d[\"parity\"][\"golden_600x1000\"][\"predict_rows_within_gate\"], d[\"roofline\"][\"per_class_ms_per_image\"])"; done; done
