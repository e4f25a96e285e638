# tools/run_ab_lib.sh LIB_A LIB_B [pytest args] -- same-box A/B of two builds of the library: x3 tests on B, then the layer bench and the headline (driver's form) alternating
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A=$1; B=$2
FRCNN_LIB_PATH=$B timeout 900 python -m pytest tests/test_gemm_x3t_gpu.py -m gpu -q -x 2>&1 | tail -2
for i in 1 2; do for l in $A $B; do echo "x3f us $l: $(FRCNN_LIB_PATH=$l timeout 300 python tools/x3f_bench.py 2>&1 | grep -v amdgpu.ids | sed -e 's/.*one-launch, channel maxima given: four//' -e 's/ (.*//' | tr '\n' ' ')"; done; done
for i in 1 2 3; do for l in $A $B; do FRCNN_LIB_PATH=$l python bench.py --steps 20 --warmup 5 --no-secondary --no-extra-legs --no-cpu-baseline 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(\"bench $l\", d[\"value\"],d[\"parity\"][\"golden_600x1000\"][\"forward_rows_within_gate\"], d[\"parity\"][\"golden_600x1000\"][\"predict_rows_within_gate\"])"; done; done
