# tools/exp_slot_priorities.sh -- HIP stream priorities per in-flight slot (FRCNN_SLOT_PRIORITIES), driver's form and steady state, 4 slots
# (the knob -- runtime.Slot creating its stream with the k-th priority of the list -- was removed after this measurement: priorities cost 1-5 %, profiles/r06/exp_slot_priorities.txt)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
run() { python bench.py --no-extra-legs --no-cpu-baseline --no-secondary --map-images 0 --roofline-images 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['timed_bursts']['min_ms'])"; }
for rep in 1 2; do
for pr in "0" "-1,0,0,0" "-1,0,0,1" "-1,-1,0,0" "-2,-1,0,1"; do
  echo "priorities $pr driver form if4: $(FRCNN_SLOT_PRIORITIES=$pr run --steps 20 --warmup 5 --inflight 4)"
done; done
for pr in "0" "-1,0,0,0" "-1,0,0,1"; do
  echo "priorities $pr steady if4: $(FRCNN_SLOT_PRIORITIES=$pr run --inflight 4)"
  echo "priorities $pr steady if3: $(FRCNN_SLOT_PRIORITIES=$pr run --inflight 3)"
done
