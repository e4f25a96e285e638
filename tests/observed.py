"""
tests/observed.py -- the committed snapshot of DETERMINISTIC measured counts (ADVICE r4 / VERDICT r4 "do this" 3).

The spec gates of the golden tests are fractions derived from the held-out sweep (a row floor, a float32-noise bound); the kernels are
deterministic, so a run's COUNTS reproduce to the row, and tests/golden/observed_counts.json holds the counts of the last measured run:

    observed.check("vgg16_600x1000_s0/forward", {"rows_within_1e-3": 300})

fails when a count falls more than `slack` below the snapshot -- a regression that stays inside the spec floor is still caught.  A count
ABOVE the snapshot passes (and is reported: re-record).  Recording: FRCNN_RECORD_OBSERVED=1 python -m pytest tests -m gpu writes
gpurun_out/observed_counts.json (merge it into tests/golden/observed_counts.json and commit); in that mode nothing is asserted here.
"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SNAPSHOT = os.path.join(ROOT, "tests", "golden", "observed_counts.json")
RECORD = os.path.join(ROOT, "gpurun_out", "observed_counts.json")


def _load(path):
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f)


def check(key, counts, slack=0):
    counts = {k: int(v) for k, v in counts.items()}
    if os.environ.get("FRCNN_RECORD_OBSERVED") == "1":
        rec = _load(RECORD)
        rec[key] = counts
        os.makedirs(os.path.dirname(RECORD), exist_ok=True)
        with open(RECORD, "w") as f:
            json.dump(rec, f, indent=1, sort_keys=True)
        return
    snap = _load(SNAPSHOT).get(key)
    assert snap is not None, "no committed snapshot for %r in %s (record one: FRCNN_RECORD_OBSERVED=1)" % (key, SNAPSHOT)
    bad = {k: (v, snap[k]) for k, v in counts.items() if k in snap and v < snap[k] - slack}
    missing = [k for k in counts if k not in snap]
    assert not bad and not missing, "observed counts fell below the committed snapshot %s: %s (missing keys %s)" % (key, bad, missing)
    better = {k: (v, snap[k]) for k, v in counts.items() if v > snap[k]}
    if better:
        print("observed counts ABOVE the snapshot for %s (re-record): %s" % (key, better))
