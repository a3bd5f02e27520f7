"""Merge a NRNERF_PIN_RECORD file (JSON lines written by tests.helpers.check_pinned on a GPU box) into
tests/golden/pinned_fp32.json:   python tools/update_pins.py gpurun_out/pins.jsonl"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN_FILE = os.path.join(REPO, "tests", "golden", "pinned_fp32.json")

pins = json.load(open(PIN_FILE)) if os.path.exists(PIN_FILE) else {}
for line in open(sys.argv[1]):
    r = json.loads(line)
    pins[r["case"]] = r["measured"]
json.dump(pins, open(PIN_FILE, "w"), indent=1, sort_keys=True)
print(f"{len(pins)} cases in {PIN_FILE}")
