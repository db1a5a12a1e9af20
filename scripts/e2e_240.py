"""The e2e_detex_240 leg of bench.py alone (8 stations x 30 days through SubSpace.detex, day units):
    python scripts/e2e_240.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = sys.argv[:1]
import bench  # noqa: E402

print(json.dumps(bench.e2e_detex_240(0), indent=1))
