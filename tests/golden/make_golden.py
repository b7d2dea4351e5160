#!/usr/bin/env python3
"""Regenerates tests/golden/*.json (run from the repo root: python tests/golden/make_golden.py)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden_impl as g  # noqa: E402

for name, fn in (("sumcheck_small.json", g.sumcheck_small), ("spartan_small.json", g.spartan_small), ("nifs_small.json", g.nifs_small),
                 ("neutronnova_small.json", g.neutronnova_small), ("neutronnova_rest.json", g.neutronnova_rest)):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(fn(), f, indent=1)
    print("wrote", name)
