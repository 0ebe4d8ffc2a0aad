"""Prints the [timing] lines of a tools/e2e_cli.py result file."""
import json
import sys

r = json.load(open(sys.argv[1]))
for k, v in r.items():
    if isinstance(v, dict):
        print(k, "%.3f s, %.2f M reads/s" % (v["seconds"], v["reads_per_s"] / 1e6))
        for t in v["timing"]:
            print("   ", t)
