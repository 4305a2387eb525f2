"""Reads bench.py output on stdin and prints the step time, split, transport, kernel time and host times of every JSON
line on one line each, prefixed by argv[1] (development aid for A/B runs inside one gpurun call)."""
import json,sys
if sys.stdin.isatty() or len(sys.argv) != 2 or sys.argv[1].endswith(".json"):
    sys.exit("usage: python bench.py | python tools/benchline.py LABEL   (reads stdin; a file argument is not read)")
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line)
        print(sys.argv[1], "value", round(d["value"]), "ms/step", round(d["ms_per_step"],4), "split", d["config"]["element_split"], d["config"]["parallelism"][-40:], "kernel_ms", round(d["roofline"]["kernel_ms_mean"],4), d["host_us_per_step"])
