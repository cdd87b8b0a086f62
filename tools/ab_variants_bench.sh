#!/bin/bash
# ms per headline step for the product library and every build/variants/lib_N.so, one gpurun call (tools/each_variant.sh)
tools/each_variant.sh bash -c 'python bench.py --no-cpu-baseline --no-side --steady-ms 300 --steps 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(\"   \", d[\"config\"][\"kernel\"], \"kernel_ms\", d[\"roofline\"][\"kernel_ms_events\"], \"steady\", d[\"extra\"][\"steady_check\"][\"kernel_ms\"])"'
