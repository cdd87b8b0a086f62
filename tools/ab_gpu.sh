#!/bin/bash
# A/B of the two interior MFMA kernels inside one GPU call: parity, stamps, steady-state bench (ws = y-buffer kernel, ab = two matrix sets)
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -4
for k in ab ws ab ws; do
  echo "== DD_MFMA_KERNEL=$k"
  DD_MFMA_KERNEL=$k python bench.py --no-cpu-baseline --no-side | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['extra']['steady_check'], d['extra']['cold_ms_per_step'])"
done
for k in ab ws; do
  echo "== stamps DD_MFMA_KERNEL=$k"
  DD_MFMA_KERNEL=$k DD_STAMPS=300 python bench.py --no-cpu-baseline --no-side --steps 3 --warmup 1 2>&1 | grep -i "stamps" | head -24
done
