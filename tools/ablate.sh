#!/bin/bash
# kernel-time ablations of the tiled minimizer kernel (results are wrong when S2K_DEBUG_SKIP is set)
for skip in 0 1 2 3 4 7; do
  S2K_DEBUG_SKIP=$skip timeout -k 10 200 python bench.py --no-cpu-baseline --verify-reads 0 --reads 500000 --steps 3 --warmup 1 > /tmp/ab.json 2>/tmp/ab.err
  python - <<PY
import json
j=json.load(open("/tmp/ab.json"))
print("skip=$skip", "hpc_kernel_ms", j["roofline"]["kernel_ms"], "reg_kernel_ms", j["other_mode"]["kernel_ms"], "km_ms", j["roofline"]["kminmer_kernel_ms"])
PY
done
