#!/bin/bash
# round-4 GPU call Z: config 5 (SDXL + colour guidance + background blend) on the final tree
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 150 python bench.py --config 5 --steps 12 --warmup 2 --no-cpu-baseline > gpurun_out/r4z_cfg5.json 2> gpurun_out/r4z_cfg5.err; echo "cfg5 rc=$?"
grep '^{' gpurun_out/r4z_cfg5.json | tail -1 | cut -c1-600
