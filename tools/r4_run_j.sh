#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -k "sdxl_vae_config5" -s 2>&1 | grep -E "passes|single pass|passed|failed|rror"
