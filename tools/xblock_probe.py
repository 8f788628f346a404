"""Times rt_op_cross_attn_block at the two SDXL shapes (bench.py's cross_attention_block leg alone) - GPU box."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    out = bench.cross_attention_block("cuda:0")
    print(json.dumps(out, indent=1))
