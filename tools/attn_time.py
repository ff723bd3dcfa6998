#!/usr/bin/env python3
"""Stripe-attention forward / backward device times for the four stage shapes (bench.py's attention_roofline)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cswin_unet_amd.config import get_config
cfg = get_config(os.path.join(bench.ROOT, "configs", "cswin_tiny_224_lite.yaml"))
r = bench.attention_roofline(int(os.environ.get("BATCH", "24")), cfg.MODEL.CSWIN, cfg.DATA.IMG_SIZE)
for row in r["per_stage"]:
    print(row)
print({k: r[k] for k in ("achieved", "frac", "time_per_step_ms", "hbm_frac_algorithmic")})
