#!/usr/bin/env python
"""Runs ON THE GPU BOX: bench.dense_1m_probe (one view of 10^6 Gaussians at 4096x3008: per-kernel times, ms per view, the texture
loop's iteration) under the current environment.   usage: python tools/dense_ab.py <label>"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
d = bench.dense_1m_probe(torch.device("cuda"))
print(sys.argv[1] if len(sys.argv) > 1 else "", d["ms_per_view"], d["kernels_us"], "tex_it", d["texture_iteration"].get("ms_per_iteration"), "pairs", d["pairs"], "longest", d["longest_tile_list"])
