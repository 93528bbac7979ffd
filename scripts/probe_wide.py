"""256-column tiles of the 8-wave MFMA kernel (tuning[2] = 16 + rows / 32) against the planner's choice and K-slice counts on
the shapes between decode and prefill.  Run on the MI355X: gpurun -- 'bash scripts/gpu.sh probe:probe_wide.py'"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0], "none"]  # probe_mma's own sweep list: nothing
import torch
from probe_mma import run

bf = torch.bfloat16
run("cfgB bf16", 8192, 8192, 4, 256, bf, [(0, 0, 0, 0), (0, 4, 20, 0), (0, 3, 20, 0), (0, 2, 20, 0), (0, 8, 24, 0), (0, 4, 24, 0)], nl=8)
run("cfgA bf16", 4096, 4096, 4, 256, bf, [(0, 0, 0, 0), (0, 4, 20, 0), (0, 8, 20, 0), (0, 2, 20, 0)], nl=32)
run("4096 M=1024", 4096, 4096, 4, 1024, bf, [(0, 0, 0, 0), (0, 1, 24, 0), (0, 2, 24, 0), (0, 1, 20, 0), (0, 2, 20, 0)], nl=16)
run("4096 M=2048", 4096, 4096, 4, 2048, bf, [(0, 0, 0, 0), (0, 1, 24, 0), (0, 2, 24, 0), (0, 1, 20, 0)], nl=16)
run("8192 M=1024", 8192, 8192, 4, 1024, bf, [(0, 0, 0, 0), (0, 1, 24, 0), (0, 2, 24, 0), (0, 1, 20, 0)], nl=8)
run("8192 M=512", 8192, 8192, 4, 512, bf, [(0, 0, 0, 0), (0, 1, 24, 0), (0, 2, 24, 0), (0, 4, 24, 0), (0, 1, 20, 0), (0, 2, 20, 0)], nl=8)
run("8192 M=2048", 8192, 8192, 4, 2048, bf, [(0, 0, 0, 0), (0, 0, 8, 0), (0, 1, 24, 0)], nl=8)
