#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point dan_forward (H2D of the uint8 planes + forward + D2H)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl4vc_amd.config import DanConfig
from dl4vc_amd.model import DanNet
from dl4vc_amd import synth
from dl4vc_amd.synth import random_state_dict
cfg = DanConfig(reads=64)
net = DanNet(cfg).load_state_dict(random_state_dict(cfg, seed=0))
b = synth.tile_sites(synth.make_sites(256, reads=64, seed=0), 8192)
net.forward_u8(*b.arrays())
t0 = time.perf_counter(); net.forward_u8(*b.arrays()); dt = time.perf_counter() - t0
print("host-pointer dan_forward: %d sites in %.3f s = %.1f sites/s (PCIe-inclusive, pageable host memory)" % (len(b), dt, len(b) / dt))
