#!/usr/bin/env python3
"""The roofline's denominator with its clock: lamd_debug_mul32_peak (dependency-free v_mad_u64_u32 on every SIMD) at 3 and 8 waves per SIMD in launches of
>= 4 ms and of ~0.3 ms.  Run it plain for the rates, and under `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE` for the shader clock of every launch
(GRBM_GUI_ACTIVE / 8 XCDs / duration; tools/mul32_peak_clock.py reads the two CSVs)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("LAMD_LANES", "1")
from lightning_amd import Engine

eng = Engine(0)
for waves, min_ms in ((3, 4.0), (8, 4.0), (3, 20.0), (8, 20.0), (8, 0.0), (3, 0.0)):
    r, ms, ck = eng.mul32_peak(waves, min_ms, 5)
    print("waves/SIMD %d  launch %.3f ms  %.2f T mul32/s  s_memtime/s_memrealtime %.2f (x 100 MHz)" % (waves, ms, r / 1e12, ck))
eng.close()
