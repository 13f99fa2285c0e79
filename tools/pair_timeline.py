#!/usr/bin/env python
"""Per-block timeline of the dominant pair launch by XCD: wall time on the device-wide 100 MHz clock and core cycles, i.e. the
shader clock each XCD really runs at under this kernel (needs a GPU and an experiment build:
    python -m torchmd_amd._build -DTMD_PAIR_TIMELINE --out=torchmd_amd/lib/exp/libtmdhip_ptl.so
    TMDHIP_LIB=$PWD/torchmd_amd/lib/exp/libtmdhip_ptl.so python tools/pair_timeline.py)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import build_system  # noqa: E402
from torchmd_amd.integrator import Integrator  # noqa: E402

dev = torch.device("cuda:0")
mol, par, system, forces, box = build_system(32, dev, torch.float32, seed=1)
forces.compute(system.pos, system.box, system.forces)
Integrator(system, forces, 1.0, dev, gamma=10.0, T=300.0).step(600)
integ = Integrator(system, forces, 1.0, dev, gamma=0.1, T=300.0)
lib = C.CDLL(os.environ["TMDHIP_LIB"])
for trial in range(3):
    integ.step(2000 + trial)  # (a long run in front: clocks and temperature as in the benchmark)
    buf = np.zeros(4 * 65536, dtype=np.uint64)
    assert lib.tmdhip_debug_pair_timeline(C.c_void_p(buf.ctypes.data), C.c_size_t(buf.nbytes)) == 0
    r = buf.reshape(-1, 4)
    grid = int(r[0, 2] >> np.uint64(32))
    r = r[:grid]
    r = r[r[:, 1] > 0]
    t0, t1, cyc = r[:, 0].astype(np.int64), r[:, 1].astype(np.int64), r[:, 3].astype(np.int64)
    xcc = (r[:, 2] & np.uint64(0xFF)).astype(np.int64)
    base = t0.min()
    print(f"last pair launch of the call: {grid} blocks ({len(r)} with work), span {(t1.max() - base) / 100:.2f} us")
    print("  last end by XCD (us)      ", [round(float(t1[xcc == x].max() - base) / 100, 1) for x in range(8)])
    print("  mean block duration (us)  ", [round(float((t1 - t0)[xcc == x].mean()) / 100, 2) for x in range(8)])
    print("  shader clock by XCD (GHz) ", [round(float(cyc[xcc == x].sum()) / float((t1 - t0)[xcc == x].sum()) / 10, 3) for x in range(8)])
