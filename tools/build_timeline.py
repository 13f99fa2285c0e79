#!/usr/bin/env python
"""Per-block timeline of the Verlet-list build on the relaxed C3 box (needs a GPU; TMDHIP_DEBUG_TIMELINE=1 is set here):
wave durations, and — per group of blocks that share a cycle counter — how many waves are alive over the launch."""
import ctypes as C
import os
import sys

os.environ["TMDHIP_DEBUG_TIMELINE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import build_system  # noqa: E402
from torchmd_amd import _lib as L  # noqa: E402
from torchmd_amd.integrator import Integrator  # noqa: E402

dev = torch.device("cuda:0")
mol, par, system, forces, box = build_system(32, dev, torch.float32, seed=1)
forces.compute(system.pos, system.box, system.forces)
Integrator(system, forces, 1.0, dev, gamma=10.0, T=300.0).step(600)
Integrator(system, forces, 1.0, dev, gamma=0.1, T=300.0).step(100)  # ends with device-side rebuilds in the MD loop
lib = L.load()


buf = np.zeros(4 * 20000, dtype=np.uint64)
nb = lib.tmdhip_debug_build_timeline(buf.ctypes.data, buf.nbytes)
print("blocks", nb)
r = buf[: 4 * nb].reshape(nb, 4)
t0, t1 = r[:, 0].astype(np.int64), r[:, 1].astype(np.int64)
xcc, hwid = (r[:, 2] & np.uint64(0xFF)).astype(np.int64), (r[:, 2] >> np.uint64(8)).astype(np.int64)
nmax, work = (r[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64), (r[:, 3] >> np.uint64(32)).astype(np.int64)
# static balance: which SIMD a block ran on (all blocks of a C3 build are resident from the start) and what it had to do
simd = xcc * 4096 + ((hwid >> 13) & 7) * 512 + ((hwid >> 12) & 1) * 256 + ((hwid >> 8) & 15) * 16 + ((hwid >> 4) & 3)
keys, inv = np.unique(simd, return_inverse=True)
wsum = np.bincount(inv, weights=work.astype(np.float64))
nblk = np.bincount(inv)
print(f"SIMDs used {len(keys)}; blocks per SIMD min {nblk.min()} mean {nblk.mean():.2f} max {nblk.max()}")
print(f"work (candidates x atoms) per block: mean {work.mean():.0f} sd {work.std():.0f} max {work.max()}")
print(f"work per SIMD: mean {wsum.mean():.0f} sd {wsum.std():.0f} max {wsum.max():.0f}  -> max / mean = {wsum.max() / wsum.mean():.3f}")
# (every block is resident from the start, so a block's duration is its exit time; the cycle counters of different
# shader engines are not synchronised, durations are)
last = np.zeros(len(keys))
np.maximum.at(last, inv, (t1 - t0).astype(np.float64))
span = float((t1 - t0).max())
print(f"per-SIMD time of the last exit / launch span: mean {last.mean() / span:.3f} p5 {np.percentile(last, 5) / span:.3f} p95 {np.percentile(last, 95) / span:.3f}")
print("corr(per-SIMD work, per-SIMD last exit)", np.corrcoef(wsum, last)[0, 1])
order = np.argsort(wsum)
q = len(order) // 4
print("last exit / span by quartile of per-SIMD work:", [round(float(last[order[k * q:(k + 1) * q]].mean() / span), 3) for k in range(4)])
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "build_timeline_raw.npy"), r)
cu = simd // 4
ck, cinv = np.unique(cu, return_inverse=True)
cw = np.bincount(cinv, weights=work.astype(np.float64))
print(f"work per CU: max / mean = {cw.max() / cw.mean():.3f} over {len(ck)} CUs")
dur = t1 - t0
print(f"block duration cycles: mean {dur.mean():.0f} p5 {np.percentile(dur, 5):.0f} p50 {np.median(dur):.0f} p95 {np.percentile(dur, 95):.0f} max {dur.max()}")
print("corr(duration, longest list of the cell)", np.corrcoef(dur, nmax)[0, 1])
print("block duration by XCD (ticks of 10 ns): mean", [int(dur[xcc == x].mean()) for x in range(8)], " p95",
      [int(np.percentile(dur[xcc == x], 95)) for x in range(8)], " max", [int(dur[xcc == x].max()) for x in range(8)])
print("last exit by XCD relative to the first entry of the launch (10 ns):", [int(t1[xcc == x].max() - t0.min()) for x in range(8)],
      " first entry:", [int(t0[xcc == x].min() - t0.min()) for x in range(8)])
print("block duration by wave slot (10 ns):", [(int(w), int((hwid & 15 == w).sum()), int(dur[hwid & 15 == w].mean())) for w in sorted(set(hwid & 15))])
print("corr(duration, work)", np.corrcoef(dur, work)[0, 1], " corr(duration, block index)", np.corrcoef(dur, np.arange(nb))[0, 1])
for x in range(8):
    m = xcc == x
    tend = t1[m].max()
    rec = m & (t0 > tend - 600000)
    s, e = t0[rec] - t0[rec].min(), t1[rec] - t0[rec].min()
    T = e.max()
    grid = np.linspace(0, T, 21)
    print(f"xcc {x}: {m.sum()} blocks, {rec.sum()} on the last block's clock; span {T} cycles; alive over time:",
          [int(((s <= b) & (e > b)).sum()) for b in grid])
    print("    entry-time percentiles / span:", [round(float(np.percentile(s, q)) / T, 2) for q in (5, 25, 50, 75, 95, 100)])
