#!/usr/bin/env python
"""Headline benchmark: ns/day (+ pair-interactions/s) of the 100k-atom TIP3P water box (config C3 of
BASELINE.json / SURVEY.md §8) on N MI355X, one independent replica per GPU (config C4, weak scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one MD time step of the whole hot path: first velocity-Verlet half step, bonded +
nonbonded forces (incl. the amortised cell/Verlet-list rebuilds), Langevin kick + second half step.
State is resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)



def usable_cpus():
    """CPUs this process may really use: its affinity mask capped by the cgroup's CPU quota.  (The GPU boxes show 256 hardware
    threads under a quota of 16 CPUs: torch's default of 128 worker threads exhausts the quota of a 100-ms period in ~12 ms and
    EVERY thread of the process — the one that paces the GPU included — is stopped for the rest of it: legs of a bench run that
    came out 3-4x slow, one run in ten.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), None])):
        try:
            with open(path) as fh:
                quota, period = parse(fh.read())
            if period is None:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                    period = fh.read().strip()
            if quota not in ("max", "-1") and int(quota) > 0:
                n = min(n, max(1, int(quota) // int(period)))
            break
        except (OSError, ValueError):
            continue
    return max(1, n)


USABLE_CPUS = usable_cpus()
# worker threads of the host-side libraries during the GPU legs (set before they are imported): well inside the quota
_THREADS_FROM_CALLER = "OMP_NUM_THREADS" in os.environ  # (torch.distributed.run sets 1 per rank)
_n_threads = os.environ.get("OMP_NUM_THREADS") or str(max(1, min(8, USABLE_CPUS // 2)))
for _k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_k, _n_threads)

import numpy as np  # noqa: E402
import torch  # noqa: E402

TERMS = ["lj", "electrostatics", "bonds", "angles"]
CUTOFF = 9.0
TIMESTEP_FS = 1.0
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


PMC_TRAFFIC_FILES_SWITCH = ("r06_switch_pmc_traffic.json",)  # the SWITCH variant of the launch (secondary.c3_switch)
PMC_TRAFFIC_FILES = ("r06_pmc_traffic.json", "r05_a_pmc_traffic.json", "r04_c_pmc_traffic.json", "r04_b_pmc_traffic.json", "r04_pmc_traffic.json", "r03_d_pmc_traffic.json", "r03_c_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json")  # newest committed PMC pass first
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 (vector)
FLOP_PER_PAIR = 50.0             # ~50 FLOP + 1 rsqrt per in-cutoff pair
SIMDS, NOMINAL_GHZ = 1024, 2.4   # 256 CUs x 4 SIMDs; one wave64 VALU instruction issues over 2 cycles per SIMD


TIMING_PASS_STEPS = 96  # launches of the dominant kernel timed in the pass BEHIND the timed region


def time_pair_launches(forces, integ, system, nsteps=TIMING_PASS_STEPS):
    """Average duration of the dominant kernel, measured live with HIP events attached to its dispatch
    (hipExtLaunchKernel) on the launch stream — in a pass of its own right BEHIND the timed region: one `step()` call of
    nsteps + 1 MD steps in which EVERY interior launch is timed, the first launch after a list build included (the
    call's last launch returns energies — another variant of the kernel — and is left out).  Round 5 timed four picked
    launches inside the timed region: the events cost that region ~3 % (profiles/r05_bench_order_ab.txt) and the four
    launches were the optimistic end of the distribution; this is the all-launch average the kernel trace gives."""
    forces.enable_timing(system.pos, True, every=1, interior_only=True)
    forces.read_timing(system.pos, reset=True)
    integ.step(nsteps + 1)
    ms, launches = forces.read_timing(system.pos, reset=True)
    forces.enable_timing(system.pos, False)
    return ms, launches


def pmc_traffic(switched=False):
    """Counter figures of the dominant kernel from the newest committed PMC pass (profiles/): counters cannot be
    collected inside the timed run, so what `rocprofv3 --pmc` measured for the same kernel + workload is reported
    with its provenance (file and the commit the pass was run on): HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE,
    the guide's gfx950 correction) and VALU wave-instructions per launch (SQ_INSTS_VALU)."""
    for name in (PMC_TRAFFIC_FILES_SWITCH if switched else PMC_TRAFFIC_FILES):
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as fh:
                d = json.load(fh)
            return (float(d["hbm_bytes_per_launch"]), os.path.relpath(path, ROOT), d.get("commit"),
                    d.get("valu_wave_instr_per_launch"))
        except Exception:
            continue
    return None, None, None, None


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start the N ranks ourselves, exactly as the
    driver would (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...`), and pass their exit
    code on.  Rank 0 of the children prints the one JSON line."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes)
    if not _THREADS_FROM_CALLER:  # the ranks share this process's CPU quota
        for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
            env[k] = str(max(1, min(4, USABLE_CPUS // (2 * args.gpus))))
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world):
    """Launcher / collective plumbing without a GPU (CPU test, `--backend gloo --dry`): the same process
    group set-up, barriers, max-over-ranks timing and observable gather as the real run, no MD."""
    import torch.distributed as dist

    from torchmd_amd.replicas import ReplicaFanout

    if world > 1:
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    fan = ReplicaFanout(total_replicas=world, device=torch.device("cpu"))
    fan.check_same_topology(np.arange(4))
    fan.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    fan.barrier()
    elapsed = fan.max_over_ranks(time.perf_counter() - t0)
    obs = fan.gather_observables([float(rank)], [float(-rank)], [300.0])
    if rank == 0:
        print(json.dumps({"metric": "dry run (launcher test)", "value": 0.0, "unit": "ns/day", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
                          "dry": True, "ranks_seen": [int(x) for x in obs[:, 0]]}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def dry_run_c5(args, rank, world):
    """`--config c5 --gpus N --dry --backend gloo`: the N-brick bench path without a GPU.  The same `DomainSet` as the real
    run — brick grid, ownership, halo plans, count + row exchanges through `DistTransport`, the asynchronous migration
    trigger, migrations, the saved-state machinery — on CPU tensors over gloo with the force engine stubbed out
    (`DryDomain`: zero forces, ballistic atoms; velocities x20 so that atoms cross brick faces inside a short window).
    Checked: every brick's halo equals the brute-force set of periodic images (before and after the run), and the
    gathered trajectory equals x0 + v t for every atom id (atoms that changed bricks included)."""
    import torch.distributed as dist

    from torchmd_amd.builders import lj_box
    from torchmd_amd.domain import DistTransport, DomainSet, LocalTransport
    from torchmd_amd.integrator import TIMEFACTOR
    from torchmd_amd.replicas import ReplicaFanout

    if world > 1:
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    fan = ReplicaFanout(total_replicas=world, device=torch.device("cpu"))
    nside = args.nside if args.nside != 32 else 36  # 46 656 atoms by default (--nside 100: the full 10^6-atom box)
    mol, pos, box = lj_box(nside, seed=0)
    n = mol.numAtoms
    rng = np.random.default_rng(1)  # (same numbers on every rank)
    vel = rng.normal(scale=20.0 * np.sqrt(0.001987191 * 85.0 / 39.95), size=(n, 3))
    dev, dt = torch.device("cpu"), torch.float64
    ds = DomainSet(box, world, dev, dt, ["lj"], CUTOFF, skin=args.skin or 2.5,
                   transport=DistTransport() if world > 1 else LocalTransport(1), dry=True)
    ds.scatter(pos, vel, np.zeros(n), np.zeros(n, dtype=np.int64), np.full(n, 39.95))

    def halos_ok(global_pos):
        _, w = ds.grid.owner(torch.as_tensor(global_pos, dtype=dt))
        ok = all(d.halo_matches_brute_force(w) for d in ds.domains.values())
        return bool(fan.max_over_ranks(0.0 if ok else 1.0) == 0.0)

    halo_before = halos_ok(pos)
    ds.compute_forces()
    ds.step(max(args.warmup, 1), timestep_fs=TIMESTEP_FS)
    m0 = ds.migrations
    fan.barrier()
    t0 = time.perf_counter()
    ds.step(args.steps, timestep_fs=TIMESTEP_FS)
    elapsed = time.perf_counter() - t0
    fan.barrier()
    elapsed = fan.max_over_ranks(elapsed)
    nsteps = max(args.warmup, 1) + args.steps
    expect = pos + vel * (TIMESTEP_FS / TIMEFACTOR) * nsteps
    ds.migrate()  # (a halo's membership is that of the last migration: compare right behind one)
    halo_after = halos_ok(expect)
    # every atom id exactly once over the ranks, at x0 + v t
    worst, owned = 0.0, 0
    for d in ds.domains.values():
        ids = d.ids.numpy()
        owned += len(ids)
        worst = max(worst, float(np.abs((d.pos + d.unwrap).numpy() - expect[ids]).max()) if len(ids) else 0.0)
    worst = fan.max_over_ranks(worst)
    owned_all = int(round(float(ds.transport.sum(torch.tensor([float(owned)]))[0]))) if world > 1 else owned
    doms = next(iter(ds.domains.values()))
    out = {
        "metric": "dry run of the C5 domain decomposition (no GPU, no forces)", "value": 0.0, "unit": "ns/day", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "dry": True,
        "backend": args.backend if world > 1 else "in-process", "natoms": n,
        "domains": {"grid": list(ds.grid.dims), "own_atoms_rank0": int(doms.nown), "halo_atoms_rank0": int(doms.local_pos.shape[1] - doms.nown),
                    "migrations_in_timed_region": ds.migrations - m0, "recoveries": ds.recoveries},
        "halo_equals_brute_force": [halo_before, halo_after], "atoms_accounted_for": owned_all == n,
        "max_abs_dx_vs_ballistic": worst,
    }
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ok = halo_before and halo_after and owned_all == n and worst < 1e-9
    if not ok:
        raise SystemExit("dry run of the domain decomposition FAILED: " + json.dumps(out))


C5_TRAFFIC_FILES = ("r06_c5_pmc_traffic.json", "r05_a_c5_pmc_traffic.json", "r04_c_c5_pmc_traffic.json", "r04_b_c5_pmc_traffic.json", "r04_c5_pmc_traffic.json", "r03_d_c5_pmc_traffic.json", "r03_c5_pmc_traffic.json")  # newest committed PMC pass first


def c5_single_gpu(args, device, cpu_budget_s=15.0, steps=None, warmup=None, defer_cpu=False):
    """Config C5 on ONE GPU: the whole 10^6-atom argon box on the single-domain engine.  Returns the fields of a bench
    line (value, ms_per_step, roofline, cpu_baseline, ...): `--config c5 --gpus 1` prints them as its line, the default
    run embeds them as `secondary.c5`."""
    from torchmd_amd.builders import argon_forcefield, lj_box
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    nside = args.nside if args.nside != 32 else 100
    mol, pos, box = lj_box(nside, seed=0)
    par = Parameters(argon_forcefield(mol), mol, ["lj"], precision=torch.float32)
    natoms = mol.numAtoms
    torch.manual_seed(1)
    vel0 = maxwell_boltzmann(par.masses, 85.0, 1)
    s = System(natoms, 1, torch.float32, device)
    s.set_positions(pos[:, :, None])
    s.set_box(box)
    s.set_velocities(vel0)
    f = Forces(par, terms=["lj"], cutoff=CUTOFF, **({} if args.skin is None else {"skin": args.skin}))
    f.compute(s.pos, s.box, s.forces)
    integ = Integrator(s, f, TIMESTEP_FS, device, gamma=1.0, T=85.0)
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    integ.step(max(warmup, 1))
    st0 = f.stats(s.pos)
    replays0 = integ.replays
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ekin, epot, temp = integ.step(steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    st1 = f.stats(s.pos)
    replays = integ.replays - replays0
    pair_ms, pair_launches = time_pair_launches(f, integ, s)
    pcut = f.count_pairs(s.pos, s.box)[0]
    # (as in the C3 line: the timed launches also make the MD step -> SURVEY 8(d)'s whole-step bytes)
    fused = st1["steps_in_pair_launch"] - st0["steps_in_pair_launch"] >= steps - 2
    alg_bytes = 4.0 * pcut + (132.0 if fused else 28.0) * natoms
    pair_avg_s = (pair_ms / max(pair_launches, 1)) * 1e-3
    achieved = alg_bytes / pair_avg_s / 1e9 if pair_avg_s > 0 else 0.0
    c5_traffic, c5_traffic_src = None, None
    for name in C5_TRAFFIC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                c5_traffic = float(json.load(fh)["hbm_bytes_per_launch"])
                c5_traffic_src = "profiles/" + name
            break
        except Exception:
            continue
    out = {
        "value": ns_per_day(steps, elapsed), "unit": "ns/day", "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "dtype": "f32", "natoms": natoms, "box": [float(x) for x in box],
        "pairs_in_cutoff": pcut, "pair_interactions_per_s": pcut * steps / elapsed,
        "roofline": {
            "kernel": "list_pair_fast_f32_kernel (fp32, LJ)" + (" with the MD step in the same launch (step blocks)" if fused else ""),
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": c5_traffic, "traffic_source": c5_traffic_src,
            "algorithmic_bytes_per_launch": alg_bytes,
            "algorithmic_bytes": "whole step, 4 B x pairs + 132 B x atoms" if fused else "pair part, 4 B x pairs + 28 B x atoms",
            "frac_pair_bytes_only": ((4.0 * pcut + 28.0 * natoms) / pair_avg_s / 1e9 / HBM_PEAK_GBS) if pair_avg_s > 0 else 0.0,
            "avg_kernel_us": pair_avg_s * 1e6, "launches_timed": int(pair_launches),
            "alu": {"flops_per_launch": 30.0 * pcut, "achieved_tflops": 30.0 * pcut / pair_avg_s / 1e12 if pair_avg_s > 0 else 0.0,
                    "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "note": "~30 FLOP per LJ-only pair"},
        },
        "list": {"rebuilds_in_timed_region": int(st1["n_rebuilds"] - st0["n_rebuilds"]),
                 "rebuild_chains_left_out": int(st1["chains_skipped"] - st0["chains_skipped"]),
                 "batches_rewound_and_repeated": int(replays),
                 "entries": int(st1["list_entries"]), "ncell": list(st1["ncell"]), "skin": st1["skin"]},
        "temperature_K": [float(temp[0])],
    }
    if not args.no_cpu_baseline and not defer_cpu:  # (defer_cpu: the caller runs the CPU legs behind all GPU legs)
        out["cpu_baseline"] = cpu_baseline_c5(natoms, budget_s=cpu_budget_s)
    f.close()
    del s, f, integ
    torch.cuda.empty_cache()
    return out


def c5_workload(natoms, box, world=1):
    return (f"C5 synthetic argon box: {natoms} atoms, L={box[0]:.1f} A, cutoff 9 A, LJ only, Langevin 85 K gamma 1/ps, "
            "timestep 1 fs" + ("; one brick per GPU, halo exchange over RCCL" if world > 1 else ""))


def run_c5(args, rank, world, local_rank, device, launched):
    """Config C5 of BASELINE.json: synthetic 10^6-atom Lennard-Jones (argon) box, cutoff 9 A, Langevin 85 K,
    1 fs.  One GPU: the whole box on the single-domain engine.  N > 1: spatial domain decomposition, one brick
    per rank, positions of the halo atoms exchanged over RCCL every step (torchmd_amd/domain.py); strong
    scaling (the box is fixed)."""
    import torch.distributed as dist

    from torchmd_amd.replicas import ReplicaFanout

    fan = ReplicaFanout(total_replicas=world, device=device)
    extra = {}
    if world == 1:
        r = c5_single_gpu(args, device)
        natoms, box, elapsed, pcut = r["natoms"], r["box"], r["ms_per_step"] * 1e-3 * args.steps, r["pairs_in_cutoff"]
        extra = {k: r[k] for k in ("roofline", "list", "temperature_K", "cpu_baseline") if k in r}
    else:
        from torchmd_amd.builders import argon_forcefield, lj_box
        from torchmd_amd.domain import DistTransport, DomainSet
        from torchmd_amd.integrator import maxwell_boltzmann
        from torchmd_amd.parameters import Parameters

        nside = args.nside if args.nside != 32 else 100
        mol, pos, box = lj_box(nside, seed=0)
        par = Parameters(argon_forcefield(mol), mol, ["lj"], precision=torch.float32)
        natoms = mol.numAtoms
        torch.manual_seed(1)
        vel0 = maxwell_boltzmann(par.masses, 85.0, 1)
        A, B = par.get_AB()
        ds = DomainSet(box, world, device, torch.float32, ["lj"], CUTOFF, A=A, B=B, skin=args.skin or 2.5,
                       transport=DistTransport())
        ds.scatter(pos, vel0[0].numpy(), par.charges.numpy(), par.mapped_atom_types.numpy(), par.masses.numpy().ravel())
        ds.compute_forces()
        ds.step(max(args.warmup, 1), timestep_fs=TIMESTEP_FS, gamma_ps=1.0, T=85.0, seed=3)
        m0 = ds.migrations
        fan.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ds.step(args.steps, timestep_fs=TIMESTEP_FS, gamma_ps=1.0, T=85.0, seed=3)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0  # this rank's K steps are complete; the slowest rank is the job's time
        fan.barrier()
        elapsed = fan.max_over_ranks(elapsed)
        d = next(iter(ds.domains.values()))
        info = torch.tensor([d.nown, d.local_pos.shape[1] - d.nown], dtype=torch.float64, device=device)
        allinfo = [torch.empty_like(info) for _ in range(world)]
        dist.all_gather(allinfo, info)
        extra["domains"] = {"grid": list(ds.grid.dims), "own_atoms": [int(x[0]) for x in allinfo],
                            "halo_atoms": [int(x[1]) for x in allinfo], "migrations_in_timed_region": ds.migrations - m0}
        pcut = None
        for dom in ds.domains.values():
            dom.forces_engine.close()
    out = {
        "metric": "ns/day, 1M-atom Lennard-Jones box, 9 A cutoff" + (", spatial domain decomposition" if world > 1 else ""),
        "value": ns_per_day(args.steps, elapsed), "unit": "ns/day", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": c5_workload(natoms, box, world), "natoms": natoms, "timestep_fs": TIMESTEP_FS},
        "pairs_in_cutoff": pcut,
        "pair_interactions_per_s": (pcut * args.steps / elapsed) if pcut else None,
    }
    out.update(extra)
    if world > 1:
        out["rccl"] = rccl_info(world, args.backend)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if launched:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


def cpu_baseline_c5(natoms_full, nside_sample=50, budget_s=15.0):
    """Reference arithmetic on the host cores for config C5, on a BOUNDED sample: the oracle's md_step (same torch
    CPU ops as the reference, sparse candidate pair list) on a 50^3 = 125 000-atom argon box at the same density,
    cutoff and thermostat; the cost of the reference's pair arithmetic is linear in the number of atoms at fixed
    density, so the figure for the 10^6-atom box is the sample's divided by the atom ratio (stated in `sample`)."""
    from oracle import torchmd_oracle as orc
    from torchmd_amd.builders import argon_forcefield, lj_box
    from torchmd_amd.integrator import maxwell_boltzmann
    from torchmd_amd.parameters import Parameters

    mol, pos, box = lj_box(nside_sample, seed=0)
    par = Parameters(argon_forcefield(mol), mol, ["lj"], precision=torch.float32)
    p = torch.tensor(pos, dtype=torch.float32)[None].contiguous()
    cbox = torch.zeros(1, 3, 3)
    for k in range(3):
        cbox[0, k, k] = float(box[k])
    torch.manual_seed(1)
    vel = maxwell_boltzmann(par.masses, 85.0, 1).to(torch.float32)
    frc = torch.zeros_like(p)
    masses = par.masses.to(torch.float32).view(-1, 1)
    pairs = orc.candidate_pairs(pos, box, CUTOFF + 0.6, None)
    dt, gamma, vcoeff = orc.integrator_constants(TIMESTEP_FS, 1.0, 85.0, masses)
    kw = dict(cutoff=CUTOFF, pairs=pairs)
    gpu_leg_threads = torch.get_num_threads()
    torch.set_num_threads(USABLE_CPUS)  # (all the CPUs the cgroup grants; the GPU legs are over)
    orc.md_step(par, p, vel, frc, cbox, masses, dt, ["lj"], gamma, vcoeff, **kw)  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        orc.md_step(par, p, vel, frc, cbox, masses, dt, ["lj"], gamma, vcoeff, **kw)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 50:
            break
    ratio = mol.numAtoms / float(natoms_full)
    torch.set_num_threads(gpu_leg_threads)
    return {
        "value": ns_per_day(n, el) * ratio, "unit": "ns/day", "cores": USABLE_CPUS, "kind": "port",
        "extrapolated": True, "s_per_step_sample": el / n,
        "source": "oracle (pinned port of the reference arithmetic; /root/reference does not exist on the GPU box)",
        "sample": f"{n} MD steps of a {mol.numAtoms}-atom argon box at the same density (oracle md_step, {len(pairs)} candidate "
        f"pairs, list build excluded); value = the sample's ns/day x {ratio:.4f} (atom ratio to the {natoms_full}-atom box: the "
        "reference's pair arithmetic is linear in N at fixed density)",
    }


def rccl_info(world, backend):
    """`"rccl": {...}` of every N > 1 line: what the ranks talk over (nccl backend = RCCL on ROCm), the library's version
    and who is there — evidence in the line itself that N processes with one GPU each took part."""
    import torch.distributed as dist

    try:
        ver = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001  (no RCCL in this build / CPU)
        ver = None
    info = {"ranks": world, "backend": backend, "version": ver, "is_rccl": bool(getattr(torch.version, "hip", None)) and backend == "nccl"}
    if dist.is_available() and dist.is_initialized():
        dev = torch.cuda.current_device() if torch.cuda.is_available() else -1
        # (gloo has no all_gather of device tensors: gather on the host unless the backend is RCCL)
        t = torch.tensor([float(dev)], device=f"cuda:{dev}" if (dev >= 0 and backend == "nccl") else "cpu")
        got = [torch.zeros_like(t) for _ in range(world)]
        try:
            dist.all_gather(got, t)
            info["local_device_of_rank"] = [int(x.item()) for x in got]
        except Exception as exc:  # noqa: BLE001  (the line must not be lost to its own provenance block)
            info["local_device_of_rank"] = None
            info["gather_error"] = f"{type(exc).__name__}: {exc}"[:200]
    return info


def ns_per_day(steps, seconds, timestep_fs=TIMESTEP_FS):
    return steps / seconds * timestep_fs * 1e-6 * 86400.0  # reference run.py:19,279 (FS2NS)


def build_system(nside, device, dtype, seed, skin=None, skin_weights="mass", switch_dist=None):
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import maxwell_boltzmann
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    mol, pos, box = tip3p_box(nside, seed=0)
    par = Parameters(water_forcefield(mol), mol, TERMS, precision=dtype)
    system = System(mol.numAtoms, 1, dtype, device)
    system.set_positions(pos[:, :, None])
    system.set_box(box)
    torch.manual_seed(seed)
    system.set_velocities(maxwell_boltzmann(par.masses, 300.0, 1))
    forces = Forces(par, terms=TERMS, cutoff=CUTOFF, rfa=True, skin_weights=skin_weights, switch_dist=switch_dist,
                    **({} if skin is None else {"skin": skin}))
    return mol, par, system, forces, box


def host_state(system):
    """Positions, velocities, forces and box of `system` on the host (what cpu_baseline starts from)."""
    return tuple(t.detach().cpu().clone() for t in (system.pos, system.vel, system.forces, system.box))


def cpu_baseline(par, state, box, budget_s=20.0):
    """Reference arithmetic on the host cores: the oracle's md_step (same torch CPU ops as the reference,
    sparse candidate pair list as in SURVEY.md §8(d)-(ii)) on the same relaxed box (`state` = host_state() of it).  The
    candidate-list build is excluded from the timing (favourable to the CPU).  Runs LAST in a bench run: its worker threads keep
    spinning for a while behind their last parallel region, and a GPU leg whose host thread paces one launch ahead of the
    device must not share the cores with them."""
    from oracle import torchmd_oracle as orc

    pos, vel, frc, cbox = state
    masses = par.masses.to(pos.dtype).view(-1, 1)
    pairs = orc.candidate_pairs(pos[0].double().numpy(), box, CUTOFF + 0.6, orc.exclusion_pairs(par))
    dt, gamma, vcoeff = orc.integrator_constants(TIMESTEP_FS, 0.1, 300.0, masses)
    kw = dict(cutoff=CUTOFF, rfa=True, pairs=pairs)
    orc.md_step(par, pos, vel, frc, cbox, masses, dt, TERMS, gamma, vcoeff, **kw)  # warm-up
    # the best the host can do: a gather / scatter workload is oversubscribed by torch's default of one thread per
    # hardware thread (round 5: 2.48 s/step on 128 threads, the reference's classes on 8 threads 1.39 s/step) — one timed
    # step at 8 / 16 / 32 / all USABLE CPUs (the cgroup quota: 16 on the GPU boxes), the bounded sample with the fastest
    # (never more threads than the cgroup lets run at once: beyond that they are only stopped and started)
    default_threads = torch.get_num_threads()
    tried = {}
    for nt in sorted({min(t, USABLE_CPUS) for t in (8, 16, 32, USABLE_CPUS)}):
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        orc.md_step(par, pos, vel, frc, cbox, masses, dt, TERMS, gamma, vcoeff, **kw)
        tried[nt] = time.perf_counter() - t0
    threads_used = min(tried, key=tried.get)
    torch.set_num_threads(threads_used)
    n, t0 = 0, time.perf_counter()
    while True:
        orc.md_step(par, pos, vel, frc, cbox, masses, dt, TERMS, gamma, vcoeff, **kw)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 50:
            break
    torch.set_num_threads(default_threads)
    ref = None  # the reference's OWN Forces + Integrator with a sparse pair list, timed in the build container
    try:         # (tools/ref_cpu_sparse.py; /root/reference does not exist on the GPU box): recorded beside the port
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r02_ref_cpu_sparse.json")) as fh:
            r = json.load(fh)
        ref = {"value": r["ns_per_day"], "unit": "ns/day", "s_per_step": r["s_per_step"], "cores": r["threads"],
               "kind": "reference", "where": r["host"], "source": "profiles/r02_ref_cpu_sparse.json"}
    except (OSError, KeyError, ValueError):
        pass
    return {
        "value": ns_per_day(n, el),
        "unit": "ns/day",
        "cores": threads_used,
        "threads_used": threads_used,
        "threads_tried_s_per_step": {str(k): v for k, v in tried.items()},
        "host_hardware_threads": os.cpu_count(),
        "host_usable_cpus": USABLE_CPUS,
        "kind": "port",
        "reference_in_build_container": ref,
        "source": "oracle (pinned port of the reference arithmetic; /root/reference does not exist on the GPU box)",
        "s_per_step": el / n,
        "sample": f"{n} MD steps of the same {pos.shape[1]}-atom box (oracle/torchmd_oracle.py md_step, "
        f"{len(pairs)} candidate pairs, list build excluded)",
    }


def c3_leg(forces, integ, system, natoms, steps, fan, pmc=True, switched=False):
    """One timed leg on the water box: `integ.step(steps)` between barrier + synchronize on both sides (nothing else in
    the region: no events, no read-backs), then — outside the region — the statistics, the pass that times the dominant
    kernel (time_pair_launches) and the pair count; returns the elapsed time, the observables and the `roofline` /
    `list` blocks of a bench line."""
    st0 = forces.stats(system.pos)
    replays0 = integ.replays
    fan.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ekin, epot, temp = integ.step(steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0  # this rank's K steps are complete (barrier + synchronize in front, synchronize
    fan.barrier()                       # + barrier behind); the job's time is the slowest rank's
    elapsed = fan.max_over_ranks(elapsed)
    st1 = forces.stats(system.pos)
    replays = integ.replays - replays0
    obs = fan.gather_observables(ekin, epot, temp)
    sp0 = forces.stats(system.pos)
    pair_ms, pair_launches = time_pair_launches(forces, integ, system)
    sp1 = forces.stats(system.pos)
    pcut = forces.count_pairs(system.pos, system.box)[0]
    st2 = forces.stats(system.pos)
    rebuilds = st1["n_rebuilds"] - st0["n_rebuilds"]

    # roofline of the dominant kernel.  Algorithmic bytes of the pair part per launch = 4 B per unique in-cutoff pair
    # (one int32 neighbour index) + 28 B per atom (16 B xyzq read, 12 B force write); since round 3 the timed
    # launches of an MD run also make the step (step blocks behind the pair blocks: kicks, drift, displacement test,
    # inline bonded terms), i.e. SURVEY.md 8(d)'s whole-step figure 4 B per pair + 132 B per atom.
    fused_steps = st1["steps_in_pair_launch"] - st0["steps_in_pair_launch"]
    fused = fused_steps >= steps - 2
    pair_bytes = 4.0 * pcut + 28.0 * natoms
    step_bytes = 4.0 * pcut + 132.0 * natoms  # whole step incl. integrator
    alg_bytes = step_bytes if fused else pair_bytes
    pair_avg_s = (pair_ms / max(pair_launches, 1)) * 1e-3
    achieved = alg_bytes / pair_avg_s / 1e9 if pair_avg_s > 0 else 0.0
    traffic, traffic_src, traffic_commit, valu_instr = pmc_traffic(switched) if pmc else (None, None, None, None)
    # the other two ceilings of SURVEY 8(d): fp32 vector ALU (50 FLOP per unique in-cutoff pair) and VALU issue
    # (wave-instructions of the PMC pass x 2 cycles per SIMD at the nominal clock)
    alu_tflops = FLOP_PER_PAIR * pcut / pair_avg_s / 1e12 if pair_avg_s > 0 else 0.0
    valu_issue = None
    if valu_instr and pair_avg_s > 0:
        cyc = pair_avg_s * NOMINAL_GHZ * 1e9 * SIMDS / valu_instr
        valu_issue = {"wave_instr_per_launch": valu_instr, "cycles_per_instr_per_simd": cyc,
                      "frac_of_2cyc_peak": 2.0 / cyc, "clock_ghz_assumed": NOMINAL_GHZ,
                      "clock_ghz_recorded_under_this_kernel": "2.23-2.37 by XCD (a recorded value: profiles/r05_pair_clock.txt, not measured in this run)",
                      "source": traffic_src}
    roofline = {
        "kernel": "list_pair_fast_f32_kernel<8> (lean scalar fp32, LJ + reaction field" + (", LJ switching function" if switched else "") + ")"
        + (" with the MD step in the same launch (step blocks: integrator + inline bonded terms)" if fused else ""),
        # `bound / achieved / peak / frac`: the HBM view the contract asks for (SURVEY 8(d)'s algorithmic bytes against 8 TB/s);
        # what the counters say limits the launch is in `limited_by`, `alu` and `valu_issue` are the other two ceilings
        "bound": "hbm",
        "bound_observed": "valu_issue+gather",
        "frac_is": "hbm: algorithmic bytes / launch time / 8 TB/s",
        "limited_by": "VALU issue + gather (texture-addresser) rate, not HBM: see `alu`, `valu_issue` and docs/history/round3.md",
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_source": traffic_src,
        "traffic_commit": traffic_commit,
        "algorithmic_bytes_per_launch": alg_bytes,
        "algorithmic_bytes": ("whole step, 4 B x pairs + 132 B x atoms (SURVEY 8(d)): the launch computes the pair forces AND "
                              "makes the MD step" if fused else "pair part, 4 B x pairs + 28 B x atoms"),
        "frac_pair_bytes_only": (pair_bytes / pair_avg_s / 1e9 / HBM_PEAK_GBS) if pair_avg_s > 0 else 0.0,
        "steps_made_by_the_pair_launch": int(fused_steps),
        "avg_kernel_us": pair_avg_s * 1e6,
        "launches_timed": int(pair_launches),
        "timing": (f"HIP start/stop events attached to the dispatch (hipExtLaunchKernel) of EVERY interior pair-kernel launch of a "
                   f"{TIMING_PASS_STEPS + 1}-step call right behind the timed region, on the launch stream (the first launch after a list "
                   f"build included: {sp1['n_rebuilds'] - sp0['n_rebuilds']} builds in the pass); no events inside the timed region"),
        "step_frac_of_hbm_roofline": (step_bytes / (elapsed / steps)) / 1e9 / HBM_PEAK_GBS,
        "alu": {"flops_per_launch": FLOP_PER_PAIR * pcut, "achieved_tflops": alu_tflops, "peak": FP32_VECTOR_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": alu_tflops / FP32_VECTOR_PEAK_TFLOPS},
        "valu_issue": valu_issue,
    }
    lst = {
        "algorithm": st2["algorithm"],
        "rebuilds_in_timed_region": int(rebuilds),
        "steps_per_rebuild": (steps / rebuilds) if rebuilds else None,
        "entries": int(st2["list_entries"]),
        "skin": st2["skin"],
        "rebuild_chains_left_out": int(st1["chains_skipped"] - st0["chains_skipped"]),
        "batches_rewound_and_repeated": int(replays),  # (a list that outlived its skin in a step without a chain: the call repeats its batch)
        "capacity_per_atom": int(st2["max_neighbours"]),
        "ncell": list(st2["ncell"]),
    }
    return {"elapsed": elapsed, "obs": obs, "pairs_in_cutoff": pcut, "roofline": roofline, "list": lst}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--nside", type=int, default=32, help="molecules per box edge (32 -> 98 304 atoms)")
    ap.add_argument("--relax-steps", type=int, default=1500)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary legs of the default run (switched C3, C5)")
    ap.add_argument("--skin", type=float, default=None, help="Verlet skin in A (default: the library's)")
    ap.add_argument("--skin-weights", default="mass", choices=["mass", "none"],
                    help="per-atom Verlet skins by mass (default) or one skin for every atom")
    ap.add_argument("--switch-dist", type=float, default=None,
                    help="profiling aid: run the MAIN leg with the LJ switching function from this distance (the line's "
                    "`config.workload` says so; the default line carries the switched box as `secondary.c3_switch`)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo with --dry)")
    ap.add_argument("--dry", action="store_true", help="launcher/collective plumbing only, no GPU work (CPU test)")
    ap.add_argument("--config", default="c3", choices=["c3", "c5"],
                    help="c3 (default, the headline): 98 304-atom TIP3P box, one replica per GPU; c5: 10^6-atom LJ box, "
                    "spatial domain decomposition for --gpus > 1")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args))

    from torchmd_amd.integrator import Integrator
    from torchmd_amd.replicas import ReplicaFanout, env_rank_world

    rank, world, local_rank = env_rank_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry:
        return dry_run_c5(args, rank, world) if args.config == "c5" else dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (the hot path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ  # torch.distributed.run / torchrun
    if launched:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(args.backend, rank=rank, world_size=world, device_id=device)  # nccl = RCCL on ROCm
    if args.config == "c5":
        return run_c5(args, rank, world, local_rank, device, launched)
    fan = ReplicaFanout(total_replicas=world, device=device)

    dtype = torch.float32
    mol, par, system, forces, box = build_system(args.nside, device, dtype, seed=1 + rank, skin=args.skin,
                                                 skin_weights=None if args.skin_weights == "none" else "mass",
                                                 switch_dist=args.switch_dist)
    fan.check_same_topology(mol.bonds, mol.angles, mol.charge)
    natoms = mol.numAtoms

    # prime forces, relax the lattice start with strong friction, then production
    forces.compute(system.pos, system.box, system.forces)
    if args.relax_steps:
        Integrator(system, forces, TIMESTEP_FS, device, gamma=10.0, T=300.0).step(args.relax_steps)
    integ = Integrator(system, forces, TIMESTEP_FS, device, gamma=0.1, T=300.0)
    if args.warmup:
        integ.step(args.warmup)

    leg = c3_leg(forces, integ, system, natoms, args.steps, fan, pmc=args.nside == 32, switched=args.switch_dist is not None)
    elapsed, obs, pcut = leg["elapsed"], leg["obs"], leg["pairs_in_cutoff"]
    steps_per_s = args.steps / elapsed
    value = ns_per_day(args.steps, elapsed) * world

    out = {
        "metric": "ns/day (aggregate over replicas), 100k-atom TIP3P water box, 9 A cutoff + reaction field",
        "value": value,
        "unit": "ns/day",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"C3/C4 synthetic TIP3P water box: {natoms} atoms, L={box[0]:.3f} A, cutoff 9 A, "
            "reaction field, terms lj+electrostatics+bonds+angles (flexible water), Langevin 300 K "
            "gamma 0.1/ps, timestep 1 fs; one independent replica per GPU"
            + (f"; LJ switching function from {args.switch_dist} A (--switch-dist: NOT the headline configuration)" if args.switch_dist else ""),
            "natoms": natoms,
            "replicas": world,
            "timestep_fs": TIMESTEP_FS,
            "relax_steps": args.relax_steps,
        },
        "ns_per_day_per_replica": ns_per_day(args.steps, elapsed),
        "pair_interactions_per_s": pcut * steps_per_s * world,
        "pairs_in_cutoff": pcut,
        "temperature_K": [float(x) for x in obs[:, 2]],
        "epot_kcal_mol": [float(x) for x in obs[:, 1]],
        "list": dict(leg["list"], skin_weights=args.skin_weights),
        "roofline": leg["roofline"],
    }
    if world > 1:
        out["rccl"] = rccl_info(world, args.backend)
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    cpu_state = host_state(system) if want_cpu else None  # (the CPU legs run behind every GPU leg: see cpu_baseline)
    secondary = {}
    if rank == 0 and world == 1 and not args.no_secondary and args.switch_dist is None:
        # Secondary leg on the SAME box and state: the reference's production settings switch the LJ term from 7.5 A
        # (tests/prod_alanine_dipeptide_amber/conf.yaml:8-9, forces.py:399-413) — the SWITCH variant of the dominant launch.
        # max(--steps, 200) steps so that the window holds the list rebuilds in their steady proportion.
        try:
            from torchmd_amd.forces import Forces

            fsw = Forces(par, terms=TERMS, cutoff=CUTOFF, rfa=True, switch_dist=7.5,
                         skin_weights=None if args.skin_weights == "none" else "mass", **({} if args.skin is None else {"skin": args.skin}))
            fsw.compute(system.pos, system.box, system.forces)
            isw = Integrator(system, fsw, TIMESTEP_FS, device, gamma=0.1, T=300.0)
            # (a fresh context behind ~0.3 s of host-side set-up: 500 untimed steps = 35 ms bring the clocks back up — with 50 the
            # launch of this leg averaged 53.4 us against 50.9 in a long run)
            sw_warm = max(args.warmup, 500)
            isw.step(sw_warm)
            sw_steps = max(args.steps, 200)
            lsw = c3_leg(fsw, isw, system, natoms, sw_steps, fan, pmc=args.nside == 32, switched=True)
            secondary["c3_switch"] = {
                "metric": "ns/day, the same water box with the LJ switching function from 7.5 A (the reference's production settings)",
                "value": ns_per_day(sw_steps, lsw["elapsed"]), "unit": "ns/day", "steps": sw_steps, "warmup": sw_warm,
                "ms_per_step": lsw["elapsed"] / sw_steps * 1e3, "dtype": "f32",
                "config": {"workload": "C3 box, terms and thermostat as the headline + switch_dist 7.5 A (explicit-force flavour of "
                           "forces.py:410-412)", "natoms": natoms, "timestep_fs": TIMESTEP_FS},
                "pairs_in_cutoff": lsw["pairs_in_cutoff"],
                "pair_interactions_per_s": lsw["pairs_in_cutoff"] * sw_steps / lsw["elapsed"],
                "temperature_K": [float(x) for x in lsw["obs"][:, 2]], "epot_kcal_mol": [float(x) for x in lsw["obs"][:, 1]],
                "list": lsw["list"], "roofline": lsw["roofline"],
            }
            fsw.close()
            del fsw, isw
        except Exception as exc:  # noqa: BLE001
            secondary["c3_switch"] = {"error": f"{type(exc).__name__}: {exc}"[:500]}
    forces.close()
    if rank == 0 and world == 1 and args.nside == 32 and not args.no_secondary:
        # Secondary configuration in the same line (headline keys untouched): BASELINE.json's config 5 on this one GPU —
        # the 10^6-atom argon box on the single-domain engine, max(--steps, 200) steps.  A failure is recorded, not raised.
        del system, forces, integ
        torch.cuda.empty_cache()
        try:
            # (at least 200 steps: the list of this box is rebuilt every ~80 steps, a 20-step window would hold none)
            c5 = c5_single_gpu(args, device, cpu_budget_s=8.0, steps=max(args.steps, 200), warmup=max(args.warmup, 50), defer_cpu=True)
            c5["metric"] = "ns/day, 1M-atom Lennard-Jones box, 9 A cutoff, 1 GPU (the cpu_baseline is extrapolated from a 125k-atom sample)"
            c5["config"] = {"workload": c5_workload(c5["natoms"], c5["box"]), "natoms": c5["natoms"], "timestep_fs": TIMESTEP_FS}
            secondary["c5"] = c5
        except Exception as exc:  # noqa: BLE001
            secondary["c5"] = {"error": f"{type(exc).__name__}: {exc}"[:500]}
    if want_cpu:
        out["cpu_baseline"] = cb = cpu_baseline(par, cpu_state, box, budget_s=15.0)
        # quoted against the FASTER of the two CPU figures (the port timed here, the reference's own classes in the build container)
        ref = cb.get("reference_in_build_container") or {}
        best_cpu = max(cb["value"], ref.get("value") or 0.0)
        out["speedup_vs_cpu_baseline"] = out["ns_per_day_per_replica"] / best_cpu
        out["speedup_is_against"] = "port" if best_cpu == cb["value"] else "reference_in_build_container"
        if isinstance(secondary.get("c5"), dict) and "natoms" in secondary["c5"]:
            secondary["c5"]["cpu_baseline"] = cpu_baseline_c5(secondary["c5"]["natoms"], budget_s=8.0)
    if secondary:
        out["secondary"] = secondary
    if rank == 0:
        print(json.dumps(out), flush=True)
    if launched:
        import torch.distributed as dist

        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
