"""ctypes binding of `libtmdhip.so` (C ABI declared in include/tmdhip.h).

The library is the product; there is no fallback.  If it cannot be loaded every entry point raises
`RuntimeError` with build instructions.
"""

from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
# TMDHIP_LIB: developer knob for A/B runs of differently built libraries (kernel experiments)
LIBPATH = os.environ.get("TMDHIP_LIB") or os.path.join(PKG, "lib", "libtmdhip.so")

ABI_VERSION = 8
F32, F64 = 0, 1
TERM_LJ, TERM_ELECTROSTATICS, TERM_REPULSION, TERM_REPULSIONCG = 1, 2, 4, 8
E_LJ, E_ELECTROSTATICS, E_REPULSION, E_REPULSIONCG, E_BONDS, E_ANGLES, E_DIHEDRALS, E_IMPROPERS = range(8)
NENERGY = 8
WANT_ENERGY, WANT_FORCES, COUNT_PAIRS, OVERWRITE_FORCES = 1, 2, 4, 8
ALL_REPLICAS = -1  # TMDHIP_ALL_REPLICAS
DD_OVERRUN = 2  # TMDHIP_DD_OVERRUN: tmdhip_dd_run measured a displacement beyond the halo's half skin
ALGO_AUTO, ALGO_ALLPAIRS, ALGO_CELLLIST = 0, 1, 2
SWITCH_REFERENCE, SWITCH_EXACT = 0, 1
OBSERVE_AFTER_RUN = 1  # TMDHIP_OBSERVE_AFTER_RUN

ENERGY_SLOT = {
    "lj": E_LJ,
    "electrostatics": E_ELECTROSTATICS,
    "repulsion": E_REPULSION,
    "repulsioncg": E_REPULSIONCG,
    "bonds": E_BONDS,
    "angles": E_ANGLES,
    "dihedrals": E_DIHEDRALS,
    "impropers": E_IMPROPERS,
}
TERM_BIT = {
    "lj": TERM_LJ,
    "electrostatics": TERM_ELECTROSTATICS,
    "repulsion": TERM_REPULSION,
    "repulsioncg": TERM_REPULSIONCG,
}


class NonbondedDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("dtype", C.c_int32),
        ("natoms", C.c_int32),
        ("ntypes", C.c_int32),
        ("nreplicas", C.c_int32),
        ("device", C.c_int32),
        ("types_host", C.c_void_p),
        ("charges_host", C.c_void_p),
        ("lj_A_host", C.c_void_p),
        ("lj_B_host", C.c_void_p),
        ("excl_offsets_host", C.c_void_p),
        ("excl_index_host", C.c_void_p),
        ("terms", C.c_uint32),
        ("rfa", C.c_int32),
        ("cutoff", C.c_double),
        ("switch_dist", C.c_double),
        ("solvent_dielectric", C.c_double),
        ("switch_mode", C.c_int32),
        ("algorithm", C.c_int32),
        ("skin", C.c_double),
    ]


class BondedDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("nbonds", C.c_int32),
        ("bond_idx_host", C.c_void_p),
        ("bond_prm_host", C.c_void_p),
        ("nangles", C.c_int32),
        ("angle_idx_host", C.c_void_p),
        ("angle_prm_host", C.c_void_p),
        ("ndihedrals", C.c_int32),
        ("dihedral_idx_host", C.c_void_p),
        ("ndihedral_terms", C.c_int32),
        ("dihedral_term_of_host", C.c_void_p),
        ("dihedral_prm_host", C.c_void_p),
        ("nimpropers", C.c_int32),
        ("improper_idx_host", C.c_void_p),
        ("nimproper_terms", C.c_int32),
        ("improper_term_of_host", C.c_void_p),
        ("improper_prm_host", C.c_void_p),
        ("n14", C.c_int32),
        ("pair14_idx_host", C.c_void_p),
        ("pair14_prm_host", C.c_void_p),
        ("terms14", C.c_uint32),
        ("bonds_use_cutoff", C.c_int32),
    ]


class MdDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("niter", C.c_int32),
        ("pos_dev", C.c_void_p),
        ("vel_dev", C.c_void_p),
        ("forces_dev", C.c_void_p),
        ("mass_dev", C.c_void_p),
        ("vcoeff_dev", C.c_void_p),
        ("box_host", C.c_void_p),
        ("dt", C.c_double),
        ("gamma", C.c_double),
        ("seed", C.c_uint64),
        ("step0", C.c_uint64),
        ("energies_dev", C.c_void_p),
        ("continuation", C.c_int32),
        ("reserved", C.c_int32),
    ]


class DdDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("dtype", C.c_int32),
        ("niter", C.c_int32),
        ("first_phases", C.c_int32),
        ("check_every", C.c_int32),
        ("reserved", C.c_int32),
        ("nown", C.c_int64),
        ("nhalo", C.c_int64),
        ("pos_dev", C.c_void_p),
        ("vel_dev", C.c_void_p),
        ("forces_dev", C.c_void_p),
        ("mass_dev", C.c_void_p),
        ("vcoeff_dev", C.c_void_p),
        ("ref_dev", C.c_void_p),
        ("disp2_dev", C.c_void_p),
        ("dt", C.c_double),
        ("gamma", C.c_double),
        ("seed", C.c_uint64),
        ("step0", C.c_uint64),
        ("nsend", C.c_int64),
        ("send_index_dev", C.c_void_p),
        ("send_shift_dev", C.c_void_p),
        ("send_buf_dev", C.c_void_p),
        ("send_counts_host", C.c_void_p),
        ("recv_counts_host", C.c_void_p),
        ("skin", C.c_double),
        ("since_migration", C.c_int64),
    ]


class DdBrick(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("dtype", C.c_int32),
        ("rank", C.c_int32),
        ("world", C.c_int32),
        ("dims", C.c_int32 * 3),
        ("ntypes_map", C.c_int32),
        ("box", C.c_double * 3),
        ("halo", C.c_double),
        ("cap_own", C.c_int64),
        ("cap_rows", C.c_int64),
        ("cap_send", C.c_int64),
        ("nown", C.c_int64),
        ("nhalo", C.c_int64),
        ("nsend", C.c_int64),
        ("ids_dev", C.c_void_p),
        ("pos_dev", C.c_void_p),
        ("unwrap_dev", C.c_void_p),
        ("vel_dev", C.c_void_p),
        ("charge_dev", C.c_void_p),
        ("type_dev", C.c_void_p),
        ("mass_dev", C.c_void_p),
        ("ref_dev", C.c_void_p),
        ("disp2_dev", C.c_void_p),
        ("send_index_dev", C.c_void_p),
        ("send_shift_dev", C.c_void_p),
        ("send_counts_host", C.c_void_p),
        ("recv_counts_host", C.c_void_p),
        ("type_map_host", C.c_void_p),
        ("need_own", C.c_int64),
        ("need_rows", C.c_int64),
        ("need_send", C.c_int64),
    ]


COMM_ID_BYTES = 128


class Stats(C.Structure):
    _fields_ = [
        ("n_compute", C.c_int64),
        ("n_rebuilds", C.c_int64),
        ("list_entries", C.c_int64),
        ("pairs_in_cutoff", C.c_int64),
        ("algorithm", C.c_int32),
        ("max_neighbours", C.c_int32),
        ("overflow", C.c_int32),
        ("ncell", C.c_int32 * 3),
        ("skin", C.c_double),
        ("chains_skipped", C.c_int64),
        ("steps_in_pair_launch", C.c_int64),
        ("fused_step_timeouts", C.c_int64),
        ("final_steps_in_pair_launch", C.c_int64),
        ("batched_launches", C.c_int64),
    ]


# name -> (restype, argtypes): every symbol include/tmdhip.h declares
SIGNATURES = {
    "tmdhip_abi_version": (C.c_int, []),
    "tmdhip_last_error": (C.c_char_p, []),
    "tmdhip_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(NonbondedDesc)]),
    "tmdhip_set_bonded": (C.c_int, [C.c_void_p, C.POINTER(BondedDesc)]),
    "tmdhip_destroy": (None, [C.c_void_p]),
    "tmdhip_compute_nonbonded": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p],
    ),
    "tmdhip_compute_bonded": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p],
    ),
    "tmdhip_compute": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_double), C.c_void_p]),
    "tmdhip_check": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "tmdhip_md_run": (C.c_int, [C.c_void_p, C.POINTER(MdDesc), C.c_void_p]),
    "tmdhip_md_restore": (C.c_int, [C.c_void_p, C.POINTER(MdDesc), C.c_void_p]),
    "tmdhip_md_observe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_void_p]),
    "tmdhip_invalidate_list": (C.c_int, [C.c_void_p, C.c_int]),
    "tmdhip_update_atoms": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "tmdhip_set_skin_weights": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tmdhip_get_stats": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Stats)]),
    "tmdhip_timing_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "tmdhip_timing_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
    "tmdhip_first_vv": (
        C.c_int,
        [C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p],
    ),
    "tmdhip_second_vv": (
        C.c_int,
        [C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p],
    ),
    "tmdhip_langevin_second_vv": (
        C.c_int,
        [
            C.c_int,
            C.c_int64,
            C.c_int64,
            C.c_void_p,
            C.c_void_p,
            C.c_void_p,
            C.c_void_p,
            C.c_double,
            C.c_double,
            C.c_uint64,
            C.c_uint64,
            C.c_void_p,
        ],
    ),
    "tmdhip_kinetic_energy": (
        C.c_int,
        [C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "tmdhip_wrap": (
        C.c_int,
        [C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p],
    ),
    "tmdhip_normal_fill": (C.c_int, [C.c_int, C.c_int64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "tmdhip_dd_step": (
        C.c_int,
        [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
         C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "tmdhip_halo_pack": (C.c_int, [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tmdhip_comm_unique_id": (C.c_int, [C.c_char_p, C.c_void_p]),
    "tmdhip_comm_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p, C.c_void_p, C.c_int, C.c_int]),
    "tmdhip_comm_destroy": (None, [C.c_void_p]),
    "tmdhip_comm_exchange": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p],
    ),
    "tmdhip_dd_run": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(DdDesc), C.POINTER(C.c_int32), C.c_void_p]),
    "tmdhip_dd_reset": (C.c_int, [C.c_void_p]),
    "tmdhip_dd_migrate": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(DdBrick), C.c_void_p]),
    "tmdhip_local_hub_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "tmdhip_local_hub_destroy": (None, [C.c_void_p]),
    "tmdhip_comm_create_local": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int]),
    "tmdhip_debug_build_timeline": (C.c_int, [C.c_void_p, C.c_size_t]),
}

_lib = None


def library_path() -> str:
    return LIBPATH


def load():
    """Load (once) and return the ctypes handle.  torch is imported first so that the HIP runtime the
    library resolves against (SONAME libamdhip64.so.7) is the one torch already mapped."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (maps torch's libamdhip64 before ours is resolved)

    if not os.path.exists(LIBPATH) and os.environ.get("TMDHIP_NO_AUTOBUILD") != "1":
        # a source checkout without the built library: compile it in-tree (hipcc, ~25 s).  This builds
        # the product; it is not a fallback — without hipcc the error below is raised.
        try:
            from . import _build

            _build.build_library()
        except Exception as exc:  # noqa: BLE001
            raise RuntimeError(
                f"{LIBPATH} is missing and could not be built ({exc}). Run `python -m torchmd_amd._build` "
                "(needs hipcc, cross-compiles for gfx950 without a GPU). torchmd_amd has no CPU or PyTorch fallback."
            ) from exc
    if not os.path.exists(LIBPATH):
        raise RuntimeError(
            f"{LIBPATH} is missing: the HIP extension has not been built. Run "
            "`python -m torchmd_amd._build` (needs hipcc, cross-compiles for gfx950 without a GPU). "
            "torchmd_amd has no CPU or PyTorch fallback."
        )
    try:
        lib = C.CDLL(LIBPATH, mode=C.RTLD_GLOBAL)
    except OSError as exc:  # pragma: no cover
        raise RuntimeError(f"could not load {LIBPATH}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.tmdhip_abi_version() != ABI_VERSION:
        raise RuntimeError("libtmdhip.so ABI version mismatch: rebuild with `python -m torchmd_amd._build --force`")
    _lib = lib
    return lib


def last_error() -> str:
    return load().tmdhip_last_error().decode("utf-8", "replace")


def check(rc: int, what: str = "tmdhip"):
    if rc < 0:
        raise RuntimeError(f"{what}: {last_error()}")
    return rc


def dtype_code(torch_dtype) -> int:
    import torch

    if torch_dtype == torch.float32:
        return F32
    if torch_dtype == torch.float64:
        return F64
    raise TypeError(f"torchmd_amd supports float32 and float64 tensors, got {torch_dtype}")


def require_device_tensor(t, name):
    if not t.is_cuda:
        raise RuntimeError(
            f"`{name}` lives on '{t.device}': torchmd_amd evaluates the hot path with HIP kernels on a ROCm "
            "device only (device='cuda'); there is no CPU fallback. Use the reference torchmd for CPU runs."
        )
