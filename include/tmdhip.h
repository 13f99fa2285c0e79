/*
 * tmdhip.h — C ABI of the MI355X (gfx950) nonbonded / integrator engine for TorchMD.
 *
 * This is the drop-in boundary of the hot path.  The reference has no native
 * layer at all: its hot path is Python/PyTorch.  Each entry point below therefore cites the
 * reference *Python* function whose arithmetic it replaces; the host-side mirror of the reference
 * API (`torchmd_amd.forces.Forces`, `torchmd_amd.integrator.Integrator`, `torchmd_amd.systems.System`)
 * binds these symbols through ctypes.
 *
 * Conventions
 *   - plain C types only; no torch / C++ types cross the boundary
 *   - `*_dev` pointers are device (HBM) pointers owned by the caller (torch tensors' data_ptr());
 *     `*_host` pointers are host memory, only read during the call
 *   - `real` = float (dtype 0) or double (dtype 1); positions/forces/velocities are [N,3] AoS
 *     contiguous exactly like the reference's `System` tensors (systems.py:12-16)
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream); no call synchronises the device unless documented
 *   - return 0 on success, negative on error; tmdhip_last_error() gives a thread-local message
 *   - one host thread per context (contexts are not re-entrant)
 */
#ifndef TMDHIP_H
#define TMDHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMDHIP_ABI_VERSION 8

/* dtype */
#define TMDHIP_F32 0
#define TMDHIP_F64 1

/* nonbonded term mask — names of reference Forces.nonbonded (forces.py:24) */
#define TMDHIP_TERM_LJ 1u
#define TMDHIP_TERM_ELECTROSTATICS 2u
#define TMDHIP_TERM_REPULSION 4u
#define TMDHIP_TERM_REPULSIONCG 8u

/* layout of the energy accumulator double[TMDHIP_NENERGY] (per replica) */
#define TMDHIP_E_LJ 0
#define TMDHIP_E_ELECTROSTATICS 1
#define TMDHIP_E_REPULSION 2
#define TMDHIP_E_REPULSIONCG 3
#define TMDHIP_E_BONDS 4
#define TMDHIP_E_ANGLES 5
#define TMDHIP_E_DIHEDRALS 6
#define TMDHIP_E_IMPROPERS 7
#define TMDHIP_NENERGY 8

/* compute flags */
#define TMDHIP_WANT_ENERGY 1 /* accumulate (+=) per-term energies into energies_dev          */
#define TMDHIP_WANT_FORCES 2 /* accumulate (+=) forces into forces_dev                        */
#define TMDHIP_COUNT_PAIRS 4 /* also count non-excluded i<j pairs with r <= cutoff (stats)    */
#define TMDHIP_OVERWRITE_FORCES 8 /* store (=) instead of +=: nonbonded, so that the caller need not zero
                                     forces_dev first (the all-pairs path zero-fills internally); bonded
                                     (ABI 4): forces_dev receives the bonded force alone                  */

/* pair-search algorithm */
#define TMDHIP_ALGO_AUTO 0
#define TMDHIP_ALGO_ALLPAIRS 1 /* tiled O(N^2) kernel: small / non-periodic / no-cutoff systems */
#define TMDHIP_ALGO_CELLLIST 2 /* cell binning + Verlet list + list pair kernel                  */

/* switching-force mode (forces.py:403-413) */
#define TMDHIP_SWITCH_REFERENCE 0 /* explicit force = S*f + E*S'/r  (upstream's extra 1/r kept) */
#define TMDHIP_SWITCH_EXACT 1     /* explicit force = S*f + E*S'    (-dE/dr)                      */

typedef struct tmdhip_ctx tmdhip_ctx;

/* Static description of one system's nonbonded interactions.
 * Replaces the state reference `Forces.__init__` derives (forces.py:27-74): the A/B tables
 * (parameters.py:449-457), charges, mapped atom types, and the exclusion set that
 * `_make_indeces` (forces.py:348-357) bakes into its dense pair tensor. */
typedef struct tmdhip_nonbonded_desc {
  int32_t struct_size; /* = sizeof(tmdhip_nonbonded_desc) */
  int32_t dtype;       /* TMDHIP_F32 / TMDHIP_F64: type of pos/forces and of the real-valued arrays below */
  int32_t natoms;
  int32_t ntypes;
  int32_t nreplicas;           /* independent copies of the system (own neighbour lists)  */
  int32_t device;              /* HIP device ordinal                                       */
  const int32_t *types_host;   /* [natoms] index into the A/B tables                       */
  const void *charges_host;    /* real [natoms], units of e                                */
  const void *lj_A_host;       /* real [ntypes*ntypes] or NULL                             */
  const void *lj_B_host;       /* real [ntypes*ntypes] or NULL                             */
  const int32_t *excl_offsets_host; /* CSR [natoms+1]; row i = sorted partners of i (both directions stored) */
  const int32_t *excl_index_host;   /* [excl_offsets[natoms]]                              */
  uint32_t terms;              /* TMDHIP_TERM_* mask                                       */
  int32_t rfa;                 /* reaction field on/off (needs cutoff)                     */
  double cutoff;               /* Angstrom; <= 0: no cutoff                                */
  double switch_dist;          /* Angstrom; <= 0: no switching                             */
  double solvent_dielectric;   /* reference default 78.5                                   */
  int32_t switch_mode;         /* TMDHIP_SWITCH_*                                          */
  int32_t algorithm;           /* TMDHIP_ALGO_*                                            */
  double skin;                 /* Verlet skin in Angstrom; <= 0: library default (1.2)     */
} tmdhip_nonbonded_desc;

/* Bonded topology (already expanded: one parameter row per instance).  Replaces the per-call
 * gathers of forces.py:122-258.  All index arrays are int32 host arrays; params are real. */
typedef struct tmdhip_bonded_desc {
  int32_t struct_size;
  int32_t nbonds;
  const int32_t *bond_idx_host;   /* [nbonds*2]                          */
  const void *bond_prm_host;      /* real [nbonds*2] = (k0, d0)          */
  int32_t nangles;
  const int32_t *angle_idx_host;  /* [nangles*3]                         */
  const void *angle_prm_host;     /* real [nangles*2] = (k0, theta0)     */
  int32_t ndihedrals;             /* distinct proper torsions            */
  const int32_t *dihedral_idx_host;   /* [ndihedrals*4]                  */
  int32_t ndihedral_terms;
  const int32_t *dihedral_term_of_host; /* [nterms] torsion each term belongs to, ascending */
  const void *dihedral_prm_host;        /* real [nterms*3] = (k0, phi0, per)                */
  int32_t nimpropers;
  const int32_t *improper_idx_host;
  int32_t nimproper_terms;
  const int32_t *improper_term_of_host;
  const void *improper_prm_host;
  int32_t n14;
  const int32_t *pair14_idx_host; /* [n14*2]                             */
  const void *pair14_prm_host;    /* real [n14*4] = (A, B, scnb, scee)   */
  uint32_t terms14;               /* TMDHIP_TERM_LJ | TMDHIP_TERM_ELECTROSTATICS: which 1-4 parts are on */
  int32_t bonds_use_cutoff;       /* reference filters bonds by `cutoff` when one is set (forces.py:128-136) */
} tmdhip_bonded_desc;

typedef struct tmdhip_stats {
  int64_t n_compute;        /* tmdhip_compute calls on this replica                         */
  int64_t n_rebuilds;       /* neighbour-list rebuilds                                      */
  int64_t list_entries;     /* entries in the current (full) Verlet list                    */
  int64_t pairs_in_cutoff;  /* result of the last TMDHIP_COUNT_PAIRS call                   */
  int32_t algorithm;        /* algorithm in use                                             */
  int32_t max_neighbours;   /* list capacity per atom                                       */
  int32_t overflow;         /* != 0: a list is currently truncated (see tmdhip_check)       */
  int32_t ncell[3];
  double skin;              /* Verlet skin in use (Angstrom)                                */
  int64_t chains_skipped;   /* MD steps whose rebuild chain the host left out (tmdhip_md_run)  */
  int64_t steps_in_pair_launch; /* MD steps made by step blocks of the pair launch instead of an integrator launch (ABI 4) */
  int64_t fused_step_timeouts;  /* batches rewound because a step block of a fused launch gave up waiting (ABI 5; whole context) */
  int64_t final_steps_in_pair_launch; /* tmdhip_md_run calls whose LAST step (final kick, bonded + kinetic energies) was made by the
                                       * step blocks of the last pair launch (ABI 8; whole context) */
  int64_t batched_launches;     /* pair + step launches that served several replicas of a cell-list context at once (ABI 8; whole context) */
} tmdhip_stats;

int tmdhip_abi_version(void);
const char *tmdhip_last_error(void);

/* Lifetime.  create copies every host array to the device. */
int tmdhip_create(tmdhip_ctx **out, const tmdhip_nonbonded_desc *desc);
int tmdhip_set_bonded(tmdhip_ctx *ctx, const tmdhip_bonded_desc *desc);
void tmdhip_destroy(tmdhip_ctx *ctx);

/* Nonbonded block of Forces.compute (forces.py:260-319) for one replica: minimum-image distances
 * (360-372), `dist <= cutoff` filter (76-81), LJ (+switch) / Coulomb / reaction field / repulsion
 * (381-491), force scatter and per-term energy sums (316-319).
 * box_host = the three box edge lengths (diagonal of box[r], forces.py:118); all zero = no wrapping.
 * replica = TMDHIP_ALL_REPLICAS: the body of the reference's `for i in range(nsystems)` loop
 * (forces.py:116) for every replica in one call — pos_dev/forces_dev are the full [R][N][3] arrays,
 * energies_dev is [R][TMDHIP_NENERGY], box_host is [R][3]; all-pairs contexts serve all replicas with
 * one launch per kernel (small systems are launch-bound). */
#define TMDHIP_ALL_REPLICAS (-1)
int tmdhip_compute_nonbonded(tmdhip_ctx *ctx, int replica, const void *pos_dev, const double *box_host,
                             void *forces_dev, double *energies_dev, int flags, void *stream);

/* Bonded block of Forces.compute (forces.py:122-258, 494-605). */
int tmdhip_compute_bonded(tmdhip_ctx *ctx, int replica, const void *pos_dev, const double *box_host,
                          void *forces_dev, double *energies_dev, int flags, void *stream);

/* Forces.compute (forces.py:83-346) for every replica in ONE call with ONE host synchronisation: bonded +
 * nonbonded forces stored into forces_dev (real [R,N,3]; NULL: energies only, `calculateForces=False`) and the
 * per-term energies returned on the host (energies_host: double [R][TMDHIP_NENERGY]).  The neighbour-list
 * validity check rides on the same read-back (for up to 16 replicas one small kernel writes everything into
 * host-mapped memory followed by a sequence word the host spins on; more replicas: copies + stream
 * synchronisation).  Returns 0 = valid; 1 = a list was truncated (capacity grown):
 * call again; negative = error. */
int tmdhip_compute(tmdhip_ctx *ctx, const void *pos_dev, const double *box_host, void *forces_dev,
                   double *energies_host, void *stream);

/* Synchronises `stream` and verifies that the neighbour lists used since the last check were valid: no
 * device-side rebuild ran out of list capacity.  Returns 0 = results valid; 1 = not valid: the capacity has
 * been grown, the next compute rebuilds, and the caller must repeat the work (a plain evaluation is simply repeated; an
 * MD batch is rewound with tmdhip_md_restore and run again); negative = error. */
int tmdhip_check(tmdhip_ctx *ctx, int replica, void *stream);

/* `niter` iterations of Integrator.step's loop (integrator.py:115-120) for every replica, enqueued from
 * C with no per-step host work:  _first_VV -> Forces.compute (bonded + nonbonded) -> langevin -> _second_VV.
 * Between steps the second half kick of step s-1, the first half step of step s and the neighbour-list
 * displacement test are ONE fused kernel.  Forces of the previous evaluation must be in forces_dev on
 * entry (as the reference requires: run.py:261 primes system.forces); on return forces_dev holds the
 * forces of the last step and velocities have received both half kicks.  (Where the lean fp32 kernel serves the
 * context, the last step — with its energies — is made by the last pair launch itself; energies_dev is complete in
 * stream order when the call returns either way, and a tmdhip_md_observe with TMDHIP_OBSERVE_AFTER_RUN finds the kinetic
 * energy already summed.  Cell-list contexts with several replicas: the pair + step blocks of all replicas share ONE
 * launch per step — ABI 8 —, every replica with its own neighbour state; TMDHIP_BATCH_REPLICAS=0 in the environment
 * keeps the replica-by-replica loop.) */
typedef struct tmdhip_md_desc {
  int32_t struct_size;
  int32_t niter;
  void *pos_dev, *vel_dev, *forces_dev; /* real [R,N,3]                                              */
  const void *mass_dev;                 /* real [N]                                                  */
  const void *vcoeff_dev;               /* real [N] = sqrt(2 gamma kB T dt / m), or NULL: no thermostat */
  const double *box_host;               /* [R*3] box edge lengths                                    */
  double dt;                            /* time step in internal units (fs / 48.88821)               */
  double gamma;                         /* friction in internal units                                */
  uint64_t seed, step0;                 /* noise stream key and position (step0 + iteration)         */
  double *energies_dev;                 /* [R*TMDHIP_NENERGY]: energies of the LAST iteration (overwritten), or NULL */
  int32_t continuation;                 /* != 0: positions and box are what the previous tmdhip_md_run of this context
                                         * left (a hint: the first step may then leave its rebuild chain out like any
                                         * other; a wrong hint costs a rewind, never a wrong result) (ABI 4) */
  int32_t reserved;
} tmdhip_md_desc;
int tmdhip_md_run(tmdhip_ctx *ctx, const tmdhip_md_desc *desc, void *stream);
/* What Integrator.step returns after its loop (integrator.py:121-125), with ONE read-back and ONE host
 * synchronisation: kinetic energy per replica (integrator.py:8-31), the per-term energies of the last step
 * (energies_dev = the buffer given to tmdhip_md_run; NULL: zeros) and the neighbour-list validity check.
 * out_host: double [R][TMDHIP_NENERGY + 1] = per-term energies, then Ekin.  Returns as tmdhip_check.
 * flags (ABI 8): TMDHIP_OBSERVE_AFTER_RUN = the caller vouches that vel_dev / mass_dev are the buffers of the tmdhip_md_run
 * call of this context that has just returned and that NOTHING has written the velocities since (no rescaling, no
 * tmdhip_second_vv, no restore).  Only then may the kinetic energy that the run's last launch has already summed be
 * reused; without the flag (0) the kinetic-energy kernel always runs. */
#define TMDHIP_OBSERVE_AFTER_RUN 1
int tmdhip_md_observe(tmdhip_ctx *ctx, const void *vel_dev, const void *mass_dev, const double *energies_dev,
                      double *out_host, int flags, void *stream);
/* Rewind: copy the state tmdhip_md_run saved at its entry (positions, velocities, forces of every replica)
 * back into desc's buffers.  Used after tmdhip_md_observe / tmdhip_check returned 1 for an MD batch (a list was
 * truncated: the capacity has been grown); the noise stream is counter based, so running the batch again gives
 * the trajectory the truncated run should have produced. */
int tmdhip_md_restore(tmdhip_ctx *ctx, const tmdhip_md_desc *desc, void *stream);

/* Atomic systems only (no exclusions, no bonded terms): replace the atom set of the context — new count,
 * types and charges; the LJ table and all options stay — and mark atoms with index >= nactive (<= 0: none)
 * as passive: they act on the others but get no neighbour list and zero force.  Used by the spatial domain
 * decomposition (a brick's own atoms followed by its halo images) at every atom migration instead of
 * re-creating the context; synchronises the device; the next compute re-plans and rebuilds. */
int tmdhip_update_atoms(tmdhip_ctx *ctx, int natoms, const int32_t *types_host, const void *charges_host,
                        int nactive);

/* Per-atom Verlet skins (cell-list path).  By default every atom may move skin/2 before the list is rebuilt and
 * every pair within cutoff + skin is listed.  With weights w_i in (0, 1] (real [natoms], host) atom i may move
 * s_i = w_i * skin/2 and pair (i, j) is listed within cutoff + s_i + s_j — equally exact (a pair that is not
 * listed cannot come within the cutoff before one of its atoms exceeds its s), but slow atoms (heavy ones: a
 * water oxygen moves 0.28 of what its hydrogens move) stop paying for the skin the fast ones need.  Inside
 * tmdhip_md_run, where velocities are known, a rebuild sizes s_i from the atom's speed as well (0.8 of the static
 * share + the path it covers in 6 fs, at most 1.2 x the largest static share; TMDHIP_VSKIN=0 switches that off).
 * NULL restores the uniform skin.  Synchronises the device; the next compute re-plans and rebuilds. */
int tmdhip_set_skin_weights(tmdhip_ctx *ctx, const void *weights_host);

/* Drop the neighbour list of a replica: the next tmdhip_compute_nonbonded rebuilds it (used after the
 * caller has changed positions out of band, and by the rebuild timing tool). */
int tmdhip_invalidate_list(tmdhip_ctx *ctx, int replica);

/* Host-synchronising query (copies a few words back). */
int tmdhip_get_stats(tmdhip_ctx *ctx, int replica, tmdhip_stats *out);

/* HIP-event timing of the dominant (pair) kernel, recorded on the launch stream.  on = 0: off; low 16 bits = n:
 * every n-th launch (1: every launch); bits 16..27 = stop after that many timed launches (0: no limit); bits 28..30 =
 * launches passed over before the first timed one; bit 31 = launches that also return energies (the last step of a
 * tmdhip_md_run call, tmdhip_compute: another variant of the kernel) are neither timed nor counted.  The
 * start / stop events are attached to the kernel's own dispatch (hipExtLaunchKernel), not recorded around it. */
int tmdhip_timing_enable(tmdhip_ctx *ctx, int on);
int tmdhip_timing_read(tmdhip_ctx *ctx, double *pair_kernel_ms, int64_t *launches, int reset);

/* Integrator kernels (stateless).  n = nreplicas*natoms rows; mass_dev is real [natoms] and is
 * indexed by (row % natoms) exactly like the broadcast of masses [N,1] in integrator.py:61-69. */
/* _first_VV (integrator.py:61-64): pos += vel*dt + 0.5*(F/m)*dt*dt ; vel += 0.5*dt*(F/m) */
int tmdhip_first_vv(int dtype, int64_t nreplicas, int64_t natoms, void *pos_dev, void *vel_dev,
                    const void *forces_dev, const void *mass_dev, double dt, void *stream);
/* _second_VV (integrator.py:67-69): vel += 0.5*dt*(F/m) */
int tmdhip_second_vv(int dtype, int64_t nreplicas, int64_t natoms, void *vel_dev, const void *forces_dev,
                     const void *mass_dev, double dt, void *stream);
/* langevin (integrator.py:72-74) fused with _second_VV:
 *   vel += -gamma*vel*dt + N(0,1)*vcoeff ; vel += 0.5*dt*(F/m)
 * N(0,1) from a counter-based Philox4x32-10 stream keyed by (seed, step, row). */
int tmdhip_langevin_second_vv(int dtype, int64_t nreplicas, int64_t natoms, void *vel_dev,
                              const void *forces_dev, const void *mass_dev, const void *vcoeff_dev,
                              double dt, double gamma, uint64_t seed, uint64_t step, void *stream);
/* kinetic_energy (integrator.py:8-31, batch=None): ke_dev[r] = sum_i 0.5*m_i*|v_ri|^2 (overwrites) */
int tmdhip_kinetic_energy(int dtype, int64_t nreplicas, int64_t natoms, const void *vel_dev,
                          const void *mass_dev, double *ke_dev, void *stream);
/* Wrapper.wrap (wrapper.py:8-30): translate every bonded group by -floor(com/box)*box (com = unweighted
 * mean of its atoms).  pos_dev real [R,N,3] in place; box_dev real [R,3,3] (device, diagonal used; an
 * all-zero box leaves the replica untouched); groups as a CSR over atoms (device int32 arrays; atoms
 * without bonds are groups of one).  has_big_groups != 0 if any group has more than 64 atoms. */
int tmdhip_wrap(int dtype, int64_t nreplicas, int64_t natoms, void *pos_dev, const void *box_dev, int32_t ngroups,
                const int32_t *group_offsets_dev, const int32_t *group_members_dev, int32_t has_big_groups,
                void *stream);
/* Fill `out_dev` (real [n]) with the N(0,1) stream used by tmdhip_langevin_second_vv (for tests). */
int tmdhip_normal_fill(int dtype, int64_t n, void *out_dev, uint64_t seed, uint64_t step, void *stream);


/* ---- spatial domain decomposition (no reference counterpart: torchmd is single-device) ----
 * One brick = one rank.  pos_dev is the position buffer of the brick's force engine, real [nown + nhalo, 3]:
 * the owned atoms first (integrated in place), then the halo rows the exchange writes.
 *
 * tmdhip_dd_step: on the owned atoms, phases bit 0 = second half kick of the previous step (preceded by the
 * Langevin update when vcoeff_dev != NULL; integrator.py:72-74, 67-69), bit 1 = first half step of this step
 * (integrator.py:61-64); same expressions and rounding as the separate kernels above.  With ref_dev (positions
 * at the last migration, real [nown, 3]) and disp2_dev, phase bit 1 also folds max_i |pos_i - ref_i|^2 into
 * *disp2_dev (float bits, atomic max; the caller zeroes it at a migration). */
int tmdhip_dd_step(int dtype, int64_t nown, void *pos_dev, void *vel_dev, const void *forces_dev,
                   const void *mass_dev, const void *vcoeff_dev, double dt, double gamma, uint64_t seed,
                   uint64_t step, int phases, const void *ref_dev, uint32_t *disp2_dev, void *stream);
/* out_dev[k, :] = pos_dev[index_dev[k], :] + shift_dev[k, :] for the `count` rows of all outgoing halo
 * messages (message order; shift = the periodic image the receiver sees). */
int tmdhip_halo_pack(int dtype, int64_t count, const void *pos_dev, const int32_t *index_dev,
                     const void *shift_dev, void *out_dev, void *stream);

/* Halo-exchange communicator: RCCL point-to-point among the `world` ranks of the brick grid (one process and
 * one GPU per rank).  librccl is opened at run time from `librccl_path` (NULL/"" = default search; pass the copy
 * the process has already mapped, e.g. PyTorch's torch/lib/librccl.so).  Rank 0 draws the id, the caller
 * distributes its TMDHIP_COMM_ID_BYTES bytes to the other ranks by any means (torch.distributed broadcast),
 * every rank then calls tmdhip_comm_create with its HIP device current. */
#define TMDHIP_COMM_ID_BYTES 128
typedef struct tmdhip_comm tmdhip_comm;
int tmdhip_comm_unique_id(const char *librccl_path, void *id_out);
int tmdhip_comm_create(tmdhip_comm **out, const char *librccl_path, const void *id, int rank, int world);
void tmdhip_comm_destroy(tmdhip_comm *comm);
/* The same communicator over an IN-PROCESS transport (ABI 5): all `world` ranks live in one process on one device, one
 * host thread per rank, each with a stream of its own.  An exchange is a rendezvous of the threads around device-side
 * copies ordered by events (no RCCL, no second GPU): what lets tmdhip_dd_run execute at world 2 / 4 / 8 on a one-GPU
 * box with the decisions it takes over RCCL.  Create one hub, then one communicator per rank (any thread); every
 * rank's thread must take part in every exchange (tmdhip_comm_exchange / tmdhip_dd_run); a rank that stays away for
 * 30 s breaks the hub (all later calls fail).  Destroy the communicators before the hub. */
typedef struct tmdhip_local_hub tmdhip_local_hub;
int tmdhip_local_hub_create(tmdhip_local_hub **out, int world);
void tmdhip_local_hub_destroy(tmdhip_local_hub *hub);
int tmdhip_comm_create_local(tmdhip_comm **out, tmdhip_local_hub *hub, int rank);
/* One grouped exchange on `stream`: rows of `width` reals; the first send_counts_host[0] rows of send_dev go to
 * rank 0, the next send_counts_host[1] to rank 1, ...; recv_dev receives recv_counts_host[p] rows from rank p
 * in rank order (the layout of an all-to-all with split sizes, as ncclSend/ncclRecv pairs between the
 * ranks that actually exchange rows — messages to oneself across the periodic boundary included). */
int tmdhip_comm_exchange(tmdhip_comm *comm, int dtype, const void *send_dev, const int64_t *send_counts_host,
                         void *recv_dev, const int64_t *recv_counts_host, int width, void *stream);

/* The step loop of one brick enqueued from C (velocity Verlet + optional Langevin over the decomposed system):
 *   per iteration: tmdhip_dd_step (kick of the previous iteration + drift of this one) -> tmdhip_halo_pack ->
 *   tmdhip_comm_exchange into the halo rows of pos_dev -> nonbonded forces on the owned atoms (ctx: open
 *   boundaries, atoms >= nown passive, see tmdhip_update_atoms);  after the last iteration the owed kick.
 *   (Since ABI 6 the loop runs these as three launches + the exchange per iteration — the update of the owned atoms
 *   with the list's displacement test, their cell-sorted records and their outgoing rows; the same for the halo rows
 *   that arrived; the pair kernel, with the rebuild chain left out while no atom is near its limit — same results
 *   bit for bit; TMDHIP_DD_FUSED=0 in the environment keeps the separate launches.)
 * Every `check_every` iterations the maximum squared displacement since the last migration is max-reduced over
 * the ranks and copied to the host asynchronously; it is examined one check later, so the loop never waits for
 * the device.  When the projected displacement exceeds skin/2 the call returns 1 with *iters_done = the number
 * of complete iterations: the next one has drifted but has neither halo nor forces yet — the caller migrates
 * atoms, evaluates the forces, counts that iteration as done and calls again with first_phases = 3 (a kick is
 * owed; niter may then be 0).  Returns 0 when all niter iterations are done, negative on error.
 * TMDHIP_DD_OVERRUN (2): a displacement that was MEASURED lies beyond skin/2 already — the projection was too
 * optimistic (hot atoms, a large check_every) and halo atoms were missing in the last iterations, so the state is
 * invalid; *iters_done = 0, tmdhip_last_error() says how far the atom went.  Every rank returns it at the same
 * iteration.  The caller goes back to a state it saved at the last migration and repeats with a smaller
 * check_every (torchmd_amd/domain.py does; counter-based noise makes the repeat reproducible). */
#define TMDHIP_DD_OVERRUN 2
typedef struct tmdhip_dd_desc {
  int32_t struct_size;
  int32_t dtype;
  int32_t niter;
  int32_t first_phases;      /* 2: velocities complete on entry; 3: the second half kick of the last step is owed */
  int32_t check_every;       /* iterations between displacement checks                                        */
  int32_t reserved;
  int64_t nown, nhalo;
  void *pos_dev;             /* real [nown + nhalo, 3]                                                        */
  void *vel_dev;             /* real [nown, 3]                                                                */
  void *forces_dev;          /* real [nown + nhalo, 3] (halo rows stay zero)                                  */
  const void *mass_dev;      /* real [nown]                                                                   */
  const void *vcoeff_dev;    /* real [nown] or NULL: no thermostat                                            */
  const void *ref_dev;       /* real [nown, 3]: positions at the last migration                               */
  uint32_t *disp2_dev;       /* running maximum of |pos - ref|^2 (float bits); zero at a migration            */
  double dt, gamma;
  uint64_t seed, step0;      /* noise stream key; iteration i kicks with counter step0 + i                    */
  int64_t nsend;             /* rows of all outgoing messages                                                 */
  const int32_t *send_index_dev;
  const void *send_shift_dev; /* real [nsend, 3]                                                              */
  void *send_buf_dev;        /* real [nsend, 3]                                                               */
  const int64_t *send_counts_host, *recv_counts_host; /* [world] rows per peer; sum(recv) = nhalo             */
  double skin;               /* halo skin: migration when the projected displacement exceeds skin / 2         */
  int64_t since_migration;   /* iterations started since the last migration, on entry                         */
} tmdhip_dd_desc;
int tmdhip_dd_run(tmdhip_ctx *ctx, tmdhip_comm *comm, const tmdhip_dd_desc *desc, int32_t *iters_done,
                  void *stream);
/* Forget the pending displacement read-back (call after every migration). */
int tmdhip_dd_reset(tmdhip_comm *comm);

/* ---- atom migration of a brick, on the device (ABI 6) ----
 * When tmdhip_dd_run returns 1 the caller re-assigns atoms to bricks.  tmdhip_dd_migrate does that for one rank's
 * brick entirely in the library (every rank of the communicator must call it): the brick of every owned atom and
 * the counts per destination, an exchange of the atoms' state rows (id, position in the caller's periodic image,
 * velocity, charge, type, mass) over the communicator, the new owned set in the order of the global ids, the halo
 * plan (which owned atoms each of the 26 neighbour directions sees, with which periodic shift: send_index /
 * send_shift / send_counts in the order tmdhip_dd_run expects), the first halo exchange (positions, charges, types)
 * and the atom set of the force engine `ctx` (what tmdhip_update_atoms does from host arrays; atoms >= nown passive).
 * The arrays are the caller's, with capacities in rows; they are rewritten in place.  All communication happens
 * before anything is overwritten: when a capacity is too small the call returns 2 with need_own / need_rows /
 * need_send set and nothing lost — grow the arrays (their contents need not be kept when need_send == 0: the owned
 * rows are then still in the library's scratch; with need_send > 0 keep the first `nown` rows) and call again with
 * the same struct.  The forces have to be evaluated afterwards (the next tmdhip_compute_nonbonded re-plans the
 * engine's grid and rebuilds its list).  Returns 0, 2 (see above) or a negative error. */
typedef struct tmdhip_dd_brick {
  int32_t struct_size;
  int32_t dtype;
  int32_t rank, world;
  int32_t dims[3];           /* brick grid px, py, pz (rank = (cx py + cy) pz + cz)                           */
  int32_t ntypes_map;        /* entries of type_map_host                                                       */
  double box[3];
  double halo;               /* cutoff + halo skin                                                             */
  int64_t cap_own, cap_rows, cap_send; /* capacities in rows: per-owned-atom arrays, pos_dev, the send list    */
  int64_t nown, nhalo, nsend;          /* in: nown; out: all three                                             */
  int64_t *ids_dev;          /* [cap_own] global atom ids                                                      */
  void *pos_dev;             /* real [cap_rows, 3]: owned rows (wrapped frame), then halo rows                 */
  void *unwrap_dev;          /* real [cap_own, 3]: caller's periodic image - wrapped position                  */
  void *vel_dev;             /* real [cap_own, 3]                                                              */
  void *charge_dev;          /* real [cap_own]                                                                 */
  int32_t *type_dev;         /* [cap_own]                                                                      */
  void *mass_dev;            /* real [cap_own]                                                                 */
  void *ref_dev;             /* real [cap_own, 3]: positions at this migration (tmdhip_dd_desc::ref_dev)       */
  uint32_t *disp2_dev;       /* zeroed                                                                         */
  int32_t *send_index_dev;   /* [cap_send]                                                                     */
  void *send_shift_dev;      /* real [cap_send, 3]                                                             */
  int64_t *send_counts_host, *recv_counts_host; /* [world], written                                            */
  const int32_t *type_map_host; /* [ntypes_map]: atom type -> LJ class of the context, or NULL (identity)      */
  int64_t need_own, need_rows, need_send; /* written on return 2                                               */
} tmdhip_dd_brick;
int tmdhip_dd_migrate(tmdhip_ctx *ctx, tmdhip_comm *comm, tmdhip_dd_brick *brick, void *stream);

/* ---- debug aid (not needed by any caller of the path) ----
 * With TMDHIP_DEBUG_TIMELINE=1 in the environment every block of the list build records {entry cycle, exit cycle,
 * hardware id, longest list | candidates x atoms << 32} (4 x uint64 per block); this copies the last build's records
 * to `out_host` (at most max_bytes) and returns the number of blocks (0: nothing recorded, < 0: error).
 * tools/build_timeline.py. */
int tmdhip_debug_build_timeline(void *out_host, size_t max_bytes);

#ifdef __cplusplus
}
#endif
#endif /* TMDHIP_H */
