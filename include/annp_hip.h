/* annp_hip.h -- C ABI of libannp_hip.so: the MI355X (gfx950) evaluation of the
 * LAMMPS `pair_style annp` hot path (descriptor build, per-atom network,
 * chain-rule forces).
 *
 * This header is the drop-in boundary.  It replaces the five free functions a
 * LAMMPS `src/GPU` pair style binds in `lib/gpu` for this potential:
 *
 *   reference (C++ linkage, static singleton)              here (C linkage, handle)
 *   ---------------------------------------------------   -------------------------
 *   annp_gpu_init      fe_v2/src/pair_annp_gpu.cpp:31-39   annp_hip_init
 *                      fe_v2/lib/lal_annp_ext.cpp:25-99
 *                      (ni adds cofsymrad/cofsymang:
 *                       ni/src/pair_annp_gpu.cpp:31-40)
 *   annp_gpu_compute   fe_v2/src/pair_annp_gpu.cpp:52-57   annp_hip_compute
 *                      fe_v2/lib/lal_annp_ext.cpp:110-119
 *   annp_gpu_compute_n fe_v2/src/pair_annp_gpu.cpp:44-49   annp_hip_compute_n
 *                      fe_v2/lib/lal_annp_ext.cpp:98-108
 *   annp_gpu_clear     fe_v2/lib/lal_annp_ext.cpp:94-96     annp_hip_clear
 *   annp_gpu_bytes     fe_v2/lib/lal_annp_ext.cpp:121-123   annp_hip_bytes
 *
 * plus device-resident entry points (annp_hip_compute_device,
 * annp_hip_neigh_build_device) for callers that already keep atoms in HBM
 * (KOKKOS/GPU-package style callers, the in-repo multi-GPU driver, bench.py).
 *
 * Conventions:
 *   - all host arrays are borrowed for the duration of the call; parameters are
 *     copied at init; device memory is owned by the handle.
 *   - forces ACCUMULATE into f (CPU pair_annp semantics, fe_v2/src/pair_annp.cpp:199,211),
 *     ghosts included (newton_pair on); eng_vdwl and eatom accumulate too.
 *   - neighbour indices are masked with NEIGHMASK (0x1FFFFFFF) as fe_v2:136 does.
 *   - return value 0 = ok; negative = error, same numbering as the reference's
 *     init codes (fe_v2/lib/lal_annp.h:28-33): -1 bad argument / not initialised,
 *     -3 out of device memory, -4 no usable gfx950 device / HIP runtime failure,
 *     -5 double precision unsupported, -7 neighbour capacity exceeded,
 *     -9 unsupported network / descriptor shape.  annp_hip_last_error() gives text.
 *     Nothing throws or exits across this boundary.
 *   - one caller thread per handle; several handles per process are fine.
 */
#ifndef ANNP_HIP_H
#define ANNP_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define ANNP_HIP_ABI_VERSION 7

#define ANNP_HIP_OK 0
#define ANNP_HIP_EARG (-1)
#define ANNP_HIP_ENOMEM (-3)
#define ANNP_HIP_EDEVICE (-4)
#define ANNP_HIP_ENODOUBLE (-5)
#define ANNP_HIP_ENEIGHCAP (-7)
#define ANNP_HIP_ESHAPE (-9)

/* which reference translation unit's arithmetic is evaluated */
#define ANNP_HIP_DESC_CHEBYSHEV 0   /* fe, fe_v2: fe_v2/src/pair_annp.cpp:633-695         */
#define ANNP_HIP_DESC_BEHLER 1      /* ni: G2/G4 in atomic units, ni/src/pair_annp.cpp:686-767 */
#define ANNP_HIP_DESC_ANNA_ADP 2    /* pair_style anna_adp: un-normalised Chebyshev descriptor -> network ->
                                       (d2, q2) of an analytic ADP form, anna-gpu-lammps/bcc_fe/src/pair_anna_adp.cpp:71-298 */

typedef struct annp_hip_handle annp_hip_handle;

/* Flat image of the arguments of annp_gpu_init.  Pointers are read during
 * annp_hip_init only. */
typedef struct annp_hip_params {
    int struct_bytes;       /* sizeof(annp_hip_params), ABI check                      */
    int descriptor;         /* ANNP_HIP_DESC_*                                          */
    int ntypes;             /* LAMMPS atom->ntypes (<= 30)                              */
    int nelements;          /* elements in the potential file (params[0].nelements; 0 is read as 1):
                               weight_all / bias_all carry one network per element, atom i is evaluated
                               with the network of element map[type[i]] (fe_v2/src/pair_annp.cpp:767-768);
                               the descriptor itself is species-blind (fe_v2:633-695)       */
    int ntl, nhl, nnod;     /* total layers, hidden layers, nodes per hidden layer      */
    int nsf, npsf, ntsf;    /* symmetry functions: all, radial, angular                 */
    int flagsym;            /* as parsed (informational; `descriptor` decides)          */
    int ni_compat;          /* BEHLER only: 1 = reproduce ni/src/pair_annp.cpp:737-738
                               (rik_m in place of rjk_m), 0 = the gradient-consistent
                               form of the reference GPU kernel ni/lib/lal_annp.cu:409-414 */
    const int *flagact;     /* [ntl-1] activation per weight layer: 0 linear, 1 tanh,
                               2 1/(1+exp(+a)), 3 1.7159 tanh(2a/3), 4 = 3 + 0.1a;
                               BEHLER: 3 and 4 are plain tanh (ni/src/pair_annp.cpp:781-808) */
    double e_scale, e_shift, e_atom;
    double cut;             /* cutmax from the potential file                           */
    const double *sfnor_scal; /* [nsf] CHEBYSHEV: 1/sqrt(cov-avg^2) (pair_annp_gpu.cpp:207-216)
                                       BEHLER: sf_max - sf_min (ni pair_annp_gpu.cpp:231-235) */
    const double *sfnor_avg;  /* [nsf] CHEBYSHEV: sfnor_avg;  BEHLER: sf_min               */
    const double *cutsq;      /* [(ntypes+1)*(ntypes+1)] LAMMPS cutsq, row-major.  init_one returns cutmax for every
                                 pair of mapped types (fe_v2:323-327), so the entries are cutmax^2 or 0 (a type that
                                 is not mapped: such atoms are neither neighbours nor centres, fe_v2:144);
                                 anything else is refused (-9)                               */
    const int *map;           /* [ntypes+1] LAMMPS type -> element, -1 = not mapped (NULL: every type -> 0) */
    const double *const *weight_all; /* [nelements*(ntl-1)] pointers, entry e*(ntl-1)+l = element e, layer l,
                                        row-major [nrow][ncol]: (nnod x nsf), (nnod x nnod)..., (1 x nnod)
                                        (the double*** weight_all[e][l] of annp_gpu_init)     */
    const double *const *bias_all;   /* [nelements*(ntl-1)] pointers, [nnod] ... [1]         */
    const double *cofsymrad;  /* BEHLER: [npsf*3] eta, Rs, Rc(Bohr); else NULL              */
    const double *cofsymang;  /* BEHLER: [ntsf*4] eta, lambda, zeta, Rc(Bohr); else NULL    */
    /* ANNA_ADP only (the arguments anna_adp_gpu_init adds, bcc_fe/src/pair_anna_adp_gpu.cpp:31-41) */
    int nout;                 /* network outputs: 2 (d2, q2); weight_all[ntl-2] is (nout x nnod)   */
    int ngp;                  /* number of analytic parameters: 17                                 */
    const double *gparams;    /* [ngp] A0 yy gamma C0 c1F c2F V0 b1 b2 delta r0 r1 hc d1 q1 d3 q3  */
    double e_base;            /* added to every atom's energy (pair_anna_adp.cpp:212); sfnor_*,
                                 e_scale/e_shift/e_atom are not used by this descriptor           */
} annp_hip_params;

/* Replaces annp_gpu_init.  device = HIP device ordinal.  nlocal/nall/max_nbors are
 * sizing hints exactly as in the reference (buffers grow on demand). */
int annp_hip_init(annp_hip_handle **handle, const annp_hip_params *params, int device,
                  int nlocal_hint, int nall_hint, int max_nbors_hint);

/* Replaces annp_gpu_compute (neighbour list built by LAMMPS on the host).
 *   ago        neighbor->ago: 0 = list was rebuilt, re-upload it
 *   host_x     LAMMPS atom->x[0]: nall*3 doubles;  host_type: nall ints
 *   ilist/numj/firstneigh: LAMMPS NeighList (full list), numj and firstneigh indexed by atom
 *   f          atom->f[0]: nall*3 doubles, accumulated into
 *   eng_vdwl   accumulated when eflag;  eatom [nall] accumulated when eatom_flag (nullable)
 *   virial     6 doubles xx yy zz xy xz yz accumulated when vflag: the ev_tally_xyz
 *              contraction of fe_v2/src/pair_annp.cpp:201-209 (nullable)
 *   vatom      atom->vatom[0]: nall*6 doubles accumulated when vatom_flag (nullable): the per-atom
 *              share of the same tally, half of each pair term to i, half to j (ev_tally_xyz) */
int annp_hip_compute(annp_hip_handle *handle, int ago, int inum, int nall, int nghost,
                     const double *host_x, const int *host_type,
                     const int *ilist, const int *numj, const int *const *firstneigh,
                     int eflag, int vflag, int eatom_flag, int vatom_flag,
                     double *f, double *eng_vdwl, double *eatom, double *virial, double *vatom);

/* Replaces annp_gpu_compute_n (neighbour list built on the device from host_x
 * when ago == 0).  cutneigh = cutoff + skin.  sublo/subhi (the sub-domain bounds the reference passes on to its
 * cell grid) are accepted for signature compatibility and not read: the build bins the bounding box of the
 * positions it is given, ghosts included.  host_type is read when the potential distinguishes atom types.
 * Owned atoms are the first inum entries of host_x, ghosts follow.  Atomic
 * systems only (no special-bond exclusions), like the potential itself. */
int annp_hip_compute_n(annp_hip_handle *handle, int ago, int inum, int nall, int nghost,
                       const double *host_x, const int *host_type,
                       const double *sublo, const double *subhi, double cutneigh,
                       int eflag, int vflag, int eatom_flag, int vatom_flag,
                       double *f, double *eng_vdwl, double *eatom, double *virial, double *vatom);

/* Device-resident evaluation: every pointer is a device pointer on the handle's GPU.
 *   d_x [nall*3], d_type [nall] (LAMMPS types 1..ntypes; nullable when every type maps to the same
 *   element, required otherwise), d_ilist [inum] (nullable = 0..inum-1)
 *   neighbours of atom i: d_neigh[d_first[i] .. d_first[i]+d_numneigh[i])
 *   d_f [nall*3] accumulated;  d_eatom [nall] accumulated (nullable)
 *   d_eng: 1 double accumulated (nullable);  d_virial: 6 doubles accumulated (nullable)
 *   d_vatom: [nall*6] accumulated (nullable)
 *   stream: hipStream_t (NULL = default stream).
 * Asynchronous: every kernel and copy is enqueued on `stream` and the call returns without waiting for
 * the device (the calling thread's current device is left as it was).  The LDS record capacity of the
 * force pass comes from the in-cutoff maximum of the PREVIOUS evaluation on the handle, read back
 * without blocking; an atom that has more neighbours than that is queued on the device and evaluated by a
 * fix-up launch on the same stream (Chebyshev: of the force pass; Behler: of both passes, by groups of four
 * atoms), so the result is complete whatever the configuration does between two calls.  Two exceptions wait
 * for the device once: the first evaluation on a handle, and the one after a capacity error.
 * Device-side capacity errors (more in-range neighbours than the largest LDS records hold: ~96 per atom for the
 * Behler kernels, 128 for anna_adp; any: a list row longer than max_numneigh) skip the affected atoms and are
 * reported as ANNP_HIP_ENEIGHCAP by the NEXT call on the handle or by annp_hip_sync, however many
 * evaluations were enqueued in between (the error word stays set on the device until it was seen). */
int annp_hip_compute_device(annp_hip_handle *handle, int inum, int nall,
                            const double *d_x, const int *d_type, const int *d_ilist,
                            const int *d_numneigh, const long long *d_first, const int *d_neigh,
                            int max_numneigh,
                            double *d_f, double *d_eatom, double *d_eng, double *d_virial, double *d_vatom,
                            void *stream);

/* Device neighbour-list build (binned, full list, r^2 <= c^2 with c = annp_hip_list_cutoff(handle, cutneigh):
 * cutneigh itself except for Behler potentials) for the first nlocal of nall atoms at d_x.  The list lives in handle-owned memory and
 * stays valid until the next build or annp_hip_clear.
 * From the second rebuild of a handle on the call does not wait for the device: the bins of the last build that
 * looked at its bounding box are used again, rows lie a fixed pitch apart, *max_numneigh is that pitch (an upper bound of the
 * longest row -- NOT the longest row, as it is after a build that waits) and the longest row itself is looked at, without waiting,
 * by the evaluations on the list (annp_hip_compute_device: the first one that finds the word landed, i.e. a step or two after the
 * build), at the latest by the NEXT build or annp_hip_sync.  A row that grew beyond the pitch in
 * between (more than 6 % + 4 entries from one rebuild to the next) was cut on the device -- nothing is indexed beyond a row -- and
 * is reported then as ANNP_HIP_ENEIGHCAP (the evaluations enqueued before that missed the entries beyond the pitch); the build after
 * that is an exact one.  ANNP_HIP_NEIGH_SYNC=1 in the environment at init
 * makes every build wait and check before it returns, as the builds inside annp_hip_compute_n always do. */
int annp_hip_neigh_build_device(annp_hip_handle *handle, int nlocal, int nall, const double *d_x,
                                double cutneigh,
                                const int **d_numneigh, const long long **d_first, const int **d_neigh,
                                int *max_numneigh, void *stream);

/* The cutoff a list built by this library for `cutneigh` = cutmax + skin is really cut at.  Chebyshev and anna_adp:
 * cutneigh.  Behler (gradient-consistent derivative): max Rc of the symmetry functions / 1.889726 + the same skin
 * (5.9 A instead of 8.5 A for the shipped Ni file): beyond Rc a neighbour contributes exactly nothing
 * (ni/src/pair_annp.cpp:693, 729), so forces and energies are those of the long list up to the order of the sums while
 * the kernels filter a third of the candidates.  ANNP_HIP_FULL_LIST=1 in the environment at init keeps cutneigh; so
 * does ni_compat = 1, whose derivative depends on the order of the list (ni:737-738). */
double annp_hip_list_cutoff(const annp_hip_handle *handle, double cutneigh);

/* Layout of the list the device built last: info4[0] 1 = rows a fixed pitch apart (a rebuild that reused the previous
 * build's pitch: one pass over the candidates), 0 = exact CSR (first build, or rows too uneven for a pitch);
 * [1] the pitch; [2] longest row; [3] atoms. */
int annp_hip_list_layout(const annp_hip_handle *handle, int *info4);

/* Copies the list the device built last (annp_hip_compute_n / annp_hip_neigh_build_device) to the host, rows packed in
 * atom order: numneigh[nlocal], first[nlocal+1] (nullable; offsets into neigh), neigh[neigh_capacity] (nullable: only
 * counts and *total).  This is what lets annp_gpu_compute_n hand LAMMPS a firstneigh (lal_base_annp.cpp:159-175).
 * Blocks until the device is idle. */
int annp_hip_neigh_to_host(annp_hip_handle *handle, int nlocal, int *numneigh, long long *first, int *neigh,
                           long long neigh_capacity, long long *total);

/* ---- the MD step around the evaluation, for callers that keep atoms in HBM --------------------------------------
 * What LAMMPS core does around Pair::compute for the reference (Comm::forward_comm / reverse_comm with newton_pair on --
 * fe_v2/src/pair_annp.cpp:199 writes ghost forces and relies on them -- and FixNVE's two half steps), as one kernel
 * each.  Device pointers, rows of 3 doubles, int32 indices; asynchronous on `stream`; the wire itself (RCCL
 * send/recv between slab neighbours) stays with the caller.  Deterministic: no atomics, fixed summation order.
 *
 * annp_hip_halo_pack           out[k] = x[idx[k]] + shift[k], k < n: the forward pack of boundary atoms into a send
 *                              buffer (shift = the periodic image offset the receiving side sees).
 * annp_hip_halo_unpack_images  x[first_image_row + k] = x[root[k]] + shift[k], k < nimg: the local periodic images
 *                              (roots are owned atoms or ghosts received over the wire, never images).  The same
 *                              launch clears n_clear doubles at d_f_clear and the word at d_eng_clear (both nullable):
 *                              Verlet::force_clear for the evaluation that follows.
 * annp_hip_reverse_fold        f[seg_dst[s]] += sum_{k = seg_start[s]}^{seg_start[s+1]-1} src[perm[k]], s < nseg, k ascending:
 *                              ghost forces back onto their owners -- image rows of f onto their roots (src = f +
 *                              3 first_image_row), then the rows that came back over the wire onto the boundary atoms
 *                              (src = receive buffer, perm = position in it).  seg_start has nseg + 1 entries.
 * annp_hip_verlet_half         v += dtf f, then x += dt v if dt != 0 (FixNVE::initial_integrate; dt = 0: final_integrate);
 *                              dtf = 0.5 dt ftm2v / mass; product and sum rounded separately (no FMA contraction). */
int annp_hip_halo_pack(annp_hip_handle *handle, int n, const int *d_idx, const double *d_shift, const double *d_x,
                       double *d_out, void *stream);
int annp_hip_halo_unpack_images(annp_hip_handle *handle, int nimg, const int *d_root, const double *d_shift, double *d_x,
                                int first_image_row, double *d_f_clear, long long n_clear, double *d_eng_clear, void *stream);
int annp_hip_reverse_fold(annp_hip_handle *handle, int nseg, const int *d_seg_dst, const int *d_seg_start, const int *d_perm,
                          const double *d_src, double *d_f, void *stream);
/* Re-planning at a list rebuild, on the device -- what LAMMPS' Comm::exchange and Comm::borders do for the reference (the pair
 * style only sees their result: owned atoms first, ghosts behind them).  Every selection keeps index order, so the arrays are
 * those of meng_zhang_amd/domain.py's torch restatement bit for bit.  The wire stays with the caller.  Each call synchronises
 * the stream once per selection (the counts size the caller's messages and buffers).
 *   annp_hip_replan_exchange   wraps the n owned positions into the periodic box in place; with world > 1 also sorts the atoms
 *                              into stay / to-left / to-right (slab of `rank` of `world` along x; has_left / has_right: whether
 *                              that neighbour exists) and packs rows [x y z | id | e0 (w0 columns) | e1 (w1 columns)] of the
 *                              stayers into d_keep and of the leavers into d_send ([to-left | to-right]); counts3 = stay, left, right
 *   annp_hip_replan_unpack     m such rows (stayers + received) -> d_x, d_ids, d_e0, d_e1
 *   annp_hip_replan_faces      indices (int32, ascending) of the owned atoms with x < lo_edge (if has_left), then of those with
 *                              x >= hi_edge (if has_right); counts2 = their numbers
 *   annp_hip_replan_images     periodic images in the dimensions of dims_mask (bit d) of rows [0, np0) of d_x and of the images made
 *                              so far: positions appended behind row np0, d_root[k] / d_shift[k][3] = the row < np0 image k stems
 *                              from and its total shift (what annp_hip_halo_unpack_images wants).  ANNP_HIP_ENEIGHCAP with an
 *                              estimate in *nimg_out when capacity_rows is too small: come back with more
 *   annp_hip_replan_fold_plan  items k < m with targets d_targets[k] in [0, nkeys) grouped by target, k ascending:
 *                              d_start[nkeys + 1], d_perm[m] for annp_hip_reverse_fold with d_seg_dst = NULL (segment s = row s) */
int annp_hip_replan_exchange(annp_hip_handle *handle, int n, double *d_x, const long long *d_ids, const double *d_e0, int w0,
                             const double *d_e1, int w1, const double *box6, const int *periodic3, int world, int rank,
                             int has_left, int has_right, double *d_keep, double *d_send, int *counts3, void *stream);
int annp_hip_replan_unpack(annp_hip_handle *handle, int m, const double *d_rows, int w0, int w1, double *d_x, long long *d_ids,
                           double *d_e0, double *d_e1, void *stream);
int annp_hip_replan_faces(annp_hip_handle *handle, int n, const double *d_x, double lo_edge, double hi_edge, int has_left, int has_right,
                          int *d_idx, int *counts2, void *stream);
int annp_hip_replan_images(annp_hip_handle *handle, int np0, double *d_x, long long capacity_rows, const double *box6, const int *periodic3,
                           double rc_halo, int dims_mask, int *d_root, double *d_shift, int *nimg_out, void *stream);
int annp_hip_replan_fold_plan(annp_hip_handle *handle, int m, const int *d_targets, int nkeys, int *d_start, int *d_perm, void *stream);
int annp_hip_verlet_half(annp_hip_handle *handle, int n, double *d_x, double *d_v, const double *d_f, double dtf, double dt,
                         void *stream);

/* ---- the halo wire --------------------------------------------------------------------------------------------------
 * RCCL point-to-point between the ranks of a spatial decomposition (one process per GPU, one handle per process), called
 * from C++ on the caller's compute stream: what LAMMPS' Comm does over MPI for the reference (`processors 2 1 1` +
 * `package gpu 2`, annp-gpu-lammps/fe_v2/performance test.zip -> in.st_test:3-4).  librccl.so.1 is opened at run time
 * (dlopen): no link-time dependency, and a process that has it loaded already (torch) shares that copy.
 *   annp_hip_comm_unique_id  128 bytes (ncclUniqueId) made on one rank; the caller brings them to every rank (MPI_Bcast, ...)
 *   annp_hip_comm_init       ncclCommInitRank on the handle's device; collective over the `world` ranks
 *   annp_hip_comm_route      one ncclGroupStart .. ncclGroupEnd of nmsg transfers: message k sends (is_send[k] != 0) or
 *                            receives ndoubles[k] doubles at device pointer d_buf[k] to / from rank peer[k] (peer == own rank
 *                            is allowed: a slab that is its own periodic neighbour); asynchronous on `stream`
 *   annp_hip_comm_destroy    also done by annp_hip_clear */
int annp_hip_comm_unique_id(char *id128);
int annp_hip_comm_init(annp_hip_handle *handle, const char *id128, int world, int rank);
int annp_hip_comm_route(annp_hip_handle *handle, int nmsg, const int *is_send, double *const *d_buf, const long long *ndoubles,
                        const int *peer, void *stream);
int annp_hip_comm_destroy(annp_hip_handle *handle);

/* Blocks until the handle's enqueued work is done and reports deferred device-side
 * errors (e.g. ANNP_HIP_ENEIGHCAP). */
int annp_hip_sync(annp_hip_handle *handle);

/* Facts about the most recent evaluation (waits for its flag words only):
 *   info4[0] largest in-cutoff neighbour count   info4[1] atoms (Behler: groups of four atoms) that went through the fix-up launch
 *   info4[2] neighbours per atom its force pass had room for (Chebyshev, moment kernels: the state the descriptor and force
 *            passes share, 96..160; pair-loop kernels and Behler: LDS records)   info4[3] what the next evaluation will use */
int annp_hip_eval_info(annp_hip_handle *handle, int *info4);

/* Which kernels the NEXT evaluation will run (after the most recent one's flag words, which this waits for):
 *   0  Chebyshev passes on the moments of the neighbourhood (the fast path: up to 160 in-cutoff neighbours per atom)
 *   1  Chebyshev passes pair by pair because the system is denser than that (about half the speed; back to 0 by itself when
 *      the maximum falls under 160 again)
 *   2  Chebyshev passes pair by pair because ANNP_HIP_FE_DESC / ANNP_HIP_FE_FORCE ask for it (developer A/B switches)
 *   3  Behler G2/G4 kernels      4  pair_style anna_adp kernels
 * The change 0 -> 1 is also announced once on the stream given to annp_hip_set_notice (annp_gpu_init passes LAMMPS' screen).
 * So is one more thing a caller would otherwise only see in its timings: atoms in no spatial order.  The force pass collects
 * forces in a table whose buckets hold eight atoms with consecutive indices; a caller that sorts its atoms in space (LAMMPS:
 * atom_modify sort, the default) needs ~18 memory requests per atom for them, atoms in random order 113 and up to 2.3 times
 * the time of that pass (results are the same).  The notice appears when more than eight contributions per atom found no bucket. */
int annp_hip_eval_path(annp_hip_handle *handle);
/* `file` is a FILE * (or NULL: silent, the default).  One line per event, prefixed "annp/hip:".  The library keeps the pointer and
 * writes to it from whichever call notices the event: pass NULL (or clear the handle, as annp_gpu_clear does)
 * before closing the file. */
int annp_hip_set_notice(annp_hip_handle *handle, void *file);

/* Kernel timing with HIP events recorded on the stream the kernels are launched on.
 * annp_hip_set_timing(h, 1) starts recording (and resets the sample count); every
 * evaluation then records four events.  ms4 = milliseconds of [0] descriptor pass,
 * [1] network pass, [2] force pass, [3] whole evaluation (ANNA_ADP: [1] is empty, [2] is the one
 * kernel that holds network, ADP sums, energy and forces).
 *   annp_hip_last_timing   the most recent evaluation
 *   annp_hip_timing_stats  mean over the evaluations since enabling (the last 64 at most) */
int annp_hip_set_timing(annp_hip_handle *handle, int enable);
int annp_hip_last_timing(annp_hip_handle *handle, double *ms4);
int annp_hip_timing_stats(annp_hip_handle *handle, double *ms4_mean, int *nsamples);

/* In-cutoff neighbour counts of the last evaluation (one int per list entry ii),
 * copied to the host: the n that SURVEY.md 8d's flop formula is evaluated with. */
int annp_hip_last_counts(annp_hip_handle *handle, int *counts, int inum);

/* Descriptor rows of the last evaluation as the descriptor pass left them, 32 doubles per list entry ii, copied to the host.
 * Chebyshev (pair_style annp Fe, anna_adp): the raw sums of fe_v2/src/pair_annp.cpp:633-695 before normalisation, radial
 * in [0,9), angular T_0..T_18 in [9,28); Behler: the G2/G4 sums.  A diagnostic: the parity tests compare two descriptor kernels
 * through it. */
int annp_hip_last_descriptors(annp_hip_handle *handle, double *rows, int inum);

/* Replaces annp_gpu_clear: frees everything; the handle is invalid afterwards. NULL ok. */
void annp_hip_clear(annp_hip_handle *handle);

/* Replaces annp_gpu_bytes: bytes of device + pinned host memory held by the handle. */
double annp_hip_bytes(const annp_hip_handle *handle);

/* Text of the last error on this handle (or of the last failed init when NULL). */
const char *annp_hip_last_error(const annp_hip_handle *handle);

int annp_hip_abi_version(void);

/* HIP devices visible to this process (0 when there is none or the runtime cannot start) */
int annp_hip_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
