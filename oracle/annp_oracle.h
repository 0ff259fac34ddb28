/* annp_oracle.h -- CPU oracle for the pair_style annp hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference CPU
 * pair style (annp-gpu-lammps/{fe,fe_v2,ni}/src/pair_annp.cpp).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it; it is the
 * checker, never the product.  The product path is the HIP library declared in
 * include/annp_hip.h and must fail loudly when that library is missing.
 *
 * Parity pin: the reference ships no tests.  This oracle is pinned
 * by (a) the reference's own published run log for its own data file
 * (fe_v2 "performance test.zip": log_relaxing_new.lammps:118-120, step-0
 * E_pair / force norm / force max of fe_st.dat) and (b) the perfect-lattice
 * energies recorded in SURVEY.md Appendix B.  See tests/test_oracle_pins.py.
 */
#ifndef ANNP_ORACLE_H
#define ANNP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ANNP_ORACLE_MAXSF 64     /* max symmetry functions            */
#define ANNP_ORACLE_MAXNOD 64    /* max nodes per hidden layer         */
#define ANNP_ORACLE_MAXLAY 6     /* max weight layers (ntl-1)          */

/* Parsed potential, restating Param_ANNP (fe_v2/src/pair_annp.h:53-62,
 * ni/src/pair_annp.h:54-64): the header of the file plus the network of ONE element. */
typedef struct annp_oracle_pot {
    int nelements;
    int ntl, nhl, nnod, nsf, npsf, ntsf;
    int flagsym;                        /* parsed from names: Ch->0, Be/BP->1, Cu->2 */
    int flagact[ANNP_ORACLE_MAXLAY];    /* li 0, hy 1, si 2, mo 3, ta 4 */
    int has_symcoef;                    /* 1 when a "#coef..." section was present (Ni) */
    double cut, mass;
    double e_scale, e_shift, e_atom;
    double norm0[ANNP_ORACLE_MAXSF];    /* line 11+ne: Fe sfnor_cov / Ni sf_min */
    double norm1[ANNP_ORACLE_MAXSF];    /* line 12+ne: Fe sfnor_avg / Ni sf_max */
    double W[ANNP_ORACLE_MAXLAY][ANNP_ORACLE_MAXNOD * ANNP_ORACLE_MAXSF]; /* row-major [nrow][ncol] */
    double B[ANNP_ORACLE_MAXLAY][ANNP_ORACLE_MAXNOD];
    double sym_rad[ANNP_ORACLE_MAXSF][3];   /* eta, Rs, Rc   */
    double sym_ang[ANNP_ORACLE_MAXSF][4];   /* eta, lambda, zeta, Rc */
    char element[16];
} annp_oracle_pot;

/* Files with several elements: pots[e] receives the shared header and the network of element e.  names = the
 * element names of the pair_coeff line (nelem_coeff of them).  by_name = 0 restates the reference parser, which stores
 * every weight block in element 0 (fe_v2/src/pair_annp.cpp:455: `type_elem` is re-declared for every line) and leaves
 * elements 1.. at zero; by_name = 1: a "#El" line selects the element of the blocks below it. */
int annp_oracle_read_file_elems(const char *path, int nelem_coeff, const char *const *names, int by_name,
                                annp_oracle_pot *pots, int maxpots);

/* kinds of arithmetic (which reference translation unit is restated) */
#define ANNP_ORACLE_FE        0   /* fe, fe_v2: Chebyshev descriptor, twisted tanh      */
#define ANNP_ORACLE_NI_COMPAT 1   /* ni: literal CPU file, including ni:737-738          */
#define ANNP_ORACLE_NI_FIXED  2   /* ni: derivative as in ni/lib/lal_annp.cu:409-414     */

/* evaluation strategy */
#define ANNP_ORACLE_LITERAL 0     /* materialised dG, forward-mode Jacobian, ref. op order */
#define ANNP_ORACLE_FAST    1     /* two-pass + reverse-mode; OpenMP over atoms            */

/* Restates read_file (fe_v2/src/pair_annp.cpp:332-523, ni/src/pair_annp.cpp:324-545).
 * nelem_coeff = number of distinct elements on the pair_coeff line (1).
 * Returns 0, or <0 on error. */
int annp_oracle_read_file(const char *path, int nelem_coeff, annp_oracle_pot *pot);

/* Restates PairANNP::compute (fe_v2/src/pair_annp.cpp:74-223, ni/src/pair_annp.cpp:74-205)
 * for a LAMMPS-style full neighbour list in CSR form.
 *   x[nall*3], f[nall*3] (f is accumulated into, += / -= as the reference does)
 *   ilist[inum], numneigh/first indexed by atom index i: neighbours of i are
 *   neigh[first[i] .. first[i]+numneigh[i])
 *   cutsq: LAMMPS cutsq[itype][jtype] (single value: cutmax^2)
 *   eatom (nullable, +=), eng (+=), virial[6] (nullable, += ; xx yy zz xy xz yz,
 *   the ev_tally_xyz contraction of fe_v2:201-209)
 *   Gout / dEdGout: nullable [inum*nsf] dumps of the normalised descriptor and dE/dG.
 *   ni_calls: how many times compute() has run on the object incl. this one (>=1);
 *             restates the in-place sf_max -= sf_min of ni:99-101.
 */
int annp_oracle_compute(const annp_oracle_pot *pot, int kind, int strategy,
                        int nall, const double *x,
                        int inum, const int *ilist, const int *numneigh,
                        const long long *first, const int *neigh,
                        double cutsq, int ni_calls,
                        double *f, double *eatom, double *eng, double *virial,
                        double *Gout, double *dEdGout, int nthreads);

/* annp_oracle_compute for a potential with several elements: type[nall] (LAMMPS types), map[ntypes+1]
 * (type -> element, -1 = not mapped: such atoms are neither neighbours nor centres). */
int annp_oracle_compute_types(const annp_oracle_pot *pots, int nelem, int kind, int strategy,
                              int nall, const double *x, const int *type, const int *map,
                              int inum, const int *ilist, const int *numneigh,
                              const long long *first, const int *neigh,
                              double cutsq, int ni_calls,
                              double *f, double *eatom, double *eng, double *virial, int nthreads);

/* Same evaluation, also tallying the per-atom virial vatom[nall*6] (+=) the way
 * ev_tally_xyz does with newton_pair on: half of each pair term to i, half to j. */
int annp_oracle_compute_vatom(const annp_oracle_pot *pot, int kind,
                              int nall, const double *x,
                              int inum, const int *ilist, const int *numneigh,
                              const long long *first, const int *neigh,
                              double cutsq, int ni_calls, double *vatom);

/* number of threads the FAST strategy will use when nthreads<=0 */
int annp_oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
