/* anna_oracle.h -- CPU oracle for pair_style anna_adp.
 *
 * TEST INFRASTRUCTURE ONLY, like annp_oracle.h: a plain-C restatement of
 * anna-gpu-lammps/bcc_fe/src/pair_anna_adp.cpp ("adp:" below).  Only tests/ may
 * call it; it is the checker, never the product.
 *
 * PARITY UNPINNED UPSTREAM: the reference ships neither tests nor a run log for
 * this pair style, and its translation unit needs LAMMPS core headers (absent),
 * so it cannot be built here.  The restatement is anchored on the reference's own
 * source (cited line by line) and on properties that must hold for it
 * (tests/test_anna_oracle.py): forces are the exact gradient of the energy with
 * the two network outputs (d2, q2) held fixed -- which is what adp:231-279
 * differentiates -- total force zero, translation invariance.
 */
#ifndef ANNA_ORACLE_H
#define ANNA_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ANNA_ORACLE_MAXSF 64
#define ANNA_ORACLE_MAXNOD 64
#define ANNA_ORACLE_MAXLAY 6
#define ANNA_ORACLE_MAXGP 32

/* ANNAPARA, adp (pair_anna_adp.h:58-67); single element */
typedef struct anna_oracle_pot {
    int nelements;
    int ntl, nhl, nnod, nout, nsf, npsf, ntsf, ngp;
    int flagsym;
    int flagact[ANNA_ORACLE_MAXLAY];
    double cut, mass;
    double e_base, e_scal;
    double gparams[ANNA_ORACLE_MAXGP];   /* A0 yy gamma C0 c1F c2F V0 b1 b2 delta r0 r1 hc d1 q1 d3 q3 */
    double W[ANNA_ORACLE_MAXLAY][ANNA_ORACLE_MAXNOD * ANNA_ORACLE_MAXSF];   /* row-major [nrow][ncol] */
    double B[ANNA_ORACLE_MAXLAY][ANNA_ORACLE_MAXNOD];
    char element[16];
} anna_oracle_pot;

/* Restates PairANNA_ADP::read_file (adp:392-566).  Returns 0 or <0. */
int anna_oracle_read_file(const char *path, int nelem_coeff, anna_oracle_pot *pot);

/* Restates PairANNA_ADP::compute (adp:71-298) for a LAMMPS-style full neighbour list in CSR form
 * (same conventions as annp_oracle_compute).  cutsq = LAMMPS cutsq[1][1] = cutmax^2.
 *   f (+=/-=), eatom (nullable, +=), eng (+=), virial[6] (nullable, +=, ev_tally_xyz contraction),
 *   vatom [nall*6] (nullable, +=, half to i half to j)
 *   Gout [inum*nsf], Lout [inum*nout] (nullable): descriptor and the network's local parameters
 *   frozen (nullable): [nout] values used INSTEAD of the network outputs for every atom (property tests) */
int anna_oracle_compute(const anna_oracle_pot *pot, int nall, const double *x,
                        int inum, const int *ilist, const int *numneigh,
                        const long long *first, const int *neigh, double cutsq,
                        double *f, double *eatom, double *eng, double *virial, double *vatom,
                        double *Gout, double *Lout, const double *frozen);

#ifdef __cplusplus
}
#endif
#endif
