/* annp_oracle.c -- CPU oracle for the pair_style annp hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see annp_oracle.h).  Plain-C restatement of the
 * reference CPU pair style; every routine cites the reference lines it follows.
 * Shorthand:  fe:N  = annp-gpu-lammps/fe_v2/src/pair_annp.cpp:N
 *             ni:N  = annp-gpu-lammps/ni/src/pair_annp.cpp:N
 *
 * Two strategies are provided:
 *   LITERAL  one atom at a time, dG materialised per list slot, forward-mode
 *            Jacobian through the network, operations in the reference's order.
 *   FAST     same formulas, but (Fe) two passes with the chain rule applied
 *            before the neighbour sums, no dG storage, OpenMP over atoms.  It is
 *            validated against LITERAL in tests/ and is what gets timed as the
 *            CPU baseline ("port") in bench.py.
 */
#include "annp_oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MY_PI 3.14159265358979323846   /* LAMMPS math_const.h MY_PI */
#define NEIGHMASK 0x1FFFFFFF           /* LAMMPS lmptype.h (stable_2Aug2023) */
#define CFLENGTH 1.889726              /* ni/src/pair_annp.h:69 */
#define CFFORCE 51.422515              /* ni/src/pair_annp.h:70 */

/* ------------------------------------------------------------------------- */
/* potential file                                                            */
/* ------------------------------------------------------------------------- */

/* getline() equivalent that keeps '\r' like std::getline does on a CRLF file */
static int read_line(FILE *fp, char **buf, size_t *cap)
{
    size_t n = 0;
    int c;
    if (feof(fp)) return 0;
    while ((c = fgetc(fp)) != EOF) {
        if (c == '\n') break;
        if (n + 2 > *cap) {
            *cap = *cap ? *cap * 2 : 4096;
            *buf = (char *)realloc(*buf, *cap);
        }
        (*buf)[n++] = (char)c;
    }
    if (n + 1 > *cap) {
        *cap = *cap ? *cap * 2 : 4096;
        *buf = (char *)realloc(*buf, *cap);
    }
    (*buf)[n] = '\0';
    if (c == EOF && n == 0) return 0;
    return 1;
}

/* "first value at column 0, then one value after every TAB that is followed by
 * a digit or '-'" : fe:393-407, fe:493-499 */
static int parse_row(const char *s, double *out, int maxn, int first_at_zero)
{
    int n = 0;
    size_t len = strlen(s);
    if (first_at_zero && n < maxn) out[n++] = atof(s);
    for (size_t j = 0; j + 1 <= len; j++) {
        char nx = s[j + 1];
        if (s[j] == '\t' && (isdigit((unsigned char)nx) || nx == '-')) {
            if (n < maxn) out[n] = atof(s + j + 1);
            n++;
        }
    }
    return n;
}

static int read_file_impl(const char *path, int nelem_coeff, const char *const *names, int by_name,
                          annp_oracle_pot *pots, int maxpots);

int annp_oracle_read_file(const char *path, int nelem_coeff, annp_oracle_pot *pot)
{
    return read_file_impl(path, nelem_coeff, NULL, 0, pot, 1);
}

int annp_oracle_read_file_elems(const char *path, int nelem_coeff, const char *const *names, int by_name,
                                annp_oracle_pot *pots, int maxpots)
{
    return read_file_impl(path, nelem_coeff, names, by_name, pots, maxpots);
}

/* pots[e] receives the header (identical for all) and the network of element e */
static int read_file_impl(const char *path, int nelem_coeff, const char *const *names, int by_name,
                          annp_oracle_pot *pots, int maxpots)
{
    FILE *fp = fopen(path, "rb");
    char *line = NULL;
    size_t cap = 0;
    annp_oracle_pot *pot = pots;
    if (!fp) return -1;
    memset(pots, 0, sizeof(*pots) * (size_t)maxpots);

    /* header: fe:347-434 (ni:339-430 identical apart from field names) */
    int ne = 0;
    for (int i = 0; i < 21 + nelem_coeff; i++) {
        if (!read_line(fp, &line, &cap)) { fclose(fp); free(line); return -2; }
        size_t len = strlen(line);
        if (i == 5) {
            pot->nelements = ne = atoi(line);
            if (ne < 1 || ne > maxpots) { fclose(fp); free(line); return -3; }
        }
        if (i >= 6 && i < 6 + ne) {                               /* fe:354-366 */
            int p = 0;
            annp_oracle_pot *q = pots + (i - 6);
            for (size_t j = 0; j < len; j++) {
                if (isalpha((unsigned char)line[j]) && p < 15) q->element[p++] = line[j];
                if (line[j] == '\t' && isdigit((unsigned char)line[j + 1])) q->mass = atof(line + j + 1);
            }
            q->element[p] = 0;
        }
        if (i == 8 + ne) {                                        /* fe:367-384 */
            int np = 1;
            pot->ntl = atoi(line);
            for (size_t j = 0; j < len; j++) {
                if (line[j] == '\t' && isdigit((unsigned char)line[j + 1])) {
                    const char *v = line + j + 1;
                    if (np == 1) pot->nhl = atoi(v);
                    if (np == 2) pot->nnod = atoi(v);
                    if (np == 3) pot->nsf = atoi(v);
                    if (np == 4) pot->npsf = atoi(v);
                    if (np == 5) pot->ntsf = atoi(v);
                    if (np == 6) pot->cut = atof(v);
                    np++;
                }
            }
            if (pot->nsf > ANNP_ORACLE_MAXSF || pot->nnod > ANNP_ORACLE_MAXNOD ||
                pot->ntl - 1 > ANNP_ORACLE_MAXLAY || pot->nsf < 1) {
                fclose(fp); free(line); return -4;
            }
        }
        if (i == 11 + ne) parse_row(line, pot->norm0, pot->nsf, 1);   /* fe:390-408 */
        if (i == 12 + ne) parse_row(line, pot->norm1, pot->nsf, 1);
        if (i == 15 + ne) {                                        /* fe:409-425 */
            int nact = 0;
            for (size_t j = 0; j + 1 < len + 1; j++) {
                char a = line[j], b = line[j + 1];
                if (a == 'C' && b == 'h') pot->flagsym = 0;
                if ((a == 'B' && b == 'e') || (a == 'B' && b == 'P')) pot->flagsym = 1;
                if (a == 'C' && b == 'u') pot->flagsym = 2;
                int act = -1;
                if (a == 'l' && b == 'i') act = 0;
                if (a == 'h' && b == 'y') act = 1;
                if (a == 's' && b == 'i') act = 2;
                if (a == 'm' && b == 'o') act = 3;
                if (a == 't' && b == 'a') act = 4;
                if (act >= 0 && nact < ANNP_ORACLE_MAXLAY) pot->flagact[nact++] = act;
            }
        }
        if (i == 18 + ne) pot->e_scale = atof(line);               /* fe:426-433 */
        if (i == 19 + ne) pot->e_shift = atof(line);
        if (i == 20 + ne) pot->e_atom = atof(line);
    }

    /* every element starts from the same header and a zero network (c_3d_matrix memsets, fe:441-448) */
    for (int e = 1; e < ne; e++) {
        char name[16];
        double mass = pots[e].mass;
        memcpy(name, pots[e].element, sizeof(name));
        pots[e] = pots[0];
        memcpy(pots[e].element, name, sizeof(name));
        pots[e].mass = mass;
    }
    /* weight / bias blocks: fe:450-517; ni:445-512 stops at "#coef".
     * Which element a block goes to: fe:455 declares `int type_elem = 0;` inside the line loop, so the match made on
     * a "#El" line (fe:457-466) is gone when the next line ("#k_(weight)") is read -- EVERY block is stored in
     * element 0 (the file's last one wins), elements 1.. keep zero weights.  (SURVEY.md 8a's "#Fe\r never equals Fe"
     * is a second reason the match would fail; it never gets that far.)  by_name = 0 restates exactly that;
     * by_name = 1 lets a "#El" line select the element of the blocks below it (test-only alternative, mirrored by the
     * product's set_blocks_by_name). */
    int cur = 0;
    while (read_line(fp, &line, &cap)) {
        if (strncmp(line, "#coef", 5) == 0) { pot->has_symcoef = 1; break; }
        if (by_name && names && line[0] == '#' && isupper((unsigned char)line[1])) {
            char nm[32];
            int q = 0;
            for (size_t i = 0; line[i] && q < 31; i++)
                if (line[i] != '#' && line[i] != '\r' && line[i] != '\n' && line[i] != '\t' && line[i] != ' ') nm[q++] = line[i];
            nm[q] = 0;
            for (int k = 0; k < nelem_coeff && k < ne; k++) if (strcmp(nm, names[k]) == 0) cur = k;
            continue;
        }
        pot = pots + cur;
        if (line[0] == '#' && isdigit((unsigned char)line[1])) {
            int no_layer = 0, flag_wb = 0;
            for (size_t i = 0; line[i]; i++) {
                if (line[i] >= '0' && line[i] <= '9') no_layer = no_layer * 10 + (line[i] - '0');
                if (line[i] == 'w') flag_wb = 0;
                if (line[i] == 'b') flag_wb = 1;
            }
            int nrow_w = pot->nnod, ncol_w = pot->nnod;
            if (no_layer == 1) ncol_w = pot->nsf;
            if (no_layer == pot->ntl - 1) nrow_w = 1;
            int nol = no_layer - 1;
            if (nol < 0 || nol >= pot->ntl - 1) { fclose(fp); free(line); return -5; }
            if (flag_wb == 0) {
                for (int r = 0; r < nrow_w; r++) {
                    if (!read_line(fp, &line, &cap)) { fclose(fp); free(line); return -6; }
                    parse_row(line, pot->W[nol] + (size_t)r * ncol_w, ncol_w, 1);
                }
            } else {
                if (!read_line(fp, &line, &cap)) { fclose(fp); free(line); return -6; }
                parse_row(line, pot->B[nol], pot->nnod, 1);
            }
        }
    }
    if (pot->has_symcoef) {                                        /* ni:524-545 */
        pots[0].has_symcoef = 1;
        pot = pots;
        read_line(fp, &line, &cap);                                /* "#rad n" */
        for (int i = 0; i < pot->npsf; i++) {
            read_line(fp, &line, &cap);
            parse_row(line, pot->sym_rad[i], 3, 0);
        }
        read_line(fp, &line, &cap);                                /* "#angl n" */
        for (int i = 0; i < pot->ntsf; i++) {
            read_line(fp, &line, &cap);
            parse_row(line, pot->sym_ang[i], 4, 0);
        }
    }
    for (int e = 1; e < ne; e++) {                                 /* the coefficient section belongs to the file */
        pots[e].has_symcoef = pots[0].has_symcoef;
        memcpy(pots[e].sym_rad, pots[0].sym_rad, sizeof(pots[0].sym_rad));
        memcpy(pots[e].sym_ang, pots[0].sym_ang, sizeof(pots[0].sym_ang));
    }
    fclose(fp);
    free(line);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* small pieces shared by both strategies                                    */
/* ------------------------------------------------------------------------- */

/* fe:590-594 */
static inline void annp_fc(double rij, double Rc, double *fc, double *dfc)
{
    double coeff_a = MY_PI / Rc * rij;
    *fc = 0.5 * (cos(coeff_a) + 1);
    *dfc = -0.5 * MY_PI / Rc * sin(coeff_a);
}

/* fe:596-611 */
static inline void annp_Tx(double x, int n, double *Tx, double *dTx)
{
    for (int i = 0; i < n; i++) {
        if (i == 0) { Tx[i] = 1; dTx[i] = 0; }
        else if (i == 1) { Tx[i] = x; dTx[i] = 1; }
        else {
            Tx[i] = 2 * x * Tx[i - 1] - Tx[i - 2];
            dTx[i] = 2 * Tx[i - 1] + 2 * x * dTx[i - 1] - dTx[i - 2];
        }
    }
}

/* fe:618-628 (called with r, not r^2, despite the parameter names) */
static inline void annp_dct_djk(double rij, double rik, const double *xij, const double *xik,
                                double cos_theta, double *dct_dj, double *dct_dk)
{
    double B = rij * rik;
    double term1 = cos_theta / (rij * rij);
    double term2 = cos_theta / (rik * rik);
    for (int i = 0; i < 3; i++) {
        dct_dj[i] = (-1.0) * xik[i] / B + term1 * xij[i];
        dct_dk[i] = (-1.0) * xij[i] / B + term2 * xik[i];
    }
}

/* activation and its derivative.  fe:709-739; the ni file treats 3 and 4 as
 * plain tanh (ni:781-808). */
static inline void annp_act(int flag, int ni_variant, double a, double *h, double *dh)
{
    const double coeff_a = 1.7159, coeff_b = 0.666666666666667, coeff_c = 0.1;
    double t;
    switch (flag) {
    case 0: *h = a; *dh = 1; break;
    case 1: *h = tanh(a); *dh = 1 - (*h) * (*h); break;
    case 2: *h = 1.0 / (1.0 + exp(a)); *dh = (*h) * (1 - (*h)); break;
    case 3:
        if (ni_variant) { t = tanh(a); *h = t; *dh = 1.0 - t * t; }
        else { t = tanh(coeff_b * a); *h = coeff_a * t; *dh = coeff_a * (1.0 - t * t) * coeff_b; }
        break;
    default:
        if (ni_variant) { t = tanh(a); *h = t; *dh = 1.0 - t * t; }
        else { t = tanh(coeff_b * a); *h = coeff_a * t + coeff_c * a; *dh = coeff_a * (1.0 - t * t) * coeff_b + coeff_c; }
        break;
    }
}

static inline void layer_dims(const annp_oracle_pot *p, int l, int *nr, int *nc)
{
    *nr = p->nnod; *nc = p->nnod;
    if (l == 0) *nc = p->nsf;
    if (l == p->ntl - 2) *nr = 1;
}

/* annp_feed_forward, fe:741-804 / ni:810-867: forward pass carrying the full
 * Jacobian d(layer)/dG ("tdE_dG"), products accumulated in the reference order. */
static void feed_forward_literal(const annp_oracle_pot *p, int ni_variant, const double *G,
                                 double *dE_dG, double *out)
{
    int nsf = p->nsf, nnod = p->nnod, nl = p->ntl - 1;
    int nmax = nsf > nnod ? nsf : nnod;
    double *J = (double *)calloc((size_t)nmax * nsf, sizeof(double));
    double *J1 = (double *)calloc((size_t)nmax * nsf, sizeof(double));
    double *hdw = (double *)calloc((size_t)nnod * nmax, sizeof(double));
    double hprev[ANNP_ORACLE_MAXSF], h[ANNP_ORACLE_MAXNOD] = {0}, hd[ANNP_ORACLE_MAXNOD];
    for (int i = 0; i < nsf; i++) J[i * nsf + i] = 1.0;
    for (int i = 0; i < nsf; i++) hprev[i] = G[i];
    for (int l = 0; l < nl; l++) {
        int nr, nc;
        layer_dims(p, l, &nr, &nc);
        const double *W = p->W[l];
        for (int r = 0; r < nr; r++) {                 /* dot_add_wxb fe:700-707 */
            double a = 0.0;
            for (int c = 0; c < nc; c++) a += W[r * nc + c] * hprev[c];
            a += p->B[l][r];
            annp_act(p->flagact[l], ni_variant, a, &h[r], &hd[r]);
        }
        /* hidly_dw = diag(hd) * W  (dot_mat_2d over a diagonal matrix: exact) */
        for (int r = 0; r < nr; r++)
            for (int c = 0; c < nc; c++) hdw[r * nc + c] = hd[r] * W[r * nc + c];
        /* tdE_dG1 = hidly_dw * tdE_dG, k ascending: fe:822-833 */
        for (int r = 0; r < nr; r++)
            for (int j = 0; j < nsf; j++) {
                double t = 0.0;
                for (int k = 0; k < nc; k++) t += hdw[r * nc + k] * J[k * nsf + j];
                J1[r * nsf + j] = t;
            }
        for (int r = 0; r < nr; r++)
            for (int j = 0; j < nsf; j++) J[r * nsf + j] = J1[r * nsf + j];
        for (int r = 0; r < nr; r++) hprev[r] = h[r];
    }
    *out = h[0];
    for (int i = 0; i < nsf; i++) dE_dG[i] = J[i];
    free(J); free(J1); free(hdw);
}

/* Same network, reverse mode (used by FAST). */
static void feed_forward_reverse(const annp_oracle_pot *p, int ni_variant, const double *G,
                                 double *dE_dG, double *out)
{
    int nl = p->ntl - 1;
    double h[ANNP_ORACLE_MAXLAY + 1][ANNP_ORACLE_MAXSF];
    double hd[ANNP_ORACLE_MAXLAY][ANNP_ORACLE_MAXNOD];
    double delta[ANNP_ORACLE_MAXSF], dprev[ANNP_ORACLE_MAXSF];
    for (int i = 0; i < p->nsf; i++) h[0][i] = G[i];
    for (int l = 0; l < nl; l++) {
        int nr, nc;
        layer_dims(p, l, &nr, &nc);
        for (int r = 0; r < nr; r++) {
            double a = 0.0;
            for (int c = 0; c < nc; c++) a += p->W[l][r * nc + c] * h[l][c];
            a += p->B[l][r];
            annp_act(p->flagact[l], ni_variant, a, &h[l + 1][r], &hd[l][r]);
        }
    }
    *out = h[nl][0];
    delta[0] = 1.0;
    for (int l = nl - 1; l >= 0; l--) {
        int nr, nc;
        layer_dims(p, l, &nr, &nc);
        for (int c = 0; c < nc; c++) dprev[c] = 0.0;
        for (int r = 0; r < nr; r++) {
            double d = delta[r] * hd[l][r];
            for (int c = 0; c < nc; c++) dprev[c] += d * p->W[l][r * nc + c];
        }
        for (int c = 0; c < nc; c++) delta[c] = dprev[c];
    }
    for (int i = 0; i < p->nsf; i++) dE_dG[i] = delta[i];
}

/* Fe normalisation scale, fe:98-108 */
static void fe_sf_scale(const annp_oracle_pot *p, double *sf_scale)
{
    for (int i = 0; i < p->nsf; i++) {
        double t_avg = p->norm1[i];
        double t_scale = sqrt(p->norm0[i] - t_avg * t_avg);
        sf_scale[i] = (t_scale <= 1.0e-10) ? 0.0 : 1.0 / t_scale;
    }
}

/* ev_tally_xyz with newton_pair on, global virial only (LAMMPS pair.cpp) */
static inline void tally_virial(double *v, double fx, double fy, double fz,
                                double dx, double dy, double dz)
{
    v[0] += dx * fx; v[1] += dy * fy; v[2] += dz * fz;
    v[3] += dx * fy; v[4] += dx * fz; v[5] += dy * fz;
}

/* ------------------------------------------------------------------------- */
/* LITERAL strategy, Fe : fe:110-220                                         */
/* ------------------------------------------------------------------------- */
static void fe_atom_literal(const annp_oracle_pot *p, const double *sf_scale, double cutsq,
                            int i, int jnum, const int *jlist, const double *x,
                            double *dG /* [jnum][nsf][3] scratch */,
                            double *f, double *evdwl, double *virial, double *Gout, double *dEdGout)
{
    int nsf = p->nsf, npsf = p->npsf, ntsf = p->ntsf;
    double G[ANNP_ORACLE_MAXSF], dE_dG[ANNP_ORACLE_MAXSF];
    double Tx[ANNP_ORACLE_MAXSF], dTx[ANNP_ORACLE_MAXSF];
    double xtmp = x[3 * i], ytmp = x[3 * i + 1], ztmp = x[3 * i + 2];
    memset(G, 0, sizeof(G));
    memset(dG, 0, sizeof(double) * (size_t)jnum * nsf * 3);

    for (int jj = 0; jj < jnum; jj++) {
        int j = jlist[jj] & NEIGHMASK;
        double xij[3], rij_unit[3], dr_dj[3], fcij, dfcij;
        xij[0] = xtmp - x[3 * j]; xij[1] = ytmp - x[3 * j + 1]; xij[2] = ztmp - x[3 * j + 2];
        double rsqij = xij[0] * xij[0] + xij[1] * xij[1] + xij[2] * xij[2];
        if (rsqij > cutsq || rsqij < 1.0e-12) continue;
        const double rijinv = 1.0 / sqrt(xij[0] * xij[0] + xij[1] * xij[1] + xij[2] * xij[2]);
        for (int m = 0; m < 3; m++) rij_unit[m] = rijinv * xij[m];
        double rij = sqrt(rsqij);
        double Rc = sqrt(cutsq);
        annp_fc(rij, Rc, &fcij, &dfcij);
        for (int m = 0; m < 3; m++) dr_dj[m] = -1.0 * xij[m] / rij;       /* fe:613-616 */

        {   /* annp_symmetry_pair fe:633-656 */
            double Rcp = p->cut;
            double xr = 2 * rij / Rcp - 1;
            annp_Tx(xr, npsf, Tx, dTx);
            for (int m = 0; m < npsf; m++) {
                G[m] += sf_scale[m] * Tx[m] * fcij;
                double term1 = (dTx[m] * 2 / Rcp * fcij + Tx[m] * dfcij) * sf_scale[m];
                for (int n = 0; n < 3; n++) dG[((size_t)jj * nsf + m) * 3 + n] += term1 * dr_dj[n];
            }
        }
        for (int kk = jj + 1; kk < jnum; kk++) {
            int k = jlist[kk] & NEIGHMASK;
            double xik[3], rik_unit[3], dr_dk[3], dct_dj[3], dct_dk[3], fcik, dfcik;
            xik[0] = xtmp - x[3 * k]; xik[1] = ytmp - x[3 * k + 1]; xik[2] = ztmp - x[3 * k + 2];
            double rsqik = xik[0] * xik[0] + xik[1] * xik[1] + xik[2] * xik[2];
            if (rsqik > cutsq || rsqik < 1.0e-12) continue;
            const double rikinv = 1.0 / sqrt(xik[0] * xik[0] + xik[1] * xik[1] + xik[2] * xik[2]);
            for (int m = 0; m < 3; m++) rik_unit[m] = rikinv * xik[m];
            double cos_theta = rij_unit[0] * rik_unit[0] + rij_unit[1] * rik_unit[1] + rij_unit[2] * rik_unit[2];
            double rik = sqrt(rsqik);
            annp_fc(rik, sqrt(cutsq), &fcik, &dfcik);
            /* annp_symmetry_trip fe:658-695 */
            double xa = 0.5 * (cos_theta + 1);
            annp_Tx(xa, ntsf, Tx, dTx);
            for (int m = 0; m < 3; m++) dr_dk[m] = -1.0 * xik[m] / rik;
            annp_dct_djk(rij, rik, xij, xik, cos_theta, dct_dj, dct_dk);
            for (int n = 0; n < ntsf; n++) {
                double s = sf_scale[n + npsf];
                G[n + npsf] += s * Tx[n] * fcij * fcik;
                double term1 = dTx[n] * 0.5 * fcij * fcik;
                double term2 = Tx[n] * dfcij * fcik;
                double term3 = Tx[n] * fcij * dfcik;
                for (int m = 0; m < 3; m++) {
                    double t_dG_dj = term1 * dct_dj[m] + term2 * dr_dj[m];
                    double t_dG_dk = term1 * dct_dk[m] + term3 * dr_dk[m];
                    dG[((size_t)jj * nsf + n + npsf) * 3 + m] += s * t_dG_dj;
                    dG[((size_t)kk * nsf + n + npsf) * 3 + m] += s * t_dG_dk;
                }
            }
        }
    }
    for (int k = 0; k < nsf; k++) G[k] = G[k] - sf_scale[k] * p->norm1[k];     /* fe:178-180 */
    double out;
    feed_forward_literal(p, 0, G, dE_dG, &out);
    *evdwl = p->e_scale * out + p->e_shift + p->e_atom;                       /* fe:790-793 */
    if (Gout) memcpy(Gout, G, sizeof(double) * nsf);
    if (dEdGout) memcpy(dEdGout, dE_dG, sizeof(double) * nsf);

    double Fi[3] = {0, 0, 0};
    for (int jj = 0; jj < jnum; jj++) {                                      /* fe:190-213 */
        double Fj[3] = {0, 0, 0};
        int j = jlist[jj] & NEIGHMASK;
        for (int k = 0; k < 3; k++) {
            for (int n = 0; n < nsf; n++)
                Fj[k] += (-1.0) * dE_dG[n] * dG[((size_t)jj * nsf + n) * 3 + k] * p->e_scale;
            Fi[k] += Fj[k];
            f[3 * j + k] += Fj[k];
        }
        if (virial)
            tally_virial(virial, -Fj[0], -Fj[1], -Fj[2],
                         x[3 * i] - x[3 * j], x[3 * i + 1] - x[3 * j + 1], x[3 * i + 2] - x[3 * j + 2]);
    }
    f[3 * i] -= Fi[0]; f[3 * i + 1] -= Fi[1]; f[3 * i + 2] -= Fi[2];
}

/* ------------------------------------------------------------------------- */
/* Ni (Behler G2/G4), ni:74-205, 686-767.  Works on whatever list it is given: */
/* LITERAL passes the full list, FAST passes the list filtered to r_m < Rc     */
/* (pairs beyond Rc contribute exactly nothing: ni:693, ni:729).               */
/* ------------------------------------------------------------------------- */
static void ni_atom(const annp_oracle_pot *p, int fixed, const double *sfden, int reverse,
                    int i, int jnum, const int *jlist, const double *x, double *dG,
                    double *f, double *evdwl, double *virial, double *Gout, double *dEdGout)
{
    int nsf = p->nsf, npsf = p->npsf, ntsf = p->ntsf;
    double G[ANNP_ORACLE_MAXSF], dE_dG[ANNP_ORACLE_MAXSF];
    double xtmp = x[3 * i], ytmp = x[3 * i + 1], ztmp = x[3 * i + 2];
    memset(G, 0, sizeof(G));
    memset(dG, 0, sizeof(double) * (size_t)jnum * nsf * 3);

    for (int jj = 0; jj < jnum; jj++) {
        int j = jlist[jj] & NEIGHMASK;
        double xij[3], rij_unit[3], dr_dj[3];
        xij[0] = xtmp - x[3 * j]; xij[1] = ytmp - x[3 * j + 1]; xij[2] = ztmp - x[3 * j + 2];
        double r2ij = xij[0] * xij[0] + xij[1] * xij[1] + xij[2] * xij[2];
        const double rijinv = 1.0 / sqrt(xij[0] * xij[0] + xij[1] * xij[1] + xij[2] * xij[2]);
        for (int m = 0; m < 3; m++) rij_unit[m] = rijinv * xij[m];
        double rij = sqrt(r2ij);
        for (int m = 0; m < 3; m++) dr_dj[m] = -1.0 * xij[m] / rij;
        {   /* annp_symmetry_pair ni:686-711 */
            double rij_m = rij * CFLENGTH;
            double Rc = p->sym_rad[0][2];
            if (rij_m < Rc) {
                for (int m = 0; m < npsf; m++) {
                    double fc, dfc;
                    double eta = p->sym_rad[m][0];
                    annp_fc(rij_m, Rc, &fc, &dfc);
                    double term1 = exp(-eta * rij_m * rij_m);
                    double term2 = term1 * (-fc * 2.0 * eta * rij_m + dfc);
                    G[m] += term1 * fc;
                    for (int n = 0; n < 3; n++) dG[((size_t)jj * nsf + m) * 3 + n] += term2 * dr_dj[n];
                }
            }
        }
        for (int kk = jj + 1; kk < jnum; kk++) {
            int k = jlist[kk] & NEIGHMASK;
            double xik[3], xjk[3], rik_unit[3];
            xik[0] = xtmp - x[3 * k]; xik[1] = ytmp - x[3 * k + 1]; xik[2] = ztmp - x[3 * k + 2];
            xjk[0] = x[3 * j] - x[3 * k]; xjk[1] = x[3 * j + 1] - x[3 * k + 1]; xjk[2] = x[3 * j + 2] - x[3 * k + 2];
            double r2ik = xik[0] * xik[0] + xik[1] * xik[1] + xik[2] * xik[2];
            double r2jk = xjk[0] * xjk[0] + xjk[1] * xjk[1] + xjk[2] * xjk[2];
            const double rikinv = 1.0 / sqrt(xik[0] * xik[0] + xik[1] * xik[1] + xik[2] * xik[2]);
            for (int m = 0; m < 3; m++) rik_unit[m] = rikinv * xik[m];
            double cos_theta = rij_unit[0] * rik_unit[0] + rij_unit[1] * rik_unit[1] + rij_unit[2] * rik_unit[2];
            double rik = sqrt(r2ik), rjk = sqrt(r2jk);
            /* annp_symmetry_trip ni:713-767 */
            double dct_dj[3], dct_dk[3], dr_dk[3], dr_djk[3];
            for (int m = 0; m < 3; m++) dr_dk[m] = -1.0 * xik[m] / rik;
            for (int m = 0; m < 3; m++) dr_djk[m] = 1.0 * xjk[m] / rjk;
            annp_dct_djk(rij, rik, xij, xik, cos_theta, dct_dj, dct_dk);
            double rij_m = rij * CFLENGTH, rik_m = rik * CFLENGTH, rjk_m = rjk * CFLENGTH;
            double r2sum = rij_m * rij_m + rik_m * rik_m + rjk_m * rjk_m;
            double Rc = p->sym_ang[0][3];
            if (rij_m < Rc && rik_m < Rc && rjk_m < Rc) {
                double fcij, fcik, fcjk, dfcij, dfcik, dfcjk;
                double term2_drj[3], term2_drk[3], term3_drj[3], term3_drk[3];
                annp_fc(rij_m, Rc, &fcij, &dfcij);
                annp_fc(rik_m, Rc, &fcik, &dfcik);
                annp_fc(rjk_m, Rc, &fcjk, &dfcjk);
                double term_fc = fcij * fcik * fcjk;
                /* ni:737-738 multiply dr_djk by rik_m; the reference GPU kernel
                 * (ni/lib/lal_annp.cu:409-414) and the true gradient use rjk_m. */
                double rjk_used = fixed ? rjk_m : rik_m;
                for (int m = 0; m < 3; m++) {
                    term2_drj[m] = 2.0 * (rij_m * dr_dj[m] + rjk_used * dr_djk[m]);
                    term2_drk[m] = 2.0 * (rik_m * dr_dk[m] - rjk_used * dr_djk[m]);
                    term3_drj[m] = fcik * (dfcij * dr_dj[m] * fcjk + fcij * dfcjk * dr_djk[m]);
                    term3_drk[m] = fcij * (dfcik * dr_dk[m] * fcjk - fcik * dfcjk * dr_djk[m]);
                }
                for (int n = 0; n < ntsf; n++) {
                    double eta = p->sym_ang[n][0], lambda = p->sym_ang[n][1], zeta = p->sym_ang[n][2];
                    double flag = (1 + lambda * cos_theta);
                    if (flag <= 0) continue;
                    double term_coe = pow(2, 1 - zeta);
                    double term_cot = term_coe * pow(flag, zeta);
                    double term_exp = exp(-eta * (r2sum));
                    G[n + npsf] += term_cot * term_exp * term_fc;
                    double term1 = lambda * term_cot * term_exp * term_fc * zeta / flag / CFLENGTH;
                    double term3 = term_cot * term_exp;
                    double term2 = term3 * term_fc * eta;
                    for (int m = 0; m < 3; m++) {
                        dG[((size_t)jj * nsf + n + npsf) * 3 + m] += term1 * dct_dj[m] - term2 * term2_drj[m] + term3 * term3_drj[m];
                        dG[((size_t)kk * nsf + n + npsf) * 3 + m] += term1 * dct_dk[m] - term2 * term2_drk[m] + term3 * term3_drk[m];
                    }
                }
            }
        }
    }
    for (int n = 0; n < nsf; n++) G[n] = (G[n] - p->norm0[n]) / sfden[n];         /* ni:168-170 */
    double out;
    if (reverse) feed_forward_reverse(p, 1, G, dE_dG, &out);
    else feed_forward_literal(p, 1, G, dE_dG, &out);
    *evdwl = out;                                                             /* ni:858-860 */
    if (Gout) memcpy(Gout, G, sizeof(double) * nsf);
    if (dEdGout) memcpy(dEdGout, dE_dG, sizeof(double) * nsf);

    double Fi[3] = {0, 0, 0};
    for (int jj = 0; jj < jnum; jj++) {                                       /* ni:180-203 */
        double Fj[3] = {0, 0, 0};
        int j = jlist[jj] & NEIGHMASK;
        for (int k = 0; k < 3; k++) {
            for (int n = 0; n < nsf; n++)
                Fj[k] += (-1.0) * dE_dG[n] * dG[((size_t)jj * nsf + n) * 3 + k] / sfden[n];
            Fi[k] += Fj[k] * CFFORCE;
            f[3 * j + k] += Fj[k] * CFFORCE;
        }
        if (virial)   /* the reference tallies the un-converted Fj (ni:190-198) */
            tally_virial(virial, -Fj[0], -Fj[1], -Fj[2],
                         x[3 * i] - x[3 * j], x[3 * i + 1] - x[3 * j + 1], x[3 * i + 2] - x[3 * j + 2]);
    }
    f[3 * i] -= Fi[0]; f[3 * i + 1] -= Fi[1]; f[3 * i + 2] -= Fi[2];
}

/* ------------------------------------------------------------------------- */
/* FAST strategy, Fe: identical formulas, chain rule applied before the sums   */
/* ------------------------------------------------------------------------- */
typedef struct { double e[3], r, fc, dfc; int j; } fe_nbr;   /* e = rij_unit = (xi-xj)/r */

static void fe_atom_fast(const annp_oracle_pot *p, const double *sf_scale, double cutsq,
                         int i, int jnum, const int *jlist, const double *x,
                         fe_nbr *nb, double *Fn /* [jnum][3] */,
                         double *fi_out, double *evdwl, double *virial, double *Gout, double *dEdGout)
{
    int nsf = p->nsf, npsf = p->npsf, ntsf = p->ntsf;
    double G[ANNP_ORACLE_MAXSF], dE_dG[ANNP_ORACLE_MAXSF], c[ANNP_ORACLE_MAXSF];
    double Tx[ANNP_ORACLE_MAXSF], dTx[ANNP_ORACLE_MAXSF];
    double xtmp = x[3 * i], ytmp = x[3 * i + 1], ztmp = x[3 * i + 2];
    double Rc = sqrt(cutsq), Rcp = p->cut;
    int n = 0;
    memset(G, 0, sizeof(G));
    for (int jj = 0; jj < jnum; jj++) {
        int j = jlist[jj] & NEIGHMASK;
        double d0 = xtmp - x[3 * j], d1 = ytmp - x[3 * j + 1], d2 = ztmp - x[3 * j + 2];
        double rsq = d0 * d0 + d1 * d1 + d2 * d2;
        if (rsq > cutsq || rsq < 1.0e-12) continue;
        double rinv = 1.0 / sqrt(rsq);
        nb[n].e[0] = rinv * d0; nb[n].e[1] = rinv * d1; nb[n].e[2] = rinv * d2;
        nb[n].r = sqrt(rsq);
        annp_fc(nb[n].r, Rc, &nb[n].fc, &nb[n].dfc);
        nb[n].j = j;
        n++;
    }
    /* pass 1: descriptor */
    for (int a = 0; a < n; a++) {
        annp_Tx(2 * nb[a].r / Rcp - 1, npsf, Tx, dTx);
        for (int m = 0; m < npsf; m++) G[m] += sf_scale[m] * Tx[m] * nb[a].fc;
        for (int b = a + 1; b < n; b++) {
            double ct = nb[a].e[0] * nb[b].e[0] + nb[a].e[1] * nb[b].e[1] + nb[a].e[2] * nb[b].e[2];
            double w = nb[a].fc * nb[b].fc;
            annp_Tx(0.5 * (ct + 1), ntsf, Tx, dTx);
            for (int m = 0; m < ntsf; m++) G[npsf + m] += sf_scale[npsf + m] * Tx[m] * w;
        }
    }
    for (int k = 0; k < nsf; k++) G[k] = G[k] - sf_scale[k] * p->norm1[k];
    double out;
    feed_forward_reverse(p, 0, G, dE_dG, &out);
    *evdwl = p->e_scale * out + p->e_shift + p->e_atom;
    if (Gout) memcpy(Gout, G, sizeof(double) * nsf);
    if (dEdGout) memcpy(dEdGout, dE_dG, sizeof(double) * nsf);
    for (int k = 0; k < nsf; k++) c[k] = p->e_scale * dE_dG[k] * sf_scale[k];

    /* pass 2: Fn[a] = sum_n c_n dG_n/dx_a  (so that F_a = -Fn[a]) */
    for (int a = 0; a < n; a++) Fn[3 * a] = Fn[3 * a + 1] = Fn[3 * a + 2] = 0.0;
    for (int a = 0; a < n; a++) {
        /* radial: dG_m/dx_a = (T'_m 2/Rc fc + T_m fc') s_m * dr_dj, dr_dj = -e */
        annp_Tx(2 * nb[a].r / Rcp - 1, npsf, Tx, dTx);
        double R = 0.0;
        for (int m = 0; m < npsf; m++) R += c[m] * (dTx[m] * 2 / Rcp * nb[a].fc + Tx[m] * nb[a].dfc);
        for (int m = 0; m < 3; m++) Fn[3 * a + m] += R * (-nb[a].e[m]);
        for (int b = a + 1; b < n; b++) {
            double ct = nb[a].e[0] * nb[b].e[0] + nb[a].e[1] * nb[b].e[1] + nb[a].e[2] * nb[b].e[2];
            annp_Tx(0.5 * (ct + 1), ntsf, Tx, dTx);
            double P = 0.0, dP = 0.0;
            for (int m = 0; m < ntsf; m++) { P += c[npsf + m] * Tx[m]; dP += c[npsf + m] * dTx[m]; }
            double t1 = dP * 0.5 * nb[a].fc * nb[b].fc;
            double t2 = P * nb[a].dfc * nb[b].fc;
            double t3 = P * nb[a].fc * nb[b].dfc;
            /* dct_dj = (-e_b + ct e_a)/r_a ... in the reference's sign convention
             * xij = r e: dct_dj = -xik/(rij rik) + ct xij/rij^2 */
            for (int m = 0; m < 3; m++) {
                double dct_da = (-nb[b].e[m] + ct * nb[a].e[m]) / nb[a].r;
                double dct_db = (-nb[a].e[m] + ct * nb[b].e[m]) / nb[b].r;
                Fn[3 * a + m] += t1 * dct_da + t2 * (-nb[a].e[m]);
                Fn[3 * b + m] += t1 * dct_db + t3 * (-nb[b].e[m]);
            }
        }
    }
    double Fi[3] = {0, 0, 0};
    for (int a = 0; a < n; a++) {
        for (int m = 0; m < 3; m++) Fi[m] += -Fn[3 * a + m];
        if (virial) {
            int j = nb[a].j;
            tally_virial(virial, Fn[3 * a], Fn[3 * a + 1], Fn[3 * a + 2],
                         x[3 * i] - x[3 * j], x[3 * i + 1] - x[3 * j + 1], x[3 * i + 2] - x[3 * j + 2]);
        }
    }
    /* caller scatters F_j = -Fn[a] to nb[a].j and fi_out (= -sum_j F_j) to i */
    fi_out[0] = -Fi[0]; fi_out[1] = -Fi[1]; fi_out[2] = -Fi[2];
    fi_out[3] = (double)n;     /* number of in-cutoff neighbours */
}

int annp_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int annp_oracle_compute(const annp_oracle_pot *pot, int kind, int strategy,
                        int nall, const double *x,
                        int inum, const int *ilist, const int *numneigh,
                        const long long *first, const int *neigh,
                        double cutsq, int ni_calls,
                        double *f, double *eatom, double *eng, double *virial,
                        double *Gout, double *dEdGout, int nthreads)
{
    int nsf = pot->nsf;
    int maxj = 0;
    (void)nall;
    for (int ii = 0; ii < inum; ii++)
        if (numneigh[ilist[ii]] > maxj) maxj = numneigh[ilist[ii]];
    if (maxj < 1) maxj = 1;

    double sfa[ANNP_ORACLE_MAXSF];
    if (kind == ANNP_ORACLE_FE) fe_sf_scale(pot, sfa);
    else {
        if (!pot->has_symcoef) return -10;
        if (ni_calls < 1) ni_calls = 1;
        for (int n = 0; n < nsf; n++) sfa[n] = pot->norm1[n] - ni_calls * pot->norm0[n];   /* ni:99-101 */
    }

    if (strategy == ANNP_ORACLE_LITERAL) {
        double *dG = (double *)malloc(sizeof(double) * (size_t)maxj * nsf * 3);
        for (int ii = 0; ii < inum; ii++) {
            int i = ilist[ii];
            double e = 0.0;
            if (kind == ANNP_ORACLE_FE)
                fe_atom_literal(pot, sfa, cutsq, i, numneigh[i], neigh + first[i], x, dG, f, &e, virial,
                                Gout ? Gout + (size_t)ii * nsf : NULL, dEdGout ? dEdGout + (size_t)ii * nsf : NULL);
            else
                ni_atom(pot, kind == ANNP_ORACLE_NI_FIXED, sfa, 0, i, numneigh[i], neigh + first[i], x, dG, f, &e, virial,
                        Gout ? Gout + (size_t)ii * nsf : NULL, dEdGout ? dEdGout + (size_t)ii * nsf : NULL);
            if (eng) *eng += e;
            if (eatom) eatom[i] += e;
        }
        free(dG);
        return 0;
    }

    /* FAST: atoms in parallel; forces scattered with atomics */
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    double etot = 0.0;
    double vtot[6] = {0, 0, 0, 0, 0, 0};
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads) reduction(+ : etot)
#endif
    {
        double vloc[6] = {0, 0, 0, 0, 0, 0};
        if (kind == ANNP_ORACLE_FE) {
            fe_nbr *nb = (fe_nbr *)malloc(sizeof(fe_nbr) * (size_t)maxj);
            double *Fn = (double *)malloc(sizeof(double) * (size_t)maxj * 3);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
            for (int ii = 0; ii < inum; ii++) {
                int i = ilist[ii];
                double e = 0.0, fi[4];
                fe_atom_fast(pot, sfa, cutsq, i, numneigh[i], neigh + first[i], x, nb, Fn, fi, &e,
                             virial ? vloc : NULL,
                             Gout ? Gout + (size_t)ii * nsf : NULL, dEdGout ? dEdGout + (size_t)ii * nsf : NULL);
                int n = (int)fi[3];
                for (int a = 0; a < n; a++)
                    for (int m = 0; m < 3; m++) {
#ifdef _OPENMP
#pragma omp atomic
#endif
                        f[3 * nb[a].j + m] += -Fn[3 * a + m];
                    }
                for (int m = 0; m < 3; m++) {
#ifdef _OPENMP
#pragma omp atomic
#endif
                    f[3 * i + m] += fi[m];
                }
                etot += e;
                if (eatom) eatom[i] += e;
            }
            free(nb); free(Fn);
        } else {
            int *jl = (int *)malloc(sizeof(int) * (size_t)maxj);
            double *dG = (double *)malloc(sizeof(double) * (size_t)maxj * nsf * 3);
            double *floc = NULL;
            double Rc = pot->sym_rad[0][2] > pot->sym_ang[0][3] ? pot->sym_rad[0][2] : pot->sym_ang[0][3];
            (void)floc;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
            for (int ii = 0; ii < inum; ii++) {
                int i = ilist[ii];
                const int *jlist = neigh + first[i];
                int n = 0;
                for (int jj = 0; jj < numneigh[i]; jj++) {
                    int j = jlist[jj] & NEIGHMASK;
                    double d0 = x[3 * i] - x[3 * j], d1 = x[3 * i + 1] - x[3 * j + 1], d2 = x[3 * i + 2] - x[3 * j + 2];
                    double r = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
                    if (r * CFLENGTH < Rc) jl[n++] = j;
                }
                double e = 0.0;
                /* forces go to a private buffer keyed by slot, then atomically out */
                double *fp = (double *)calloc((size_t)(n + 1) * 3, sizeof(double));
                {
                    /* run ni_atom on a compact private coordinate set: slot a -> index a+1, centre -> 0 */
                    double *xp = (double *)malloc(sizeof(double) * (size_t)(n + 1) * 3);
                    int *jp = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
                    for (int m = 0; m < 3; m++) xp[m] = x[3 * i + m];
                    for (int a = 0; a < n; a++) {
                        for (int m = 0; m < 3; m++) xp[3 * (a + 1) + m] = x[3 * jl[a] + m];
                        jp[a] = a + 1;
                    }
                    ni_atom(pot, kind == ANNP_ORACLE_NI_FIXED, sfa, 1, 0, n, jp, xp, dG, fp, &e,
                            virial ? vloc : NULL,
                            Gout ? Gout + (size_t)ii * nsf : NULL, dEdGout ? dEdGout + (size_t)ii * nsf : NULL);
                    free(xp); free(jp);
                }
                for (int a = 0; a <= n; a++) {
                    int tgt = a == 0 ? i : jl[a - 1];
                    for (int m = 0; m < 3; m++) {
#ifdef _OPENMP
#pragma omp atomic
#endif
                        f[3 * tgt + m] += fp[3 * a + m];
                    }
                }
                free(fp);
                etot += e;
                if (eatom) eatom[i] += e;
            }
            free(jl); free(dG);
        }
        if (virial) {
#ifdef _OPENMP
#pragma omp critical
#endif
            for (int m = 0; m < 6; m++) vtot[m] += vloc[m];
        }
    }
    if (eng) *eng += etot;
    if (virial) for (int m = 0; m < 6; m++) virial[m] += vtot[m];
    return 0;
}

/* Potentials with several elements: atom i is evaluated with the network of element map[type[i]]
 * (fe:767-768 `params->all_annp[itype]`), the descriptor is species-blind.  A type that is not mapped (map < 0) has
 * cutsq = 0 against everything (fe:144 `rsqij > cutsq[ritype][rjtype]` then rejects every pair with it): such atoms are
 * neither neighbours nor centres.  pots[e] = element e's network (annp_oracle_read_file_elems). */
int annp_oracle_compute_types(const annp_oracle_pot *pots, int nelem, int kind, int strategy,
                              int nall, const double *x, const int *type, const int *map,
                              int inum, const int *ilist, const int *numneigh,
                              const long long *first, const int *neigh,
                              double cutsq, int ni_calls,
                              double *f, double *eatom, double *eng, double *virial, int nthreads)
{
    long long tot = 0;
    for (int ii = 0; ii < inum; ii++) tot += numneigh[ilist[ii]];
    int *nn = (int *)calloc((size_t)nall, sizeof(int));
    long long *ff = (long long *)calloc((size_t)nall + 1, sizeof(long long));
    int *ng = (int *)malloc(sizeof(int) * (size_t)(tot > 0 ? tot : 1));
    int *il = (int *)malloc(sizeof(int) * (size_t)(inum > 0 ? inum : 1));
    if (!nn || !ff || !ng || !il) { free(nn); free(ff); free(ng); free(il); return -3; }
    long long w = 0;
    for (int ii = 0; ii < inum; ii++) {                       /* rows without the unmapped neighbours, order kept */
        const int i = ilist[ii];
        ff[i] = w;
        for (int jj = 0; jj < numneigh[i]; jj++) {
            const int jraw = neigh[first[i] + jj];
            if (map[type[jraw & NEIGHMASK]] >= 0) ng[w++] = jraw;
        }
        nn[i] = (int)(w - ff[i]);
    }
    int rc = 0;
    for (int e = 0; e < nelem && rc == 0; e++) {
        int m = 0;
        for (int ii = 0; ii < inum; ii++) if (map[type[ilist[ii]]] == e) il[m++] = ilist[ii];
        if (m > 0)
            rc = annp_oracle_compute(pots + e, kind, strategy, nall, x, m, il, nn, ff, ng, cutsq, ni_calls,
                                     f, eatom, eng, virial, NULL, NULL, nthreads);
    }
    free(nn); free(ff); free(ng); free(il);
    return rc;
}

/* Per-atom virial, LITERAL strategy re-run with a force probe: the pair term of (i, j) is
 * v = (x_i - x_j) (x) (-F_j^(i)) where F_j^(i) is what atom i's energy puts on neighbour j
 * (fe:201-209, ni:190-198).  F_j^(i) is obtained by evaluating atom i alone into a scratch
 * force array, so this routine shares every formula with annp_oracle_compute. */
int annp_oracle_compute_vatom(const annp_oracle_pot *pot, int kind,
                              int nall, const double *x,
                              int inum, const int *ilist, const int *numneigh,
                              const long long *first, const int *neigh,
                              double cutsq, int ni_calls, double *vatom)
{
    double *ftmp = (double *)calloc((size_t)nall * 3, sizeof(double));
    if (!ftmp) return -3;
    for (int ii = 0; ii < inum; ii++) {
        const int i = ilist[ii];
        int one = i;
        int rc = annp_oracle_compute(pot, kind, ANNP_ORACLE_LITERAL, nall, x, 1, &one, numneigh, first, neigh,
                                     cutsq, ni_calls, ftmp, NULL, NULL, NULL, NULL, NULL, 1);
        if (rc) { free(ftmp); return rc; }
        const int *jl = neigh + first[i];
        const double fscale = (kind == ANNP_ORACLE_FE) ? 1.0 : 1.0 / CFFORCE;   /* ni tallies the un-converted force */
        for (int jj = 0; jj < numneigh[i]; jj++) {
            const int j = jl[jj] & NEIGHMASK;
            const double fx = -ftmp[3 * j] * fscale, fy = -ftmp[3 * j + 1] * fscale, fz = -ftmp[3 * j + 2] * fscale;
            const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
            const double v[6] = {dx * fx, dy * fy, dz * fz, dx * fy, dx * fz, dy * fz};
            for (int k = 0; k < 6; k++) { vatom[6 * i + k] += 0.5 * v[k]; vatom[6 * j + k] += 0.5 * v[k]; }
            ftmp[3 * j] = ftmp[3 * j + 1] = ftmp[3 * j + 2] = 0.0;
        }
        ftmp[3 * i] = ftmp[3 * i + 1] = ftmp[3 * i + 2] = 0.0;
    }
    free(ftmp);
    return 0;
}
