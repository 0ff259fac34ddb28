/* anna_oracle.c -- CPU oracle for pair_style anna_adp.  TEST INFRASTRUCTURE ONLY (see anna_oracle.h).
 *
 * Restates anna-gpu-lammps/bcc_fe/src/pair_anna_adp.cpp ("adp:N" = line N of that file):
 * a Chebyshev descriptor (no normalisation) feeds a small network whose two outputs are the
 * decay constants d2, q2 of the dipole and quadrupole functions of an analytic ADP form; the
 * forces differentiate that form with d2, q2 held fixed.
 */
#define _GNU_SOURCE
#include "anna_oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MY_PI 3.14159265358979323846   /* LAMMPS MathConst::MY_PI */
#define NEIGHMASK 0x1FFFFFFF

/* std::getline on a CRLF file keeps the '\r' */
static int read_line(FILE *fp, char **buf, size_t *cap)
{
    size_t n = 0;
    int c;
    if (feof(fp)) return 0;
    while ((c = fgetc(fp)) != EOF) {
        if (c == '\n') break;
        if (n + 2 > *cap) { *cap = *cap ? *cap * 2 : 4096; *buf = (char *)realloc(*buf, *cap); }
        (*buf)[n++] = (char)c;
    }
    if (n + 1 > *cap) { *cap = *cap ? *cap * 2 : 4096; *buf = (char *)realloc(*buf, *cap); }
    (*buf)[n] = '\0';
    return !(c == EOF && n == 0);
}

/* one value at column 0, then one after every TAB followed by a digit or '-' (adp:475-481, 538-545) */
static int parse_row(const char *s, double *out, int maxn)
{
    int n = 0;
    if (n < maxn) out[n] = atof(s);
    n++;
    for (size_t j = 0; s[j]; j++) {
        char nx = s[j + 1];
        if (s[j] == '\t' && (isdigit((unsigned char)nx) || nx == '-')) {
            if (n < maxn) out[n] = atof(s + j + 1);
            n++;
        }
    }
    return n;
}

int anna_oracle_read_file(const char *path, int nelem_coeff, anna_oracle_pot *pot)
{
    FILE *fp = fopen(path, "rb");
    char *line = NULL;
    size_t cap = 0;
    int ne = 0;
    if (!fp) return -1;
    memset(pot, 0, sizeof(*pot));
    for (int i = 0; i < 19 + nelem_coeff; i++) {                       /* adp:406 */
        if (!read_line(fp, &line, &cap)) { fclose(fp); free(line); return -2; }
        size_t len = strlen(line);
        if (i == 5) {
            pot->nelements = ne = atoi(line);
            if (ne != 1) { fclose(fp); free(line); return -3; }
        }
        if (i >= 6 && i < 6 + ne) {                                    /* adp:413-425 */
            int p = 0;
            for (size_t j = 0; j < len; j++) {
                if (isalpha((unsigned char)line[j]) && p < 15) pot->element[p++] = line[j];
                if (line[j] == '\t' && isdigit((unsigned char)line[j + 1])) pot->mass = atof(line + j + 1);
            }
            pot->element[p] = 0;
        }
        if (i == 8 + ne) {                                             /* adp:426-443: TL HL nodes nout nsf npsf ntsf cut */
            int np = 1;
            pot->ntl = atoi(line);
            for (size_t j = 0; j < len; j++)
                if (line[j] == '\t' && isdigit((unsigned char)line[j + 1])) {
                    const char *v = line + j + 1;
                    if (np == 1) pot->nhl = atoi(v);
                    if (np == 2) pot->nnod = atoi(v);
                    if (np == 3) pot->nout = atoi(v);
                    if (np == 4) pot->nsf = atoi(v);
                    if (np == 5) pot->npsf = atoi(v);
                    if (np == 6) pot->ntsf = atoi(v);
                    if (np == 7) pot->cut = atof(v);
                    np++;
                }
            if (pot->nsf < 1 || pot->nsf > ANNA_ORACLE_MAXSF || pot->nnod < 1 || pot->nnod > ANNA_ORACLE_MAXNOD ||
                pot->ntl < 2 || pot->ntl - 1 > ANNA_ORACLE_MAXLAY || pot->nout < 1 || pot->nout > ANNA_ORACLE_MAXNOD) {
                fclose(fp); free(line); return -4;
            }
        }
        if (i == 11 + ne) {                                            /* adp:444-461: every two-character window */
            int nact = 0;
            for (size_t j = 0; j < len; j++) {
                char a = line[j], b = line[j + 1];
                if (a == 'C' && b == 'h') pot->flagsym = 0;
                if (a == 'B' && (b == 'e' || b == 'P')) pot->flagsym = 1;
                if (a == 'C' && b == 'u') pot->flagsym = 2;
                int act = -1;
                if (a == 'l' && b == 'i') act = 0;
                if (a == 'h' && b == 'y') act = 1;
                if (a == 's' && b == 'i') act = 2;
                if (a == 'm' && b == 'o') act = 3;
                if (a == 't' && b == 'a') act = 4;
                if (act >= 0 && nact < ANNA_ORACLE_MAXLAY) pot->flagact[nact++] = act;
            }
        }
        if (i == 14 + ne) {                                            /* adp:462-470 */
            pot->e_base = atof(line);
            for (size_t j = 0; j < len; j++)
                if (line[j] == '\t' && isdigit((unsigned char)line[j + 1])) pot->e_scal = atof(line + j + 1);
        }
        if (i == 17 + ne) {
            pot->ngp = atoi(line);
            if (pot->ngp < 17 || pot->ngp > ANNA_ORACLE_MAXGP) { fclose(fp); free(line); return -4; }
        }
        if (i == 18 + ne) parse_row(line, pot->gparams, pot->ngp);     /* adp:473-482 */
    }
    /* weight / bias blocks, adp:497-553: the last layer has nout rows */
    while (read_line(fp, &line, &cap)) {
        if (line[0] == '#' && isdigit((unsigned char)line[1])) {
            int no_layer = 0, flag_wb = 0;
            for (size_t i = 0; line[i]; i++) {
                if (line[i] >= '0' && line[i] <= '9') no_layer = no_layer * 10 + (line[i] - '0');
                if (line[i] == 'w') flag_wb = 0;
                if (line[i] == 'b') flag_wb = 1;
            }
            int nrow_w = pot->nnod, ncol_w = pot->nnod, ncol_b = pot->nnod;
            if (no_layer == 1) ncol_w = pot->nsf;
            if (no_layer == pot->ntl - 1) { nrow_w = pot->nout; ncol_b = pot->nout; }
            const int l = no_layer - 1;
            if (l < 0 || l >= pot->ntl - 1) { fclose(fp); free(line); return -5; }
            if (!flag_wb) {
                for (int r = 0; r < nrow_w; r++) {
                    if (!read_line(fp, &line, &cap)) { fclose(fp); free(line); return -6; }
                    parse_row(line, pot->W[l] + (size_t)r * ncol_w, ncol_w);
                }
            } else {
                if (!read_line(fp, &line, &cap)) { fclose(fp); free(line); return -6; }
                parse_row(line, pot->B[l], ncol_b);
            }
        }
    }
    fclose(fp);
    free(line);
    return 0;
}

/* adp:607-631 */
static void anna_act(int flag, int n, const double *wxb, double *h)
{
    const double coeff_a = 1.7, coeff_b = 0.3;
    for (int i = 0; i < n; i++) {
        if (flag == 0) h[i] = wxb[i];
        if (flag == 1) h[i] = tanh(wxb[i]);
        if (flag == 2) h[i] = 1.0 / (1.0 + exp(wxb[i]));
        if (flag == 3 || flag == 4) h[i] = coeff_a * tanh(coeff_b * wxb[i]);
    }
}

/* adp:633-667 */
static void anna_feed_forward(const anna_oracle_pot *p, const double *G, double *lparams)
{
    double h[2][ANNA_ORACLE_MAXNOD], wxb[ANNA_ORACLE_MAXNOD];
    const int nl = p->ntl - 1;
    for (int l = 0; l < nl; l++) {
        const int nr = (l == nl - 1) ? p->nout : p->nnod;
        const int nc = (l == 0) ? p->nsf : p->nnod;
        const double *in = (l == 0) ? G : h[(l - 1) & 1];
        for (int r = 0; r < nr; r++) {
            double a = 0.0;
            for (int c = 0; c < nc; c++) a += p->W[l][(size_t)r * nc + c] * in[c];   /* adp:599-604 */
            wxb[r] = a + p->B[l][r];
        }
        anna_act(p->flagact[l], nr, wxb, h[l & 1]);
    }
    for (int o = 0; o < p->nout; o++) lparams[o] = h[(nl - 1) & 1][o];
}

int anna_oracle_compute(const anna_oracle_pot *pot, int nall, const double *x,
                        int inum, const int *ilist, const int *numneigh,
                        const long long *first, const int *neigh, double cutsq,
                        double *f, double *eatom, double *eng, double *virial, double *vatom,
                        double *Gout, double *Lout, const double *frozen)
{
    const int nsf = pot->nsf, npsf = pot->npsf, ntsf = pot->ntsf;
    const double Rc = pot->cut, coeff_b = MY_PI / Rc;
    const double *g = pot->gparams;
    const double A0 = g[0], yy = g[1], gamma = g[2], C0 = g[3], c1F = g[4], c2F = g[5], V0 = g[6], b1 = g[7];
    const double b2 = g[8], delta = g[9], r0 = g[10], r1 = g[11], hc = g[12], d1 = g[13], q1 = g[14], d3 = g[15], q3 = g[16];
    if (pot->nout < 2 || npsf + ntsf != nsf) return -1;
    (void)nall;
    int status = 0;
#pragma omp parallel for schedule(dynamic, 8)
    for (int ii = 0; ii < inum; ii++) {
        const int i = ilist[ii];
        const int jnum = numneigh[i];
        const int *jlist = neigh + first[i];
        double G[ANNA_ORACLE_MAXSF], Tx[ANNA_ORACLE_MAXSF], lparams[ANNA_ORACLE_MAXNOD];
        double(*all)[4] = (double(*)[4])malloc(sizeof(double[4]) * (size_t)(jnum > 0 ? jnum : 1));
        if (!all) { status = -2; continue; }
        memset(G, 0, sizeof(G));
        /* descriptor, adp:117-158 */
        for (int jj = 0; jj < jnum; jj++) {
            const int j = jlist[jj] & NEIGHMASK;
            double xij[3] = {x[3 * i] - x[3 * j], x[3 * i + 1] - x[3 * j + 1], x[3 * i + 2] - x[3 * j + 2]};
            const double rsqij = xij[0] * xij[0] + xij[1] * xij[1] + xij[2] * xij[2];
            all[jj][0] = xij[0]; all[jj][1] = xij[1]; all[jj][2] = xij[2]; all[jj][3] = sqrt(rsqij);
            if (rsqij > cutsq || rsqij < 1.0e-12) continue;
            const double rijinv = 1.0 / sqrt(rsqij);
            const double uj[3] = {rijinv * xij[0], rijinv * xij[1], rijinv * xij[2]};
            const double rij = all[jj][3];
            const double fcij = 0.5 * (cos(coeff_b * rij) + 1.0);
            {   /* adp:584-597 */
                const double xx = 2 * rij / Rc - 1;
                for (int m = 0; m < npsf; m++) Tx[m] = m == 0 ? 1 : m == 1 ? xx : 2 * xx * Tx[m - 1] - Tx[m - 2];
                for (int m = 0; m < npsf; m++) G[m] += Tx[m] * fcij;
            }
            for (int kk = jj + 1; kk < jnum; kk++) {
                const int k = jlist[kk] & NEIGHMASK;    /* the reference does not mask k (adp:137); bits set there would read out of bounds */
                const double xik[3] = {x[3 * i] - x[3 * k], x[3 * i + 1] - x[3 * k + 1], x[3 * i + 2] - x[3 * k + 2]};
                const double rsqik = xik[0] * xik[0] + xik[1] * xik[1] + xik[2] * xik[2];
                if (rsqik > cutsq || rsqik < 1.0e-12) continue;
                const double rikinv = 1.0 / sqrt(rsqik);
                const double uk[3] = {rikinv * xik[0], rikinv * xik[1], rikinv * xik[2]};
                const double cos_theta = uj[0] * uk[0] + uj[1] * uk[1] + uj[2] * uk[2];
                const double rik = sqrt(rsqik);
                const double fcik = 0.5 * (cos(coeff_b * rik) + 1.0);
                const double xx = 0.5 * (cos_theta + 1);        /* adp:599-612 */
                for (int n = 0; n < ntsf; n++) Tx[n] = n == 0 ? 1 : n == 1 ? xx : 2 * xx * Tx[n - 1] - Tx[n - 2];
                for (int n = 0; n < ntsf; n++) G[n + npsf] += Tx[n] * fcij * fcik;
            }
        }
        anna_feed_forward(pot, G, lparams);                     /* adp:161-162 */
        if (Gout) memcpy(Gout + (size_t)ii * nsf, G, sizeof(double) * nsf);
        if (frozen) for (int o = 0; o < pot->nout; o++) lparams[o] = frozen[o];
        if (Lout) memcpy(Lout + (size_t)ii * pot->nout, lparams, sizeof(double) * pot->nout);
        const double d2 = lparams[0], q2 = lparams[1];

        /* per-atom sums, adp:165-196 */
        double mu[3] = {0, 0, 0}, lam[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        double rho_i = 0.0, adp_repul_eng = 0.0;
        const double coeff_repul = V0 / (b2 - b1);
        for (int jj = 0; jj < jnum; jj++) {
            const double r = all[jj][3];
            if (r > Rc || r < 1.0e-12) continue;
            const double sx = (r - Rc) / hc;
            const double stpf = pow(sx, 4) / (1 + pow(sx, 4));
            const double u = stpf * (d1 * exp(-d2 * r) + d3);
            const double w = stpf * (q1 * exp(-q2 * r) + q3);
            for (int a = 0; a < 3; a++) mu[a] += u * all[jj][a];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) lam[a][b] += w * all[jj][a] * all[jj][b];
            const double rho_z = r - r0;
            const double exp_z = exp(-gamma * rho_z);
            rho_i += stpf * (A0 * pow(rho_z, yy) * exp_z * (1 + exp_z) + C0);
            const double repul_z = r / r1;
            adp_repul_eng += stpf * (coeff_repul * (b2 / pow(repul_z, b1) - b1 / pow(repul_z, b2)) + delta);
        }
        const double v_i = lam[0][0] + lam[1][1] + lam[2][2];
        double sum_mu = 0.0, sum_lam = 0.0;
        for (int a = 0; a < 3; a++) {
            sum_mu += mu[a] * mu[a];
            for (int b = 0; b < 3; b++) sum_lam += pow(lam[a][b], 2);
        }
        const double f_v = -1.0 / 3.0 * v_i;
        const double rep_coeff = V0 / (b2 - b1);
        const double adp_angular_eng = 0.5 * sum_mu + 0.5 * sum_lam - 1.0 / 6.0 * v_i * v_i;
        const double adp_embed_eng = c1F * sqrt(rho_i) + c2F * pow(rho_i, 2);
        const double evdwl = 0.5 * adp_repul_eng + adp_embed_eng + adp_angular_eng + pot->e_base;   /* adp:212 */

        /* forces, adp:215-280 */
        double fi[3] = {0, 0, 0}, vi[6] = {0, 0, 0, 0, 0, 0};
        for (int jj = 0; jj < jnum; jj++) {
            const int j = jlist[jj] & NEIGHMASK;
            const double rij = all[jj][3];
            if (rij > Rc || rij < 1.0e-12) continue;
            const double xi = all[jj][0], yi = all[jj][1], zi = all[jj][2];
            const double sx = (rij - Rc) / hc;
            const double t1 = 1 + pow(sx, 4);
            const double stpf = pow(sx, 4) / t1;
            const double d_stpf = 4 * pow(sx, 3) / pow(t1, 2) / hc;
            const double rho_z = rij - r0;
            const double exp_z = exp(-gamma * rho_z);
            const double z_yy = A0 * pow(rho_z, yy);
            const double ga_zyy = z_yy * gamma;
            const double d_rho = exp_z * (1.0 + exp_z) * (z_yy * (d_stpf + stpf * yy / rho_z) - ga_zyy) + C0 * d_stpf - ga_zyy * exp_z * exp_z;
            const double d_embed = (0.5 * c1F * pow(rho_i, -0.5) + 2.0 * c2F * rho_i) * d_rho;
            const double repul_z = rij / r1;
            const double zb1 = pow(repul_z, b1), zb2 = pow(repul_z, b2);
            const double drep_t = b2 * b1 / r1;
            const double rep_t1 = rep_coeff * (b2 / zb1 - b1 / zb2) + delta;
            const double d_repul = d_stpf * rep_t1 + stpf * rep_coeff * (drep_t / repul_z * (-1.0 / zb1 + 1.0 / zb2));
            const double u_term = d1 * exp(-d2 * rij), w_term = q1 * exp(-q2 * rij);
            const double adp_u = stpf * (u_term + d3);
            const double adp_w = 2.0 * stpf * (w_term + q3);
            const double d_adp_u = d_stpf * (u_term + d3) + stpf * (-d2 * u_term);
            const double d_adp_w = d_stpf * (w_term + q3) + stpf * (-q2 * w_term);
            const double lamb1 = d_adp_w * (lam[0][0] * xi * xi + lam[1][1] * yi * yi + lam[2][2] * zi * zi);
            const double lamb2 = d_adp_w * (lam[0][1] * xi * yi + lam[0][2] * xi * zi + lam[1][2] * yi * zi) * 2.0 + lamb1;
            const double df1 = 0.5 * d_repul + d_embed + d_adp_u * (mu[0] * xi + mu[1] * yi + mu[2] * zi) + lamb2;
            const double df3 = f_v * (d_adp_w * rij + adp_w);
            const double fx = df1 * xi / rij + adp_w * (yi * lam[0][1] + zi * lam[0][2] + xi * lam[0][0]) + mu[0] * adp_u + xi * df3;
            const double fy = df1 * yi / rij + adp_w * (yi * lam[1][1] + zi * lam[1][2] + xi * lam[0][1]) + mu[1] * adp_u + yi * df3;
            const double fz = df1 * zi / rij + adp_w * (yi * lam[1][2] + zi * lam[2][2] + xi * lam[0][2]) + mu[2] * adp_u + zi * df3;
            fi[0] -= fx; fi[1] -= fy; fi[2] -= fz;
#pragma omp atomic
            f[3 * j] += fx;
#pragma omp atomic
            f[3 * j + 1] += fy;
#pragma omp atomic
            f[3 * j + 2] += fz;
            /* ev_tally_xyz(i, j, nlocal, newton, 0, 0, -fx, -fy, -fz, delx, dely, delz), adp:276-278 */
            const double v[6] = {xi * -fx, yi * -fy, zi * -fz, xi * -fy, xi * -fz, yi * -fz};
            for (int c = 0; c < 6; c++) vi[c] += v[c];
            if (vatom)
                for (int c = 0; c < 6; c++) {
#pragma omp atomic
                    vatom[6 * (size_t)j + c] += 0.5 * v[c];
                }
        }
        for (int c = 0; c < 3; c++) {
#pragma omp atomic
            f[3 * i + c] += fi[c];
        }
        if (vatom)
            for (int c = 0; c < 6; c++) {
#pragma omp atomic
                vatom[6 * (size_t)i + c] += 0.5 * vi[c];
            }
        if (virial)
            for (int c = 0; c < 6; c++) {
#pragma omp atomic
                virial[c] += vi[c];
            }
        if (eatom) {
#pragma omp atomic
            eatom[i] += evdwl;
        }
        if (eng) {
#pragma omp atomic
            eng[0] += evdwl;
        }
        free(all);
    }
    return status;
}
