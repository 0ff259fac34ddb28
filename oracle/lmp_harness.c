/* lmp_harness.c -- the data a LAMMPS run hands to Pair::compute(), for tests.
 *
 * TEST INFRASTRUCTURE ONLY.  Not part of the reference and not part of the
 * product: it plays the role LAMMPS core plays around the pair style (ghost
 * atoms for periodic boundaries as Comm::borders makes them, and a binned
 * *full* neighbour list as requested at fe_v2/src/pair_annp.cpp:317), so the
 * oracle and the HIP path can be fed identical inputs without a LAMMPS build.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Periodic images of the n owned atoms that lie within rc outside the box.
 * box = {xlo,ylo,zlo,xhi,yhi,zhi}; periodic[d] in {0,1}.
 * Writes up to cap ghosts (positions xg[3*g], owner[g]) and returns how many exist. */
long long harness_ghosts(int n, const double *x, const double *box, const int *periodic,
                         double rc, long long cap, double *xg, int *owner)
{
    long long ng = 0;
    double L[3] = {box[3] - box[0], box[4] - box[1], box[5] - box[2]};
    for (int i = 0; i < n; i++) {
        for (int sx = -1; sx <= 1; sx++)
            for (int sy = -1; sy <= 1; sy++)
                for (int sz = -1; sz <= 1; sz++) {
                    int s[3] = {sx, sy, sz};
                    if (!sx && !sy && !sz) continue;
                    int ok = 1;
                    double p[3];
                    for (int d = 0; d < 3 && ok; d++) {
                        if (s[d] && !periodic[d]) ok = 0;
                        p[d] = x[3 * i + d] + s[d] * L[d];
                        if (p[d] < box[d] - rc || p[d] >= box[3 + d] + rc) ok = 0;
                    }
                    if (!ok) continue;
                    if (ng < cap && xg) {
                        xg[3 * ng] = p[0]; xg[3 * ng + 1] = p[1]; xg[3 * ng + 2] = p[2];
                        owner[ng] = i;
                    }
                    ng++;
                }
    }
    return ng;
}

/* Full neighbour list of atoms [0,nlocal) over all nall atoms, r^2 <= rc^2, i != j.
 * Call once with neigh == NULL to obtain numneigh[] (and the total as return value),
 * build first[] as its exclusive prefix sum, then call again to fill neigh[]. */
long long harness_neigh(int nlocal, int nall, const double *x, double rc,
                        int *numneigh, const long long *first, int *neigh)
{
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i = 0; i < nall; i++)
        for (int d = 0; d < 3; d++) {
            if (x[3 * i + d] < lo[d]) lo[d] = x[3 * i + d];
            if (x[3 * i + d] > hi[d]) hi[d] = x[3 * i + d];
        }
    int nb[3];
    for (int d = 0; d < 3; d++) {
        nb[d] = (int)floor((hi[d] - lo[d]) / rc) + 1;
        if (nb[d] < 1) nb[d] = 1;
    }
    long long nbins = (long long)nb[0] * nb[1] * nb[2];
    int *binof = (int *)malloc(sizeof(int) * (size_t)nall);
    long long *bstart = (long long *)calloc((size_t)nbins + 1, sizeof(long long));
    int *bitems = (int *)malloc(sizeof(int) * (size_t)nall);
    for (int i = 0; i < nall; i++) {
        int c[3];
        for (int d = 0; d < 3; d++) {
            c[d] = (int)floor((x[3 * i + d] - lo[d]) / rc);
            if (c[d] >= nb[d]) c[d] = nb[d] - 1;
            if (c[d] < 0) c[d] = 0;
        }
        binof[i] = (c[2] * nb[1] + c[1]) * nb[0] + c[0];
        bstart[binof[i] + 1]++;
    }
    for (long long b = 0; b < nbins; b++) bstart[b + 1] += bstart[b];
    {
        long long *fill = (long long *)malloc(sizeof(long long) * (size_t)nbins);
        memcpy(fill, bstart, sizeof(long long) * (size_t)nbins);
        for (int i = 0; i < nall; i++) bitems[fill[binof[i]]++] = i;
        free(fill);
    }
    double rc2 = rc * rc;
    long long total = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : total)
#endif
    for (int i = 0; i < nlocal; i++) {
        int b = binof[i];
        int c0 = b % nb[0], c1 = (b / nb[0]) % nb[1], c2 = b / (nb[0] * nb[1]);
        int cnt = 0;
        int *out = neigh ? neigh + first[i] : NULL;
        for (int z = c2 - 1; z <= c2 + 1; z++) {
            if (z < 0 || z >= nb[2]) continue;
            for (int y = c1 - 1; y <= c1 + 1; y++) {
                if (y < 0 || y >= nb[1]) continue;
                for (int xx = c0 - 1; xx <= c0 + 1; xx++) {
                    if (xx < 0 || xx >= nb[0]) continue;
                    long long bb = ((long long)z * nb[1] + y) * nb[0] + xx;
                    for (long long s = bstart[bb]; s < bstart[bb + 1]; s++) {
                        int j = bitems[s];
                        if (j == i) continue;
                        double d0 = x[3 * i] - x[3 * j], d1 = x[3 * i + 1] - x[3 * j + 1], d2 = x[3 * i + 2] - x[3 * j + 2];
                        if (d0 * d0 + d1 * d1 + d2 * d2 <= rc2) {
                            if (out) out[cnt] = j;
                            cnt++;
                        }
                    }
                }
            }
        }
        if (!neigh) numneigh[i] = cnt;
        total += cnt;
    }
    free(binof); free(bstart); free(bitems);
    return total;
}
