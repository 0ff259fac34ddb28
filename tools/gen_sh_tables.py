#!/usr/bin/env python3
"""Generator of meng_zhang_amd/csrc/sh_tables.hpp: the constants of the spherical-harmonic form of the Chebyshev angular
descriptor (fe_sh_kernels.hpp).  Exact rational arithmetic (fractions), rounded once to double.

  sum_{a<b} fc_a fc_b T_n((cos_ab + 1)/2)  =  1/2 ( sum_l q[n][l] pw_l  -  sum_a fc_a^2 )              (fe:658-695's sums)
  pw_l = sum_{a,b} fc_a fc_b P_l(cos_ab)   =  sum_m kappa[l][m] ( Ac_lm^2 + As_lm^2 )                  (addition theorem)
  Ac_lm + i As_lm = sum_a fc_a Pm_{l-m}(z_a) (x_a + i y_a)^m,   Pm_k monic: Pm_k = z Pm_{k-1} - gamma(m,k) Pm_{k-2}
  (Pm_k is d^m P_l / dz^m, l = m + k, divided by its leading coefficient lambda_lm = (2l)! / (2^l l! (l-m)!))

    python tools/gen_sh_tables.py            # rewrites the header
    python tools/gen_sh_tables.py --check    # exit 1 if the header on disk differs
"""
import os
import sys
from decimal import Decimal, getcontext
from fractions import Fraction as Fr
from math import comb
from math import factorial as fac

L = 18
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "meng_zhang_amd", "csrc", "sh_tables.hpp")


def gamma(m, k):
    return Fr((k - 1) * (k - 1 + 2 * m), (2 * k + 2 * m - 1) * (2 * k + 2 * m - 3))


def kappa(l, m):
    lam = Fr(fac(2 * l), 2 ** l * fac(l) * fac(l - m))
    return (1 if m == 0 else 2) * Fr(fac(l - m), fac(l + m)) * lam * lam


def cheb(n):
    """integer coefficients of T_n(x), ascending"""
    t = [[1], [0, 1]]
    for k in range(2, n + 1):
        a = [0] + [2 * c for c in t[k - 1]]
        b = t[k - 2] + [0] * (len(a) - len(t[k - 2]))
        t.append([x - y for x, y in zip(a, b)])
    return t[n]


def shifted(n):
    """T_n((z+1)/2) in powers of z"""
    out = [Fr(0)] * (n + 1)
    for k, ck in enumerate(cheb(n)):
        for j in range(k + 1):
            out[j] += Fr(ck * comb(k, j), 2 ** k)
    return out


def dfac(n):
    r = 1
    while n > 1:
        r *= n
        n -= 2
    return r


def mono_to_legendre(k):
    """z^k = sum_l c_l P_l(z)"""
    out = [Fr(0)] * (k + 1)
    for l in range(k, -1, -2):
        s = (k - l) // 2
        out[l] = Fr((2 * l + 1) * fac(k), 2 ** s * fac(s) * dfac(k + l + 1))
    return out


def qmat():
    q = [[Fr(0)] * (L + 1) for _ in range(L + 1)]
    for n in range(L + 1):
        for k, c in enumerate(shifted(n)):
            for l, v in enumerate(mono_to_legendre(k)):
                q[n][l] += c * v
    return q


def pm_monomials(m):
    """Pm^(m)_k(z), k = 0..L-m, as exact coefficient lists in ascending powers of z (same parity as k only)"""
    polys = [[Fr(1)], [Fr(0), Fr(1)]]
    for k in range(2, L + 1 - m):
        a = [Fr(0)] + polys[k - 1]
        b = polys[k - 2] + [Fr(0)] * (len(a) - len(polys[k - 2]))
        polys.append([x - gamma(m, k) * y for x, y in zip(a, b)])
    return polys[:L + 1 - m]


def shf_toff(m):
    """first entry of column m in the force pass's table: columns m = L..0, in a column the powers z^j, j = K-1..0"""
    return (L - m) * (L + 1 - m) // 2


def conv_records():
    """The change of basis of the force pass, b_mj = sum_{k>=j, k=j mod 2} M^(m)_kj B_mk (Pm^(m)_k = sum_j M_kj z^j), cut into the
    records its kernel walks: for column m, for the blocks of 16 powers j = blk..blk+15, for t = 0,1,..: 16 doubles
    M^(m)_{j+2t, j} (0 where j+2t > K-1).  Returns (records, first record of (m, blk))."""
    recs, first = [], {}
    for m in range(L + 1):
        K = L + 1 - m
        polys = pm_monomials(m)
        for blk in (0, 16):
            if blk >= K:
                continue
            first[(m, blk)] = len(recs)
            for t in range((K - blk + 1) // 2):
                row = []
                for jl in range(16):
                    j, k = blk + jl, blk + jl + 2 * t
                    row.append(polys[k][j] if k < K else Fr(0))
                recs.append(row)
    return recs, first


def tail_schedule():
    """The descriptor pass sums MONOMIAL moments sum_b fc_b z_b^j w_b^m (round 4b) and turns them into the moments of the
    Pm^(m)_k afterwards, A_(m+k,m) = sum_{j = k, k-2, ..} M^(m)_kj Mom_jm, sixteen entries (m, k) per round, one per lane of an
    atom, in place.  Entries in descending k: an entry only reads powers j <= k of its own column, so nothing a round reads
    was overwritten by an earlier one, and the lanes of a round have nearly the same number of terms.
    Returns (rounds, trips): rounds[r] = list of 16 (m, k) or None; trips[r] = terms a lane of round r walks."""
    ent = sorted(((m, k) for m in range(L + 1) for k in range(L + 1 - m)), key=lambda e: (-e[1], e[0]))
    rounds = [ent[i:i + 16] for i in range(0, len(ent), 16)]
    trips = [r[0][1] // 2 + 1 for r in rounds]
    rounds = [r + [None] * (16 - len(r)) for r in rounds]
    return rounds, trips


def tail_tables():
    """info[r][lane] = position of entry (m, k) in the moment row | (m + k) << 8 | 1 << 15 (a lane without an entry: position of
    (0, 0), no flag); kappa[r][lane]; coef[first trip of r + t][lane] = M^(m)_(k, k-2t), 0 past the last term."""
    rounds, trips = tail_schedule()
    polys = {m: pm_monomials(m) for m in range(L + 1)}
    info, kap, coef = [], [], []
    for r, T in zip(rounds, trips):
        info.append([(shf_toff(e[0]) + L - e[0] - e[1]) | ((e[0] + e[1]) << 8) | (1 << 15) if e else shf_toff(0) + L for e in r])
        kap.append([kappa(e[0] + e[1], e[0]) if e else Fr(0) for e in r])
        for t in range(T):
            coef.append([polys[e[0]][e[1]][e[1] - 2 * t] if e and e[1] - 2 * t >= 0 else Fr(0) for e in r])
    return info, kap, coef, trips


# ---- the same change of basis in three groups of columns, each through LDS as soon as its columns are summed (round 6) ----------
# annp_fe_desc_sh<.., GROUPED>: the monomial totals of a group of columns wait in LDS (there is room for a third of them beside the
# neighbours' state up to 112 neighbours per atom) instead of the atom's moment row in HBM; after the group's last column its entries
# change basis -- rounds of sixteen as above, per group -- and only kappa A goes to memory, once.  Columns are summed m = 0, 1, ..:
# the groups are runs of m, cut so that each fills its rounds (54 + 58 + 78 entries in 4 + 4 + 5 rounds: 13, one more than ungrouped).
SHG_GROUPS = [tuple(range(0, 3)), tuple(range(3, 7)), tuple(range(7, L + 1))]


def group_schedule():
    """[(columns, rounds, trips)] per group: tail_schedule restricted to the group's columns"""
    out = []
    for cols in SHG_GROUPS:
        ent = sorted(((m, k) for m in cols for k in range(L + 1 - m)), key=lambda e: (-e[1], e[0]))
        rounds = [ent[i:i + 16] for i in range(0, len(ent), 16)]
        trips = [r[0][1] // 2 + 1 for r in rounds]
        out.append((cols, [r + [None] * (16 - len(r)) for r in rounds], trips))
    return out


def group_tables():
    """As tail_tables, rounds of all groups one after the other; positions are LOCAL to the group's LDS buffer (entry (m, k) of a group
    whose highest column is mh sits at shf_toff(m) + L - m - k - shf_toff(mh)); a lane without an entry reads the group's last position
    with zero coefficients.  Returns (info, kappa, coef, trips, first round of every group, base position of every group, entries per group)."""
    polys = {m: pm_monomials(m) for m in range(L + 1)}
    info, kap, coef, trips_all, rfirst, gbase, nent = [], [], [], [], [0], [], []
    for cols, rounds, trips in group_schedule():
        base = shf_toff(max(cols))
        n = sum(L + 1 - m for m in cols)
        gbase.append(base); nent.append(n)
        for r, T in zip(rounds, trips):
            info.append([(shf_toff(e[0]) + L - e[0] - e[1] - base) | ((e[0] + e[1]) << 8) | (1 << 15) if e else n - 1 for e in r])
            kap.append([kappa(e[0] + e[1], e[0]) if e else Fr(0) for e in r])
            for t in range(T):
                coef.append([polys[e[0]][e[1]][e[1] - 2 * t] if e and e[1] - 2 * t >= 0 else Fr(0) for e in r])
            trips_all.append(T)
        rfirst.append(len(trips_all))
    return info, kap, coef, trips_all, rfirst, gbase, nent


def sqrt_to_double(x):
    getcontext().prec = 80
    return float((Decimal(x.numerator) / Decimal(x.denominator)).sqrt())


def render():
    q = qmat()
    lines = ["// sh_tables.hpp -- GENERATED by tools/gen_sh_tables.py (exact rationals rounded once); do not edit.",
             "#pragma once", "", "namespace annp {", "",
             "constexpr int SH_LMAX = %d;" % L, "",
             "// entries of the moment row / of the force pass's table: one per (m, k), 190",
             "constexpr int SHF_NE = %d;" % sum(L + 1 - m for m in range(L + 1)), ""]
    lines += [
              "// force pass on monomials (annp_fe_force_sh, round 4): its table holds, for column m = 18..0, the coefficients of z^j,",
              "// j = K-1..0 (K = 19-m), of beta_m(z) -- the order Horner's rule walks them in; entry (m, j) sits at shf_toff(m) + K-1-j.",
              "// The moment buffer of annp_fe_desc_sh has the same order: (cosine, sine) of (l = m+k, m) at 2 (shf_toff(m) + K-1-k).",
              "__host__ __device__ constexpr int shf_toff(int m) { return (SH_LMAX - m) * (SH_LMAX + 1 - m) / 2; }",
              "// l = m + k of every entry, in that order",
              "#define ANNP_SHF_L_INIT { " + ", ".join(str(m + k) for m in range(L, -1, -1) for k in range(L - m, -1, -1)) + " }",
              "// Pm^(m)_k(z) = sum_j M_kj z^j in the records the kernel's change of basis walks (conv_records of the generator):",
              "// SHF_CONV_FIRST[m][blk/16] = first record of column m, powers blk..blk+15; a record = 16 doubles M_(j+2t, j)",
              "constexpr int SHF_CONV_NREC = %d;" % len(conv_records()[0]),
              "constexpr int SHF_CONV_FIRST[SH_LMAX + 1][2] = {" + ", ".join("{%d, %d}" % (conv_records()[1].get((m, 0), -1), conv_records()[1].get((m, 16), -1)) for m in range(L + 1)) + "};",
              "#define ANNP_SHF_CONV_INIT { \\"] + ["    " + ", ".join(float(v).hex() for v in row) + ", \\" for row in conv_records()[0]] + ["}", "",
              "// descriptor pass, round 4b: monomial moments -> moments of the Pm^(m)_k (tail_schedule / tail_tables of the generator):",
              "// SHD_NROUND rounds of 16 entries, SHD_TRIPS[r] terms per lane, SHD_TFIRST[r] = first row of round r in the coefficients",
              "constexpr int SHD_NROUND = %d;" % len(tail_tables()[3]),
              "constexpr int SHD_TRIPS[SHD_NROUND] = {" + ", ".join(str(t) for t in tail_tables()[3]) + "};",
              "constexpr int SHD_TFIRST[SHD_NROUND + 1] = {" + ", ".join(str(sum(tail_tables()[3][:r])) for r in range(len(tail_tables()[3]) + 1)) + "};",
              "#define ANNP_SHD_INFO_INIT { " + ", ".join(str(v) for row in tail_tables()[0] for v in row) + " }",
              "#define ANNP_SHD_KAPPA_INIT { \\"] + ["    " + ", ".join(float(v).hex() for v in row) + ", \\" for row in tail_tables()[1]] + ["}",
              "#define ANNP_SHD_COEF_INIT { \\"] + ["    " + ", ".join(float(v).hex() for v in row) + ", \\" for row in tail_tables()[2]] + ["}", "",
              "// the same in three groups of columns through LDS (round 6; group_schedule / group_tables of the generator): SHG_COLS[g] = columns of",
              "// group g as [first, last], SHG_RFIRST[g] = its first round, SHG_GBASE[g] = position of its first entry in the moment row, SHG_NENT[g]",
              "// = its entries per atom; positions in ANNP_SHG_INFO_INIT are local to the group (row position - SHG_GBASE[g])",
              "constexpr int SHG_NGROUP = %d;" % len(SHG_GROUPS),
              "constexpr int SHG_COLS[SHG_NGROUP][2] = {" + ", ".join("{%d, %d}" % (min(c), max(c)) for c in SHG_GROUPS) + "};",
              "constexpr int SHG_NROUND = %d;" % len(group_tables()[3]),
              "constexpr int SHG_RFIRST[SHG_NGROUP + 1] = {" + ", ".join(str(v) for v in group_tables()[4]) + "};",
              "constexpr int SHG_GBASE[SHG_NGROUP] = {" + ", ".join(str(v) for v in group_tables()[5]) + "};",
              "constexpr int SHG_NENT[SHG_NGROUP] = {" + ", ".join(str(v) for v in group_tables()[6]) + "};",
              "constexpr int SHG_TRIPS[SHG_NROUND] = {" + ", ".join(str(t) for t in group_tables()[3]) + "};",
              "constexpr int SHG_TFIRST[SHG_NROUND + 1] = {" + ", ".join(str(sum(group_tables()[3][:r])) for r in range(len(group_tables()[3]) + 1)) + "};",
              "#define ANNP_SHG_INFO_INIT { " + ", ".join(str(v) for row in group_tables()[0] for v in row) + " }",
              "#define ANNP_SHG_KAPPA_INIT { \\"] + ["    " + ", ".join(float(v).hex() for v in row) + ", \\" for row in group_tables()[1]] + ["}",
              "#define ANNP_SHG_COEF_INIT { \\"] + ["    " + ", ".join(float(v).hex() for v in row) + ", \\" for row in group_tables()[2]] + ["}", "",
              "// q[n][l]: T_n((z+1)/2) = sum_{l<=n} q[n][l] P_l(z)",
              "#define ANNP_SH_Q_INIT { \\"]
    for n in range(L + 1):
        lines.append("    " + ", ".join(float(q[n][l]).hex() for l in range(L + 1)) + ", \\")
    lines += ["}", "", "}  // namespace annp", ""]
    return "\n".join(lines)


if __name__ == "__main__":
    text = render()
    if "--check" in sys.argv:
        sys.exit(0 if open(OUT).read() == text else 1)
    with open(OUT, "w") as fh:
        fh.write(text)
