/*
 * ff_oracle_score.c -- CPU oracle (TEST INFRASTRUCTURE; see ff_oracle.h).  Part 2: per-hit and per-guide
 * off-target scores, restated string-for-string from the reference's scoring package, and a
 * java.lang.Double.toString-compatible formatter.  Paths relative to /root/reference/src/main/scala.
 */
#define _GNU_SOURCE
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ff_oracle_internal.h"
#include "cfd_tables.inc" /* values of scoring/Doench2016CFDScore.scala:173-214, see tools/gen_score_tables.py */

/* ---- Doench2016CFDScore ---- */
static double cfd_mm_lookup(const char *key) { /* mmLookup(key), :143-146 */
    for (int i = 0; i < 240; i++)
        if (strcmp(FFO_CFD_MM[i].key, key) == 0) return FFO_CFD_MM[i].w;
    ffo_set_error("Missing key %s in mm Lookup table", key);
    return NAN;
}

double ffo_cfd_pam(const char *pam2) { /* pamLookup, :211-214 */
    for (int i = 0; i < 16; i++)
        if (FFO_CFD_PAM[i].pam[0] == pam2[0] && FFO_CFD_PAM[i].pam[1] == pam2[1]) return FFO_CFD_PAM[i].w;
    return NAN;
}

static char special_reverse_comp_base(char c) { /* :159 */
    return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'U' ? 'A' : c;
}

double ffo_cfd_score_pair(const char *guide20, const char *ot20) { /* scoreCFD :132-151 */
    double score = 1.0;
    for (int i = 0; i < 20; i++) {
        char g = guide20[i] == 'T' ? 'U' : guide20[i]; /* replace('T','U') :136-137 */
        char o = ot20[i] == 'T' ? 'U' : ot20[i];
        if (g != o) {
            char key[16];
            snprintf(key, sizeof key, "r%c:d%c,%d", g, special_reverse_comp_base(o), i + 1); /* :142 */
            score *= cfd_mm_lookup(key);                                                    /* :146 */
        }
    }
    return score;
}

/* ---- CrisprMitEduOffTarget ---- */
static const double HSU_COEFF[20] = { /* offtargetCoeff :43-47 */
    0.0, 0.0, 0.014, 0.0, 0.0, 0.395, 0.317, 0.0, 0.389, 0.079,
    0.445, 0.508, 0.613, 0.851, 0.732, 0.828, 0.615, 0.804, 0.685, 0.583};

double ffo_hsu_score_offtarget(const ffo_pack *p, const char *guide_bases, uint64_t ot) { /* scoreOffTarget :107-148 */
    char s[32];
    ffo_bit_decode(ot, p->scan_len, s);
    int mismatches = 0, dist_sum = 0, dist_n = 0, last = -1;
    double part_one = 1.0;
    for (int i = 0; i < 20; i++) { /* guideSize = 20 :197 */
        if (s[i] != guide_bases[i]) {
            part_one = part_one * (1.0 - HSU_COEFF[i]); /* :120 */
            mismatches += 1;
            if (last >= 0) { dist_sum += i - last; dist_n++; } /* :123-125 */
            last = i;
        }
    }
    double part_two;
    if (mismatches < 2) part_two = 1.0; /* :132 */
    else {
        double avg = (double)dist_sum / (double)dist_n;      /* :133 */
        part_two = 1.0 / ((((19 - avg) / 19.0) * 4.0) + 1.0); /* :134 */
    }
    double part_three = mismatches == 0 ? 1.0 : 1.0 / pow((double)mismatches, 2); /* :138 */
    double total = part_one * part_two * part_three * 100.0;                       /* :140 */
    /* pamToAdjustment :53 on str.slice(21,23) ; default 0.01 :200 */
    const char *pam = s + 21;
    double adj = 0.01;
    if (pam[1] == 'G') {
        if (pam[0] == 'G') adj = 1.0;
        else if (pam[0] == 'A') adj = 0.26;
        else if (pam[0] == 'C') adj = 0.11;
        else if (pam[0] == 'T') adj = 0.01;
    }
    return total * adj; /* :147 */
}

/* ---- Jost & Santos CRISPRi: scoring/JostAndSantosCRISPRi.scala ---- */
#include "jost_table.inc"

static char jost_comp(char b) { return b == 'A' ? 'T' : b == 'C' ? 'G' : b == 'G' ? 'C' : b == 'T' ? 'A' : b; } /* utils/Utils.scala compBase */

/* scoreMapping(ScoreLookup(position, baseA, baseB)): the entry (position, "r<baseA, T as U>:d<baseB>") (:385-389) */
static double jost_lookup(int position, char base_a, char base_b) {
    char key[8];
    snprintf(key, sizeof key, "r%c:d%c", base_a == 'T' ? 'U' : base_a, base_b);
    for (int i = 0; i < 228; i++)
        if (FFO_JOST[i].pos == position && !strcmp(FFO_JOST[i].key, key)) return FFO_JOST[i].mean;
    return NAN; /* NoSuchElementException in the reference; cannot happen for two different ACGT bases */
}

double ffo_jost_calc_score(const ffo_pack *p, const char *target, const char *off) { /* calc_score :92-127 */
    double total = 1.0;
    if ((int)strlen(target) != p->scan_len || (int)strlen(off) != p->scan_len) return NAN; /* asserts :94-95 */
    if (p->scan_len == 23) {            /* cas9ScanLength20mer: the first base is skipped, positions 1..19 = string offsets 1..19 (:100-109) */
        for (int index = 0; index < 19; index++) {
            char base = off[index + 1];
            if (target[index + 1] != base) total *= jost_lookup(index + 1, base, jost_comp(target[index + 1]));
        }
    } else if (p->scan_len == 22) {     /* cas9ScanLength19mer: positions 1..19 = string offsets 0..18 (:112-123) */
        for (int index = 0; index < 19; index++) {
            char base = off[index];
            if (target[index] != base) total *= jost_lookup(index + 1, base, jost_comp(target[index]));
        }
    } else return NAN;                  /* IllegalStateException :125 */
    return total;
}

int ffo_score_guide(const ffo_pack *p, uint64_t guide, const uint64_t *hits, int n, ffo_guide_scores *out,
                    double *per_hit_cfd) {
    char bases[32], ot[32];
    ffo_bit_decode(guide, p->scan_len, bases);
    memset(out, 0, sizeof *out);

    /* Doench2016CFDScore.scoreGuide :53-88 ; validOverEnzyme :96-98 */
    out->cfd_valid = p->cas9_23;
    if (per_hit_cfd) for (int i = 0; i < n; i++) per_hit_cfd[i] = NAN;
    if (out->cfd_valid) {
        double sum = 0.0, mx = 0.0; /* .sum folds from 0.0 ; .max over the list */
        int any = 0;
        for (int i = 0; i < n; i++) {
            ffo_bit_decode(hits[i], p->scan_len, ot);
            if (strncmp(ot, bases, 20) != 0) {                      /* :67 */
                double pam = ffo_cfd_pam(ot + (p->scan_len - 2));   /* :69 */
                double cand = ffo_cfd_score_pair(bases, ot);        /* :71 */
                double sc = pam * cand;                             /* :72-73 */
                if (per_hit_cfd) per_hit_cfd[i] = sc;
                sum += sc * (double)ffo_get_count(hits[i]);         /* :79 */
                if (!any || sc > mx) mx = sc;                       /* :80 */
                any = 1;
            }
        }
        out->cfd_spec = any ? 1.0 / (1.0 + sum) : 1.0;
        out->cfd_max = any ? mx : 0.0;
    }

    /* CrisprMitEduOffTarget.score_crispr :60-105 ; considerOnTarget is never set (quirk 1) */
    out->hsu_valid = p->cas9_23;
    if (out->hsu_valid) {
        double sum = 0.0;
        for (int i = 0; i < n; i++)
            if (ffo_mismatches(p, guide, hits[i], FFO_STRING_MASK) != 0) sum += ffo_hsu_score_offtarget(p, bases, hits[i]); /* :90-91 */
        out->hsu = (100.0 / (100.0 + sum)) * 100.0; /* :104 */
    }

    /* ClosestHit.scoreGuide :43-76 */
    int closest = INT_MAX, count = 0;
    for (int i = 0; i < n; i++) {
        int mm = ffo_mismatches(p, hits[i], guide, FFO_STRING_MASK);
        int c = ffo_get_count(hits[i]);
        if (mm <= 4) out->hist[mm] += c;        /* :57-59 */
        if (mm < closest && mm > 0) { closest = mm; count = c; } /* :62-64 */
        else if (mm == closest) count += c;     /* :65-67 */
    }
    out->closest = closest;
    out->closest_count = closest == INT_MAX ? 0 : count;

    /* DangerousSequences :61-65 */
    for (int i = 0; i < n; i++)
        if (ffo_mismatches(p, hits[i], guide, FFO_STRING_MASK) == 0) out->in_genome += ffo_get_count(hits[i]);

    /* JostAndSantosCRISPRi.scoreGuide :27-46 ; validOverEnzyme :53-58 */
    out->jost_valid = p->index != 1 && (p->scan_len == 23 || p->scan_len == 22);
    if (out->jost_valid) {
        double sum = 0.0, mx = 0.0;
        int any = 0;
        for (int i = 0; i < n; i++) {
            if (ffo_mismatches(p, hits[i], guide, FFO_STRING_MASK) > 0) {   /* filter :40 */
                ffo_bit_decode(hits[i], p->scan_len, ot);
                double sc = ffo_jost_calc_score(p, bases, ot);               /* :36 */
                sum += sc * (double)ffo_get_count(hits[i]);                  /* :42 */
                if (!any || sc > mx) mx = sc;                                /* :43 */
                any = 1;
            }
        }
        out->jost_spec = 1.0 / (1.0 + sum);
        out->jost_max = any ? mx : 0.0;
    }
    return 0;
}

int ffo_result_score_guide(const ffo_result *r, const ffo_db *db, int g, ffo_guide_scores *out, double *per_hit_cfd) {
    const ffo_guide_ot *go = &r->guides[g];
    uint64_t *t = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(go->n_hits > 0 ? go->n_hits : 1));
    for (int i = 0; i < go->n_hits; i++) t[i] = go->hits[i].target;
    int rc = ffo_score_guide(db->pack, go->encoding, t, go->n_hits, out, per_hit_cfd);
    free(t);
    return rc;
}

/* ---- java.lang.Double.toString ----
 * Used wherever the reference prints a Double with .toString (Doench2016CFDScore.scala:72,84,
 * CrisprMitEduOffTarget.scala:60, DangerousSequences.scala:54).  Shortest decimal that round-trips, then
 * Java's layout rules.  JDK <= 18 occasionally prints one digit more than the shortest (JDK-4511638); that
 * anomaly is not reproduced. */
int ffo_java_double_to_string(double d, char *out) {
    if (isnan(d)) return sprintf(out, "NaN");
    if (isinf(d)) return sprintf(out, d < 0 ? "-Infinity" : "Infinity");
    if (d == 0.0) return sprintf(out, signbit(d) ? "-0.0" : "0.0");
    char buf[64], digits[32];
    int prec, exp10 = 0;
    for (prec = 1; prec <= 17; prec++) {
        snprintf(buf, sizeof buf, "%.*e", prec - 1, d);
        if (strtod(buf, NULL) == d) break;
    }
    /* buf = [-]d.ddddde[+-]xx */
    const char *q = buf;
    int neg = 0, nd = 0;
    if (*q == '-') { neg = 1; q++; }
    for (; *q && *q != 'e'; q++)
        if (*q >= '0' && *q <= '9') digits[nd++] = *q;
    exp10 = atoi(q + 1);
    while (nd > 1 && digits[nd - 1] == '0') nd--;
    digits[nd] = 0;
    char *w = out;
    if (neg) *w++ = '-';
    double a = fabs(d);
    if (a >= 1e-3 && a < 1e7) {
        if (exp10 >= 0) {
            for (int i = 0; i <= exp10; i++) *w++ = i < nd ? digits[i] : '0';
            *w++ = '.';
            if (nd > exp10 + 1) for (int i = exp10 + 1; i < nd; i++) *w++ = digits[i];
            else *w++ = '0';
        } else {
            *w++ = '0'; *w++ = '.';
            for (int i = 0; i < -exp10 - 1; i++) *w++ = '0';
            for (int i = 0; i < nd; i++) *w++ = digits[i];
        }
        *w = 0;
    } else {
        *w++ = digits[0]; *w++ = '.';
        if (nd > 1) for (int i = 1; i < nd; i++) *w++ = digits[i];
        else *w++ = '0';
        w += sprintf(w, "E%d", exp10);
    }
    return (int)(w - out);
}


/* ---- config C5: mismatches + one bulge (this repository's specification, see ff_oracle.h) ------------------------------
 * Plain strings, one loop per alignment: g = the 20 guide bases after the PAM, t = the 20 target bases after the PAM,
 * position 0 next to the PAM.
 *   RNA bulge k: guide base k unpaired      -> g[i] ~ t[i] (i < k),  g[i] ~ t[i-1] (i > k)
 *   DNA bulge k: target base k unpaired     -> g[i] ~ t[i] (i < k),  g[i] ~ t[i+1] (k <= i <= 18); g[19] has no stored partner */
int ffo_bulge_align(const ffo_pack *p, uint64_t guide, uint64_t target, int max_bulge, int *type, int *pos) {
    char gs[32], ts[32];
    ffo_bit_decode(guide, p->scan_len, gs);
    ffo_bit_decode(target, p->scan_len, ts);
    const int n = p->guide_hi - p->guide_lo; /* 20 */
    const char *g = gs + p->guide_lo, *t = ts + p->guide_lo;
    int best = 0, bt = 0, bp = 0;
    for (int i = 0; i < n; i++) best += g[i] != t[i];
    if (max_bulge > 0) {
        for (int kind = 1; kind <= 2; kind++)       /* RNA first: it wins ties against DNA */
            for (int k = 1; k <= n - 2; k++) {
                int mm = 0;
                for (int i = 0; i < k; i++) mm += g[i] != t[i];
                if (kind == 1) { for (int i = k + 1; i < n; i++) mm += g[i] != t[i - 1]; }
                else           { for (int i = k; i <= n - 2; i++) mm += g[i] != t[i + 1]; }
                if (mm < best) { best = mm; bt = kind; bp = k; }
            }
    }
    if (type) *type = bt;
    if (pos) *pos = bp;
    return best;
}
