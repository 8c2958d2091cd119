/*
 * ff_oracle_io.c -- CPU oracle (TEST INFRASTRUCTURE; see ff_oracle.h).  Part 3: on-disk database
 * (.header text + BGZF body), guide discovery in FASTA, the tab-delimited table writer/reader and the
 * three file-level drivers (index / discover / score).  Paths relative to /root/reference/src/main/scala.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "ff_oracle_internal.h"

/* ------------------------------------------------------------------------------------------------
 * BGZF (htsjdk BlockCompressed{Output,Input}Stream; SAM spec 4.1).  PARITY UNPINNED: codec not in the
 * reference tree, no reference test covers it.  Call sites: DatabaseWriter.scala:69,80,91,99-100.
 * ---------------------------------------------------------------------------------------------- */
#define BGZF_BLOCK 0xff00

typedef struct bgzf_writer {
    FILE *f;
    uint8_t buf[BGZF_BLOCK];
    int buffered;
    uint64_t block_address;
} bgzf_writer;

static int bgzf_flush_block(bgzf_writer *w) {
    if (w->buffered == 0) return 0;
    uint8_t out[BGZF_BLOCK + 1024];
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, 5, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return -1;
    zs.next_in = w->buf; zs.avail_in = (uInt)w->buffered;
    zs.next_out = out + 18; zs.avail_out = sizeof out - 18 - 8;
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { deflateEnd(&zs); return -2; }
    size_t clen = zs.total_out;
    deflateEnd(&zs);
    size_t total = 18 + clen + 8;
    static const uint8_t hdr[12] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0};
    memcpy(out, hdr, 12);
    out[12] = 'B'; out[13] = 'C'; out[14] = 2; out[15] = 0;
    out[16] = (uint8_t)((total - 1) & 0xff); out[17] = (uint8_t)((total - 1) >> 8);
    uint32_t crc = (uint32_t)crc32(crc32(0L, NULL, 0), w->buf, (uInt)w->buffered);
    uint32_t isz = (uint32_t)w->buffered;
    for (int i = 0; i < 4; i++) { out[18 + clen + i] = (uint8_t)(crc >> (8 * i)); out[22 + clen + i] = (uint8_t)(isz >> (8 * i)); }
    if (fwrite(out, 1, total, w->f) != total) return -3;
    w->block_address += total;
    w->buffered = 0;
    return 0;
}
static uint64_t bgzf_position(const bgzf_writer *w) { return (w->block_address << 16) | (uint64_t)w->buffered; } /* getPosition */
static int bgzf_write(bgzf_writer *w, const uint8_t *data, size_t n) {
    while (n > 0) {
        size_t space = (size_t)(BGZF_BLOCK - w->buffered), c = n < space ? n : space;
        memcpy(w->buf + w->buffered, data, c);
        w->buffered += (int)c; data += c; n -= c;
        if (w->buffered == BGZF_BLOCK && bgzf_flush_block(w)) return -1;
    }
    return 0;
}
static int bgzf_close(bgzf_writer *w) {
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (bgzf_flush_block(w)) return -1;
    if (fwrite(eof, 1, 28, w->f) != 28) return -2;
    return fclose(w->f);
}

/* whole-file BGZF reader: inflates every member, keeps (compressed start -> uncompressed start) */
typedef struct bgzf_image {
    uint8_t *data; size_t n;
    uint64_t *cstart; uint64_t *ustart; size_t nblocks;
} bgzf_image;

static void bgzf_image_free(bgzf_image *im) { free(im->data); free(im->cstart); free(im->ustart); }

static int bgzf_read_all(const char *path, bgzf_image *im) {
    memset(im, 0, sizeof *im);
    FILE *f = fopen(path, "rb");
    if (!f) { ffo_set_error("cannot open %s", path); return -1; }
    fseek(f, 0, SEEK_END);
    long fsz = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *raw = (uint8_t *)malloc((size_t)fsz + 1);
    if (fread(raw, 1, (size_t)fsz, f) != (size_t)fsz) { fclose(f); free(raw); ffo_set_error("short read %s", path); return -2; }
    fclose(f);
    size_t cap = 1 << 20, capb = 1024;
    im->data = (uint8_t *)malloc(cap);
    im->cstart = (uint64_t *)malloc(capb * 8);
    im->ustart = (uint64_t *)malloc(capb * 8);
    size_t off = 0;
    while (off + 18 <= (size_t)fsz) {
        const uint8_t *h = raw + off;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) { ffo_set_error("bad BGZF member at %zu", off); goto fail; }
        int xlen = h[10] | (h[11] << 8), bsize = -1;
        for (int x = 0; x + 4 <= xlen;) {
            const uint8_t *s = h + 12 + x;
            int slen = s[2] | (s[3] << 8);
            if (s[0] == 'B' && s[1] == 'C' && slen == 2) bsize = (s[4] | (s[5] << 8)) + 1;
            x += 4 + slen;
        }
        if (bsize < 0 || off + (size_t)bsize > (size_t)fsz) { ffo_set_error("BGZF member without BC field at %zu", off); goto fail; }
        const uint8_t *cdata = h + 12 + xlen;
        size_t clen = (size_t)bsize - 12 - (size_t)xlen - 8;
        uint32_t isz = 0, crc = 0;
        for (int i = 0; i < 4; i++) { crc |= (uint32_t)h[bsize - 8 + i] << (8 * i); isz |= (uint32_t)h[bsize - 4 + i] << (8 * i); }
        if (im->n + isz > cap) { while (im->n + isz > cap) cap *= 2; im->data = (uint8_t *)realloc(im->data, cap); }
        if (im->nblocks == capb) { capb *= 2; im->cstart = (uint64_t *)realloc(im->cstart, capb * 8); im->ustart = (uint64_t *)realloc(im->ustart, capb * 8); }
        im->cstart[im->nblocks] = off; im->ustart[im->nblocks] = im->n; im->nblocks++;
        if (isz > 0) {
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            inflateInit2(&zs, -15);
            zs.next_in = (Bytef *)cdata; zs.avail_in = (uInt)clen;
            zs.next_out = im->data + im->n; zs.avail_out = isz;
            int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || zs.total_out != isz) { ffo_set_error("BGZF inflate failed at %zu", off); goto fail; }
            if ((uint32_t)crc32(crc32(0L, NULL, 0), im->data + im->n, isz) != crc) { ffo_set_error("BGZF crc mismatch at %zu", off); goto fail; }
            im->n += isz;
        }
        off += (size_t)bsize;
    }
    free(raw);
    return 0;
fail:
    free(raw);
    bgzf_image_free(im);
    return -3;
}

static int64_t bgzf_linear_offset(const bgzf_image *im, uint64_t vpos) { /* seek(virtual pointer) */
    uint64_t c = vpos >> 16, within = vpos & 0xffff;
    size_t lo = 0, hi = im->nblocks;
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (im->cstart[mid] < c) lo = mid + 1; else hi = mid; }
    if (lo == im->nblocks || im->cstart[lo] != c) {
        /* a pointer to the end of the file (after the last data member) is legal for an empty tail */
        return -1;
    }
    return (int64_t)(im->ustart[lo] + within);
}

/* ------------------------------------------------------------------------------------------------
 * database files -- reference/binary/BinaryHeader.scala:69-160, DatabaseWriter.scala:58-111
 * ---------------------------------------------------------------------------------------------- */
#define FFO_MAGIC 0x1234ABCDE123890LL /* BinaryConstants.scala:26 */

int ffo_db_write(const ffo_db *db, const char *path) {
    bgzf_writer *w = (bgzf_writer *)calloc(1, sizeof *w);
    w->f = fopen(path, "wb");
    if (!w->f) { free(w); ffo_set_error("cannot create %s", path); return -1; }
    uint64_t *vpos = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)db->n_bins);
    for (int b = 0; b < db->n_bins; b++) { /* DatabaseWriter.scala:76-97 */
        const ffo_bin *bin = &db->bins[b];
        if (!bin->longs) { ffo_set_error("bin %d not set", b); fclose(w->f); free(w); free(vpos); return -2; }
        vpos[b] = bgzf_position(w);                                  /* oldPos :80 */
        uint8_t *bytes = (uint8_t *)malloc(bin->n_longs * 8);
        ffo_longs_to_bytes(bin->longs, bin->n_longs, bytes);         /* :91 */
        int rc = bgzf_write(w, bytes, bin->n_longs * 8);
        free(bytes);
        if (rc) { fclose(w->f); free(w); free(vpos); ffo_set_error("write failed"); return -3; }
    }
    if (bgzf_close(w)) { free(w); free(vpos); ffo_set_error("close failed"); return -4; }
    free(w);
    char hp[4096];
    snprintf(hp, sizeof hp, "%s.header", path);
    FILE *h = fopen(hp, "w");
    if (!h) { free(vpos); ffo_set_error("cannot create %s", hp); return -5; }
    fprintf(h, "%lld\n1\n%d\n%d\n", (long long)FFO_MAGIC, db->pack->index, db->n_bins); /* BinaryHeader.scala:72-79 */
    char name[16];
    for (int b = 0; b < db->n_bins; b++) {                                             /* :82-92 */
        ffo_bin_name(db->bin_width, (uint32_t)b, name);
        fprintf(h, "%s=%llu,%zu,%d\n", name, (unsigned long long)vpos[b], db->bins[b].n_longs * 8, db->bins[b].n_targets);
    }
    for (int c = 0; c < db->n_contigs; c++) fprintf(h, "%s=%d\n", db->contigs[c], c + 1); /* :94-96 */
    fclose(h);
    free(vpos);
    return 0;
}

static int read_line(FILE *f, char *buf, size_t n) {
    if (!fgets(buf, (int)n, f)) return 0;
    size_t l = strlen(buf);
    while (l && (buf[l - 1] == '\n' || buf[l - 1] == '\r')) buf[--l] = 0;
    return 1;
}

/* header only (what `score` needs, ScoreResults.scala:91); body == NULL skips the BGZF file */
static ffo_db *db_read_impl(const char *path, int with_body) { /* readHeader :115-160 */
    char hp[4096], line[8192];
    snprintf(hp, sizeof hp, "%s.header", path);
    FILE *h = fopen(hp, "r");
    if (!h) { ffo_set_error("cannot open %s", hp); return NULL; }
    long long magic = 0, version = 0, nb = 0;
    int enz = 0;
    if (!read_line(h, line, sizeof line) || (magic = atoll(line)) != FFO_MAGIC) { fclose(h); ffo_set_error("Binary file %s doesn't have the magic number expected at the top of the file", hp); return NULL; }
    if (!read_line(h, line, sizeof line) || (version = atoll(line)) != 1) { fclose(h); ffo_set_error("Binary file %s doesn't have the correct version", hp); return NULL; }
    if (!read_line(h, line, sizeof line)) { fclose(h); return NULL; }
    enz = atoi(line);
    if (!read_line(h, line, sizeof line)) { fclose(h); return NULL; }
    nb = atoll(line);
    int width = (int)(log((double)nb) / log(4.0)); /* :132 */
    ffo_db *db = ffo_db_new(enz, width);
    if (!db) { fclose(h); return NULL; }
    uint64_t *vpos = (uint64_t *)calloc((size_t)db->n_bins, 8);
    size_t *usize = (size_t *)calloc((size_t)db->n_bins, sizeof(size_t));
    char name[16];
    for (int b = 0; b < db->n_bins; b++) { /* :140-150 */
        ffo_bin_name(width, (uint32_t)b, name);
        unsigned long long vp = 0, us = 0;
        int nt = 0;
        char got[64];
        if (!read_line(h, line, sizeof line) || sscanf(line, "%63[^=]=%llu,%llu,%d", got, &vp, &us, &nt) != 4) {
            ffo_set_error("Missing line for bin %s", name); goto fail;
        }
        if (strcmp(got, name) != 0) { ffo_set_error("Failed to verify bin name, expected: %s isn't what we got %s", name, got); goto fail; } /* :147 */
        vpos[b] = vp; usize[b] = (size_t)us; db->bins[b].n_targets = nt;
    }
    while (read_line(h, line, sizeof line)) { /* :152-155: ids re-assigned in file order */
        if (!line[0]) continue;
        char *eq = strchr(line, '=');
        if (eq) *eq = 0;
        ffo_db_add_contig(db, line);
    }
    fclose(h); h = NULL;
    if (with_body) {
        bgzf_image im;
        if (bgzf_read_all(path, &im)) goto fail;
        for (int b = 0; b < db->n_bins; b++) { /* fillBlock: seek + read uncompressedSize bytes, SeekTraverser.scala:113-120 */
            int64_t lin = usize[b] ? bgzf_linear_offset(&im, vpos[b]) : 0;
            if (lin < 0 || (size_t)lin + usize[b] > im.n || usize[b] % 8) { ffo_set_error("bin %d: bad block pointer", b); bgzf_image_free(&im); goto fail; }
            db->bins[b].n_longs = usize[b] / 8;
            db->bins[b].longs = (int64_t *)malloc(usize[b] ? usize[b] : 8);
            ffo_bytes_to_longs(im.data + lin, usize[b], db->bins[b].longs); /* Utils.byteArrayToLong */
        }
        bgzf_image_free(&im);
    }
    free(vpos); free(usize);
    return db;
fail:
    if (h) fclose(h);
    free(vpos); free(usize);
    ffo_db_free(db);
    return NULL;
}
ffo_db *ffo_db_read(const char *path) { return db_read_impl(path, 1); }

/* ------------------------------------------------------------------------------------------------
 * guide discovery -- reference/ReferenceEncoder.scala:104-175 with the regexes of
 * standards/StandardScanParameters.scala:104-106,126-128,148-150,170-172,192-194,209-211
 * ---------------------------------------------------------------------------------------------- */
static int is_acgt(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }
static int all_acgt(const char *s, int n) { for (int i = 0; i < n; i++) if (!is_acgt(s[i])) return 0; return 1; }

static int fwd_match(const ffo_pack *p, const char *s, size_t rem) {
    int L = p->scan_len;
    if (rem < (size_t)L) return 0;
    switch (p->index) {
        case 1: return s[0] == 'T' && s[1] == 'T' && s[2] == 'T' && all_acgt(s + 3, 21);               /* (T)(?=(TT[ACGT]{21})) */
        case 2: case 5: return all_acgt(s, L - 2) && (s[L - 2] == 'A' || s[L - 2] == 'G') && s[L - 1] == 'G'; /* N{21|20}[AG]G */
        case 3: case 6: return all_acgt(s, L - 2) && s[L - 2] == 'G' && s[L - 1] == 'G';
        case 4: return all_acgt(s, L - 2) && s[L - 2] == 'A' && s[L - 1] == 'G';
    }
    return 0;
}
static int rev_match(const ffo_pack *p, const char *s, size_t rem) {
    int L = p->scan_len;
    if (rem < (size_t)L) return 0;
    switch (p->index) {
        case 1: return all_acgt(s, 21) && s[21] == 'A' && s[22] == 'A' && s[23] == 'A';                /* ([ACGT])(?=([ACGT]{20}AAA)) */
        case 2: case 5: return s[0] == 'C' && (s[1] == 'C' || s[1] == 'T') && all_acgt(s + 2, L - 2); /* ([C])(?=([CT][ACGT]{21|20})) */
        case 3: case 6: return s[0] == 'C' && s[1] == 'C' && all_acgt(s + 2, L - 2);
        case 4: return s[0] == 'C' && s[1] == 'T' && all_acgt(s + 2, L - 2);
    }
    return 0;
}
static void revcomp(const char *s, int n, char *out) { /* Utils.reverseCompString, utils/Utils.scala:81-88 */
    for (int i = 0; i < n; i++) {
        char c = s[n - 1 - i];
        out[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
    }
    out[n] = 0;
}

int ffo_find_sites(const ffo_pack *p, const char *seq, size_t len, int flank, ffo_site *out, int cap) { /* SimpleSiteFinder.reset :114-169 */
    int n = 0, L = p->scan_len;
    for (int pass = 0; pass < 2; pass++) /* all forward matches first (:121-141), then reverse (:145-163) */
        for (size_t i = 0; i + (size_t)L <= len; i++) {
            int ok = pass == 0 ? fwd_match(p, seq + i, len - i) : rev_match(p, seq + i, len - i);
            if (!ok) continue;
            if (n < cap) {
                ffo_site *s = &out[n];
                s->start = (int)i;
                s->forward = pass == 0;
                size_t cs = i >= (size_t)flank ? i - (size_t)flank : 0;             /* math.max(0, start - flank) */
                size_t ce = i + (size_t)L + (size_t)flank; if (ce > len) ce = len;   /* slice clamps */
                s->has_context = (int)(ce - cs) == L + 2 * flank && (L + 2 * flank) < (int)sizeof s->context; /* :131-134 */
                if (pass == 0) {
                    memcpy(s->bases, seq + i, (size_t)L); s->bases[L] = 0;
                    if (s->has_context) { memcpy(s->context, seq + cs, ce - cs); s->context[ce - cs] = 0; }
                } else {
                    revcomp(seq + i, L, s->bases);
                    if (s->has_context) revcomp(seq + cs, (int)(ce - cs), s->context);
                }
                if (!s->has_context) s->context[0] = 0;
            }
            n++;
        }
    return n;
}

/* FASTA reader: contig name = header minus '>' with ' ' and '\t' -> '_' (:56), lines upper-cased (:63) */
typedef struct fasta_contig { char *name; char *seq; size_t len; } fasta_contig;
static int read_fasta(const char *path, fasta_contig **out) {
    FILE *f = fopen(path, "r");
    if (!f) { ffo_set_error("cannot open %s", path); return -1; }
    fasta_contig *cs = NULL;
    int n = 0;
    size_t cap = 0;
    char *line = NULL;
    size_t lcap = 0;
    ssize_t r;
    while ((r = getline(&line, &lcap, f)) >= 0) {
        while (r && (line[r - 1] == '\n' || line[r - 1] == '\r')) line[--r] = 0;
        if (line[0] == '>') {
            cs = (fasta_contig *)realloc(cs, sizeof(fasta_contig) * (size_t)(n + 1));
            for (char *c = line + 1; *c; c++) if (*c == ' ' || *c == '\t') *c = '_';
            cs[n].name = strdup(line + 1); cs[n].seq = NULL; cs[n].len = 0; cap = 0; n++;
        } else if (n > 0) {
            fasta_contig *c = &cs[n - 1];
            if (c->len + (size_t)r + 1 > cap) { cap = (c->len + (size_t)r + 1) * 2; c->seq = (char *)realloc(c->seq, cap); }
            for (ssize_t i = 0; i < r; i++) c->seq[c->len + (size_t)i] = (char)toupper((unsigned char)line[i]);
            c->len += (size_t)r;
            c->seq[c->len] = 0;
        }
    }
    free(line);
    fclose(f);
    *out = cs;
    return n;
}
static void free_fasta(fasta_contig *cs, int n) { for (int i = 0; i < n; i++) { free(cs[i].name); free(cs[i].seq); } free(cs); }

/* ------------------------------------------------------------------------------------------------
 * index -- modules/BuildOffTargetDatabase.scala:57-89, BlockReader.scala:87-159
 * ---------------------------------------------------------------------------------------------- */
typedef struct idx_site { char bases[25]; uint64_t pos; size_t seq; } idx_site;
static int idx_cmp(const void *a, const void *b) { /* CRISPRSite.compare = bases compare (CRISPRSite.scala:47); stable by discovery order */
    const idx_site *x = (const idx_site *)a, *y = (const idx_site *)b;
    int c = strcmp(x->bases, y->bases);
    if (c) return c;
    return x->seq < y->seq ? -1 : x->seq > y->seq;
}

int ffo_index_fasta(const char *fasta, const char *db_path, const char *enzyme, int bin_width) {
    const ffo_pack *p = ffo_pack_by_name(enzyme);
    if (!p) { ffo_set_error("Unable to find the correct parameter pack for enzyme: %s", enzyme); return -1; }
    fasta_contig *cs;
    int nc = read_fasta(fasta, &cs);
    if (nc < 0) return -2;
    ffo_db *db = ffo_db_new(p->index, bin_width);
    idx_site *sites = NULL;
    size_t ns = 0, cap = 0;
    for (int c = 0; c < nc; c++) {
        int id = ffo_db_add_contig(db, cs[c].name);
        int n = ffo_find_sites(p, cs[c].seq ? cs[c].seq : "", cs[c].len, 0, NULL, 0);
        ffo_site *tmp = (ffo_site *)malloc(sizeof(ffo_site) * (size_t)(n > 0 ? n : 1));
        ffo_find_sites(p, cs[c].seq ? cs[c].seq : "", cs[c].len, 0, tmp, n);
        if (ns + (size_t)n > cap) { cap = (ns + (size_t)n) * 2 + 16; sites = (idx_site *)realloc(sites, sizeof(idx_site) * cap); }
        for (int i = 0; i < n; i++) {
            strcpy(sites[ns].bases, tmp[i].bases);
            sites[ns].pos = ffo_pos_encode(id, (uint32_t)tmp[i].start, p->scan_len, tmp[i].forward); /* BlockReader.scala:113 */
            sites[ns].seq = ns;
            ns++;
        }
        free(tmp);
    }
    free_fasta(cs, nc);
    qsort(sites, ns, sizeof(idx_site), idx_cmp);
    /* dedup runs of identical bases: TargetPos.combine, BlockReader.scala:138-159 (count and positions capped at Short.MaxValue) */
    uint64_t *targets = (uint64_t *)malloc(8 * (ns ? ns : 1)), *positions = (uint64_t *)malloc(8 * (ns ? ns : 1));
    size_t nt = 0, np = 0;
    for (size_t i = 0; i < ns;) {
        size_t j = i;
        while (j < ns && strcmp(sites[j].bases, sites[i].bases) == 0) j++;
        size_t cnt = j - i;
        if (cnt > 32767) cnt = 32767;
        uint64_t enc;
        ffo_bit_encode(sites[i].bases, p->scan_len, (int)cnt, &enc);
        targets[nt++] = enc;
        for (size_t k = 0; k < cnt; k++) positions[np++] = sites[i + k].pos;
        i = j;
    }
    free(sites);
    int rc = ffo_db_build_from_sorted(db, targets, positions, nt, 500); /* maxTargetsPerLinearBin = 500, DatabaseWriter.scala:66 */
    free(targets); free(positions);
    if (!rc) rc = ffo_db_write(db, db_path);
    ffo_db_free(db);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * table writer -- targetio/TabDelimitedHandler.scala:103-159, crispr/CRISPRHit.scala:54-104
 * ---------------------------------------------------------------------------------------------- */
typedef struct score_cols { /* which models were requested, in order (ScoreResults.scala:101-121) */
    int n;
    int kind[8]; /* 0 hsu2013, 1 doench2016cfd, 2 minot, 3 dangerous, 4 jostandsantos, 5 reciprocalofftargets */
    int max_reciprocal; /* ReciprocalOffTargets.maxMismatch (ScoreResults.scala:85-87, default 1) */
} score_cols;

static double gc_content(const char *s) { /* utils/Utils.scala:46 */
    int gc = 0, n = (int)strlen(s);
    for (int i = 0; i < n; i++) { char c = (char)toupper((unsigned char)s[i]); if (c == 'C' || c == 'G') gc++; }
    return (double)gc / (double)n;
}

static void write_hit(FILE *f, const ffo_db *db, const ffo_guide_ot *g, int hi, int positions) { /* CRISPRHit.toOutput :54-88 */
    const ffo_pack *p = db->pack;
    const ffo_hit *h = &g->hits[hi];
    char s[32];
    int cnt = ffo_bit_decode(h->target, p->scan_len, s);
    fprintf(f, "%s_%d_%d", s, cnt, ffo_mismatches(p, g->encoding, h->target, FFO_STRING_MASK));
    if (!positions) return;
    if (g->valid_coords && h->n_pos > 0) { /* :58-74 */
        fputc('<', f);
        for (int k = 0; k < h->n_pos; k++) {
            int cid, size, fwd;
            uint32_t start;
            ffo_pos_decode(h->positions[k], &cid, &start, &size, &fwd);
            const char *cn = ffo_db_contig(db, cid);
            fprintf(f, "%s%s:%d^%s", k ? "|" : "", cn ? cn : "?", (int)start, fwd ? "F" : "R");
        }
        fputc('>', f);
    }
    if (g->hit_cfd && !isnan(g->hit_cfd[hi])) { /* toOutputScores :93-104 -- only CFD attaches a per-hit score */
        char d[40];
        ffo_java_double_to_string(g->hit_cfd[hi], d);
        fprintf(f, "{Doench2016CFDScore=%s}", d);
    }
}

static void write_table(FILE *f, const ffo_db *db, ffo_guide_ot *guides, const int *order, int n, const score_cols *sc,
                        int write_ots, int write_positions) {
    const ffo_pack *p = db->pack;
    fprintf(f, "contig\tstart\tstop\ttarget\tcontext\toverflow\torientation"); /* default_columns :79 */
    for (int m = 0; sc && m < sc->n; m++) switch (sc->kind[m]) {
        case 0: fprintf(f, "\tHsu2013"); break;
        case 1: fprintf(f, "\tDoenchCFD_maxOT\tDoenchCFD_specificityscore"); break;
        case 2: fprintf(f, "\tbasesDiffToClosestHit\tclosestHitCount\t0-1-2-3-4_mismatch"); break;
        case 3: fprintf(f, "\tdangerous_GC\tdangerous_polyT\tdangerous_in_genome"); break;
        case 4: fprintf(f, "\tJostCRISPRi_maxOT\tJostCRISPRi_specificityscore"); break; /* JostAndSantosCRISPRi.scala:132-134 */
        case 5: fprintf(f, "\tReciprocalOffTargets"); break;                              /* ReciprocalOffTargets.scala:98 */
    }
    fprintf(f, write_ots ? "\totCount\toffTargets\n" : "\totCount\n"); /* :122-125 */
    for (int r = 0; r < n; r++) {
        ffo_guide_ot *g = &guides[order[r]];
        int L = (int)strlen(g->bases);
        int full = g->current_total >= g->overflow;
        fprintf(f, "%s\t%d\t%d\t%s\t%s\t%s\t%s\t", g->contig, g->start, g->start + L, g->bases,
                g->context ? g->context : "NONE", (full || g->inherited_overflow) ? "OVERFLOW" : "OK", g->forward ? "FWD" : "RVS"); /* :133-139 */
        if (sc && sc->n) {
            uint64_t *t = (uint64_t *)malloc(8 * (size_t)(g->n_hits > 0 ? g->n_hits : 1));
            for (int i = 0; i < g->n_hits; i++) t[i] = g->hits[i].target;
            ffo_guide_scores s;
            ffo_score_guide(p, g->encoding, t, g->n_hits, &s, NULL);
            free(t);
            char d[40];
            for (int m = 0; m < sc->n; m++) switch (sc->kind[m]) {
                case 0: ffo_java_double_to_string(s.hsu, d); fprintf(f, "%s\t", d); break;
                case 1:
                    if (s.cfd_max >= 0.023) { ffo_java_double_to_string(s.cfd_max, d); fprintf(f, "%s\t", d); } /* Doench2016CFDScore.scala:83-87 */
                    else fprintf(f, "0.0\t");
                    ffo_java_double_to_string(s.cfd_spec, d); fprintf(f, "%s\t", d);
                    break;
                case 2:
                    if (s.closest == INT_MAX) fprintf(f, "UNK\t0\t"); else fprintf(f, "%d\t%d\t", s.closest, s.closest_count); /* ClosestHit.scala:71-75 */
                    fprintf(f, "%d,%d,%d,%d,%d\t", s.hist[0], s.hist[1], s.hist[2], s.hist[3], s.hist[4]);
                    break;
                case 3: { /* DangerousSequences.scala:49-68 (annotated form) */
                    double gc = gc_content(g->bases);
                    if (gc < .25 || gc > .75) { ffo_java_double_to_string(gc, d); fprintf(f, "GC_%s\t", d); } else fprintf(f, "NONE\t");
                    char guide[32];
                    int gl = p->guide_hi - p->guide_lo;
                    memcpy(guide, g->bases + p->guide_lo, (size_t)gl); guide[gl] = 0;
                    fprintf(f, strstr(guide, "TTTT") ? "PolyT\t" : "NONE\t");
                    if (g->n_hits > 0 && s.in_genome > 0) fprintf(f, "IN_GENOME=%d\t", s.in_genome); else fprintf(f, "NONE\t");
                } break;
                case 4: /* JostAndSantosCRISPRi.scoreGuide :42-45 (0.0.toString when nothing was scored) */
                    ffo_java_double_to_string(s.jost_max, d); fprintf(f, "%s\t", d);
                    ffo_java_double_to_string(s.jost_spec, d); fprintf(f, "%s\t", d);
                    break;
                case 5: { /* ReciprocalOffTargets.scoreGuides :54-62: every OTHER guide of the file, in file order, within maxMismatch */
                    int any = 0;
                    for (int o = 0; o < n; o++) {
                        int mmr = ffo_mismatches(p, g->encoding, guides[o].encoding, FFO_STRING_MASK);
                        if (mmr != 0 && mmr <= sc->max_reciprocal) { fprintf(f, "%s%s", any ? "," : "", guides[o].bases); any = 1; }
                    }
                    fprintf(f, any ? "\t" : "NA\t"); /* SingleGuideScoreModel.missingAnnotation, TabDelimitedHandler.scala:142 */
                } break;
            }
        }
        long total = 0;
        for (int i = 0; i < g->n_hits; i++) total += g->hits[i].n_pos; /* :147 */
        fprintf(f, "%ld", total);
        if (write_ots) {
            fputc('\t', f);
            for (int i = 0; i < g->n_hits; i++) { if (i) fputc(',', f); write_hit(f, db, g, i, write_positions); } /* :149-151 */
        }
        fputc('\n', f);
    }
}

static const ffo_guide_ot *g_sort_base;
static int cmp_start(const void *a, const void *b) { /* CRISPRSiteOT.compare :64 -- by start; ties keep input order (the reference's quicksort is unstable) */
    int x = *(const int *)a, y = *(const int *)b;
    if (g_sort_base[x].start != g_sort_base[y].start) return g_sort_base[x].start < g_sort_base[y].start ? -1 : 1;
    return x < y ? -1 : x > y;
}

/* ------------------------------------------------------------------------------------------------
 * discover -- modules/OffTargetDiscovery.scala:79-153
 * ---------------------------------------------------------------------------------------------- */
int ffo_discover_fasta(const char *db_path, const char *fasta, const char *out_path, int max_mm, int max_ot, int flank,
                       int position_output, int force_linear, double min_gc, double max_gc) {
    ffo_db *db = ffo_db_read(db_path); /* :89 */
    if (!db) return -1;
    const ffo_pack *p = db->pack;
    fasta_contig *cs;
    int nc = read_fasta(fasta, &cs);
    if (nc < 0) { ffo_db_free(db); return -2; }
    ffo_guide_ot *meta = NULL;
    uint64_t *enc = NULL;
    int ng = 0;
    for (int c = 0; c < nc; c++) { /* findTargetSites :93 */
        const char *seq = cs[c].seq ? cs[c].seq : "";
        int n = ffo_find_sites(p, seq, cs[c].len, flank, NULL, 0);
        ffo_site *tmp = (ffo_site *)malloc(sizeof(ffo_site) * (size_t)(n > 0 ? n : 1));
        ffo_find_sites(p, seq, cs[c].len, flank, tmp, n);
        meta = (ffo_guide_ot *)realloc(meta, sizeof(ffo_guide_ot) * (size_t)(ng + n + 1));
        enc = (uint64_t *)realloc(enc, 8 * (size_t)(ng + n + 1));
        for (int i = 0; i < n; i++) {
            double gc = gc_content(tmp[i].bases); /* filter_by_GC :96, GuideMemoryStorage.scala:41-50 */
            if (!(gc >= min_gc && gc <= max_gc)) continue;
            memset(&meta[ng], 0, sizeof(ffo_guide_ot));
            meta[ng].contig = strdup(cs[c].name);
            meta[ng].bases = strdup(tmp[i].bases);
            meta[ng].context = tmp[i].has_context ? strdup(tmp[i].context) : NULL;
            meta[ng].start = tmp[i].start;
            meta[ng].forward = tmp[i].forward;
            ffo_bit_encode(tmp[i].bases, p->scan_len, 1, &enc[ng]); /* :100-102 */
            ng++;
        }
        free(tmp);
    }
    free_fasta(cs, nc);
    ffo_result *r = ffo_discover(db, enc, ng, max_mm, max_ot, force_linear);
    if (!r) { ffo_db_free(db); free(enc); free(meta); return -3; }
    for (int i = 0; i < ng; i++) {
        r->guides[i].contig = meta[i].contig; r->guides[i].bases = meta[i].bases; r->guides[i].context = meta[i].context;
        r->guides[i].start = meta[i].start; r->guides[i].forward = meta[i].forward; r->guides[i].valid_coords = 1;
    }
    int *order = (int *)malloc(sizeof(int) * (size_t)(ng > 0 ? ng : 1));
    for (int i = 0; i < ng; i++) order[i] = i;
    g_sort_base = r->guides;
    qsort(order, (size_t)ng, sizeof(int), cmp_start); /* ResultsAggregator.scala:35 */
    FILE *f = fopen(out_path, "w");
    int rc = 0;
    if (!f) { ffo_set_error("cannot create %s", out_path); rc = -4; }
    else { write_table(f, db, r->guides, order, ng, NULL, 1, position_output); fclose(f); } /* :141-152 */
    free(order); free(enc); free(meta);
    ffo_result_free(r);
    ffo_db_free(db);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * table reader + score -- targetio/TabDelimitedHandler.scala:169-335, modules/ScoreResults.scala:90-154
 * ---------------------------------------------------------------------------------------------- */
static int split_java(char *line, char sep, char **tok, int cap) { /* String.split: trailing empty strings removed */
    int n = 0;
    char *s = line;
    for (;;) {
        if (n < cap) tok[n] = s;
        n++;
        char *e = strchr(s, sep);
        if (!e) break;
        *e = 0;
        s = e + 1;
    }
    if (n > cap) n = cap;
    while (n > 0 && tok[n - 1][0] == 0) n--;
    return n;
}

static int contig_id(const ffo_db *db, const char *name) {
    for (int i = 0; i < db->n_contigs; i++) if (strcmp(db->contigs[i], name) == 0) return i + 1;
    return -1;
}

static int add_offtarget_token(const ffo_db *db, ffo_guide_ot *g, char *token, int max_mm) { /* addOffTargets :277-334 */
    const ffo_pack *p = db->pack;
    char *scores = strchr(token, '{');
    char *us1 = strchr(token, '_');
    if (!us1) { ffo_set_error("bad off-target token"); return -1; }
    char *us2 = strchr(us1 + 1, '_');
    if (!us2) { ffo_set_error("bad off-target token"); return -1; }
    *us1 = 0; *us2 = 0;
    const char *seq = token;
    int count = atoi(us1 + 1);
    char *lt = strchr(us2 + 1, '<');
    int mm = atoi(us2 + 1); /* digits up to '<' or end (:287-291) */
    if (mm > max_mm) return 0; /* :293 */
    if (count > 32767) { ffo_set_error("The count was too large to encode in a Scala Short value"); return -2; }
    uint64_t enc;
    if (ffo_bit_encode(seq, (int)strlen(seq), count, &enc)) return -3;
    uint64_t *pos;
    int npos;
    if (lt) { /* :296-309 */
        char *gt = strchr(lt, '>');
        if (gt) *gt = 0;
        npos = 1;
        for (char *c = lt + 1; *c; c++) if (*c == '|') npos++;
        pos = (uint64_t *)malloc(8 * (size_t)npos);
        char *ptok[1];
        (void)ptok;
        int k = 0;
        for (char *s = lt + 1; s && *s;) {
            char *bar = strchr(s, '|');
            if (bar) *bar = 0;
            char *colon = strrchr(s, ':');
            char *caret = strchr(s, '^');
            if (!colon || !caret) { free(pos); ffo_set_error("bad position token"); return -4; }
            /* contig = split(':')(0): contigs containing ':' would break the reference too */
            colon = strchr(s, ':');
            *colon = 0; *caret = 0;
            int cid = contig_id(db, s);
            if (cid < 0) { free(pos); ffo_set_error("Unknown contig: %s", s); return -5; } /* BitPosition.encode :52 */
            pos[k++] = ffo_pos_encode(cid, (uint32_t)atoi(colon + 1), (int)strlen(seq), caret[1] == 'F' && caret[2] == 0);
            s = bar ? bar + 1 : NULL;
        }
        npos = k;
    } else { /* :310-316 zero-filled coordinates, validOffTargetCoordinates = false */
        npos = count;
        pos = (uint64_t *)calloc((size_t)(npos > 0 ? npos : 1), 8);
        g->valid_coords = 0;
    }
    g->owned = (uint64_t **)realloc(g->owned, sizeof(uint64_t *) * (size_t)(g->n_owned + 1));
    g->owned[g->n_owned++] = pos;
    if (!(g->current_total >= g->overflow)) { /* if (!ot.full) ot.addOT(otHit) :305-306 */
        if (g->n_hits == g->cap_hits) { g->cap_hits = g->cap_hits ? g->cap_hits * 2 : 16; g->hits = (ffo_hit *)realloc(g->hits, sizeof(ffo_hit) * (size_t)g->cap_hits); }
        g->hits[g->n_hits].target = enc; g->hits[g->n_hits].positions = pos; g->hits[g->n_hits].n_pos = npos;
        g->n_hits++;
        g->current_total += npos;
    }
    (void)scores; (void)p; /* per-hit scores of the input are re-computed by the models that run; foreign keys are dropped */
    return 0;
}

int ffo_score_file(const char *db_path, const char *in_path, const char *out_path, const char *metrics_csv, int max_mm,
                   int include_ots, int max_reciprocal) {
    ffo_db *db = db_read_impl(db_path, 0); /* header only, ScoreResults.scala:91 */
    if (!db) return -1;
    const ffo_pack *p = db->pack;
    score_cols sc = {0, {0}, max_reciprocal};
    int want_cfd = 0;
    {
        char *m = strdup(metrics_csv), *save = NULL;
        for (char *t = strtok_r(m, ",", &save); t; t = strtok_r(NULL, ",", &save)) { /* getRegisteredScoringMetric :159-226 */
            int kind = -1;
            if (!strcasecmp(t, "hsu2013")) kind = 0;
            else if (!strcasecmp(t, "doench2016cfd")) kind = 1;
            else if (!strcasecmp(t, "minot")) kind = 2;
            else if (!strcasecmp(t, "dangerous")) kind = 3;
            else if (!strcasecmp(t, "jostandsantos")) kind = 4;
            else if (!strcasecmp(t, "reciprocalofftargets")) kind = 5;
            else { ffo_set_error("Unknown scoring metric: %s", t); free(m); ffo_db_free(db); return -2; }
            if ((kind == 0 || kind == 1) && !p->cas9_23) continue; /* validOverEnzyme -> dropped :111-118 */
            if (kind == 4 && !(p->index != 1 && (p->scan_len == 23 || p->scan_len == 22))) continue; /* JostAndSantosCRISPRi.scala:53-58 */
            if (kind == 1) want_cfd = 1;
            if (sc.n < 8) sc.kind[sc.n++] = kind;
        }
        free(m);
    }
    FILE *f = fopen(in_path, "r");
    if (!f) { ffo_set_error("cannot open %s", in_path); ffo_db_free(db); return -3; }
    char *line = NULL;
    size_t lcap = 0;
    ssize_t r = getline(&line, &lcap, f);
    if (r < 0) { fclose(f); ffo_db_free(db); return -4; }
    while (r && (line[r - 1] == '\n' || line[r - 1] == '\r')) line[--r] = 0;
    char *htok[64];
    int nh = split_java(line, '\t', htok, 64);
    int with_ots = nh >= 2 && !strcmp(htok[nh - 2], "otCount") && !strcmp(htok[nh - 1], "offTargets"); /* :191-192 */
    int n_annot = nh - 7 - (with_ots ? 2 : 1);                                                       /* :198 */
    int with_ot_token_len = nh;
    ffo_guide_ot *guides = NULL;
    int ng = 0, rc = 0;
    while (!rc && (r = getline(&line, &lcap, f)) >= 0) { /* extractCRISPRSiteOT :225-268 */
        while (r && (line[r - 1] == '\n' || line[r - 1] == '\r')) line[--r] = 0;
        char *tok[80];
        int nt = split_java(line, '\t', tok, 80);
        if (nt < 8 + n_annot) { ffo_set_error("Unable to parse line"); rc = -5; break; }
        int overflowed = strcmp(tok[5], "OK") != 0;                                  /* :240 */
        int otc = atoi(tok[7 + n_annot]);
        ffo_guide_ot g;
        memset(&g, 0, sizeof g);
        g.overflow = overflowed ? otc : otc + 1;                                     /* :241-245 */
        g.inherited_overflow = overflowed;
        g.valid_coords = 1;
        if (ffo_bit_encode(tok[3], (int)strlen(tok[3]), 1, &g.encoding)) { rc = -6; break; }
        if (with_ots && nt == with_ot_token_len) {                                   /* :253 */
            char *s = tok[nt - 1];
            while (s && *s && !rc) {
                char *comma = strchr(s, ',');
                if (comma) *comma = 0;
                if (add_offtarget_token(db, &g, s, max_mm)) rc = -7;
                s = comma ? comma + 1 : NULL;
            }
        }
        int keep = !g.inherited_overflow && !(g.current_total >= g.overflow);       /* filterOutOverflowedGuides=true :259 */
        if (keep && !rc) {
            g.contig = strdup(tok[0]); g.start = atoi(tok[1]); g.bases = strdup(tok[3]);
            g.context = strcmp(tok[4], "NONE") ? strdup(tok[4]) : NULL;
            g.forward = !strcmp(tok[6], "FWD");
            if (want_cfd) { /* Doench2016CFDScore attaches pam*cfd to each scored hit (:72) */
                g.hit_cfd = (double *)malloc(sizeof(double) * (size_t)(g.n_hits > 0 ? g.n_hits : 1));
                uint64_t *t = (uint64_t *)malloc(8 * (size_t)(g.n_hits > 0 ? g.n_hits : 1));
                for (int i = 0; i < g.n_hits; i++) t[i] = g.hits[i].target;
                ffo_guide_scores s;
                ffo_score_guide(p, g.encoding, t, g.n_hits, &s, g.hit_cfd);
                free(t);
            }
            guides = (ffo_guide_ot *)realloc(guides, sizeof(ffo_guide_ot) * (size_t)(ng + 1));
            guides[ng++] = g;
        } else {
            free(g.hits);
            for (int k = 0; k < g.n_owned; k++) free(g.owned[k]);
            free(g.owned);
        }
    }
    free(line);
    fclose(f);
    if (!rc) {
        int *order = (int *)malloc(sizeof(int) * (size_t)(ng > 0 ? ng : 1));
        for (int i = 0; i < ng; i++) order[i] = i;
        g_sort_base = guides;
        qsort(order, (size_t)ng, sizeof(int), cmp_start); /* ScoreResults.scala:137 */
        FILE *o = fopen(out_path, "w");
        if (!o) { ffo_set_error("cannot create %s", out_path); rc = -8; }
        else { write_table(o, db, guides, order, ng, &sc, include_ots, 1); fclose(o); } /* :142-153 (writePositions = true) */
        free(order);
    }
    ffo_result tmp = {guides, ng, 0};
    ffo_result *heap = (ffo_result *)malloc(sizeof *heap);
    *heap = tmp;
    ffo_result_free(heap);
    ffo_db_free(db);
    return rc;
}
