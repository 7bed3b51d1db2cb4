/* hamgnn_tables.h -- the on-disk container of kernel launch tables (hamgnn_amd/export.py) and a minimal C loader.
 *
 * The launch tables of include/hamgnn_hip.h (segment / block / phase / group / item / part / row tables + the packed weight blob) are built by the
 * Python planner (hamgnn_amd/plan.py) from a model's irreps and weights.  A host that is not Python loads them from a file written once by
 * `python -c "from hamgnn_amd import export; export.export_tp_is(device_program, 'block.hgprog', rows)"`:
 *
 *   bytes 0..7    magic "HGPROG1\0"
 *   bytes 8..15   uint64 (little-endian) H = length of the JSON header
 *   bytes 16..    JSON header (ASCII, padded with spaces): {"format": 1, "entry": "hg_tp_is", "hidden": .., "out_dim": .., "lds_bytes": ..,
 *                 "nparts": .., "zero_fill_out": true|false, "arrays": [{"name": "weights", "dtype": "f32", "shape": [n], "offset": bytes from the
 *                 start of the file (64-byte aligned), "nbytes": ..}, {"name": "seg_table", "dtype": "i32", "shape": [nseg, 8], ...}, ...]}
 *   then the raw little-endian arrays at their offsets: weights, seg_table, block_table, phase_table, group_table, item_table, part_table, row_table
 *   (the arguments of hg_tp_is of the same names; part_table doubles as part_table_host).
 *
 * The loader below does not parse general JSON: it finds `"name": "<array>"` and reads the integer fields that follow it, which is all the format
 * needs (the writer emits the keys in a fixed order).  C99, no dependencies.  examples/run_tp_is.c is a complete host built on it.                  */
#ifndef HAMGNN_TABLES_H
#define HAMGNN_TABLES_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { const void* data; int64_t nbytes; int64_t shape[2]; int is_f32; } HgArray;
typedef struct {
    unsigned char* file;              /* the whole file (owned; free with hg_prog_free) */
    int64_t file_bytes;
    int hidden, out_dim, lds_bytes, nparts, zero_fill_out;
    HgArray weights, seg_table, block_table, phase_table, group_table, item_table, part_table, row_table;
} HgProgFile;

static long long hg__int_after(const char* hdr, const char* from, const char* key) {
    const char* p = strstr(from ? from : hdr, key);
    if (!p) return -1;
    p += strlen(key);
    while (*p == ' ' || *p == ':' || *p == '[') ++p;
    if (!strncmp(p, "true", 4)) return 1;
    if (!strncmp(p, "false", 5)) return 0;
    return strtoll(p, NULL, 10);
}

static int hg__array(const HgProgFile* f, const char* hdr, const char* name, HgArray* a) {
    char key[64];
    snprintf(key, sizeof key, "\"name\": \"%s\"", name);
    const char* p = strstr(hdr, key);
    if (!p) return -1;
    const char* sh = strstr(p, "\"shape\"");
    if (!sh) return -1;
    sh = strchr(sh, '[') + 1;
    a->shape[0] = strtoll(sh, (char**)&sh, 10);
    a->shape[1] = (*sh == ',') ? strtoll(sh + 1, NULL, 10) : 1;
    a->is_f32 = strstr(p, "\"dtype\": \"f32\"") != NULL && strstr(p, "\"dtype\": \"f32\"") < strstr(p, "\"shape\"");
    long long off = hg__int_after(hdr, p, "\"offset\""), nb = hg__int_after(hdr, p, "\"nbytes\"");
    if (off < 0 || nb < 0 || off > f->file_bytes || nb > f->file_bytes - off || (off & 63)) return -1;       /* (no off + nb: cannot wrap) */
    if (a->shape[0] < 0 || a->shape[1] < 1 || (a->shape[0] && a->shape[1] > (INT64_MAX / 4) / a->shape[0]) || nb != a->shape[0] * a->shape[1] * 4) return -1;   /* nbytes == shape product x 4 */
    a->data = f->file + off;
    a->nbytes = nb;
    return 0;
}

static void hg_prog_free(HgProgFile* f) { free(f->file); memset(f, 0, sizeof *f); }

/* 0 on success; negative: cannot open / not a container / truncated or inconsistent (nothing stays allocated on failure) */
static int hg_prog_load(const char* path, HgProgFile* f) {
    memset(f, 0, sizeof *f);
    FILE* fp = fopen(path, "rb");
    if (!fp) return -1;
    fseek(fp, 0, SEEK_END);
    f->file_bytes = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    if (f->file_bytes < 0) { fclose(fp); return -2; }
    f->file = (unsigned char*)malloc((size_t)f->file_bytes + 1);
    if (!f->file || fread(f->file, 1, (size_t)f->file_bytes, fp) != (size_t)f->file_bytes) { fclose(fp); hg_prog_free(f); return -2; }
    fclose(fp);
    if (f->file_bytes < 16 || memcmp(f->file, "HGPROG1\0", 8)) { hg_prog_free(f); return -3; }
    uint64_t hlen = 0;
    for (int i = 7; i >= 0; --i) hlen = (hlen << 8) | f->file[8 + i];
    if (hlen > (uint64_t)f->file_bytes - 16) { hg_prog_free(f); return -4; }      /* (not 16 + hlen: a corrupt length must not wrap) */
    char* hdr = (char*)malloc((size_t)hlen + 1);
    if (!hdr) { hg_prog_free(f); return -2; }
    memcpy(hdr, f->file + 16, hlen);
    hdr[hlen] = 0;
    f->hidden = (int)hg__int_after(hdr, NULL, "\"hidden\"");
    f->out_dim = (int)hg__int_after(hdr, NULL, "\"out_dim\"");
    f->lds_bytes = (int)hg__int_after(hdr, NULL, "\"lds_bytes\"");
    f->nparts = (int)hg__int_after(hdr, NULL, "\"nparts\"");
    f->zero_fill_out = (int)hg__int_after(hdr, NULL, "\"zero_fill_out\"");
    int rc = hg__array(f, hdr, "weights", &f->weights) | hg__array(f, hdr, "seg_table", &f->seg_table) | hg__array(f, hdr, "block_table", &f->block_table) |
             hg__array(f, hdr, "phase_table", &f->phase_table) | hg__array(f, hdr, "group_table", &f->group_table) | hg__array(f, hdr, "item_table", &f->item_table) |
             hg__array(f, hdr, "part_table", &f->part_table) | hg__array(f, hdr, "row_table", &f->row_table);
    free(hdr);
    if (rc) { hg_prog_free(f); return -5; }
    return 0;
}
#endif
