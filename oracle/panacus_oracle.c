/*
 * panacus_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See panacus_oracle.h for scope, parity status and the golden vectors that pin it.
 * Plain C11, serial, written to mirror the reference's loop order literally; speed is a
 * non-goal.  Citations are file:line in the reference tree (marschall-lab/panacus v0.4.1).
 */
#define _GNU_SOURCE
#include "panacus_oracle.h"

#include <math.h>
#include <stdarg.h>
#include <ctype.h>
#include <regex.h>
#include <sys/stat.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static char g_err[512];
static void set_err(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char *orc_graph_last_error(void) { return g_err; }
void orc_free(void *p) { free(p); }

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) {
        fprintf(stderr, "oracle: out of memory (%zu bytes)\n", n);
        abort();
    }
    return p;
}
static void *xcalloc(size_t n, size_t s) {
    void *p = calloc(n ? n : 1, s ? s : 1);
    if (!p) {
        fprintf(stderr, "oracle: out of memory\n");
        abort();
    }
    return p;
}
static void *xrealloc(void *q, size_t n) {
    void *p = realloc(q, n ? n : 1);
    if (!p) {
        fprintf(stderr, "oracle: out of memory\n");
        abort();
    }
    return p;
}
static char *xstrndup(const char *s, size_t n) {
    char *p = xmalloc(n + 1);
    memcpy(p, s, n);
    p[n] = 0;
    return p;
}

/* ------------------------------------------------------------------------------------ */
/* byte-string -> u64 map (stands in for HashMap<Vec<u8>, ItemId>, graph.rs:167)         */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    char **keys;
    size_t *klen;
    uint64_t *vals;
    size_t cap, n;
} smap;

static uint64_t fnv1a(const char *s, size_t n) {
    uint64_t h = 1469598103934665603ULL;
    for (size_t i = 0; i < n; i++) {
        h ^= (unsigned char)s[i];
        h *= 1099511628211ULL;
    }
    return h;
}
static void smap_init(smap *m, size_t cap) {
    m->cap = 16;
    while (m->cap < cap * 2) m->cap <<= 1;
    m->keys = xcalloc(m->cap, sizeof *m->keys);
    m->klen = xcalloc(m->cap, sizeof *m->klen);
    m->vals = xcalloc(m->cap, sizeof *m->vals);
    m->n = 0;
}
static void smap_free(smap *m) {
    for (size_t i = 0; i < m->cap; i++) free(m->keys[i]);
    free(m->keys);
    free(m->klen);
    free(m->vals);
    memset(m, 0, sizeof *m);
}
static int smap_put(smap *m, const char *k, size_t n, uint64_t v); /* fwd */
static void smap_grow(smap *m) {
    smap o = *m;
    smap_init(m, o.cap);
    for (size_t i = 0; i < o.cap; i++)
        if (o.keys[i]) {
            size_t j = fnv1a(o.keys[i], o.klen[i]) & (m->cap - 1);
            while (m->keys[j]) j = (j + 1) & (m->cap - 1);
            m->keys[j] = o.keys[i];
            m->klen[j] = o.klen[i];
            m->vals[j] = o.vals[i];
            m->n++;
        }
    free(o.keys);
    free(o.klen);
    free(o.vals);
}
/* returns 1 if inserted, 0 if key existed (value left unchanged) */
static int smap_put(smap *m, const char *k, size_t n, uint64_t v) {
    if ((m->n + 1) * 2 > m->cap) smap_grow(m);
    size_t j = fnv1a(k, n) & (m->cap - 1);
    while (m->keys[j]) {
        if (m->klen[j] == n && memcmp(m->keys[j], k, n) == 0) return 0;
        j = (j + 1) & (m->cap - 1);
    }
    m->keys[j] = xstrndup(k, n);
    m->klen[j] = n;
    m->vals[j] = v;
    m->n++;
    return 1;
}
static int smap_get(const smap *m, const char *k, size_t n, uint64_t *v) {
    size_t j = fnv1a(k, n) & (m->cap - 1);
    while (m->keys[j]) {
        if (m->klen[j] == n && memcmp(m->keys[j], k, n) == 0) {
            *v = m->vals[j];
            return 1;
        }
        j = (j + 1) & (m->cap - 1);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* canonical edge -> id map (HashMap<Edge, ItemId>, graph.rs:276-306)                    */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    uint64_t u, v;
    uint8_t o1, o2; /* 0 = Forward, 1 = Backward (graph.rs:20-24) */
} edge_t;

typedef struct {
    edge_t *keys;
    uint64_t *vals; /* 0 = empty */
    size_t cap, n;
} emap;

static uint64_t edge_hash(edge_t e) {
    uint64_t h = e.u * 0x9E3779B97F4A7C15ULL;
    h ^= (e.v + 0x7F4A7C15ULL) * 0xC2B2AE3D27D4EB4FULL;
    h ^= ((uint64_t)e.o1 << 1 | e.o2) * 0x165667B19E3779F9ULL;
    h ^= h >> 29;
    return h;
}
static int edge_eq(edge_t a, edge_t b) {
    return a.u == b.u && a.v == b.v && a.o1 == b.o1 && a.o2 == b.o2;
}
static void emap_init(emap *m, size_t cap) {
    m->cap = 16;
    while (m->cap < cap * 2) m->cap <<= 1;
    m->keys = xcalloc(m->cap, sizeof *m->keys);
    m->vals = xcalloc(m->cap, sizeof *m->vals);
    m->n = 0;
}
static void emap_grow(emap *m) {
    emap o = *m;
    emap_init(m, o.cap);
    for (size_t i = 0; i < o.cap; i++)
        if (o.vals[i]) {
            size_t j = edge_hash(o.keys[i]) & (m->cap - 1);
            while (m->vals[j]) j = (j + 1) & (m->cap - 1);
            m->keys[j] = o.keys[i];
            m->vals[j] = o.vals[i];
            m->n++;
        }
    free(o.keys);
    free(o.vals);
}
static int emap_put(emap *m, edge_t e, uint64_t v) {
    if ((m->n + 1) * 2 > m->cap) emap_grow(m);
    size_t j = edge_hash(e) & (m->cap - 1);
    while (m->vals[j]) {
        if (edge_eq(m->keys[j], e)) return 0;
        j = (j + 1) & (m->cap - 1);
    }
    m->keys[j] = e;
    m->vals[j] = v;
    m->n++;
    return 1;
}
static uint64_t emap_get(const emap *m, edge_t e) {
    size_t j = edge_hash(e) & (m->cap - 1);
    while (m->vals[j]) {
        if (edge_eq(m->keys[j], e)) return m->vals[j];
        j = (j + 1) & (m->cap - 1);
    }
    return 0;
}

/* Edge::canonical, graph.rs:142-148 */
static edge_t edge_canonical(uint64_t u, uint8_t o1, uint64_t v, uint8_t o2) {
    edge_t e;
    if (u > v || (u == v && o1 == 1)) {
        e.u = v;
        e.o1 = (uint8_t)!o2;
        e.v = u;
        e.o2 = (uint8_t)!o1;
    } else {
        e.u = u;
        e.o1 = o1;
        e.v = v;
        e.o2 = o2;
    }
    return e;
}

/* ------------------------------------------------------------------------------------ */
/* PathSegment (graph.rs:472-627)                                                        */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    char *sample;
    char *haplotype; /* NULL = None */
    char *seqid;     /* NULL = None */
    int has_start, has_end;
    uint64_t start, end;
} pathseg;

static void pathseg_free(pathseg *p) {
    free(p->sample);
    free(p->haplotype);
    free(p->seqid);
    memset(p, 0, sizeof *p);
}

static int all_digits(const char *s, size_t n) {
    if (n == 0) return 0;
    for (size_t i = 0; i < n; i++)
        if (s[i] < '0' || s[i] > '9') return 0;
    return 1;
}
static int parse_usize(const char *s, size_t n, uint64_t *out) {
    /* usize::from_str(..).ok(): None on overflow */
    uint64_t v = 0;
    if (n == 0) return 0;
    for (size_t i = 0; i < n; i++) {
        if (s[i] < '0' || s[i] > '9') return 0;
        uint64_t d = (uint64_t)(s[i] - '0');
        if (v > (UINT64_MAX - d) / 10) return 0;
        v = v * 10 + d;
    }
    *out = v;
    return 1;
}

/* PATHID_COORDS = ^(.+):([0-9]+)-([0-9]+)$ (graph.rs:18).  On match returns 1 and the
 * length of group 1 plus the two numbers (has_* = 0 when usize parsing overflowed). */
static int match_coords(const char *s, size_t n, size_t *g1len, int *hs, uint64_t *st, int *he,
                        uint64_t *en) {
    /* group 3: maximal digit suffix must reach back to the last '-' */
    size_t i = n;
    while (i > 0 && s[i - 1] >= '0' && s[i - 1] <= '9') i--;
    if (i == n || i == 0 || s[i - 1] != '-') return 0;
    size_t dash = i - 1;
    size_t j = dash;
    while (j > 0 && s[j - 1] >= '0' && s[j - 1] <= '9') j--;
    if (j == dash || j == 0 || s[j - 1] != ':') return 0;
    size_t colon = j - 1;
    if (colon == 0) return 0; /* (.+) needs at least one char */
    (void)all_digits;
    *g1len = colon;
    *hs = parse_usize(s + j, dash - j, st);
    *he = parse_usize(s + dash + 1, n - dash - 1, en);
    return 1;
}

/* PathSegment::from_str, graph.rs:495-549.  PATHID_PANSN = ^([^#]+)(#[^#]+)?(#[^#].*)?$ */
static pathseg pathseg_from_str(const char *s, size_t n) {
    pathseg r;
    memset(&r, 0, sizeof r);
    r.sample = xstrndup(s, n);

    size_t a = 0;
    while (a < n && s[a] != '#') a++;
    if (a == 0) return r; /* group 1 needs >= 1 non-'#' char: no match */
    /* candidate groups */
    const char *g2 = NULL, *g3 = NULL;
    size_t g2n = 0, g3n = 0;
    int matched = 0;
    if (a == n) {
        matched = 1; /* only group 1 */
    } else {
        /* try group 2 = '#' + maximal non-'#' run (non-empty) */
        size_t b = a + 1;
        while (b < n && s[b] != '#') b++;
        if (b > a + 1) {
            if (b == n) {
                g2 = s + a;
                g2n = b - a;
                matched = 1;
            } else if (b + 1 < n && s[b + 1] != '#') {
                g2 = s + a;
                g2n = b - a;
                g3 = s + b;
                g3n = n - b;
                matched = 1;
            }
        }
        if (!matched) {
            /* backtrack: group 2 absent, group 3 = '#' [^#] .* from a */
            if (a + 1 < n && s[a + 1] != '#') {
                g3 = s + a;
                g3n = n - a;
                matched = 1;
            }
        }
    }
    if (!matched) return r;

    int nseg = 2 + (g2 != NULL) + (g3 != NULL);
    size_t g1len;
    int hs, he;
    uint64_t st, en;
    if (nseg == 4) {
        free(r.sample);
        r.sample = xstrndup(s, a);
        r.haplotype = xstrndup(g2 + 1, g2n - 1);
        if (match_coords(g3 + 1, g3n - 1, &g1len, &hs, &st, &he, &en)) {
            r.seqid = xstrndup(g3 + 1, g1len);
            r.has_start = hs;
            r.start = st;
            r.has_end = he;
            r.end = en;
        } else {
            r.seqid = xstrndup(g3 + 1, g3n - 1);
        }
    } else if (nseg == 3) {
        const char *seg = g2 ? g2 : g3;
        size_t segn = g2 ? g2n : g3n;
        free(r.sample);
        r.sample = xstrndup(s, a);
        if (match_coords(seg + 1, segn - 1, &g1len, &hs, &st, &he, &en)) {
            r.haplotype = xstrndup(seg + 1, g1len);
            r.has_start = hs;
            r.start = st;
            r.has_end = he;
            r.end = en;
        } else {
            r.haplotype = xstrndup(seg + 1, segn - 1);
        }
    } else { /* nseg == 2: segments[1] is group 1 == whole string (no '#') */
        if (match_coords(s, a, &g1len, &hs, &st, &he, &en)) {
            free(r.sample);
            r.sample = xstrndup(s, g1len);
            r.has_start = hs;
            r.start = st;
            r.has_end = he;
            r.end = en;
        }
    }
    return r;
}

/* PathSegment::id(), graph.rs:558-579 */
static char *pathseg_id(const pathseg *p) {
    char *out;
    if (p->haplotype) {
        if (p->seqid) {
            if (asprintf(&out, "%s#%s#%s", p->sample, p->haplotype, p->seqid) < 0) abort();
        } else {
            if (asprintf(&out, "%s#%s", p->sample, p->haplotype) < 0) abort();
        }
    } else if (p->seqid) {
        if (asprintf(&out, "%s#*#%s", p->sample, p->seqid) < 0) abort();
    } else {
        out = xstrndup(p->sample, strlen(p->sample));
    }
    return out;
}
/* identity of clear_coords() (graph.rs:581-589): (sample, Option<hap>, Option<seqid>) */
static char *pathseg_clearkey(const pathseg *p) {
    char *out;
    if (asprintf(&out, "%s\x01%c%s\x01%c%s", p->sample, p->haplotype ? '1' : '0',
                 p->haplotype ? p->haplotype : "", p->seqid ? '1' : '0',
                 p->seqid ? p->seqid : "") < 0)
        abort();
    return out;
}

/* ------------------------------------------------------------------------------------ */
/* GraphStorage                                                                           */
/* ------------------------------------------------------------------------------------ */
struct orc_graph {
    char *gfa_file;
    smap node2id;
    uint32_t *node_lens;
    uint64_t n_nodes;
    int has_edges;
    emap edge2id;
    uint64_t n_edges;
    pathseg *paths;
    char **path_disp;
    uint64_t n_paths;
    /* grouping state */
    char **path_group; /* group name per path */
    char **group_names;
    uint64_t n_groups;
};

uint64_t orc_graph_n_nodes(const orc_graph *g) { return g->n_nodes; }
uint64_t orc_graph_n_edges(const orc_graph *g) { return g->n_edges; }
uint64_t orc_graph_n_paths(const orc_graph *g) { return g->n_paths; }
const uint32_t *orc_graph_node_lens(const orc_graph *g) { return g->node_lens; }
const char *orc_graph_path_display(const orc_graph *g, uint64_t i) { return g->path_disp[i]; }
const char *orc_graph_group_name(const orc_graph *g, uint64_t gid) {
    return gid < g->n_groups ? g->group_names[gid] : NULL;
}

static size_t field_end(const char *s, size_t from, size_t n) {
    while (from < n && s[from] != '\t') from++;
    return from;
}

/* parse_walk_segment / parse_walk_identifier (graph.rs:381-412; util.rs:366-395) */
static int parse_walk_ident(const char *line, size_t n, pathseg *out, size_t *walk_off) {
    size_t col[7];
    size_t pos = 0;
    for (int k = 0; k < 6; k++) {
        size_t e = field_end(line, pos, n);
        if (e >= n) return 0;
        col[k] = pos;
        pos = e + 1;
    }
    col[6] = pos;
    memset(out, 0, sizeof *out);
    out->sample = xstrndup(line + col[1], col[2] - col[1] - 1);
    out->haplotype = xstrndup(line + col[2], col[3] - col[2] - 1);
    out->seqid = xstrndup(line + col[3], col[4] - col[3] - 1);
    size_t l4 = col[5] - col[4] - 1, l5 = col[6] - col[5] - 1;
    if (!(l4 == 1 && line[col[4]] == '*')) {
        if (!parse_usize(line + col[4], l4, &out->start)) return 0;
        out->has_start = 1;
    }
    if (!(l5 == 1 && line[col[5]] == '*')) {
        if (!parse_usize(line + col[5], l5, &out->end)) return 0;
        out->has_end = 1;
    }
    *walk_off = pos;
    return 1;
}

static void graph_add_path(orc_graph *g, pathseg p, size_t *cap) {
    if (g->n_paths == *cap) {
        *cap = *cap ? *cap * 2 : 64;
        g->paths = xrealloc(g->paths, *cap * sizeof *g->paths);
    }
    g->paths[g->n_paths++] = p;
}

/* GraphStorage::from_gfa (graph.rs:195-220): parse_nodes_gfa (308-375) then, for
 * Edge/All, parse_edge_gfa (276-306) in a second pass over the file. */
orc_graph *orc_graph_from_gfa(const char *gfa_file, int index_edges) {
    FILE *f = fopen(gfa_file, "rb");
    if (!f) {
        set_err("cannot open %s", gfa_file);
        return NULL;
    }
    orc_graph *g = xcalloc(1, sizeof *g);
    g->gfa_file = xstrndup(gfa_file, strlen(gfa_file));
    smap_init(&g->node2id, 1024);
    size_t lens_cap = 1024, path_cap = 0;
    g->node_lens = xmalloc(lens_cap * sizeof *g->node_lens);
    g->node_lens[0] = 0;
    uint64_t node_id = 1;

    char *line = NULL;
    size_t lcap = 0;
    ssize_t n;
    while ((n = getline(&line, &lcap, f)) > 0) {
        if (line[0] == 'S') {
            size_t e = field_end(line, 2, (size_t)n);
            if (!smap_put(&g->node2id, line + 2, e - 2, node_id)) {
                set_err("Segment with ID %.*s occurs multiple times in GFA", (int)(e - 2), line + 2);
                free(line);
                fclose(f);
                orc_graph_free(g);
                return NULL;
            }
            size_t s0 = e + 1, s1 = s0;
            while (s1 < (size_t)n && line[s1] != '\t' && line[s1] != '\n' && line[s1] != '\r') s1++;
            if (node_id >= lens_cap) {
                lens_cap *= 2;
                g->node_lens = xrealloc(g->node_lens, lens_cap * sizeof *g->node_lens);
            }
            g->node_lens[node_id] = (uint32_t)(s1 - s0);
            node_id++;
        } else if (line[0] == 'P') {
            size_t s0 = field_end(line, 0, (size_t)n) + 1;
            size_t s1 = field_end(line, s0, (size_t)n);
            graph_add_path(g, pathseg_from_str(line + s0, s1 - s0), &path_cap);
        } else if (line[0] == 'W') {
            pathseg p;
            size_t off;
            if (!parse_walk_ident(line, (size_t)n, &p, &off)) {
                set_err("malformed W line");
                free(line);
                fclose(f);
                orc_graph_free(g);
                return NULL;
            }
            graph_add_path(g, p, &path_cap);
        }
    }
    g->n_nodes = node_id - 1;
    g->path_disp = xcalloc(g->n_paths, sizeof *g->path_disp);
    for (uint64_t i = 0; i < g->n_paths; i++) {
        char *id = pathseg_id(&g->paths[i]);
        if (g->paths[i].has_start && g->paths[i].has_end) {
            if (asprintf(&g->path_disp[i], "%s:%llu-%llu", id, (unsigned long long)g->paths[i].start,
                         (unsigned long long)g->paths[i].end) < 0)
                abort();
            free(id);
        } else {
            g->path_disp[i] = id;
        }
    }

    if (index_edges) {
        g->has_edges = 1;
        emap_init(&g->edge2id, 1024);
        uint64_t edge_id = 1;
        rewind(f);
        while ((n = getline(&line, &lcap, f)) > 0) {
            if (line[0] != 'L') continue;
            /* Edge::from_link, graph.rs:100-136 */
            size_t a0 = 2, a1 = field_end(line, a0, (size_t)n);
            uint64_t u, v;
            if (!smap_get(&g->node2id, line + a0, a1 - a0, &u)) {
                set_err("unknown node %.*s", (int)(a1 - a0), line + a0);
                goto fail;
            }
            uint8_t o1 = line[a1 + 1] == '+' ? 0 : 1;
            size_t b0 = a1 + 3, b1 = field_end(line, b0, (size_t)n);
            if (!smap_get(&g->node2id, line + b0, b1 - b0, &v)) {
                set_err("unknown node %.*s", (int)(b1 - b0), line + b0);
                goto fail;
            }
            uint8_t o2 = line[b1 + 1] == '+' ? 0 : 1;
            edge_t e = edge_canonical(u, o1, v, o2);
            if (emap_put(&g->edge2id, e, edge_id)) edge_id++; /* duplicates warn + skip */
        }
        g->n_edges = edge_id - 1;
    }
    free(line);
    fclose(f);
    return g;
fail:
    free(line);
    fclose(f);
    orc_graph_free(g);
    return NULL;
}

static void graph_clear_groups(orc_graph *g) {
    if (g->path_group) {
        for (uint64_t i = 0; i < g->n_paths; i++) free(g->path_group[i]);
        free(g->path_group);
        g->path_group = NULL;
    }
    if (g->group_names) {
        for (uint64_t i = 0; i < g->n_groups; i++) free(g->group_names[i]);
        free(g->group_names);
        g->group_names = NULL;
    }
    g->n_groups = 0;
}

void orc_graph_free(orc_graph *g) {
    if (!g) return;
    graph_clear_groups(g);
    free(g->gfa_file);
    if (g->node2id.keys) smap_free(&g->node2id);
    free(g->node_lens);
    if (g->has_edges) {
        free(g->edge2id.keys);
        free(g->edge2id.vals);
    }
    for (uint64_t i = 0; i < g->n_paths; i++) {
        pathseg_free(&g->paths[i]);
        if (g->path_disp) free(g->path_disp[i]);
    }
    free(g->paths);
    free(g->path_disp);
    free(g);
}

/* ------------------------------------------------------------------------------------ */
/* GraphMask::load_groups (abacus.rs:242-308) + get_path_order (310-347) + group ids     */
/* (555-559).  Subset/exclude lists are not restated (no golden outputs in the reference) */
/* ------------------------------------------------------------------------------------ */
/* parse_bed_to_path_segments with use_block_info = true (io.rs:35-119): 1 column = a name,
 * >= 3 columns = name, start, end (from_str_start_end overrides coordinates parsed from the
 * name), exactly 12 columns = one segment per block.  Returns -1 where the reference panics. */
typedef struct {
    pathseg *v;
    size_t n, cap;
} segvec;
static void segvec_push(segvec *a, pathseg p) {
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 16;
        a->v = xrealloc(a->v, a->cap * sizeof *a->v);
    }
    a->v[a->n++] = p;
}
static void segvec_free(segvec *a) {
    for (size_t i = 0; i < a->n; i++) pathseg_free(&a->v[i]);
    free(a->v);
    memset(a, 0, sizeof *a);
}
/* usize::from_str: digits with an optional leading '+' */
static int parse_usize_rust(const char *s, size_t n, uint64_t *out) {
    if (n > 0 && s[0] == '+') {
        s++;
        n--;
    }
    return parse_usize(s, n, out);
}
/* "a,b,c,".split(',').filter_map(|s| usize::from_str(s.trim()).ok()) */
static size_t parse_usize_list(const char *s, size_t n, uint64_t **out) {
    size_t cnt = 0, cap = 0, i = 0;
    *out = NULL;
    for (;;) {
        size_t e = i;
        while (e < n && s[e] != ',') e++;
        size_t b = i, t = e;
        while (b < t && isspace((unsigned char)s[b])) b++;
        while (t > b && isspace((unsigned char)s[t - 1])) t--;
        uint64_t v;
        if (parse_usize_rust(s + b, t - b, &v)) {
            if (cnt == cap) {
                cap = cap ? cap * 2 : 8;
                *out = xrealloc(*out, cap * sizeof **out);
            }
            (*out)[cnt++] = v;
        }
        if (e >= n) break;
        i = e + 1;
    }
    return cnt;
}
static pathseg pathseg_with_coords(const char *name, size_t n, uint64_t st, uint64_t en) {
    pathseg ps = pathseg_from_str(name, n);
    ps.has_start = ps.has_end = 1;
    ps.start = st;
    ps.end = en;
    return ps;
}
static int parse_bed(const char *file, segvec *out) {
    FILE *f = fopen(file, "rb");
    if (!f) {
        set_err("cannot open list file %s", file);
        return -1;
    }
    char *line = NULL;
    size_t lcap = 0;
    ssize_t n;
    int lineno = 0, rc = 0;
    while ((n = getline(&line, &lcap, f)) > 0) {
        lineno++;
        /* BufRead::lines strips \n and \r\n */
        if (n > 0 && line[n - 1] == '\n') {
            n--;
            if (n > 0 && line[n - 1] == '\r') n--;
        }
        size_t fb[13], fe[13], nf = 0, pos = 0, total = 0;
        for (;;) {
            size_t e = field_end(line, pos, (size_t)n);
            if (nf < 13) {
                fb[nf] = pos;
                fe[nf] = e;
                nf++;
            }
            total++;
            if (e >= (size_t)n) break;
            pos = e + 1;
        }
        const char *name = line + fb[0];
        size_t nn = fe[0] - fb[0];
        if ((nn >= 8 && memcmp(name, "browser ", 8) == 0) || (nn >= 6 && memcmp(name, "track ", 6) == 0) ||
            (nn >= 1 && name[0] == '#'))
            continue;
        if (total == 1) {
            segvec_push(out, pathseg_from_str(name, nn));
        } else if (total >= 3) {
            uint64_t st, en;
            if (!parse_usize_rust(line + fb[1], fe[1] - fb[1], &st) || !parse_usize_rust(line + fb[2], fe[2] - fb[2], &en)) {
                set_err("error line %d: start/end is not an usize", lineno);
                rc = -1;
                break;
            }
            if (total == 12) {
                uint64_t bc = 0, *sizes, *starts;
                if (!parse_usize_rust(line + fb[9], fe[9] - fb[9], &bc)) bc = 0; /* unwrap_or(0) */
                size_t ns = parse_usize_list(line + fb[10], fe[10] - fb[10], &sizes);
                size_t nt = parse_usize_list(line + fb[11], fe[11] - fb[11], &starts);
                if (bc != ns || bc != nt) {
                    set_err("error in block sizes/starts in line %d: counts do not match", lineno);
                    free(sizes);
                    free(starts);
                    rc = -1;
                    break;
                }
                for (size_t k = 0; k < ns; k++)
                    segvec_push(out, pathseg_with_coords(name, nn, st + starts[k], st + starts[k] + sizes[k]));
                free(sizes);
                free(starts);
            } else {
                segvec_push(out, pathseg_with_coords(name, nn, st, en));
            }
        } else {
            set_err("error in line %d: row must have either 1, 3, or 12 columns, but has 2", lineno);
            rc = -1;
            break;
        }
    }
    free(line);
    fclose(f);
    return rc;
}

/* GraphMask::load_coord_list (abacus.rs:212-240): the text is a BED file if such a file exists,
 * otherwise a regular expression over the displayed path names (`re.is_match(&path.to_string())`,
 * an unanchored search); the matching graph paths, with their own coordinates, are the list.
 * The reference uses the Rust `regex` crate; POSIX extended syntax is used here, which agrees with
 * it on literals, anchors, classes, alternation and greedy repetition. */
static int load_coord_list(const orc_graph *g, const char *text, segvec *out) {
    struct stat st;
    if (stat(text, &st) == 0 && S_ISREG(st.st_mode)) return parse_bed(text, out);
    regex_t re;
    if (regcomp(&re, text, REG_EXTENDED | REG_NOSUB) != 0) {
        set_err("string %s is not valid! Neither as a file name nor as a regex", text);
        return -1;
    }
    for (uint64_t i = 0; i < g->n_paths; i++)
        if (regexec(&re, g->path_disp[i], 0, NULL, 0) == 0) {
            const pathseg *q = &g->paths[i]; /* .cloned() */
            pathseg c = *q;
            c.sample = xstrndup(q->sample, strlen(q->sample));
            c.haplotype = q->haplotype ? xstrndup(q->haplotype, strlen(q->haplotype)) : NULL;
            c.seqid = q->seqid ? xstrndup(q->seqid, strlen(q->seqid)) : NULL;
            segvec_push(out, c);
        }
    regfree(&re);
    return 0;
}

/* reads a path/group list (BED, see parse_bed) and marks the paths it names:
 * complement_with_group_assignments (abacus.rs:152-206) -- a name that is a path selects that
 * path, a name that is a group selects all paths of the group (and must not carry coordinates).
 * When `exact_coords` is set a path entry only matches a graph path whose coordinates are
 * equal too (HashSet<&PathSegment> in abacus.rs:329-336).  visit (optional) receives the
 * entries in file order as path indices. */
static int read_path_list(const orc_graph *g, const char *file, const smap *key2path, char **keys,
                          int exact_coords, uint8_t *mark, uint64_t **visit, uint64_t *nvisit) {
    segvec segs = {0};
    if (load_coord_list(g, file, &segs) != 0) {
        segvec_free(&segs);
        return -1;
    }
    size_t vcap = 0;
    uint64_t P = g->n_paths;
    (void)keys;
    for (size_t si = 0; si < segs.n; si++) {
        const pathseg ps = segs.v[si];
        char *k = pathseg_clearkey(&ps);
        uint64_t pi;
        if (smap_get(key2path, k, strlen(k), &pi)) {
            /* all graph paths sharing this clear-coords identity */
            for (uint64_t i = 0; i < P; i++) {
                char *ki = pathseg_clearkey(&g->paths[i]);
                int same = strcmp(ki, k) == 0;
                free(ki);
                if (!same) continue;
                if (exact_coords && (g->paths[i].has_start != ps.has_start || g->paths[i].has_end != ps.has_end ||
                                     (ps.has_start && g->paths[i].start != ps.start) ||
                                     (ps.has_end && g->paths[i].end != ps.end)))
                    continue;
                mark[i] = 1;
            }
            if (visit) {
                if (*nvisit == vcap) {
                    vcap = vcap ? vcap * 2 : 64;
                    *visit = xrealloc(*visit, vcap * sizeof **visit);
                }
                (*visit)[(*nvisit)++] = pi;
            }
        } else {
            char *id = pathseg_id(&ps);
            int hit = 0, is_group = 0;
            for (uint64_t i = 0; i < P && !is_group; i++)
                is_group = g->path_group[i] && strcmp(g->path_group[i], id) == 0;
            if (is_group && ps.has_start && ps.has_end) {
                set_err("invalid coordinate \"%s\": group identifiers are not allowed to have start/stop information!", id);
                free(id);
                free(k);
                segvec_free(&segs);
                return -1;
            }
            for (uint64_t i = 0; i < P; i++)
                if (g->path_group[i] && strcmp(g->path_group[i], id) == 0) {
                    /* group members come from the keys of `groups`, i.e. WITHOUT coordinates */
                    if (exact_coords && (g->paths[i].has_start || g->paths[i].has_end)) continue;
                    mark[i] = 1;
                    if (visit && !hit) {
                        if (*nvisit == vcap) {
                            vcap = vcap ? vcap * 2 : 64;
                            *visit = xrealloc(*visit, vcap * sizeof **visit);
                        }
                        (*visit)[(*nvisit)++] = i;
                    }
                    hit = 1;
                }
            free(id);
        }
        free(k);
    }
    segvec_free(&segs);
    return 0;
}

int64_t orc_graph_path_order(orc_graph *g, int group_mode, const char *group_file,
                             const char *order_file, uint64_t *path_idx, uint64_t *group_id,
                             uint64_t *n_out) {
    return orc_graph_path_order_masked(g, group_mode, group_file, order_file, NULL, NULL, path_idx, group_id, n_out);
}

/* With subset / exclude path lists (whole paths only):
 *   order source (abacus.rs:324-337): -O list, else the subset list, else all paths that are
 *   not in the exclude list; every entry emits the whole bucket of its group;
 *   paths outside the subset have an EMPTY item-table entry in the reference
 *   (util.rs:88-105: skipped during the parse) -- here they are dropped from the order, which
 *   yields the same countables and the same groups. */
int64_t orc_graph_path_order_masked(orc_graph *g, int group_mode, const char *group_file,
                                    const char *order_file, const char *subset_file, const char *exclude_file,
                                    uint64_t *path_idx, uint64_t *group_id, uint64_t *n_out) {
    graph_clear_groups(g);
    uint64_t P = g->n_paths;
    g->path_group = xcalloc(P, sizeof *g->path_group);

    /* clear_coords key -> first path index with that key */
    smap key2path;
    smap_init(&key2path, P + 1);
    char **keys = xcalloc(P, sizeof *keys);
    for (uint64_t i = 0; i < P; i++) {
        keys[i] = pathseg_clearkey(&g->paths[i]);
        smap_put(&key2path, keys[i], strlen(keys[i]), i);
    }

    int64_t ret = -1;
    smap file_groups; /* clearkey -> index into fg_names */
    char **fg_names = NULL;
    size_t fg_n = 0, fg_cap = 0;
    int have_fg = 0;

    if (group_mode == ORC_GROUP_HAPLOTYPE) {
        for (uint64_t i = 0; i < P; i++)
            if (asprintf(&g->path_group[i], "%s#%s", g->paths[i].sample,
                         g->paths[i].haplotype ? g->paths[i].haplotype : "") < 0)
                abort();
    } else if (group_mode == ORC_GROUP_SAMPLE) {
        for (uint64_t i = 0; i < P; i++)
            g->path_group[i] = xstrndup(g->paths[i].sample, strlen(g->paths[i].sample));
    } else if (group_mode == ORC_GROUP_FILE) {
        FILE *f = fopen(group_file, "rb");
        if (!f) {
            set_err("cannot open group file %s", group_file);
            goto done;
        }
        smap_init(&file_groups, 64);
        have_fg = 1;
        char *line = NULL;
        size_t lcap = 0;
        ssize_t n;
        int lineno = 1;
        while ((n = getline(&line, &lcap, f)) > 0) {
            /* parse_groups, io.rs:121-151: strip ONE trailing \n or \r */
            if (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) n--;
            size_t tabs = 0, t0 = 0;
            for (ssize_t k = 0; k < n; k++)
                if (line[k] == '\t') {
                    if (!tabs) t0 = (size_t)k;
                    tabs++;
                }
            if (tabs != 1) {
                set_err("error in line %d: table must have exactly two columns", lineno);
                free(line);
                fclose(f);
                goto done;
            }
            pathseg ps = pathseg_from_str(line, t0);
            char *k = pathseg_clearkey(&ps);
            uint64_t idx;
            const char *grp = line + t0 + 1;
            size_t grpn = (size_t)n - t0 - 1;
            if (smap_get(&file_groups, k, strlen(k), &idx)) {
                if (strlen(fg_names[idx]) != grpn || memcmp(fg_names[idx], grp, grpn) != 0) {
                    set_err("path cannot be assigned to more than one group (line %d)", lineno);
                    free(k);
                    pathseg_free(&ps);
                    free(line);
                    fclose(f);
                    goto done;
                }
            } else {
                if (fg_n == fg_cap) {
                    fg_cap = fg_cap ? fg_cap * 2 : 64;
                    fg_names = xrealloc(fg_names, fg_cap * sizeof *fg_names);
                }
                fg_names[fg_n] = xstrndup(grp, grpn);
                smap_put(&file_groups, k, strlen(k), fg_n);
                fg_n++;
            }
            free(k);
            pathseg_free(&ps);
            lineno++;
        }
        free(line);
        fclose(f);
        for (uint64_t i = 0; i < P; i++) {
            uint64_t idx;
            if (smap_get(&file_groups, keys[i], strlen(keys[i]), &idx))
                g->path_group[i] = xstrndup(fg_names[idx], strlen(fg_names[idx]));
            else
                g->path_group[i] = pathseg_id(&g->paths[i]); /* abacus.rs:290-294 */
        }
    } else {
        for (uint64_t i = 0; i < P; i++) g->path_group[i] = pathseg_id(&g->paths[i]);
    }
    /* `groups` is a HashMap keyed by clear_coords(): paths sharing a key share the entry.
     * With -S/-H/default the value is a pure function of the key, so nothing to reconcile. */

    /* get_path_order: bucket path indices by group name (file order inside a bucket) */
    {
        smap grp2bucket;
        smap_init(&grp2bucket, P + 1);
        uint64_t nb = 0;
        uint64_t *bucket_of = xcalloc(P, sizeof *bucket_of);
        for (uint64_t i = 0; i < P; i++) {
            uint64_t b;
            const char *gn = g->path_group[i];
            if (!smap_get(&grp2bucket, gn, strlen(gn), &b)) {
                b = nb++;
                smap_put(&grp2bucket, gn, strlen(gn), b);
            }
            bucket_of[i] = b;
        }
        uint8_t *bucket_done = xcalloc(nb ? nb : 1, 1);

        /* the order source: a list of group names to visit */
        uint64_t *visit = NULL; /* bucket ids in visiting order */
        uint64_t nvisit = 0, vcap = 0;
#define PUSH_VISIT(b)                                          \
    do {                                                       \
        if (nvisit == vcap) {                                  \
            vcap = vcap ? vcap * 2 : 64;                       \
            visit = xrealloc(visit, vcap * sizeof *visit);     \
        }                                                      \
        visit[nvisit++] = (b);                                 \
    } while (0)
        if (order_file) {
            FILE *f = fopen(order_file, "rb");
            if (!f) {
                set_err("cannot open order file %s", order_file);
                free(bucket_of);
                free(bucket_done);
                smap_free(&grp2bucket);
                goto done;
            }
            char *line = NULL;
            size_t lcap = 0;
            ssize_t n;
            while ((n = getline(&line, &lcap, f)) > 0) {
                /* BufRead::lines strips \n and \r\n (io.rs:42) */
                if (n > 0 && line[n - 1] == '\n') n--;
                if (n > 0 && line[n - 1] == '\r') n--;
                size_t e = field_end(line, 0, (size_t)n);
                if ((e >= 8 && memcmp(line, "browser ", 8) == 0) ||
                    (e >= 6 && memcmp(line, "track ", 6) == 0) || (e >= 1 && line[0] == '#'))
                    continue;
                pathseg ps = pathseg_from_str(line, e);
                char *k = pathseg_clearkey(&ps);
                uint64_t pi, b;
                /* complement_with_group_assignments, abacus.rs:152-206 */
                if (smap_get(&key2path, k, strlen(k), &pi)) {
                    PUSH_VISIT(bucket_of[pi]);
                } else {
                    char *id = pathseg_id(&ps);
                    if (smap_get(&grp2bucket, id, strlen(id), &b)) PUSH_VISIT(b);
                    /* else: unknown path/group -> logged and skipped */
                    free(id);
                }
                free(k);
                pathseg_free(&ps);
            }
            free(line);
            fclose(f);
        } else if (subset_file) {
            uint64_t *sv = NULL, nsv = 0;
            uint8_t *tmpmark = xcalloc(P ? P : 1, 1);
            if (read_path_list(g, subset_file, &key2path, keys, 0, tmpmark, &sv, &nsv) != 0) {
                free(tmpmark);
                free(sv);
                free(bucket_of);
                free(bucket_done);
                free(visit);
                smap_free(&grp2bucket);
                goto done;
            }
            for (uint64_t k = 0; k < nsv; k++) PUSH_VISIT(bucket_of[sv[k]]);
            free(tmpmark);
            free(sv);
        } else {
            uint8_t *ex = xcalloc(P ? P : 1, 1);
            if (exclude_file && read_path_list(g, exclude_file, &key2path, keys, 1, ex, NULL, NULL) != 0) {
                free(ex);
                free(bucket_of);
                free(bucket_done);
                free(visit);
                smap_free(&grp2bucket);
                goto done;
            }
            for (uint64_t i = 0; i < P; i++)
                if (!ex[i]) PUSH_VISIT(bucket_of[i]);
            free(ex);
        }
#undef PUSH_VISIT
        uint8_t *in_subset = NULL;
        if (subset_file) {
            in_subset = xcalloc(P ? P : 1, 1);
            if (read_path_list(g, subset_file, &key2path, keys, 0, in_subset, NULL, NULL) != 0) {
                free(in_subset);
                free(bucket_of);
                free(bucket_done);
                free(visit);
                smap_free(&grp2bucket);
                goto done;
            }
        }

        uint64_t out = 0, ngroups = 0;
        size_t gcap = 0;
        for (uint64_t vi = 0; vi < nvisit; vi++) {
            uint64_t b = visit[vi];
            if (bucket_done[b]) continue; /* group_to_paths.remove(..).unwrap_or_default() */
            bucket_done[b] = 1;
            for (uint64_t i = 0; i < P; i++) {
                if (bucket_of[i] != b) continue;
                const char *gn = g->path_group[i];
                /* abacus.rs:555-559: new group id whenever the emitted name changes */
                if (ngroups == 0 || strcmp(g->group_names[ngroups - 1], gn) != 0) {
                    if (ngroups == gcap) {
                        gcap = gcap ? gcap * 2 : 64;
                        g->group_names = xrealloc(g->group_names, gcap * sizeof *g->group_names);
                    }
                    g->group_names[ngroups++] = xstrndup(gn, strlen(gn));
                }
                if (in_subset && !in_subset[i]) continue; /* empty item-table entry in the reference */
                path_idx[out] = i;
                group_id[out] = ngroups - 1;
                out++;
            }
        }
        free(in_subset);
        g->n_groups = ngroups;
        *n_out = out;
        ret = (int64_t)ngroups;
        free(visit);
        free(bucket_of);
        free(bucket_done);
        smap_free(&grp2bucket);
    }
done:
    if (have_fg) {
        smap_free(&file_groups);
        for (size_t i = 0; i < fg_n; i++) free(fg_names[i]);
        free(fg_names);
    }
    for (uint64_t i = 0; i < P; i++) free(keys[i]);
    free(keys);
    smap_free(&key2path);
    return ret;
}

/* ------------------------------------------------------------------------------------ */
/* ItemTable: parse_gfa_paths_walks_multiple without subset/exclude (util.rs:22-206);     */
/* node steps via get_path_segment_ids / get_walk_segment_ids (util.rs:1048-1142), edge   */
/* steps via update_tables_edgecount (util.rs:723-795) with include=[(0,MAX)], exclude=[] */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    uint64_t *v;
    size_t n, cap;
} u64vec;
static void u64vec_push(u64vec *a, uint64_t x) {
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 1024;
        a->v = xrealloc(a->v, a->cap * sizeof *a->v);
    }
    a->v[a->n++] = x;
}

/* the step column of a P line / the walk of a W line as (segment id, orientation) pairs:
 * parse_path_seq_to_item_vec / parse_walk_seq_to_item_vec (util.rs:797-850, 963-1046) */
static int parse_steps(const orc_graph *g, const char *line, size_t n, u64vec *sids, u64vec *oris) {
    sids->n = 0;
    oris->n = 0;
    size_t pos, end;
    if (line[0] == 'P') {
        size_t s0 = field_end(line, 0, n) + 1;
        pos = field_end(line, s0, n) + 1;
        end = pos;
        while (end < n && line[end] != '\t' && line[end] != '\n' && line[end] != '\r') end++;
        /* steps "name+,name-": get_segment_id util.rs:1017-1031 */
        while (pos < end) {
            size_t e = pos;
            while (e < end && line[e] != ',') e++;
            if (e > pos) {
                char o = line[e - 1];
                uint64_t id;
                if ((o != '+' && o != '-') || !smap_get(&g->node2id, line + pos, e - 1 - pos, &id)) {
                    set_err("unknown node %.*s", (int)(e - pos), line + pos);
                    return -1;
                }
                u64vec_push(sids, id);
                u64vec_push(oris, o == '+' ? 0 : 1);
            }
            pos = e + 1;
        }
    } else {
        pathseg ps;
        if (!parse_walk_ident(line, n, &ps, &pos)) {
            set_err("malformed W line");
            return -1;
        }
        pathseg_free(&ps);
        end = pos;
        while (end < n && line[end] != '\t' && line[end] != '\n' && line[end] != '\r') end++;
        /* walk ">name<name": get_walk_segment_id util.rs:1033-1046 */
        while (pos < end) {
            size_t e = pos + 1;
            while (e < end && line[e] != '>' && line[e] != '<') e++;
            char o = line[pos];
            uint64_t id;
            if ((o != '>' && o != '<') || !smap_get(&g->node2id, line + pos + 1, e - pos - 1, &id)) {
                set_err("unknown node %.*s", (int)(e - pos), line + pos);
                return -1;
            }
            u64vec_push(sids, id);
            u64vec_push(oris, o == '>' ? 0 : 1);
            pos = e;
        }
    }
    return 0;
}

int64_t orc_graph_item_table(const orc_graph *g, int count_type, uint64_t **items_out,
                             uint64_t *prefsum) {
    if (count_type == ORC_EDGE && !g->has_edges) {
        set_err("graph was loaded without edge index");
        return -1;
    }
    FILE *f = fopen(g->gfa_file, "rb");
    if (!f) {
        set_err("cannot open %s", g->gfa_file);
        return -1;
    }
    u64vec items = {0};
    u64vec sids = {0};
    u64vec oris = {0};
    uint64_t num_path = 0;
    prefsum[0] = 0;
    char *line = NULL;
    size_t lcap = 0;
    ssize_t n;
    while ((n = getline(&line, &lcap, f)) > 0) {
        if (line[0] != 'P' && line[0] != 'W') continue;
        if (parse_steps(g, line, (size_t)n, &sids, &oris) != 0) goto fail;
        uint64_t added = 0;
        if (count_type == ORC_EDGE) {
            /* util.rs:723-795 with include=[(0,usize::MAX)], exclude=[], offset = path start */
            if (sids.n > 0) {
                uint64_t p = g->paths[num_path].has_start && g->paths[num_path].has_end
                                 ? g->paths[num_path].start
                                 : 0;
                p += g->node_lens[sids.v[0]];
                for (size_t k = 0; k + 1 < sids.n; k++) {
                    uint64_t l = g->node_lens[sids.v[k + 1]];
                    edge_t e = edge_canonical(sids.v[k], (uint8_t)oris.v[k], sids.v[k + 1],
                                              (uint8_t)oris.v[k + 1]);
                    uint64_t eid = emap_get(&g->edge2id, e);
                    if (!eid) {
                        set_err("unknown edge %c%llu%c%llu", e.o1 ? '<' : '>', (unsigned long long)e.u,
                                e.o2 ? '<' : '>', (unsigned long long)e.v);
                        goto fail;
                    }
                    if (0 < p + l) { /* include_coords[0].0 < p + l */
                        u64vec_push(&items, eid);
                        added++;
                    }
                    p += l;
                }
            }
        } else {
            for (size_t k = 0; k < sids.n; k++) u64vec_push(&items, sids.v[k]);
            added = sids.n;
        }
        prefsum[num_path + 1] = prefsum[num_path] + added;
        num_path++;
    }
    free(line);
    free(sids.v);
    free(oris.v);
    fclose(f);
    *items_out = items.v ? items.v : xmalloc(8);
    return (int64_t)items.n;
fail:
    free(line);
    free(sids.v);
    free(oris.v);
    free(items.v);
    fclose(f);
    return -1;
}

/* ------------------------------------------------------------------------------------ */
/* AbacusByTotal::item_table_to_abacus + coverage (abacus.rs:539-586, 719-744)           */
/* ------------------------------------------------------------------------------------ */
void orc_coverage(const uint64_t *items, const uint64_t *prefsum, const uint64_t *path_idx,
                  const uint64_t *group_id, uint64_t n_ordered, uint64_t n_items,
                  const uint8_t *exclude, uint32_t *countable) {
    uint64_t *last = xmalloc((n_items + 1) * sizeof *last);
    for (uint64_t i = 0; i <= n_items; i++) {
        countable[i] = 0;
        last[i] = UINT64_MAX;
    }
    countable[0] = UINT32_MAX;
    for (uint64_t k = 0; k < n_ordered; k++) {
        uint64_t start = prefsum[path_idx[k]], end = prefsum[path_idx[k] + 1];
        uint64_t grp = group_id[k];
        for (uint64_t j = start; j < end; j++) {
            uint64_t sid = items[j];
            if (last[sid] != grp && (!exclude || !exclude[sid])) {
                countable[sid] += 1;
                last[sid] = grp;
            }
        }
    }
    free(last);
}

/* construct_hist / construct_hist_bps (abacus.rs:746-787) */
void orc_hist(const uint32_t *countable, uint64_t n_items, uint64_t n_groups,
              const uint32_t *weights, uint64_t *hist) {
    for (uint64_t i = 0; i <= n_groups; i++) hist[i] = 0;
    for (uint64_t i = 0; i <= n_items; i++) {
        uint64_t cov = countable[i];
        if (cov >= n_groups + 1) continue; /* index 0 holds u32::MAX and is skipped here */
        hist[cov] += weights ? weights[i] : 1;
    }
}

/* ------------------------------------------------------------------------------------ */
/* closed-form growth (graph_broker/hist.rs:21-187)                                      */
/* ------------------------------------------------------------------------------------ */
/* Threshold::to_absolute / to_relative, util.rs:350-363 */
static uint64_t thr_to_absolute(int kind, double val, uint64_t n) {
    if (kind == ORC_THR_ABSOLUTE) return (uint64_t)val;
    return (uint64_t)ceil((double)n * val);
}
static double thr_to_relative(int kind, double val, uint64_t n) {
    if (kind == ORC_THR_RELATIVE) return val;
    return (double)(uint64_t)val / (double)n;
}

/* hist.rs:21-36 */
double orc_choose(uint64_t n, uint64_t k) {
    double res = 0.0;
    if (k > n) return 0.0;
    if (k > n - k) k = n - k;
    double nf = (double)n;
    for (uint64_t i = 0; i < k; i++) {
        res += log2(nf - (double)i);
        res -= log2((double)i + 1.0);
    }
    return res;
}

/* hist.rs:89-114 */
void orc_growth_union(const uint64_t *cov, uint64_t n, int ck, double cv, double *out) {
    uint64_t c = thr_to_absolute(ck, cv, n);
    if (c < 1) c = 1;
    double n_fall_m = 0.0;
    uint64_t tot_i = 0;
    for (uint64_t i = c; i <= n; i++) tot_i += cov[i];
    double tot = (double)tot_i;
    double *perc_mult = xcalloc(n + 1, sizeof *perc_mult);
    for (uint64_t m = 1; m < n + 1; m++) {
        double y = 0.0;
        n_fall_m += log2((double)n - (double)m + 1.0);
        for (uint64_t i = c; i < n - m + 1; i++) {
            perc_mult[i] += log2((double)n - (double)m - (double)i + 1.0);
            y += exp2(log2((double)cov[i]) + perc_mult[i] - n_fall_m);
        }
        out[m - 1] = tot - y;
    }
    free(perc_mult);
}

/* hist.rs:116-138 */
void orc_growth_core(const uint64_t *cov, uint64_t n, int ck, double cv, double *out) {
    uint64_t c = thr_to_absolute(ck, cv, n + 1);
    if (c < 1) c = 1;
    double n_fall_m = 0.0;
    double *perc_mult = xcalloc(n + 1, sizeof *perc_mult);
    for (uint64_t m = 1; m < n + 1; m++) {
        double y = 0.0;
        n_fall_m += log2((double)n - (double)m + 1.0);
        for (uint64_t i = (m > c ? m : c); i < n + 1; i++) {
            perc_mult[i] += log2((double)i - (double)m + 1.0);
            y += exp2(log2((double)cov[i]) + perc_mult[i] - n_fall_m);
        }
        out[m - 1] = y;
    }
    free(perc_mult);
}

/* hist.rs:140-187 */
void orc_growth_quorum(const uint64_t *cov, uint64_t n, int ck, double cv, int qk, double qv,
                          double *out) {
    uint64_t c = thr_to_absolute(ck, cv, n);
    if (c < 1) c = 1;
    double quorum = thr_to_relative(qk, qv, n);
    double n_fall_m = 0.0, m_fact = 0.0;
    double *perc_mult = xcalloc(n + 1, sizeof *perc_mult);
    double *q = xcalloc((n + 1) * (n + 1), sizeof *q);
    for (uint64_t m = 1; m < n + 1; m++) {
        m_fact += log2((double)m);
        uint64_t m_quorum = (uint64_t)ceil((double)m * quorum);
        double yl = 0.0;
        n_fall_m += log2((double)n - (double)m + 1.0);
        for (uint64_t i = (m > c ? m : c); i < n + 1; i++) {
            perc_mult[i] += log2((double)i - (double)m + 1.0);
            yl += exp2(log2((double)cov[i]) + perc_mult[i] - n_fall_m);
        }
        double yr = 0.0;
        for (uint64_t i = m_quorum; i < n; i++) {
            double sum_q = 0.0;
            int add = 0;
            for (uint64_t j = (m_quorum > c ? m_quorum : c); j < m; j++) {
                if (n + j + 1 > i + m && j <= i) {
                    double *qij = &q[i * (n + 1) + j];
                    if (*qij == 0.0) *qij = orc_choose(i, j);
                    *qij += log2((double)n - (double)i - (double)m + 1.0 + (double)j);
                    *qij -= log2((double)m - (double)j);
                    sum_q += exp2(*qij + m_fact - n_fall_m);
                    add = 1;
                }
            }
            if (add) yr += exp2(log2((double)cov[i]) + log2(sum_q));
        }
        out[m - 1] = yl + yr;
    }
    free(perc_mult);
    free(q);
}

/* Hist::calc_growth dispatch, hist.rs:51-66 */
int64_t orc_growth(const uint64_t *hist, uint64_t hist_len, int cov_kind, double cov_val,
                   int quo_kind, double quo_val, double *out) {
    if (hist_len == 0) return 0;
    uint64_t n = hist_len - 1;
    if (n == 0) return 0;
    uint64_t quorum = thr_to_absolute(quo_kind, quo_val, n);
    if (quorum < 1) quorum = 1;
    if (quorum == 1)
        orc_growth_union(hist, n, cov_kind, cov_val, out);
    else if (quorum >= n)
        orc_growth_core(hist, n, cov_kind, cov_val, out);
    else
        orc_growth_quorum(hist, n, cov_kind, cov_val, quo_kind, quo_val, out);
    return (int64_t)n;
}

/* ------------------------------------------------------------------------------------ */
/* AbacusByGroup: compute_row_storage_space (abacus.rs:859-899) +                        */
/* compute_column_values (901-986, report_values=false) restated with the same           */
/* "last slot is the write cursor" trick so that any quirk of it is reproduced.           */
/* ------------------------------------------------------------------------------------ */
static int64_t by_group_impl(const uint64_t *items, const uint64_t *prefsum, const uint64_t *path_idx,
                             const uint64_t *group_id, uint64_t n_ordered, uint64_t n_items,
                             const uint8_t *exclude, uint64_t *r, uint64_t **c_out, uint32_t **v_out) {
    uint64_t *last = xmalloc((n_items + 1) * sizeof *last);
    for (uint64_t i = 0; i <= n_items; i++) last[i] = UINT64_MAX;
    for (uint64_t i = 0; i < n_items + 2; i++) r[i] = 0;
    for (uint64_t k = 0; k < n_ordered; k++) {
        uint64_t start = prefsum[path_idx[k]], end = prefsum[path_idx[k] + 1];
        for (uint64_t j = start; j < end; j++) {
            uint64_t sid = items[j];
            if (last[sid] != group_id[k] && (!exclude || !exclude[sid])) {
                r[sid] += 1;
                last[sid] = group_id[k];
            }
        }
    }
    free(last);
    uint64_t acc = 0;
    for (uint64_t i = 0; i < n_items + 2; i++) {
        uint64_t tmp = r[i];
        r[i] = acc;
        acc += tmp;
    }
    uint64_t nnz = r[n_items + 1];
    uint64_t *c = xmalloc((nnz ? nnz : 1) * sizeof *c);
    for (uint64_t i = 0; i < nnz; i++) c[i] = UINT64_MAX;
    /* report_values (abacus.rs:907-916): one counter per slot */
    uint32_t *v = v_out ? xcalloc(nnz ? nnz : 1, sizeof *v) : NULL;
    for (uint64_t k = 0; k < n_ordered; k++) {
        uint64_t start = prefsum[path_idx[k]], end = prefsum[path_idx[k] + 1];
        uint64_t grp = group_id[k];
        for (uint64_t j = start; j < end; j++) {
            uint64_t sid = items[j];
            uint64_t cv_start = r[sid], cv_end = r[sid + 1];
            if (cv_end == cv_start) continue;
            uint64_t p = c[cv_end - 1];
            if (c[cv_end - 1] == UINT64_MAX) {
                c[cv_start] = grp;
                if (cv_start < cv_end - 1) c[cv_end - 1] = 0;
                if (v) v[cv_start] += 1;
            } else if (cv_start + p < cv_end - 1) {
                if (c[cv_start + p] < grp) {
                    c[cv_end - 1] += 1;
                    p += 1;
                    c[cv_start + p] = grp;
                }
                if (v) v[cv_start + p] += 1;
            } else if (v) {
                v[cv_end - 1] += 1; /* "make sure it points to the last element and not beyond" */
            }
        }
    }
    *c_out = c;
    if (v_out) *v_out = v;
    return (int64_t)nnz;
}

int64_t orc_by_group(const uint64_t *items, const uint64_t *prefsum, const uint64_t *path_idx,
                     const uint64_t *group_id, uint64_t n_ordered, uint64_t n_items,
                     const uint8_t *exclude, uint64_t *r, uint64_t **c_out) {
    return by_group_impl(items, prefsum, path_idx, group_id, n_ordered, n_items, exclude, r, c_out, NULL);
}

int64_t orc_by_group_values(const uint64_t *items, const uint64_t *prefsum, const uint64_t *path_idx,
                            const uint64_t *group_id, uint64_t n_ordered, uint64_t n_items,
                            const uint8_t *exclude, uint64_t *r, uint64_t **c_out, uint32_t **v_out) {
    return by_group_impl(items, prefsum, path_idx, group_id, n_ordered, n_items, exclude, r, c_out, v_out);
}


/* AbacusByGroup::calc_growth, abacus.rs:989-1032 */
void orc_ordered_growth(const uint64_t *r, const uint64_t *c, uint64_t n_items,
                        uint64_t n_groups, int cov_kind, double cov_val, int quo_kind,
                        double quo_val, const uint32_t *weights, double *out) {
    for (uint64_t j = 0; j < n_groups; j++) out[j] = 0.0;
    uint64_t cthr = thr_to_absolute(cov_kind, cov_val, n_groups);
    if (cthr < 1) cthr = 1;
    double q = thr_to_relative(quo_kind, quo_val, n_groups);
    if (!(q > 0.0)) q = 0.0; /* f64::max(0.0, q) */
    for (uint64_t i = 1; i <= n_items; i++) {
        uint64_t start = r[i], end = r[i + 1];
        if (end - start >= cthr) {
            uint64_t k = start;
            for (uint64_t j = c[start]; j < n_groups; j++) {
                if (k < end - 1 && c[k + 1] <= j) k += 1;
                if (k - start + 1 >= (uint64_t)ceil(((double)c[k] + 1.0) * q))
                    out[j] += weights ? (double)weights[i] : 1.0;
            }
        }
    }
}

/* ActiveTable of an exclude list of whole paths (abacus.rs:447-458; util.rs:1171-1181,
 * 785-787): every item on an excluded path is flagged.  Needs orc_graph_path_order* to have
 * run (group names).  flags has n_items+1 entries. */
int orc_graph_exclude_flags(orc_graph *g, int count_type, const char *exclude_file, uint8_t *flags) {
    uint64_t P = g->n_paths;
    uint64_t n_items = count_type == ORC_EDGE ? g->n_edges : g->n_nodes;
    memset(flags, 0, n_items + 1);
    smap key2path;
    smap_init(&key2path, P + 1);
    for (uint64_t i = 0; i < P; i++) {
        char *k = pathseg_clearkey(&g->paths[i]);
        smap_put(&key2path, k, strlen(k), i);
        free(k);
    }
    uint8_t *ex = xcalloc(P ? P : 1, 1);
    int rc = read_path_list(g, exclude_file, &key2path, NULL, 0, ex, NULL, NULL);
    smap_free(&key2path);
    if (rc != 0) {
        free(ex);
        return -1;
    }
    uint64_t *items = NULL;
    uint64_t *pre = xmalloc((P + 1) * sizeof *pre);
    int64_t n = orc_graph_item_table(g, count_type, &items, pre);
    if (n < 0) {
        free(ex);
        free(pre);
        return -1;
    }
    for (uint64_t p = 0; p < P; p++)
        if (ex[p])
            for (uint64_t j = pre[p]; j < pre[p + 1]; j++) flags[items[j]] = 1;
    free(items);
    free(pre);
    free(ex);
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* Subset / exclude with BED coordinates: parse_gfa_paths_walks             */
/* (util.rs:208-366; the _multiple variant :22-206 makes the same decisions per count),    */
/* update_tables (:569-722), update_tables_edgecount (:723-795), ActiveTable and           */
/* IntervalContainer (src/util.rs:118-310), quantify_uncovered_bps (abacus.rs:1187-1229).   */
/* usize arithmetic wraps like the reference's release build (no overflow checks).         */
/* No reference-produced output exists for these options: parity unpinned.                 */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    uint64_t s, e;
} ival;
typedef struct {
    ival *v;
    size_t n, cap;
} ivec;
static void ivec_insert(ivec *a, size_t i, ival x) {
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 4;
        a->v = xrealloc(a->v, a->cap * sizeof *a->v);
    }
    memmove(a->v + i + 1, a->v + i, (a->n - i) * sizeof *a->v);
    a->v[i] = x;
    a->n++;
}
static void ivec_remove(ivec *a, size_t i) {
    memmove(a->v + i, a->v + i + 1, (a->n - i - 1) * sizeof *a->v);
    a->n--;
}
static void ivec_clear(ivec *a) {
    free(a->v);
    memset(a, 0, sizeof *a);
}
/* IntervalContainer::add (src/util.rs:215-258); an absent key is an empty list here */
static void icont_add(ivec *x, uint64_t start, uint64_t end) {
    if (x->n == 0) {
        ivec_insert(x, 0, (ival){start, end});
        return;
    }
    /* binary_search_by_key(&start, |(y, _)| y).unwrap_or_else(|z| z): starts are unique */
    size_t i = 0;
    while (i < x->n && x->v[i].s < start) i++;
    if (i > 0 && x->v[i - 1].e >= start) {
        if (x->v[i - 1].e < end) {
            uint64_t stop = end;
            while (i < x->n && x->v[i].s <= end) {
                if (x->v[i].e > stop) stop = x->v[i].e;
                ivec_remove(x, i);
            }
            x->v[i - 1].e = stop;
        }
    } else if (i < x->n && x->v[i].e >= start && x->v[i].s <= end) {
        if (start < x->v[i].s) x->v[i].s = start;
        uint64_t stop = x->v[i].e > end ? x->v[i].e : end;
        while (i + 1 < x->n && x->v[i + 1].s <= end) {
            if (x->v[i + 1].e > stop) stop = x->v[i + 1].e;
            ivec_remove(x, i + 1);
        }
        x->v[i].e = stop;
    } else {
        ivec_insert(x, i, (ival){start, end});
    }
}
/* IntervalContainer::total_coverage (src/util.rs:272-305); ex == NULL is `None` */
static uint64_t icont_total_coverage(const ivec *v, const ival *ex, size_t nex, int have_ex) {
    uint64_t res = 0;
    if (!have_ex) {
        for (size_t k = 0; k < v->n; k++) res = res + v->v[k].e - v->v[k].s;
        return res;
    }
    size_t i = 0;
    for (size_t k = 0; k < v->n; k++) {
        const uint64_t start = v->v[k].s, end = v->v[k].e;
        while (i < nex && ex[i].e <= start) i++;
        if (i < nex && ex[i].s < end) {
            uint64_t m = ex[i].s - 1; /* wraps for 0 like the release build */
            if (end < m) m = end;
            res += m - start;
            if (ex[i].e < end) res += end - ex[i].e + 1;
        } else {
            res += end - start;
        }
    }
    return res;
}
/* ActiveTable (src/util.rs:118-207): items[] + optional annotation per item */
typedef struct {
    uint8_t *items;
    ivec *ann; /* NULL = without annotation */
} active_table;
static void active_annotate(active_table *t, uint64_t id, uint64_t item_len, uint64_t start, uint64_t end) {
    ivec *m = &t->ann[id];
    if (end - start == item_len) {
        t->items[id] = 1;
        ivec_clear(m);
    } else {
        if (start <= end) icont_add(m, start, end);
        /* m.get(&id).unwrap()[0]: the reference panics when nothing was ever added */
        if (m->n > 0 && m->v[0].s == 0 && m->v[0].e == item_len) {
            ivec_clear(m);
            t->items[id] = 1;
        }
    }
}
/* GraphMask::build_subpath_map (abacus.rs:354-382): id -> sorted, merged intervals */
typedef struct {
    smap idx;
    ivec *lists;
    size_t n, cap;
} subpath_map;
static int ival_cmp(const void *a, const void *b) {
    const ival *x = a, *y = b;
    if (x->s != y->s) return x->s < y->s ? -1 : 1;
    if (x->e != y->e) return x->e < y->e ? -1 : 1;
    return 0;
}
static void subpath_map_add(subpath_map *m, const char *id, ival x) {
    uint64_t k;
    if (!smap_get(&m->idx, id, strlen(id), &k)) {
        if (m->n == m->cap) {
            m->cap = m->cap ? m->cap * 2 : 16;
            m->lists = xrealloc(m->lists, m->cap * sizeof *m->lists);
        }
        k = m->n++;
        memset(&m->lists[k], 0, sizeof m->lists[k]);
        smap_put(&m->idx, id, strlen(id), k);
    }
    ivec *l = &m->lists[k];
    ivec_insert(l, l->n, x);
}
static void subpath_map_finish(subpath_map *m) {
    for (size_t k = 0; k < m->n; k++) {
        ivec *v = &m->lists[k];
        qsort(v->v, v->n, sizeof *v->v, ival_cmp);
        size_t w = 0; /* HashSet: drop exact duplicates */
        for (size_t i = 0; i < v->n; i++)
            if (w == 0 || ival_cmp(&v->v[w - 1], &v->v[i]) != 0) v->v[w++] = v->v[i];
        v->n = w;
        size_t i = 1;
        while (i < v->n) {
            if (v->v[i - 1].e >= v->v[i].s) {
                ival x = v->v[i];
                ivec_remove(v, i);
                if (x.e > v->v[i - 1].e) v->v[i - 1].e = x.e;
            } else {
                i++;
            }
        }
    }
}
static const ivec *subpath_map_get(const subpath_map *m, const char *id) {
    uint64_t k;
    return smap_get(&m->idx, id, strlen(id), &k) ? &m->lists[k] : NULL;
}
static void subpath_map_free(subpath_map *m) {
    for (size_t k = 0; k < m->n; k++) ivec_clear(&m->lists[k]);
    free(m->lists);
    smap_free(&m->idx);
}
/* load_coord_list_file + complement_with_group_assignments (abacus.rs:152-210) into a
 * subpath map.  Needs the group names of a preceding orc_graph_path_order* call. */
static int load_subpath_map(const orc_graph *g, const char *file, subpath_map *out) {
    segvec segs = {0};
    smap_init(&out->idx, 64);
    out->lists = NULL;
    out->n = out->cap = 0;
    if (load_coord_list(g, file, &segs) != 0) {
        segvec_free(&segs);
        return -1;
    }
    const uint64_t P = g->n_paths;
    char **keys = xcalloc(P ? P : 1, sizeof *keys);
    for (uint64_t i = 0; i < P; i++) keys[i] = pathseg_clearkey(&g->paths[i]);
    int rc = 0;
    for (size_t si = 0; si < segs.n && rc == 0; si++) {
        const pathseg *ps = &segs.v[si];
        char *k = pathseg_clearkey(ps);
        char *id = pathseg_id(ps);
        int is_path = 0, is_group = 0;
        for (uint64_t i = 0; i < P && !is_path; i++) is_path = strcmp(keys[i], k) == 0;
        if (is_path) {
            ival x = {0, UINT64_MAX};
            if (ps->has_start && ps->has_end) x = (ival){ps->start, ps->end};
            subpath_map_add(out, id, x);
        } else {
            for (uint64_t i = 0; i < P && !is_group; i++)
                is_group = g->path_group && g->path_group[i] && strcmp(g->path_group[i], id) == 0;
            if (is_group && ps->has_start && ps->has_end) {
                set_err("invalid coordinate \"%s\": group identifiers are not allowed to have start/stop information!", id);
                rc = -1;
            } else if (is_group) {
                for (uint64_t i = 0; i < P; i++)
                    if (strcmp(g->path_group[i], id) == 0) {
                        char *pid = pathseg_id(&g->paths[i]);
                        subpath_map_add(out, pid, (ival){0, UINT64_MAX});
                        free(pid);
                    }
            } /* else: unknown path/group -> logged and skipped */
        }
        free(k);
        free(id);
    }
    for (uint64_t i = 0; i < P; i++) free(keys[i]);
    free(keys);
    segvec_free(&segs);
    if (rc == 0) subpath_map_finish(out);
    return rc;
}
/* intersects / is_contained (src/util.rs:370-398) over sorted, disjoint intervals */
static int ivs_intersects(const ival *v, size_t n, ival el) {
    for (size_t k = 0; k < n; k++)
        if (v[k].s <= el.e && v[k].e >= el.s) return 1;
    return 0;
}
static int ivs_is_contained(const ival *v, size_t n, ival el) {
    for (size_t k = 0; k < n; k++)
        if (v[k].s <= el.s && v[k].e >= el.e) return 1;
    return 0;
}

int64_t orc_graph_masked_table(const orc_graph *g, int count_type, const char *subset_file, const char *exclude_file,
                               uint64_t **items_out, uint64_t *prefsum, uint8_t *exclude,
                               uint64_t **uncov_ids, uint64_t **uncov_bps, uint64_t *n_uncov) {
    *items_out = NULL;
    if (uncov_ids) *uncov_ids = NULL;
    if (uncov_bps) *uncov_bps = NULL;
    if (n_uncov) *n_uncov = 0;
    if (count_type == ORC_EDGE && !g->has_edges) {
        set_err("graph was loaded without edge index");
        return -1;
    }
    const uint64_t n_items = count_type == ORC_EDGE ? g->n_edges : g->n_nodes;
    subpath_map inc, exc;
    int have_inc = 0, have_exc = 0;
    if (subset_file) {
        if (load_subpath_map(g, subset_file, &inc) != 0) {
            subpath_map_free(&inc);
            return -1;
        }
        have_inc = 1;
    }
    if (exclude_file) {
        if (load_subpath_map(g, exclude_file, &exc) != 0) {
            subpath_map_free(&exc);
            if (have_inc) subpath_map_free(&inc);
            return -1;
        }
        have_exc = 1;
    }
    /* load_optional_subsetting (abacus.rs:384-425) */
    ivec *covered = NULL; /* subset_covered_bps: bp count with a subset only */
    if (count_type == ORC_BP && have_inc) covered = xcalloc(n_items + 1, sizeof *covered);
    active_table ex = {NULL, NULL};
    if (have_exc) {
        ex.items = xcalloc(n_items + 1, 1);
        if (count_type == ORC_BP) ex.ann = xcalloc(n_items + 1, sizeof *ex.ann);
    }

    int64_t ret = -1;
    FILE *f = fopen(g->gfa_file, "rb");
    if (!f) {
        set_err("cannot open %s", g->gfa_file);
        goto done;
    }
    u64vec items = {0}, sids = {0}, oris = {0};
    uint64_t num_path = 0;
    prefsum[0] = 0;
    char *line = NULL;
    size_t lcap = 0;
    ssize_t n;
    const ival complete = {0, UINT64_MAX};
    int failed = 0;
    while ((n = getline(&line, &lcap, f)) > 0) {
        if (line[0] != 'P' && line[0] != 'W') continue;
        const pathseg *ps = &g->paths[num_path];
        char *id = pathseg_id(ps);
        const ival *ic = &complete, *ec = NULL;
        size_t nic = 1, nec = 0;
        if (have_inc) {
            const ivec *l = subpath_map_get(&inc, id);
            ic = l ? l->v : NULL;
            nic = l ? l->n : 0;
        }
        if (have_exc) {
            const ivec *l = subpath_map_get(&exc, id);
            ec = l ? l->v : NULL;
            nec = l ? l->n : 0;
        }
        free(id);
        const ival span = (ps->has_start && ps->has_end) ? (ival){ps->start, ps->end} : complete;
        uint64_t added = 0;
        /* neither part of the subset nor of the exclude list: an empty entry (util.rs:272-286) */
        if (have_inc && !ivs_intersects(ic, nic, span) && !ivs_intersects(ec, nec, span)) {
            prefsum[num_path + 1] = prefsum[num_path];
            num_path++;
            continue;
        }
        if (parse_steps(g, line, (size_t)n, &sids, &oris) != 0) {
            failed = 1;
            break;
        }
        if (count_type != ORC_EDGE && (!have_inc || ivs_is_contained(ic, nic, span)) &&
            (!have_exc || ivs_is_contained(ec, nec, span))) {
            /* whole path (util.rs:288-315, 1186-1250): every step is pushed; with exclude
             * coordinates for this path all of its nodes are flagged as excluded */
            for (size_t k = 0; k < sids.n; k++) {
                u64vec_push(&items, sids.v[k]);
                if (nec > 0) ex.items[sids.v[k]] = 1;
            }
            added = sids.n;
        } else if (count_type != ORC_EDGE) {
            /* update_tables, util.rs:569-722 */
            size_t i = 0, j = 0;
            uint64_t p = span.s;
            for (size_t k = 0; k < sids.n; k++) {
                const uint64_t sid = sids.v[k];
                const uint64_t l = g->node_lens[sid];
                int stop_here = 0;
                while (i < nic && ic[i].s < p + l && !stop_here) {
                    if (ic[i].e > p) {
                        uint64_t a = ic[i].s > p ? ic[i].s - p : 0, b;
                        if (ic[i].e < p + l) {
                            i++;
                            b = ic[i - 1].e - p;
                        } else {
                            stop_here = 1;
                            b = l;
                        }
                        if (oris.v[k]) { /* backward: mirror the interval inside the node */
                            const uint64_t a2 = l - b, b2 = l - a;
                            a = a2;
                            b = b2;
                        }
                        u64vec_push(&items, sid);
                        added++;
                        if (covered) {
                            if (b - a == l)
                                ivec_clear(&covered[sid]);
                            else
                                icont_add(&covered[sid], a, b);
                        }
                    } else {
                        i++;
                    }
                }
                stop_here = 0;
                while (j < nec && ec[j].s < p + l && !stop_here) {
                    if (ec[j].e > p) {
                        uint64_t a = ec[j].s > p ? ec[j].s - p : 0, b;
                        if (ec[j].e < p + l) {
                            j++;
                            b = ec[j - 1].e - p;
                        } else {
                            stop_here = 1;
                            b = l;
                        }
                        if (oris.v[k]) {
                            const uint64_t a2 = l - b, b2 = l - a;
                            a = a2;
                            b = b2;
                        }
                        if (ex.items) {
                            if (ex.ann)
                                active_annotate(&ex, sid, l, a, b);
                            else
                                ex.items[sid] = 1;
                        }
                    } else {
                        j++;
                    }
                }
                if (i >= nic && j >= nec) break;
                p += l;
            }
        } else if (sids.n > 0) {
            /* update_tables_edgecount, util.rs:723-795 */
            size_t i = 0, j = 0;
            uint64_t p = span.s + g->node_lens[sids.v[0]];
            for (size_t k = 0; k + 1 < sids.n; k++) {
                while (i < nic && ic[i].e <= p) i++;
                while (j < nec && ec[j].e <= p) j++;
                const uint64_t l = g->node_lens[sids.v[k + 1]];
                edge_t e = edge_canonical(sids.v[k], (uint8_t)oris.v[k], sids.v[k + 1], (uint8_t)oris.v[k + 1]);
                const uint64_t eid = emap_get(&g->edge2id, e);
                if (!eid) {
                    set_err("unknown edge %c%llu%c%llu", e.o1 ? '<' : '>', (unsigned long long)e.u,
                            e.o2 ? '<' : '>', (unsigned long long)e.v);
                    failed = 1;
                    break;
                }
                if (i < nic && ic[i].s < p + l) {
                    u64vec_push(&items, eid);
                    added++;
                }
                if (ex.items && j < nec && ec[j].s < p + l)
                    ex.items[eid] = 1;
                else if (i >= nic && j >= nec)
                    break;
                p += l;
            }
            if (failed) break;
        }
        prefsum[num_path + 1] = prefsum[num_path] + added;
        num_path++;
    }
    free(line);
    free(sids.v);
    free(oris.v);
    fclose(f);
    if (failed) {
        free(items.v);
        goto done;
    }
    if (exclude) {
        memset(exclude, 0, n_items + 1);
        if (ex.items) memcpy(exclude, ex.items, n_items + 1);
    }
    /* quantify_uncovered_bps, abacus.rs:1187-1229 */
    if (covered && uncov_ids && uncov_bps && n_uncov) {
        u64vec ids = {0}, bps = {0};
        for (uint64_t sid = 1; sid <= n_items; sid++) {
            if (covered[sid].n == 0) continue;                   /* not a key of the container */
            if (ex.items && ex.items[sid]) continue;             /* completely excluded */
            const uint64_t l = g->node_lens[sid];
            ival whole = {0, l};
            const ival *av = NULL;
            size_t an = 0;
            if (ex.items) { /* get_active_intervals: items[sid] is false here */
                av = ex.ann ? ex.ann[sid].v : NULL;
                an = ex.ann ? ex.ann[sid].n : 0;
            }
            (void)whole;
            const uint64_t cov = icont_total_coverage(&covered[sid], av, an, ex.items != NULL);
            if (cov > l) continue; /* "oops, total coverage is larger than node length": skipped */
            u64vec_push(&ids, sid);
            u64vec_push(&bps, l - cov);
        }
        *uncov_ids = ids.v ? ids.v : xmalloc(8);
        *uncov_bps = bps.v ? bps.v : xmalloc(8);
        *n_uncov = ids.n;
    }
    *items_out = items.v ? items.v : xmalloc(8);
    ret = (int64_t)items.n;
done:
    if (covered) {
        for (uint64_t k = 0; k <= n_items; k++) ivec_clear(&covered[k]);
        free(covered);
    }
    if (ex.ann) {
        for (uint64_t k = 0; k <= n_items; k++) ivec_clear(&ex.ann[k]);
        free(ex.ann);
    }
    free(ex.items);
    if (have_inc) subpath_map_free(&inc);
    if (have_exc) subpath_map_free(&exc);
    return ret;
}

/* construct_hist_bps' fix-up (abacus.rs:779-785): hist[countable[id]] -= uncov; hist[0] += uncov */
void orc_hist_apply_uncovered(const uint32_t *countable, const uint64_t *uncov_ids, const uint64_t *uncov_bps,
                              uint64_t n_uncov, uint64_t *hist) {
    for (uint64_t k = 0; k < n_uncov; k++) {
        hist[countable[uncov_ids[k]]] -= uncov_bps[k];
        hist[0] += uncov_bps[k];
    }
}

/* ------------------------------------------------------------------------------------ */
/* pansyn-v1: counter-based synthetic pangenome generator (integer-only)      */
/* Similarity::set_table up to the Jaccard table, similarity.rs:119-165.  The reference keeps
 * the sums in HashMaps keyed by (x << 64 | y); dense arrays hold the same numbers. */
int orc_similarity(const uint64_t *r, const uint64_t *c, uint64_t n_items, uint64_t n_groups,
                   const uint32_t *node_lens, uint64_t *inter, uint64_t *lens, float *table) {
    const uint64_t G = n_groups;
    uint8_t *seen = xmalloc(G ? G : 1);
    memset(seen, 0, G ? G : 1);
    for (uint64_t k = 0; k < G * G; k++) inter[k] = 0;
    for (uint64_t g = 0; g < G; g++) lens[g] = 0;
    /* tuple_windows over r: index = item id, 0..=n_items (similarity.rs:125,130) */
    for (uint64_t index = 0; index <= n_items; index++) {
        const uint64_t w = node_lens ? node_lens[index] : 1; /* :131,134-138 */
        for (uint64_t a = r[index]; a < r[index + 1]; a++) {
            const uint64_t x = c[a];
            lens[x] += w;
            seen[x] = 1;
            for (uint64_t b = r[index]; b < r[index + 1]; b++) inter[x * G + c[b]] += w; /* :139-149 */
        }
    }
    int rc = 0;
    for (uint64_t g = 0; g < G; g++)
        if (!seen[g]) rc = -1; /* path_lens[&(i as u64)] on a missing key panics (:163) */
    if (rc == 0 && table)
        for (uint64_t i = 0; i < G; i++)
            for (uint64_t j = 0; j < G; j++) {
                const uint64_t x = inter[i * G + j];
                table[i * G + j] = (float)x / (float)(lens[i] + lens[j] - x); /* :162-163 */
            }
    free(seen);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* Similarity::set_table after the Jaccard table (similarity.rs:166-182): Euclidean distances   */
/* between the table's rows, hierarchical clustering, rows / columns / labels reordered.        */
/*                                                                                              */
/* The clustering is `kodama::linkage` (crate kodama 0.3.0, Cargo.toml:34 -- a third-party      */
/* dependency that is NOT in the reference tree).  kodama is a port of Muellner's fastcluster   */
/* (D. Muellner, "Modern hierarchical, agglomerative clustering algorithms", arXiv:1109.2378):  */
/*   Method::Single                       -> minimum spanning tree (Prim from observation 0)    */
/*   Complete / Average / Weighted / Ward -> nearest-neighbour chain                            */
/*   Centroid / Median                    -> the "generic" algorithm = merge the globally       */
/*                                           closest pair of active clusters at every step      */
/* with the Lance-Williams updates of kodama's method.rs in f32, Ward / Centroid / Median on    */
/* squared distances.  MST and NN-chain steps are then stably sorted by dissimilarity; clusters  */
/* get SciPy labels (observation i = i, the cluster made by step k = n + k) and every step lists */
/* its smaller label first.  Restated from the published algorithm; ties between EQUAL          */
/* dissimilarities are broken towards the lower index here (kodama's generic algorithm breaks    */
/* them by the layout of its binary heap, which is not reproduced): parity unpinned, the        */
/* reference holds no similarity output to pin it to.                                           */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    uint64_t c1, c2;
    float d;
} orc_step;

static inline size_t cond_idx(size_t n, size_t i, size_t j) { /* i < j */
    return n * i - i * (i + 1) / 2 + (j - i - 1);
}
static inline float *cond_at(float *dis, size_t n, size_t a, size_t b) {
    return a < b ? &dis[cond_idx(n, a, b)] : &dis[cond_idx(n, b, a)];
}

/* similarity.rs:238-254: f32 throughout; (v1 - v2).powf(2.0) == the correctly rounded square */
static void sim_distances(const float *table, size_t n, float *condensed) {
    size_t k = 0;
    for (size_t row = 0; row + 1 < n; row++)
        for (size_t col = row + 1; col < n; col++) {
            float sum = 0.0f;
            for (size_t x = 0; x < n; x++) {
                const float d = table[row * n + x] - table[col * n + x];
                sum += d * d;
            }
            condensed[k++] = sqrtf(sum);
        }
}

/* kodama method.rs: a = d(x, removed cluster), *b = d(x, kept cluster) */
static void lw_update(int method, float a, float *b, float merged, size_t sa, size_t sb, size_t sx) {
    const float fa = (float)sa, fb = (float)sb, fx = (float)sx;
    switch (method) {
        case 0: if (a < *b) *b = a; break;                                 /* single */
        case 1: if (a > *b) *b = a; break;                                 /* complete */
        case 2: *b = (fa * a + fb * *b) / (fa + fb); break;                /* average */
        case 3: *b = 0.5f * (a + *b); break;                               /* weighted */
        case 4: {                                                          /* ward (squared) */
            const float num = ((fx + fa) * a) + ((fx + fb) * *b) - (fx * merged);
            *b = num / (fa + fb + fx);
            break;
        }
        case 5: {                                                          /* centroid (squared) */
            const float fab = fa + fb;
            *b = (((fa * a) + (fb * *b)) / fab) - ((fa * fb * merged) / (fab * fab));
            break;
        }
        default: *b = (0.5f * (a + *b)) - (merged * 0.25f); break;         /* median (squared) */
    }
}

/* merge `a` into `b` (a is removed, b keeps the merged cluster) */
static void lw_merge(int method, float *dis, size_t n, uint8_t *active, size_t *sizes, size_t a, size_t b, float merged) {
    active[a] = 0;
    for (size_t x = 0; x < n; x++) {
        if (!active[x] || x == b) continue;
        lw_update(method, *cond_at(dis, n, x, a), cond_at(dis, n, x, b), merged, sizes[a], sizes[b], sizes[x]);
    }
    sizes[b] += sizes[a];
}

static size_t link_mst(float *dis, size_t n, orc_step *steps) {
    uint8_t *active = xmalloc(n);
    float *mind = xmalloc(n * sizeof *mind);
    for (size_t i = 0; i < n; i++) {
        active[i] = 1;
        mind[i] = INFINITY;
    }
    size_t cluster = 0, ns = 0;
    active[0] = 0;
    for (size_t it = 0; it + 1 < n; it++) {
        size_t min_obs = n;
        float min_dist = 0.0f;
        for (size_t x = 0; x < n; x++) {
            if (!active[x]) continue;
            const float d = *cond_at(dis, n, x, cluster);
            if (d < mind[x]) mind[x] = d;
            if (min_obs == n || mind[x] < min_dist) {
                min_obs = x;
                min_dist = mind[x];
            }
        }
        steps[ns++] = (orc_step){min_obs, cluster, min_dist};
        active[min_obs] = 0;
        cluster = min_obs;
    }
    free(active);
    free(mind);
    return ns;
}

static size_t link_nnchain(int method, float *dis, size_t n, orc_step *steps) {
    uint8_t *active = xmalloc(n);
    size_t *sizes = xmalloc(n * sizeof *sizes), *chain = xmalloc((n + 1) * sizeof *chain), clen = 0, ns = 0;
    for (size_t i = 0; i < n; i++) {
        active[i] = 1;
        sizes[i] = 1;
    }
    for (size_t it = 0; it + 1 < n; it++) {
        size_t a, b;
        float min;
        if (clen < 4) {
            a = 0;
            while (!active[a]) a++;
            clen = 0;
            chain[clen++] = a;
            b = a + 1;
            while (!active[b]) b++;
            min = dis[cond_idx(n, a, b)];
            for (size_t i = b + 1; i < n; i++)
                if (active[i] && dis[cond_idx(n, a, i)] < min) {
                    min = dis[cond_idx(n, a, i)];
                    b = i;
                }
        } else {
            clen -= 2;
            b = chain[--clen];
            a = chain[clen - 1];
            min = *cond_at(dis, n, a, b);
        }
        for (;;) {
            chain[clen++] = b;
            for (size_t x = 0; x < n; x++) {
                if (!active[x] || x == b) continue;
                const float d = *cond_at(dis, n, x, b);
                if (d < min) {
                    min = d;
                    a = x;
                }
            }
            b = a;
            a = chain[clen - 1];
            if (b == chain[clen - 2]) break;
        }
        steps[ns++] = (orc_step){a, b, min};
        if (a > b) {
            const size_t t = a;
            a = b;
            b = t;
        }
        lw_merge(method, dis, n, active, sizes, a, b, min);
    }
    free(active);
    free(sizes);
    free(chain);
    return ns;
}

/* stable sort by dissimilarity + SciPy labels through a union-find (kodama: LinkageUnionFind::relabel) */
static void link_relabel(orc_step *steps, size_t ns, size_t n) {
    for (size_t i = 1; i < ns; i++) { /* insertion sort: stable */
        const orc_step s = steps[i];
        size_t j = i;
        while (j > 0 && steps[j - 1].d > s.d) {
            steps[j] = steps[j - 1];
            j--;
        }
        steps[j] = s;
    }
    size_t *parent = xmalloc((2 * n) * sizeof *parent);
    for (size_t i = 0; i < 2 * n; i++) parent[i] = i;
    size_t next = n;
    for (size_t i = 0; i < ns; i++) {
        size_t r1 = steps[i].c1, r2 = steps[i].c2;
        while (parent[r1] != r1) r1 = parent[r1];
        while (parent[r2] != r2) r2 = parent[r2];
        parent[r1] = parent[r2] = next++;
        steps[i].c1 = r1 < r2 ? r1 : r2;
        steps[i].c2 = r1 < r2 ? r2 : r1;
    }
    free(parent);
}

/* centroid / median: the closest active pair at every step, labels assigned as the merges happen */
static size_t link_generic(int method, float *dis, size_t n, orc_step *steps) {
    uint8_t *active = xmalloc(n);
    size_t *sizes = xmalloc(n * sizeof *sizes), *label = xmalloc(n * sizeof *label), ns = 0;
    for (size_t i = 0; i < n; i++) {
        active[i] = 1;
        sizes[i] = 1;
        label[i] = i;
    }
    for (size_t it = 0; it + 1 < n; it++) {
        size_t a = n, b = n;
        float min = 0.0f;
        for (size_t i = 0; i < n; i++) {
            if (!active[i]) continue;
            for (size_t j = i + 1; j < n; j++) {
                if (!active[j]) continue;
                const float d = dis[cond_idx(n, i, j)];
                if (a == n || d < min) {
                    min = d;
                    a = i;
                    b = j;
                }
            }
        }
        const size_t l1 = label[a], l2 = label[b];
        steps[ns++] = (orc_step){l1 < l2 ? l1 : l2, l1 < l2 ? l2 : l1, min};
        lw_merge(method, dis, n, active, sizes, a, b, min);
        label[b] = n + it;
    }
    free(active);
    free(sizes);
    free(label);
    return ns;
}

/* table: n x n f32 (the Jaccard table in group order); method: 0 single, 1 complete, 2 average,
 * 3 weighted, 4 ward, 5 centroid (the reference's default, analysis_parameter.rs:287-291), 6 median.
 * perm_out[k] = the group (index in the input order) that ends up in row / column k of the printed
 * table; table is reordered in place like the reference does (rows, then every row). */
int orc_similarity_order(float *table, uint64_t n_groups, int method, uint64_t *perm_out) {
    const size_t n = (size_t)n_groups;
    if (n == 0) return -1; /* `table.len() - 1` underflows in calculate_distances (similarity.rs:248) */
    const size_t nd = n * (n - 1) / 2;
    float *dis = xmalloc((nd ? nd : 1) * sizeof *dis);
    sim_distances(table, n, dis);
    if (method >= 4)
        for (size_t k = 0; k < nd; k++) dis[k] = dis[k] * dis[k];
    orc_step *steps = xmalloc((n ? n : 1) * sizeof *steps);
    size_t ns;
    if (method == 0) ns = link_mst(dis, n, steps);
    else if (method <= 4) ns = link_nnchain(method, dis, n, steps);
    else ns = link_generic(method, dis, n, steps);
    if (method <= 4) link_relabel(steps, ns, n);
    /* get_order_from_dendrogram (similarity.rs:205-217) */
    size_t *leaf = xmalloc((n ? n : 1) * sizeof *leaf), nl = 0;
    for (size_t i = 0; i < ns; i++) {
        if (steps[i].c1 < n) leaf[nl++] = steps[i].c1;
        if (steps[i].c2 < n) leaf[nl++] = steps[i].c2;
    }
    /* :172-174: enumerate, sort by the observation, keep the positions = the inverse permutation */
    size_t *order = xmalloc((n ? n : 1) * sizeof *order);
    for (size_t k = 0; k < nl; k++) order[leaf[k]] = k;
    /* sort_by_indices (:194-203) applied to the rows, to every row, and to the labels (:175-179);
     * an index list shorter than the table (n = 1: no steps at all) leaves it untouched */
    size_t *who = xmalloc(n * sizeof *who), *ind = xmalloc(n * sizeof *ind);
    for (size_t i = 0; i < n; i++) who[i] = i;
    for (size_t i = 0; i < nl; i++) ind[i] = order[i];
    for (size_t i = 0; i < nl; i++)
        while (i != ind[i]) {
            const size_t ni = ind[i], t = ind[i];
            ind[i] = ind[ni];
            ind[ni] = t;
            const size_t w = who[i];
            who[i] = who[ni];
            who[ni] = w;
        }
    float *tmp = xmalloc(n * n * sizeof *tmp);
    for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < n; j++) tmp[i * n + j] = table[who[i] * n + who[j]];
    memcpy(table, tmp, n * n * sizeof *tmp);
    for (size_t i = 0; i < n; i++) perm_out[i] = who[i];
    free(tmp);
    free(who);
    free(ind);
    free(order);
    free(leaf);
    free(steps);
    free(dis);
    return 0;
}

/* AbacusByGroup::to_tsv, node/bp branch without `total` and without multiplicities
 * (abacus.rs:1093-1112) */
void orc_table_row(const uint64_t *r, const uint64_t *c, uint64_t i, uint64_t n_groups,
                   uint64_t bp, uint64_t *out) {
    uint64_t k = r[i];
    const uint64_t end = r[i + 1];
    for (uint64_t j = 0; j < n_groups; j++) {
        if (k == end || j < c[k]) out[j] = 0;
        else if (j == c[k]) {
            out[j] = bp;
            k++;
        }
    }
}

/* the same with AbacusByGroup.v (abacus.rs:1098-1108 for node/bp, :1158-1166 for edge).  The edge
   branch indexes v by the GROUP id j instead of the slot k -- restated as it is; returns -1 where
   that runs past the end of v (the reference panics). */
int orc_table_row_values(const uint64_t *r, const uint64_t *c, const uint32_t *v, uint64_t nnz, uint64_t i,
                         uint64_t n_groups, uint64_t bp, int is_edge, uint64_t *out) {
    uint64_t k = r[i];
    const uint64_t end = r[i + 1];
    for (uint64_t j = 0; j < n_groups; j++) {
        if (k == end || j < c[k]) out[j] = 0;
        else if (j == c[k]) {
            if (is_edge) {
                if (j >= nnz) return -1;
                out[j] = v[j];
            } else {
                out[j] = (uint64_t)v[k] * bp;
            }
            k++;
        }
    }
    return 0;
}

void orc_exp2(const double *x, double *y, uint64_t n) {
    for (uint64_t k = 0; k < n; k++) y[k] = exp2(x[k]);
}

void orc_log2(const double *x, double *y, uint64_t n) {
    for (uint64_t k = 0; k < n; ++k) y[k] = log2(x[k]);
}

/* ------------------------------------------------------------------------------------ */
uint64_t pansyn_splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static uint64_t pansyn_key(uint64_t seed, uint64_t stream) {
    return pansyn_splitmix64(seed ^ (0xA0761D6478BD642FULL * (stream + 1)));
}
#define PANSYN_ONE53 (1ULL << 53)

uint32_t pansyn_node_len(uint64_t seed, uint64_t i) {
    uint64_t t = pansyn_splitmix64(pansyn_key(seed, 1) + i) >> 11;
    if (t < PANSYN_ONE53 / 100 * 55) return 1;
    uint64_t e = pansyn_splitmix64(pansyn_key(seed, 2) + i);
    uint64_t k = e ? (uint64_t)__builtin_clzll(e) : 63;
    if (k > 63) k = 63;
    uint64_t f16 = e & 0xFFFF;
    uint64_t len = 1 + ((180 * ((k << 16) + f16)) >> 16);
    return (uint32_t)(len > 50000 ? 50000 : len);
}

void pansyn_node_lens(uint64_t seed, uint64_t n_nodes, uint32_t *out /* n_nodes+1 */) {
    out[0] = 0;
    for (uint64_t i = 1; i <= n_nodes; i++) out[i] = pansyn_node_len(seed, i);
}

uint64_t pansyn_node_thr(uint64_t seed, uint64_t i, uint64_t n_paths) {
    uint64_t t = pansyn_splitmix64(pansyn_key(seed, 3) + i) >> 11;
    if (t < PANSYN_ONE53 / 100 * 20) return PANSYN_ONE53;           /* core: in every path */
    if (t < PANSYN_ONE53 / 100 * 65) return PANSYN_ONE53 / n_paths; /* near-singleton */
    return pansyn_splitmix64(pansyn_key(seed, 4) + i) >> 11;        /* shell: uniform freq */
}

int64_t pansyn_generate(uint64_t seed, uint64_t n_nodes, uint64_t n_paths, uint64_t **items_out,
                        uint64_t *prefsum) {
    u64vec items = {0};
    uint64_t *thr = xmalloc((n_nodes + 1) * sizeof *thr);
    for (uint64_t i = 1; i <= n_nodes; i++) thr[i] = pansyn_node_thr(seed, i, n_paths);
    uint64_t k5 = pansyn_key(seed, 5);
    prefsum[0] = 0;
    for (uint64_t p = 0; p < n_paths; p++) {
        uint64_t kp = pansyn_splitmix64(k5 + p);
        size_t begin = items.n;
        for (uint64_t i = 1; i <= n_nodes; i++) {
            uint64_t h = pansyn_splitmix64(kp + i);
            if ((h >> 11) < thr[i]) {
                u64vec_push(&items, i);
                if ((h & 63) == 0) u64vec_push(&items, i);
            }
        }
        if (p % 16 == 15) { /* descending path: reverse */
            for (size_t a = begin, b = items.n; a + 1 < b; a++, b--) {
                uint64_t t = items.v[a];
                items.v[a] = items.v[b - 1];
                items.v[b - 1] = t;
            }
        }
        prefsum[p + 1] = items.n;
    }
    free(thr);
    *items_out = items.v ? items.v : xmalloc(8);
    return (int64_t)items.n;
}

/* pansyn-v1r: the paths of pansyn-v1 REARRANGED, position by position, the way real pangenome paths stray from the order of
 * the ids -- a model of inversions, duplications and translocations, not of any reference function (the reference's sweep,
 * abacus.rs:727-742, does not care in which order a group's steps come).  A path is taken in blocks of 64 steps; block B of
 * path p draws r = h mod 10000, h = splitmix64(splitmix64(key(seed, 8) + p) + B), and (the first and the last block of a
 * path stay as they are)
 *   r < 100         the block is reversed in place                                  (1 %: a local inversion)
 *   100 <= r < 110  the block becomes a copy of block (h >> 20) mod B of the same path  (0.1 %: a jump back, a duplication)
 *   110 <= r < 115  every id of the block is moved by 1 + (h >> 24) mod (n_nodes - 1), wrapping round  (0.05 %: a translocation)
 * Copies are taken from the path as pansyn-v1 made it.  Path lengths do not change. */
void pansyn_rearrange(uint64_t seed, uint64_t n_nodes, uint64_t n_paths, uint64_t *items, const uint64_t *prefsum) {
    const uint64_t k8 = pansyn_key(seed, 8);
    for (uint64_t p = 0; p < n_paths; p++) {
        const uint64_t s0 = prefsum[p], len = prefsum[p + 1] - s0;
        if (len == 0) continue;
        const uint64_t n_blocks = (len + 63) / 64;
        const uint64_t kp = pansyn_splitmix64(k8 + p);
        uint64_t *orig = xmalloc(len * sizeof *orig);
        memcpy(orig, items + s0, len * sizeof *orig);
        for (uint64_t B = 1; B + 1 < n_blocks; B++) {
            const uint64_t h = pansyn_splitmix64(kp + B), r = h % 10000;
            const uint64_t b0 = B * 64, bl = len - b0 < 64 ? len - b0 : 64;
            if (r < 100) {
                for (uint64_t j = 0; j < bl; j++) items[s0 + b0 + j] = orig[b0 + bl - 1 - j];
            } else if (r < 110) {
                const uint64_t src = ((h >> 20) % B) * 64;
                for (uint64_t j = 0; j < bl; j++) items[s0 + b0 + j] = orig[src + j];
            } else if (r < 115 && n_nodes > 1) {
                const uint64_t off = 1 + (h >> 24) % (n_nodes - 1);
                for (uint64_t j = 0; j < bl; j++) items[s0 + b0 + j] = (orig[b0 + j] - 1 + off) % n_nodes + 1;
            }
        }
        free(orig);
    }
}
