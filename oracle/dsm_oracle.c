/* ORACLE / TEST INFRASTRUCTURE -- see dsm_oracle.h.  "FF.cpp" below is
 * /root/reference/surfel_fusion/src/fusion_functions.cpp, "SM.cpp" is surfel_map.cpp.
 *
 * Expression typing follows C's usual arithmetic conversions exactly as the
 * reference's C++ does (FLT_EVAL_METHOD == 0, built with -ffp-contract=off):
 * the reference's macro constants are double literals, so any expression that
 * touches one is evaluated in double and rounded when stored to a float.
 */
#include "dsm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define CELL 8      /* SP_SIZE,        fusion_functions.h:10 */
#define SWEEPS 3    /* ITERATION_NUM,  fusion_functions.h:8  */
#define WORKERS 10  /* THREAD_NUM,     fusion_functions.h:9  */
#define ANGLE_COS 0.1 /* MAX_ANGLE_COS, fusion_functions.h:11 */

struct dsmo_ctx {
    int w, h, gw, gh, n_seed;
    float fx, fy, cx, cy, far_d, near_d;
    double huber, baseline, disp_err, min_tol;
    const uint8_t *img;
    size_t img_step;
    const float *dep;
    size_t dep_step;
    dsmo_seed *seed;
    int32_t *label;
    float *space; /* reference keeps doubles (FF.h:34); every stored value is float-exact */
    float *nmap;
    float *scratch; /* per-seed gather buffers, 4 x 256 x 3 floats */
};

static inline float px_img(const dsmo_ctx *c, int r, int col) { return (float)c->img[(size_t)r * c->img_step + col]; }
static inline float px_dep(const dsmo_ctx *c, int r, int col) {
    return *(const float *)((const char *)c->dep + (size_t)r * c->dep_step + (size_t)col * 4);
}

/* worker k's [begin,end) over n items: FF.cpp:198-202 / 392-396 / 471-475 */
static void chunk(int n, int k, int *b, int *e) {
    int step = n / WORKERS;
    *b = step * k;
    *e = (k == WORKERS - 1) ? n : *b + step;
}

dsmo_ctx *dsmo_create(int w, int h, float fx, float fy, float cx, float cy, float far_d, float near_d) {
    dsmo_ctx *c = (dsmo_ctx *)calloc(1, sizeof(*c));
    c->w = w; c->h = h; c->gw = w / CELL; c->gh = h / CELL; /* FF.cpp:14-15 */
    c->n_seed = c->gw * c->gh;
    c->fx = fx; c->fy = fy; c->cx = cx; c->cy = cy; c->far_d = far_d; c->near_d = near_d;
    c->huber = 0.4; c->baseline = 0.5; c->disp_err = 4.0; c->min_tol = 0.1; /* fusion_functions.h:13-16 */
    /* One record in front of the table: image sizes with (size mod 8) > 4 leave border pixels without a candidate
     * cell, and pixels whose every cost exceeds the 1e6 sentinel have none either; the reference then labels them -1
     * and touches superpixel_seeds[-1] (FF.cpp:400,442-451, 242) -- memory in front of the vector.  The restatement
     * pins that record: all zero at the start of a frame, i.e. never stable, no normal (fusion skips it after the
     * free-space test).  PARITY UNPINNED for those pixels: the reference's behaviour there is undefined. */
    c->seed = (dsmo_seed *)calloc((size_t)c->n_seed + 1, sizeof(dsmo_seed)) + 1;
    c->label = (int32_t *)calloc((size_t)w * h, sizeof(int32_t));
    c->space = (float *)calloc((size_t)w * h * 3, sizeof(float));
    c->nmap = (float *)calloc((size_t)w * h * 3, sizeof(float));
    c->scratch = (float *)calloc(4 * 256 * 3, sizeof(float));
    return c;
}

void dsmo_destroy(dsmo_ctx *c) {
    if (!c) return;
    free(c->seed - 1); free(c->label); free(c->space); free(c->nmap); free(c->scratch); free(c);
}

void dsmo_set_constants(dsmo_ctx *c, double huber, double baseline, double disparity_error, double min_tolerate) {
    c->huber = huber; c->baseline = baseline; c->disp_err = disparity_error; c->min_tol = min_tolerate;
}

void dsmo_set_frame(dsmo_ctx *c, const uint8_t *img, size_t img_step, const float *depth, size_t depth_step) {
    c->img = img; c->img_step = img_step; c->dep = depth; c->dep_step = depth_step;
}

/* ------------------------------------------------------------------ seeds */

/* window [8g-4, 8g+12) clipped to [0, dim-1): note the *exclusive* dim-1, the last
 * row/column never contributes (FF.cpp:482-489, 602-609). */
static void clipped_window(const dsmo_ctx *c, int gx, int gy, int *x0, int *x1, int *y0, int *y1) {
    *x0 = gx * CELL + CELL / 2 - CELL; *y0 = gy * CELL + CELL / 2 - CELL;
    *x1 = *x0 + 2 * CELL; *y1 = *y0 + 2 * CELL;
    if (*x0 < 0) *x0 = 0;
    if (*y0 < 0) *y0 = 0;
    if (*x1 > c->w - 1) *x1 = c->w - 1;
    if (*y1 > c->h - 1) *y1 = c->h - 1;
}

/* FF.cpp:577-629.  Fields the reference leaves unassigned are pinned to zero. */
void dsmo_initialize_seeds(dsmo_ctx *c) {
    for (int s = 0; s < c->n_seed; s++) {
        int gx = s % c->gw, gy = s / c->gw;
        int ix = gx * CELL + CELL / 2, iy = gy * CELL + CELL / 2;
        if (ix > c->w - 1) ix = c->w - 1;
        if (iy > c->h - 1) iy = c->h - 1;
        dsmo_seed sd;
        memset(&sd, 0, sizeof(sd));
        sd.x = (float)ix; sd.y = (float)iy;
        sd.mean_intensity = px_img(c, iy, ix);
        sd.mean_depth = px_dep(c, iy, ix);
        if ((double)sd.mean_depth < 0.01) { /* FF.cpp:600 */
            int x0, x1, y0, y1, found = 0;
            clipped_window(c, gx, gy, &x0, &x1, &y0, &y1);
            for (int y = y0; y < y1 && !found; y++)
                for (int x = x0; x < x1; x++) {
                    float d = px_dep(c, y, x);
                    if ((double)d > 0.01) { sd.mean_depth = d; found = 1; break; }
                }
        }
        c->seed[s] = sd;
    }
}

/* FF.cpp:364-387.  Returns whether the depth term applied. */
static int pixel_cost(const dsmo_ctx *c, int s, float pix_i, float pix_invd, int x, int y, float *no_d, float *with_d) {
    const dsmo_seed *sd = &c->seed[s];
    float ddx = sd->x - (float)x, ddy = sd->y - (float)y;
    float dist = ddx * ddx + ddy * ddy;
    float cost = 0.0f;
    cost += dist / (float)((CELL / 2) * (CELL / 2));
    float di = sd->mean_intensity - pix_i;
    cost = (float)((double)cost + (double)(di * di) / 100.0); /* FF.cpp:376 */
    *no_d = cost;
    *with_d = cost;
    if (sd->mean_depth > 0 && pix_invd > 0) {
        float dd = (float)(1.0 / (double)sd->mean_depth - (double)pix_invd); /* FF.cpp:380 */
        *with_d = (float)((double)cost + (double)(dd * dd) * 400.0);       /* FF.cpp:381 */
        return 1;
    }
    return 0;
}

/* FF.cpp:389-453, all ten row strips in index order == one row-major scan.
 * The `stable` test and the `stable=false` write act on live state (Gauss-Seidel). */
void dsmo_update_pixels(dsmo_ctx *c) {
    for (int y = 0; y < c->h; y++)
        for (int x = 0; x < c->w; x++) {
            int32_t *lab = &c->label[y * c->w + x];
            if (c->seed[*lab].stable) continue; /* FF.cpp:400 */
            float pi = px_img(c, y, x);
            float d = px_dep(c, y, x);
            float invd = 0.0f;
            if ((double)d > 0.01) invd = (float)(1.0 / (double)d); /* FF.cpp:404-405 */
            int bx = x / CELL, by = y / CELL;
            float best_d = 1e6f, best_n = 1e6f;
            int arg_d = -1, arg_n = -1, all_depth = 1;
            for (int ox = -1; ox <= 1; ox++)      /* x offset outer, FF.cpp:413 */
                for (int oy = -1; oy <= 1; oy++) { /* y offset inner, FF.cpp:414 */
                    int gx = bx + ox, gy = by + oy;
                    int ax = abs(gx * CELL + CELL / 2 - x), ay = abs(gy * CELL + CELL / 2 - y);
                    if (!(ax < CELL && ay < CELL && gx >= 0 && gx < c->gw && gy >= 0 && gy < c->gh)) continue;
                    float cn, cd;
                    all_depth &= pixel_cost(c, gy * c->gw + gx, pi, invd, x, y, &cn, &cd);
                    if (cd < best_d) { best_d = cd; arg_d = gy * c->gw + gx; }
                    if (cn < best_n) { best_n = cn; arg_n = gy * c->gw + gx; }
                }
            int pick = all_depth ? arg_d : arg_n; /* FF.cpp:442-451 */
            *lab = pick;
            c->seed[pick].stable = 0;
        }
}

/* FF.cpp:468-562 for one worker's chunk; returns at the first unstable seed that owns no pixel (FF.cpp:516-517). */
static void update_seed_chunk(dsmo_ctx *c, int b, int e) {
    float *dlist = c->scratch;
    for (int s = b; s < e; s++) {
        dsmo_seed *sd = &c->seed[s];
        if (sd->stable) continue;
        int x0, x1, y0, y1;
        clipped_window(c, s % c->gw, s / c->gw, &x0, &x1, &y0, &y1);
        float sx = 0, sy = 0, si = 0, ni = 0, sdp = 0, nd = 0;
        int n = 0;
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                if (c->label[y * c->w + x] != s) continue;
                sx += (float)x; sy += (float)y; ni += 1.0f;
                si += px_img(c, y, x);
                float d = px_dep(c, y, x);
                if ((double)d > 0.1) { dlist[n++] = d; sdp += d; nd += 1.0f; } /* FF.cpp:508-513 */
            }
        if (ni == 0) return; /* sic */
        si /= ni; sx /= ni; sy /= ni;
        float pi = sd->mean_intensity, px = sd->x, py = sd->y;
        sd->mean_intensity = si; sd->x = sx; sd->y = sy;
        float moved = fabsf(pi - si) + fabsf(px - sx) + fabsf(py - sy);
        if ((double)moved < 0.2) sd->stable = 1;
        if (nd > 0) {
            float md = sdp / nd;
            for (int it = 0; it < 5; it++) { /* damped Huber-Newton, FF.cpp:534-554 */
                float a = 0, bb = 0;
                for (int k = 0; k < n; k++) {
                    float r = md - dlist[k];
                    if ((double)r < c->huber && (double)r > -c->huber) { a += 2 * r; bb += 2; }
                    else a = (float)((double)a + (r > 0 ? c->huber : -1 * c->huber));
                }
                float delta = (float)((double)(-a) / ((double)bb + 10.0));
                md = md + delta;
                if ((double)delta < 0.01 && (double)delta > -0.01) break;
            }
            sd->mean_depth = md;
        } else {
            sd->mean_depth = 0.0f;
        }
    }
}

void dsmo_update_seeds(dsmo_ctx *c) {
    for (int k = 0; k < WORKERS; k++) {
        int b, e;
        chunk(c->n_seed, k, &b, &e);
        update_seed_chunk(c, b, e);
    }
}

/* ---------------------------------------------------------------- normals */

static inline void back_project_f(const dsmo_ctx *c, float u, float v, float d, float *x, float *y, float *z) {
    *x = (u - c->cx) / c->fx * d; /* FF.cpp:94-96: float arithmetic, stored to double */
    *y = (v - c->cy) / c->fy * d;
    *z = d;
}

/* FF.cpp:644-662 */
static void fill_space(dsmo_ctx *c) {
    for (int y = 0; y < c->h; y++)
        for (int x = 0; x < c->w; x++) {
            float *p = &c->space[(size_t)(y * c->w + x) * 3];
            back_project_f(c, (float)x, (float)y, px_dep(c, y, x), &p[0], &p[1], &p[2]);
        }
}

/* FF.cpp:664-712; the ten strips together cover rows 1..h-2 (one row is visited twice, idempotently). */
static void fill_pixel_normals(dsmo_ctx *c) {
    for (int y = 1; y < c->h - 1; y++)
        for (int x = 1; x < c->w - 1; x++) {
            size_t i = (size_t)(y * c->w + x) * 3;
            const float *p = &c->space[i], *pr = &c->space[i + 3], *pd = &c->space[i + (size_t)c->w * 3];
            if ((double)p[2] < 0.1 || (double)pr[2] < 0.1 || (double)pd[2] < 0.1) continue;
            float rx = pr[0] - p[0], ry = pr[1] - p[1], rz = pr[2] - p[2];
            float dx = pd[0] - p[0], dy = pd[1] - p[1], dz = pd[2] - p[2];
            float nx = ry * dz - rz * dy, ny = rz * dx - rx * dz, nz = rx * dy - ry * dx;
            float len = sqrtf(nx * nx + ny * ny + nz * nz);
            nx /= len; ny /= len; nz /= len;
            float va = (nx * p[0] + ny * p[1] + nz * p[2]) / sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
            if ((double)va > -ANGLE_COS && (double)va < ANGLE_COS) continue;
            c->nmap[i] = nx; c->nmap[i + 1] = ny; c->nmap[i + 2] = nz;
        }
}

/* column-major general 4x4 inverse, adjugate over determinant (stands in for Eigen, FF.cpp:59,176) */
#define DEF_INVERSE4(NAME, T)                                                                        \
    static void NAME(const T *a, T *o) {                                                             \
        T s0 = a[0] * a[5] - a[1] * a[4], s1 = a[0] * a[9] - a[1] * a[8], s2 = a[0] * a[13] - a[1] * a[12];       \
        T s3 = a[4] * a[9] - a[5] * a[8], s4 = a[4] * a[13] - a[5] * a[12], s5 = a[8] * a[13] - a[9] * a[12];     \
        T c5 = a[10] * a[15] - a[11] * a[14], c4 = a[6] * a[15] - a[7] * a[14], c3 = a[6] * a[11] - a[7] * a[10]; \
        T c2 = a[2] * a[15] - a[3] * a[14], c1 = a[2] * a[11] - a[3] * a[10], c0 = a[2] * a[7] - a[3] * a[6];     \
        T det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;                           \
        T id = (T)1 / det;                                                                           \
        o[0] = (a[5] * c5 - a[9] * c4 + a[13] * c3) * id;                                            \
        o[4] = (-a[4] * c5 + a[8] * c4 - a[12] * c3) * id;                                           \
        o[8] = (a[7] * s5 - a[11] * s4 + a[15] * s3) * id;                                           \
        o[12] = (-a[6] * s5 + a[10] * s4 - a[14] * s3) * id;                                         \
        o[1] = (-a[1] * c5 + a[9] * c2 - a[13] * c1) * id;                                           \
        o[5] = (a[0] * c5 - a[8] * c2 + a[12] * c1) * id;                                            \
        o[9] = (-a[3] * s5 + a[11] * s2 - a[15] * s1) * id;                                          \
        o[13] = (a[2] * s5 - a[10] * s2 + a[14] * s1) * id;                                          \
        o[2] = (a[1] * c4 - a[5] * c2 + a[13] * c0) * id;                                            \
        o[6] = (-a[0] * c4 + a[4] * c2 - a[12] * c0) * id;                                           \
        o[10] = (a[3] * s4 - a[7] * s2 + a[15] * s0) * id;                                           \
        o[14] = (-a[2] * s4 + a[6] * s2 - a[14] * s0) * id;                                          \
        o[3] = (-a[1] * c3 + a[5] * c1 - a[9] * c0) * id;                                            \
        o[7] = (a[0] * c3 - a[4] * c1 + a[8] * c0) * id;                                             \
        o[11] = (-a[3] * s3 + a[7] * s1 - a[11] * s0) * id;                                          \
        o[15] = (a[2] * s3 - a[6] * s1 + a[10] * s0) * id;                                           \
    }
DEF_INVERSE4(inverse4f, float)
DEF_INVERSE4(inverse4d, double)
void dsmo_inverse4f(const float *a, float *out) { inverse4f(a, out); }

/* FF.cpp:104-188: Huber-weighted Gauss-Newton plane fit n.p + b = 0 on centred points (modified in place). */
static void huber_plane(const dsmo_ctx *c, float *nx, float *ny, float *nz, float *nb, float *pt, int n) {
    float mx = 0, my = 0, mz = 0;
    for (int i = 0; i < n; i++) { mx += pt[3 * i]; my += pt[3 * i + 1]; mz += pt[3 * i + 2]; }
    mx /= (float)n; my /= (float)n; mz /= (float)n;
    *nb = 0;
    for (int i = 0; i < n; i++) { pt[3 * i] -= mx; pt[3 * i + 1] -= my; pt[3 * i + 2] -= mz; }
    const double hr = c->huber;
    for (int it = 0; it < 5; it++) {
        double H[16], J[4], Hi[16];
        memset(H, 0, sizeof H); memset(J, 0, sizeof J);
        for (int i = 0; i < n; i++) {
            float p0 = pt[3 * i], p1 = pt[3 * i + 1], p2 = pt[3 * i + 2];
            float r = p0 * *nx + p1 * *ny + p2 * *nz + *nb;
            if ((double)r < hr && (double)r > -1 * hr) {
                J[0] += (double)(2 * r * p0); J[1] += (double)(2 * r * p1); J[2] += (double)(2 * r * p2); J[3] += (double)(2 * r);
                /* H is column-major here: H[col*4+row]; the reference fills all 16 entries (FF.cpp:140-155) */
                H[0] += (double)(2 * p0 * p0); H[4] += (double)(2 * p0 * p1); H[8] += (double)(2 * p0 * p2); H[12] += (double)(2 * p0);
                H[1] += (double)(2 * p1 * p0); H[5] += (double)(2 * p1 * p1); H[9] += (double)(2 * p1 * p2); H[13] += (double)(2 * p1);
                H[2] += (double)(2 * p2 * p0); H[6] += (double)(2 * p2 * p1); H[10] += (double)(2 * p2 * p2); H[14] += (double)(2 * p2);
                H[3] += (double)(2 * p0); H[7] += (double)(2 * p1); H[11] += (double)(2 * p2); H[15] += 2;
            } else if ((double)r >= hr) {
                J[0] += hr * (double)p0; J[1] += hr * (double)p1; J[2] += hr * (double)p2; J[3] += hr;
            } else if ((double)r <= -1 * hr) {
                J[0] += -1 * hr * (double)p0; J[1] += -1 * hr * (double)p1; J[2] += -1 * hr * (double)p2; J[3] += -1 * hr;
            }
        }
        H[0] += 5; H[5] += 5; H[10] += 5; H[15] += 5;
        inverse4d(H, Hi);
        double u[4];
        for (int i = 0; i < 4; i++) u[i] = ((Hi[i] * J[0] + Hi[4 + i] * J[1]) + Hi[8 + i] * J[2]) + Hi[12 + i] * J[3];
        *nx = (float)((double)*nx - u[0]); *ny = (float)((double)*ny - u[1]);
        *nz = (float)((double)*nz - u[2]); *nb = (float)((double)*nb - u[3]);
    }
    *nb = *nb - (*nx * mx + *ny * my + *nz * mz);
    float len = sqrtf(*nx * *nx + *ny * *ny + *nz * *nz);
    *nx /= len; *ny /= len; *nz /= len; *nb /= len;
}

/* FF.cpp:792-914.  The reference does not clip x and only range-checks the linear index, so a
 * window can wrap into a neighbouring row; wrapped pixels can never carry label == s once the
 * grid is at least 3 cells wide (they sit at the opposite image edge), so skipping x outside
 * [0,w) is equivalent. */
static void fit_seed_planes(dsmo_ctx *c) {
    float *dlist = c->scratch, *nlist = c->scratch + 256, *plist = c->scratch + 256 * 4, *inl = c->scratch + 256 * 7;
    const int npx = c->w * c->h;
    for (int s = 0; s < c->n_seed; s++) {
        dsmo_seed *sd = &c->seed[s];
        int x0 = (s % c->gw) * CELL + CELL / 2 - CELL, y0 = (s / c->gw) * CELL + CELL / 2 - CELL;
        int n = 0;
        float far2 = 0;
        for (int y = y0; y < y0 + 2 * CELL; y++)
            for (int x = x0; x < x0 + 2 * CELL; x++) {
                int i = y * c->w + x;
                if (i < 0 || i >= npx) continue;
                if (x < 0 || x >= c->w) continue; /* see note above */
                if (c->label[i] != s) continue;
                float ex = (float)x - sd->x, ey = (float)y - sd->y;
                float d2 = ex * ex + ey * ey;
                if (d2 > far2) far2 = d2;
                float d = px_dep(c, y, x);
                if ((double)d > 0.05) {
                    dlist[n] = d;
                    memcpy(&nlist[3 * n], &c->nmap[(size_t)i * 3], 12);
                    memcpy(&plist[3 * n], &c->space[(size_t)i * 3], 12);
                    n++;
                }
            }
        if ((float)n < 16) continue; /* FF.cpp:841 */
        float md = sd->mean_depth;
        float nx = 0, ny = 0, nz = 0, nb = 0, n_in = 0;
        int m = 0;
        for (int k = 0; k < n; k++) {
            float r = md - dlist[k];
            if ((double)r < c->huber && (double)r > -c->huber) {
                nx += nlist[3 * k]; ny += nlist[3 * k + 1]; nz += nlist[3 * k + 2];
                n_in += 1;
                memcpy(&inl[3 * m], &plist[3 * k], 12);
                m++;
            }
        }
        if ((double)(n_in / (float)n) < 0.8) continue; /* FF.cpp:862 */
        float len = sqrtf(nx * nx + ny * ny + nz * nz);
        nx = nx / len; ny = ny / len; nz = nz / len;
        huber_plane(c, &nx, &ny, &nz, &nb, inl, m);
        float bx, by, bz;
        back_project_f(c, sd->x, sd->y, md, &bx, &by, &bz);
        double ax = bx, ay = by, az = bz;
        float k = (float)(-1 * (ax * (double)nx + ay * (double)ny + az * (double)nz) - (double)nb); /* FF.cpp:890 */
        ax += (double)(k * nx); ay += (double)(k * ny); az += (double)(k * nz);
        md = (float)az;
        float vc = (float)(-1.0 * ((double)nx * ax + (double)ny * ay + (double)nz * az) / sqrt(ax * ax + ay * ay + az * az));
        if (vc < 0) { vc = -vc; nx = -nx; ny = -ny; nz = -nz; }
        sd->norm_x = nx; sd->norm_y = ny; sd->norm_z = nz;
        sd->posi_x = (float)ax; sd->posi_y = (float)ay; sd->posi_z = (float)az;
        sd->mean_depth = md;
        sd->view_cos = vc;
        sd->size = sqrtf(far2);
    }
}

void dsmo_calculate_norms(dsmo_ctx *c) { /* FF.cpp:916-958 */
    fill_space(c);
    fill_pixel_normals(c);
    fit_seed_planes(c);
}

void dsmo_generate_super_pixels(dsmo_ctx *c) { /* FF.cpp:960-975 */
    memset(c->seed - 1, 0, sizeof(dsmo_seed) * ((size_t)c->n_seed + 1));
    memset(c->label, 0, sizeof(int32_t) * (size_t)c->w * c->h);
    memset(c->nmap, 0, sizeof(float) * 3 * (size_t)c->w * c->h);
    dsmo_initialize_seeds(c);
    for (int it = 0; it < SWEEPS; it++) {
        dsmo_update_pixels(c);
        dsmo_update_seeds(c);
    }
    dsmo_calculate_norms(c);
}

/* ----------------------------------------------------------------- fusion */

static inline void xform_point(const float *m, const float *p, float *o) { /* 4x4 * (x,y,z,1), FF.cpp:220 */
    for (int i = 0; i < 3; i++) o[i] = ((m[i] * p[0] + m[4 + i] * p[1]) + m[8 + i] * p[2]) + m[12 + i] * 1.0f;
}
static inline void xform_dir(const float *m, const float *v, float *o) { /* block<3,3> * v, FF.cpp:228 */
    for (int i = 0; i < 3; i++) o[i] = (m[i] * v[0] + m[4 + i] * v[1]) + m[8 + i] * v[2];
}
static inline float depth_weight(float d) { /* FF.cpp:99-102 */
    double w = 1.0 / (double)d / (double)d;
    return (float)(1.0 < w ? 1.0 : w); /* std::min(w, 1.0): a NaN w is returned as is */
}

/* FF.cpp:190-313 over the whole array (the ten chunks are independent). */
static void fuse_local(dsmo_ctx *c, int ref_idx, const float *pose, const float *inv, dsmo_surfel *ls, int n) {
    for (int i = 0; i < n; i++) {
        dsmo_surfel *e = &ls[i];
        if (ref_idx - e->last_update > 5 && e->update_times < 5) { e->update_times = 0; continue; }
        if (e->update_times == 0) continue;
        float pw[3] = {e->px, e->py, e->pz}, pc[3], nw[3] = {e->nx, e->ny, e->nz}, nc[3];
        xform_point(inv, pw, pc);
        if (pc[2] < c->near_d || pc[2] > c->far_d) continue;
        xform_dir(inv, nw, nc);
        float u = pc[0] * c->fx / pc[2] + c->cx, v = pc[1] * c->fy / pc[2] + c->cy; /* FF.cpp:85-89 */
        double ud = (double)u + 0.5, vd = (double)v + 0.5;
        /* int(x) of NaN/out-of-range is INT_MIN on x86-64; either way the bounds test below rejects */
        int ui = (ud >= -2147483648.0 && ud < 2147483648.0) ? (int)ud : (-2147483647 - 1);
        int vi = (vd >= -2147483648.0 && vd < 2147483648.0) ? (int)vd : (-2147483647 - 1);
        if (ui < 1 || ui > c->w - 2 || vi < 1 || vi > c->h - 2) continue;
        if ((double)pc[2] < (double)px_dep(c, vi, ui) - 1.0) { e->update_times = 0; continue; }
        dsmo_seed *sd = &c->seed[c->label[vi * c->w + ui]];
        if (sd->norm_x == 0 && sd->norm_y == 0 && sd->norm_z == 0) continue;
        if ((double)sd->view_cos < ANGLE_COS) continue;
        float cam_f = (float)((double)(fabsf(c->fx) + fabsf(c->fy)) / 2.0);
        float tol = (float)((double)(pc[2] * pc[2]) / (c->baseline * (double)cam_f) * c->disp_err);
        tol = (float)((double)tol < c->min_tol ? c->min_tol : (double)tol);
        if (pc[2] < sd->mean_depth - tol) continue;
        if (pc[2] > sd->mean_depth + tol) continue;
        float ncos = nc[0] * sd->norm_x + nc[1] * sd->norm_y + nc[2] * sd->norm_z;
        if ((double)ncos < ANGLE_COS) { e->update_times = 0; continue; }
        float w0 = e->weight, w1 = depth_weight(sd->mean_depth), ws = w0 + w1;
        float sc[3] = {sd->posi_x, sd->posi_y, sd->posi_z}, sw[3];
        xform_point(pose, sc, sw);
        float fpx = (e->px * w0 + w1 * sw[0]) / ws, fpy = (e->py * w0 + w1 * sw[1]) / ws, fpz = (e->pz * w0 + w1 * sw[2]) / ws;
        float fn[3] = {nc[0] * w0 + w1 * sd->norm_x, nc[1] * w0 + w1 * sd->norm_y, nc[2] * w0 + w1 * sd->norm_z};
        double len = (double)sqrtf(fn[0] * fn[0] + fn[1] * fn[1] + fn[2] * fn[2]);
        fn[0] = (float)((double)fn[0] / len); fn[1] = (float)((double)fn[1] / len); fn[2] = (float)((double)fn[2] / len);
        float fw[3];
        xform_dir(pose, fn, fw);
        e->px = fpx; e->py = fpy; e->pz = fpz;
        e->nx = fw[0]; e->ny = fw[1]; e->nz = fw[2];
        e->weight = ws;
        e->color = sd->mean_intensity;
        float nsz = sd->size * fabsf(sd->mean_depth / (cam_f * sd->view_cos));
        if (nsz < e->size) e->size = nsz;
        e->last_update = ref_idx;
        e->update_times += 1;
        sd->fused = 1;
    }
}

/* FF.cpp:315-361 */
static int spawn_surfels(dsmo_ctx *c, int ref_idx, const float *pose, dsmo_surfel *out, int cap) {
    int k = 0;
    for (int s = 0; s < c->n_seed; s++) {
        const dsmo_seed *sd = &c->seed[s];
        if (sd->mean_depth == 0) continue;
        if (sd->fused) continue;
        if ((double)sd->view_cos < ANGLE_COS) continue;
        if (sd->norm_x == 0 && sd->norm_y == 0 && sd->norm_z == 0) continue;
        float pc[3] = {sd->posi_x, sd->posi_y, sd->posi_z}, nc[3] = {sd->norm_x, sd->norm_y, sd->norm_z}, pw[3], nw[3];
        xform_point(pose, pc, pw);
        xform_dir(pose, nc, nw);
        float cam_f = (float)((double)(fabsf(c->fx) + fabsf(c->fy)) / 2.0);
        if (k >= cap) return -1;
        dsmo_surfel *e = &out[k++];
        e->px = pw[0]; e->py = pw[1]; e->pz = pw[2];
        e->nx = nw[0]; e->ny = nw[1]; e->nz = nw[2];
        e->size = sd->size * fabsf(sd->mean_depth / (cam_f * sd->view_cos));
        e->color = sd->mean_intensity;
        e->weight = depth_weight(sd->mean_depth);
        e->update_times = 1;
        e->last_update = ref_idx;
    }
    return k;
}

int dsmo_fuse_initialize_map(dsmo_ctx *c, int ref_idx, const uint8_t *img, size_t img_step, const float *depth,
                             size_t depth_step, const float *pose16, dsmo_surfel *local, int n_local,
                             dsmo_surfel *new_out, int new_cap, int *n_new) {
    float inv[16];
    dsmo_set_frame(c, img, img_step, depth, depth_step);
    dsmo_generate_super_pixels(c);
    inverse4f(pose16, inv); /* FF.cpp:59 */
    fuse_local(c, ref_idx, pose16, inv, local, n_local);
    int k = spawn_surfels(c, ref_idx, pose16, new_out, new_cap);
    if (k < 0) return -1;
    *n_new = k;
    return 0;
}

/* SM.cpp:1077-1109, the serial loop as written */
int dsmo_compact(dsmo_surfel *local, int *n_local, int cap, const dsmo_surfel *fresh, int n_fresh) {
    int n = *n_local, nh = 0;
    int *holes = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++)
        if (local[i].update_times == 0) holes[nh++] = i;
    for (int j = 0; j < n_fresh; j++) {
        if (fresh[j].update_times == 0) continue;
        if (nh > 0) local[holes[--nh]] = fresh[j];
        else {
            if (n >= cap) { free(holes); return -1; }
            local[n++] = fresh[j];
        }
    }
    while (nh > 0) { local[holes[--nh]] = local[n - 1]; n--; }
    free(holes);
    *n_local = n;
    return 0;
}

int dsmo_fuse_map(dsmo_ctx *c, int ref_idx, const uint8_t *img, size_t img_step, const float *depth,
                  size_t depth_step, const float *pose16, dsmo_surfel *local, int *n_local, int cap, int *n_new) {
    dsmo_surfel *fresh = (dsmo_surfel *)malloc(sizeof(dsmo_surfel) * (size_t)c->n_seed);
    int k = 0;
    int rc = dsmo_fuse_initialize_map(c, ref_idx, img, img_step, depth, depth_step, pose16, local, *n_local, fresh,
                                      c->n_seed, &k);
    if (rc == 0) rc = dsmo_compact(local, n_local, cap, fresh, k);
    free(fresh);
    *n_new = k;
    return rc;
}

/* SM.cpp:750-789 (active) and 712-733 (inactive): p' = M * (p,1), n' = M[0:3,0:3] * n */
void dsmo_warp(dsmo_surfel *s, int n, const float *m) {
    for (int i = 0; i < n; i++) {
        float p[3] = {s[i].px, s[i].py, s[i].pz}, v[3] = {s[i].nx, s[i].ny, s[i].nz}, o[3], w[3];
        xform_point(m, p, o);
        xform_dir(m, v, w);
        s[i].px = o[0]; s[i].py = o[1]; s[i].pz = o[2];
        s[i].nx = w[0]; s[i].ny = w[1]; s[i].nz = w[2];
    }
}

/* SM.cpp:1476-1497 */
int dsmo_extract_key(dsmo_surfel *local, int n, int key, dsmo_surfel *out) {
    int k = 0;
    for (int i = 0; i < n; i++)
        if (local[i].update_times > 0 && local[i].last_update == key) {
            out[k++] = local[i];
            local[i].update_times = 0;
        }
    return k;
}

/* ------------------------------------------------------------------- taps */
void dsmo_get_labels(dsmo_ctx *c, int32_t *out) { memcpy(out, c->label, sizeof(int32_t) * (size_t)c->w * c->h); }
void dsmo_set_labels(dsmo_ctx *c, const int32_t *in) { memcpy(c->label, in, sizeof(int32_t) * (size_t)c->w * c->h); }
void dsmo_get_seeds(dsmo_ctx *c, void *out) { memcpy(out, c->seed, sizeof(dsmo_seed) * (size_t)c->n_seed); }
void dsmo_set_seeds(dsmo_ctx *c, const void *in) { memcpy(c->seed, in, sizeof(dsmo_seed) * (size_t)c->n_seed); }
void dsmo_get_norm_map(dsmo_ctx *c, float *out) { memcpy(out, c->nmap, sizeof(float) * 3 * (size_t)c->w * c->h); }
