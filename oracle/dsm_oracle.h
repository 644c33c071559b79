/* ORACLE / TEST INFRASTRUCTURE -- never linked into or called by the product path.
 *
 * Plain-C restatement of the reference's per-frame surfel-fusion hot path
 * (surfel_fusion/src/fusion_functions.cpp + the compaction in
 * surfel_fusion/src/surfel_map.cpp:1077-1109).  Schedule: the reference's ten
 * worker threads executed in index order, i.e. the "serial schedule" that
 * oracle/_ref/libdsm_ref_serial.so pins.
 *
 * Pinning status: validated bit-for-bit (labels, seed table, surfel arrays)
 * against the reference's own translation unit compiled in place
 * (oracle/ref_driver.cpp -> oracle/_ref/) by tests/test_oracle_vs_ref.py and
 * against the committed vectors in tests/golden/.  The only arithmetic not
 * pinned by reference code is Eigen's 4x4 inverse (Eigen3 is un-vendored and
 * absent): both checkers use the adjugate/determinant closed form.
 */
#ifndef DSM_ORACLE_H
#define DSM_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* elements.h:22-31 -- 44 bytes */
typedef struct {
    float px, py, pz;
    float nx, ny, nz;
    float size, color, weight;
    int32_t update_times, last_update;
} dsmo_surfel;

/* elements.h:5-20 -- 60 bytes */
typedef struct {
    float x, y;
    float size;
    float norm_x, norm_y, norm_z;
    float posi_x, posi_y, posi_z;
    float view_cos;
    float mean_depth;
    float mean_intensity;
    uint8_t fused, stable;
    uint8_t pad_[2];
    float min_eigen_value, max_eigen_value;
} dsmo_seed;

typedef struct dsmo_ctx dsmo_ctx;

dsmo_ctx *dsmo_create(int w, int h, float fx, float fy, float cx, float cy, float far_d, float near_d);
void dsmo_destroy(dsmo_ctx *c);
/* fusion_functions.h:13-21: HUBER_RANGE, BASELINE, DISPARITY_ERROR, MIN_TOLERATE_DIFF */
void dsmo_set_constants(dsmo_ctx *c, double huber, double baseline, double disparity_error, double min_tolerate);

int dsmo_fuse_initialize_map(dsmo_ctx *c, int ref_idx, const uint8_t *img, size_t img_step, const float *depth,
                             size_t depth_step, const float *pose16, dsmo_surfel *local, int n_local,
                             dsmo_surfel *new_out, int new_cap, int *n_new);
int dsmo_fuse_map(dsmo_ctx *c, int ref_idx, const uint8_t *img, size_t img_step, const float *depth,
                  size_t depth_step, const float *pose16, dsmo_surfel *local, int *n_local, int cap, int *n_new);
/* surfel_map.cpp:1077-1109 on its own */
int dsmo_compact(dsmo_surfel *local, int *n_local, int cap, const dsmo_surfel *fresh, int n_fresh);

void dsmo_get_labels(dsmo_ctx *c, int32_t *out);
void dsmo_set_labels(dsmo_ctx *c, const int32_t *in);
void dsmo_get_seeds(dsmo_ctx *c, void *out);
void dsmo_set_seeds(dsmo_ctx *c, const void *in);
void dsmo_get_norm_map(dsmo_ctx *c, float *out);
void dsmo_set_frame(dsmo_ctx *c, const uint8_t *img, size_t img_step, const float *depth, size_t depth_step);
void dsmo_generate_super_pixels(dsmo_ctx *c);
void dsmo_initialize_seeds(dsmo_ctx *c);
void dsmo_update_pixels(dsmo_ctx *c);
void dsmo_update_seeds(dsmo_ctx *c);
void dsmo_calculate_norms(dsmo_ctx *c);

/* surfel_map.cpp:750-789 / 712-733: rigid warp of positions and normals by a column-major 4x4 float
 * matrix (Eigen MatrixXf products; accumulation left to right as everywhere in the shims -- Eigen3 itself
 * is absent, so this stays "parity unpinned" like the 4x4 inverse). */
void dsmo_warp(dsmo_surfel *s, int n, const float *m16);
/* surfel_map.cpp:1476-1497: move the live surfels whose last_update == key out of `local` (in index
 * order) into `out`, marking their slots deleted (update_times = 0); returns how many. */
int dsmo_extract_key(dsmo_surfel *local, int n, int key, dsmo_surfel *out);

/* general 4x4 inverse (adjugate / determinant), column-major */
void dsmo_inverse4f(const float *a, float *out);

#ifdef __cplusplus
}
#endif
#endif
