// TEST INFRASTRUCTURE: the product's node-level host logic (densesurfelmapping_amd/csrc/dsm_surfel_map.cpp,
// included verbatim below) on top of a CPU stand-in for the engine calls it makes, so that the pose-graph /
// stamp-matching / bookkeeping code can be checked against the reference node on a box without a GPU.  The
// stand-in implements the handful of dsm_* entry points with std::vector storage and the C restatement
// oracle (oracle/dsm_oracle.c) for the per-frame arithmetic.  Never shipped, never loaded by the package.
#include "../densesurfelmapping_amd/csrc/dsm_surfel_map.cpp"

#include "../oracle/dsm_oracle.h"

struct dsm_handle {
    dsm_config cfg;
    dsmo_ctx *ctx = nullptr;
    std::vector<dsm_surfel> local, store;
    std::vector<float> cloud; // 4 per point
    std::vector<uint8_t> image;
    std::vector<float> depth;
    std::string err;
};

extern "C" {

int dsm_config_init(dsm_config *cfg, int width, int height, float fx, float fy, float cx, float cy, float far_dist, float near_dist, int rgbd) {
    memset(cfg, 0, sizeof *cfg);
    cfg->width = width; cfg->height = height;
    cfg->fx = fx; cfg->fy = fy; cfg->cx = cx; cfg->cy = cy;
    cfg->far_dist = far_dist; cfg->near_dist = near_dist;
    cfg->huber_range = rgbd ? 0.05 : 0.4;
    cfg->baseline = rgbd ? 0.08 : 0.5;
    cfg->disparity_error = rgbd ? 1.0 : 4.0;
    cfg->min_tolerate_diff = rgbd ? 0.05 : 0.1;
    return DSM_OK;
}
int dsm_create(const dsm_config *cfg, dsm_handle **out) {
    dsm_handle *h = new dsm_handle();
    h->cfg = *cfg;
    h->ctx = dsmo_create(cfg->width, cfg->height, cfg->fx, cfg->fy, cfg->cx, cfg->cy, cfg->far_dist, cfg->near_dist);
    dsmo_set_constants(h->ctx, cfg->huber_range, cfg->baseline, cfg->disparity_error, cfg->min_tolerate_diff);
    *out = h;
    return DSM_OK;
}
void dsm_destroy(dsm_handle *h) {
    if (!h) return;
    dsmo_destroy(h->ctx);
    delete h;
}
const char *dsm_last_error(const dsm_handle *h) { return h ? h->err.c_str() : ""; }
int dsm_host_alloc(void **out, size_t bytes) {
    *out = malloc(bytes ? bytes : 1);
    return *out ? DSM_OK : DSM_E_HIP;
}
void dsm_host_free(void *p) { free(p); }
int dsm_map_upload(dsm_handle *h, const dsm_surfel *s, int32_t n) {
    h->local.assign(s, s + n);
    return DSM_OK;
}
int dsm_map_size(dsm_handle *h, int32_t *n) {
    *n = (int32_t)h->local.size();
    return DSM_OK;
}
int dsm_map_capacity(const dsm_handle *, int32_t *cap) {
    *cap = 1 << 30;
    return DSM_OK;
}
int dsm_map_download(dsm_handle *h, dsm_surfel *out, int32_t cap, int32_t *n) {
    *n = (int32_t)h->local.size();
    if (*n > cap) return DSM_E_CAPACITY;
    if (*n) memcpy(out, h->local.data(), sizeof(dsm_surfel) * h->local.size());
    return DSM_OK;
}
int dsm_frame_upload(dsm_handle *h, int, const uint8_t *image, size_t img_step, const float *depth, size_t depth_step) {
    const int w = h->cfg.width, hh = h->cfg.height;
    h->image.resize((size_t)w * hh);
    h->depth.resize((size_t)w * hh);
    for (int y = 0; y < hh; y++) {
        memcpy(&h->image[(size_t)y * w], image + (size_t)y * img_step, (size_t)w);
        memcpy(&h->depth[(size_t)y * w], (const uint8_t *)depth + (size_t)y * depth_step, (size_t)w * 4);
    }
    return DSM_OK;
}
int dsm_fuse_frame_resident(dsm_handle *h, int, int reference_frame_index, const float *pose16) {
    const int w = h->cfg.width;
    int n = (int)h->local.size(), n_new = 0;
    const int cap = n + (w / 8) * (h->cfg.height / 8);
    h->local.resize((size_t)cap);
    const int rc = dsmo_fuse_map(h->ctx, reference_frame_index, h->image.data(), (size_t)w, h->depth.data(), (size_t)w * 4, pose16,
                                 (dsmo_surfel *)h->local.data(), &n, cap, &n_new);
    h->local.resize((size_t)n);
    return rc == 0 ? DSM_OK : DSM_E_INVALID;
}
int dsm_map_warp(dsm_handle *h, const float *warp16) {
    dsmo_warp((dsmo_surfel *)h->local.data(), (int)h->local.size(), warp16);
    return DSM_OK;
}
int dsm_store_size(dsm_handle *h, int32_t *n) {
    *n = (int32_t)h->store.size();
    return DSM_OK;
}
int dsm_store_deactivate(dsm_handle *h, int32_t key, int32_t *begin, int32_t *n) {
    std::vector<dsm_surfel> out(h->local.size() + 1);
    const int k = dsmo_extract_key((dsmo_surfel *)h->local.data(), (int)h->local.size(), key, (dsmo_surfel *)out.data());
    *begin = (int32_t)h->store.size();
    *n = k;
    for (int i = 0; i < k; i++) {
        h->store.push_back(out[(size_t)i]);
        h->cloud.push_back(out[(size_t)i].px); h->cloud.push_back(out[(size_t)i].py);
        h->cloud.push_back(out[(size_t)i].pz); h->cloud.push_back(out[(size_t)i].color);
    }
    return DSM_OK;
}
int dsm_store_activate(dsm_handle *h, int32_t begin, int32_t n) {
    h->local.insert(h->local.end(), h->store.begin() + begin, h->store.begin() + begin + n);
    return DSM_OK;
}
int dsm_store_erase(dsm_handle *h, int32_t begin, int32_t n) {
    h->store.erase(h->store.begin() + begin, h->store.begin() + begin + n);
    h->cloud.erase(h->cloud.begin() + 4 * (size_t)begin, h->cloud.begin() + 4 * ((size_t)begin + (size_t)n));
    return DSM_OK;
}
int dsm_store_warp(dsm_handle *h, int32_t n_groups, const int32_t *offsets, const float *mats16, const uint8_t *changed) {
    for (int g = 0; g < n_groups; g++) {
        if (!changed[g]) continue;
        const int b = offsets[g], e = offsets[g + 1];
        dsmo_warp((dsmo_surfel *)h->store.data() + b, e - b, mats16 + 16 * g);
        for (int i = b; i < e - 1; i++) { // all but the last point (surfel_map.cpp:742)
            h->cloud[4 * (size_t)i + 0] = h->store[(size_t)i].px;
            h->cloud[4 * (size_t)i + 1] = h->store[(size_t)i].py;
            h->cloud[4 * (size_t)i + 2] = h->store[(size_t)i].pz;
            h->cloud[4 * (size_t)i + 3] = h->store[(size_t)i].color;
        }
    }
    return DSM_OK;
}
int dsm_store_download(dsm_handle *h, int32_t begin, int32_t n, dsm_surfel *surfels_out, float *xyzi_out) {
    if (n && surfels_out) memcpy(surfels_out, h->store.data() + begin, sizeof(dsm_surfel) * (size_t)n);
    if (n && xyzi_out) memcpy(xyzi_out, h->cloud.data() + 4 * (size_t)begin, 16 * (size_t)n);
    return DSM_OK;
}

} // extern "C"
