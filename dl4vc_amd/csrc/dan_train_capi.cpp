// C ABI of the training step (include/dl4vc_dan_train.h): parameter table, per-step schedule of the kernels of
// dan_train.hip (and the read-axis reductions / highway forward / pooling kernels shared with the inference path).
// Host code only.  Reference: dl4vc/trainer.py:109-439 (loop body), dl4vc/model.py:434-961 (forward), main.py:99-117.
#include "../../include/dl4vc_dan_train.h"
#include "dan_kernels.h"
#include "dan_train.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace dan;

namespace {

struct Param {
    std::string name;
    std::vector<int64_t> shape;      // state-dict shape
    int64_t off = 0;                 // offset (floats) into the flat buffers
    int64_t rows = 1, cols = 0, ld = 0;   // stored as rows x ld with cols valid (ld > cols only for the two FC matrices)
    bool trainable = true;
    int64_t stored() const { return rows * ld; }
    int64_t numel() const { return rows * cols; }
};

struct View {                        // a state-dict tensor = a window of a stored parameter (the six heads share one matrix)
    int param = -1;
    int64_t off = 0, rows = 1, cols = 0;
    std::vector<int64_t> shape;
};

struct LayerP {                      // indices into the parameter table, -1 = absent
    int conv_w = -1, conv_b = -1, bn_g = -1, bn_b = -1, res_w = -1, res_b = -1, bot_w = -1, bot_b = -1, cmp_w = -1, cmp_b = -1;
    int cin = 0, cout = 0, dil = 1, kg = 0;
    bool residual = false;
};

std::string g_create_error;

}  // namespace

struct dan_trainer {
    dan_config cfg{};
    dan_train_hyper hp{};
    int max_batch = 0;
    bool finalized = false;
    mutable std::string err;
    int64_t step = 0;
    int F = 0, n0 = 0, n1 = 0;
    int64_t F_stride = 0, n0_stride = 0, n1_stride = 0;
    std::vector<Param> params;
    std::map<std::string, View> views;                        // by state-dict name
    std::vector<LayerP> layers;
    int p_emb = -1, p_fc0w = -1, p_fc0b = -1, p_fc1w = -1, p_fc1b = -1, p_hw = -1, p_hb = -1;
    int64_t n_flat = 0;
    std::map<std::string, std::vector<float>> init;          // tensors set before finalize
    std::vector<float> pe_host;
    // running statistics (not optimised): [layers][CPAD] each
    float *d_rmean = nullptr, *d_rvar = nullptr;
    // flat buffers
    float *P = nullptr, *G = nullptr, *M = nullptr, *V = nullptr;
    std::vector<void*> allocs;
    // constants / packed weights (refreshed every step)
    float* d_pe = nullptr;
    int *d_inv = nullptr, *d_canon = nullptr;                 // layer-1 channel maps: canonical -> reference (48), reference -> canonical
    std::vector<float*> pk_conv_f, pk_conv_d, pk_res_f, pk_res_d, pk_bot_f, pk_bot_d;
    std::vector<float*> pk_wino_f, pk_wino_d;                 // Winograd F(2,3) forms of the conv / its data gradient (dilation-2 layers)
    std::vector<int> wino_layer;
    PackJob* d_pack_jobs = nullptr;          // the step's re-packing as one launch (build_pack_jobs)
    long long* d_cmpb_off = nullptr;         // [layers] offsets of the compression biases in the flat buffers
    long long* d_cmpw_off = nullptr;         // [layers] offsets of the compression weights
    int n_pack_jobs = 0, n_pack_blocks = 0;
    std::vector<int> lazy_x;                 // [l] 1: x_l = bn(a_l) is never written -- its consumers form it from a_l as they load
    float* d_xtap = nullptr;                 // scratch for the "act:x<l>" debug tap of such a layer
    float* d_wino_u = nullptr;                                // [128][128][4] scratch of the weight transform
    float *d_wc_pk = nullptr, *d_wct = nullptr, *d_bc_pad = nullptr;
    float* d_bias = nullptr;                                  // [layers][3][CPAD]: conv bias, residual bias, bottleneck bias (padded)
    float *d_coef_f = nullptr, *d_coef_b = nullptr, *d_smean = nullptr, *d_sinv = nullptr;
    // inputs / targets
    uint8_t* d_in = nullptr;
    uint8_t* h_stage = nullptr;                  // pinned mirror of d_in | d_tg8 | d_tgf: a step's inputs go up in three asynchronous copies
    uint8_t* d_tg8 = nullptr;
    float* d_tgf = nullptr;
    uint8_t* d_mask[3] = {nullptr, nullptr, nullptr};
    // activations
    std::vector<float*> d_a, d_x, d_pool;
    float* d_h = nullptr;
    float *d_stats = nullptr;
    int stat_entries_max = 0;                                  // entries d_stats holds (dan_train_finalize); every producer is checked against it
    double* d_bp = nullptr;
    float *d_feat = nullptr, *d_featd = nullptr, *d_hid0 = nullptr, *d_hid0d = nullptr, *d_hid1 = nullptr, *d_hid1d = nullptr;
    float *d_logits = nullptr, *d_dlogits = nullptr, *d_losses = nullptr, *d_site_terms = nullptr;
    uint8_t* d_close = nullptr;
    // gradients of activations
    float *d_dhid1d = nullptr, *d_dhid1 = nullptr, *d_dhid0d = nullptr, *d_dhid0 = nullptr, *d_dfeatd = nullptr, *d_dfeat = nullptr;
    float *d_du = nullptr, *d_g[2] = {nullptr, nullptr}, *d_dn = nullptr, *d_dpool = nullptr, *d_dh = nullptr;
    float *d_partial = nullptr, *d_bias_partial = nullptr, *d_hw_partial = nullptr;
    float* d_clip = nullptr;
    float* d_dhw = nullptr;                                    // [layers][rows][HPAD]
    float* d_split_hw = nullptr;                               // split-K partials of the highway weight-gradient GEMM
    long long split_hw_floats = 0;
    float* d_split_ws = nullptr;                               // split-K partials of the forward FC GEMMs
    long long split_ws_floats = 0;
    double* d_l0tot = nullptr;                                  // layer 1's backward by bins: the totals (launch_l0_backward)
    float* d_l0tab = nullptr;                                   // layer 1's forward by table: rebuilt every step (launch_l0_train_tables)
    int last_B = 0;
    // dan_train_backward_begin / _end: the step in flight and the event behind which the FC-side gradients are final
    bool pending = false;
    hipEvent_t ev_tail = nullptr;
    float dp_mean_sites = 0.f, dp_ce_den[2] = {0.f, 0.f};     // dan_train_set_global_batch: for the next backward only
};

namespace {

int failt(const dan_trainer* t, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (t) t->err = buf; else g_create_error = buf;
    return code;
}

// launch_train_row's result: the statistics entries it wrote, or < 0 when it refused (nothing was launched then)
int check_row_launch(const dan_trainer* t, int entries, const char* what, int layer) {
    if (entries == TRAIN_ROW_ERR_STATS)
        return failt(t, DAN_ERR_STATE, "%s launch of layer %d would write more statistics entries than d_stats holds", what, layer + 1);
    if (entries == TRAIN_ROW_ERR_FORM)
        return failt(t, DAN_ERR_STATE, "%s launch of layer %d carries a second product (w3) that the selected kernel form ignores", what, layer + 1);
    return entries < 0 ? failt(t, DAN_ERR_STATE, "%s launch of layer %d refused (%d)", what, layer + 1, entries) : DAN_OK;
}

#define HIPT(t, call)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return failt(t, DAN_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <typename T>
int talloc(dan_trainer* t, T** p, size_t count, bool zero = true) {
    void* q = nullptr;
    const size_t bytes = std::max<size_t>(count * sizeof(T), 256);
    HIPT(t, hipMalloc(&q, bytes));
    t->allocs.push_back(q);
    if (zero) HIPT(t, hipMemset(q, 0, bytes));
    *p = (T*)q;
    return DAN_OK;
}

int build_pack_jobs(dan_trainer* t);
bool pool_after(const dan_config& c, int l1) { return (c.pool_layers_mask >> l1) & 1u; }
bool is_residual(const dan_config& c, int l1) {
    return c.residual_start > 0 && l1 >= c.residual_start && !(l1 == c.layers && c.c_init != c.c_final);
}
void layer_dims(const dan_config& c, int l1, int* cin, int* cout, int* dil) {
    const int in0 = 2 * EMBED + (c.use_q ? 1 : 0) + (c.use_strand ? 1 : 0) + (c.use_mask ? 3 : 0);
    if (l1 == 1) { *cin = in0; *cout = c.c_init; *dil = 1; }
    else if (l1 < c.layers) { *cin = c.c_init; *cout = c.c_init; *dil = c.dil_mid; }
    else { *cin = c.c_init; *cout = c.c_final; *dil = c.dil_final; }
}

int add_param(dan_trainer* t, const std::string& name, std::vector<int64_t> shape, int64_t rows, int64_t cols, int64_t ld,
              bool trainable = true) {
    Param p;
    p.name = name; p.shape = std::move(shape); p.rows = rows; p.cols = cols; p.ld = ld; p.trainable = trainable;
    p.off = t->n_flat;
    t->n_flat += (p.stored() + 3) / 4 * 4;                   // 16-byte aligned tensors
    View v;
    v.param = (int)t->params.size(); v.rows = rows; v.cols = cols; v.shape = p.shape;
    t->views[name] = v;
    t->params.push_back(p);
    return (int)t->params.size() - 1;
}

int add_vec(dan_trainer* t, const std::string& name, std::vector<int64_t> shape, bool trainable = true) {
    int64_t n = 1;
    for (auto s : shape) n *= s;
    return add_param(t, name, std::move(shape), 1, n, n, trainable);
}

float* pp(dan_trainer* t, int idx) { return t->P + t->params[idx].off; }
float* gp(dan_trainer* t, int idx) { return t->G + t->params[idx].off; }

}  // namespace

extern "C" {

const char* dan_train_last_error(const dan_trainer_t* t) { return t ? t->err.c_str() : g_create_error.c_str(); }

int dan_train_create(const dan_config* cfg, const dan_train_hyper* hyper, int32_t max_batch, dan_trainer_t** out) {
    if (!cfg || !hyper || !out) return failt(nullptr, DAN_ERR_INVALID_ARG, "dan_train_create: null argument");
    *out = nullptr;
    const dan_config& c = *cfg;
    if (c.precision != 0) return failt(nullptr, DAN_ERR_INVALID_ARG, "training runs in fp32 only (precision 0)");
    if (c.layers < 1 || c.layers > DAN_MAX_LAYERS) return failt(nullptr, DAN_ERR_INVALID_ARG, "layers must be in 1..%d", DAN_MAX_LAYERS);
    if (c.reads < 1 || c.length < 8 || c.length > MPOS) return failt(nullptr, DAN_ERR_INVALID_ARG, "reads >= 1 and length in 8..%d required", MPOS);
    if (c.c_init < 1 || c.c_init > CPAD || c.c_final < 1 || c.c_final > CPAD) return failt(nullptr, DAN_ERR_INVALID_ARG, "channel counts must be in 1..%d", CPAD);
    if (c.layers == 1 && c.c_init != c.c_final)
        return failt(nullptr, DAN_ERR_INVALID_ARG, "a single conv layer needs init_conv_channels == final_conv_channels (model.py:214,257)");
    if (c.bottleneck < 0 || c.bottleneck > HPAD) return failt(nullptr, DAN_ERR_INVALID_ARG, "bottleneck must be in 0..%d", HPAD);
    if (c.dil_mid < 1 || c.dil_mid > HALO || c.dil_final < 1 || c.dil_final > HALO) return failt(nullptr, DAN_ERR_INVALID_ARG, "dilations must be in 1..%d", HALO);
    if (c.residual_start == 1 || c.residual_start < 0) return failt(nullptr, DAN_ERR_INVALID_ARG, "Do not allow residuals starting at conv layer %d", c.residual_start);
    if ((c.pool_layers_mask & 1u) || (c.pool_layers_mask >> c.layers)) return failt(nullptr, DAN_ERR_INVALID_ARG, "pool layers must lie in 1..layers-1");
    if (c.fc_sizes[0] < 1 || c.fc_sizes[1] < 1) return failt(nullptr, DAN_ERR_INVALID_ARG, "fc_sizes must be positive");
    if (max_batch < 1 || max_batch > 4096) return failt(nullptr, DAN_ERR_INVALID_ARG, "max_batch must be in 1..4096");
    if (hyper->dropout < 0.f || hyper->dropout >= 1.f) return failt(nullptr, DAN_ERR_INVALID_ARG, "dropout must be in [0, 1)");
    if (c.layers > 1 && c.c_init != c.c_final && is_residual(c, c.layers)) return failt(nullptr, DAN_ERR_INVALID_ARG, "internal: residual on a narrowing layer");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || c.device_id < 0 || c.device_id >= ndev)
        return failt(nullptr, DAN_ERR_NO_DEVICE, "no HIP device %d (found %d): the training step has no CPU path", c.device_id, ndev);
    dan_trainer* t = new dan_trainer();
    t->cfg = c; t->hp = *hyper; t->max_batch = max_batch;
    t->F = 2 * c.c_final * c.length + c.layers * c.bottleneck * c.reads;
    t->F_stride = ((int64_t)t->F + 15) / 16 * 16;
    t->n0 = c.fc_sizes[0]; t->n1 = c.fc_sizes[1];
    t->n0_stride = (t->n0 + 15) / 16 * 16;
    t->n1_stride = (t->n1 + 15) / 16 * 16;
    // ---- parameter table (state-dict names; SURVEY.md section 8b "Weights contract")
    const int L = c.length, H = c.bottleneck;
    t->p_emb = add_vec(t, "embeddings.weight", {VOCAB, EMBED});
    t->layers.resize(c.layers);
    for (int l = 0; l < c.layers; ++l) {
        LayerP& lp = t->layers[l];
        layer_dims(c, l + 1, &lp.cin, &lp.cout, &lp.dil);
        lp.kg = (l == 0) ? KG0 : KGC;
        lp.residual = is_residual(c, l + 1);
        const std::string s = std::to_string(l);
        lp.conv_w = add_vec(t, "conv1D_layers." + s + ".weight", {lp.cout, lp.cin, 1, 3});
        lp.conv_b = add_vec(t, "conv1D_layers." + s + ".bias", {lp.cout});
        // the reference builds the BatchNorm modules whether or not it calls them (model.py:217): they are part of the state
        lp.bn_g = add_vec(t, "bn1D_layers." + s + ".weight", {lp.cout}, c.use_bn != 0);
        lp.bn_b = add_vec(t, "bn1D_layers." + s + ".bias", {lp.cout}, c.use_bn != 0);
        if (lp.residual) {
            const std::string q = "residual_conv_layers." + std::to_string(l + 1 - c.residual_start);     // model.py:760
            lp.res_w = add_vec(t, q + ".weight", {lp.cout, lp.cout, 1, 1});
            lp.res_b = add_vec(t, q + ".bias", {lp.cout});
        }
        if (H > 0) {
            lp.bot_w = add_vec(t, "conv1D_bottleneck_layers." + s + ".weight", {H, lp.cout, 1, 1});
            lp.bot_b = add_vec(t, "conv1D_bottleneck_layers." + s + ".bias", {H});
            lp.cmp_w = add_vec(t, "conv1D_compression_layers." + s + ".weight", {H, H, 1, L});
            lp.cmp_b = add_vec(t, "conv1D_compression_layers." + s + ".bias", {H});
        }
    }
    t->p_fc0w = add_param(t, "fc.0.weight", {t->n0, t->F}, t->n0, t->F, t->F_stride);
    t->p_fc0b = add_vec(t, "fc.0.bias", {t->n0});
    t->p_fc1w = add_param(t, "fc.1.weight", {t->n1, t->n0}, t->n1, t->n0, t->n0_stride);
    t->p_fc1b = add_vec(t, "fc.1.bias", {t->n1});
    static const struct { const char* name; int n; } heads[] = {{"fcHidden2BinTarget", 2}, {"fcHidden2VT", 3}, {"fcHidden2AF", 1},
                                                                 {"fcHidden2Coverage", 1}, {"fcHidden2VB", VOCAB}, {"fcHidden2VR", VOCAB}};
    // the six heads are ONE stored matrix [27][fc1] and ONE bias vector [27]; their state-dict tensors are windows of it
    t->p_hw = add_vec(t, "heads.weight", {NHEAD, t->n1});
    t->p_hb = add_vec(t, "heads.bias", {NHEAD});
    t->views.erase("heads.weight");
    t->views.erase("heads.bias");
    int row = 0;
    for (auto& hd : heads) {
        View w; w.param = t->p_hw; w.off = (int64_t)row * t->n1; w.rows = 1; w.cols = (int64_t)hd.n * t->n1; w.shape = {hd.n, t->n1};
        View b; b.param = t->p_hb; b.off = row; b.rows = 1; b.cols = hd.n; b.shape = {hd.n};
        t->views[std::string(hd.name) + ".weight"] = w;
        t->views[std::string(hd.name) + ".bias"] = b;
        row += hd.n;
    }
    *out = t;
    return DAN_OK;
}

int dan_train_set_tensor(dan_trainer_t* t, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
    if (!t || !name || !data || ndim < 0 || ndim > 8 || (ndim > 0 && !shape)) return failt(t, DAN_ERR_INVALID_ARG, "dan_train_set_tensor: bad argument");
    if (t->finalized) return failt(t, DAN_ERR_STATE, "dan_train_set_tensor('%s') after dan_train_finalize (use dan_train_put_tensor)", name);
    std::vector<int64_t> shp(shape, shape + ndim);
    int64_t n = 1;
    for (auto s : shp) n *= s;
    const std::string nm(name);
    std::vector<int64_t> want;
    auto it = t->views.find(nm);
    if (it != t->views.end()) want = it->second.shape;
    else if (nm == "pe") want = {1, t->cfg.length, EMBED};
    else if (nm.rfind("bn1D_layers.", 0) == 0 && (nm.find("running_mean") != std::string::npos || nm.find("running_var") != std::string::npos)) {
        const int l = atoi(nm.c_str() + 12);
        if (l < 0 || l >= t->cfg.layers) return failt(t, DAN_ERR_INVALID_ARG, "tensor '%s': no such layer", name);
        want = {t->layers[l].cout};
    } else return failt(t, DAN_ERR_INVALID_ARG, "tensor '%s' is not part of this configuration's state", name);
    if (shp != want) {
        std::string got, exp;
        for (auto s : shp) got += std::to_string(s) + ",";
        for (auto s : want) exp += std::to_string(s) + ",";
        return failt(t, DAN_ERR_SHAPE, "tensor '%s' has shape (%s) but the configuration needs (%s)", name, got.c_str(), exp.c_str());
    }
    t->init[nm].assign(data, data + n);
    return DAN_OK;
}

int dan_train_finalize(dan_trainer_t* t) {
    if (!t) return DAN_ERR_INVALID_ARG;
    if (t->finalized) return failt(t, DAN_ERR_STATE, "dan_train_finalize called twice");
    const dan_config& c = t->cfg;
    HIPT(t, hipSetDevice(c.device_id));
    const int L = c.length, R = c.reads, H = c.bottleneck, NL = c.layers, B = t->max_batch;
    const size_t rows = (size_t)B * R;
    int rc;
    // ---- flat parameter / gradient / moment buffers
    if ((rc = talloc(t, &t->P, t->n_flat)) || (rc = talloc(t, &t->G, t->n_flat)) || (rc = talloc(t, &t->M, t->n_flat)) || (rc = talloc(t, &t->V, t->n_flat))) return rc;
    for (const auto& kv : t->views) {
        const View& v = kv.second;
        const Param& p = t->params[v.param];
        auto it = t->init.find(kv.first);
        if (it == t->init.end()) return failt(t, DAN_ERR_MISSING_TENSOR, "missing tensor '%s'", kv.first.c_str());
        HIPT(t, hipMemcpy2D(t->P + p.off + v.off, p.ld * sizeof(float), it->second.data(), v.cols * sizeof(float), v.cols * sizeof(float), v.rows, hipMemcpyHostToDevice));
    }
    auto pe = t->init.find("pe");
    if (pe == t->init.end()) return failt(t, DAN_ERR_MISSING_TENSOR, "missing tensor 'pe'");
    if ((rc = talloc(t, &t->d_pe, (size_t)L * EMBED))) return rc;
    HIPT(t, hipMemcpy(t->d_pe, pe->second.data(), (size_t)L * EMBED * sizeof(float), hipMemcpyHostToDevice));
    t->pe_host = pe->second;
    if ((rc = talloc(t, &t->d_rmean, (size_t)NL * CPAD)) || (rc = talloc(t, &t->d_rvar, (size_t)NL * CPAD))) return rc;
    for (int l = 0; l < NL; ++l) {
        const std::string q = "bn1D_layers." + std::to_string(l);
        auto m = t->init.find(q + ".running_mean"), v = t->init.find(q + ".running_var");
        if (m == t->init.end() || v == t->init.end()) return failt(t, DAN_ERR_MISSING_TENSOR, "missing tensor '%s.running_mean/var'", q.c_str());
        HIPT(t, hipMemcpy(t->d_rmean + (size_t)l * CPAD, m->second.data(), m->second.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPT(t, hipMemcpy(t->d_rvar + (size_t)l * CPAD, v->second.data(), v->second.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    t->init.clear();
    // ---- layer-1 channel maps (canonical 48-channel order of the encode stage; model.py:517,543,558,625)
    {
        std::vector<int> canon;
        for (int i = 0; i < 2 * EMBED; ++i) canon.push_back(i);
        if (c.use_q) canon.push_back(40);
        if (c.use_strand) canon.push_back(41);
        if (c.use_mask) { canon.push_back(42); canon.push_back(43); canon.push_back(44); }
        std::vector<int> inv(CIN0, -1);
        for (size_t i = 0; i < canon.size(); ++i) inv[canon[i]] = (int)i;
        if ((rc = talloc(t, &t->d_inv, CIN0)) || (rc = talloc(t, &t->d_canon, canon.size()))) return rc;
        HIPT(t, hipMemcpy(t->d_inv, inv.data(), CIN0 * sizeof(int), hipMemcpyHostToDevice));
        HIPT(t, hipMemcpy(t->d_canon, canon.data(), canon.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    // ---- packed weights and padded constants
    t->pk_conv_f.resize(NL); t->pk_conv_d.resize(NL); t->pk_res_f.assign(NL, nullptr); t->pk_res_d.assign(NL, nullptr);
    t->pk_bot_f.assign(NL, nullptr); t->pk_bot_d.assign(NL, nullptr);
    t->pk_wino_f.assign(NL, nullptr); t->pk_wino_d.assign(NL, nullptr); t->wino_layer.assign(NL, 0);
    if ((rc = talloc(t, &t->d_wino_u, (size_t)CPAD * CPAD * 4))) return rc;
    for (int l = 0; l < NL; ++l) {
        const LayerP& lp = t->layers[l];
        if ((rc = talloc(t, &t->pk_conv_f[l], (size_t)3 * lp.kg * KGC * 256)) || (rc = talloc(t, &t->pk_conv_d[l], (size_t)3 * KGC * KGC * 256))) return rc;
        // 3 taps at dilation 2 on 128-channel rows CAN run in Winograd F(2,3) form (conv_algo 2: forward conv and data gradient,
        // +5 % step rate).  It is opt-in for training: its rounding noise is a few times the direct form's, and in TRAINING a
        // pre-activation that lands on the other side of zero flips a ReLU mask and moves a gradient by a whole element --
        // at production width that is up to 1.7e-2 of a tensor's max against the float64 oracle where the direct form stays at
        // 2e-6 (tools/train_diff.py), although both pass the reference fixtures.  Parity first: the default is the direct form.
        if (l > 0 && lp.dil == 2 && c.conv_algo == 2) {
            t->wino_layer[l] = 1;
            if ((rc = talloc(t, &t->pk_wino_f[l], (size_t)4 * KGC * KGC * 256)) || (rc = talloc(t, &t->pk_wino_d[l], (size_t)4 * KGC * KGC * 256))) return rc;
        }
        if (lp.residual && ((rc = talloc(t, &t->pk_res_f[l], (size_t)KGC * KGC * 256)) || (rc = talloc(t, &t->pk_res_d[l], (size_t)KGC * KGC * 256)))) return rc;
        if (H > 0 && ((rc = talloc(t, &t->pk_bot_f[l], (size_t)KGC * 2 * 256)) || (rc = talloc(t, &t->pk_bot_d[l], (size_t)2 * KGC * 256)))) return rc;
    }
    if (H > 0) {
        if ((rc = talloc(t, &t->d_wc_pk, (size_t)NL * L * 2 * 2 * 256)) || (rc = talloc(t, &t->d_wct, (size_t)NL * L * HPAD * HPAD)) ||
            (rc = talloc(t, &t->d_bc_pad, (size_t)NL * HPAD))) return rc;
    }
    if ((rc = talloc(t, &t->d_bias, (size_t)NL * 3 * CPAD)) || (rc = talloc(t, &t->d_coef_f, (size_t)NL * 3 * CPAD)) ||
        (rc = talloc(t, &t->d_coef_b, (size_t)3 * CPAD)) || (rc = talloc(t, &t->d_smean, (size_t)NL * CPAD)) || (rc = talloc(t, &t->d_sinv, (size_t)NL * CPAD))) return rc;
    if (!c.use_bn) {                                         // no BatchNorm: the "normalised" activation is the activation itself
        std::vector<float> id((size_t)NL * 3 * CPAD, 0.f);
        for (int l = 0; l < NL; ++l)
            for (int ch = 0; ch < t->layers[l].cout; ++ch) id[(size_t)l * 3 * CPAD + ch] = 1.f;
        HIPT(t, hipMemcpy(t->d_coef_f, id.data(), id.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    // ---- inputs, targets, masks
    const size_t in_site = (size_t)3 * R * L + 3 * L;
    if ((rc = talloc(t, &t->d_in, (size_t)B * in_site)) || (rc = talloc(t, &t->d_tg8, (size_t)B * 4)) || (rc = talloc(t, &t->d_tgf, (size_t)B * 3))) return rc;
    HIPT(t, hipHostMalloc((void**)&t->h_stage, (size_t)B * (in_site + 4 + 3 * sizeof(float)), hipHostMallocDefault));
    if ((rc = talloc(t, &t->d_mask[0], (size_t)B * t->F)) || (rc = talloc(t, &t->d_mask[1], (size_t)B * t->n0)) || (rc = talloc(t, &t->d_mask[2], (size_t)B * t->n1))) return rc;
    // ---- activations
    const size_t rowf = (size_t)L * CPAD;
    t->d_a.resize(NL); t->d_x.resize(NL); t->d_pool.assign(NL + 1, nullptr);
    // A BatchNorm output x_l that feeds nothing but the next layer's convolution and its own highway branch is never written: the
    // pointwise launch behind layer l still forms it (for h_l) but stores only h_l, and its three consumers -- the next conv
    // launch, that layer's weight gradient, the bottleneck's weight gradient -- form it again from a_l with the same three
    // per-channel constants as they stage their operands (one multiply-add per element they load anyway): 658 MB less HBM write
    // per layer at 64 sites.  Not for a residual layer (x_l comes out of a 1x1 GEMM), a pooled one (the read mean reads x_l)
    // or the last one (the final pools read it).
    t->lazy_x.assign(NL, 0);
    for (int l = 0; l + 1 < NL; ++l) t->lazy_x[l] = !t->layers[l].residual && !pool_after(c, l + 1) && H > 0;
    for (int l = 0; l < NL; ++l) {
        if ((rc = talloc(t, &t->d_a[l], rows * rowf, false))) return rc;
        t->d_x[l] = nullptr;
        if (!t->lazy_x[l] && (rc = talloc(t, &t->d_x[l], rows * rowf, false))) return rc;
        if (pool_after(c, l + 1) && (rc = talloc(t, &t->d_pool[l + 1], (size_t)B * rowf, false))) return rc;
    }
    if (H > 0 && ((rc = talloc(t, &t->d_h, (size_t)NL * rows * L * HPAD, false)) || (rc = talloc(t, &t->d_dh, (size_t)NL * rows * L * HPAD, false)))) return rc;
    // entries of a statistics pass: one per read (whole-read form), TWO per read (train_rowh_kernel's half-read units, whatever
    // the window length: at L < 128 that is more than the tiles), or one per 64-position tile (pointwise form)
    const size_t stat_max = std::max<size_t>(2 * rows, (rows * L + 63) / 64);
    t->stat_entries_max = (int)stat_max;
    if ((rc = talloc(t, &t->d_stats, stat_max * 2 * CPAD)) || (rc = talloc(t, &t->d_bp, (stat_max / 32 + 2) * 2 * CPAD + (size_t)(t->n_flat / (256 * 64) + 2)))) return rc;
    if ((rc = talloc(t, &t->d_feat, (size_t)B * t->F_stride)) || (rc = talloc(t, &t->d_featd, (size_t)B * t->F_stride)) ||
        (rc = talloc(t, &t->d_hid0, (size_t)B * t->n0_stride)) || (rc = talloc(t, &t->d_hid0d, (size_t)B * t->n0_stride)) ||
        (rc = talloc(t, &t->d_hid1, (size_t)B * t->n1_stride)) || (rc = talloc(t, &t->d_hid1d, (size_t)B * t->n1_stride)) ||
        (rc = talloc(t, &t->d_logits, (size_t)B * NHEAD)) || (rc = talloc(t, &t->d_dlogits, (size_t)B * NHEAD)) ||
        (rc = talloc(t, &t->d_losses, 8)) || (rc = talloc(t, &t->d_site_terms, (size_t)B * 8)) || (rc = talloc(t, &t->d_close, (size_t)B * 2))) return rc;
    if ((rc = talloc(t, &t->d_dhid1d, (size_t)B * t->n1_stride)) || (rc = talloc(t, &t->d_dhid1, (size_t)B * t->n1_stride)) ||
        (rc = talloc(t, &t->d_dhid0d, (size_t)B * t->n0_stride)) || (rc = talloc(t, &t->d_dhid0, (size_t)B * t->n0_stride)) ||
        (rc = talloc(t, &t->d_dfeatd, (size_t)B * t->F_stride)) || (rc = talloc(t, &t->d_dfeat, (size_t)B * t->F_stride))) return rc;
    if ((rc = talloc(t, &t->d_du, rows * rowf, false)) || (rc = talloc(t, &t->d_g[0], rows * rowf, false)) || (rc = talloc(t, &t->d_g[1], rows * rowf, false)) ||
        (rc = talloc(t, &t->d_dn, rows * rowf, false)) || (rc = talloc(t, &t->d_dpool, (size_t)B * rowf, false))) return rc;
    if ((rc = talloc(t, &t->d_partial, (size_t)TRAIN_PARTIAL_WGS * 3 * CPAD * CPAD, false)) || (rc = talloc(t, &t->d_bias_partial, (size_t)2 * TRAIN_PARTIAL_WGS * CPAD, false)) ||
        (rc = talloc(t, &t->d_hw_partial, (size_t)8 * HPAD * L * HPAD + 64 * HPAD, false)) ||
        (rc = talloc(t, &t->d_l0tot, (size_t)L0_BINS_TOTALS)) || (rc = talloc(t, &t->d_l0tab, l0_tab_floats(L))) ||
        (rc = talloc(t, &t->d_clip, 4))) return rc;
    if (H > 0) {
        t->split_hw_floats = (long long)8 * HPAD * L * HPAD;
        if ((rc = talloc(t, &t->d_dhw, (size_t)NL * rows * HPAD, false)) || (rc = talloc(t, &t->d_split_hw, (size_t)t->split_hw_floats, false))) return rc;
    }
    // (split-K partials of the FC GEMMs; at <= 16 sites the FC1 forms of fc_skinny_*_kernel: one partial per 1024-float slab of
    // the feature row forward, up to 12 row groups of the weight matrix in the data gradient)
    const long long skinny = (long long)std::min(B, 16) * std::max((t->F_stride + 1023) / 1024 * (long long)t->n0, 12 * (long long)t->F_stride);
    t->split_ws_floats = std::max((long long)32 * B * std::max(t->n0, t->n1), skinny);
    if ((rc = talloc(t, &t->d_split_ws, (size_t)t->split_ws_floats, false))) return rc;
    if ((rc = build_pack_jobs(t))) return rc;
    HIPT(t, hipDeviceSynchronize());
    HIPT(t, hipEventCreateWithFlags(&t->ev_tail, hipEventDisableTiming));
    t->finalized = true;
    return DAN_OK;
}

void dan_train_destroy(dan_trainer_t* t) {
    if (!t) return;
    (void)hipSetDevice(t->cfg.device_id);
    (void)hipDeviceSynchronize();
    for (void* p : t->allocs) (void)hipFree(p);
    if (t->ev_tail) (void)hipEventDestroy(t->ev_tail);
    if (t->h_stage) (void)hipHostFree(t->h_stage);
    delete t;
}

}  // extern "C"

namespace {

// weights -> the fragment orders the kernels consume, biases -> padded copies; run at the start of every step
// The same re-packing as refresh_packed below as a job table for ONE launch (built at finalize: every pointer is fixed by then).
// The Winograd transforms (opt-in form) go through a scratch buffer and stay separate launches.
int build_pack_jobs(dan_trainer* t) {
    const dan_config& c = t->cfg;
    const int L = c.length, H = c.bottleneck;
    std::vector<PackJob> jobs;
    auto frag = [&](float* dst, const float* src, int taps, int kg, int tiles, int n_out, int n_in, long long so, long long sc, long long st,
                    int flip, const int* omap, const int* cmap) {
        PackJob q{}; q.type = 0; q.dst = dst; q.src = src; q.i[0] = taps; q.i[1] = kg; q.i[2] = tiles; q.i[3] = n_out; q.i[4] = n_in; q.i[5] = flip;
        q.l[0] = so; q.l[1] = sc; q.l[2] = st; q.omap = omap; q.cmap = cmap; jobs.push_back(q);
    };
    auto pad = [&](float* dst, const float* src, int n, int n_pad) { PackJob q{}; q.type = 1; q.dst = dst; q.src = src; q.i[0] = n; q.i[1] = n_pad; jobs.push_back(q); };
    auto wc = [&](int type, float* dst, const float* src) { PackJob q{}; q.type = type; q.dst = dst; q.src = src; q.i[0] = H; q.i[1] = L; jobs.push_back(q); };
    for (int l = 0; l < c.layers; ++l) {
        const LayerP& lp = t->layers[l];
        const float* W = pp(t, lp.conv_w);
        const int* cmap = (l == 0) ? t->d_inv : nullptr;
        const int cin_canon = (l == 0) ? CIN0 : lp.cin;
        frag(t->pk_conv_f[l], W, 3, lp.kg, KGC, lp.cout, cin_canon, (long long)lp.cin * 3, 3, 1, 0, nullptr, cmap);
        frag(t->pk_conv_d[l], W, 3, KGC, KGC, cin_canon, lp.cout, 3, (long long)lp.cin * 3, 1, 1, cmap, nullptr);
        float* b = t->d_bias + (size_t)l * 3 * CPAD;
        pad(b, pp(t, lp.conv_b), lp.cout, CPAD);
        if (lp.residual) {
            const float* Wr = pp(t, lp.res_w);
            frag(t->pk_res_f[l], Wr, 1, KGC, KGC, lp.cout, lp.cout, lp.cout, 1, 0, 0, nullptr, nullptr);
            frag(t->pk_res_d[l], Wr, 1, KGC, KGC, lp.cout, lp.cout, 1, lp.cout, 0, 0, nullptr, nullptr);
            pad(b + CPAD, pp(t, lp.res_b), lp.cout, CPAD);
        }
        if (H > 0) {
            const float* Wb = pp(t, lp.bot_w);
            frag(t->pk_bot_f[l], Wb, 1, KGC, 2, H, lp.cout, lp.cout, 1, 0, 0, nullptr, nullptr);
            frag(t->pk_bot_d[l], Wb, 1, 2, KGC, lp.cout, H, 1, lp.cout, 0, 0, nullptr, nullptr);
            pad(b + 2 * CPAD, pp(t, lp.bot_b), H, HPAD);
            wc(2, t->d_wc_pk + (size_t)l * L * 2 * 2 * 256, pp(t, lp.cmp_w));
            wc(3, t->d_wct + (size_t)l * L * HPAD * HPAD, pp(t, lp.cmp_w));
            pad(t->d_bc_pad + (size_t)l * HPAD, pp(t, lp.cmp_b), H, HPAD);
        }
    }
    if (H > 0) {
        std::vector<long long> off(c.layers);
        for (int l = 0; l < c.layers; ++l) off[l] = t->params[t->layers[l].cmp_b].off;
        int rc2 = talloc(t, &t->d_cmpb_off, off.size(), false);
        if (rc2) return rc2;
        HIPT(t, hipMemcpy(t->d_cmpb_off, off.data(), off.size() * sizeof(long long), hipMemcpyHostToDevice));
        for (int l = 0; l < c.layers; ++l) off[l] = t->params[t->layers[l].cmp_w].off;
        if ((rc2 = talloc(t, &t->d_cmpw_off, off.size(), false))) return rc2;
        HIPT(t, hipMemcpy(t->d_cmpw_off, off.data(), off.size() * sizeof(long long), hipMemcpyHostToDevice));
    }
    int blocks = 0;
    for (PackJob& q : jobs) { q.first_block = blocks; blocks += pack_job_blocks(q); }
    t->n_pack_jobs = (int)jobs.size(); t->n_pack_blocks = blocks;
    int rc = talloc(t, (uint8_t**)&t->d_pack_jobs, jobs.size() * sizeof(PackJob), false);
    if (rc) return rc;
    HIPT(t, hipMemcpy(t->d_pack_jobs, jobs.data(), jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice));
    return DAN_OK;
}

void refresh_packed(dan_trainer* t, hipStream_t s) {
    const dan_config& c = t->cfg;
    const int L = c.length, H = c.bottleneck;
    if (t->d_pack_jobs) {
        launch_pack_jobs(t->d_pack_jobs, t->n_pack_jobs, t->n_pack_blocks, s);
        for (int l = 0; l < c.layers; ++l) {
            if (!t->wino_layer[l]) continue;
            const LayerP& lp = t->layers[l];
            const float* W = pp(t, lp.conv_w);
            launch_wino_u(t->d_wino_u, W, lp.cout, lp.cin, (long long)lp.cin * 3, 3, 0, s);
            launch_pack_frag(t->pk_wino_f[l], t->d_wino_u, 4, KGC, KGC, lp.cout, lp.cin, (long long)lp.cin * 4, 4, 1, 0, nullptr, nullptr, s);
            launch_wino_u(t->d_wino_u, W, lp.cin, lp.cout, 3, (long long)lp.cin * 3, 1, s);
            launch_pack_frag(t->pk_wino_d[l], t->d_wino_u, 4, KGC, KGC, lp.cin, lp.cout, (long long)lp.cout * 4, 4, 1, 0, nullptr, nullptr, s);
        }
        return;
    }
    for (int l = 0; l < c.layers; ++l) {
        const LayerP& lp = t->layers[l];
        const float* W = pp(t, lp.conv_w);
        const int* cmap = (l == 0) ? t->d_inv : nullptr;
        const int cin_canon = (l == 0) ? CIN0 : lp.cin;
        // forward: W(o, c, tap) = W[o][c][tap]
        launch_pack_frag(t->pk_conv_f[l], W, 3, lp.kg, KGC, lp.cout, cin_canon, (long long)lp.cin * 3, 3, 1, 0, nullptr, cmap, s);
        // data gradient: du[p][c] = sum_{o,t'} W[o][c][2 - t'] dz[p + (t' - 1) d][o]: outputs c (canonical for layer 1), k = o
        launch_pack_frag(t->pk_conv_d[l], W, 3, KGC, KGC, cin_canon, lp.cout, 3, (long long)lp.cin * 3, 1, 1, cmap, nullptr, s);
        if (t->wino_layer[l]) {
            // forward: U(o, c);  data gradient: outputs c, inputs o, flipped taps
            launch_wino_u(t->d_wino_u, W, lp.cout, lp.cin, (long long)lp.cin * 3, 3, 0, s);
            launch_pack_frag(t->pk_wino_f[l], t->d_wino_u, 4, KGC, KGC, lp.cout, lp.cin, (long long)lp.cin * 4, 4, 1, 0, nullptr, nullptr, s);
            launch_wino_u(t->d_wino_u, W, lp.cin, lp.cout, 3, (long long)lp.cin * 3, 1, s);
            launch_pack_frag(t->pk_wino_d[l], t->d_wino_u, 4, KGC, KGC, lp.cin, lp.cout, (long long)lp.cout * 4, 4, 1, 0, nullptr, nullptr, s);
        }
        float* b = t->d_bias + (size_t)l * 3 * CPAD;
        launch_pad_copy(b, pp(t, lp.conv_b), lp.cout, CPAD, s);
        if (lp.residual) {
            const float* Wr = pp(t, lp.res_w);
            launch_pack_frag(t->pk_res_f[l], Wr, 1, KGC, KGC, lp.cout, lp.cout, lp.cout, 1, 0, 0, nullptr, nullptr, s);
            launch_pack_frag(t->pk_res_d[l], Wr, 1, KGC, KGC, lp.cout, lp.cout, 1, lp.cout, 0, 0, nullptr, nullptr, s);
            launch_pad_copy(b + CPAD, pp(t, lp.res_b), lp.cout, CPAD, s);
        }
        if (H > 0) {
            const float* Wb = pp(t, lp.bot_w);
            launch_pack_frag(t->pk_bot_f[l], Wb, 1, KGC, 2, H, lp.cout, lp.cout, 1, 0, 0, nullptr, nullptr, s);
            launch_pack_frag(t->pk_bot_d[l], Wb, 1, 2, KGC, lp.cout, H, 1, lp.cout, 0, 0, nullptr, nullptr, s);      // out = c, k = o2
            launch_pad_copy(b + 2 * CPAD, pp(t, lp.bot_b), H, HPAD, s);
            launch_pack_wc(t->d_wc_pk + (size_t)l * L * 2 * 2 * 256, pp(t, lp.cmp_w), H, L, s);
            launch_pack_wct(t->d_wct + (size_t)l * L * HPAD * HPAD, pp(t, lp.cmp_w), H, L, s);
            launch_pad_copy(t->d_bc_pad + (size_t)l * HPAD, pp(t, lp.cmp_b), H, HPAD, s);
        }
    }
}

void fill_encode(RowArgs& a, const dan_trainer* t, int B) {
    const size_t rl = (size_t)t->cfg.reads * t->cfg.length, L = t->cfg.length;
    const uint8_t* d = t->d_in;
    a.reads = d; a.qual = d + B * rl; a.strand = d + 2 * B * rl; a.ref = d + 3 * B * rl; a.ref_mask = a.ref + B * L; a.var_mask = a.ref_mask + B * L;
    a.emb = t->P + t->params[t->p_emb].off; a.pe = t->d_pe;
}

}  // namespace

extern "C" {

int dan_train_backward_begin(dan_trainer_t* t, const uint8_t* reads, const uint8_t* qual, const uint8_t* strand, const uint8_t* ref,
                             const uint8_t* ref_mask, const uint8_t* var_mask, int64_t n_sites, const dan_train_targets* tg,
                             const uint8_t* const* dropout_masks, uint64_t seed) {
    if (!t) return DAN_ERR_INVALID_ARG;
    if (!t->finalized) return failt(t, DAN_ERR_STATE, "dan_train_backward before dan_train_finalize");
    if (t->pending) return failt(t, DAN_ERR_STATE, "dan_train_backward_begin: the previous step has not been ended (dan_train_backward_end)");
    if (n_sites < 1 || n_sites > t->max_batch) return failt(t, DAN_ERR_INVALID_ARG, "a training batch holds 1..%d sites, got %lld", t->max_batch, (long long)n_sites);
    if (!reads || !qual || !strand || !ref || !ref_mask || !var_mask) return failt(t, DAN_ERR_INVALID_ARG, "null input plane");
    if (!tg || !tg->label || !tg->var_type || !tg->allele_freq || !tg->coverage || !tg->var_base_enum || !tg->var_ref_enum || !tg->weight)
        return failt(t, DAN_ERR_INVALID_ARG, "null target array");
    const dan_config& c = t->cfg;
    const dan_train_hyper& hp = t->hp;
    HIPT(t, hipSetDevice(c.device_id));
    hipStream_t s = nullptr;
    const int B = (int)n_sites, L = c.length, R = c.reads, H = c.bottleneck, NL = c.layers;
    const int n_rows = B * R;
    const size_t rl = (size_t)R * L;
    const double n_pos = (double)n_rows * L;
    t->last_B = B;
    // ---- inputs
    {   // the seventeen caller arrays are gathered in the pinned mirror (same layout as on the device) and go up in three
        // stream-ordered copies; the mirror is free again: the previous step's copies completed before its dan_train_backward_end returned
        uint8_t* h = t->h_stage;
        size_t o = 0;
        auto stage = [&](const void* src, size_t n) { memcpy(h + o, src, n); o += n; };
        stage(reads, B * rl); stage(qual, B * rl); stage(strand, B * rl);
        stage(ref, (size_t)B * L); stage(ref_mask, (size_t)B * L); stage(var_mask, (size_t)B * L);
        const size_t n_in = o;
        stage(tg->label, B); stage(tg->var_type, B); stage(tg->var_base_enum, B); stage(tg->var_ref_enum, B);
        const size_t n_tg8 = o - n_in;
        stage(tg->allele_freq, B * sizeof(float)); stage(tg->coverage, B * sizeof(float)); stage(tg->weight, B * sizeof(float));
        HIPT(t, hipMemcpyAsync(t->d_in, h, n_in, hipMemcpyHostToDevice, s));
        HIPT(t, hipMemcpyAsync(t->d_tg8, h + n_in, n_tg8, hipMemcpyHostToDevice, s));
        HIPT(t, hipMemcpyAsync(t->d_tgf, h + n_in + n_tg8, o - n_in - n_tg8, hipMemcpyHostToDevice, s));
    }
    int rc = DAN_OK;
    const int stat_cap = t->stat_entries_max;
    const bool drop = hp.dropout > 0.f;
    const float dscale = drop ? 1.f / (1.f - hp.dropout) : 1.f;
    const int64_t mcols[3] = {t->F, t->n0, t->n1};
    if (drop) {
        for (int i = 0; i < 3; ++i) {
            if (dropout_masks && dropout_masks[i]) HIPT(t, hipMemcpy(t->d_mask[i], dropout_masks[i], (size_t)B * mcols[i], hipMemcpyHostToDevice));
            else if (dropout_masks) return failt(t, DAN_ERR_INVALID_ARG, "dropout mask %d is null", i);
            else launch_dropout_mask(t->d_mask[i], (long long)B * mcols[i], hp.dropout, seed, (unsigned long long)t->step * 3 + i, s);
        }
    }
    const uint8_t* mk[3] = {drop ? t->d_mask[0] : nullptr, drop ? t->d_mask[1] : nullptr, drop ? t->d_mask[2] : nullptr};
    HIPT(t, hipMemsetAsync(t->G, 0, (size_t)t->n_flat * sizeof(float), s));
    refresh_packed(t, s);

    int stat_entries = 0;                                    // entries of d_stats the last statistics-producing launch wrote
    // =========================================== forward (train mode) ===========================================
    for (int l = 0; l < NL; ++l) {
        const LayerP& lp = t->layers[l];
        const float* bias = t->d_bias + (size_t)l * 3 * CPAD;
        float* coef_f = t->d_coef_f + (size_t)l * 3 * CPAD;
        {   // a_l = relu(conv(u_l) + b), u_l = x_{l-1} (+ read-mean of x_{l-1} when layer l-1 pools; model.py:742,749)
            RowArgs a{};
            a.R = R; a.L = L;
            if (l == 0) {
                // layer 1 by table: the tables from this step's weights, then the walk (dan_train.hip)
                fill_encode(a, t, B);
                launch_l0_train_tables(pp(t, lp.conv_w), t->d_inv, a.emb, a.pe, L, lp.cout, lp.cin, t->d_l0tab, s);
                stat_entries = launch_l0_train_forward(a, t->d_l0tab, bias, n_rows, t->d_a[l], c.use_bn ? t->d_stats : nullptr, s);
                if (stat_entries > stat_cap) return failt(t, DAN_ERR_STATE, "layer 1's forward writes %d statistics entries, the buffer holds %d", stat_entries, stat_cap);
            } else {
            if (t->lazy_x[l - 1]) { a.mode = 1; a.src1 = t->d_a[l - 1]; a.s1_stride = CPAD; a.coef = t->d_coef_f + (size_t)(l - 1) * 3 * CPAD; }
            else { a.mode = 1; a.src1 = t->d_x[l - 1]; a.s1_stride = CPAD; a.pool_in = t->d_pool[l]; }
            a.w1 = t->pk_conv_f[l]; a.taps = 3; a.kg = lp.kg; a.dil = lp.dil;
            if (t->wino_layer[l]) { a.w1 = t->pk_wino_f[l]; a.wino = 1; }
            a.bias1 = bias; a.relu_out = 1; a.out1 = t->d_a[l];
            a.stats = c.use_bn ? t->d_stats : nullptr;
            stat_entries = launch_train_row(a, n_rows, s, stat_cap);
            if ((rc = check_row_launch(t, stat_entries, "conv forward", l))) return rc;
            }
        }
        if (c.use_bn) {                                      // batch statistics over (B, R, L) per channel (model.py:750-751, train mode)
            int nb = 0;
            launch_stats_partial(t->d_stats, stat_entries, t->d_bp, &nb, s);
            launch_bn_forward_finalize(t->d_bp, nb, n_pos, pp(t, lp.bn_g), pp(t, lp.bn_b), lp.cout, coef_f, t->d_smean + (size_t)l * CPAD,
                                       t->d_sinv + (size_t)l * CPAD, t->d_rmean + (size_t)l * CPAD, t->d_rvar + (size_t)l * CPAD, s);
        }
        {   // x_l = [W_r] bn(a_l) [+ b_r + x_{l-1}]  (model.py:753-761);  h_l = relu(W_b x_l + b_b)  (model.py:774)
            RowArgs a{};
            a.R = R; a.L = L; a.mode = 1; a.src1 = t->d_a[l]; a.s1_stride = CPAD; a.coef = coef_f;
            if (lp.residual) { a.w1 = t->pk_res_f[l]; a.taps = 1; a.kg = KGC; a.dil = 0; a.bias1 = bias + CPAD; a.add1 = t->d_x[l - 1]; }
            if (lp.residual && t->lazy_x[l - 1]) { a.add1 = t->d_a[l - 1]; a.add1_coef = t->d_coef_f + (size_t)(l - 1) * 3 * CPAD; }
            a.out1 = t->d_x[l];                              // (nullptr for a lazy layer: the launch then only writes h_l)
            if (H > 0) { a.w2 = t->pk_bot_f[l]; a.bias2 = bias + 2 * CPAD; a.out2 = t->d_h + (size_t)l * n_rows * L * HPAD; }
            if ((rc = check_row_launch(t, launch_train_row(a, n_rows, s, stat_cap), "BatchNorm-apply / residual / bottleneck", l))) return rc;
        }
        if (pool_after(c, l + 1)) launch_read_mean(t->d_x[l], t->d_pool[l + 1], B, R, L, nullptr, s);
    }
    const long long h_layer = (long long)n_rows * L * HPAD;
    launch_final_pool(t->d_x[NL - 1], t->d_feat, t->F_stride, B, R, L, c.c_final, nullptr, s);
    if (H > 0)
        launch_highway(t->d_h, h_layer, t->d_wc_pk, (long long)L * 2 * 2 * 256, t->d_bc_pad, t->d_feat, t->F_stride, 2 * c.c_final * L, B, R, L, H, NL, nullptr, s);
    // FC stack: Dropout -> Linear -> ReLU -> Dropout -> Linear -> ReLU -> Dropout (model.py:369-377)
    launch_dropout(t->d_feat, mk[0], dscale, t->d_featd, B, t->F, t->F_stride, s);
    launch_gemm(t->d_featd, t->F_stride, 0, pp(t, t->p_fc0w), t->F_stride, 0, pp(t, t->p_fc0b), t->d_hid0, t->n0_stride, B, t->n0, (int)t->F_stride, 1, t->d_split_ws, t->split_ws_floats, s);
    launch_dropout(t->d_hid0, mk[1], dscale, t->d_hid0d, B, t->n0, t->n0_stride, s);
    launch_gemm(t->d_hid0d, t->n0_stride, 0, pp(t, t->p_fc1w), t->n0_stride, 0, pp(t, t->p_fc1b), t->d_hid1, t->n1_stride, B, t->n1, (int)t->n0_stride, 1, t->d_split_ws, t->split_ws_floats, s);
    launch_dropout(t->d_hid1, mk[2], dscale, t->d_hid1d, B, t->n1, t->n1_stride, s);
    {
        LossArgs a{};
        a.B = B; a.hid = t->n1; a.hid_stride = (int)t->n1_stride; a.hidden = t->d_hid1d; a.wh = pp(t, t->p_hw); a.bh = pp(t, t->p_hb);
        a.label = t->d_tg8; a.var_type = t->d_tg8 + B; a.var_base = t->d_tg8 + 2 * B; a.var_ref = t->d_tg8 + 3 * B;
        a.allele_freq = t->d_tgf; a.coverage = t->d_tgf + B; a.weight = t->d_tgf + 2 * B;
        a.label_smoothing = hp.label_smoothing; a.close_window = hp.close_match_window; a.focal_alpha = hp.focal_alpha; a.focal_gamma = hp.focal_gamma;
        a.fp_weight = hp.fp_train_weight; a.binary_weight = hp.binary_weight; a.aux_weight = hp.aux_weight;
        a.aux_bases_weight = hp.aux_bases_weight; a.aux_allele_weight = hp.aux_allele_weight;
        a.logits = t->d_logits; a.dlogits = t->d_dlogits; a.losses = t->d_losses; a.close = t->d_close; a.site_terms = t->d_site_terms;
        a.mean_sites = t->dp_mean_sites; a.ce_den[0] = t->dp_ce_den[0]; a.ce_den[1] = t->dp_ce_den[1];
        t->dp_mean_sites = 0.f; t->dp_ce_den[0] = t->dp_ce_den[1] = 0.f;
        launch_heads_loss(a, s);
    }

    // ================================================= backward =================================================
    // heads (their gradient tensors are contiguous in the flat buffer, like the weights)
    launch_heads_bwd(t->d_dlogits, t->d_hid1d, pp(t, t->p_hw), B, t->n1, (int)t->n1_stride, t->d_dhid1d, gp(t, t->p_hw), gp(t, t->p_hb), s);
    launch_dropout_relu_bwd(t->d_dhid1d, mk[2], dscale, t->d_hid1, 1, t->d_dhid1, B, t->n1, t->n1_stride, s);
    // FC2: gW1[n][k] = sum_b d1[b][n] hid0d[b][k];  d(hid0d) = d1 W1
    launch_gemm(t->d_dhid1, t->n1_stride, 1, t->d_hid0d, t->n0_stride, 1, nullptr, gp(t, t->p_fc1w), t->n0_stride, t->n1, t->n0, B, 0, t->d_split_ws, t->split_ws_floats, s);
    launch_colsum(t->d_dhid1, B, t->n1, t->n1_stride, gp(t, t->p_fc1b), s);
    launch_gemm(t->d_dhid1, t->n1_stride, 0, pp(t, t->p_fc1w), t->n0_stride, 1, nullptr, t->d_dhid0d, t->n0_stride, B, t->n0, t->n1, 0, t->d_split_ws, t->split_ws_floats, s);
    launch_dropout_relu_bwd(t->d_dhid0d, mk[1], dscale, t->d_hid0, 1, t->d_dhid0, B, t->n0, t->n0_stride, s);
    // FC1
    launch_gemm(t->d_dhid0, t->n0_stride, 1, t->d_featd, t->F_stride, 1, nullptr, gp(t, t->p_fc0w), t->F_stride, t->n0, t->F, B, 0, t->d_split_ws, t->split_ws_floats, s);
    launch_colsum(t->d_dhid0, B, t->n0, t->n0_stride, gp(t, t->p_fc0b), s);
    HIPT(t, hipEventRecord(t->ev_tail, s));                  // bucket 0 (FC stack + heads: the tail of the flat buffer) is final
    launch_gemm(t->d_dhid0, t->n0_stride, 0, pp(t, t->p_fc0w), t->F_stride, 1, nullptr, t->d_dfeatd, t->F_stride, B, t->F, t->n0, 0, t->d_split_ws, t->split_ws_floats, s);
    launch_dropout_relu_bwd(t->d_dfeatd, mk[0], dscale, nullptr, 0, t->d_dfeat, B, t->F, t->F_stride, s);
    // highway compression
    const int hw_off = 2 * c.c_final * L;
    if (H > 0) {
        // dhw = dfeat_hw * (feat_hw > 0) as [layer][row][HPAD]; then one launch for every layer's dh (a 165-MB write stream per layer at
        // 64 sites) and one for every layer's weight gradient (h read once, no split, straight into the torch layout)
        launch_highway_dhw(t->d_dfeat, t->d_feat, t->F_stride, hw_off, t->d_dhw, B, R, H, NL, s);
        launch_highway_dh(t->d_dhw, t->d_wct, t->d_dh, n_rows, L, NL, s);
        launch_highway_gwc(t->d_dhw, t->d_h, t->G, t->d_cmpw_off, n_rows, L, H, NL, s);
        // the compression biases' gradients of all layers (column sums of dhw) in two launches
        launch_highway_bias_grad_all(t->d_dfeat, t->d_feat, t->F_stride, hw_off, t->d_bias_partial, t->G, t->d_cmpb_off, B, R, H, NL, s);
    }
    // final max + mean pool (model.py:824-839)
    launch_final_pool_bwd(t->d_x[NL - 1], t->d_dfeat, t->F_stride, t->d_du, B, R, L, c.c_final, s);
    int cur = 0;                                             // d_g[cur] receives g_l; d_g[cur ^ 1] holds g_{l+1}
    bool fused_g = false;                                    // the previous iteration's data-gradient launch already wrote g_l into d_g[cur]
    for (int l = NL - 1; l >= 0; --l) {
        const LayerP& lp = t->layers[l];
        const bool next_res = (l + 1 < NL) && t->layers[l + 1].residual;
        const bool pooled = pool_after(c, l + 1);
        float* g = t->d_g[cur];
        const float* g_next = t->d_g[cur ^ 1];
        if (fused_g) {       // g_l (and its statistics) came out of layer l+1's data-gradient launch, below in the previous iteration
            fused_g = false;
        } else {   // g_l = du_{l+1} (or the pool gradient) [+ g_{l+1} through the residual skip] [+ mean_r du_{l+1}] [+ W_b^T (dh_l * (h_l > 0))]
            RowArgs a{};
            a.R = R; a.L = L; a.mode = 1;
            if (H > 0) {
                a.src1 = t->d_dh + (size_t)l * h_layer; a.s1_stride = HPAD; a.src2 = t->d_h + (size_t)l * h_layer; a.mask_src2 = 1;
                a.w1 = t->pk_bot_d[l]; a.taps = 1; a.kg = 2; a.dil = 0;
                a.add1 = t->d_du;
            } else {
                a.src1 = t->d_du; a.s1_stride = CPAD;
            }
            a.add2 = next_res ? g_next : nullptr;
            a.addb = pooled ? t->d_dpool : nullptr;
            a.out1 = g;
            if (!lp.residual) { a.stats = t->d_stats; a.stat_aux = t->d_a[l]; }
            const int e = launch_train_row(a, n_rows, s, stat_cap);
            if ((rc = check_row_launch(t, e, "g accumulation", l))) return rc;
            if (!lp.residual) stat_entries = e;
        }
        const float* dn = g;
        if (lp.residual) {   // x_l = W_r n_l + b_r + x_{l-1}:  dn_l = W_r^T g_l;  gW_r = g_l n_l^T;  g_{l-1} += g_l (next iteration's add2)
            RowArgs a{};
            a.R = R; a.L = L; a.mode = 1; a.src1 = g; a.s1_stride = CPAD;
            a.w1 = t->pk_res_d[l]; a.taps = 1; a.kg = KGC; a.dil = 0;
            a.out1 = t->d_dn; a.stats = t->d_stats; a.stat_aux = t->d_a[l];
            stat_entries = launch_train_row(a, n_rows, s, stat_cap);
            if ((rc = check_row_launch(t, stat_entries, "residual data gradient", l))) return rc;
            dn = t->d_dn;
            WgradArgs w{};
            w.R = R; w.L = L; w.n_rows = n_rows; w.a1 = g; w.a_stride = CPAD;
            w.b_mode = 1; w.b1 = t->d_a[l]; w.b_coef = t->d_coef_f + (size_t)l * 3 * CPAD;
            w.taps = 1; w.dil = 0; w.o_tiles = KGC; w.c_tiles = KGC; w.partial = t->d_partial; w.bias_partial = t->d_bias_partial;
            const int wgs = launch_train_wgrad(w, s);
            launch_wgrad_reduce(t->d_partial, t->d_bias_partial, wgs, 1, CPAD, CPAD, lp.cout, lp.cout, nullptr, gp(t, lp.res_w), gp(t, lp.res_b), s);
        }
        {   // BatchNorm backward coefficients from sum(dn), sum(dn * a)  (identity without BatchNorm)
            int nb = 0;
            launch_stats_partial(t->d_stats, stat_entries, t->d_bp, &nb, s);
            launch_bn_backward_coef(t->d_bp, nb, n_pos, pp(t, lp.bn_g), t->d_smean + (size_t)l * CPAD, t->d_sinv + (size_t)l * CPAD, lp.cout, c.use_bn,
                                    t->d_coef_b, gp(t, lp.bn_g), gp(t, lp.bn_b), s);
        }
        if (H > 0) {         // h_l = relu(W_b x_l + b_b):  gW_b = (dh_l * (h_l > 0)) x_l^T
            WgradArgs w{};
            w.R = R; w.L = L; w.n_rows = n_rows; w.a1 = t->d_dh + (size_t)l * h_layer; w.a_stride = HPAD; w.a2 = t->d_h + (size_t)l * h_layer; w.a_mask = 1;
            w.b_mode = 1; w.b1 = t->d_x[l];
            if (t->lazy_x[l]) { w.b1 = t->d_a[l]; w.b_coef = t->d_coef_f + (size_t)l * 3 * CPAD; }
            w.taps = 1; w.dil = 0; w.o_tiles = 2; w.c_tiles = KGC; w.partial = t->d_partial; w.bias_partial = t->d_bias_partial;
            const int wgs = launch_train_wgrad(w, s);
            launch_wgrad_reduce(t->d_partial, t->d_bias_partial, wgs, 1, 2 * 16, CPAD, H, lp.cout, nullptr, gp(t, lp.bot_w), gp(t, lp.bot_b), s);
        }
        if (l == 0) {
            // layer 1: its weight and bias gradients AND the embedding gradient from one pass over dz_1 (dan_train.hip: layer 1's
            // backward by bins) -- no encode-form GEMM, no data-gradient launch (its only reader was the embedding gradient)
            WgradArgs w{};
            w.R = R; w.L = L; w.n_rows = n_rows; w.a1 = dn; w.a_stride = CPAD; w.a2 = t->d_a[l]; w.a_coef = t->d_coef_b; w.a_mask = 1;
            RowArgs e{};
            fill_encode(e, t, B);
            w.reads = e.reads; w.qual = e.qual; w.strand = e.strand; w.ref = e.ref; w.ref_mask = e.ref_mask; w.var_mask = e.var_mask;
            w.emb = e.emb; w.pe = e.pe; w.partial = t->d_partial;
            const int e0 = launch_l0_backward(w, B, t->d_l0tot, pp(t, lp.conv_w), t->d_canon, lp.cout, lp.cin, gp(t, lp.conv_w), gp(t, lp.conv_b), gp(t, t->p_emb), s);
            if (e0) return failt(t, DAN_ERR_HIP, "layer 1's backward: the device refuses %d B of dynamic LDS for l0_bins_kernel: %s", l0_bins_lds_bytes(), hipGetErrorString((hipError_t)e0));
            cur ^= 1;
            continue;
        }
        {   // conv weight gradient: dz_l = (A dn + B a_l + C) * (a_l > 0);  gW[o][c][t] = sum_p dz[p][o] u_l[p + (t-1) d][c]
            WgradArgs w{};
            w.R = R; w.L = L; w.n_rows = n_rows; w.a1 = dn; w.a_stride = CPAD; w.a2 = t->d_a[l]; w.a_coef = t->d_coef_b; w.a_mask = 1;
            w.b_mode = 1; w.b1 = t->d_x[l - 1]; w.b_pool = t->d_pool[l];
            if (t->lazy_x[l - 1]) { w.b1 = t->d_a[l - 1]; w.b_coef = t->d_coef_f + (size_t)(l - 1) * 3 * CPAD; }
            w.taps = 3; w.dil = lp.dil; w.o_tiles = KGC; w.c_tiles = KGC; w.partial = t->d_partial; w.bias_partial = t->d_bias_partial;
            const int wgs = launch_train_wgrad(w, s);
            launch_wgrad_reduce(t->d_partial, t->d_bias_partial, wgs, 3, CPAD, w.c_tiles * 16, lp.cout, lp.cin, nullptr, gp(t, lp.conv_w), gp(t, lp.conv_b), s);
        }
        {   // data gradient du_l = conv^T(dz_l)   (for layer 1: the gradient of the 48 encoded channels)
            RowArgs a{};
            a.R = R; a.L = L; a.mode = 1; a.src1 = dn; a.s1_stride = CPAD; a.src2 = t->d_a[l]; a.coef = t->d_coef_b; a.mask_src2 = 1;
            a.w1 = t->pk_conv_d[l]; a.taps = 3; a.kg = KGC; a.dil = lp.dil;
            if (t->wino_layer[l]) { a.w1 = t->pk_wino_d[l]; a.wino = 1; }
            a.out1 = t->d_du;
            // g_{l-1} = du_l + W_b^T (dh_{l-1} * (h_{l-1} > 0)) [+ g_l through the residual skip], with the statistics of layer l-1's
            // BatchNorm backward: folded into this launch (half-read direct form) unless layer l-1 pools -- the read mean of du_l
            // must exist before g_{l-1} does -- so that du_l is never written
            // (train_row_fuses_second_product: the launcher's own predicate for the only kernel form that honours w3 -- the two
            // conditions cannot drift apart, and launch_train_row refuses a w3 it would ignore)
            if (l > 0 && H > 0 && !pool_after(c, l) && train_row_fuses_second_product(a)) {
                const LayerP& lq = t->layers[l - 1];
                a.w3 = t->pk_bot_d[l - 1]; a.src3 = t->d_dh + (size_t)(l - 1) * h_layer; a.src4 = t->d_h + (size_t)(l - 1) * h_layer;
                a.add2 = lp.residual ? g : nullptr;
                a.out1 = t->d_g[cur ^ 1];
                if (!lq.residual) { a.stats = t->d_stats; a.stat_aux = t->d_a[l - 1]; }
                fused_g = true;
            }
            const int e = launch_train_row(a, n_rows, s, stat_cap);
            if ((rc = check_row_launch(t, e, "conv data gradient", l))) return rc;
            if (fused_g && !t->layers[l - 1].residual) stat_entries = e;
        }
        if (l > 0 && pool_after(c, l)) launch_read_mean(t->d_du, t->d_dpool, B, R, L, nullptr, s);     // u_l = x_{l-1} + mean_r x_{l-1}
        cur ^= 1;
    }
    HIPT(t, hipGetLastError());
    t->pending = true;
    return DAN_OK;
}

int dan_train_backward_end(dan_trainer_t* t, float* losses, uint8_t* close) {
    if (!t) return DAN_ERR_INVALID_ARG;
    if (!t->pending) return failt(t, DAN_ERR_STATE, "dan_train_backward_end without dan_train_backward_begin");
    HIPT(t, hipSetDevice(t->cfg.device_id));
    t->pending = false;
    HIPT(t, hipStreamSynchronize(nullptr));
    if (losses) HIPT(t, hipMemcpy(losses, t->d_losses, 7 * sizeof(float), hipMemcpyDeviceToHost));
    if (close) HIPT(t, hipMemcpy(close, t->d_close, (size_t)t->last_B * 2, hipMemcpyDeviceToHost));
    return DAN_OK;
}

int dan_train_backward(dan_trainer_t* t, const uint8_t* reads, const uint8_t* qual, const uint8_t* strand, const uint8_t* ref,
                       const uint8_t* ref_mask, const uint8_t* var_mask, int64_t n_sites, const dan_train_targets* tg,
                       const uint8_t* const* dropout_masks, uint64_t seed, float* losses, uint8_t* close) {
    const int rc = dan_train_backward_begin(t, reads, qual, strand, ref, ref_mask, var_mask, n_sites, tg, dropout_masks, seed);
    if (rc) return rc;
    return dan_train_backward_end(t, losses, close);
}

int dan_train_set_global_batch(dan_trainer_t* t, float sites_per_rank, float vb_weight_per_rank, float vr_weight_per_rank) {
    if (!t || !t->finalized) return DAN_ERR_STATE;
    if (t->pending) return failt(t, DAN_ERR_STATE, "dan_train_set_global_batch with a step in flight");
    if (sites_per_rank < 0.f || vb_weight_per_rank < 0.f || vr_weight_per_rank < 0.f)
        return failt(t, DAN_ERR_INVALID_ARG, "dan_train_set_global_batch: negative normaliser");
    t->dp_mean_sites = sites_per_rank; t->dp_ce_den[0] = vb_weight_per_rank; t->dp_ce_den[1] = vr_weight_per_rank;
    return DAN_OK;
}

int dan_train_grad_bucket(dan_trainer_t* t, int32_t bucket, int64_t* offset, int64_t* count) {
    if (!t || !t->finalized) return DAN_ERR_STATE;
    if (bucket < 0 || bucket > 1) return failt(t, DAN_ERR_INVALID_ARG, "gradient buckets are 0 (FC stack + heads) and 1 (everything before them)");
    const int64_t cut = t->params[t->p_fc0w].off;
    if (offset) *offset = bucket == 0 ? cut : 0;
    if (count) *count = bucket == 0 ? t->n_flat - cut : cut;
    return DAN_OK;
}

int dan_train_wait_bucket(dan_trainer_t* t, int32_t bucket) {
    if (!t) return DAN_ERR_INVALID_ARG;
    if (!t->pending) return failt(t, DAN_ERR_STATE, "dan_train_wait_bucket: no step in flight");
    if (bucket < 0 || bucket > 1) return failt(t, DAN_ERR_INVALID_ARG, "gradient buckets are 0 and 1");
    HIPT(t, hipSetDevice(t->cfg.device_id));
    if (bucket == 0) HIPT(t, hipEventSynchronize(t->ev_tail));
    else HIPT(t, hipStreamSynchronize(nullptr));
    return DAN_OK;
}

int dan_train_apply(dan_trainer_t* t, float* grad_norm) {
    if (!t) return DAN_ERR_INVALID_ARG;
    if (!t->finalized) return failt(t, DAN_ERR_STATE, "dan_train_apply before dan_train_finalize");
    if (t->pending) return failt(t, DAN_ERR_STATE, "dan_train_apply with a step in flight: call dan_train_backward_end first");
    HIPT(t, hipSetDevice(t->cfg.device_id));
    hipStream_t s = nullptr;
    const dan_train_hyper& hp = t->hp;
    int nb = 0;
    launch_sumsq(t->G, t->n_flat, t->d_bp, &nb, s);
    launch_clip_coef(t->d_bp, nb, hp.grad_clip, t->d_clip, s);
    t->step += 1;
    const float bc1 = 1.f - std::pow(hp.beta1, (float)t->step), bc2 = 1.f - std::pow(hp.beta2, (float)t->step);
    launch_adam(t->P, t->G, t->M, t->V, t->n_flat, t->d_clip, hp.lr, hp.beta1, hp.beta2, hp.adam_eps, bc1, bc2, s);
    HIPT(t, hipGetLastError());
    HIPT(t, hipDeviceSynchronize());
    if (grad_norm) HIPT(t, hipMemcpy(grad_norm, t->d_clip, sizeof(float), hipMemcpyDeviceToHost));
    return DAN_OK;
}

int dan_train_step(dan_trainer_t* t, const uint8_t* reads, const uint8_t* qual, const uint8_t* strand, const uint8_t* ref,
                   const uint8_t* ref_mask, const uint8_t* var_mask, int64_t n_sites, const dan_train_targets* targets,
                   const uint8_t* const* dropout_masks, uint64_t seed, float* losses, uint8_t* close, float* grad_norm) {
    int rc = dan_train_backward(t, reads, qual, strand, ref, ref_mask, var_mask, n_sites, targets, dropout_masks, seed, losses, close);
    if (rc) return rc;
    return dan_train_apply(t, grad_norm);
}

int dan_train_set_lr(dan_trainer_t* t, float lr) {
    if (!t) return DAN_ERR_INVALID_ARG;
    if (!(lr >= 0.f)) return failt(t, DAN_ERR_INVALID_ARG, "learning rate must be >= 0");
    t->hp.lr = lr;
    return DAN_OK;
}

void* dan_train_grad_buffer(dan_trainer_t* t, int64_t* n_floats) {
    if (!t || !t->finalized) return nullptr;
    if (n_floats) *n_floats = t->n_flat;
    return t->G;
}

static int find_tensor(dan_trainer* t, const std::string& name, float** base, const Param** par, int64_t* n, int64_t* cols, int64_t* ld, int64_t* rows) {
    *par = nullptr;
    std::string nm = name;
    float* buf = t->P;
    if (nm.rfind("grad:", 0) == 0) { buf = t->G; nm = nm.substr(5); }
    else if (nm.rfind("m:", 0) == 0) { buf = t->M; nm = nm.substr(2); }
    else if (nm.rfind("v:", 0) == 0) { buf = t->V; nm = nm.substr(2); }
    auto it = t->views.find(nm);
    if (it != t->views.end()) {
        const View& v = it->second;
        const Param& p = t->params[v.param];
        *base = buf + p.off + v.off; *par = &p; *rows = v.rows; *cols = v.cols; *ld = p.ld; *n = v.rows * v.cols;
        return DAN_OK;
    }
    if (buf == t->P && nm.rfind("bn1D_layers.", 0) == 0) {
        const int l = atoi(nm.c_str() + 12);
        if (l >= 0 && l < t->cfg.layers) {
            const bool mean = nm.find("running_mean") != std::string::npos, var = nm.find("running_var") != std::string::npos;
            if (mean || var) {
                *base = (mean ? t->d_rmean : t->d_rvar) + (size_t)l * CPAD;
                *rows = 1; *cols = *ld = *n = t->layers[l].cout;
                return DAN_OK;
            }
        }
    }
    return failt(t, DAN_ERR_INVALID_ARG, "no tensor '%s' in the training state", name.c_str());
}

int64_t dan_train_get_tensor(dan_trainer_t* t, const char* name, float* dst, int64_t capacity) {
    if (!t || !name || !dst || capacity < 0) return DAN_ERR_INVALID_ARG;
    if (!t->finalized) return failt(t, DAN_ERR_STATE, "dan_train_get_tensor before dan_train_finalize");
    HIPT(t, hipSetDevice(t->cfg.device_id));
    HIPT(t, hipDeviceSynchronize());
    const std::string nm(name);
    const dan_config& c = t->cfg;
    const int B = t->last_B;
    const size_t rows = (size_t)B * c.reads;
    const float* src = nullptr;
    int64_t n = 0;
    if (nm == "pe") { n = std::min<int64_t>(capacity, (int64_t)t->pe_host.size()); memcpy(dst, t->pe_host.data(), n * sizeof(float)); return n; }
    if (nm.rfind("act:", 0) == 0 && nm.size() > 5) {
        const int l = atoi(nm.c_str() + 5) - 1;
        if (l < 0 || l >= c.layers) return failt(t, DAN_ERR_INVALID_ARG, "'%s': no such layer", name);
        if (nm[4] == 'a') { src = t->d_a[l]; n = rows * c.length * CPAD; }
        else if (nm[4] == 'x' && t->lazy_x[l]) {              // a layer whose x is never written: formed here for the tap
            int rc0 = 0;
            if (!t->d_xtap && (rc0 = talloc(t, &t->d_xtap, (size_t)t->max_batch * c.reads * c.length * CPAD, false))) return rc0;
            RowArgs a{};
            a.R = c.reads; a.L = c.length; a.mode = 1; a.src1 = t->d_a[l]; a.s1_stride = CPAD; a.coef = t->d_coef_f + (size_t)l * 3 * CPAD;
            a.out1 = t->d_xtap;
            launch_train_row(a, (int)rows, nullptr);
            HIPT(t, hipDeviceSynchronize());
            src = t->d_xtap; n = rows * c.length * CPAD;
        } else if (nm[4] == 'x') { src = t->d_x[l]; n = rows * c.length * CPAD; }
        else if (nm[4] == 'h' && t->d_h) { src = t->d_h + (size_t)l * rows * c.length * HPAD; n = rows * c.length * HPAD; }
    } else if (nm == "feature") { src = t->d_feat; n = (int64_t)B * t->F_stride; }
    else if (nm == "dfeature") { src = t->d_dfeat; n = (int64_t)B * t->F_stride; }
    else if (nm == "du0")      // (was: the gradient of the encoded input; layer 1's backward no longer forms it -- dan_train.hip, layer 1's backward by bins)
        return failt(t, DAN_ERR_INVALID_ARG, "debug buffer 'du0' no longer exists: layer 1's gradients come from binned sums of dz_1, the encoded input's gradient is never formed");
    else if (nm == "logits") { src = t->d_logits; n = (int64_t)B * NHEAD; }
    else if (nm == "dlogits") { src = t->d_dlogits; n = (int64_t)B * NHEAD; }
    if (src) {
        n = std::min(n, capacity);
        HIPT(t, hipMemcpy(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
        return n;
    }
    float* base = nullptr;
    const Param* par = nullptr;
    int64_t cols = 0, ld = 0, prow = 0;
    int rc = find_tensor(t, nm, &base, &par, &n, &cols, &ld, &prow);
    if (rc) return rc;
    if (capacity < n) return failt(t, DAN_ERR_INVALID_ARG, "'%s' holds %lld floats, capacity %lld", name, (long long)n, (long long)capacity);
    HIPT(t, hipMemcpy2D(dst, cols * sizeof(float), base, ld * sizeof(float), cols * sizeof(float), prow, hipMemcpyDeviceToHost));
    return n;
}

int dan_train_put_tensor(dan_trainer_t* t, const char* name, const float* src, int64_t count) {
    if (!t || !name || !src) return DAN_ERR_INVALID_ARG;
    if (!t->finalized) return failt(t, DAN_ERR_STATE, "dan_train_put_tensor before dan_train_finalize");
    HIPT(t, hipSetDevice(t->cfg.device_id));
    float* base = nullptr;
    const Param* par = nullptr;
    int64_t n = 0, cols = 0, ld = 0, prow = 0;
    int rc = find_tensor(t, name, &base, &par, &n, &cols, &ld, &prow);
    if (rc) return rc;
    if (count != n) return failt(t, DAN_ERR_SHAPE, "'%s' holds %lld floats, got %lld", name, (long long)n, (long long)count);
    HIPT(t, hipMemcpy2D(base, ld * sizeof(float), src, cols * sizeof(float), cols * sizeof(float), prow, hipMemcpyHostToDevice));
    return DAN_OK;
}

int64_t dan_train_query(const dan_trainer_t* t, const char* what) {
    if (!t || !what) return DAN_ERR_INVALID_ARG;
    const std::string w(what);
    if (w == "step") return t->step;
    if (w == "max_batch") return t->max_batch;
    if (w == "num_param_floats") return t->n_flat;
    if (w == "feature_stride") return t->F_stride;
    if (w == "feature_width") return t->F;
    return failt(t, DAN_ERR_INVALID_ARG, "dan_train_query: unknown key '%s'", what);
}

}  // extern "C"
