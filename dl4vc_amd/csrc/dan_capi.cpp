// C ABI of libdl4vc_dan.so (include/dl4vc_dan.h): handle lifecycle, checkpoint validation and
// MFMA-order weight packing, chunked execution plan.  Host code only; kernels are in dan_kernels.hip.
#include "../../include/dl4vc_dan.h"
#include "dan_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

using namespace dan;

namespace {

struct Tensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
    int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

std::string g_create_error;

struct EventPair { hipEvent_t a, b; };
struct KernelStat {
    std::vector<EventPair> pending;
    int64_t launches = 0;
    double ms = 0.0;
};

}  // namespace

struct dan_handle {
    dan_config cfg{};
    std::map<std::string, Tensor> tensors;
    bool finalized = false;
    mutable std::string err;
    int n_segments = 0;
    std::vector<int> seg_begin, seg_end;     // 0-based layer ranges
    int F = 0;                               // feature width
    int64_t F_stride = 0;                    // padded to a multiple of 16
    int n0_stride = 0;                       // fc_sizes[0] padded to a multiple of 16 (K of the second FC)
    int chunk = 0, max_batch = 0;
    bool chunk_auto = false;                                   // the chunk was sized by dan_create (free device memory), not by the caller
    int wino = 0;                            // dilation-2 layers in Winograd F(2,3) form
    int tap_layer = -1;
    int last_chunk_sites = 0;
    int64_t last_batch = 0;
    // device memory
    std::vector<void*> allocs;
    float* d_wl = nullptr;                   // [layers][LAYER_STRIDE] weight blocks (fp32 path)
    char* d_wlp = nullptr;                   // [layers][WP_LAYER_BYTES] 32x32x16 fragments of the ping-pong bf16 kernel (precision 2)
    char* d_wlr = nullptr;                   // the same blocks as 16x16x32 fragments (sixteen-wave form, dan_config.bf16_form = 1)
    float* d_wc16 = nullptr;                 // compression weights in the channel order of a 16-byte bf16 load of h
    float *d_wpool = nullptr, *d_cols = nullptr, *d_cp = nullptr, *d_zero = nullptr;   // conv(read-mean): weights [segment][128][384], scratch, result
    bool use_p = false;                      // precision 2 on dan_kernels_bf16p.hip: y and h cross HBM as bf16
    bool use_x = false;                      // precision 1 on dan_kernels_bf16x.hip: y crosses HBM as two bf16 planes (hi, lo)
    char* d_wlx = nullptr;                   // [layers][WX_LAYER_BYTES] hi / lo 16x16x32 fragments of the bf16x3 kernel
    unsigned res_mask = 0;
    float *d_emb = nullptr, *d_pe = nullptr;
    float* d_l0tab = nullptr;                // fp32 path: layer 1 by table (dan_kernels.h L0_*)
    float *d_y = nullptr, *d_pool = nullptr, *d_h = nullptr, *d_tap = nullptr;
    float* d_y2 = nullptr;                   // fp32 windows above MPOS columns (two units per read): the segments' y alternates between d_y and d_y2
    bool split = false;
    int* d_rowsrc = nullptr;                 // empty-row map of the current chunk (skip_empty_rows)
    int *d_work = nullptr, *d_work_count = nullptr;   // ... and the list of rows to compute
    int n_cus = 0;
    float *d_wc = nullptr, *d_bc = nullptr;
    float *d_feat = nullptr, *d_hid0 = nullptr, *d_hid1 = nullptr;
    float* d_fc_ws = nullptr;                                  // split-k partial sums of FC1 [2][max_batch][fc0]
    float *d_w0 = nullptr, *d_b0 = nullptr, *d_w1 = nullptr, *d_b1 = nullptr, *d_wh = nullptr, *d_bh = nullptr;
    uint16_t* d_w0x = nullptr;               // precision >= 1: FC1's weights as two bf16 planes [2][n0][F_stride] (hi, lo) for launch_fcx
    // staging for the host-pointer entry points
    uint8_t* d_in = nullptr;
    float* d_out = nullptr;
    // asynchronous host-pointer path (dan_forward_async / dan_wait): two slots, allocated on first use
    struct Slot {
        uint8_t *pin_in = nullptr, *dev_in = nullptr;
        float *pin_out = nullptr, *dev_out = nullptr;
        hipEvent_t ev_h2d = nullptr, ev_comp = nullptr, ev_done = nullptr;
        std::vector<hipEvent_t> ev_slice;        // one per conv chunk of a macro-batch: "this chunk's inputs are on the device"
        int64_t ticket = -1, n = 0;
        float* dst[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    } slot[2];
    bool async_ready = false;
    hipStream_t s_h2d = nullptr, s_comp = nullptr, s_d2h = nullptr;
    int64_t next_ticket = 0;
    // profiling
    bool profiling = false;
    std::map<std::string, KernelStat> stats;
    std::vector<EventPair> event_pool;
};

namespace {

int fail(const dan_handle* h, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(h, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(h, DAN_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <typename T>
int dev_alloc(dan_handle* h, T** p, size_t count) {
    void* q = nullptr;
    size_t bytes = std::max<size_t>(count * sizeof(T), 256);
    HIPCHK(h, hipMalloc(&q, bytes));
    h->allocs.push_back(q);
    *p = (T*)q;
    return DAN_OK;
}

template <typename T>
int dev_upload(dan_handle* h, T** p, const std::vector<T>& v) {
    int rc = dev_alloc(h, p, v.size());
    if (rc) return rc;
    HIPCHK(h, hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return DAN_OK;
}

bool pool_after(const dan_config& c, int l1) { return (c.pool_layers_mask >> l1) & 1u; }   // 1-based layer

bool is_residual(const dan_config& c, int l1) {
    return c.residual_start > 0 && l1 >= c.residual_start && !(l1 == c.layers && c.c_init != c.c_final);
}

void layer_dims(const dan_config& c, int l1, int* cin, int* cout, int* dil) {
    const int in0 = 2 * EMBED + (c.use_q ? 1 : 0) + (c.use_strand ? 1 : 0) + (c.use_mask ? 3 : 0);
    if (l1 == 1) { *cin = in0; *cout = c.c_init; *dil = 1; }
    else if (l1 < c.layers) { *cin = c.c_init; *cout = c.c_init; *dil = c.dil_mid; }
    else { *cin = c.c_init; *cout = c.c_final; *dil = c.dil_final; }
}

const Tensor* need(dan_handle* h, const std::string& name, std::initializer_list<int64_t> shape, int* rc) {
    auto it = h->tensors.find(name);
    if (it == h->tensors.end()) {
        *rc = fail(h, DAN_ERR_MISSING_TENSOR, "missing tensor '%s'", name.c_str());
        return nullptr;
    }
    const Tensor& t = it->second;
    std::vector<int64_t> want(shape);
    if (t.shape != want) {
        std::string got, exp;
        for (auto s : t.shape) got += std::to_string(s) + ",";
        for (auto s : want) exp += std::to_string(s) + ",";
        *rc = fail(h, DAN_ERR_SHAPE, "tensor '%s' has shape (%s) but the configuration needs (%s)", name.c_str(),
                   got.c_str(), exp.c_str());
        return nullptr;
    }
    return &t;
}

// MFMA A-fragment order for v_mfma_f32_16x16x4_f32 with the k order of a 16-channel group permuted:
//   packed[((tap*kg + g)*tiles + n)*64 + lane][s] = W[o = 16n + (lane&15)][c = 16g + 4(lane>>4) + s][tap]
// W(o,c,tap) is supplied by the functor (zero outside the real extents).
template <typename F>
std::vector<float> pack_frag(int taps, int kg, int tiles, F W) {
    std::vector<float> out((size_t)taps * kg * tiles * 64 * 4);
    size_t i = 0;
    for (int t = 0; t < taps; ++t)
        for (int g = 0; g < kg; ++g)
            for (int n = 0; n < tiles; ++n)
                for (int lane = 0; lane < 64; ++lane)
                    for (int s = 0; s < 4; ++s) out[i++] = W(16 * n + (lane & 15), 16 * g + 4 * (lane >> 4) + s, t);
    return out;
}

hipStream_t as_stream(void* s) { return (hipStream_t)s; }

// fp32 -> bf16 bits, round to nearest even (what v_cvt_pk_bf16_f32 does); NaN stays NaN
uint16_t bf16_bits(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
// MFMA 32x32x16 bf16 A-fragment order of the ping-pong kernel (dan_kernels.h): fragment ((ks * taps + t) * nq + q), lane, j
// (channel-group major: layer 1's three groups are the first nine steps of the walk):
//   row m = lane & 31 -> output channel 32 q + 16 ((m >> 2) & 1) + 4 (m >> 3) + (m & 3),  k = 16 ks + 8 (lane >> 5) + j
template <typename F>
void pack_fragp(uint16_t* dst, int taps, int ksteps, int nq, F W);

float bf16_float(uint16_t b) { uint32_t u = (uint32_t)b << 16; float x; memcpy(&x, &u, 4); return x; }

template <typename F>
void pack_fragp(uint16_t* dst, int taps, int ksteps, int nq, F W) {
    for (int t = 0; t < taps; ++t)
        for (int ks = 0; ks < ksteps; ++ks)
            for (int q = 0; q < nq; ++q) {
                uint16_t* f = dst + ((size_t)(ks * taps + t) * nq + q) * (WP_FRAG / 2);
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int m = lane & 31;
                        const int o = 32 * q + 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3);
                        f[lane * 8 + j] = bf16_bits(W(o, 16 * ks + 8 * (lane >> 5) + j, t));
                    }
            }
}

// MFMA 16x16x32 bf16 A-fragment order of the sixteen-wave form: fragment ((ks * taps + t) * n_ct + ct), lane, j with
//   row r = lane & 15 of channel tile ct -> output channel 32 (ct >> 1) + 8 (r >> 2) + 4 (ct & 1) + (r & 3)   (a lane's two tiles of a
//   column are 8 consecutive channels),  k = 32 ks + 8 (lane >> 4) + j
template <typename F>
void pack_fragr(uint16_t* dst, int taps, int ksteps, int n_ct, F W) {
    for (int ks = 0; ks < ksteps; ++ks)
        for (int t = 0; t < taps; ++t)
            for (int ct = 0; ct < n_ct; ++ct) {
                uint16_t* f = dst + ((size_t)(ks * taps + t) * n_ct + ct) * (WP_FRAG / 2);
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int r = lane & 15;
                        const int o = 32 * (ct >> 1) + 8 * (r >> 2) + 4 * (ct & 1) + (r & 3);
                        f[lane * 8 + j] = bf16_bits(W(o, 32 * ks + 8 * (lane >> 4) + j, t));
                    }
            }
}

// MFMA 16x16x32 bf16 A-fragment order of the bf16x3 kernel (dan_kernels_bf16x.hip): pack_fragr's row order, hi and lo plane of a
// tile side by side -- fragment ((ks * taps + t) * n_ct + ct) * 2 + plane, lo = bf16(w - hi)
template <typename F>
void pack_fragx(uint16_t* dst, int taps, int ksteps, int n_ct, F W) {
    for (int ks = 0; ks < ksteps; ++ks)
        for (int t = 0; t < taps; ++t)
            for (int ct = 0; ct < n_ct; ++ct) {
                uint16_t* f = dst + ((size_t)(ks * taps + t) * n_ct + ct) * 2 * (WP_FRAG / 2);
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int r = lane & 15;
                        const int o = 32 * (ct >> 1) + 8 * (r >> 2) + 4 * (ct & 1) + (r & 3);
                        const float w = W(o, 32 * ks + 8 * (lane >> 4) + j, t);
                        const uint16_t hi = bf16_bits(w);
                        f[lane * 8 + j] = hi;
                        f[WP_FRAG / 2 + lane * 8 + j] = bf16_bits(w - bf16_float(hi));
                    }
            }
}

int prof_begin(dan_handle* h, const char* k, hipStream_t s, EventPair* ev) {
    if (!h->profiling) return 0;
    if (!h->event_pool.empty()) { *ev = h->event_pool.back(); h->event_pool.pop_back(); }
    else { HIPCHK(h, hipEventCreate(&ev->a)); HIPCHK(h, hipEventCreate(&ev->b)); }
    HIPCHK(h, hipEventRecord(ev->a, s));
    (void)k;
    return 0;
}

int prof_end(dan_handle* h, const char* k, hipStream_t s, EventPair* ev) {
    if (!h->profiling) return 0;
    HIPCHK(h, hipEventRecord(ev->b, s));
    h->stats[k].pending.push_back(*ev);
    return 0;
}

int prof_collect(dan_handle* h, KernelStat& st) {
    for (auto& ev : st.pending) {
        HIPCHK(h, hipEventSynchronize(ev.b));
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, ev.a, ev.b));
        st.ms += ms;
        st.launches += 1;
        h->event_pool.push_back(ev);
    }
    st.pending.clear();
    return 0;
}

// receptive-field radius of layers [l_begin, l_end) (0-based): the sum of their dilations (3 taps each; model.py:214-231)
int segment_halo(const dan_config& c, int l_begin, int l_end) {
    int halo = 0;
    for (int l = l_begin; l < l_end; ++l) halo += (l == 0) ? 1 : (l + 1 < c.layers ? c.dil_mid : c.dil_final);
    return halo;
}

}  // namespace

extern "C" {

int dan_abi_version(void) { return DAN_ABI_VERSION; }

#ifndef DAN_SOURCE_HASH
#define DAN_SOURCE_HASH "unknown"
#endif
const char* dan_source_hash(void) { return DAN_SOURCE_HASH; }

const char* dan_last_error(const dan_t* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int dan_create(const dan_config* cfg, dan_t** out) {
    if (!cfg || !out) return fail(nullptr, DAN_ERR_INVALID_ARG, "dan_create: null argument");
    *out = nullptr;
    const dan_config& c = *cfg;
    if (c.layers < 1 || c.layers > DAN_MAX_LAYERS) return fail(nullptr, DAN_ERR_INVALID_ARG, "layers must be in 1..%d", DAN_MAX_LAYERS);
    if (c.reads < 1) return fail(nullptr, DAN_ERR_INVALID_ARG, "reads must be >= 1");
    if (c.precision < 0 || c.precision > 2)
        return fail(nullptr, DAN_ERR_INVALID_ARG, "precision %d unknown (0 = fp32 MFMA, 1 = bf16x3 split, 2 = bf16)", c.precision);
    // precisions 0 and 1 above MPOS columns: every read as two overlapping units (dan_kernels.h plan_units; checked per segment below)
    const int max_len = P_LMAX;
    if (c.length < 8 || c.length > max_len)
        return fail(nullptr, DAN_ERR_INVALID_ARG, "length %d unsupported by the LDS-resident path at precision %d (8..%d)", c.length, c.precision, max_len);
    if (c.c_init < 1 || c.c_init > CPAD || c.c_final < 1 || c.c_final > CPAD)
        return fail(nullptr, DAN_ERR_INVALID_ARG, "channel counts must be in 1..%d", CPAD);
    if (c.layers == 1 && c.c_init != c.c_final)
        return fail(nullptr, DAN_ERR_INVALID_ARG, "a single conv layer needs init_conv_channels == final_conv_channels (model.py:214,257)");
    if (c.bottleneck < 0 || c.bottleneck > HPAD) return fail(nullptr, DAN_ERR_INVALID_ARG, "bottleneck must be in 0..%d", HPAD);
    if (c.dil_mid < 1 || c.dil_mid > HALO || c.dil_final < 1 || c.dil_final > HALO)
        return fail(nullptr, DAN_ERR_INVALID_ARG, "dilations must be in 1..%d", HALO);
    if (c.fc_sizes[0] < 1 || c.fc_sizes[1] < 1) return fail(nullptr, DAN_ERR_INVALID_ARG, "fc_sizes must be positive");
    if (c.residual_start == 1 || c.residual_start < 0)
        return fail(nullptr, DAN_ERR_INVALID_ARG, "Do not allow residuals starting at conv layer %d", c.residual_start);   // model.py:209
    if ((c.pool_layers_mask & 1u) || (c.pool_layers_mask >> c.layers))
        return fail(nullptr, DAN_ERR_INVALID_ARG, "pool layers must lie in 1..layers-1");
    // Winograd F(2,3) form (dan_kernels.hip): fp32 path, and every conv after the first must have dilation 2
    const bool wino_ok = c.precision == 0 && (c.layers < 3 || c.dil_mid == 2) && (c.layers < 2 || c.dil_final == 2);
    if (c.conv_algo < 0 || c.conv_algo > 2) return fail(nullptr, DAN_ERR_INVALID_ARG, "conv_algo %d unknown (0 = auto, 1 = direct, 2 = winograd)", c.conv_algo);
    if (c.skip_empty_rows < 0 || c.skip_empty_rows > 1) return fail(nullptr, DAN_ERR_INVALID_ARG, "skip_empty_rows must be 0 or 1");
    if (c.bf16_form < 0 || c.bf16_form > 1) return fail(nullptr, DAN_ERR_INVALID_ARG, "bf16_form %d unknown (0 = eight waves, 1 = sixteen waves)", c.bf16_form);
    if (c.bf16_form != 0 && c.precision != 2) return fail(nullptr, DAN_ERR_INVALID_ARG, "bf16_form selects a form of the precision-2 kernel");
    if (c.conv_algo == 2 && !wino_ok)
        return fail(nullptr, DAN_ERR_INVALID_ARG, "conv_algo 2 (winograd) needs precision 0 and dilation 2 on every conv layer after the first");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || c.device_id < 0 || c.device_id >= ndev)
        return fail(nullptr, DAN_ERR_NO_DEVICE, "no HIP device %d (found %d): the DAN forward has no CPU path", c.device_id, ndev);
    dan_handle* h = new dan_handle();
    h->cfg = c;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c.device_id) == hipSuccess) h->n_cus = cus; }
    h->max_batch = c.max_batch > 0 ? c.max_batch : 4096;
    if (c.chunk_sites > 0) {
        h->chunk = c.chunk_sites;
    } else {
        // default: the largest power-of-two chunk (<= the FC macro-batch) whose segment-boundary activations y and bottleneck
        // outputs h stay under 48 GB -- 288 GB of HBM per GPU: fewer, larger launches (2048 sites at 64 x 201: +1.7 % over 128) --
        // and under a third of what the device has FREE now (a shared or smaller device sizes down instead of failing in hipMalloc;
        // the feature matrix, FC workspaces and weights take their share of the rest)
        // (precision 2 keeps y and h as bf16: half the bytes per site, twice the sites per chunk -- 1024 at 128 x 301)
        const double elem = c.precision == 2 ? 2.0 : 4.0;
        const double y_copies = (c.precision != 2 && c.length > MPOS) ? 2.0 : 1.0;       // (two units per read: y out of place)
        const double per_site = (double)c.reads * c.length * (y_copies * CPAD + (double)c.layers * (c.bottleneck > 0 ? HPAD : 0)) * elem;
        double budget = 48e9;
        size_t free_b = 0, total_b = 0;
        // (hipMemGetInfo reports on the CURRENT device: switch for the question, then back -- dan_create leaves the calling
        // thread's device as it found it; the result depends on what else holds memory at this moment, so the choice is
        // reported by dan_query("chunk_sites") and measurements key on that, never on the default being a given number)
        int prev_dev = -1;
        (void)hipGetDevice(&prev_dev);
        if (hipSetDevice(c.device_id) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0)
            budget = std::min(budget, (double)free_b / 3.0);
        if (prev_dev >= 0 && prev_dev != c.device_id) (void)hipSetDevice(prev_dev);
        h->chunk_auto = true;
        int chunk = 128;
        while (chunk * 2 <= h->max_batch && (double)(chunk * 2) * per_site <= budget) chunk *= 2;
        while (chunk > 1 && (double)chunk * per_site > budget) chunk /= 2;
        h->chunk = chunk;
    }
    // the read-axis reductions put the chunk's site index in gridDim.y (limit 65 535): no chunk exceeds 32 768 sites,
    // whatever the caller or the automatic sizing asks for (results do not depend on the chunking)
    h->chunk = std::min(h->chunk, 32768);
    h->wino = wino_ok && c.conv_algo != 1;
    h->max_batch = ((h->max_batch + h->chunk - 1) / h->chunk) * h->chunk;
    h->F = 2 * c.c_final * c.length + c.layers * c.bottleneck * c.reads;
    h->F_stride = ((int64_t)h->F + 15) / 16 * 16;
    h->n0_stride = (c.fc_sizes[0] + 15) / 16 * 16;
    int b = 0;
    for (int l1 = 1; l1 <= c.layers; ++l1)
        if (l1 == c.layers || pool_after(c, l1)) { h->seg_begin.push_back(b); h->seg_end.push_back(l1); b = l1; }
    h->n_segments = (int)h->seg_begin.size();
    if (c.precision != 2 && c.length > MPOS) {
        // every segment's two units must fit the 208-row image: half the window + the segment's receptive-field radius
        for (int sg = 0; sg < h->n_segments; ++sg) {
            SegmentArgs probe{};
            const int halo = segment_halo(c, h->seg_begin[sg], h->seg_end[sg]);
            if (!plan_units(probe, c.length, halo)) {
                const int rc = fail(nullptr, DAN_ERR_INVALID_ARG, "length %d unsupported at precisions 0 and 1 with these dilations: layers %d..%d reach %d "
                                    "columns sideways, half the window plus that exceeds the %d-column LDS image", c.length,
                                    h->seg_begin[sg] + 1, h->seg_end[sg], halo, MPOS);
                delete h;
                return rc;
            }
        }
        h->split = true;
    }
    *out = h;
    return DAN_OK;
}

int dan_set_tensor(dan_t* h, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
    if (!h || !name || !data || (ndim > 0 && !shape) || ndim < 0 || ndim > 8)
        return fail(h, DAN_ERR_INVALID_ARG, "dan_set_tensor: bad argument");
    if (h->finalized) return fail(h, DAN_ERR_STATE, "dan_set_tensor('%s') after dan_finalize", name);
    Tensor t;
    t.shape.assign(shape, shape + ndim);
    for (auto s : t.shape)
        if (s < 0) return fail(h, DAN_ERR_SHAPE, "tensor '%s': negative dimension", name);
    t.data.assign(data, data + t.numel());
    h->tensors[name] = std::move(t);
    return DAN_OK;
}

int dan_finalize(dan_t* h) {
    if (!h) return DAN_ERR_INVALID_ARG;
    if (h->finalized) return fail(h, DAN_ERR_STATE, "dan_finalize called twice");
    const dan_config& c = h->cfg;
    HIPCHK(h, hipSetDevice(c.device_id));
    int rc = DAN_OK;
    const int L = c.length, R = c.reads, H = c.bottleneck;

    // ---- embeddings + positional encoding (model.py:143-145,154-162)
    const Tensor* emb = need(h, "embeddings.weight", {VOCAB, EMBED}, &rc); if (!emb) return rc;
    const Tensor* pe = need(h, "pe", {1, L, EMBED}, &rc); if (!pe) return rc;
    if ((rc = dev_upload(h, &h->d_emb, emb->data))) return rc;
    if ((rc = dev_upload(h, &h->d_pe, pe->data))) return rc;

    // ---- conv stack: one fixed-stride weight block per layer (dan_kernels.h)
    std::vector<float> wl((size_t)c.layers * LAYER_STRIDE, 0.f);
    // precision 2: the ping-pong kernel (bf16 y / h in HBM); precision 1: the split kernel (y as two bf16 planes)
    h->use_p = c.precision == 2;
    h->use_x = c.precision == 1;
    const bool conv_pool = h->use_p || h->use_x;                 // the read-mean enters the layer behind it as conv(pool)
    std::vector<char> wlp(h->use_p ? (size_t)c.layers * WP_LAYER_BYTES : 0, 0);
    std::vector<char> wlr(h->use_p ? (size_t)c.layers * WP_LAYER_BYTES : 0, 0);
    std::vector<char> wlx(h->use_x ? (size_t)c.layers * WX_LAYER_BYTES : 0, 0);
    std::vector<float> wpool_all(conv_pool ? (size_t)h->n_segments * CPAD * 3 * CPAD : 0, 0.f);
    std::vector<float> wc16_all;
    const size_t wc16_layer = (size_t)L * 2 * 2 * 64 * 4;    // floats: [pos][n 2][plane 2][lane 64][8 bf16]
    if (h->use_p && H > 0) wc16_all.resize((size_t)c.layers * wc16_layer);
    std::vector<float> wc_all, bc_all((size_t)c.layers * HPAD, 0.f);
    const size_t wc_layer = (size_t)L * 2 * 2 * 64 * 4;      // [g = 2L][tile 2][lane 64][4]
    if (H > 0) wc_all.resize((size_t)c.layers * wc_layer);
    // canonical position of the reference's layer-1 input channels (model.py:517,543,558,625)
    std::vector<int> canon;
    for (int i = 0; i < 2 * EMBED; ++i) canon.push_back(i);
    if (c.use_q) canon.push_back(40);
    if (c.use_strand) canon.push_back(41);
    if (c.use_mask) { canon.push_back(42); canon.push_back(43); canon.push_back(44); }
    h->res_mask = 0;

    for (int l = 0; l < c.layers; ++l) {
        const int l1 = l + 1;
        int cin, cout, dil;
        layer_dims(c, l1, &cin, &cout, &dil);
        float* blk = wl.data() + (size_t)l * LAYER_STRIDE;
        const std::string p = "conv1D_layers." + std::to_string(l);
        const Tensor* w = need(h, p + ".weight", {cout, cin, 1, 3}, &rc); if (!w) return rc;
        const Tensor* b = need(h, p + ".bias", {cout}, &rc); if (!b) return rc;
        const int kg = (l == 0) ? KG0 : KGC;
        std::vector<int> inv(kg * 16, -1);                   // canonical channel -> reference channel
        for (int i = 0; i < cin; ++i) inv[l == 0 ? canon[i] : i] = i;
        auto Wf = [&](int o, int cc, int t) -> float {
            if (o >= cout || cc >= (int)inv.size() || inv[cc] < 0) return 0.f;
            return w->data[((size_t)o * cin + inv[cc]) * 3 + t];
        };
        std::vector<float> packed = pack_frag(3, kg, KGC, Wf);
        std::copy(packed.begin(), packed.end(), blk + W_OFF);
        if (l == 0 && c.precision == 0) {
            // layer 1 by table (dan_kernels.h L0_*): conv1 is linear in the terms its input column is a sum of; sums in double
            std::vector<float> tab(l0_tab_floats(L), 0.f);
            for (int t = 0; t < 3; ++t)
                for (int o = 0; o < CPAD; ++o) {
                    for (int ta = 0; ta < VOCAB; ++ta)
                        for (int tb = 0; tb < VOCAB; ++tb) {
                            double v = 0.0;
                            for (int e = 0; e < EMBED; ++e)
                                v += (double)Wf(o, e, t) * emb->data[(size_t)ta * EMBED + e] + (double)Wf(o, EMBED + e, t) * emb->data[(size_t)tb * EMBED + e];
                            tab[L0_TJ_OFF + ((size_t)t * L0_NTJ + ta * 10 + tb) * CPAD + o] = (float)v;
                        }
                    for (int k = 0; k < 5; ++k) tab[L0_WSC_OFF + ((size_t)k * 3 + t) * CPAD + o] = Wf(o, 2 * EMBED + k, t);
                }
            // the positional term of tap t at column w: sum_e (W[o][e][t] + W[o][20 + e][t]) pe[w][e]; PE[variant][w] = the taps whose
            // column w + t - 1 exists for a unit that has / lacks a left / right neighbour at w (and lies inside the window anyway)
            std::vector<double> pet((size_t)3 * L * CPAD);
            for (int t = 0; t < 3; ++t)
                for (int wcol = 0; wcol < L; ++wcol)
                    for (int o = 0; o < CPAD; ++o) {
                        double v = 0.0;
                        for (int e = 0; e < EMBED; ++e) v += ((double)Wf(o, e, t) + (double)Wf(o, EMBED + e, t)) * pe->data[(size_t)wcol * EMBED + e];
                        pet[((size_t)t * L + wcol) * CPAD + o] = v;
                    }
            for (int var = 0; var < 3; ++var)
                for (int wcol = 0; wcol < L; ++wcol)
                    for (int o = 0; o < CPAD; ++o) {
                        double v = pet[((size_t)1 * L + wcol) * CPAD + o];
                        if (var != 1 && wcol - 1 >= 0) v += pet[((size_t)0 * L + wcol - 1) * CPAD + o];
                        if (var != 2 && wcol + 1 < L) v += pet[((size_t)2 * L + wcol + 1) * CPAD + o];
                        tab[L0_PE_OFF + ((size_t)var * L + wcol) * CPAD + o] = (float)v;
                    }
            if ((rc = dev_upload(h, &h->d_l0tab, tab))) return rc;
        }
        if (l > 0) {
            // Winograd F(2,3) weight transform U = [g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2], formed in double
            auto Wu = [&](int o, int cc, int k) -> float {
                const double g0 = Wf(o, cc, 0), g1 = Wf(o, cc, 1), g2 = Wf(o, cc, 2);
                return k == 0 ? (float)g0 : k == 1 ? (float)((g0 + g1 + g2) * 0.5) : k == 2 ? (float)((g0 - g1 + g2) * 0.5) : (float)g2;
            };
            std::vector<float> pw = pack_frag(4, KGC, KGC, Wu);
            std::copy(pw.begin(), pw.end(), blk + WW_OFF);
        }
        char* blkp = h->use_p ? wlp.data() + (size_t)l * WP_LAYER_BYTES : nullptr;
        if (blkp) pack_fragp((uint16_t*)(blkp + WP_CONV_OFF), 3, l == 0 ? P_KS0 : P_KSC, 4, Wf);
        char* blkr = h->use_p ? wlr.data() + (size_t)l * WP_LAYER_BYTES : nullptr;
        if (blkr) pack_fragr((uint16_t*)(blkr + WP_CONV_OFF), 3, l == 0 ? 2 : 4, 8, Wf);
        char* blkx = h->use_x ? wlx.data() + (size_t)l * WX_LAYER_BYTES : nullptr;
        if (blkx) pack_fragx((uint16_t*)(blkx + WX_CONV_OFF), 3, l == 0 ? X_KS0 : X_KS, 8, Wf);
        if (conv_pool && l > 0)
            for (int sg = 1; sg < h->n_segments; ++sg)
                if (h->seg_begin[sg] == l) {                     // the layer behind a pool layer: its weights, [o][t * 128 + c] -- bf16-rounded for the
                    float* wp = wpool_all.data() + (size_t)sg * CPAD * 3 * CPAD;   // plain-bf16 kernel (its oracle's "storage" mode), fp32 for bf16x3
                    for (int o = 0; o < CPAD; ++o)
                        for (int t = 0; t < 3; ++t)
                            for (int cc = 0; cc < CPAD; ++cc)
                                wp[((size_t)o * 3 + t) * CPAD + cc] = h->use_p ? bf16_float(bf16_bits(Wf(o, cc, t))) : Wf(o, cc, t);
                }
        float* cst = blk + CST_OFF;
        for (int o = 0; o < cout; ++o) { cst[CST_BIAS + o] = b->data[o]; cst[CST_SCALE + o] = 1.f; }
        if (c.use_bn) {                                      // eval-mode BN after the ReLU, eps 1e-5 (model.py:750-751)
            const std::string q = "bn1D_layers." + std::to_string(l);
            const Tensor* g = need(h, q + ".weight", {cout}, &rc); if (!g) return rc;
            const Tensor* be = need(h, q + ".bias", {cout}, &rc); if (!be) return rc;
            const Tensor* mu = need(h, q + ".running_mean", {cout}, &rc); if (!mu) return rc;
            const Tensor* var = need(h, q + ".running_var", {cout}, &rc); if (!var) return rc;
            for (int o = 0; o < cout; ++o) {
                const double sc = (double)g->data[o] / std::sqrt((double)var->data[o] + 1e-5);
                cst[CST_SCALE + o] = (float)sc;
                cst[CST_SHIFT + o] = (float)((double)be->data[o] - (double)mu->data[o] * sc);
            }
        }
        if (is_residual(c, l1)) {
            const std::string q = "residual_conv_layers." + std::to_string(l1 - c.residual_start);   // model.py:760
            const Tensor* wr = need(h, q + ".weight", {cout, cout, 1, 1}, &rc); if (!wr) return rc;
            const Tensor* br = need(h, q + ".bias", {cout}, &rc); if (!br) return rc;
            auto Wr = [&](int o, int cc, int) -> float { return (o < cout && cc < cout) ? wr->data[(size_t)o * cout + cc] : 0.f; };
            std::vector<float> pr = pack_frag(1, KGC, KGC, Wr);
            std::copy(pr.begin(), pr.end(), blk + WRES_OFF);
            if (blkp) pack_fragp((uint16_t*)(blkp + WP_RES_OFF), 1, P_KSC, 4, Wr);
            if (blkr) pack_fragr((uint16_t*)(blkr + WP_RES_OFF), 1, 4, 8, Wr);
            if (blkx) pack_fragx((uint16_t*)(blkx + WX_RES_OFF), 1, X_KS, 8, Wr);
            for (int o = 0; o < cout; ++o) cst[CST_BRES + o] = br->data[o];
            h->res_mask |= 1u << l;
        }
        if (H > 0) {
            const std::string q = "conv1D_bottleneck_layers." + std::to_string(l);
            const Tensor* wb = need(h, q + ".weight", {H, cout, 1, 1}, &rc); if (!wb) return rc;
            const Tensor* bb = need(h, q + ".bias", {H}, &rc); if (!bb) return rc;
            auto Wb = [&](int o, int cc, int) -> float { return (o < H && cc < cout) ? wb->data[(size_t)o * cout + cc] : 0.f; };
            std::vector<float> pb = pack_frag(1, KGC, 2, Wb);
            std::copy(pb.begin(), pb.end(), blk + WBOT_OFF);
            if (blkp) pack_fragp((uint16_t*)(blkp + WP_BOT_OFF), 1, P_KSC, 1, Wb);
            if (blkr) pack_fragr((uint16_t*)(blkr + WP_BOT_OFF), 1, 4, 2, Wb);
            if (blkx) pack_fragx((uint16_t*)(blkx + WX_BOT_OFF), 1, X_KS, 2, Wb);
            for (int o = 0; o < H; ++o) cst[CST_BBOT + o] = bb->data[o];
            const std::string z = "conv1D_compression_layers." + std::to_string(l);
            const Tensor* wcm = need(h, z + ".weight", {H, H, 1, L}, &rc); if (!wcm) return rc;
            const Tensor* bcm = need(h, z + ".bias", {H}, &rc); if (!bcm) return rc;
            // k = p*32 + c, k-group g = 2p + (c >> 4):  packed[g][n][lane][s] = Wc[o][c][p]
            float* dst = wc_all.data() + (size_t)l * wc_layer;
            size_t i = 0;
            for (int g = 0; g < 2 * L; ++g)
                for (int n = 0; n < 2; ++n)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int s = 0; s < 4; ++s) {
                            const int o = 16 * n + (lane & 15), cc = 16 * (g & 1) + 4 * (lane >> 4) + s, pp = g >> 1;
                            dst[i++] = (o < H && cc < H) ? wcm->data[((size_t)o * H + cc) * L + pp] : 0.f;
                        }
            for (int o = 0; o < H; ++o) bc_all[(size_t)l * HPAD + o] = bcm->data[o];
            if (h->use_p) {
                uint16_t* d16 = (uint16_t*)(wc16_all.data() + (size_t)l * wc16_layer);
                for (int pp = 0; pp < L; ++pp)
                    for (int n = 0; n < 2; ++n)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int s8 = 0; s8 < 8; ++s8) {
                                const int o = 16 * n + (lane & 15), cc = 8 * (lane >> 4) + s8;
                                const float w = (o < H && cc < H) ? wcm->data[((size_t)o * H + cc) * L + pp] : 0.f;
                                const uint16_t hi = bf16_bits(w);
                                const size_t base = (((size_t)pp * 2 + n) * 2) * 64 * 8;
                                d16[base + (size_t)lane * 8 + s8] = hi;                                   // plane 0
                                d16[base + 64 * 8 + (size_t)lane * 8 + s8] = bf16_bits(w - bf16_float(hi));   // plane 1
                            }
            }
        }
    }
    if (h->use_p) {
        for (int l = 0; l < c.layers; ++l)
        {
            memcpy(wlp.data() + (size_t)l * WP_LAYER_BYTES + WP_CST_OFF, wl.data() + (size_t)l * LAYER_STRIDE + CST_OFF,
                   CST_FLOATS * sizeof(float));
            memcpy(wlr.data() + (size_t)l * WP_LAYER_BYTES + WP_CST_OFF, wl.data() + (size_t)l * LAYER_STRIDE + CST_OFF,
                   CST_FLOATS * sizeof(float));
        }
        if ((rc = dev_upload(h, &h->d_wlp, wlp))) return rc;
        if ((rc = dev_upload(h, &h->d_wlr, wlr))) return rc;
        if (H > 0 && (rc = dev_upload(h, &h->d_wc16, wc16_all))) return rc;
    }
    if (h->use_x) {
        for (int l = 0; l < c.layers; ++l) {
            float* cx = (float*)(wlx.data() + (size_t)l * WX_LAYER_BYTES + WX_CST_OFF);
            memcpy(cx, wl.data() + (size_t)l * LAYER_STRIDE + CST_OFF, CST_FLOATS * sizeof(float));
            // this family's epilogue computes relu(acc + b) * sc + sh as max(acc, -b) * sc + (sh + b * sc) (dan_kernels_bf16x.hip):
            // the bias slot holds -b, the shift slot sh + b * sc (formed in double)
            for (int ch = 0; ch < CPAD; ++ch) {
                const double b = cx[CST_BIAS + ch];
                cx[CST_SHIFT + ch] = (float)((double)cx[CST_SHIFT + ch] + b * (double)cx[CST_SCALE + ch]);
                cx[CST_BIAS + ch] = (float)-b;
            }
        }
        if ((rc = dev_upload(h, &h->d_wlx, wlx))) return rc;
    }
    if (conv_pool && h->n_segments > 1) {
        if ((rc = dev_upload(h, &h->d_wpool, wpool_all))) return rc;
        std::vector<float> zeros(CPAD, 0.f);
        if ((rc = dev_upload(h, &h->d_zero, zeros))) return rc;
    }
    if ((rc = dev_upload(h, &h->d_wl, wl))) return rc;
    if (H > 0) {
        if ((rc = dev_upload(h, &h->d_wc, wc_all)) || (rc = dev_upload(h, &h->d_bc, bc_all))) return rc;
    }

    // ---- FC stack + heads (model.py:362-377,406-415)
    const int n0 = c.fc_sizes[0], n1 = c.fc_sizes[1];
    const Tensor* w0 = need(h, "fc.0.weight", {n0, h->F}, &rc); if (!w0) return rc;
    const Tensor* b0 = need(h, "fc.0.bias", {n0}, &rc); if (!b0) return rc;
    const Tensor* w1 = need(h, "fc.1.weight", {n1, n0}, &rc); if (!w1) return rc;
    const Tensor* b1 = need(h, "fc.1.bias", {n1}, &rc); if (!b1) return rc;
    if (c.precision == 0) {
        if ((rc = dev_alloc(h, &h->d_w0, (size_t)n0 * h->F_stride))) return rc;
        HIPCHK(h, hipMemset(h->d_w0, 0, (size_t)n0 * h->F_stride * sizeof(float)));
        HIPCHK(h, hipMemcpy2D(h->d_w0, h->F_stride * sizeof(float), w0->data.data(), (size_t)h->F * sizeof(float),
                              (size_t)h->F * sizeof(float), n0, hipMemcpyHostToDevice));
    } else {
        // the bf16 modes run FC1 (99.5 % of the FC stack's FLOPs) on the bf16 matrix cores with split operands: the weights as two
        // bf16 planes, hi = bf16(w), lo = bf16(w - hi) -- the bytes of the fp32 matrix
        const size_t plane = (size_t)n0 * h->F_stride;
        std::vector<uint16_t> wx(2 * plane, 0);
        for (int o = 0; o < n0; ++o)
            for (int64_t k = 0; k < h->F; ++k) {
                const float w = w0->data[(size_t)o * h->F + k];
                const uint16_t hi = bf16_bits(w);
                wx[(size_t)o * h->F_stride + k] = hi;
                wx[plane + (size_t)o * h->F_stride + k] = bf16_bits(w - bf16_float(hi));
            }
        if ((rc = dev_upload(h, &h->d_w0x, wx))) return rc;
    }
    if ((rc = dev_alloc(h, &h->d_w1, (size_t)n1 * h->n0_stride))) return rc;
    HIPCHK(h, hipMemset(h->d_w1, 0, (size_t)n1 * h->n0_stride * sizeof(float)));
    HIPCHK(h, hipMemcpy2D(h->d_w1, h->n0_stride * sizeof(float), w1->data.data(), (size_t)n0 * sizeof(float),
                          (size_t)n0 * sizeof(float), n1, hipMemcpyHostToDevice));
    if ((rc = dev_upload(h, &h->d_b0, b0->data)) || (rc = dev_upload(h, &h->d_b1, b1->data))) return rc;
    static const struct { const char* name; int n; } heads[] = {{"fcHidden2BinTarget", 2}, {"fcHidden2VT", 3}, {"fcHidden2AF", 1},
                                                                 {"fcHidden2Coverage", 1}, {"fcHidden2VB", VOCAB}, {"fcHidden2VR", VOCAB}};
    std::vector<float> wh, bh;
    for (auto& hd : heads) {
        const Tensor* w = need(h, std::string(hd.name) + ".weight", {hd.n, n1}, &rc); if (!w) return rc;
        const Tensor* b = need(h, std::string(hd.name) + ".bias", {hd.n}, &rc); if (!b) return rc;
        wh.insert(wh.end(), w->data.begin(), w->data.end());
        bh.insert(bh.end(), b->data.begin(), b->data.end());
    }
    if ((rc = dev_upload(h, &h->d_wh, wh)) || (rc = dev_upload(h, &h->d_bh, bh))) return rc;

    // ---- activations / workspaces, sized for one chunk (conv) and one macro-batch (FC)
    const size_t read_floats = (size_t)L * CPAD;
    const size_t act_div = h->use_p ? 2 : 1;                 // (bf16 y / h: half a float per element)
    if ((rc = dev_alloc(h, &h->d_y, (size_t)h->chunk * R * read_floats / act_div))) return rc;
    if (h->split && (rc = dev_alloc(h, &h->d_y2, (size_t)h->chunk * R * read_floats))) return rc;
    if ((rc = dev_alloc(h, &h->d_pool, (size_t)h->chunk * read_floats))) return rc;
    if (conv_pool && h->n_segments > 1) {
        if ((rc = dev_alloc(h, &h->d_cp, (size_t)h->chunk * read_floats))) return rc;
        if ((rc = dev_alloc(h, &h->d_cols, (size_t)h->chunk * L * 3 * CPAD))) return rc;
    }
    if (c.skip_empty_rows) {
        float* tmp = nullptr;
        if ((rc = dev_alloc(h, &tmp, (size_t)h->chunk * R))) return rc;     // int32 per pileup row
        h->d_rowsrc = (int*)tmp;
        if ((rc = dev_alloc(h, &tmp, (size_t)h->chunk * R + 4))) return rc;
        h->d_work = (int*)tmp;
        h->d_work_count = h->d_work + (size_t)h->chunk * R;
    }
    if (H > 0 && (rc = dev_alloc(h, &h->d_h, (size_t)c.layers * h->chunk * R * L * HPAD / act_div))) return rc;
    if ((rc = dev_alloc(h, &h->d_feat, (size_t)h->max_batch * h->F_stride))) return rc;
    HIPCHK(h, hipMemset(h->d_feat, 0, (size_t)h->max_batch * h->F_stride * sizeof(float)));
    if ((rc = dev_alloc(h, &h->d_hid0, (size_t)h->max_batch * h->n0_stride)) || (rc = dev_alloc(h, &h->d_hid1, (size_t)h->max_batch * n1))) return rc;
    HIPCHK(h, hipMemset(h->d_hid0, 0, (size_t)h->max_batch * h->n0_stride * sizeof(float)));
    if ((rc = dev_alloc(h, &h->d_fc_ws, (size_t)2 * h->max_batch * h->cfg.fc_sizes[0]))) return rc;
    const size_t in_site = (size_t)3 * R * L + 3 * L;
    if ((rc = dev_alloc(h, &h->d_in, (size_t)h->max_batch * in_site))) return rc;
    if ((rc = dev_alloc(h, &h->d_out, (size_t)h->max_batch * (2 + 3 + 3 + 1 + 22)))) return rc;
    HIPCHK(h, hipDeviceSynchronize());
    h->tensors.clear();
    h->finalized = true;
    return DAN_OK;
}

void dan_destroy(dan_t* h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device_id);
    (void)hipDeviceSynchronize();
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->async_ready) {
        for (auto& sl : h->slot) {
            (void)hipHostFree(sl.pin_in); (void)hipHostFree(sl.pin_out);
            (void)hipEventDestroy(sl.ev_h2d); (void)hipEventDestroy(sl.ev_comp); (void)hipEventDestroy(sl.ev_done);
            for (auto& e : sl.ev_slice) (void)hipEventDestroy(e);
        }
        (void)hipStreamDestroy(h->s_h2d); (void)hipStreamDestroy(h->s_comp); (void)hipStreamDestroy(h->s_d2h);
    }
    for (auto& kv : h->stats) for (auto& ev : kv.second.pending) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    for (auto& ev : h->event_pool) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    delete h;
}

// feed (may be null): called in front of the k-th conv chunk's first launch; it brings that chunk's inputs onto the device (on a
// stream of its own) and hands back the event the compute stream waits for.  The asynchronous host path uploads a macro-batch
// chunk by chunk THROUGH this: chunk k's launches are queued before the host touches chunk k + 1, so the forward of chunk 0 starts
// after one chunk's staging and the staging of chunk k + 1 runs under chunk k's kernels.
struct ChunkFeed {
    int (*fn)(void* ctx, int64_t first_site, int n_sites, int64_t k, hipEvent_t* ready);
    void* ctx;
};
static int forward_device_impl(dan_t* h, const uint8_t* reads, const uint8_t* qual, const uint8_t* strand, const uint8_t* ref,
                               const uint8_t* ref_mask, const uint8_t* var_mask, int64_t n_sites, float* bin_logits,
                               float* vt_logits, float* vt_prob, float* bp, float* aux, void* stream, const ChunkFeed* feed);

int dan_forward_device(dan_t* h, const uint8_t* reads, const uint8_t* qual, const uint8_t* strand, const uint8_t* ref,
                       const uint8_t* ref_mask, const uint8_t* var_mask, int64_t n_sites, float* bin_logits,
                       float* vt_logits, float* vt_prob, float* bp, float* aux, void* stream) {
    return forward_device_impl(h, reads, qual, strand, ref, ref_mask, var_mask, n_sites, bin_logits, vt_logits, vt_prob, bp, aux, stream, nullptr);
}

static int forward_device_impl(dan_t* h, const uint8_t* reads, const uint8_t* qual, const uint8_t* strand, const uint8_t* ref,
                               const uint8_t* ref_mask, const uint8_t* var_mask, int64_t n_sites, float* bin_logits,
                               float* vt_logits, float* vt_prob, float* bp, float* aux, void* stream, const ChunkFeed* feed) {
    if (!h) return DAN_ERR_INVALID_ARG;
    if (!h->finalized) return fail(h, DAN_ERR_STATE, "dan_forward before dan_finalize");
    if (n_sites < 0) return fail(h, DAN_ERR_INVALID_ARG, "negative site count");
    if (n_sites == 0) return DAN_OK;
    if (!reads || !qual || !strand || !ref || !ref_mask || !var_mask) return fail(h, DAN_ERR_INVALID_ARG, "null input plane");
    const dan_config& c = h->cfg;
    HIPCHK(h, hipSetDevice(c.device_id));
    hipStream_t s = as_stream(stream);
    const int L = c.length, R = c.reads, H = c.bottleneck;
    const size_t rl = (size_t)R * L;
    const long long h_layer_stride = (long long)h->chunk * R * L * HPAD;
    for (int64_t mb = 0; mb < n_sites; mb += h->max_batch) {
        const int nb = (int)std::min<int64_t>(h->max_batch, n_sites - mb);
        for (int c0 = 0; c0 < nb; c0 += h->chunk) {
            const int ns = std::min(h->chunk, nb - c0);
            const int64_t g0 = mb + c0;                      // first site of the chunk in the caller's arrays
            if (feed) {
                hipEvent_t ready;
                int rcf = feed->fn(feed->ctx, g0, ns, g0 / h->chunk, &ready); if (rcf) return rcf;
                HIPCHK(h, hipStreamWaitEvent(s, ready, 0));
            }
            if (h->d_rowsrc) {
                EventPair evm{};
                int rcm = prof_begin(h, "row_map", s, &evm); if (rcm) return rcm;
                launch_row_map(reads + g0 * rl, qual + g0 * rl, strand + g0 * rl, h->d_rowsrc, h->d_work, h->d_work_count, ns, R, L, s);
                rcm = prof_end(h, "row_map", s, &evm); if (rcm) return rcm;
            }
            float* y_seg = h->d_y;                           // where the segment just launched left its output
            for (int sg = 0; sg < h->n_segments; ++sg) {
                SegmentArgs a{};
                a.wl = h->d_wl;
                a.l_begin = h->seg_begin[sg]; a.l_end = h->seg_end[sg];
                a.n_layers = c.layers; a.dil_mid = c.dil_mid; a.dil_final = c.dil_final;
                a.res_mask = h->res_mask; a.has_hw = H > 0;
                a.R = R; a.L = L;
                a.reads = reads + g0 * rl; a.qual = qual + g0 * rl; a.strand = strand + g0 * rl;
                a.ref = ref + g0 * L; a.ref_mask = ref_mask + g0 * L; a.var_mask = var_mask + g0 * L;
                a.emb = h->d_emb; a.pe = h->d_pe;
                // one unit per read and y in place -- or, above MPOS columns at precision 0, two units per read and y alternating
                // between two buffers (a unit reads the other unit's columns of the input while that one stores its output)
                float* const y_in = y_seg;                   // the previous segment's output
                if (h->split) y_seg = (y_seg == h->d_y) ? h->d_y2 : h->d_y;
                a.y = y_in; a.y_out = y_seg;
                if (c.precision != 2 && !plan_units(a, L, segment_halo(c, a.l_begin, a.l_end)))
                    return fail(h, DAN_ERR_STATE, "unit plan failed for segment %d", sg);
                a.pool = sg > 0 ? h->d_pool : nullptr;
                a.h = h->d_h; a.h_layer_stride = h_layer_stride;
                const bool tap_here = h->tap_layer >= 0 &&
                                      ((h->tap_layer == 0 && sg == 0) || (h->tap_layer > a.l_begin && h->tap_layer <= a.l_end));
                a.tap = tap_here ? h->d_tap : nullptr;
                a.tap_layer = h->tap_layer;
                a.wino = h->wino;
                a.l0_tab = h->d_l0tab;
                a.work = h->d_work; a.work_count = h->d_rowsrc ? h->d_work_count : nullptr;
                EventPair ev{};
                int rc = prof_begin(h, "conv_segment", s, &ev); if (rc) return rc;
                if (c.precision == 0) {
                    launch_segment(a, ns, h->n_cus, s);
                } else if (h->use_x) {
                    SegmentXArgs b{};
                    b.wl = h->d_wlx; b.l_begin = a.l_begin; b.l_end = a.l_end; b.n_layers = a.n_layers;
                    b.dil_mid = a.dil_mid; b.dil_final = a.dil_final; b.res_mask = a.res_mask; b.has_hw = a.has_hw;
                    b.R = a.R; b.L = a.L; b.reads = a.reads; b.qual = a.qual; b.strand = a.strand; b.ref = a.ref;
                    b.ref_mask = a.ref_mask; b.var_mask = a.var_mask; b.emb = a.emb; b.pe = a.pe;
                    b.y = (uint16_t*)a.y; b.y_out = (uint16_t*)a.y_out; b.pool = sg > 0 ? h->d_cp : nullptr; b.h = h->d_h; b.h_layer_stride = a.h_layer_stride;
                    b.units = a.units; b.Lw = a.Lw;               // (plan_units above: b.L = the unit length)
                    for (int u = 0; u < 2; ++u) { b.u_off[u] = a.u_off[u]; b.u_len[u] = a.u_len[u]; b.own_lo[u] = a.own_lo[u]; b.own_hi[u] = a.own_hi[u]; }
                    b.tap = a.tap; b.tap_layer = a.tap_layer; b.work = a.work; b.work_count = a.work_count;
                    b.stagger = -1;                              // (the launcher's own start offsets)
                    launch_segmentx(b, ns, h->n_cus, s);
                } else if (h->use_p) {
                    SegmentPArgs b{};
                    b.wl = h->d_wlp; b.wlr = h->d_wlr; b.form = c.bf16_form; b.l_begin = a.l_begin; b.l_end = a.l_end; b.n_layers = a.n_layers;
                    b.dil_mid = a.dil_mid; b.dil_final = a.dil_final; b.res_mask = a.res_mask; b.has_hw = a.has_hw;
                    b.R = a.R; b.L = a.L; b.reads = a.reads; b.qual = a.qual; b.strand = a.strand; b.ref = a.ref;
                    b.ref_mask = a.ref_mask; b.var_mask = a.var_mask; b.emb = a.emb; b.pe = a.pe;
                    b.y = (uint16_t*)h->d_y; b.pool = sg > 0 ? h->d_cp : nullptr; b.h = (uint16_t*)h->d_h; b.h_layer_stride = a.h_layer_stride;
                    b.tap = a.tap; b.tap_layer = a.tap_layer; b.work = a.work; b.work_count = a.work_count;
                    launch_segmentp(b, ns, h->n_cus, s);
                }
                rc = prof_end(h, "conv_segment", s, &ev); if (rc) return rc;
                if (sg + 1 < h->n_segments) {
                    rc = prof_begin(h, "pool", s, &ev); if (rc) return rc;
                    if (h->use_p || h->use_x) {
                        if (h->use_x) launch_read_meanx((const uint16_t*)y_seg, h->d_pool, ns, R, L, h->d_rowsrc, s);
                        else launch_read_mean16((const uint16_t*)h->d_y, h->d_pool, ns, R, L, h->d_rowsrc, s);
                        const int ln = h->seg_begin[sg + 1];                      // 0-based layer behind the pool: its dilation
                        launch_conv_pool(h->d_pool, h->d_wpool + (size_t)(sg + 1) * CPAD * 3 * CPAD, h->d_zero, h->d_cols, h->d_cp, ns, L,
                                         ln + 1 < c.layers ? c.dil_mid : c.dil_final, s);
                    } else launch_read_mean(y_seg, h->d_pool, ns, R, L, h->d_rowsrc, s);
                    rc = prof_end(h, "pool", s, &ev); if (rc) return rc;
                    HIPCHK(h, hipGetLastError());            // a refused launch must not let garbage flow on to the FC
                }
            }
            float* feat = h->d_feat + (size_t)c0 * h->F_stride;
            EventPair ev{};
            int rc = prof_begin(h, "pool", s, &ev); if (rc) return rc;
            if (h->use_x) launch_final_poolx((const uint16_t*)y_seg, feat, h->F_stride, ns, R, L, c.c_final, h->d_rowsrc, s);
            else if (h->use_p) launch_final_pool16((const uint16_t*)h->d_y, feat, h->F_stride, ns, R, L, c.c_final, h->d_rowsrc, s);
            else launch_final_pool(y_seg, feat, h->F_stride, ns, R, L, c.c_final, h->d_rowsrc, s);
            rc = prof_end(h, "pool", s, &ev); if (rc) return rc;
            HIPCHK(h, hipGetLastError());
            if (H > 0) {
                rc = prof_begin(h, "highway", s, &ev); if (rc) return rc;
                if (h->use_p)
                    launch_highway16((const uint16_t*)h->d_h, h_layer_stride, h->d_wc16, (long long)L * 2 * 2 * 64 * 4, h->d_bc, feat,
                                     h->F_stride, 2 * c.c_final * L, ns, R, L, H, c.layers, h->d_rowsrc, s);
                else
                    launch_highway(h->d_h, h_layer_stride, h->d_wc, (long long)L * 2 * 2 * 64 * 4, h->d_bc, feat, h->F_stride,
                                   2 * c.c_final * L, ns, R, L, H, c.layers, h->d_rowsrc, s);
                rc = prof_end(h, "highway", s, &ev); if (rc) return rc;
            }
            h->last_chunk_sites = ns;
        }
        EventPair ev{};
        int rc = prof_begin(h, "fc", s, &ev); if (rc) return rc;
        if (h->d_w0x)
            launch_fcx(h->d_feat, h->F_stride, h->d_w0x, h->F_stride, (long long)c.fc_sizes[0] * h->F_stride, h->d_b0, h->d_hid0, h->n0_stride,
                       nb, c.fc_sizes[0], (int)h->F_stride, 1, s, h->d_fc_ws, (long long)2 * h->max_batch * c.fc_sizes[0]);
        else
            launch_fc(h->d_feat, h->F_stride, h->d_w0, h->F_stride, h->d_b0, h->d_hid0, h->n0_stride, nb, c.fc_sizes[0],
                      (int)h->F_stride, 1, s, h->d_fc_ws, (long long)2 * h->max_batch * c.fc_sizes[0]);
        launch_fc(h->d_hid0, h->n0_stride, h->d_w1, h->n0_stride, h->d_b1, h->d_hid1, c.fc_sizes[1], nb, c.fc_sizes[1],
                  h->n0_stride, 1, s);
        launch_heads(h->d_hid1, c.fc_sizes[1], h->d_wh, h->d_bh, nb, bin_logits ? bin_logits + mb * 2 : nullptr,
                     vt_logits ? vt_logits + mb * 3 : nullptr, vt_prob ? vt_prob + mb * 3 : nullptr,
                     bp ? bp + mb : nullptr, aux ? aux + mb * 22 : nullptr, s);
        rc = prof_end(h, "fc", s, &ev); if (rc) return rc;
        h->last_batch = nb;
    }
    HIPCHK(h, hipGetLastError());
    return DAN_OK;
}

int dan_forward_aux(dan_t* h, const uint8_t* reads, const uint8_t* qual, const uint8_t* strand, const uint8_t* ref,
                    const uint8_t* ref_mask, const uint8_t* var_mask, int64_t n_sites, float* bin_logits,
                    float* vt_logits, float* vt_prob, float* bp, float* aux) {
    if (!h) return DAN_ERR_INVALID_ARG;
    if (!h->finalized) return fail(h, DAN_ERR_STATE, "dan_forward before dan_finalize");
    if (n_sites < 0) return fail(h, DAN_ERR_INVALID_ARG, "negative site count");
    if (n_sites == 0) return DAN_OK;
    if (!reads || !qual || !strand || !ref || !ref_mask || !var_mask) return fail(h, DAN_ERR_INVALID_ARG, "null input plane");
    if (h->slot[0].ticket >= 0 || h->slot[1].ticket >= 0)      // (the workspaces are shared and the streams differ)
        return fail(h, DAN_ERR_STATE, "dan_forward while asynchronous batches are in flight: dan_wait for them first");
    const dan_config& c = h->cfg;
    HIPCHK(h, hipSetDevice(c.device_id));
    const size_t rl = (size_t)c.reads * c.length, L = c.length;
    for (int64_t mb = 0; mb < n_sites; mb += h->max_batch) {
        const size_t nb = (size_t)std::min<int64_t>(h->max_batch, n_sites - mb);
        uint8_t* d = h->d_in;
        uint8_t *dr = d, *dq = dr + nb * rl, *ds = dq + nb * rl, *df = ds + nb * rl, *drm = df + nb * L, *dvm = drm + nb * L;
        HIPCHK(h, hipMemcpy(dr, reads + mb * rl, nb * rl, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(dq, qual + mb * rl, nb * rl, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(ds, strand + mb * rl, nb * rl, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(df, ref + mb * L, nb * L, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(drm, ref_mask + mb * L, nb * L, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(dvm, var_mask + mb * L, nb * L, hipMemcpyHostToDevice));
        float *o_bin = h->d_out, *o_vt = o_bin + nb * 2, *o_p = o_vt + nb * 3, *o_bp = o_p + nb * 3, *o_aux = o_bp + nb;
        int rc = dan_forward_device(h, dr, dq, ds, df, drm, dvm, (int64_t)nb, o_bin, o_vt, o_p, o_bp, o_aux, nullptr);
        if (rc) return rc;
        HIPCHK(h, hipDeviceSynchronize());
        if (bin_logits) HIPCHK(h, hipMemcpy(bin_logits + mb * 2, o_bin, nb * 2 * sizeof(float), hipMemcpyDeviceToHost));
        if (vt_logits) HIPCHK(h, hipMemcpy(vt_logits + mb * 3, o_vt, nb * 3 * sizeof(float), hipMemcpyDeviceToHost));
        if (vt_prob) HIPCHK(h, hipMemcpy(vt_prob + mb * 3, o_p, nb * 3 * sizeof(float), hipMemcpyDeviceToHost));
        if (bp) HIPCHK(h, hipMemcpy(bp + mb, o_bp, nb * sizeof(float), hipMemcpyDeviceToHost));
        if (aux) HIPCHK(h, hipMemcpy(aux + mb * 22, o_aux, nb * 22 * sizeof(float), hipMemcpyDeviceToHost));
    }
    return DAN_OK;
}

int dan_forward(dan_t* h, const uint8_t* reads, const uint8_t* qual, const uint8_t* strand, const uint8_t* ref,
                const uint8_t* ref_mask, const uint8_t* var_mask, int64_t n_sites, float* bin_logits, float* vt_logits,
                float* vt_prob, float* bp) {
    return dan_forward_aux(h, reads, qual, strand, ref, ref_mask, var_mask, n_sites, bin_logits, vt_logits, vt_prob, bp, nullptr);
}

static const int OUT_W[5] = {2, 3, 3, 1, 22};            // floats per site of bin_logits, vt_logits, vt_prob, bp, aux

static int async_init(dan_handle* h) {
    const dan_config& c = h->cfg;
    const size_t in_site = (size_t)3 * c.reads * c.length + 3 * c.length;
    HIPCHK(h, hipStreamCreateWithFlags(&h->s_h2d, hipStreamNonBlocking));
    HIPCHK(h, hipStreamCreateWithFlags(&h->s_comp, hipStreamNonBlocking));
    HIPCHK(h, hipStreamCreateWithFlags(&h->s_d2h, hipStreamNonBlocking));
    for (auto& sl : h->slot) {
        HIPCHK(h, hipHostMalloc((void**)&sl.pin_in, (size_t)h->max_batch * in_site, hipHostMallocDefault));
        HIPCHK(h, hipHostMalloc((void**)&sl.pin_out, (size_t)h->max_batch * 31 * sizeof(float), hipHostMallocDefault));
        int rc = dev_alloc(h, &sl.dev_in, (size_t)h->max_batch * in_site); if (rc) return rc;
        rc = dev_alloc(h, &sl.dev_out, (size_t)h->max_batch * 31); if (rc) return rc;
        HIPCHK(h, hipEventCreateWithFlags(&sl.ev_h2d, hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&sl.ev_comp, hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming));
        sl.ev_slice.resize((size_t)(h->max_batch + h->chunk - 1) / h->chunk);
        for (auto& e : sl.ev_slice) HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    h->async_ready = true;
    return DAN_OK;
}

// One conv chunk of an asynchronous batch: caller's pageable planes -> the slot's pinned mirror -> the device, on the copy stream.
// The mirror is filled by up to four threads (one core moves pageable memory at 5-10 GB/s: 118 MB per chunk at 128 x 301).  Nothing
// here allocates: a fixed table of pieces and of threads, so that no std::bad_alloc can leave through the C ABI; a thread that
// cannot be started (std::system_error) leaves its pieces to the calling thread.
struct AsyncStage {
    dan_handle* h; dan_handle::Slot* sl;
    const uint8_t* src[6]; size_t off[6]; size_t per_site[6];
};

static int stage_chunk(void* ctx, int64_t first_site, int n_sites, int64_t k, hipEvent_t* ready) {
    AsyncStage& st = *static_cast<AsyncStage*>(ctx);
    dan_handle* h = st.h;
    dan_handle::Slot& sl = *st.sl;
    if (k < 0 || (size_t)k >= sl.ev_slice.size()) return fail(h, DAN_ERR_STATE, "chunk %lld outside the slot's event table", (long long)k);
    const size_t s0 = (size_t)first_site, ns = (size_t)n_sites;
    constexpr int MAX_THR = 4;
    struct Piece { uint8_t* dst; const uint8_t* src; size_t n; };
    Piece pieces[3 * MAX_THR + 3];
    int n_pieces = 0;
    size_t total = 0;
    for (int i = 0; i < 6; ++i) total += ns * st.per_site[i];
    const int n_thr = total >= ((size_t)32 << 20) ? MAX_THR : 1;
    for (int i = 0; i < 6; ++i) {
        const size_t o = st.off[i] + s0 * st.per_site[i], n = ns * st.per_site[i];
        const int parts = (i < 3) ? n_thr : 1;                   // the three read planes are the bytes; the site planes ride with thread 0
        for (int t = 0; t < parts; ++t) {
            const size_t a0 = n * t / parts, a1 = n * (t + 1) / parts;
            pieces[n_pieces++] = {sl.pin_in + o + a0, st.src[i] + s0 * st.per_site[i] + a0, a1 - a0};
        }
    }
    // pieces 0 .. 3 n_thr - 1 are the read planes' parts (piece j belongs to thread j % n_thr), the rest the site planes
    std::thread workers[MAX_THR];
    bool started[MAX_THR] = {false, false, false, false};
    const Piece* pc = pieces;
    for (int t = 1; t < n_thr; ++t) {
        try {
            workers[t] = std::thread([pc, t, n_thr] { for (int j = t; j < 3 * n_thr; j += n_thr) memcpy(pc[j].dst, pc[j].src, pc[j].n); });
            started[t] = true;
        } catch (...) {}
    }
    for (int j = 0; j < n_pieces; ++j)
        if (j >= 3 * n_thr || !started[j % n_thr]) memcpy(pieces[j].dst, pieces[j].src, pieces[j].n);
    for (int t = 1; t < n_thr; ++t) if (started[t]) workers[t].join();
    for (int i = 0; i < 6; ++i) {
        const size_t o = st.off[i] + s0 * st.per_site[i];
        HIPCHK(h, hipMemcpyAsync(sl.dev_in + o, sl.pin_in + o, ns * st.per_site[i], hipMemcpyHostToDevice, h->s_h2d));
    }
    HIPCHK(h, hipEventRecord(sl.ev_slice[(size_t)k], h->s_h2d));
    *ready = sl.ev_slice[(size_t)k];
    return DAN_OK;
}

int dan_forward_async(dan_t* h, const uint8_t* reads, const uint8_t* qual, const uint8_t* strand, const uint8_t* ref,
                      const uint8_t* ref_mask, const uint8_t* var_mask, int64_t n_sites, float* bin_logits,
                      float* vt_logits, float* vt_prob, float* bp, float* aux, int64_t* ticket) {
    if (!h || !ticket) return DAN_ERR_INVALID_ARG;
    if (!h->finalized) return fail(h, DAN_ERR_STATE, "dan_forward_async before dan_finalize");
    if (n_sites < 0 || n_sites > h->max_batch)
        return fail(h, DAN_ERR_INVALID_ARG, "dan_forward_async takes 0..max_batch (%d) sites per call, got %lld", h->max_batch, (long long)n_sites);
    if (n_sites > 0 && (!reads || !qual || !strand || !ref || !ref_mask || !var_mask)) return fail(h, DAN_ERR_INVALID_ARG, "null input plane");
    const dan_config& c = h->cfg;
    HIPCHK(h, hipSetDevice(c.device_id));
    if (!h->async_ready) { int rc = async_init(h); if (rc) return rc; }
    dan_handle::Slot& sl = h->slot[h->next_ticket & 1];
    if (sl.ticket >= 0)
        return fail(h, DAN_ERR_STATE, "two batches are already in flight: dan_wait(%lld) first", (long long)sl.ticket);
    const size_t nb = (size_t)n_sites, rl = (size_t)c.reads * c.length, L = c.length;
    if (nb) {
        AsyncStage st{};
        st.h = h; st.sl = &sl;
        const uint8_t* src[6] = {reads, qual, strand, ref, ref_mask, var_mask};
        const size_t off[7] = {0, nb * rl, 2 * nb * rl, 3 * nb * rl, 3 * nb * rl + nb * L, 3 * nb * rl + 2 * nb * L, nb * (3 * rl + 3 * L)};
        for (int i = 0; i < 6; ++i) { st.src[i] = src[i]; st.off[i] = off[i]; st.per_site[i] = i < 3 ? rl : L; }
        const ChunkFeed feed{stage_chunk, &st};
        uint8_t* d = sl.dev_in;
        float *o_bin = sl.dev_out, *o_vt = o_bin + nb * 2, *o_p = o_vt + nb * 3, *o_bp = o_p + nb * 3, *o_aux = o_bp + nb;
        int rc = forward_device_impl(h, d + off[0], d + off[1], d + off[2], d + off[3], d + off[4], d + off[5], n_sites, o_bin, o_vt,
                                     o_p, o_bp, o_aux, (void*)h->s_comp, &feed);
        if (rc) return rc;
        HIPCHK(h, hipEventRecord(sl.ev_h2d, h->s_h2d));
        HIPCHK(h, hipEventRecord(sl.ev_comp, h->s_comp));
        HIPCHK(h, hipStreamWaitEvent(h->s_d2h, sl.ev_comp, 0));
        HIPCHK(h, hipMemcpyAsync(sl.pin_out, sl.dev_out, nb * 31 * sizeof(float), hipMemcpyDeviceToHost, h->s_d2h));
        HIPCHK(h, hipEventRecord(sl.ev_done, h->s_d2h));
    }
    sl.ticket = h->next_ticket++;
    sl.n = n_sites;
    sl.dst[0] = bin_logits; sl.dst[1] = vt_logits; sl.dst[2] = vt_prob; sl.dst[3] = bp; sl.dst[4] = aux;
    *ticket = sl.ticket;
    return DAN_OK;
}

int dan_wait(dan_t* h, int64_t ticket) {
    if (!h) return DAN_ERR_INVALID_ARG;
    if (!h->async_ready || ticket < 0) return fail(h, DAN_ERR_STATE, "dan_wait(%lld): no such batch in flight", (long long)ticket);
    dan_handle::Slot& sl = h->slot[ticket & 1];
    if (sl.ticket != ticket) return fail(h, DAN_ERR_STATE, "dan_wait(%lld): no such batch in flight", (long long)ticket);
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const size_t nb = (size_t)sl.n;
    if (nb) {
        HIPCHK(h, hipEventSynchronize(sl.ev_done));
        const float* src = sl.pin_out;
        for (int i = 0; i < 5; ++i) {
            if (sl.dst[i]) memcpy(sl.dst[i], src, nb * OUT_W[i] * sizeof(float));
            src += nb * OUT_W[i];
        }
    }
    sl.ticket = -1;
    return DAN_OK;
}

int dan_set_tap(dan_t* h, int32_t layer) {
    if (!h) return DAN_ERR_INVALID_ARG;
    if (layer < -1 || layer > h->cfg.layers) return fail(h, DAN_ERR_INVALID_ARG, "tap layer %d out of range", layer);
    if (layer >= 0 && !h->d_tap) {
        if (!h->finalized) return fail(h, DAN_ERR_STATE, "dan_set_tap before dan_finalize");
        HIPCHK(h, hipSetDevice(h->cfg.device_id));
        int rc = dev_alloc(h, &h->d_tap, (size_t)h->chunk * h->cfg.reads * h->cfg.length * CPAD);
        if (rc) return rc;
    }
    h->tap_layer = layer;
    return DAN_OK;
}

int64_t dan_query(const dan_t* h, const char* what) {
    if (!h || !what) return DAN_ERR_INVALID_ARG;
    const std::string w(what);
    if (w == "feature_width") return h->F;
    if (w == "feature_stride") return h->F_stride;
    if (w == "chunk_sites") return h->chunk;
    if (w == "chunk_sites_auto") return h->chunk_auto ? 1 : 0;
    if (w == "max_batch") return h->max_batch;
    if (w == "cpad") return CPAD;
    if (w == "tap_sites") return h->last_chunk_sites;
    if (w == "segments") return h->n_segments;
    if (w == "bf16_pingpong") return h->use_p ? 1 : 0;
    if (w == "bf16x3_split_kernel") return h->use_x ? 1 : 0;
    if (w == "hidden0_stride") return h->n0_stride;
    return fail(h, DAN_ERR_INVALID_ARG, "dan_query: unknown key '%s'", what);
}

int64_t dan_read_buffer(dan_t* h, const char* name, float* dst, int64_t capacity) {
    if (!h || !name || !dst || capacity < 0) return DAN_ERR_INVALID_ARG;
    if (!h->finalized) return fail(h, DAN_ERR_STATE, "dan_read_buffer before dan_finalize");
    const dan_config& c = h->cfg;
    const std::string w(name);
    const float* src = nullptr;
    int64_t n = 0;
    if (w == "tap") { if (!h->d_tap) return fail(h, DAN_ERR_STATE, "no tap was requested"); src = h->d_tap; n = (int64_t)h->last_chunk_sites * c.reads * c.length * CPAD; }
    else if (w == "feature") { src = h->d_feat; n = h->last_batch * h->F_stride; }
    else if (w == "hidden0") { src = h->d_hid0; n = h->last_batch * h->n0_stride; }
    else if (w == "hidden1") { src = h->d_hid1; n = h->last_batch * c.fc_sizes[1]; }
    else if (w == "pool") { src = h->d_pool; n = (int64_t)h->last_chunk_sites * c.length * CPAD; }
    else if (w == "h") {
        // bottleneck outputs of the last chunk, [layer][site][read][L][HPAD] (layer stride = the chunk's capacity); bf16 on the
        // ping-pong kernel: widened here
        if (!h->d_h) return fail(h, DAN_ERR_STATE, "the network has no highway bottleneck");
        const int64_t per_layer = (int64_t)h->last_chunk_sites * c.reads * c.length * HPAD;
        const int64_t stride = (int64_t)h->chunk * c.reads * c.length * HPAD;
        if (per_layer == 0) return fail(h, DAN_ERR_STATE, "dan_read_buffer('h') before the first forward");
        if (capacity < per_layer)
            return fail(h, DAN_ERR_INVALID_ARG, "dan_read_buffer('h'): capacity %lld is less than one layer (%lld floats)", (long long)capacity,
                        (long long)per_layer);
        n = std::min<int64_t>(per_layer * c.layers, capacity) / per_layer * per_layer;
        HIPCHK(h, hipSetDevice(c.device_id));
        HIPCHK(h, hipDeviceSynchronize());
        std::vector<uint16_t> tmp(h->use_p ? (size_t)per_layer : 0);
        for (int64_t l = 0; l * per_layer < n; ++l) {
            if (h->use_p) {
                HIPCHK(h, hipMemcpy(tmp.data(), (const uint16_t*)h->d_h + l * stride, (size_t)per_layer * 2, hipMemcpyDeviceToHost));
                for (int64_t i = 0; i < per_layer; ++i) dst[l * per_layer + i] = bf16_float(tmp[(size_t)i]);
            } else {
                HIPCHK(h, hipMemcpy(dst + l * per_layer, h->d_h + l * stride, (size_t)per_layer * sizeof(float), hipMemcpyDeviceToHost));
            }
        }
        return n;
    }
    else return fail(h, DAN_ERR_INVALID_ARG, "dan_read_buffer: unknown buffer '%s'", name);
    n = std::min(n, capacity);
    HIPCHK(h, hipSetDevice(c.device_id));
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return n;
}

int dan_profile_enable(dan_t* h, int32_t on) {
    if (!h) return DAN_ERR_INVALID_ARG;
    h->profiling = on != 0;
    for (auto& kv : h->stats) {
        int rc = prof_collect(h, kv.second);
        if (rc) return rc;
        kv.second.launches = 0;
        kv.second.ms = 0.0;
    }
    return DAN_OK;
}

int dan_kernel_stats(dan_t* h, const char* kernel, int64_t* launches, double* total_ms) {
    if (!h || !kernel) return DAN_ERR_INVALID_ARG;
    KernelStat& st = h->stats[kernel];
    int rc = prof_collect(h, st);
    if (rc) return rc;
    if (launches) *launches = st.launches;
    if (total_ms) *total_ms = st.ms;
    return DAN_OK;
}

}  // extern "C"
