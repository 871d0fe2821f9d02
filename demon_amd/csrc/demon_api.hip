// demon_api.hip -- context, weight packing, network topology and the C ABI of libdemon_hip.so.
//
// The three reference networks (python/depthmotionnet/networks_original.py:22-255) are built here as
// fixed kernel sequences over one device arena: weights are repacked once for the MFMA kernels, every
// activation lives at a fixed address, layer outputs are written straight into the channel slices of
// their concat buffers, and each sequence is captured into a hipGraph.  Nothing is allocated and no
// host round trip happens between the stages of examples/example.py:87-99.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library itself is dlopen'ed on first use (see rccl() below)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <functional>
#include <mutex>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/demon_hip.h"
#include "internal.h"

using namespace demon;

namespace {

thread_local std::string g_create_error;
// kernel family the last run_layer / run_pair call launched ("conv_mfma", "conv_patch", "deconv4", "conv_small", "conv_pair",
// with "+splitk" when a split-K reduce launch follows); demon_profile_full reads it to attribute time per kernel
unsigned long long *g_timeline_dev = nullptr;  // diagnostic build (demon_debug_timeline): where the kernels write their records
thread_local const char *g_last_kernel = nullptr;
thread_local char g_kernel_tag[48];
// suffix of a kernel tag: "+splitk" = a conv_splitk_reduce launch follows
const char *split_suffix(int ksplit) { return ksplit > 1 ? "+splitk" : ""; }
void set_kernel_tag(const char *family, int bm, int bn, int taps, bool splitk)
{
    // e.g. "conv_mfma<128x32>+splitk" (rocprofv3: conv_mfma_kernel<128, 32, ...>), "conv_patch<64x128,t5>", "deconv4<32x128>"
    if (taps > 0) snprintf(g_kernel_tag, sizeof g_kernel_tag, "%s<%dx%d,t%d>%s", family, bm, bn, taps, splitk ? "+splitk" : "");
    else snprintf(g_kernel_tag, sizeof g_kernel_tag, "%s<%dx%d>%s", family, bm, bn, splitk ? "+splitk" : "");
    g_last_kernel = g_kernel_tag;
}

struct Layer {
    std::string name;  // "<scope>/<layer>"
    enum Kind { CONV, DECONV, DENSE } kind = CONV;
    int Cin = 0, Cout = 0, kh = 1, kw = 1, sh = 1, sw = 1, ph = 0, pw = 0, act = 0;
    View in, out;
    const float *scale = nullptr;
    int K = 0, Kpad = 0, Krows = 0, Mpad = 0, ncls = 1;
    float *d_wp = nullptr, *d_bias = nullptr;
    KEntry *d_ktab = nullptr;
    // conv_stream.hip (Cin % 16 == 0): the same weights in MFMA fragment order, refreshed from d_wp whenever that changes, and
    // the page of zeros that taps outside the image read
    float *d_wf = nullptr;
    const float *zero = nullptr;
    mutable bool wf_dirty = true;
    int Cin4() const { return (Cin + 3) / 4 * 4; }
    // 1-D minimal filtering (conv_wino.hip): k x 1 / 1 x k layers with 3 taps stride 1 or 5 / 7 / 9 taps stride 2; U[e][Cin4][Mpad] in d_w1
    float *d_w1 = nullptr;
    mutable bool w1_dirty = true;
    // F(4,3) weights of the 3 x 3 stride-1 layers for conv_wino3.hip: U43[ky][e][Cin4][Mpad], 6 planes per kernel row; for the 3 x 3
    // stride-2 layers (wino3_stride2) the polyphase F(4,2) + F(4,1) weights, 9 planes per kernel row
    float *d_w3 = nullptr;
    mutable bool w3_dirty = true;
    bool wino3_stride2() const { return kind == CONV && !scale && Cin >= 16 && kh == 3 && kw == 3 && sh == 2 && sw == 2 && ph == 1 && pw == 1; }
    // four-outputs-per-window weights of the 3-tap stride-1 / 5-tap stride-2 1-D layers for conv_wino4.hip: U[e][Cin4][Mpad], 6 / 11 planes
    float *d_w4 = nullptr;
    mutable bool w4_dirty = true;
    int wino4_kind_of() const
    {
        if (kind != CONV || scale || Cin < 16 || (kh != 1 && kw != 1) || (kh == 1 && kw == 1)) return -1;
        if ((kw == 1 && sw != 1) || (kh == 1 && sh != 1)) return -1;
        return wino4_kind(kw == 1 ? kh : kw, kw == 1 ? sh : sw);
    }
    // weight-streaming dense kernel (dense_stream.hip): the weights re-blocked to [Mpad / 128][Cin][128]
    float *d_wd = nullptr;
    mutable bool wd_dirty = true;
    int wino1d_axis() const { return kw == 1 ? 0 : 1; }
    int wino1d_cross() const { return (kh == 3 && kw == 3) ? 3 : 1; }   // 3 x 3 stride 1: three 1 x 3 filters summed in the same accumulators
    int wino1d_kind_of() const
    {
        if (kind != CONV || scale || Cin < 16) return -1;
        if (kh == 3 && kw == 3) return (sh == 1 && sw == 1 && ph == 1 && pw == 1) ? 0 : -1;
        if ((kh != 1 && kw != 1) || (kh == 1 && kw == 1)) return -1;
        if (kw == 1 && sw != 1) return -1;
        if (kh == 1 && sh != 1) return -1;
        return wino1d_kind(kw == 1 ? kh : kw, kw == 1 ? sh : sw);
    }
    bool have_kernel = false, have_bias = false;
    std::vector<int64_t> kernel_dims;  // TF layout
    int force_tile = -1, force_split = 0;  // tuning override (demon_bench_layer)
    // autotuned choice per batch size: kind 0 = im2col kernel (tile, ksplit), 1 = patch kernel (patch tile)
    struct Tuned { int kind, tile, ksplit; };
    std::map<int, Tuned> tuned;
    int ntaps() const { return kind == DECONV ? 4 : (kind == DENSE ? 1 : kh * kw); }
    bool stream_ok() const { return d_wf != nullptr && zero != nullptr; }
    static bool stream_shape_ok(Kind kind, int Cin, int kh, int kw) { return Cin % 16 == 0 && (kind != CONV || kh * kw <= 9); }
};

// The launch plan of a layer for batch n: the entry measured at exactly n, else the one of the NEAREST tuned batch size (by ratio) --
// a ragged last batch inside a context tuned for its full batch (runtime.get_context installs one plan) keeps the tuned kernel
// families and variants instead of falling back to the untuned heuristics; kernels clamp a split-K that does not fit.
inline std::map<int, Layer::Tuned>::const_iterator nearest_tuned(const Layer *L, int n)
{
    auto it = L->tuned.find(n);
    if (it != L->tuned.end() || L->tuned.empty()) return it;
    auto best = L->tuned.begin();
    double best_r = 1e30;
    for (auto jt = L->tuned.begin(); jt != L->tuned.end(); ++jt) {
        const double r = jt->first > n ? (double)jt->first / n : (double)n / jt->first;
        if (r < best_r) { best_r = r; best = jt; }
    }
    return best;
}

struct Step {
    std::string name;
    std::string kernel;
    double flops_per_sample = 0, bytes_per_sample = 0, bytes_fixed = 0;
    std::function<void(int n, hipStream_t s)> fn;
    // iterative nets only: 1 = conv1 / conv2 (depend on image_pair and weights only, identical in every iteration);
    // 2 = copy of the cached conv2 output into the concat buffer (runs instead of them when the option is on)
    int image_only = 0;
    // k x 1 / 1 x k pairs that conv_pair.hip can run as one launch exist twice in the list: pair = 1 is the fused step (runs when
    // option fused_pairs is on), pair = 2 marks the two ordinary layer steps (run when it is off)
    int pair = 0;
    // empty: the fused form always applies (conv_pair.hip); else: decided per batch size by the launch plan (chained pairs)
    std::function<bool(int n)> pair_applies;
    // Side branch: small kernel chains that do not depend on the steps that follow them on the main stream (the motion head next
    // to the decoder of the depth+motion block, predict_flow5 -> upsample next to refine4 of the flow block) run on a second
    // stream.  side = 1 marks them; fork = 1 on the first one (side stream waits for everything enqueued so far on the main
    // stream); join = 1 on the first MAIN step that needs their results (main stream waits for the side stream).
    int side = 0, fork = 0, join = 0;
    // extra-input assembly exists twice: fused = 1 is the single fused launch (option fused_inputs, default), fused = 2 marks the
    // chain of stand-alone op launches it replaces (depth_to_flow / warp2d / copy_channels / flow_to_depth / upsample)
    int fused = 0;
};

struct Variable {
    std::string name;
    Layer *layer;
    bool is_bias;
    int64_t dims[4];
    int ndim;
    int64_t count;
};

}  // namespace

struct demon_ctx {
    unsigned long long serial = 0;   // unique per created context (keys of demon_lanes_run_group's graphs: an address can come back)
    int device = 0, max_batch = 0, H = 0, W = 0;
    int variant = 1;  // 1 = networks_original.py / blocks_original.py, 2 = v2/networks.py / v2/blocks.py
    hipStream_t stream = nullptr;
    std::string err;
    std::vector<void *> allocations;
    // poison harness (tests/test_poison_gpu.py; environment DEMON_POISON_GUARD=1 at creation): every device allocation of the context
    // sits flush between two guard zones filled with a quiet-NaN canary -- a read outside a tensor poisons the result, a write outside
    // one is found by check_guards()
    size_t guard_bytes = 0;
    std::vector<std::pair<char *, size_t>> guarded;   // (raw allocation, payload bytes)
    std::map<std::string, View> buffers;
    std::vector<std::unique_ptr<Layer>> layers;
    std::vector<Variable> variables;
    std::map<std::string, int> var_index;
    std::vector<Step> net_boot, net_iter, net_refine;
    int opt_hipgraph = 1, opt_f2d_method = 0, opt_reuse_image = 0;
    std::map<std::string, hipGraphExec_t> graphs;
    // io / state
    View image_pair, image2_2, flowconf5, flowconf2, depth2, normal2, depth0, normal0;
    float *d_rot = nullptr, *d_trans = nullptr, *d_scale = nullptr, *d_motion = nullptr, *d_intrinsics = nullptr;
    float *d_ws = nullptr;  // split-K workspace
    float *d_ws_side = nullptr;  // split-K workspace of the side branch (runs concurrently with the main one)
    hipStream_t side_stream = nullptr;
    std::vector<hipStream_t> tune_streams;   // throughput-mode autotune (option tune_lanes): concurrent replays need streams of their own
    std::vector<hipEvent_t> tune_events;
    int opt_tune_lanes = 1;
    // demon_set_cu_mask: the context's streams are created with hipExtStreamCreateWithCUMask on this mask (bit i -> XCD i % 8, CU slot
    // i / 8 of that XCD); empty = every CU (plain streams).  A lane group can give each lane a share of every XCD (demon_amd/lanes.py)
    std::vector<uint32_t> cu_mask;
    std::map<std::string, hipGraphExec_t> group_graphs;   // demon_lanes_run_group: one graph holding several lanes' passes as parallel branches (owned by lane 0)
    std::vector<hipEvent_t> group_events;
    std::vector<hipStream_t> placeholder_streams;   // demon_lanes_apply / demon_lanes_calibrate: idle streams that shift the lanes' stream -> hardware-queue mapping (owned by lane 0)
    std::vector<hipEvent_t> events;  // fork / join events, one per use inside a sequence
    int opt_side_branches = 1;
    int opt_fused_pairs = 1;  // conv_pair.hip for the pairs conv_pair_applies() selects
    int opt_fused_inputs = 1;  // one launch for the extra-input assembly of the iterative blocks / the refinement input
    // all packed kernels and biases of the networks live in ONE device slab (alloc_weight_slab), so that
    // demon_broadcast_weights is a single RCCL broadcast of device-resident, already packed data
    float *w_slab = nullptr;
    size_t w_slab_floats = 0;
    float *d_zero = nullptr;  // shared zero page of the conv_stream layers
    std::vector<std::pair<Layer *, Layer *>> chain_pairs;  // stride-1 k x 1 / 1 x k pairs that can run as one chained launch
    std::vector<std::pair<Layer *, Layer *>> fused_pairs;  // pairs conv_pair.hip runs as one launch unless the plan of the k x 1 layer is kind 12
};

namespace {

#define HIP_TRY(ctx, expr)                                                                              \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                             \
            return DEMON_ERR_HIP;                                                                       \
        }                                                                                               \
    } while (0)

int fail(demon_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    else g_create_error = msg;
    return code;
}

constexpr unsigned kGuardCanary = 0x7fc0dead;   // a quiet NaN: whatever reads it and computes with it yields NaN

float *dev_alloc(demon_ctx *c, size_t bytes)
{
    void *p = nullptr;
    if (!bytes) bytes = 16;
    if (c->guard_bytes) {   // [guard | payload, flush on both sides (rounded up to 4 bytes) | guard], all of it canary until the owner writes its payload
        const size_t g = c->guard_bytes, pay = (bytes + 3) / 4 * 4;
        if (hipMalloc(&p, 2 * g + pay) != hipSuccess) return nullptr;
        c->allocations.push_back(p);
        if (hipMemsetD32((hipDeviceptr_t)p, (int)kGuardCanary, (2 * g + pay) / 4) != hipSuccess) return nullptr;
        c->guarded.emplace_back((char *)p, pay);
        return (float *)((char *)p + g);
    }
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    c->allocations.push_back(p);
    return (float *)p;
}

// number of guard zones that no longer hold the canary (0 = nothing wrote outside its tensor); `where` names the first one
int check_guards(demon_ctx *c, std::string *where)
{
    int bad = 0;
    const size_t g = c->guard_bytes;
    if (!g) return 0;
    std::vector<unsigned> host(g / 4);
    for (size_t i = 0; i < c->guarded.size(); ++i)
        for (int side = 0; side < 2; ++side) {
            const char *zone = c->guarded[i].first + (side ? g + c->guarded[i].second : 0);
            if (hipMemcpy(host.data(), zone, g, hipMemcpyDeviceToHost) != hipSuccess) { if (where && !bad) *where = "hipMemcpy of a guard zone failed"; return bad + 1; }
            for (size_t w = 0; w < host.size(); ++w)
                if (host[w] != kGuardCanary) {
                    if (where && !bad)
                        *where = "allocation #" + std::to_string(i) + " (" + std::to_string(c->guarded[i].second) + " bytes): word " + std::to_string(w) +
                                 (side ? " BEHIND its end" : " of the zone IN FRONT of it") + " was overwritten";
                    ++bad;
                    break;
                }
        }
    return bad;
}

size_t guard_bytes_from_env()
{
    const char *e = getenv("DEMON_POISON_GUARD");
    if (!e || !*e || !atoi(e)) return 0;
    return (size_t)4 << 20;   // 4 MiB on each side: more than any kernel's run-ahead (16 weight rows; 8 channel planes of a 192 x 256 map = 1.5 MiB)
}

// split-K workspace: kSplitKWorkspaceFloats of partial sums [cls][slice][Mpad][P], finished by conv_splitk_reduce
float *alloc_splitk_workspace(demon_ctx *c);

// named activation buffer, sized for max_batch; same name -> same memory (the five sub-nets run one
// after the other on one stream and share their encoder/decoder buffers)
View buffer(demon_ctx *c, const std::string &key, int C, int H, int W)
{
    auto it = c->buffers.find(key);
    if (it != c->buffers.end()) return it->second;
    View v;
    // tensors stay below 2^31 - 2^24 elements: buffer-resource extents (internal.h: rsrc_bytes) and the kernels' per-lane offsets are 32-bit
    if ((size_t)c->max_batch * C * H * W + 8ul * H * W >= (1ul << 31) - (1ul << 24)) {
        if (c->err.empty()) c->err = "max_batch too large: activation buffer '" + key + "' would exceed 2^31 elements";
        return v;
    }
    // + 8 planes of slack: the patch kernel's last channel chunk may read (and discard) up to CKS-1 planes
    // past the last channel of the last sample
    // (poison harness: no slack -- the tensor ends flush against the canary zone, so a kernel that USES what it reads there is found)
    v.base = dev_alloc(c, sizeof(float) * ((size_t)c->max_batch * C * H * W + (c->guard_bytes ? 0ul : 8ul * H * W)));
    v.Ctot = C; v.c0 = 0; v.C = C; v.H = H; v.W = W;
    c->buffers[key] = v;
    return v;
}

int round_up(int x, int m) { return (x + m - 1) / m * m; }

void add_variable(demon_ctx *c, Layer *L, bool is_bias)
{
    Variable v;
    v.name = L->name + (is_bias ? "/bias" : "/kernel");
    v.layer = L;
    v.is_bias = is_bias;
    v.ndim = 0;
    v.count = 1;
    if (is_bias) {
        v.dims[v.ndim++] = L->Cout;
    } else {
        for (int64_t d : L->kernel_dims) v.dims[v.ndim++] = d;
    }
    for (int i = 0; i < v.ndim; ++i) v.count *= v.dims[i];
    c->var_index[v.name] = (int)c->variables.size();
    c->variables.push_back(v);
}

// ---- layer planning (geometry, K table) ------------------------------------------------------------
bool plan_layer(demon_ctx *c, Layer *L, bool alloc_weights = true)
{
    const int H = L->in.H, W = L->in.W;
    std::vector<KEntry> tab;
    if (L->kind == Layer::CONV) {
        L->K = L->kh * L->kw * L->Cin;
        L->ncls = 1;
    } else if (L->kind == Layer::DECONV) {
        L->K = 4 * L->Cin;
        L->ncls = 4;
    } else {
        L->K = L->Cin;
        L->ncls = 1;
    }
    L->Kpad = round_up(L->K, 16);
    L->Mpad = round_up(L->Cout, 32);
    tab.assign((size_t)L->ncls * L->Kpad, KEntry{0, (int)((unsigned)(-30000) << 16)});
    auto pack = [](int dy, int dx) { return (int)(((unsigned)dy << 16) | ((unsigned)dx & 0xffffu)); };
    if (L->kind == Layer::CONV) {
        for (int a = 0; a < L->kh; ++a)
            for (int b = 0; b < L->kw; ++b)
                for (int ci = 0; ci < L->Cin; ++ci) {
                    const int k = (a * L->kw + b) * L->Cin + ci;
                    const int dy = a - L->ph, dx = b - L->pw;
                    tab[k] = KEntry{ci * H * W + dy * W + dx, pack(dy, dx)};
                }
    } else if (L->kind == Layer::DECONV) {
        // output (2y+py, 2x+px) of the cropped k4 s2 transposed conv reads input rows
        //   py = 0: taps a = 1 (dy = 0), a = 3 (dy = -1);   py = 1: a = 0 (dy = +1), a = 2 (dy = 0)
        static const int tap_d[2][2] = {{0, -1}, {1, 0}};
        for (int cls = 0; cls < 4; ++cls) {
            const int py = cls >> 1, px = cls & 1;
            for (int ty = 0; ty < 2; ++ty)
                for (int tx = 0; tx < 2; ++tx)
                    for (int ci = 0; ci < L->Cin; ++ci) {
                        const int k = (ty * 2 + tx) * L->Cin + ci;
                        const int dy = tap_d[py][ty], dx = tap_d[px][tx];
                        tab[(size_t)cls * L->Kpad + k] = KEntry{ci * H * W + dy * W + dx, pack(dy, dx)};
                    }
        }
    } else {
        for (int k = 0; k < L->K; ++k) tab[k] = KEntry{k, pack(0, 0)};
    }
    L->d_ktab = (KEntry *)dev_alloc(c, tab.size() * sizeof(KEntry));
    L->Krows = L->Kpad + 16;  // 16 zero rows of slack: the patch kernel's last channel chunk may read past K
    if (!L->d_ktab) return false;
    if (hipMemcpy(L->d_ktab, tab.data(), tab.size() * sizeof(KEntry), hipMemcpyHostToDevice) != hipSuccess) return false;
    if (Layer::stream_shape_ok(L->kind, L->Cin, L->kh, L->kw) && !getenv("DEMON_NO_STREAM")) {
        L->d_wf = dev_alloc(c, sizeof(float) * (size_t)L->ncls * L->K * L->Mpad);
        L->wf_dirty = true;
        if (alloc_weights) {  // stand-alone layer (demon_op_*, demon_bench_layer): its own zero page
            const size_t zf = (size_t)(L->Cin + 2) * H * W;
            float *z = dev_alloc(c, sizeof(float) * zf);
            if (z && hipMemset(z, 0, sizeof(float) * zf) == hipSuccess) L->zero = z;
        }
    }
    if (L->wino1d_kind_of() >= 0 && !getenv("DEMON_NO_WINO")) {
        const size_t nu = ((size_t)L->wino1d_cross() * wino1d_nuv(L->wino1d_kind_of()) * L->Cin4() + kWinoWeightSlackRows) * L->Mpad;   // + 16 rows: a K-step of KG groups reads 4 KG - 4 rows past Cin4 at most (KG <= 4), times zero inputs
        L->d_w1 = dev_alloc(c, sizeof(float) * nu);
        if (!L->d_w1 || hipMemset(L->d_w1, 0, sizeof(float) * nu) != hipSuccess) return false;
        L->w1_dirty = true;
        if (L->wino4_kind_of() >= 0) {
            const size_t n4 = ((size_t)wino4_nuv(L->wino4_kind_of()) * L->Cin4() + kWinoWeightSlackRows) * L->Mpad;
            L->d_w4 = dev_alloc(c, sizeof(float) * n4);
            if (!L->d_w4 || hipMemset(L->d_w4, 0, sizeof(float) * n4) != hipSuccess) return false;
            L->w4_dirty = true;
        }
        if (L->wino1d_cross() == 3 && L->in.W % 4 == 0 && L->in.W >= 32) {
            const size_t n3 = ((size_t)3 * 6 * L->Cin4() + kWinoWeightSlackRows) * L->Mpad;
            L->d_w3 = dev_alloc(c, sizeof(float) * n3);
            if (!L->d_w3 || hipMemset(L->d_w3, 0, sizeof(float) * n3) != hipSuccess) return false;
            L->w3_dirty = true;
        }
    }
    if (L->wino3_stride2() && L->in.W % 8 == 0 && L->in.W >= 64 && !getenv("DEMON_NO_WINO")) {
        const size_t n3 = ((size_t)3 * 9 * L->Cin4() + kWinoWeightSlackRows) * L->Mpad;
        L->d_w3 = dev_alloc(c, sizeof(float) * n3);
        if (!L->d_w3 || hipMemset(L->d_w3, 0, sizeof(float) * n3) != hipSuccess) return false;
        L->w3_dirty = true;
    }
    if (L->kind == Layer::DENSE && L->in.H * L->in.W == 1 && L->out.H * L->out.W == 1 && !L->scale && dense_stream_geometry_ok(L->Cin, L->Mpad, 1) &&
        (long)L->Cin * L->Mpad >= (1l << 20)) {   // >= 4 MB of weights: below that a dense layer is launch bound whatever streams them
        L->d_wd = dev_alloc(c, sizeof(float) * (size_t)L->Cin * L->Mpad);
        if (!L->d_wd) return false;
        L->wd_dirty = true;
    }
    if (!alloc_weights) return true;  // network layers: alloc_weight_slab() places d_wp / d_bias once all layers exist
    L->d_wp = dev_alloc(c, sizeof(float) * (size_t)L->ncls * L->Krows * L->Mpad);
    L->d_bias = dev_alloc(c, sizeof(float) * L->Mpad);
    if (!L->d_wp || !L->d_bias) return false;
    if (hipMemset(L->d_wp, 0, sizeof(float) * (size_t)L->ncls * L->Krows * L->Mpad) != hipSuccess) return false;
    if (hipMemset(L->d_bias, 0, sizeof(float) * L->Mpad) != hipSuccess) return false;
    return true;
}

// One slab for the packed kernels and biases of every network layer (zero-filled: padded rows / columns stay zero).
bool alloc_weight_slab(demon_ctx *c)
{
    auto a64 = [](size_t n) { return (n + 63) / 64 * 64; };  // 256-byte aligned pieces
    size_t total = 0;
    for (auto &L : c->layers) total += a64((size_t)L->ncls * L->Krows * L->Mpad) + a64(L->Mpad);
    c->w_slab = dev_alloc(c, sizeof(float) * total);
    if (!c->w_slab || hipMemset(c->w_slab, 0, sizeof(float) * total) != hipSuccess) return false;
    c->w_slab_floats = total;
    size_t zf = 0;
    for (auto &L : c->layers)
        if (L->d_wf) zf = std::max(zf, (size_t)(L->Cin + 2) * L->in.H * L->in.W);
    if (zf) {
        c->d_zero = dev_alloc(c, sizeof(float) * zf);
        if (!c->d_zero || hipMemset(c->d_zero, 0, sizeof(float) * zf) != hipSuccess) return false;
        for (auto &L : c->layers) if (L->d_wf) L->zero = c->d_zero;
    }
    size_t off = 0;
    for (auto &L : c->layers) {
        L->d_wp = c->w_slab + off;
        off += a64((size_t)L->ncls * L->Krows * L->Mpad);
        L->d_bias = c->w_slab + off;
        off += a64(L->Mpad);
    }
    return true;
}

// TF layout -> packed [cls][Kpad][Mpad] (host side, then one upload)
int upload_kernel(demon_ctx *c, Layer *L, const float *w)
{
    std::vector<float> wp((size_t)L->ncls * L->Krows * L->Mpad, 0.0f);
    if (L->kind == Layer::CONV || L->kind == Layer::DENSE) {
        // HWIO [kh][kw][Cin][Cout] (dense: [in][out]) is already [k][co] with k = (a*kw+b)*Cin+ci
        for (int k = 0; k < L->K; ++k) memcpy(&wp[(size_t)k * L->Mpad], w + (size_t)k * L->Cout, sizeof(float) * L->Cout);
    } else {
        static const int tap_a[2][2] = {{1, 3}, {0, 2}};
        for (int cls = 0; cls < 4; ++cls) {
            const int py = cls >> 1, px = cls & 1;
            for (int ty = 0; ty < 2; ++ty)
                for (int tx = 0; tx < 2; ++tx) {
                    const int a = tap_a[py][ty], b = tap_a[px][tx];
                    for (int ci = 0; ci < L->Cin; ++ci) {
                        const int k = (ty * 2 + tx) * L->Cin + ci;
                        float *dst = &wp[((size_t)cls * L->Krows + k) * L->Mpad];
                        for (int co = 0; co < L->Cout; ++co) dst[co] = w[(((size_t)a * 4 + b) * L->Cout + co) * L->Cin + ci];
                    }
                }
        }
    }
    HIP_TRY(c, hipMemcpy(L->d_wp, wp.data(), wp.size() * sizeof(float), hipMemcpyHostToDevice));
    L->have_kernel = true;
    L->wf_dirty = true;
    L->w1_dirty = true;
    L->w3_dirty = true;
    L->w4_dirty = true;
    L->wd_dirty = true;
    return DEMON_OK;
}


float *alloc_splitk_workspace(demon_ctx *c)
{
    return dev_alloc(c, sizeof(float) * kSplitKWorkspaceFloats);
}

void fill_conv_args(const Layer *L, int n, float *ws, ConvArgs &a)
{
    a.ws = ws;
    a.in = L->in.ptr();
    a.out = L->out.ptr();
    a.wp = L->d_wp;
    a.bias = L->d_bias;
    a.ktab = L->d_ktab;
    a.scale = L->scale;
    a.N = n;
    a.H = L->in.H;
    a.W = L->in.W;
    a.in_n_stride = L->in.n_stride();
    a.Cout = L->Cout;
    a.Mpad = L->Mpad;
    a.Kpad = L->Kpad;
    a.Ho = L->out.H;
    a.Wo = L->out.W;
    a.out_n_stride = L->out.n_stride();
    a.out_plane = (long)L->out.H * L->out.W;
    a.act = L->act;
    a.cls_w_stride = (long)L->Krows * L->Mpad;
    a.ksplit = 1;
    a.tl = g_timeline_dev;
    static const int xcd_order = getenv("DEMON_XCD_ORDER") ? atoi(getenv("DEMON_XCD_ORDER")) : 1;
    a.xcd = xcd_order;
    if (L->kind == Layer::DECONV) {
        a.Hp = L->in.H; a.Wp = L->in.W; a.sy = 1; a.sx = 1; a.osy = 2; a.osx = 2;
    } else {
        a.Hp = L->out.H; a.Wp = L->out.W; a.sy = L->sh; a.sx = L->sw; a.osy = 1; a.osx = 1;
    }
}

// Geometry of the patch-staged kernel for one layer and batch size: picks the Cout x pixel tile, the pixel
// tile shape (G images x TH rows x TW cols) with the least staged input per output pixel, and split-K.
struct PatchPlan { bool ok = false; int tile = 0, ntaps = 0; PatchArgs a; };

// only_tw >= 0 pins the pixel tile shape (index into the TW candidates below) instead of the cost model's choice: the autotuner
// measures them (a plan stores it as ksplit + 1000 * (only_tw + 1))
bool plan_patch(const Layer *L, int n, float *ws, PatchPlan &pp, int only_tile = -1, int only_tw = -1)
{
    static const int enabled = getenv("DEMON_CONV_PATCH") ? atoi(getenv("DEMON_CONV_PATCH")) : 1;
    pp.ok = false;
    if (!enabled || L->kind == Layer::DENSE) return false;
    const int ntaps = L->kind == Layer::DECONV ? 4 : L->kh * L->kw;
    if (L->kind == Layer::CONV && L->kh > 1 && L->kw > 1 && !(L->kh == 3 && L->kw == 3)) return false;
    int chunks = 0;
    ConvArgs ca;
    fill_conv_args(L, n, ws, ca);
    const int Hp = ca.Hp, Wp = ca.Wp, sh = ca.sy, sw = ca.sx;
    int ext_y, ext_x;  // tap extent
    if (L->kind == Layer::DECONV) { ext_y = 2; ext_x = 2; } else { ext_y = L->kh; ext_x = L->kw; }
    float best_cost = 1e30f;
    for (int tile = 0; tile < PTILE_COUNT; ++tile) {
        if (only_tile >= 0 && tile != only_tile) continue;
        const int bm = patch_tile_bm(tile), bn = patch_tile_bn(tile), nt = patch_tile_threads(tile);
        if (L->Mpad % bm) continue;
        const bool dc4 = patch_tile_is_dc4(tile);  // all four classes of a transposed conv in one workgroup
        if (dc4 && (L->kind != Layer::DECONV || only_tile < 0)) continue;  // chosen by the autotuner / a plan only
        if (dc4) { ext_y = 3; ext_x = 3; } else if (L->kind == Layer::DECONV) { ext_y = 2; ext_x = 2; }
        // the 16-row MFMA tile is for Cout <= 16 heads, and there it replaces the 32-row tiles (half of their MFMA rows are zeros)
        const bool m16_ok = L->Cout <= 16 && patch_cks(ntaps, PTILE_16x128) != 0;
        if (only_tile < 0 && (bm == 16) != m16_ok) continue;
        if (bm == 16 && L->Cout > 16) continue;
        const int tcks = patch_cks(ntaps, tile);
        if (!tcks) continue;
        const int tchunks = (L->Cin + tcks - 1) / tcks;
        if ((float)L->Cin / (tchunks * tcks) < 0.7f) continue;  // too much zero padding in K: the im2col kernel is better
        for (int tw_sel = 0; tw_sel < 4; ++tw_sel) {
            if (only_tw >= 0 && tw_sel != only_tw) continue;
            int TW = tw_sel == 0 ? (Wp < bn ? Wp : bn) : (bn >> tw_sel);  // Wp/bn, bn/2, bn/4, bn/8
            if (TW < 8 || TW > Wp || TW > bn) continue;
            int TH = bn / TW;
            if (TH > Hp) TH = Hp;
            int G = 1;
            if (TH == Hp && TW == Wp) { G = bn / (TH * TW); if (G < 1) G = 1; if (G > n) G = n; }
            const int PH = (TH - 1) * sh + ext_y, PW = (TW - 1) * sw + ext_x, PS = PH * PW;
            const long elems = (long)G * tcks * PS;
            if ((elems + nt - 1) / nt > PATCH_EPT) continue;
            if (patch_lds_bytes(tile, ntaps, G, PS) > 64 * 1024) continue;
            const int tiles_y = (Hp + TH - 1) / TH, tiles_x = (Wp + TW - 1) / TW, groups = (n + G - 1) / G;
            const double useful = (double)n * Hp * Wp;
            const double tiles = (double)groups * tiles_y * tiles_x;
            const double mfma_waste = tiles * bn / useful;          // >= 1: padded MFMA columns
            const double staged = tiles * elems / useful / tcks;    // staged input floats per output pixel per channel
            // cost model: MFMA time dominates; staging adds ~ (staged / ntaps) relative to one MFMA column, and small Cout
            // tiles pay more for it; calibrated loosely on layer sweeps
            const float cost = (float)(mfma_waste * (1.0 + 0.35 * staged / ntaps * (128.0 / bm)) * (bm == 128 ? 1.0 : (bm == 64 ? 1.04 : 1.10)));
            if (cost < best_cost) {
                best_cost = cost;
                pp.tile = tile;
                chunks = tchunks;
                PatchArgs &a = pp.a;
                a.G = G; a.TH = TH; a.TW = TW; a.tiles_y = tiles_y; a.tiles_x = tiles_x; a.PH = PH; a.PW = PW; a.PS = PS;
                pp.ok = true;
            }
        }
    }
    if (!pp.ok) return false;
    pp.ntaps = ntaps;
    PatchArgs &a = pp.a;
    a.in = ca.in; a.out = ca.out; a.wp = ca.wp; a.bias = ca.bias; a.scale = ca.scale; a.ws = ws;
    a.N = n; a.Cin = L->Cin; a.H = ca.H; a.W = ca.W; a.in_n_stride = ca.in_n_stride;
    a.Hp = Hp; a.Wp = Wp; a.sh = sh; a.sw = sw;
    a.Cout = L->Cout; a.Mpad = L->Mpad; a.cls_w_stride = ca.cls_w_stride;
    a.Ho = ca.Ho; a.Wo = ca.Wo; a.out_n_stride = ca.out_n_stride; a.osy = ca.osy; a.osx = ca.osx;
    a.act = L->act; a.nsteps_total = chunks; a.xcd = ca.xcd; a.tl = ca.tl;
    const bool fused_classes = patch_tile_is_dc4(pp.tile);
    {
        // LDS layout of the patch: row pitch PWL >= PW and plane stride PS >= PH * PWL chosen so that the 32 lanes of one
        // B-fragment read (32 pixels of the tile; 16 pixels x 2 channels for the 16-row MFMA tile) fall into as many different
        // banks as possible.  With the natural pitch a tile narrower than 32 pixels puts its rows PW = TW + halo floats apart and
        // neighbouring rows collide (rocprofv3 SQ_LDS_BANK_CONFLICT: 27-35 % of the LDS cycles of this kernel).
        static const int pad_enabled = getenv("DEMON_LDS_PAD") ? atoi(getenv("DEMON_LDS_PAD")) : 1;
        const bool m16 = patch_tile_bm(pp.tile) == 16;
        const int pxw = m16 ? 16 : 32, cks = patch_cks(ntaps, pp.tile);
        const int tile_pixels = a.G * a.TH * a.TW, blocks = (tile_pixels + pxw - 1) / pxw;
        // the search is pure host arithmetic but runs on every eager launch: memoised per geometry
        static std::mutex pad_mutex;
        static std::map<std::array<int, 10>, std::pair<int, int>> pad_cache;
        const std::array<int, 10> key = {a.TH, a.TW, a.G, a.PW, a.PH, a.sh, a.sw, m16 ? 1 : 0, cks, pp.tile};
        int best_p = 0, best_q = 0;
        bool cached = false;
        {
            std::lock_guard<std::mutex> lock(pad_mutex);
            auto it = pad_cache.find(key);
            if (it != pad_cache.end()) { best_p = it->second.first; best_q = it->second.second; cached = true; }
        }
        if (!cached && pad_enabled) {
            long best = -1;
            const int qmax = (m16 || a.G > 1) ? 32 : 1;
            for (int q = 0; q < qmax; ++q)
                for (int p = 0; p < 32; ++p) {
                    const int PWL = a.PW + p, PS = a.PH * PWL + q;
                    if (patch_lds_bytes(pp.tile, ntaps, a.G, PS) > 64 * 1024) continue;
                    long cost = 0;
                    for (int b = 0; b < blocks; ++b) {
                        int cnt[32] = {0};
                        int worst = 0;
                        for (int l = 0; l < 32; ++l) {
                            int pix = b * pxw + l % pxw;
                            const int k = m16 ? l / 16 : 0;
                            if (pix >= tile_pixels) pix = 0;
                            const int g = pix / (a.TH * a.TW), rem = pix - g * (a.TH * a.TW);
                            const int py = rem / a.TW, px = rem - py * a.TW;
                            const int addr = g * cks * PS + py * a.sh * PWL + px * a.sw + k * PS;
                            if (++cnt[addr & 31] > worst) worst = cnt[addr & 31];
                        }
                        cost += worst;
                    }
                    const long score = cost * 100000 + (long)a.G * cks * PS;  // fewest conflicts first, then the smallest patch
                    if (best < 0 || score < best) { best = score; best_p = p; best_q = q; }
                }
            std::lock_guard<std::mutex> lock(pad_mutex);
            pad_cache[key] = {best_p, best_q};
        }
        a.PWL = a.PW + best_p;
        a.PS = a.PH * a.PWL + best_q;

    }
    for (int cls = 0; cls < 4; ++cls) {
        if (L->kind == Layer::DECONV) {
            static const int tap_d[2][2] = {{0, -1}, {1, 0}};
            const int py = cls >> 1, px = cls & 1;
            a.oy0[cls] = fused_classes ? -1 : (py ? 0 : -1);  // fused: one union patch with origin (-1, -1) for all classes
            a.ox0[cls] = fused_classes ? -1 : (px ? 0 : -1);
            for (int ty = 0; ty < 2; ++ty)
                for (int tx = 0; tx < 2; ++tx)
                    a.tapoff[cls][ty * 2 + tx] = (tap_d[py][ty] - a.oy0[cls]) * a.PWL + (tap_d[px][tx] - a.ox0[cls]);
        } else {
            a.oy0[cls] = -L->ph;
            a.ox0[cls] = -L->pw;
            for (int ta = 0; ta < L->kh; ++ta)
                for (int tb = 0; tb < L->kw; ++tb) a.tapoff[cls][ta * L->kw + tb] = ta * a.PWL + tb;
        }
    }
    // split-K over channel chunks when the grid is too small to fill the chip
    const long groups = (n + a.G - 1) / a.G;
    const long wgs = groups * a.tiles_y * a.tiles_x * patch_tile_mtiles(pp.tile, L->Cout, L->Mpad) * (fused_classes ? 1 : L->ncls);
    int split = 1;
    if (wgs < 384) {
        split = (int)((512 + wgs - 1) / wgs);
        const int smax = chunks / 4 > 1 ? chunks / 4 : 1;
        if (split > smax) split = smax;
        const long P = (long)n * Hp * Wp;
        while (split > 1 && (!ws || (long)L->ncls * split * L->Mpad * P > kSplitKWorkspaceFloats)) --split;
    }
    a.ksplit = split;
    auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };  // 0 <=> d == 1
    a.m_plane = magic(a.PH * a.PW); a.m_pw = magic(a.PW); a.m_thtw = magic(a.TH * a.TW); a.m_tw = magic(a.TW);
    a.m_tilesx = magic(a.tiles_x); a.m_tilesy = magic(a.tiles_y);
    return true;
}

void launch_patch_plan(const Layer *L, PatchPlan &pp, ConvArgs &a, long P, float *ws, hipStream_t s)
{
    while (pp.a.ksplit > 1 && (!ws || (long)L->ncls * pp.a.ksplit * L->Mpad * P > kSplitKWorkspaceFloats)) --pp.a.ksplit;
    launch_conv_patch(pp.a, pp.tile, pp.ntaps, L->ncls, s);
    set_kernel_tag(patch_tile_is_dc4(pp.tile) ? "deconv4" : "conv_patch", patch_tile_bm(pp.tile), patch_tile_bn(pp.tile),
                   patch_tile_is_dc4(pp.tile) ? 0 : pp.ntaps, pp.a.ksplit > 1);
    if (pp.a.ksplit > 1) {
        a.ksplit = pp.a.ksplit;
        launch_splitk_reduce(a, L->ncls, s);
    }
}

bool small_applies(const Layer *L)
{
    return L->kind == Layer::CONV && conv_small_applies(L->kh, L->kw, L->sh, L->sw, L->Cin, L->Cout);
}

void run_small(const Layer *L, const ConvArgs &a, hipStream_t s)
{
    SmallConvArgs sa;
    sa.in = a.in; sa.out = a.out; sa.wp = a.wp; sa.bias = a.bias; sa.scale = a.scale;
    sa.Cin = L->Cin; sa.Cout = L->Cout; sa.Mpad = L->Mpad; sa.H = a.H; sa.W = a.W; sa.act = a.act;
    sa.in_n_stride = a.in_n_stride; sa.out_n_stride = a.out_n_stride;
    launch_conv_small(sa, a.N, s);
    g_last_kernel = "conv_small";
}

// d_wp -> fragment order for conv_stream.hip; outside graph capture for network layers (prepare_stream_weights), on the spot
// for stand-alone layers
void refresh_stream_weights(const Layer *L, hipStream_t s)
{
    if (L->d_w1 && L->w1_dirty) {   // transformed weights of the 1-D minimal-filtering kernel (conv_wino.hip)
        launch_wino1d_repack(L->d_w1, L->d_wp, L->wino1d_kind_of(), L->Cin, L->Cin4(), L->Mpad, L->wino1d_cross(), s);
        L->w1_dirty = false;
    }
    if (L->d_w3 && L->w3_dirty) {   // F(4,3) weights of conv_wino3.hip
        launch_wino3_repack43(L->d_w3, L->d_wp, L->Cin, L->Cin4(), L->Mpad, L->wino3_stride2() ? 2 : 1, s);
        L->w3_dirty = false;
    }
    if (L->d_w4 && L->w4_dirty) {   // four-outputs-per-window weights of conv_wino4.hip
        launch_wino4_repack(L->d_w4, L->d_wp, L->wino4_kind_of(), L->Cin, L->Cin4(), L->Mpad, s);
        L->w4_dirty = false;
    }
    if (L->d_wd && L->wd_dirty) {   // re-blocked weights of the weight-streaming dense kernel (dense_stream.hip)
        launch_dense_repack(L->d_wd, L->d_wp, L->Cin, L->Mpad, s);
        L->wd_dirty = false;
    }
    if (!L->d_wf || !L->wf_dirty) return;
    launch_stream_repack(L->d_wf, L->d_wp, L->ncls, L->K, L->Mpad, (long)L->Krows * L->Mpad, s);
    L->wf_dirty = false;
}

int fill_stream_args(const Layer *L, const ConvArgs &a, int ksplit, StreamArgs &sa, hipStream_t s)
{
    refresh_stream_weights(L, s);
    sa.c = a;
    sa.wf = L->d_wf;
    sa.zero = L->zero;
    sa.dbg = getenv("DEMON_STREAM_DBG") ? atoi(getenv("DEMON_STREAM_DBG")) : 0;
    sa.cls_wf_stride = (long)L->K * L->Mpad;
    sa.ntaps = L->ntaps();
    sa.csteps = L->Cin / 16;
    sa.nsteps = sa.ntaps * sa.csteps;
    memset(sa.tapdy, 0, sizeof sa.tapdy);
    memset(sa.tapdx, 0, sizeof sa.tapdx);
    if (L->kind == Layer::DECONV) {
        static const int tap_d[2][2] = {{0, -1}, {1, 0}};  // as plan_layer's K table
        for (int cls = 0; cls < 4; ++cls)
            for (int ty = 0; ty < 2; ++ty)
                for (int tx = 0; tx < 2; ++tx) {
                    sa.tapdy[cls][ty * 2 + tx] = tap_d[cls >> 1][ty];
                    sa.tapdx[cls][ty * 2 + tx] = tap_d[cls & 1][tx];
                }
    } else if (L->kind == Layer::CONV) {
        for (int ta = 0; ta < L->kh; ++ta)
            for (int tb = 0; tb < L->kw; ++tb) {
                sa.tapdy[0][ta * L->kw + tb] = ta - L->ph;
                sa.tapdx[0][ta * L->kw + tb] = tb - L->pw;
            }
    }
    if (ksplit < 1) ksplit = 1;
    if (ksplit > sa.nsteps) ksplit = sa.nsteps;
    return ksplit;
}

void run_stream(const Layer *L, const ConvArgs &a, int variant, int ksplit, hipStream_t s)
{
    StreamArgs sa;
    ksplit = fill_stream_args(L, a, ksplit, sa, s);
    launch_conv_stream(sa, variant, ksplit, L->ncls, s);
    // e.g. "conv_stream<256x32,w4k1>": tile, waves along Cout x K-splitting wave groups (rocprofv3: conv_stream_kernel<NW, TM, TN, KW>)
    snprintf(g_kernel_tag, sizeof g_kernel_tag, "conv_stream<%dx%d,w%dk%d>%s", stream_variant_bm(variant), stream_variant_bn(variant),
             stream_variant_waves(variant) / stream_variant_kw(variant), stream_variant_kw(variant), split_suffix(ksplit));
    g_last_kernel = g_kernel_tag;
}

void run_frag(const Layer *L, const ConvArgs &a, int variant, int ksplit, hipStream_t s)
{
    StreamArgs sa;
    ksplit = fill_stream_args(L, a, ksplit, sa, s);
    launch_conv_frag(sa, variant, ksplit, L->ncls, s);
    snprintf(g_kernel_tag, sizeof g_kernel_tag, "conv_frag<%dx%d,v%d>%s", frag_variant_bm(variant), frag_variant_bn(variant), variant, split_suffix(ksplit));
    g_last_kernel = g_kernel_tag;
}

// minimal-filtering transposed conv (conv_wino.hip), plan kind 8: variant = tiles per workgroup, ksplit slices over the input channels
bool wino_applies(const Layer *L) { return L->kind == Layer::DECONV && L->Cin >= 16 && !L->scale; }

bool run_wino(const Layer *L, const ConvArgs &a, int variant, int ksplit, hipStream_t s)
{
    WinoArgs w;
    w.in = a.in; w.out = a.out; w.wp = a.wp; w.bias = a.bias; w.ws = a.ws;
    w.N = a.N; w.Cin = L->Cin; w.H = a.H; w.W = a.W; w.in_n_stride = a.in_n_stride;
    w.Cout = L->Cout; w.Mpad = L->Mpad; w.cls_w_stride = a.cls_w_stride;
    w.Ho = a.Ho; w.Wo = a.Wo; w.out_n_stride = a.out_n_stride; w.out_plane = a.out_plane;
    w.act = a.act; w.xcd = a.xcd;
    w.nsteps_total = (L->Cin + 3) / 4;
    if (!wino_plan_geometry(w, variant, a.N)) return false;
    if (ksplit < 1 || wino_variant_kh(variant) == 2) ksplit = 1;   // (variant 6 splits the reduction inside the workgroup: nothing across workgroups on top)
    if (ksplit > w.nsteps_total) ksplit = w.nsteps_total;
    w.ksplit = ksplit;
    launch_wino_deconv(w, variant, s);
    snprintf(g_kernel_tag, sizeof g_kernel_tag, "wino_deconv<%dx%d%s>%s", 16 * wino_variant_mb(variant), 16 * wino_variant_tn(variant), wino_variant_kh(variant) == 2 ? ",kh2" : "", split_suffix(ksplit));
    g_last_kernel = g_kernel_tag;
    if (ksplit > 1) {
        ConvArgs r = a;
        r.ksplit = ksplit;
        launch_splitk_reduce(r, L->ncls, s);
    }
    return true;
}

// 1-D minimal filtering for the k x 1 / 1 x k convs (conv_wino.hip), plan kind 10: variant = workgroup shape
bool wino1d_applies(const Layer *L) { return L->d_w1 != nullptr && L->wino1d_kind_of() >= 0; }

bool fill_wino1d_args(const Layer *L, const ConvArgs &a, int variant, Wino1Args &w)
{
    w.in = a.in; w.out = a.out; w.wu = L->d_w1; w.bias = a.bias; w.ws = a.ws;
    w.N = a.N; w.Cin = L->Cin; w.Cin4 = L->Cin4(); w.H = a.H; w.W = a.W; w.Ho = a.Ho; w.Wo = a.Wo; w.in_n_stride = a.in_n_stride;
    w.Cout = L->Cout; w.Mpad = L->Mpad; w.out_n_stride = a.out_n_stride; w.out_plane = a.out_plane;
    w.act = a.act; w.xcd = a.xcd; w.tl = a.tl;
    w.pad = L->wino1d_axis() == 0 ? L->ph : L->pw;
    const int cks = 4 * wino1d_variant_kg(variant);
    w.cross = L->wino1d_cross();
    w.cross_pad = w.cross > 1 ? L->ph : 0;
    w.csteps = (L->Cin + cks - 1) / cks;
    w.nsteps_total = w.cross * w.csteps;
    w.ksplit = 1;
    return wino1d_plan_geometry(w, L->wino1d_kind_of(), variant, L->wino1d_axis(), a.N);
}

bool run_wino1d(const Layer *L, const ConvArgs &a, int variant, int ksplit, hipStream_t s)
{
    refresh_stream_weights(L, s);
    Wino1Args w;
    if (variant < 0 || variant >= WINO1D_VARIANTS || !fill_wino1d_args(L, a, variant, w)) return false;
    if (ksplit < 1) ksplit = 1;
    if (ksplit > w.nsteps_total) ksplit = w.nsteps_total;
    w.ksplit = ksplit;
    if (!launch_wino1d(w, L->wino1d_kind_of(), variant, L->wino1d_axis(), s)) return false;
    snprintf(g_kernel_tag, sizeof g_kernel_tag, "wino1d<t%d%s,v%d>%s", L->wino1d_axis() == 0 ? L->kh : L->kw, L->wino1d_cross() > 1 ? "x3" : "", variant, split_suffix(ksplit));
    g_last_kernel = g_kernel_tag;
    if (ksplit > 1) {
        ConvArgs r = a;
        r.ksplit = ksplit;
        launch_splitk_reduce(r, L->ncls, s);
    }
    return true;
}

// 3 x 3 stride-1 convs with the transformed input rows stationary (conv_wino3.hip), plan kind 15: variant = workgroup shape.  Uses the
// transformed weights of the wino1d kernel (d_w1, cross = 3).  Variants 16 ..: the 3 x 3 stride-2 layers (weights in d_w3).
bool wino3_applies(const Layer *L)
{
    if (L->wino3_stride2()) return L->d_w3 != nullptr;
    return L->d_w1 != nullptr && L->wino1d_kind_of() == 0 && L->wino1d_cross() == 3 && (L->in.W & 1) == 0;
}

bool fill_wino3_args(const Layer *L, const ConvArgs &a, int variant, Wino3Args &w)
{
    if (variant < 0 || variant >= WINO3_VARIANTS) return false;
    const int form = wino3_variant_form(variant);
    if ((form == 2) != L->wino3_stride2()) return false;
    if (form >= 1 && !L->d_w3) return false;
    w.in = a.in; w.out = a.out; w.wu = form >= 1 ? L->d_w3 : L->d_w1; w.bias = a.bias;
    w.N = a.N; w.Cin = L->Cin; w.Cin4 = L->Cin4(); w.H = a.H; w.W = a.W; w.Ho = a.Ho; w.Wo = a.Wo; w.stride = form == 2 ? 2 : 1; w.in_n_stride = a.in_n_stride;
    w.Cout = L->Cout; w.Mpad = L->Mpad; w.out_n_stride = a.out_n_stride; w.out_plane = a.out_plane;
    w.act = a.act; w.xcd = a.xcd;
    return wino3_plan_geometry(w, variant);
}

bool run_wino3(const Layer *L, const ConvArgs &a, int variant, hipStream_t s)
{
    refresh_stream_weights(L, s);
    Wino3Args w;
    if (!wino3_applies(L) || !fill_wino3_args(L, a, variant, w)) return false;
    if (!launch_wino3(w, variant, s)) return false;
    snprintf(g_kernel_tag, sizeof g_kernel_tag, "wino3rows<%s,v%d>", wino3_variant_form(variant) == 2 ? "s2t3x3" : (wino3_variant_f4(variant) ? "f4t3x3" : "t3x3"), variant);
    g_last_kernel = g_kernel_tag;
    return true;
}

// k x 1 / 1 x k convs with four outputs per window (conv_wino4.hip), plan kind 16: variant = workgroup shape
bool wino4_applies(const Layer *L) { return L->d_w4 != nullptr && L->wino4_kind_of() >= 0; }

bool fill_wino4_args(const Layer *L, const ConvArgs &a, int variant, Wino4Args &w, bool flat = false)
{
    w.in = a.in; w.out = a.out; w.wu = L->d_w4; w.bias = a.bias;
    w.N = a.N; w.Cin = L->Cin; w.Cin4 = L->Cin4(); w.H = a.H; w.W = a.W; w.Ho = a.Ho; w.Wo = a.Wo; w.in_n_stride = a.in_n_stride;
    w.Cout = L->Cout; w.Mpad = L->Mpad; w.out_n_stride = a.out_n_stride; w.out_plane = a.out_plane;
    w.act = a.act; w.xcd = a.xcd;
    w.pad = L->wino1d_axis() == 0 ? L->ph : L->pw;
    return wino4_plan_geometry(w, L->wino4_kind_of(), variant, L->wino1d_axis(), flat);
}

// ksplit field of the plan entry: 2 = tile-walking workgroups (round 6; falls back to the plain launch where that form does not exist or
// the tiles fit the chip in one round), 3 = flat line order (round 6: the lines of all images as one sequence, for maps whose lines per image
// do not fill a workgroup; falls back to the plain launch where that saves no line block)
bool run_wino4(const Layer *L, const ConvArgs &a, int variant, hipStream_t s, int mode = 1)
{
    if (!wino4_applies(L)) return false;
    refresh_stream_weights(L, s);
    Wino4Args w;
    const bool flat = mode == 3 && fill_wino4_args(L, a, variant, w, true);
    if (!flat && !fill_wino4_args(L, a, variant, w)) return false;
    const bool walked = mode == 2 && launch_wino4(w, L->wino4_kind_of(), variant, L->wino1d_axis(), s, true);
    if (!walked && !launch_wino4(w, L->wino4_kind_of(), variant, L->wino1d_axis(), s)) return false;
    snprintf(g_kernel_tag, sizeof g_kernel_tag, walked ? "wino4<t%d,v%d,walk>" : (flat ? "wino4<t%d,v%d,flat>" : "wino4<t%d,v%d>"), L->wino1d_axis() == 0 ? L->kh : L->kw, variant);
    g_last_kernel = g_kernel_tag;
    return true;
}

// weight-streaming dense layer (dense_stream.hip), plan kind 11: ksplit = K slices across workgroups (dense_reduce_kernel adds them)
bool dense_stream_applies(const Layer *L) { return L->d_wd != nullptr; }

bool run_dense_stream(const Layer *L, const ConvArgs &a, int variant, int ksplit, hipStream_t s)
{
    if (variant < 0 || variant >= DENSE_VARIANTS) return false;
    refresh_stream_weights(L, s);   // (a no-op for network layers: prepare_stream_weights ran before any capture)
    if (ksplit < 1) ksplit = 1;
    while (ksplit > 1 && !dense_stream_geometry_ok(L->Cin, L->Mpad, ksplit)) --ksplit;
    if (!dense_stream_geometry_ok(L->Cin, L->Mpad, ksplit)) return false;
    DenseArgs d;
    d.x = a.in; d.out = a.out; d.wd = L->d_wd; d.bias = a.bias; d.ws = a.ws;
    d.N = a.N; d.K = L->Cin; d.Cout = L->Cout; d.Mpad = L->Mpad;
    d.x_n_stride = a.in_n_stride; d.out_n_stride = a.out_n_stride;
    d.act = a.act; d.ksplit = ksplit;
    launch_dense_stream(d, variant, s);
    snprintf(g_kernel_tag, sizeof g_kernel_tag, "dense_stream<128x32,v%d>%s", variant, split_suffix(ksplit));
    g_last_kernel = g_kernel_tag;
    return true;
}

// the blocks' first layer with the weights in registers (conv_thin.hip), plan kind 12.  On the k x 1 layer of a pair that conv_pair.hip
// can fuse, a kind-12 plan entry also means: run the pair as its two layers (demon_autotune measures both forms)
bool thin_applies(const Layer *L)
{
    return L->kind == Layer::CONV && !L->scale && conv_thin_shape_ok(L->kh, L->kw, L->sh, L->sw, L->ph, L->pw, L->Cin, L->Mpad, L->in.W, L->out.W);
}

bool run_thin(const Layer *L, const ConvArgs &a, hipStream_t s)
{
    ThinArgs t;
    t.in = a.in; t.out = a.out; t.wp = a.wp; t.bias = a.bias;
    t.N = a.N; t.Cin = L->Cin; t.H = a.H; t.W = a.W; t.in_n_stride = a.in_n_stride;
    t.Cout = L->Cout; t.Mpad = L->Mpad; t.Ho = a.Ho; t.Wo = a.Wo; t.out_n_stride = a.out_n_stride; t.out_plane = a.out_plane;
    t.pad = L->ph; t.act = a.act; t.xcd = a.xcd; t.tiles_y = t.tiles_x = 0;
    launch_conv_thin(t, s);
    snprintf(g_kernel_tag, sizeof g_kernel_tag, "conv_thin<32x512,t%d>", L->kh);
    g_last_kernel = g_kernel_tag;
    return true;
}

// 1 x 7 / 1 x 9 stride-2 convs with <= 32 channels on both sides, whole reduction out of LDS (conv_row.hip), plan kind 13; reads the
// transformed weights of the 1-D minimal-filtering kernel (d_w1)
bool row_applies(const Layer *L)
{
    return L->kind == Layer::CONV && !L->scale && L->d_w1 != nullptr && L->wino1d_cross() == 1 &&
           conv_row_shape_ok(L->kh, L->kw, L->sh, L->sw, L->ph, L->pw, L->Cin, L->Mpad, L->in.W, L->out.W);
}

bool run_row(const Layer *L, const ConvArgs &a, hipStream_t s)
{
    refresh_stream_weights(L, s);
    RowArgs r;
    r.in = a.in; r.out = a.out; r.wu = L->d_w1; r.bias = a.bias;
    r.N = a.N; r.Cin = L->Cin; r.Cin4 = L->Cin4(); r.H = a.H; r.W = a.W; r.in_n_stride = a.in_n_stride;
    r.Cout = L->Cout; r.Ho = a.Ho; r.Wo = a.Wo; r.out_n_stride = a.out_n_stride; r.out_plane = a.out_plane;
    r.pad = L->pw; r.act = a.act; r.tiles_y = r.tiles_x = 0;
    if (!launch_conv_row(r, L->kw, s)) return false;
    snprintf(g_kernel_tag, sizeof g_kernel_tag, "conv_row<32x128,t%d>", L->kw);
    g_last_kernel = g_kernel_tag;
    return true;
}

void run_mfma(const ConvArgs &a_in, ConvPlan plan, int ncls, hipStream_t s)
{
    launch_conv_mfma(a_in, plan, ncls, s);
    snprintf(g_kernel_tag, sizeof g_kernel_tag, "conv_mfma<%dx%d>%s", conv_tile_bm(plan.tile), conv_tile_bn(plan.tile), split_suffix(plan.ksplit));
    g_last_kernel = g_kernel_tag;
}

void run_layer(const Layer *L, int n, hipStream_t s, float *ws)
{
    ConvArgs a;
    fill_conv_args(L, n, ws, a);
    const long P = (long)n * a.Hp * a.Wp;
    auto clamp_split = [&](int k) {
        k %= 1000;
        while (k > 1 && (!ws || (long)L->ncls * k * L->Mpad * P > kSplitKWorkspaceFloats)) --k;
        return k;
    };
    // test hook: DEMON_FORCE_PLAN="kind,tile,ksplit" forces one kernel variant for every layer it applies to
    if (const char *fp = getenv("DEMON_FORCE_PLAN")) {
        int kind = 0, tile = 0, ks = 1;
        if (sscanf(fp, "%d,%d,%d", &kind, &tile, &ks) == 3) {
            if (kind == 3) {
                if (small_applies(L)) { run_small(L, a, s); return; }
            } else if (kind == 4) {
                if (L->stream_ok() && tile >= 0 && tile < STREAM_VARIANTS && L->Mpad % stream_variant_bm(tile) == 0) {
                    run_stream(L, a, tile, clamp_split(ks), s);
                    return;
                }
            } else if (kind == 5) {
                if (L->stream_ok() && tile >= 0 && tile < FRAG_VARIANTS && L->Mpad % frag_variant_bm(tile) == 0) {
                    run_frag(L, a, tile, clamp_split(ks), s);
                    return;
                }
            } else if (kind == 10) {
                if (wino1d_applies(L) && run_wino1d(L, a, tile, clamp_split(ks % 1000), s)) return;
            } else if (kind == 15) {
                if (run_wino3(L, a, tile, s)) return;
            } else if (kind == 16) {
                if (run_wino4(L, a, tile, s, ks)) return;
            } else if (kind == 13) {
                if (row_applies(L) && run_row(L, a, s)) return;
            } else if (kind == 12) {
                if (thin_applies(L) && run_thin(L, a, s)) return;
            } else if (kind == 11) {
                if (dense_stream_applies(L) && run_dense_stream(L, a, tile, clamp_split(ks % 1000), s)) return;
            } else if (kind == 8) {
                if (wino_applies(L) && tile >= 0 && tile < WINO_VARIANTS && run_wino(L, a, tile, clamp_split(ks % 1000), s)) return;
            } else if (kind == 1) {
                PatchPlan pp;
                if (tile >= 0 && tile < PTILE_COUNT && plan_patch(L, n, ws, pp, tile, ks / 1000 - 1)) {
                    if (ks % 1000 > 0) pp.a.ksplit = ks % 1000 < pp.a.nsteps_total ? ks % 1000 : pp.a.nsteps_total;
                    launch_patch_plan(L, pp, a, P, ws, s);
                    return;
                }
            } else if (tile >= 0 && tile < TILE_COUNT && L->Mpad % conv_tile_bm(tile) == 0) {
                run_mfma(a, ConvPlan{tile, clamp_split(ks < 1 ? 1 : std::min(ks % 1000, L->Kpad / 16))}, L->ncls, s);
                return;
            }
        }
    }
    if (L->force_tile < 0) {
        auto it = nearest_tuned(L, n);
        if (it != L->tuned.end()) {  // measured choice (demon_autotune), of this batch size or the nearest tuned one
            const Layer::Tuned &t = it->second;
            if (t.kind == 3 && small_applies(L)) {
                run_small(L, a, s);
                return;
            }
            // kinds 6 / 7 = chained with the 1 x k partner (run_chain); reached here only when the chain did not apply
            if ((t.kind == 4 || t.kind == 7) && L->stream_ok() && L->Mpad % stream_variant_bm(t.tile) == 0) {
                run_stream(L, a, t.tile, t.kind == 7 ? 1 : clamp_split(t.ksplit), s);
                return;
            }
            if ((t.kind == 5 || t.kind == 6) && L->stream_ok() && L->Mpad % frag_variant_bm(t.tile) == 0) {
                run_frag(L, a, t.tile, t.kind == 6 ? 1 : clamp_split(t.ksplit), s);
                return;
            }
            if (t.kind == 8 && wino_applies(L) && run_wino(L, a, t.tile, clamp_split(t.ksplit), s)) return;
            if (t.kind == 10 && wino1d_applies(L) && run_wino1d(L, a, t.tile, clamp_split(t.ksplit), s)) return;
            if (t.kind == 12 && thin_applies(L) && run_thin(L, a, s)) return;
            if (t.kind == 13 && row_applies(L) && run_row(L, a, s)) return;
            if (t.kind == 15 && run_wino3(L, a, t.tile, s)) return;
            if (t.kind == 16 && run_wino4(L, a, t.tile, s, t.ksplit)) return;
            if (t.kind == 11 && dense_stream_applies(L) && run_dense_stream(L, a, t.tile, clamp_split(t.ksplit), s)) return;
            if (t.kind == 1) {
                PatchPlan pp;
                const int tw = t.ksplit / 1000 - 1, ks = t.ksplit % 1000;
                if (plan_patch(L, n, ws, pp, t.tile, tw)) {
                    if (ks > 0) pp.a.ksplit = ks < pp.a.nsteps_total ? ks : pp.a.nsteps_total;
                    launch_patch_plan(L, pp, a, P, ws, s);
                    return;
                }
            } else if (t.kind == 0) {
                run_mfma(a, ConvPlan{t.tile, clamp_split(t.ksplit)}, L->ncls, s);
                return;
            }
        }
    }
    if (L->force_tile >= 400) {  // demon_bench_layer: minimal-filtering variant force_tile - 400 (transposed conv / k x 1, 1 x k, 3 x 3 conv)
        const int v = L->force_tile - 400;
        if (wino_applies(L) && v < WINO_VARIANTS && run_wino(L, a, v, clamp_split(L->force_split), s)) return;
        if (wino1d_applies(L) && v < WINO1D_VARIANTS && run_wino1d(L, a, v, clamp_split(L->force_split), s)) return;
        if (dense_stream_applies(L) && v < DENSE_VARIANTS && run_dense_stream(L, a, v, clamp_split(L->force_split), s)) return;
        if (thin_applies(L) && v == 0 && run_thin(L, a, s)) return;
        if (row_applies(L) && v == 100 && run_row(L, a, s)) return;   // (tile 500: the other variants of this layer are the 1-D kernel's)
    }
    if (L->force_tile >= 300 && L->force_tile < 400) {  // demon_bench_layer: fragment-tiled kernel variant force_tile - 300
        const int v = L->force_tile - 300;
        if (L->stream_ok() && v < FRAG_VARIANTS && L->Mpad % frag_variant_bm(v) == 0) { run_frag(L, a, v, clamp_split(L->force_split), s); return; }
    } else if (L->force_tile >= 200) {  // demon_bench_layer: streaming kernel variant force_tile - 200
        const int v = L->force_tile - 200;
        if (L->stream_ok() && v < STREAM_VARIANTS && L->Mpad % stream_variant_bm(v) == 0) { run_stream(L, a, v, clamp_split(L->force_split), s); return; }
    }
    if (L->force_tile < 0 && small_applies(L)) { run_small(L, a, s); return; }  // Cout <= 4 heads: VALU direct conv
    if (L->force_tile < 0 || (L->force_tile >= 100 && L->force_tile < 200)) {
        PatchPlan pp;
        if (plan_patch(L, n, ws, pp, L->force_tile >= 100 ? L->force_tile - 100 : -1)) {
            if (L->force_split > 0) pp.a.ksplit = L->force_split;
            launch_patch_plan(L, pp, a, P, ws, s);
            return;
        }
    }
    ConvPlan plan = choose_conv_plan(L->Mpad, P, L->ncls, L->Kpad, ws ? kSplitKWorkspaceFloats : 0);
    if (L->force_tile >= 0 && L->force_tile < TILE_COUNT) plan.tile = L->force_tile;
    if (L->force_split > 0) plan.ksplit = L->force_split;
    plan.ksplit = clamp_split(plan.ksplit);
    run_mfma(a, plan, L->ncls, s);
}

// Wall time (ms) of `body` -- captured five times into one hipGraph, like the real sequences run, so that the host-side planning of an
// eager launch does not leak into the comparison -- replayed on the context stream.  Option "tune_lanes" = L > 1 ("throughput
// mode"): L instances of the graph replay CONCURRENTLY on L streams and the time until all are done is returned.  That is the cost
// of a candidate when several passes are in flight on the GPU (demon_amd/lanes.py): a kernel that is fastest alone because it spreads
// over every CU, or splits K into many short workgroups plus a reduce launch, may lose to one that finishes its work in fewer
// SIMD cycles.  (The instances write the same outputs and share the split-K workspace: timing runs only.)
float time_replay(demon_ctx *c, const std::function<void()> &body, bool &failed)
{
    const int lanes = std::max(1, std::min(c->opt_tune_lanes, 8));
    while ((int)c->tune_streams.size() < lanes - 1) {
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { failed = true; return 1e30f; }
        c->tune_streams.push_back(st);
    }
    while ((int)c->tune_events.size() < lanes + 1) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) { failed = true; return 1e30f; }
        c->tune_events.push_back(e);
    }
    hipEvent_t e0 = c->tune_events[0], e1 = c->tune_events[1];
    hipGraph_t graph = nullptr;
    std::vector<hipGraphExec_t> execs((size_t)lanes, nullptr);
    bool ok = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
        for (int i = 0; i < 5; ++i) body();
        ok = hipStreamEndCapture(c->stream, &graph) == hipSuccess;
        for (int i = 0; ok && i < lanes; ++i) ok = hipGraphInstantiate(&execs[i], graph, nullptr, nullptr, 0) == hipSuccess;
    }
    auto stream_of = [&](int i) { return i == 0 ? c->stream : c->tune_streams[i - 1]; };
    if (ok) {
        for (int i = 0; i < lanes; ++i) hipGraphLaunch(execs[i], stream_of(i));  // warm-up
        for (int i = 1; i < lanes; ++i) hipStreamSynchronize(stream_of(i));
        hipEventRecord(e0, c->stream);
        for (int i = 1; i < lanes; ++i) hipStreamWaitEvent(stream_of(i), e0, 0);
        for (int i = 0; i < lanes; ++i) hipGraphLaunch(execs[i], stream_of(i));
        for (int i = 1; i < lanes; ++i) {
            hipEventRecord(c->tune_events[i + 1], stream_of(i));
            hipStreamWaitEvent(c->stream, c->tune_events[i + 1], 0);
        }
        hipEventRecord(e1, c->stream);
        ok = hipEventSynchronize(e1) == hipSuccess && hipGetLastError() == hipSuccess;
    }
    for (hipGraphExec_t e : execs) if (e) hipGraphExecDestroy(e);
    if (graph) hipGraphDestroy(graph);
    float ms = 1e30f;
    if (!ok || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { failed = true; return 1e30f; }
    return ms;
}

// Measures every applicable (kernel, tile, split-K) variant of one layer at batch n and remembers the fastest.
int mask_cus(const demon_ctx *c);
int autotune_layer(demon_ctx *c, Layer *L, int n)
{
    g_active_cus = mask_cus(c);
    struct Cand { int kind, tile, ksplit; };
    std::vector<Cand> cands;
    ConvArgs a;
    fill_conv_args(L, n, c->d_ws, a);
    const long P = (long)n * a.Hp * a.Wp;
    const ConvPlan heur = choose_conv_plan(L->Mpad, P, L->ncls, L->Kpad, kSplitKWorkspaceFloats);
    for (int t = 0; t < TILE_COUNT; ++t) {
        if (L->Mpad % conv_tile_bm(t)) continue;
        if (conv_tile_bn(t) > 32 && P * 2 <= conv_tile_bn(t)) continue;
        const long wgs = (long)(L->Mpad / conv_tile_bm(t)) * ((P + conv_tile_bn(t) - 1) / conv_tile_bn(t)) * L->ncls;
        const int nsteps = L->Kpad / 16;
        for (int ks : {1, 2, 3, 4, 6, 8, 12, 16, 24, 32}) {
            if (ks > 1 && (ks > nsteps / 4 || wgs * ks > 2048 || (long)L->ncls * ks * L->Mpad * P > kSplitKWorkspaceFloats)) continue;
            if (wgs * ks < 96 && !(t == heur.tile && ks == heur.ksplit)) continue;  // hopeless: less than 3/8 of the CUs busy
            cands.push_back({0, t, ks});
        }
    }
    for (int t = 0; t < PTILE_COUNT; ++t) {
        PatchPlan pp;
        if (!plan_patch(L, n, c->d_ws, pp, t)) continue;
        cands.push_back({1, t, 0});  // 0 = the planner's own pixel tile shape and split-K
        const long groups = (n + pp.a.G - 1) / pp.a.G;
        const long wgs = groups * pp.a.tiles_y * pp.a.tiles_x * patch_tile_mtiles(t, L->Cout, L->Mpad) * L->ncls;
        for (int ks : {1, 2, 3, 4, 6, 8}) {
            if (ks == pp.a.ksplit || ks > pp.a.nsteps_total / 2 || wgs * ks > 2048) continue;
            if (ks > 1 && (long)L->ncls * ks * L->Mpad * P > kSplitKWorkspaceFloats) continue;
            if (wgs * ks < 96) continue;
            cands.push_back({1, t, ks});
        }
        // the other pixel tile shapes (TW = tile width candidates of plan_patch), planner's split-K
        for (int tw = 0; tw < 4; ++tw) {
            PatchPlan q;
            if (!plan_patch(L, n, c->d_ws, q, t, tw)) continue;
            if (q.a.TW == pp.a.TW && q.a.TH == pp.a.TH && q.a.G == pp.a.G) continue;  // the shape already measured
            cands.push_back({1, t, 1000 * (tw + 1)});
        }
    }
    if (small_applies(L)) cands.push_back({3, 0, 0});
    if (L->stream_ok()) {
        const int nsteps = L->K / 16;
        for (int v = 0; v < STREAM_VARIANTS; ++v) {
            if (L->Mpad % stream_variant_bm(v)) continue;
            const long wgs = (long)(L->Mpad / stream_variant_bm(v)) * ((P + stream_variant_bn(v) - 1) / stream_variant_bn(v)) * L->ncls;
            const long waves = wgs * stream_variant_waves(v);
            const int kw = stream_variant_kw(v);
            for (int ks : {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48}) {
                if (ks > 1 && (ks > nsteps / 3 || (long)L->ncls * ks * L->Mpad * P > kSplitKWorkspaceFloats)) continue;
                if (kw > 1 && (ks > 2 || nsteps < 2 * kw * ks)) continue;  // in-workgroup split-K is there to AVOID the reduce launch
                if (waves * ks < (kw > 1 ? 128 : 512) || waves * ks > 16384) continue;  // enough waves to matter, at most 16 per SIMD
                cands.push_back({4, v, ks});
            }
        }
    }
    if (L->stream_ok()) {
        const int nsteps = L->K / 16;
        for (int v = 0; v < FRAG_VARIANTS; ++v) {
            if (L->Mpad % frag_variant_bm(v)) continue;
            if (frag_variant_bn(v) > 32 && P * 2 <= frag_variant_bn(v)) continue;
            const long wgs = (long)(L->Mpad / frag_variant_bm(v)) * ((P + frag_variant_bn(v) - 1) / frag_variant_bn(v)) * L->ncls;
            const int kw = frag_variant_kw(v);
            for (int ks : {1, 2, 3, 4, 6, 8, 12, 16, 24, 32}) {
                if (ks > 1 && (ks > nsteps / 4 || (long)L->ncls * ks * L->Mpad * P > kSplitKWorkspaceFloats)) continue;
                if (kw > 1 && (ks > 2 || nsteps < 4 * kw * ks)) continue;  // in-workgroup split-K is there to avoid the reduce launch
                if (wgs * ks * kw < 96 || wgs * ks > 4096) continue;
                cands.push_back({5, v, ks});
            }
        }
    }
    if (wino1d_applies(L)) {
        for (int v = 0; v < WINO1D_VARIANTS; ++v) {
            Wino1Args w;
            if (!fill_wino1d_args(L, a, v, w)) continue;
            const int nsteps = w.csteps * wino1d_variant_kg(v);
            const long wgs = wino1d_workgroups(w, v);
            for (int ks : {1, 2, 3, 4, 6, 8}) {
                if (ks > 1 && (ks > nsteps / 8 || wgs * ks > 4096 || (long)ks * L->Mpad * P > kSplitKWorkspaceFloats)) continue;
                if (wgs * ks < 64) continue;
                cands.push_back({10, v, ks});
            }
        }
    }
    if (wino4_applies(L)) {
        for (int v = 0; v < WINO4_VARIANTS; ++v) {
            Wino4Args w;
            if (fill_wino4_args(L, a, v, w) && wino4_workgroups(w, v) >= 128) {
                cands.push_back({16, v, 1});
                w.gx = (int)wino4_workgroups(w, v); w.gy = 1;
                if (L->Cin % (4 * wino4_variant_kg(v)) == 0 && wino4_persist_grid(w, L->wino4_kind_of(), v) > 0) cands.push_back({16, v, 2});   // tile-walking form
            }
            if (fill_wino4_args(L, a, v, w, true) && wino4_workgroups(w, v) >= 96) cands.push_back({16, v, 3});   // flat line order (only where it saves line blocks)
        }
    }
    if (wino3_applies(L)) {
        for (int v = 0; v < WINO3_VARIANTS; ++v) {
            Wino3Args w;
            if (fill_wino3_args(L, a, v, w) && wino3_workgroups(w, v) >= 128) cands.push_back({15, v, 1});
        }
    }
    if (thin_applies(L)) cands.push_back({12, 0, 1});
    if (row_applies(L)) cands.push_back({13, 0, 1});
    if (dense_stream_applies(L)) {
        const long blocks = (long)(L->Mpad / 128) * ((n + 31) / 32);
        for (int ks : {1, 2, 3, 4, 6, 8, 9, 12, 16, 18, 24, 32, 36, 48, 64}) {
            if (!dense_stream_geometry_ok(L->Cin, L->Mpad, ks)) continue;
            if (ks > 1 && ((long)ks * L->Mpad * P > kSplitKWorkspaceFloats || blocks * ks > 4096)) continue;
            if (blocks * ks < 64 && ks < 64) continue;
            for (int v = 0; v < DENSE_VARIANTS; ++v) cands.push_back({11, v, ks});
        }
    }
    if (wino_applies(L) && !getenv("DEMON_NO_WINO")) {
        const int nsteps = (L->Cin + 3) / 4;
        for (int v = 0; v < WINO_VARIANTS; ++v) {
            WinoArgs w;
            w.N = n; w.H = a.H; w.W = a.W; w.Mpad = L->Mpad; w.Cin = L->Cin; w.nsteps_total = nsteps;
            if (!wino_plan_geometry(w, v, n)) continue;
            const long wgs = wino_workgroups(w, v);
            for (int ks : {1, 2, 3, 4, 6, 8, 12, 16}) {
                if (ks > 1 && wino_variant_kh(v) == 2) continue;
                if (ks > 1 && (ks > nsteps / 8 || wgs * ks > 4096 || (long)L->ncls * ks * L->Mpad * P > kSplitKWorkspaceFloats)) continue;
                if (wgs * ks < 96) continue;
                cands.push_back({8, v, ks});
            }
        }
    }
    float best = 1e30f;
    Layer::Tuned best_t{0, heur.tile, heur.ksplit};
    bool failed = false;
    for (size_t i = 0; i < cands.size() && !failed; ++i) {
        const Cand &cd = cands[i];
        L->tuned[n] = Layer::Tuned{cd.kind, cd.tile, cd.ksplit};
        const float ms = time_replay(c, [&] { run_layer(L, n, c->stream, c->d_ws); }, failed);
        if (!failed && ms < best) { best = ms; best_t = Layer::Tuned{cd.kind, cd.tile, cd.ksplit}; }
    }
    if (failed) return DEMON_ERR_HIP;
    if (const char *pk = getenv("DEMON_TUNE_PICK")) {  // test hook: deterministic choice = candidate index
        const Cand &cd = cands[(size_t)atoi(pk) % cands.size()];
        best_t = Layer::Tuned{cd.kind, cd.tile, cd.ksplit};
    }
    L->tuned[n] = best_t;
    return DEMON_OK;
}

// k x 1 + 1 x k pair as one launch (conv_pair.hip); false -> the caller runs the two layers one after the other
bool run_pair(const Layer *Ly, const Layer *Lx, int n, hipStream_t s)
{
    PairArgs a;
    a.in = Ly->in.ptr(); a.out = Lx->out.ptr();
    a.w1 = Ly->d_wp; a.b1 = Ly->d_bias; a.w2 = Lx->d_wp; a.b2 = Lx->d_bias;
    a.N = n; a.Cin = Ly->Cin; a.H = Ly->in.H; a.W = Ly->in.W; a.in_n_stride = Ly->in.n_stride();
    a.CM = Ly->Cout; a.CMk = Lx->Cin; a.CO = Lx->Cout; a.Mpad1 = Ly->Mpad; a.Mpad2 = Lx->Mpad;
    a.Hm = Ly->out.H; a.Ho = Lx->out.H; a.Wo = Lx->out.W; a.out_n_stride = Lx->out.n_stride();
    a.ph = Ly->ph; a.pw = Lx->pw;
    conv_pair_tiles(a.Ho, a.Wo, a.tiles_y, a.tiles_x);
    const int cks = conv_pair_cks(Ly->kh);
    a.steps1 = (a.Cin + cks - 1) / cks;
    a.steps2 = (a.CM + cks - 1) / cks;
    static const int xcd_order = getenv("DEMON_XCD_ORDER") ? atoi(getenv("DEMON_XCD_ORDER")) : 1;
    a.xcd = xcd_order;
    if (!launch_conv_pair(a, Ly->kh, Ly->sh, s)) return false;
    g_last_kernel = "conv_pair";
    return true;
}

// ---- chained pairs: the k x 1 and the 1 x k conv of a stride-1 pair as ONE launch (conv_frag_chain_kernel / conv_stream_chain_kernel)
// when a workgroup tile of the chosen variant holds all channels of whole rows.  The plan of the k x 1 layer says so: kind 6 =
// chained on conv_frag variant `tile`, kind 7 = on conv_stream variant `tile`.
bool chain_shape_ok(const Layer *Ly, const Layer *Lx, int kind, int v)
{
    if (!Ly->stream_ok() || !Lx->stream_ok() || Ly->kind != Layer::CONV || Lx->kind != Layer::CONV) return false;
    if (Ly->sh != 1 || Ly->sw != 1 || Lx->sh != 1 || Lx->sw != 1 || Lx->kh != 1 || Ly->Mpad != Lx->Mpad) return false;
    if (Ly->out.H != Lx->out.H || Ly->out.W != Lx->out.W || Ly->scale || Lx->scale) return false;
    if (kind == 6) return v >= 0 && v < FRAG_VARIANTS && frag_variant_kw(v) == 1 && frag_variant_bm(v) == Ly->Mpad && frag_variant_bn(v) % Ly->out.W == 0;
    if (kind == 7) return v >= 0 && v < STREAM_VARIANTS && stream_variant_kw(v) == 1 && stream_variant_bm(v) == Ly->Mpad && stream_variant_bn(v) % Ly->out.W == 0;
    return false;
}

bool chain_choice(const Layer *Ly, const Layer *Lx, int n, int &kind, int &v)
{
    if (const char *fp = getenv("DEMON_FORCE_PLAN")) {  // test hook: "6,v,1" / "7,v,1" chain every pair the variant fits; other plans: no chains
        int ks = 0;
        if (sscanf(fp, "%d,%d,%d", &kind, &v, &ks) == 3) return (kind == 6 || kind == 7) && chain_shape_ok(Ly, Lx, kind, v);
    }
    auto it = nearest_tuned(Ly, n);
    if (it == Ly->tuned.end()) return false;
    kind = it->second.kind;
    v = it->second.tile;
    return (kind == 6 || kind == 7) && chain_shape_ok(Ly, Lx, kind, v);
}

bool run_chain(const Layer *Ly, const Layer *Lx, int n, int kind, int v, hipStream_t s, float *ws)
{
    ConvArgs a1, a2;
    fill_conv_args(Ly, n, ws, a1);
    fill_conv_args(Lx, n, ws, a2);
    StreamArgs s1, s2;
    fill_stream_args(Ly, a1, 1, s1, s);
    fill_stream_args(Lx, a2, 1, s2, s);
    if (!(kind == 6 ? launch_conv_frag_chain(s1, s2, v, s) : launch_conv_stream_chain(s1, s2, v, s))) return false;
    if (kind == 6) snprintf(g_kernel_tag, sizeof g_kernel_tag, "conv_frag_chain<%dx%d,v%d>", frag_variant_bm(v), frag_variant_bn(v), v);
    else snprintf(g_kernel_tag, sizeof g_kernel_tag, "conv_stream_chain<%dx%d,w%d>", stream_variant_bm(v), stream_variant_bn(v), stream_variant_waves(v));
    g_last_kernel = g_kernel_tag;
    return true;
}

// ---- topology builder ---------------------------------------------------------------------------------
struct Builder {
    demon_ctx *c;
    std::vector<Step> *steps;
    std::string scope;
    bool ok = true;
    int tag = 0;  // image_only tag given to the steps created while it is set
    // false: helpers.py:70-153 zero-pad k//2 on both sides then VALID; true: v2/helpers.py:24-91 padding='same', which for the
    // even input sizes of this net puts (k - stride) // 2 zeros in front and the rest behind
    bool same = false;
    int side = 0, fork_next = 0, join_next = 0;  // see Step::side
    int fused = 0;  // Step::fused of the steps created while it is set
    int fork_hold = 0;  // the fused step and the first step of the chain it replaces both carry the fork
    void stamp(Step &st)
    {
        st.side = side; st.fork = fork_next; st.join = join_next; st.fused = fused;
        fork_next = 0; join_next = 0;
    }

    Layer *make(const std::string &name, Layer::Kind kind, View in, View out, int kh, int kw, int sh, int sw, int act,
                const float *scale = nullptr, bool add_step = true)
    {
        auto L = std::make_unique<Layer>();
        L->name = scope + "/" + name;
        L->kind = kind;
        L->Cin = in.C; L->Cout = out.C;
        L->kh = kh; L->kw = kw; L->sh = sh; L->sw = sw; L->ph = kh / 2; L->pw = kw / 2; L->act = act;
        if (same) { L->ph = kh > sh ? (kh - sh) / 2 : 0; L->pw = kw > sw ? (kw - sw) / 2 : 0; }
        L->in = in; L->out = out; L->scale = scale;
        if (kind == Layer::CONV) L->kernel_dims = {kh, kw, in.C, out.C};
        else if (kind == Layer::DECONV) L->kernel_dims = {4, 4, out.C, in.C};
        else L->kernel_dims = {in.C, out.C};
        if (!plan_layer(c, L.get(), false)) ok = false;
        Layer *p = L.get();
        c->layers.push_back(std::move(L));
        add_variable(c, p, false);
        add_variable(c, p, true);
        Step st;
        st.name = p->name;
        st.kernel = "conv_mfma";
        const double pix = (kind == Layer::DECONV) ? 4.0 * in.H * in.W : (double)out.H * out.W;
        const double kreal = (kind == Layer::DECONV) ? 4.0 * in.C : (double)p->K;
        st.flops_per_sample = 2.0 * out.C * kreal * pix;
        st.bytes_per_sample = 4.0 * ((double)in.C * in.H * in.W + (double)out.C * out.H * out.W);
        st.bytes_fixed = 4.0 * ((double)p->K * p->ncls * out.C + out.C);
        float *ws = side ? c->d_ws_side : c->d_ws;
        st.fn = [p, ws](int n, hipStream_t s) { run_layer(p, n, s, ws); };
        st.image_only = tag;
        if (add_step) { stamp(st); steps->push_back(st); }
        return p;
    }
    // helpers.py:70-102
    Layer *conv(const std::string &name, View in, View out, int k, int stride, int act, const float *scale = nullptr)
    {
        return make(name, Layer::CONV, in, out, k, k, stride, stride, act, scale);
    }
    // helpers.py:105-153: <name>y = k x 1 stride (s,1), <name>x = 1 x k stride (1,s), both leaky relu;
    // v2/helpers.py:44-91: the same pair with `cy` outputs of the first filter ((24,32), (48,64), ...)
    void conv2(const std::string &name, View in, View out, int k, int s, int cy = 0)
    {
        const int Hmid = (in.H + 2 * (k / 2) - k) / s + 1;  // == ceil(H / s), the 'same' size, for the even H of this net
        if (!cy) cy = out.C;
        char key[64];
        snprintf(key, sizeof key, "tmp_y_%dx%dx%d", cy, Hmid, in.W);
        View mid = buffer(c, key, cy, Hmid, in.W);
        // large maps with few channels (levels 1-2): one fused launch for the pair (conv_pair.hip) when the option is on
        const bool fusable = conv_pair_applies(k, s, in.C, cy, out.C) && (long)out.H * out.W >= 2048;
        // stride-1 pairs of the deeper levels: one chained launch when the launch plan says so (run_chain)
        const bool chainable = !fusable && s == 1 && Layer::stream_shape_ok(Layer::CONV, in.C, k, 1) && Layer::stream_shape_ok(Layer::CONV, cy, 1, k) &&
                               (cy + 31) / 32 == (out.C + 31) / 32;
        if (!fusable && !chainable) {
            make(name + "y", Layer::CONV, in, mid, k, 1, s, 1, 1);
            make(name + "x", Layer::CONV, mid, out, 1, k, 1, s, 1);
            return;
        }
        // both forms go into the list (Step::pair): the fused step first, then the two layers
        const int jn = join_next, fk = fork_next;
        Layer *Ly = make(name + "y", Layer::CONV, in, mid, k, 1, s, 1, 1, nullptr, false);
        Layer *Lx = make(name + "x", Layer::CONV, mid, out, 1, k, 1, s, 1, nullptr, false);
        Step st;
        st.name = Ly->name + "+x";
        st.kernel = "conv_mfma";
        st.flops_per_sample = 2.0 * Ly->Cout * Ly->K * Ly->out.H * Ly->out.W + 2.0 * Lx->Cout * Lx->K * Lx->out.H * Lx->out.W;
        st.bytes_per_sample = 4.0 * ((double)in.C * in.H * in.W + (double)out.C * out.H * out.W);
        st.bytes_fixed = 4.0 * ((double)Ly->K * Ly->Cout + (double)Lx->K * Lx->Cout + Ly->Cout + Lx->Cout);
        float *ws = side ? c->d_ws_side : c->d_ws;
        std::function<bool(int)> applies;
        if (chainable) {
            c->chain_pairs.push_back({Ly, Lx});
            applies = [Ly, Lx](int n) { int kind, v; return chain_choice(Ly, Lx, n, kind, v); };
            st.bytes_per_sample += 8.0 * (double)cy * Hmid * in.W;  // the intermediate is still written, and read back out of L2
            st.fn = [Ly, Lx, ws](int n, hipStream_t s2) {
                int kind, v;
                if (chain_choice(Ly, Lx, n, kind, v) && run_chain(Ly, Lx, n, kind, v, s2, ws)) return;
                run_layer(Ly, n, s2, ws);
                run_layer(Lx, n, s2, ws);
            };
        } else {
            // plan kind 12 on the k x 1 layer (conv_thin.hip): the pair runs as its two layers, not as the fused launch
            c->fused_pairs.push_back({Ly, Lx});
            applies = [Ly](int n) { auto it = nearest_tuned(Ly, n); return !(it != Ly->tuned.end() && it->second.kind == 12 && thin_applies(Ly)); };
            st.fn = [Ly, Lx, ws](int n, hipStream_t s2) {
                if (run_pair(Ly, Lx, n, s2)) return;
                run_layer(Ly, n, s2, ws);
                run_layer(Lx, n, s2, ws);
            };
        }
        st.pair_applies = applies;
        st.image_only = tag;
        st.pair = 1;
        join_next = jn; fork_next = fk;
        stamp(st);
        steps->push_back(st);
        for (Layer *L : {Ly, Lx}) {
            Step sl;
            sl.name = L->name;
            sl.kernel = "conv_mfma";
            sl.flops_per_sample = 2.0 * L->Cout * L->K * L->out.H * L->out.W;
            sl.bytes_per_sample = 4.0 * ((double)L->in.C * L->in.H * L->in.W + (double)L->out.C * L->out.H * L->out.W);
            sl.bytes_fixed = 4.0 * ((double)L->K * L->Cout + L->Cout);
            sl.fn = [L, ws](int n, hipStream_t s2) { run_layer(L, n, s2, ws); };
            sl.image_only = tag;
            sl.pair = 2;
            sl.pair_applies = applies;
            if (L == Ly) { join_next = jn; fork_next = fk; }  // the same fork / join as the fused form
            stamp(sl);
            steps->push_back(sl);
        }
    }
    // blocks_original.py:97-110 (lrelu) and :64-75 (linear)
    Layer *deconv(const std::string &name, View in, View out, int act) { return make(name, Layer::DECONV, in, out, 4, 4, 2, 2, act); }
    Layer *dense(const std::string &name, View in, View out, int act, bool add_step = true) { return make(name, Layer::DENSE, in, out, 1, 1, 1, 1, act, nullptr, add_step); }
    void op(const std::string &name, const std::string &kernel, double bytes_per_sample, std::function<void(int, hipStream_t)> fn)
    {
        Step st;
        st.name = scope + "/" + name;
        st.kernel = kernel;
        st.bytes_per_sample = bytes_per_sample;
        st.fn = std::move(fn);
        st.image_only = tag;
        stamp(st);
        steps->push_back(st);
    }
};


// Option "reuse_image_features": conv1 / conv2 of netFlow2 and netDM2 see only image_pair and their weights, so they give the
// same result in all iterations of one forward pass.  After them (tag 1) the result is saved to a cache buffer (tag 3, runs
// with them in the first iteration) and a copy from the cache (tag 2) stands in for them in the later iterations.
void add_image_cache_steps(Builder &b, const char *key, View conv2_out)
{
    View cache = buffer(b.c, key, conv2_out.C, conv2_out.H, conv2_out.W);
    const long hw = (long)conv2_out.H * conv2_out.W;
    const int C = conv2_out.C;
    b.tag = 3;
    b.op("save_image_features", "copy_channels", 8.0 * C * hw, [=](int n, hipStream_t s) {
        launch_copy_channels(cache.ptr(), cache.n_stride(), conv2_out.ptr(), conv2_out.n_stride(), n, C, hw, s);
    });
    b.tag = 2;
    b.op("reuse_image_features", "copy_channels", 8.0 * C * hw, [=](int n, hipStream_t s) {
        launch_copy_channels(conv2_out.ptr(), conv2_out.n_stride(), cache.ptr(), cache.n_stride(), n, C, hw, s);
    });
    b.tag = 0;
}

// Channel plan of the shared encoder: blocks_original.py:141-153 / v2/blocks.py:141-195 (v2: asymmetric 1-D pairs, 384 at level 5)
struct EncPlan {
    int c1y, c2y_boot, c3y, c4y, c5;
};
EncPlan enc_plan(const demon_ctx *c)
{
    return c->variant == 2 ? EncPlan{24, 48, 96, 192, 384} : EncPlan{32, 64, 128, 256, 512};
}

// v2/blocks.py:197-213 (flow block), :395-411 (depth+motion block): the first 96 channels of conv5_1, flattened in C,H,W order
// (= the first 96*h5*w5 floats of each sample of the NCHW buffer), go through a square dense layer and come back as 96 more
// channels behind conv5_1.  `feat` is the [384 + 96, h5, w5] buffer whose first 384 channels conv5_1 has just written.
void add_dense5(Builder &b, View feat)
{
    const int hw = feat.H * feat.W, units = 96 * hw;
    View in = feat, out = feat;
    in.Ctot = out.Ctot = feat.Ctot * hw;
    in.H = in.W = out.H = out.W = 1;
    in.c0 = 0; in.C = units;
    out.c0 = (feat.Ctot - 96) * hw; out.C = units;
    b.dense("dense5", in, out, 1);
}

// blocks_original.py:121-235; v2/blocks.py:120-253
void build_flow(demon_ctx *c, std::vector<Step> *steps, const std::string &scope, bool iterative)
{
    Builder b{c, steps, scope};
    const bool v2 = c->variant == 2;
    const EncPlan ep = enc_plan(c);
    b.same = v2;
    const int H = c->H, W = c->W, h1 = H / 2, w1 = W / 2, h2 = H / 4, w2 = W / 4, h3 = H / 8, w3 = W / 8, h4 = H / 16,
              w4 = W / 16, h5 = H / 32, w5 = W / 32;
    View conv1 = buffer(c, "conv1", 32, h1, w1);
    View conv2cat = buffer(c, "conv2cat", 64, h2, w2);
    View concat2 = buffer(c, "concat2", 128, h2, w2);
    View concat3 = buffer(c, "concat3", 256, h3, w3);
    View concat4f = buffer(c, "concat4_flow", 514, h4, w4);
    View conv3 = buffer(c, "conv3", 128, h3, w3), conv4 = buffer(c, "conv4", 256, h4, w4);
    View conv5 = buffer(c, "conv5", ep.c5, h5, w5);
    View feat5 = buffer(c, "conv5_1", v2 ? ep.c5 + 96 : ep.c5, h5, w5);  // v2: [conv5_1 384, dense5 96] (v2/blocks.py:213)
    View conv5_1 = feat5.slice(0, ep.c5);
    if (iterative) b.tag = 1;
    const size_t block_begin = steps->size();
    b.conv2("conv1", c->image_pair, conv1, 9, 2, ep.c1y);
    if (!iterative) {
        b.conv2("conv2", conv1, conv2cat, 7, 2, ep.c2y_boot);  // 64 outputs (:144; v2 :144 (48,64))
    } else {
        b.conv2("conv2", conv1, conv2cat.slice(0, 32), 7, 2);
        add_image_cache_steps(b, "flow2_conv2_cache", conv2cat.slice(0, 32));
        // side branch: the extra-input chain (previous depth / motion -> flow -> warp -> conv2_extra_inputs: five small launches)
        // does not depend on conv1 / conv2, which see only the images; it is enqueued in front of them and joins at conv2_1
        std::vector<Step> side_steps;
        b.steps = &side_steps;
        b.side = 1; b.fork_next = 1;
        View extra = buffer(c, "extra_flow", 9, h2, w2);  // [warped 3, flow 2, depth 1, normal 3] (:180; v2 :180)
        View img2 = c->image2_2, depth2 = c->depth2, normal2 = c->normal2;
        float *rot = c->d_rot, *trans = c->d_trans, *intr = c->d_intrinsics;
        const double px = (double)h2 * w2 * 4;
        {
            View dn = depth2;  // depth2 / normal2 are the slices [0,1) / [1,4) of one 4-channel buffer
            b.fused = 1;
            b.op("assemble_inputs", "assemble_flow_inputs", px * (4 + 3 + 9), [=](int n, hipStream_t s) {
                launch_assemble_flow_inputs(extra.ptr(), extra.n_stride(), img2.ptr(), img2.n_stride(), dn.ptr(), dn.n_stride(), intr, rot, trans,
                                            n, h2, w2, s);
            });
            b.fused = 2;
            b.fork_next = 1;
        }
        b.op("depth_to_flow", "depth_to_flow", px * 3, [=](int n, hipStream_t s) {
            launch_depth_to_flow(extra.slice(3, 2).ptr(), depth2.ptr(), depth2.n_stride(), intr, rot, trans, n, h2, w2,
                                 extra.n_stride(), 1, 1, 1, s);
        });
        b.op("warp2d", "warp2d", px * 8, [=](int n, hipStream_t s) {
            launch_warp2d(extra.ptr(), extra.n_stride(), img2.ptr(), img2.n_stride(), extra.slice(3, 2).ptr(),
                          extra.n_stride(), n, 3, h2, w2, 1, 1, 0.0f, s);
        });
        b.op("concat_prev", "copy_channels", px * 8, [=](int n, hipStream_t s) {
            launch_copy_channels(extra.slice(5, 1).ptr(), extra.n_stride(), depth2.ptr(), depth2.n_stride(), n, 1,
                                 (long)h2 * w2, s);
            launch_copy_channels(extra.slice(6, 3).ptr(), extra.n_stride(), normal2.ptr(), normal2.n_stride(), n, 3,
                                 (long)h2 * w2, s);
        });
        b.fused = 0;
        b.conv2("conv2_extra_inputs", extra, conv2cat.slice(32, 32), 3, 1);
        b.side = 0;
        b.steps = steps;
        steps->insert(steps->begin() + block_begin, side_steps.begin(), side_steps.end());
        b.join_next = 1;
    }
    b.conv2("conv2_1", conv2cat, concat2.slice(64, 64), 3, 1);
    b.conv2("conv3", concat2.slice(64, 64), conv3, 5, 2, ep.c3y);
    b.conv2("conv3_1", conv3, concat3.slice(128, 128), 3, 1);
    b.conv2("conv4", concat3.slice(128, 128), conv4, 5, 2, ep.c4y);
    b.conv2("conv4_1", conv4, concat4f.slice(256, 256), 3, 1);
    b.conv2("conv5", concat4f.slice(256, 256), conv5, 5, 2);
    b.conv2("conv5_1", conv5, conv5_1, 3, 1);
    if (v2) add_dense5(b, feat5);
    View pf5 = buffer(c, "predict5_tmp", 24, h5, w5);
    // side branch: the level-5 flow head and its upsampling (three launches on a 6x8 map) next to refine4/upconv; refine3/upconv
    // reads the whole concat4 buffer, so it joins
    b.side = 1; b.fork_next = 1;
    b.conv("predict_flow5/conv1", feat5, pf5, 3, 1, 1);
    b.conv("predict_flow5/conv2", pf5, c->flowconf5, 3, 1, 0);
    b.deconv("upsample_flow5to4/upconv", c->flowconf5, concat4f.slice(512, 2), 0);
    b.side = 0;
    b.deconv("refine4/upconv", feat5, concat4f.slice(0, 256), 1);
    b.join_next = 1;
    b.deconv("refine3/upconv", concat4f, concat3.slice(0, 128), 1);
    b.deconv("refine2/upconv", concat3, concat2.slice(0, 64), 1);
    View pf2 = buffer(c, "predict2_tmp", 24, h2, w2);
    b.conv("predict_flow2/conv1", concat2, pf2, 3, 1, 1);
    b.conv("predict_flow2/conv2", pf2, c->flowconf2, 3, 1, 0);
    if (!b.ok) c->err = "device allocation failed while building " + scope;
}

// blocks_original.py:299-448; v2/blocks.py:317-494
void build_dm(demon_ctx *c, std::vector<Step> *steps, const std::string &scope, bool iterative)
{
    Builder b{c, steps, scope};
    const bool v2 = c->variant == 2;
    const EncPlan ep = enc_plan(c);
    b.same = v2;
    const int H = c->H, W = c->W, h1 = H / 2, w1 = W / 2, h2 = H / 4, w2 = W / 4, h3 = H / 8, w3 = W / 8, h4 = H / 16,
              w4 = W / 16, h5 = H / 32, w5 = W / 32;
    View conv1 = buffer(c, "conv1", 32, h1, w1);
    View conv2cat = buffer(c, "conv2cat", 64, h2, w2);
    View concat2 = buffer(c, "concat2", 128, h2, w2);
    View concat3 = buffer(c, "concat3", 256, h3, w3);
    View concat4 = buffer(c, "concat4_dm", 512, h4, w4);
    View conv3 = buffer(c, "conv3", 128, h3, w3), conv4 = buffer(c, "conv4", 256, h4, w4);
    View conv5 = buffer(c, "conv5", ep.c5, h5, w5);
    View feat5 = buffer(c, "conv5_1", v2 ? ep.c5 + 96 : ep.c5, h5, w5);
    View conv5_1 = feat5.slice(0, ep.c5);
    if (iterative) b.tag = 1;
    const size_t block_begin = steps->size();
    b.conv2("conv1", c->image_pair, conv1, 9, 2, ep.c1y);
    b.conv2("conv2", conv1, conv2cat.slice(0, 32), 7, 2);
    if (iterative) add_image_cache_steps(b, "dm2_conv2_cache", conv2cat.slice(0, 32));
    b.tag = 0;
    // side branch (as in the flow block): warp / flow_to_depth / conv2_extra_inputs next to conv1 / conv2, joined at conv2_1
    std::vector<Step> side_steps;
    b.steps = &side_steps;
    b.side = 1; b.fork_next = 1;
    const int nextra = iterative ? 8 : 7;  // [warped 3, flowconf 4, depth_from_flow 1] (:341, :362; v2 :359, :381)
    View extra = buffer(c, iterative ? "extra_dm8" : "extra_dm7", nextra, h2, w2);
    View img2 = c->image2_2, flowconf2 = c->flowconf2;
    const double px = (double)h2 * w2 * 4;
    {
        float *rot = c->d_rot, *trans = c->d_trans, *intr = c->d_intrinsics;
        demon_ctx *cc = c;
        const int with_depth = iterative ? 1 : 0;
        b.fused = 1;
        b.op("assemble_inputs", "assemble_dm_inputs", px * (4 + 3 + nextra), [=](int n, hipStream_t s) {
            launch_assemble_dm_inputs(extra.ptr(), extra.n_stride(), img2.ptr(), img2.n_stride(), flowconf2.ptr(), flowconf2.n_stride(), intr, rot,
                                      trans, n, h2, w2, with_depth, v2 ? 1 : cc->opt_f2d_method, v2 ? 50.0f : 0.0f, s);
        });
        b.fused = 2;
        b.fork_next = 1;
    }
    b.op("warp2d", "warp2d", px * 8, [=](int n, hipStream_t s) {
        launch_warp2d(extra.ptr(), extra.n_stride(), img2.ptr(), img2.n_stride(), flowconf2.ptr(), flowconf2.n_stride(), n,
                      3, h2, w2, 1, 1, 0.0f, s);
    });
    b.op("concat_flowconf", "copy_channels", px * 8, [=](int n, hipStream_t s) {
        launch_copy_channels(extra.slice(3, 4).ptr(), extra.n_stride(), flowconf2.ptr(), flowconf2.n_stride(), n, 4,
                             (long)h2 * w2, s);
    });
    if (iterative) {
        float *rot = c->d_rot, *trans = c->d_trans, *intr = c->d_intrinsics;
        demon_ctx *cc = c;
        // v2: flow_to_depth2 followed by clip_by_value(0, 50) (v2/blocks.py:362-379), regardless of the option
        b.op("flow_to_depth", "flow_to_depth", px * 3, [=](int n, hipStream_t s) {
            launch_flow_to_depth(extra.slice(7, 1).ptr(), extra.n_stride(), flowconf2.ptr(), flowconf2.n_stride(), intr,
                                 rot, trans, n, h2, w2, 1, 1, v2 ? 1 : cc->opt_f2d_method, v2 ? 50.0f : 0.0f, s);
        });
    }
    b.fused = 0;
    b.conv2("conv2_extra_inputs", extra, conv2cat.slice(32, 32), 3, 1);
    b.side = 0;
    b.steps = steps;
    steps->insert(steps->begin() + block_begin, side_steps.begin(), side_steps.end());
    b.join_next = 1;
    b.conv2("conv2_1", conv2cat, concat2.slice(64, 64), 3, 1);
    b.conv2("conv3", concat2.slice(64, 64), conv3, 5, 2, ep.c3y);
    b.conv2("conv3_1", conv3, concat3.slice(128, 128), 3, 1);
    b.conv2("conv4", concat3.slice(128, 128), conv4, 5, 2, ep.c4y);
    b.conv2("conv4_1", conv4, concat4.slice(256, 256), 3, 1);
    b.conv2("conv5", concat4.slice(256, 256), conv5, 3, 2);  // k = 3 (:375; v2 :392)
    b.conv2("conv5_1", conv5, conv5_1, 3, 1);
    // motion head (:380-412; v2 :413-457); flatten is C,H,W order = NCHW memory order
    View mconv = buffer(c, "motion_conv1", 128, h5, w5);
    // side branch: the whole motion head (small maps, dense layers) next to the decoder; predict_depthnormal2/conv2 needs the
    // predicted scale, so it joins.  The branch has its own split-K workspace.
    b.side = 1; b.fork_next = 1;
    if (!v2) {
        b.conv("motion_conv1", conv5_1, mconv, 3, 1, 1);
    } else {
        add_dense5(b, feat5);
        View m3 = buffer(c, "motion_conv3", 64, h3, w3), m4 = buffer(c, "motion_conv4", 64, h4, w4);
        b.conv2("motion_conv3", concat2.slice(64, 64), m3, 5, 2);
        b.conv2("motion_conv4", m3, m4, 5, 2);
        b.conv2("motion_conv5a", m4, mconv.slice(0, 64), 3, 2);
        b.conv("motion_conv5b", feat5, mconv.slice(64, 64), 3, 1, 1);
    }
    View fc_in = mconv;
    fc_in.C = fc_in.Ctot = 128 * h5 * w5; fc_in.H = fc_in.W = 1;
    View fc1 = buffer(c, "motion_fc1", 1024, 1, 1), fc2 = buffer(c, "motion_fc2", 128, 1, 1);
    View fc3;
    fc3.base = c->d_motion; fc3.Ctot = fc3.C = 7; fc3.c0 = 0; fc3.H = fc3.W = 1;
    b.dense("motion_fc1", fc_in, fc1, 1);
    // motion_fc2 + motion_fc3 + the rotation / translation / scale split (:397-412) run as ONE small kernel; the layers are
    // still created so that their variables exist and are packed like every other layer
    Layer *L2 = b.dense("motion_fc2", fc1, fc2, 1, false);
    Layer *L3 = b.dense("motion_fc3", fc2, fc3, 0, false);
    {
        float *motion = c->d_motion, *rot = c->d_rot, *trans = c->d_trans, *scale = c->d_scale;
        const float *x = fc1.base;
        b.op("motion_fc2+fc3+split", "motion_tail", 4.0 * (1024 + 7), [=](int n, hipStream_t s) {
            launch_motion_tail(x, L2->d_wp, L2->d_bias, L3->d_wp, L3->d_bias, motion, rot, trans, scale, n, 1024, L2->Mpad, L3->Mpad, s);
        });
    }
    b.side = 0;
    b.deconv("refine4/upconv", conv5_1, concat4.slice(0, 256), 1);  // v2 too: conv5_1 without dense5 (v2/blocks.py:462)
    b.deconv("refine3/upconv", concat4, concat3.slice(0, 128), 1);
    b.deconv("refine2/upconv", concat3, concat2.slice(0, 64), 1);
    View pd = buffer(c, "predict2_tmp", 24, h2, w2);
    View dn = buffer(c, "depthnormal2", 4, h2, w2);  // ch 0 = scale*depth, ch 1:4 = normal (:278-287; v2 :294-305)
    b.conv("predict_depthnormal2/conv1", concat2, pd, 3, 1, 1);
    b.join_next = 1;
    b.conv("predict_depthnormal2/conv2", pd, dn, 3, 1, 0, c->d_scale);
    if (!b.ok) c->err = "device allocation failed while building " + scope;
}

// blocks_original.py:452-513; v2/blocks.py:499-562 (4 output channels: depth + normal)
void build_refine(demon_ctx *c, std::vector<Step> *steps)
{
    Builder b{c, steps, "netRefine"};
    b.same = c->variant == 2;
    const int H = c->H, W = c->W, h1 = H / 2, w1 = W / 2, h2 = H / 4, w2 = W / 4;
    View inp = buffer(c, "refine_in", 4, H, W);  // [image1 3, depth2 upsampled 1] (:482; v2 :527)
    View concat0 = buffer(c, "refine_concat0", 64, H, W);
    View concat1 = buffer(c, "refine_concat1", 128, h1, w1);
    View r1 = buffer(c, "refine_conv1", 64, h1, w1), r2 = buffer(c, "refine_conv2", 128, h2, w2),
         r2_1 = buffer(c, "refine_conv2_1", 128, h2, w2);
    View image_pair = c->image_pair, depth2 = c->depth2;
    b.fused = 1;
    b.op("assemble_input", "assemble_refine_input", 4.0 * (4.0 * H * W + 3.0 * H * W + h2 * w2), [=](int n, hipStream_t s) {
        launch_assemble_refine_input(inp.ptr(), inp.n_stride(), image_pair.ptr(), image_pair.n_stride(), depth2.ptr(), depth2.n_stride(), n, H, W, 4, s);
    });
    b.fused = 2;
    b.op("assemble_input", "upsample_nearest", 4.0 * (4.0 * H * W + 3.0 * H * W + h2 * w2), [=](int n, hipStream_t s) {
        launch_copy_channels(inp.ptr(), inp.n_stride(), image_pair.ptr(), image_pair.n_stride(), n, 3, (long)H * W, s);
        launch_upsample_nearest(inp.slice(3, 1).ptr(), inp.n_stride(), depth2.ptr(), depth2.n_stride(), n, 1, h2, w2, 4, s);
    });
    b.fused = 0;
    b.conv("conv0", inp, concat0.slice(32, 32), 3, 1, 1);
    b.conv("conv1", concat0.slice(32, 32), r1, 3, 2, 1);
    b.conv("conv1_1", r1, concat1.slice(64, 64), 3, 1, 1);
    b.conv("conv2", concat1.slice(64, 64), r2, 3, 2, 1);
    b.conv("conv2_1", r2, r2_1, 3, 1, 1);
    b.deconv("refine1/upconv", r2_1, concat1.slice(0, 64), 1);
    b.deconv("refine0/upconv", concat1, concat0.slice(0, 32), 1);
    View p0 = buffer(c, "predict0_tmp", 16, H, W);
    b.conv("predict_depth0/conv1", concat0, p0, 3, 1, 1);
    // v1: c->depth0 is a 1-channel buffer; v2: [depth0, normal0 xyz] in one 4-channel buffer (v2/blocks.py:560)
    View head = c->depth0;
    if (c->variant == 2) head.C = 4;
    b.conv("predict_depth0/conv2", p0, head, 3, 1, 0);
    if (!b.ok) c->err = "device allocation failed while building netRefine";
}

bool weights_ready(demon_ctx *c, std::string *missing)
{
    for (auto &L : c->layers)
        if (!L->have_kernel || !L->have_bias) {
            if (missing) *missing = L->name;
            return false;
        }
    return true;
}

// mode 0: plain (every layer, no cache traffic); 1: first iteration with the option on (layers + save); 2: later iterations
// (cached conv2 output instead of the image-only layers)
// fork / join events: one per use inside a sequence (a captured event must not be re-recorded), created on demand so that any
// iteration count fits; a failed creation or record / wait turns the side branches off for the rest of the sequence and is
// reported by the caller (c->err)
hipEvent_t next_event(demon_ctx *c, size_t &ev)
{
    if (ev >= c->events.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        c->events.push_back(e);
    }
    hipEvent_t e = c->events[ev];
    if (e) ++ev;
    return e;
}

// dst waits for everything enqueued on src so far; false (and an error note) when an event call fails
bool stream_wait(demon_ctx *c, hipStream_t src, hipStream_t dst, size_t &ev)
{
    hipEvent_t e = next_event(c, ev);
    if (e && hipEventRecord(e, src) == hipSuccess && hipStreamWaitEvent(dst, e, 0) == hipSuccess) return true;
    if (c->err.empty()) c->err = "fork / join event of the side stream failed";
    return false;
}

// of a k x 1 / 1 x k pair either the fused step (pair == 1) or its two layer steps (pair == 2) run
bool pair_step_skipped(const demon_ctx *c, const Step &st, int n)
{
    if (!st.pair) return false;
    const bool fused = c->opt_fused_pairs && (!st.pair_applies || st.pair_applies(n));
    return st.pair == 1 ? !fused : fused;
}

int mask_cus(const demon_ctx *c)
{
    int k = 0;
    for (uint32_t w : c->cu_mask) k += __builtin_popcount(w);
    return k;
}

void run_steps(demon_ctx *c, const std::vector<Step> &steps, int n, hipStream_t s, int mode, size_t &ev)
{
    g_active_cus = mask_cus(c);
    bool branches = c->opt_side_branches && c->side_stream && s == c->stream;
    bool side_open = false;  // work on the side stream that the main stream has not waited for yet
    for (const Step &st : steps) {
        if (st.image_only == 1 && mode == 2) continue;
        if (st.image_only == 2 && mode != 2) continue;
        if (st.image_only == 3 && mode != 1) continue;
        if (pair_step_skipped(c, st, n)) continue;
        if ((st.fused == 1 && !c->opt_fused_inputs) || (st.fused == 2 && c->opt_fused_inputs)) continue;
        // diagnostic hook (tools/ablate_lanes.py): DEMON_SKIP_STEPS="sub1,sub2" leaves out the steps whose name contains one of the
        // substrings -- WRONG results, used only to read off what a group of layers costs while several passes are in flight
        static const std::string skip = [] {
            const char *e = getenv("DEMON_SKIP_STEPS");
            if (e && *e) fprintf(stderr, "libdemon_hip: DEMON_SKIP_STEPS=%s is set -- steps are LEFT OUT of every pass, all results are WRONG (diagnostic for tools/ablate_lanes.py only)\n", e);
            return std::string(e ? e : "");
        }();
        if (!skip.empty()) {
            bool drop = false;
            for (size_t b = 0; b < skip.size() && !drop;) {
                size_t e = skip.find(',', b);
                if (e == std::string::npos) e = skip.size();
                if (e > b) {
                    std::string pat = skip.substr(b, e - b);
                    const bool anchored = pat.back() == '$';   // "name$": the step name must END with it
                    if (anchored) pat.pop_back();
                    const size_t at = anchored ? (st.name.size() >= pat.size() ? st.name.rfind(pat) : std::string::npos) : st.name.find(pat);
                    drop = at != std::string::npos && (!anchored || at + pat.size() == st.name.size());
                }
                b = e + 1;
            }
            if (drop) continue;
        }
        if (!branches) { st.fn(n, s); continue; }
        if (st.fork) {
            if (stream_wait(c, s, c->side_stream, ev)) side_open = true;
            else if (!side_open) branches = false;  // nothing pending on the side stream: carry on single-stream
        }
        if (st.join && side_open) {
            // a failed join would leave the side work unordered: fall back to a full wait outside a capture is impossible
            // here, so the error is kept in c->err and surfaces through run_sequence
            if (stream_wait(c, c->side_stream, s, ev)) side_open = false;
        }
        st.fn(n, (st.side && side_open) ? c->side_stream : s);
    }
    if (side_open) stream_wait(c, c->side_stream, s, ev);  // never leave the side stream dangling (a capture must be joined)
}

enum SeqKind { SEQ_BOOT = 0, SEQ_ITER, SEQ_REFINE, SEQ_FULL };

void enqueue_sequence(demon_ctx *c, int kind, int n, int iterations, hipStream_t s)
{
    size_t ev = 0;  // every fork / join of the sequence takes its own event
    if (kind == SEQ_BOOT || kind == SEQ_FULL) run_steps(c, c->net_boot, n, s, 0, ev);
    if (kind == SEQ_ITER) run_steps(c, c->net_iter, n, s, 0, ev);
    if (kind == SEQ_FULL)
        for (int i = 0; i < iterations; ++i) run_steps(c, c->net_iter, n, s, c->opt_reuse_image ? (i == 0 ? 1 : 2) : 0, ev);
    if (kind == SEQ_REFINE || kind == SEQ_FULL) run_steps(c, c->net_refine, n, s, 0, ev);
}

// one hipGraph per (sequence, batch, iterations): the whole kernel chain becomes a single launch
int run_sequence(demon_ctx *c, int kind, int n, int iterations)
{
    c->err.clear();
    if (!c->opt_hipgraph) {
        enqueue_sequence(c, kind, n, iterations, c->stream);
        if (hipGetLastError() != hipSuccess || !c->err.empty()) {
            if (c->err.empty()) c->err = "kernel launch failed";
            return DEMON_ERR_HIP;
        }
        return DEMON_OK;
    }
    char key[64];
    snprintf(key, sizeof key, "%d:%d:%d:%d:%d:%d:%d:%d", kind, n, iterations, c->opt_f2d_method, c->opt_reuse_image, c->opt_side_branches,
             c->opt_fused_pairs, c->opt_fused_inputs);
    auto it = c->graphs.find(key);
    if (it == c->graphs.end()) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        HIP_TRY(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        enqueue_sequence(c, kind, n, iterations, c->stream);
        HIP_TRY(c, hipStreamEndCapture(c->stream, &graph));
        if (!c->err.empty()) { hipGraphDestroy(graph); return DEMON_ERR_HIP; }
        HIP_TRY(c, hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        hipGraphDestroy(graph);
        it = c->graphs.emplace(key, exec).first;
    }
    HIP_TRY(c, hipGraphLaunch(it->second, c->stream));
    return DEMON_OK;
}

// fragment-order copies of the weights (conv_stream.hip) follow every weight change; done here, before a sequence is captured
void prepare_stream_weights(demon_ctx *c)
{
    for (auto &L : c->layers) refresh_stream_weights(L.get(), c->stream);
}

// a context's stream: on its CU mask when it has one (demon_set_cu_mask), else a plain non-blocking stream
hipError_t make_stream(const demon_ctx *c, hipStream_t *st)
{
    if (!c->cu_mask.empty()) return hipExtStreamCreateWithCUMask(st, (uint32_t)c->cu_mask.size(), c->cu_mask.data());
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}

void drop_group_graphs(demon_ctx *c)
{
    for (auto &g : c->group_graphs) hipGraphExecDestroy(g.second);
    c->group_graphs.clear();
}

// between demon_release_streams and demon_acquire_streams a context has no stream: nothing may fall onto the null stream
const char *const kNoStream = "the context gave its HIP streams back (demon_release_streams); call demon_acquire_streams first";

int check_batch(demon_ctx *c, int n)
{
    if (!c) return DEMON_ERR_INVALID;
    if (n < 1 || n > c->max_batch) return fail(c, DEMON_ERR_INVALID, "batch size out of range [1, max_batch]");
    if (!c->stream) return fail(c, DEMON_ERR_NOT_READY, kNoStream);
    std::string missing;
    if (!weights_ready(c, &missing)) return fail(c, DEMON_ERR_NOT_READY, "weights not set for layer " + missing);
    hipSetDevice(c->device);
    prepare_stream_weights(c);
    return DEMON_OK;
}

int h2d(demon_ctx *c, const View &v, const float *host, int n)
{
    if (!host) return fail(c, DEMON_ERR_INVALID, "null input pointer");
    if (!c->stream) return fail(c, DEMON_ERR_NOT_READY, kNoStream);
    const size_t row = sizeof(float) * (size_t)v.C * v.H * v.W;
    HIP_TRY(c, hipMemcpy2DAsync(v.ptr(), sizeof(float) * v.n_stride(), host, row, row, n, hipMemcpyHostToDevice, c->stream));
    return DEMON_OK;
}

int d2h(demon_ctx *c, float *host, const View &v, int n)
{
    if (!host) return DEMON_OK;
    if (!c->stream) return fail(c, DEMON_ERR_NOT_READY, kNoStream);
    const size_t row = sizeof(float) * (size_t)v.C * v.H * v.W;
    HIP_TRY(c, hipMemcpy2DAsync(host, row, v.ptr(), sizeof(float) * v.n_stride(), row, n, hipMemcpyDeviceToHost, c->stream));
    return DEMON_OK;
}

int download_outputs(demon_ctx *c, int n, const demon_outputs *o)
{
    if (!o) return DEMON_OK;
    int r;
    if ((r = d2h(c, o->predict_flow5, c->flowconf5.slice(0, 2), n))) return r;
    if ((r = d2h(c, o->predict_conf5, c->flowconf5.slice(2, 2), n))) return r;
    if ((r = d2h(c, o->predict_flow2, c->flowconf2.slice(0, 2), n))) return r;
    if ((r = d2h(c, o->predict_conf2, c->flowconf2.slice(2, 2), n))) return r;
    if ((r = d2h(c, o->predict_depth2, c->depth2, n))) return r;
    if ((r = d2h(c, o->predict_normal2, c->normal2, n))) return r;
    if (o->predict_rotation) HIP_TRY(c, hipMemcpyAsync(o->predict_rotation, c->d_rot, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, c->stream));
    if (o->predict_translation) HIP_TRY(c, hipMemcpyAsync(o->predict_translation, c->d_trans, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, c->stream));
    if (o->predict_scale) HIP_TRY(c, hipMemcpyAsync(o->predict_scale, c->d_scale, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    return DEMON_OK;
}

struct TmpDev {
    std::vector<void *> ptrs;
    ~TmpDev() { for (void *p : ptrs) hipFree(p); }
    float *alloc(size_t nfloats)
    {
        void *p = nullptr;
        if (hipMalloc(&p, sizeof(float) * (nfloats ? nfloats : 4)) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return (float *)p;
    }
    float *upload(const float *h, size_t nfloats)
    {
        float *d = alloc(nfloats);
        if (d && hipMemcpy(d, h, sizeof(float) * nfloats, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        return d;
    }
};

}  // namespace

// =====================================================================================================
extern "C" {

static int create_impl(demon_ctx **out, int device, int max_batch, int height, int width, int variant)
{
    if (!out) return fail(nullptr, DEMON_ERR_INVALID, "null ctx pointer");
    *out = nullptr;
    if (max_batch < 1 || height < 32 || width < 32 || height % 32 || width % 32)
        return fail(nullptr, DEMON_ERR_INVALID, "max_batch must be >= 1 and height/width positive multiples of 32");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, DEMON_ERR_HIP, "no HIP device visible (libdemon_hip.so needs an MI355X / gfx950 GPU)");
    if (device < 0 || device >= ndev) return fail(nullptr, DEMON_ERR_INVALID, "device index out of range");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, DEMON_ERR_HIP, "hipSetDevice failed");
    std::unique_ptr<demon_ctx> c(new demon_ctx);
    static std::atomic<unsigned long long> next_serial{1};
    c->serial = next_serial.fetch_add(1);
    c->device = device; c->max_batch = max_batch; c->H = height; c->W = width; c->variant = variant;
    c->guard_bytes = guard_bytes_from_env();   // poison harness: every allocation of this context between NaN canaries
    if (const char *fp = getenv("DEMON_FUSED_PAIRS")) c->opt_fused_pairs = atoi(fp) ? 1 : 0;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)
        return fail(nullptr, DEMON_ERR_HIP, "hipStreamCreate failed");
    demon_ctx *p = c.get();
    const int h2 = height / 4, w2 = width / 4, h5 = height / 32, w5 = width / 32;
    p->image_pair = buffer(p, "image_pair", 6, height, width);
    p->image2_2 = buffer(p, "image2_2", 3, h2, w2);
    p->flowconf5 = buffer(p, "flowconf5", 4, h5, w5);
    p->flowconf2 = buffer(p, "flowconf2", 4, h2, w2);
    // the depth+normal head writes [depth, normal xyz] into one 4-channel buffer; the state the next
    // stage reads (prev depth2 / normal2, blocks_original.py:180) are slices of it, so nothing is copied
    p->depth2 = buffer(p, "depthnormal2", 4, h2, w2).slice(0, 1);
    p->normal2 = buffer(p, "depthnormal2", 4, h2, w2).slice(1, 3);
    p->depth0 = buffer(p, "depth0", variant == 2 ? 4 : 1, height, width).slice(0, 1);
    p->normal0 = buffer(p, "depth0", variant == 2 ? 4 : 1, height, width).slice(1, 3);  // v2 only
    p->d_rot = dev_alloc(p, sizeof(float) * 3 * max_batch);
    p->d_trans = dev_alloc(p, sizeof(float) * 3 * max_batch);
    p->d_scale = dev_alloc(p, sizeof(float) * max_batch);
    p->d_motion = dev_alloc(p, sizeof(float) * 7 * max_batch);
    p->d_intrinsics = dev_alloc(p, sizeof(float) * 4 * max_batch);
    p->d_ws = alloc_splitk_workspace(p);
    p->d_ws_side = alloc_splitk_workspace(p);
    if (hipStreamCreateWithFlags(&p->side_stream, hipStreamNonBlocking) != hipSuccess) p->side_stream = nullptr;
    p->events.clear();  // fork / join events are created on demand (next_event)
    if (!p->d_ws_side) p->d_ws_side = p->d_ws, p->opt_side_branches = 0;
    if (!p->d_ws || !p->d_rot || !p->d_trans || !p->d_scale || !p->d_motion || !p->d_intrinsics) {
        demon_destroy(c.release());
        return fail(nullptr, DEMON_ERR_HIP, "device allocation failed");
    }
    {
        // networks_original.py:108-109
        std::vector<float> intr((size_t)4 * max_batch);
        for (int i = 0; i < max_batch; ++i) {
            intr[4 * i + 0] = 0.89115971f; intr[4 * i + 1] = 1.18821287f; intr[4 * i + 2] = 0.5f; intr[4 * i + 3] = 0.5f;
        }
        hipMemcpy(p->d_intrinsics, intr.data(), intr.size() * sizeof(float), hipMemcpyHostToDevice);
    }
    build_flow(p, &p->net_boot, "netFlow1", false);
    build_dm(p, &p->net_boot, "netDM1", false);
    build_flow(p, &p->net_iter, "netFlow2", true);
    build_dm(p, &p->net_iter, "netDM2", true);
    build_refine(p, &p->net_refine);
    if (p->err.empty() && !alloc_weight_slab(p)) p->err = "device allocation failed (weight slab)";
    if (!p->err.empty() || hipDeviceSynchronize() != hipSuccess) {
        std::string e = p->err.empty() ? "device error while building the networks" : p->err;
        demon_destroy(c.release());
        return fail(nullptr, DEMON_ERR_HIP, e);
    }
    *out = c.release();
    return DEMON_OK;
}

int demon_create(demon_ctx **out, int device, int max_batch, int height, int width)
{
    return create_impl(out, device, max_batch, height, width, 1);
}

int demon_create_v2(demon_ctx **out, int device, int max_batch, int height, int width)
{
    return create_impl(out, device, max_batch, height, width, 2);
}

int demon_create_ops(demon_ctx **out, int device)
{
    if (!out) return fail(nullptr, DEMON_ERR_INVALID, "null ctx pointer");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, DEMON_ERR_HIP, "no HIP device visible (libdemon_hip.so needs an MI355X / gfx950 GPU)");
    if (device < 0 || device >= ndev) return fail(nullptr, DEMON_ERR_INVALID, "device index out of range");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, DEMON_ERR_HIP, "hipSetDevice failed");
    std::unique_ptr<demon_ctx> c(new demon_ctx);
    c->device = device; c->max_batch = 0; c->variant = 0; c->opt_side_branches = 0;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)
        return fail(nullptr, DEMON_ERR_HIP, "hipStreamCreate failed");
    c->d_ws = alloc_splitk_workspace(c.get());
    c->d_ws_side = c->d_ws;
    if (!c->d_ws) {
        demon_destroy(c.release());
        return fail(nullptr, DEMON_ERR_HIP, "device allocation failed");
    }
    *out = c.release();
    return DEMON_OK;
}

int demon_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int demon_variant(const demon_ctx *c) { return c ? c->variant : 0; }

int demon_destroy(demon_ctx *c)
{
    if (!c) return DEMON_OK;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto &g : c->graphs) hipGraphExecDestroy(g.second);
    drop_group_graphs(c);
    for (hipEvent_t e : c->group_events) hipEventDestroy(e);
    if (c->side_stream) { hipStreamSynchronize(c->side_stream); hipStreamDestroy(c->side_stream); }
    for (hipStream_t st : c->tune_streams) { hipStreamSynchronize(st); hipStreamDestroy(st); }
    for (hipStream_t st : c->placeholder_streams) hipStreamDestroy(st);
    for (hipEvent_t e : c->tune_events) hipEventDestroy(e);
    for (hipEvent_t e : c->events) if (e) hipEventDestroy(e);
    for (void *p : c->allocations) hipFree(p);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return DEMON_OK;
}

const char *demon_last_error(const demon_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }
int demon_device(const demon_ctx *c) { return c ? c->device : -1; }

int demon_num_variables(const demon_ctx *c) { return c ? (int)c->variables.size() : 0; }

int demon_variable_info(const demon_ctx *c, int index, char *name, int name_cap, int64_t dims[4], int *ndim)
{
    if (!c || index < 0 || index >= (int)c->variables.size()) return DEMON_ERR_INVALID;
    const Variable &v = c->variables[index];
    if (name && name_cap > 0) { strncpy(name, v.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    for (int i = 0; i < 4; ++i) dims[i] = i < v.ndim ? v.dims[i] : 1;
    if (ndim) *ndim = v.ndim;
    return DEMON_OK;
}

int demon_set_weight(demon_ctx *c, const char *tf_name, const float *host, const int64_t *dims, int ndim)
{
    if (!c || !tf_name || !host) return fail(c, DEMON_ERR_INVALID, "null argument");
    auto it = c->var_index.find(tf_name);
    if (it == c->var_index.end()) return fail(c, DEMON_ERR_NOT_FOUND, std::string("unknown variable ") + tf_name);
    const Variable &v = c->variables[it->second];
    if (dims) {
        bool same = ndim == v.ndim;
        for (int i = 0; same && i < ndim; ++i) same = dims[i] == v.dims[i];
        if (!same) return fail(c, DEMON_ERR_INVALID, std::string("shape mismatch for variable ") + tf_name);
    }
    hipSetDevice(c->device);
    for (auto &g : c->graphs) hipGraphExecDestroy(g.second);  // packed pointers stay, but be safe
    c->graphs.clear();
    if (v.is_bias) {
        HIP_TRY(c, hipMemcpy(v.layer->d_bias, host, sizeof(float) * v.layer->Cout, hipMemcpyHostToDevice));
        v.layer->have_bias = true;
        return DEMON_OK;
    }
    return upload_kernel(c, v.layer, host);
}

int64_t demon_weights_blob_size(const demon_ctx *c)
{
    if (!c) return 0;
    int64_t total = 0;
    for (const Variable &v : c->variables) total += v.count;
    return total;
}

int demon_set_weights_blob(demon_ctx *c, const float *blob, int64_t nfloats)
{
    if (!c || !blob) return fail(c, DEMON_ERR_INVALID, "null argument");
    if (nfloats != demon_weights_blob_size(c)) return fail(c, DEMON_ERR_INVALID, "weight blob size mismatch");
    int64_t off = 0;
    for (const Variable &v : c->variables) {
        int r = demon_set_weight(c, v.name.c_str(), blob + off, v.dims, v.ndim);
        if (r) return r;
        off += v.count;
    }
    return DEMON_OK;
}

int demon_set_weights_blob_device(demon_ctx *c, const void *dblob, int64_t nfloats)
{
    if (!c || !dblob) return fail(c, DEMON_ERR_INVALID, "null argument");
    if (nfloats != demon_weights_blob_size(c)) return fail(c, DEMON_ERR_INVALID, "weight blob size mismatch");
    hipSetDevice(c->device);
    std::vector<float> host((size_t)nfloats);
    HIP_TRY(c, hipMemcpy(host.data(), dblob, sizeof(float) * (size_t)nfloats, hipMemcpyDeviceToHost));
    return demon_set_weights_blob(c, host.data(), nfloats);
}

// ---- multi-GPU: RCCL through the C ABI ---------------------------------------------------------------------
// One process per GPU.  The path shards by image pair with no data-path collective; the one
// collective is the broadcast of the weights at start-up.  librccl.so is loaded on first use (dlopen), so single-GPU
// users of libdemon_hip.so have no RCCL link dependency; rccl.h supplies types and prototypes only.
}  // extern "C"
namespace {
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};
static_assert(sizeof(ncclUniqueId) == DEMON_COMM_ID_BYTES, "include/demon_hip.h: DEMON_COMM_ID_BYTES must match ncclUniqueId");

RcclApi &rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // An RCCL that is already in the process is the one to use (PyTorch brings its own copy, torch/lib/librccl.so: two copies of the library in
        // one process answer each other's symbols and end in "double free or corruption" at exit -- seen in round 6 when this loader ran BEFORE
        // `import torch` and had put /opt/rocm's copy into the global scope).  So: first ask for a loaded one (RTLD_NOLOAD), then load one --
        // RTLD_LOCAL, so that a host that imports PyTorch afterwards still gets PyTorch's copy for PyTorch.
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *name : names) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
            if (api.lib) break;
        }
        for (const char *name : names) {
            if (api.lib) break;
            api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!api.lib) { api.err = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : ""); return; }
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
        api.Broadcast = (decltype(api.Broadcast))dlsym(api.lib, "ncclBroadcast");
        api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
        api.CommCount = (decltype(api.CommCount))dlsym(api.lib, "ncclCommCount");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.Broadcast || !api.AllReduce || !api.CommCount)
            api.err = "librccl.so lacks the nccl* entry points";
    });
    return api;
}
int rccl_fail(demon_ctx *c, const char *what, int rc)
{
    RcclApi &r = rccl();
    std::string msg = std::string(what) + ": " + (rc && r.GetErrorString ? r.GetErrorString((ncclResult_t)rc) : r.err.c_str());
    return fail(c, DEMON_ERR_HIP, msg);
}
}  // namespace
extern "C" {

int demon_comm_get_unique_id(char *id)
{
    RcclApi &r = rccl();
    if (!id) return fail(nullptr, DEMON_ERR_INVALID, "null id");
    if (!r.err.empty()) return rccl_fail(nullptr, "rccl", 0);
    ncclUniqueId u;
    memset(&u, 0, sizeof u);
    int rc = r.GetUniqueId(&u);
    if (rc) return rccl_fail(nullptr, "ncclGetUniqueId", rc);
    memcpy(id, u.internal, DEMON_COMM_ID_BYTES);
    return DEMON_OK;
}

int demon_comm_init_rank(void **nccl_comm, int nranks, const char *id, int rank, int device)
{
    RcclApi &r = rccl();
    if (!nccl_comm || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(nullptr, DEMON_ERR_INVALID, "bad argument");
    *nccl_comm = nullptr;
    if (!r.err.empty()) return rccl_fail(nullptr, "rccl", 0);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, DEMON_ERR_HIP, "hipSetDevice failed");
    ncclUniqueId u;
    memcpy(u.internal, id, DEMON_COMM_ID_BYTES);
    int rc = r.CommInitRank((ncclComm_t *)nccl_comm, nranks, u, rank);
    if (rc) return rccl_fail(nullptr, "ncclCommInitRank", rc);
    return DEMON_OK;
}

int demon_comm_destroy(void *nccl_comm)
{
    RcclApi &r = rccl();
    if (!nccl_comm) return DEMON_OK;
    if (!r.err.empty()) return rccl_fail(nullptr, "rccl", 0);
    int rc = r.CommDestroy((ncclComm_t)nccl_comm);
    return rc ? rccl_fail(nullptr, "ncclCommDestroy", rc) : DEMON_OK;
}

int demon_comm_count(void *nccl_comm, int *nranks)
{
    RcclApi &r = rccl();
    if (!nccl_comm || !nranks) return fail(nullptr, DEMON_ERR_INVALID, "null argument");
    if (!r.err.empty()) return rccl_fail(nullptr, "rccl", 0);
    int rc = r.CommCount((ncclComm_t)nccl_comm, nranks);
    return rc ? rccl_fail(nullptr, "ncclCommCount", rc) : DEMON_OK;
}
}  // extern "C"
namespace {
// Identity of a context's packed weight slab: model variant, image size and, per layer, name, class count, packed rows / columns
// and the offsets of its kernel and bias inside the slab (FNV-1a).  Two contexts with equal values can exchange slabs byte for byte.
uint64_t slab_layout_hash(const demon_ctx *c)
{
    uint64_t h = 1469598103934665603ull;
    auto mix = [&h](uint64_t v) { for (int i = 0; i < 8; ++i) { h ^= (v >> (8 * i)) & 0xff; h *= 1099511628211ull; } };
    mix((uint64_t)c->variant); mix((uint64_t)c->H); mix((uint64_t)c->W); mix((uint64_t)c->w_slab_floats);
    for (auto &L : c->layers) {
        for (char ch : L->name) mix((uint64_t)(unsigned char)ch);
        mix((uint64_t)L->ncls); mix((uint64_t)L->Krows); mix((uint64_t)L->Mpad);
        mix((uint64_t)(L->d_wp - c->w_slab)); mix((uint64_t)(L->d_bias - c->w_slab));
    }
    return h;
}
// What a receiver does once a packed slab has arrived from elsewhere (another rank's broadcast, another context's copy): every
// variable counts as set -- no demon_set_weight ever ran here -- and the fragment-order copies of the conv_stream / conv_frag
// layers are stale; captured graphs keep their pointers but are dropped like after any other weight change.
void slab_arrived(demon_ctx *c)
{
    for (auto &g : c->graphs) hipGraphExecDestroy(g.second);
    c->graphs.clear();
    for (auto &L : c->layers) { L->have_kernel = L->have_bias = true; L->wf_dirty = true; L->w1_dirty = true; L->w3_dirty = true; L->w4_dirty = true; L->wd_dirty = true; }
}
}  // namespace
extern "C" {

uint64_t demon_weights_slab_layout(const demon_ctx *c) { return c && c->w_slab ? slab_layout_hash(c) : 0; }

// One ncclBroadcast of the packed weight slab (device to device over xGMI; no host staging, no repacking on the receivers:
// every rank built the same layer table, so the slab layout is identical -- which is CHECKED first: the root's layout hash is
// broadcast, every rank compares it with its own, and a MIN all-reduce of the verdicts makes all ranks refuse together instead
// of leaving some of them inside the big collective).  The root must have all weights set.
int demon_broadcast_weights(demon_ctx *c, void *nccl_comm, int root, int rank)
{
    if (!c || !nccl_comm) return fail(c, DEMON_ERR_INVALID, "null argument");
    if (!c->w_slab) return fail(c, DEMON_ERR_INVALID, "context has no networks (demon_create_ops)");
    if (!c->stream) return fail(c, DEMON_ERR_NOT_READY, kNoStream);
    RcclApi &r = rccl();
    if (!r.err.empty()) return rccl_fail(c, "rccl", 0);
    std::string missing;
    const bool root_ready = rank != root || weights_ready(c, &missing);
    hipSetDevice(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // header: [layout hash lo, hi, root ready] from the root, then the agreement flag
    TmpDev tmp;
    int *hdr = (int *)tmp.alloc(4);
    if (!hdr) return fail(c, DEMON_ERR_HIP, "temporary device allocation failed");
    const uint64_t mine = slab_layout_hash(c);
    int h_hdr[4] = {(int)(mine & 0xffffffffu), (int)(mine >> 32), root_ready ? 1 : 0, 0};
    HIP_TRY(c, hipMemcpyAsync(hdr, h_hdr, sizeof h_hdr, hipMemcpyHostToDevice, c->stream));
    int rc = r.Broadcast(hdr, hdr, 3, ncclInt32, root, (ncclComm_t)nccl_comm, c->stream);
    if (rc) return rccl_fail(c, "ncclBroadcast (header)", rc);
    int got[4] = {0, 0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(got, hdr, sizeof got, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const bool same = got[0] == h_hdr[0] && got[1] == h_hdr[1];
    int verdict = (same ? 1 : 0) | (got[2] ? 2 : 0);  // bit 0: my layout equals the root's, bit 1: the root has its weights
    HIP_TRY(c, hipMemcpyAsync(hdr + 3, &verdict, sizeof verdict, hipMemcpyHostToDevice, c->stream));
    rc = r.AllReduce(hdr + 3, hdr + 3, 1, ncclInt32, ncclMin, (ncclComm_t)nccl_comm, c->stream);
    if (rc) return rccl_fail(c, "ncclAllReduce (agreement)", rc);
    int all = 0;
    HIP_TRY(c, hipMemcpyAsync(&all, hdr + 3, sizeof all, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!got[2]) return fail(c, DEMON_ERR_NOT_READY, rank == root ? "root rank: weights not set for layer " + missing : std::string("root rank: weights not set"));
    if (!same) return fail(c, DEMON_ERR_INVALID, "weight slab layout differs from the root's (other model variant, image size or library build)");
    if ((all & 1) == 0) return fail(c, DEMON_ERR_INVALID, "another rank's weight slab layout differs from the root's: broadcast refused on all ranks");
    rc = r.Broadcast(c->w_slab, c->w_slab, c->w_slab_floats, ncclFloat32, root, (ncclComm_t)nccl_comm, c->stream);
    if (rc) return rccl_fail(c, "ncclBroadcast", rc);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    slab_arrived(c);
    return DEMON_OK;
}

// The receiver side of demon_broadcast_weights without a second GPU: the packed slab of `src` (same model variant / image size,
// any device of this process) is copied device to device into `dst`, which then does exactly what a non-root rank does after the
// collective.  `dst` needs no demon_set_weight call at all.
int demon_copy_weights_from(demon_ctx *dst, const demon_ctx *src)
{
    if (!dst || !src) return fail(dst, DEMON_ERR_INVALID, "null argument");
    if (!dst->w_slab || !src->w_slab) return fail(dst, DEMON_ERR_INVALID, "context has no networks (demon_create_ops)");
    if (slab_layout_hash(dst) != slab_layout_hash(src)) return fail(dst, DEMON_ERR_INVALID, "weight slab layouts differ (other model variant or image size)");
    if (!dst->stream || !src->stream) return fail(dst, DEMON_ERR_NOT_READY, kNoStream);
    std::string missing;
    if (!weights_ready(const_cast<demon_ctx *>(src), &missing)) return fail(dst, DEMON_ERR_NOT_READY, "source context: weights not set for layer " + missing);
    hipSetDevice(src->device);
    HIP_TRY(dst, hipStreamSynchronize(src->stream));
    hipSetDevice(dst->device);
    HIP_TRY(dst, hipStreamSynchronize(dst->stream));
    HIP_TRY(dst, hipMemcpyPeerAsync(dst->w_slab, dst->device, src->w_slab, src->device, sizeof(float) * dst->w_slab_floats, dst->stream));
    HIP_TRY(dst, hipStreamSynchronize(dst->stream));
    slab_arrived(dst);
    return DEMON_OK;
}

int64_t demon_weights_slab_bytes(const demon_ctx *c) { return c ? (int64_t)(c->w_slab_floats * sizeof(float)) : 0; }

int demon_set_option(demon_ctx *c, const char *key, int value)
{
    if (!c || !key) return DEMON_ERR_INVALID;
    if (!strcmp(key, "hipgraph")) { c->opt_hipgraph = value ? 1 : 0; return DEMON_OK; }
    if (!strcmp(key, "flow_to_depth_method")) {
        if (value != 0 && value != 1) return fail(c, DEMON_ERR_INVALID, "flow_to_depth_method must be 0 or 1");
        c->opt_f2d_method = value;
        return DEMON_OK;
    }
    if (!strcmp(key, "reuse_image_features")) { c->opt_reuse_image = value ? 1 : 0; return DEMON_OK; }
    if (!strcmp(key, "fused_pairs")) { c->opt_fused_pairs = value ? 1 : 0; return DEMON_OK; }
    if (!strcmp(key, "fused_inputs")) { c->opt_fused_inputs = value ? 1 : 0; return DEMON_OK; }
    if (!strcmp(key, "tune_lanes")) {
        if (value < 1 || value > 8) return fail(c, DEMON_ERR_INVALID, "tune_lanes must be in [1, 8]");
        c->opt_tune_lanes = value;
        if (value == 1) {   // back to one pass at a time: the extra streams of the concurrent replays go (they take part in the stream -> queue mapping)
            for (hipStream_t st : c->tune_streams) { hipStreamSynchronize(st); hipStreamDestroy(st); }
            c->tune_streams.clear();
        }
        return DEMON_OK;
    }
    if (!strcmp(key, "side_branches")) { c->opt_side_branches = (value && c->side_stream && c->d_ws_side != c->d_ws) ? 1 : 0; return DEMON_OK; }
    return fail(c, DEMON_ERR_NOT_FOUND, std::string("unknown option ") + key);
}

int demon_get_option(const demon_ctx *c, const char *key, int *value)
{
    if (!c || !key || !value) return DEMON_ERR_INVALID;
    if (!strcmp(key, "hipgraph")) *value = c->opt_hipgraph;
    else if (!strcmp(key, "flow_to_depth_method")) *value = c->opt_f2d_method;
    else if (!strcmp(key, "reuse_image_features")) *value = c->opt_reuse_image;
    else if (!strcmp(key, "fused_pairs")) *value = c->opt_fused_pairs;
    else if (!strcmp(key, "fused_inputs")) *value = c->opt_fused_inputs;
    else if (!strcmp(key, "tune_lanes")) *value = c->opt_tune_lanes;
    else if (!strcmp(key, "side_branches")) *value = c->opt_side_branches;
    else return DEMON_ERR_NOT_FOUND;
    return DEMON_OK;
}

// A chainable pair: the two separately tuned launches against every chained variant that fits; the k x 1 layer's plan becomes
// kind 6 / 7 when a chain is faster.
int autotune_chain(demon_ctx *c, Layer *Ly, Layer *Lx, int n)
{
    bool failed = false;
    auto measure = [&](const std::function<void()> &body) -> float { return time_replay(c, body, failed); };
    float best = measure([&] { run_layer(Ly, n, c->stream, c->d_ws); run_layer(Lx, n, c->stream, c->d_ws); });
    Layer::Tuned best_t{-1, 0, 0};
    for (int kind : {6, 7})
        for (int v = 0; v < (kind == 6 ? (int)FRAG_VARIANTS : (int)STREAM_VARIANTS) && !failed; ++v) {
            if (!chain_shape_ok(Ly, Lx, kind, v)) continue;
            const float ms = measure([&] { run_chain(Ly, Lx, n, kind, v, c->stream, c->d_ws); });
            if (ms < best) { best = ms; best_t = Layer::Tuned{kind, v, 1}; }
        }
    if (failed) return DEMON_ERR_HIP;
    if (best_t.kind > 0) Ly->tuned[n] = best_t;
    return DEMON_OK;
}

// A pair conv_pair.hip can fuse whose k x 1 layer also has the weights-in-registers kernel (kind 12): one launch or two?  Measured
// like autotune_layer does (a replayed hipGraph of five repetitions).  Two launches win: the k x 1 layer's plan becomes kind 12 (which
// is what selects that form); the fused launch wins: it keeps / gets back its best other kernel.
int autotune_fused_pair(demon_ctx *c, Layer *Ly, Layer *Lx, int n)
{
    if (!thin_applies(Ly) || !c->opt_fused_pairs) return DEMON_OK;
    bool failed = false;
    auto measure = [&](bool fused) -> float {
        return time_replay(c, [&] {
            if (fused && run_pair(Ly, Lx, n, c->stream)) return;
            run_layer(Ly, n, c->stream, c->d_ws);
            run_layer(Lx, n, c->stream, c->d_ws);
        }, failed);
    };
    const bool had = Ly->tuned.count(n) > 0;
    const Layer::Tuned before = had ? Ly->tuned[n] : Layer::Tuned{14, 0, 1};
    Ly->tuned[n] = Layer::Tuned{12, 0, 1};
    const float two = measure(false);
    const float one = measure(true);
    if (failed) return DEMON_ERR_HIP;
    if (one <= two) {
        // the fused launch wins: the k x 1 layer gets back the entry it had, or -- when nothing else was measured for it -- the explicit
        // marker kind 14 ("pair runs fused at this batch size; alone, the layer is served by the heuristics"), so that nearest_tuned()
        // cannot pick up another batch size's kind 12 for a pair that just measured faster as one launch
        Ly->tuned[n] = (had && before.kind != 12) ? before : Layer::Tuned{14, 0, 1};
    }
    return DEMON_OK;
}

int demon_autotune(demon_ctx *c, int n)
{
    if (!c || n < 1 || n > c->max_batch) return fail(c, DEMON_ERR_INVALID, "batch size out of range [1, max_batch]");
    if (!c->stream) return fail(c, DEMON_ERR_NOT_READY, kNoStream);   // (the repack kernels and every timed replay run on the context's stream)
    hipSetDevice(c->device);
    for (auto &g : c->graphs) hipGraphExecDestroy(g.second);  // captured launches embed the old choices
    c->graphs.clear();
    prepare_stream_weights(c);
    // DEMON_TUNE_ONLY=<substring>[,<substring> ...]: re-tune only the layers whose name contains one of them (the others keep their installed plan entries)
    const char *only_env = getenv("DEMON_TUNE_ONLY");
    std::vector<std::string> only_list;
    for (std::string rest = only_env ? only_env : ""; !rest.empty();) {
        const size_t e = rest.find(',');
        if (e) only_list.push_back(rest.substr(0, e));
        rest = e == std::string::npos ? "" : rest.substr(e + 1);
    }
    struct OnlyFilter {
        const std::vector<std::string> &pats;
        bool any;
        bool skips(const std::string &name) const
        {
            if (!any) return false;
            for (const std::string &p : pats) if (name.find(p) != std::string::npos) return false;
            return true;
        }
    } filter{only_list, !only_list.empty()};
    for (auto &L : c->layers) {
        if (filter.skips(L->name)) continue;
        int r = autotune_layer(c, L.get(), n);
        if (r) return fail(c, r, "autotune failed at layer " + L->name);
    }
    for (auto &pr : c->chain_pairs) {
        if (filter.skips(pr.first->name)) continue;
        int r = autotune_chain(c, pr.first, pr.second, n);
        if (r) return fail(c, r, "autotune failed at pair " + pr.first->name);
    }
    for (auto &pr : c->fused_pairs) {
        if (filter.skips(pr.first->name)) continue;
        int r = autotune_fused_pair(c, pr.first, pr.second, n);
        if (r) return fail(c, r, "autotune failed at pair " + pr.first->name);
    }
    return DEMON_OK;
}

int demon_num_layers(const demon_ctx *c) { return c ? (int)c->layers.size() : 0; }

int demon_plan_get(const demon_ctx *c, int n, int layer_index, char *name, int name_cap, int *kind, int *tile, int *ksplit)
{
    if (!c || layer_index < 0 || layer_index >= (int)c->layers.size()) return DEMON_ERR_INVALID;
    const Layer *L = c->layers[layer_index].get();
    if (name && name_cap > 0) { strncpy(name, L->name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    auto it = L->tuned.find(n);
    if (it == L->tuned.end()) return DEMON_ERR_NOT_FOUND;
    if (kind) *kind = it->second.kind;
    if (tile) *tile = it->second.tile;
    if (ksplit) *ksplit = it->second.ksplit;
    return DEMON_OK;
}

int demon_plan_clear(demon_ctx *c, int n)
{
    if (!c || n < 1 || n > c->max_batch) return fail(c, DEMON_ERR_INVALID, "batch size out of range [1, max_batch]");
    for (auto &g : c->graphs) hipGraphExecDestroy(g.second);   // captured launches embed the old choices
    c->graphs.clear();
    for (auto &L : c->layers) L->tuned.erase(n);
    return DEMON_OK;
}

int demon_plan_set(demon_ctx *c, int n, const char *layer_name, int kind, int tile, int ksplit)
{
    if (!c || !layer_name || n < 1 || n > c->max_batch) return fail(c, DEMON_ERR_INVALID, "bad argument");
    // kinds: 0 im2col, 1 patch-staged, 3 small-Cout, 4 streaming, 5 fragment-tiled, 6 / 7 = kind 5 / 4 chained with the 1 x k partner
    // 8 = minimal-filtering transposed conv, 10 = 1-D minimal filtering (conv_wino.hip); 9 = the removed F(2x2,3x3) kernel (docs/experiments)
    // 11 = weight-streaming dense layer (dense_stream.hip; tile 0 / 1 = default / non-temporal weight loads)
    // 12 = the blocks' first layer with the weights in registers (conv_thin.hip; tile 0); its pair then runs as two launches
    // 13 = 1 x 7 / 1 x 9 stride-2 conv with <= 32 channels, whole reduction out of LDS (conv_row.hip; tile 0)
    // 14 = marker on the k x 1 layer of a conv_pair.hip pair: the fused launch was measured faster at this batch size (the layer alone: heuristics)
    // 15 = 3 x 3 stride-1 conv, transformed input rows stationary (conv_wino3.hip; tile = workgroup shape)
    // 16 = k x 1 / 1 x k conv with four outputs per window (conv_wino4.hip: 3 taps stride 1, 5 taps stride 2; tile = workgroup shape)
    if (kind < 0 || kind > 16 || kind == 2 || kind == 9 || tile < 0 ||
        tile >= (kind == 16 ? (int)WINO4_VARIANTS : kind == 15 ? (int)WINO3_VARIANTS : kind >= 12 ? 1 : kind == 11 ? (int)DENSE_VARIANTS : kind == 10 ? (int)WINO1D_VARIANTS : kind == 8 ? (int)WINO_VARIANTS : (kind == 1 ? (int)PTILE_COUNT : ((kind == 4 || kind == 7) ? (int)STREAM_VARIANTS : ((kind == 5 || kind == 6) ? (int)FRAG_VARIANTS : (int)TILE_COUNT)))) || ksplit < 0)
        return fail(c, DEMON_ERR_INVALID, "bad plan entry");
    if (kind != 1 && ksplit >= 1000)   // (until round 3 "+ 1000" on kinds 0 / 4 / 5 selected a split-K form that no longer exists)
        return fail(c, DEMON_ERR_INVALID, "ksplit >= 1000 is only meaningful for the patch-staged kernel (pixel-tile shape)");
    for (auto &L : c->layers)
        if (L->name == layer_name) {
            if (kind == 0 && L->Mpad % conv_tile_bm(tile)) return fail(c, DEMON_ERR_INVALID, "tile does not divide Cout");
            if (kind == 3 && !small_applies(L.get())) return fail(c, DEMON_ERR_INVALID, "the small-Cout kernel does not apply to this layer");
            if (kind == 8 && !wino_applies(L.get())) return fail(c, DEMON_ERR_INVALID, "the minimal-filtering kernel applies to transposed convs only");
            if (kind == 10 && !wino1d_applies(L.get())) return fail(c, DEMON_ERR_INVALID, "no 1-D minimal-filtering form for this layer");
            if (kind == 16 && !wino4_applies(L.get())) return fail(c, DEMON_ERR_INVALID, "conv_wino4.hip applies to k x 1 / 1 x k convs with 3 taps stride 1 or 5 taps stride 2 and >= 16 input channels only");
            if (kind == 15 && !wino3_applies(L.get())) return fail(c, DEMON_ERR_INVALID, "conv_wino3.hip applies to 3 x 3 convs with >= 16 input channels, stride 1 (rows of even length) or stride 2 (rows of a multiple of 8 pixels, >= 64)");
            if (kind == 13 && !row_applies(L.get())) return fail(c, DEMON_ERR_INVALID, "conv_row.hip applies to 1 x 7 / 1 x 9 stride-2 convs with at most 32 channels on both sides only");
            if (kind == 12 && !thin_applies(L.get())) return fail(c, DEMON_ERR_INVALID, "conv_thin.hip applies to the 9 x 1 stride-2 first layer (Cin <= 6, Cout <= 32) only");
            if (kind == 11 && !dense_stream_applies(L.get())) return fail(c, DEMON_ERR_INVALID, "the weight-streaming kernel applies to dense layers only");
            if (kind == 4 && (!L->stream_ok() || L->Mpad % stream_variant_bm(tile))) return fail(c, DEMON_ERR_INVALID, "the streaming kernel does not apply to this layer");
            if (kind == 5 && (!L->stream_ok() || L->Mpad % frag_variant_bm(tile))) return fail(c, DEMON_ERR_INVALID, "the fragment-tiled kernel does not apply to this layer");
            if (kind == 6 || kind == 7) {
                bool ok = false;
                for (auto &pr : c->chain_pairs) ok = ok || (pr.first == L.get() && chain_shape_ok(pr.first, pr.second, kind, tile));
                if (!ok) return fail(c, DEMON_ERR_INVALID, "this layer cannot run chained with a 1 x k partner on that variant");
            }
            for (auto &g : c->graphs) hipGraphExecDestroy(g.second);
            c->graphs.clear();
            L->tuned[n] = Layer::Tuned{kind, tile, (kind == 0 || kind == 4 || kind == 5) && ksplit < 1 ? 1 : ksplit};
            return DEMON_OK;
        }
    return fail(c, DEMON_ERR_NOT_FOUND, std::string("unknown layer ") + layer_name);
}

int demon_upload_inputs(demon_ctx *c, int n, const float *image_pair, const float *image2_2)
{
    int r = check_batch(c, n);
    if (r) return r;
    hipSetDevice(c->device);
    if ((r = h2d(c, c->image_pair, image_pair, n))) return r;
    if ((r = h2d(c, c->image2_2, image2_2, n))) return r;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DEMON_OK;
}

// Asynchronous variants for pipelines that overlap the copies of one context with the kernels of another (two contexts, two
// streams): nothing here waits; host buffers must be page-locked (demon_host_register) and stay valid until demon_synchronize.
int demon_upload_inputs_async(demon_ctx *c, int n, const float *image_pair, const float *image2_2)
{
    int r = check_batch(c, n);
    if (r) return r;
    hipSetDevice(c->device);
    if ((r = h2d(c, c->image_pair, image_pair, n))) return r;
    return h2d(c, c->image2_2, image2_2, n);
}

int demon_download_outputs_async(demon_ctx *c, int n, const demon_outputs *o, float *depth0)
{
    if (!c || n < 1 || n > c->max_batch) return fail(c, DEMON_ERR_INVALID, "bad batch");
    hipSetDevice(c->device);
    int r = download_outputs(c, n, o);
    if (r) return r;
    return d2h(c, depth0, c->depth0, n);
}

int demon_host_register(void *ptr, int64_t bytes)
{
    if (!ptr || bytes <= 0) return DEMON_ERR_INVALID;
    return hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault) == hipSuccess ? DEMON_OK : DEMON_ERR_HIP;
}

int demon_host_unregister(void *ptr)
{
    if (!ptr) return DEMON_ERR_INVALID;
    return hipHostUnregister(ptr) == hipSuccess ? DEMON_OK : DEMON_ERR_HIP;
}

int demon_run_full(demon_ctx *c, int n, int iterations)
{
    int r = check_batch(c, n);
    if (r) return r;
    if (iterations < 0 || iterations > 64) return fail(c, DEMON_ERR_INVALID, "iterations out of range");
    hipSetDevice(c->device);
    return run_sequence(c, SEQ_FULL, n, iterations);
}

int demon_run_bootstrap(demon_ctx *c, int n)
{
    int r = check_batch(c, n);
    if (r) return r;
    hipSetDevice(c->device);
    return run_sequence(c, SEQ_BOOT, n, 0);
}

int demon_synchronize(demon_ctx *c)
{
    if (!c) return DEMON_ERR_INVALID;
    hipSetDevice(c->device);
    if (c->stream) HIP_TRY(c, hipStreamSynchronize(c->stream));   // (no stream between release and acquire: nothing is in flight)
    return DEMON_OK;
}

// The runtime binds a HIP stream to one of a few hardware queues when the stream is created, by a rule that depends on every stream
// alive in the process; two busy streams on one queue serialise.  A lane group (demon_amd/lanes.py) that measures a poor mapping
// gives its streams back (demon_release_streams on every lane), optionally creates a few placeholder streams, and takes new ones
// (demon_acquire_streams, lane by lane).  Cached hipGraph execs are dropped and captured again on the new streams at the next run call.
int demon_release_streams(demon_ctx *c)
{
    if (!c) return DEMON_ERR_INVALID;
    hipSetDevice(c->device);
    if (c->stream) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipStreamDestroy(c->stream)); c->stream = nullptr; }
    if (c->side_stream) { hipStreamSynchronize(c->side_stream); hipStreamDestroy(c->side_stream); c->side_stream = nullptr; }
    // graph execs captured across BOTH streams (side branches) do not survive the streams they were captured on: launching one after
    // the exchange aborted inside the HIP runtime ("pure virtual method called", round 5, tools/e2e_bisect.py).  All cached execs go;
    // the next run call captures again on the new streams (a few milliseconds, set-up time like the exchange itself).
    for (auto &g : c->graphs) hipGraphExecDestroy(g.second);
    c->graphs.clear();
    drop_group_graphs(c);
    // the throughput-mode tuner's streams take part in the stream -> hardware-queue mapping as well (re-created on demand)
    for (hipStream_t st : c->tune_streams) { hipStreamSynchronize(st); hipStreamDestroy(st); }
    c->tune_streams.clear();
    return DEMON_OK;
}

int demon_acquire_streams(demon_ctx *c)
{
    if (!c) return DEMON_ERR_INVALID;
    hipSetDevice(c->device);
    if (!c->stream) HIP_TRY(c, make_stream(c, &c->stream));
    if (!c->side_stream && c->variant != 0 && make_stream(c, &c->side_stream) != hipSuccess) {
        c->side_stream = nullptr;
        c->opt_side_branches = 0;
    }
    return DEMON_OK;
}

// ---- lanes: several contexts on one GPU, their streams mapped onto hardware queues by measurement (demon_amd/lanes.py is the
// Python face of these two; INTEGRATION.md section 6 shows them from C)
namespace {
int lanes_check(demon_ctx *const *ctxs, int nctx)
{
    if (!ctxs || nctx < 1 || nctx > DEMON_LANES_MAX || !ctxs[0]) return DEMON_ERR_INVALID;
    for (int i = 0; i < nctx; ++i) {
        if (!ctxs[i]) return fail(ctxs[0], DEMON_ERR_INVALID, "null context in the lane array");
        if (ctxs[i]->device != ctxs[0]->device) return fail(ctxs[0], DEMON_ERR_INVALID, "the lanes of a group live on one device");
        for (int j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) return fail(ctxs[0], DEMON_ERR_INVALID, "the same context twice in the lane array");
    }
    return DEMON_OK;
}
}  // namespace

int demon_lanes_apply(demon_ctx *const *ctxs, int nctx, int placeholder_streams)
{
    int r = lanes_check(ctxs, nctx);
    if (r) return r;
    demon_ctx *c0 = ctxs[0];
    if (placeholder_streams < 0 || placeholder_streams > DEMON_LANES_MAX_PLACEHOLDERS) return fail(c0, DEMON_ERR_INVALID, "placeholder_streams out of range");
    hipSetDevice(c0->device);
    for (int i = 0; i < nctx; ++i)
        if ((r = demon_release_streams(ctxs[i]))) return r == DEMON_ERR_HIP ? fail(c0, r, "lane " + std::to_string(i) + ": " + ctxs[i]->err) : r;
    for (hipStream_t st : c0->placeholder_streams) hipStreamDestroy(st);
    c0->placeholder_streams.clear();
    for (int i = 0; i < placeholder_streams; ++i) {
        hipStream_t st = nullptr;
        HIP_TRY(c0, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        c0->placeholder_streams.push_back(st);
    }
    for (int i = 0; i < nctx; ++i)
        if ((r = demon_acquire_streams(ctxs[i]))) return r == DEMON_ERR_HIP ? fail(c0, r, "lane " + std::to_string(i) + ": " + ctxs[i]->err) : r;
    return DEMON_OK;
}


int demon_hw_queues_hint(int under_launcher) { return under_launcher ? 16 : 8; }

// ---- CU masks: a context's streams restricted to a subset of the compute units (hipExtStreamCreateWithCUMask).  Bit i of the mask
// is CU slot i / 8 of XCD i % 8 (the driver deals mask bits round robin to the XCDs); a mask must leave every XCD some CUs.  The
// streams are exchanged at once (nothing may be in flight); nwords = 0 removes the mask.  Cached graphs are dropped.
int demon_set_cu_mask(demon_ctx *c, const uint32_t *mask, int nwords)
{
    if (!c || nwords < 0 || nwords > 16 || (nwords && !mask)) return DEMON_ERR_INVALID;
    if (nwords) {
        // every XCD keeps at least one CU: a queue without CUs on an XCD that is still dealt workgroups would never finish
        unsigned per_xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 32 * nwords; ++i) per_xcd[i & 7] += (mask[i >> 5] >> (i & 31)) & 1u;
        for (int x = 0; x < 8; ++x)
            if (!per_xcd[x]) return fail(c, DEMON_ERR_INVALID, "CU mask leaves XCD " + std::to_string(x) + " without compute units");
    }
    const bool had = c->stream != nullptr;
    int r = demon_release_streams(c);
    if (r) return r;
    c->cu_mask.assign(mask, mask + nwords);
    return had ? demon_acquire_streams(c) : DEMON_OK;
}

// ---- one hipGraph for a GROUP of lanes: the passes of ctxs[0 .. nctx) as parallel branches of a single graph, launched on lane 0's
// stream (one hipGraphLaunch = nctx steps; the lanes join at the end of the graph).  The alternative to feeding the lanes' own graphs
// round robin (demon_run_full on each): no stream -> hardware-queue calibration of ours, the runtime places the branches itself.
int demon_lanes_run_group(demon_ctx *const *ctxs, int nctx, int n, int iterations, int bootstrap_only)
{
    int r = lanes_check(ctxs, nctx);
    if (r) return r;
    demon_ctx *c0 = ctxs[0];
    if (iterations < 0 || iterations > 64) return fail(c0, DEMON_ERR_INVALID, "iterations out of range");
    for (int i = 0; i < nctx; ++i)
        if ((r = check_batch(ctxs[i], n))) return i ? fail(c0, r, "lane " + std::to_string(i) + ": " + ctxs[i]->err) : r;
    hipSetDevice(c0->device);
    std::string key = std::to_string(bootstrap_only ? 1 : 0) + ":" + std::to_string(n) + ":" + std::to_string(iterations);
    for (int i = 0; i < nctx; ++i) key += ":" + std::to_string(ctxs[i]->serial) + "." + std::to_string(ctxs[i]->opt_side_branches);
    auto it = c0->group_graphs.find(key);
    if (it == c0->group_graphs.end()) {
        while ((int)c0->group_events.size() < nctx) {
            hipEvent_t e = nullptr;
            HIP_TRY(c0, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            c0->group_events.push_back(e);
        }
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        for (int i = 0; i < nctx; ++i) ctxs[i]->err.clear();
        HIP_TRY(c0, hipStreamBeginCapture(c0->stream, hipStreamCaptureModeThreadLocal));
        bool ok = hipEventRecord(c0->group_events[0], c0->stream) == hipSuccess;
        for (int i = 1; ok && i < nctx; ++i) ok = hipStreamWaitEvent(ctxs[i]->stream, c0->group_events[0], 0) == hipSuccess;   // the lanes' streams join the capture
        for (int i = 0; ok && i < nctx; ++i) enqueue_sequence(ctxs[i], bootstrap_only ? SEQ_BOOT : SEQ_FULL, n, bootstrap_only ? 0 : iterations, ctxs[i]->stream);
        for (int i = 1; ok && i < nctx; ++i)
            ok = hipEventRecord(c0->group_events[i], ctxs[i]->stream) == hipSuccess && hipStreamWaitEvent(c0->stream, c0->group_events[i], 0) == hipSuccess;
        const hipError_t ec = hipStreamEndCapture(c0->stream, &graph);
        for (int i = 0; i < nctx; ++i)
            if (!ctxs[i]->err.empty()) { if (i) c0->err = ctxs[i]->err; ok = false; }
        if (!ok || ec != hipSuccess) {
            if (graph) hipGraphDestroy(graph);
            return fail(c0, DEMON_ERR_HIP, c0->err.empty() ? std::string("capture of the lane group failed: ") + hipGetErrorString(ec) : c0->err);
        }
        HIP_TRY(c0, hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        hipGraphDestroy(graph);
        it = c0->group_graphs.emplace(key, exec).first;
    }
    HIP_TRY(c0, hipGraphLaunch(it->second, c0->stream));
    return DEMON_OK;
}

int demon_lanes_calibrate(demon_ctx *const *ctxs, int nctx, int n, int iterations, int bootstrap_only, int steps_per_lane,
                          int max_placeholders, unsigned lanes_mask, demon_lanes_result *res)
{
    int r = lanes_check(ctxs, nctx);
    if (r) return r;
    demon_ctx *c0 = ctxs[0];
    if (!res || steps_per_lane < 1 || max_placeholders < 0 || max_placeholders > DEMON_LANES_MAX_PLACEHOLDERS) return fail(c0, DEMON_ERR_INVALID, "bad argument");
    if (iterations < 0 || iterations > 64) return fail(c0, DEMON_ERR_INVALID, "iterations out of range");
    for (int i = 0; i < nctx; ++i)
        if ((r = check_batch(ctxs[i], n))) return i ? fail(c0, r, "lane " + std::to_string(i) + ": " + ctxs[i]->err) : r;
    hipSetDevice(c0->device);
    memset(res, 0, sizeof *res);
    auto sync = [&](int k) -> int {
        for (int i = 0; i < k; ++i) HIP_TRY(c0, hipStreamSynchronize(ctxs[i]->stream));
        return DEMON_OK;
    };
    auto rate = [&](int k, float *out) -> int {
        double best = 0.0;
        for (int round = 0; round < 2; ++round) {   // (the first round also instantiates graphs / warms caches)
            int rr = sync(nctx);
            if (rr) return rr;
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < steps_per_lane * k; ++i) {
                demon_ctx *c = ctxs[i % k];
                if ((rr = run_sequence(c, bootstrap_only ? SEQ_BOOT : SEQ_FULL, n, bootstrap_only ? 0 : iterations)))
                    return rr == DEMON_ERR_HIP && c != c0 ? fail(c0, rr, c->err) : rr;
            }
            if ((rr = sync(k))) return rr;
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            best = std::max(best, (double)n * steps_per_lane * k / s);
        }
        *out = (float)best;
        return DEMON_OK;
    };
    int current = (int)c0->placeholder_streams.size();
    if (nctx == 1) max_placeholders = 0;     // one lane does not care where its stream lands
    res->lanes = 1;
    res->placeholder_streams = 0;
    res->pairs_per_s = -1.0f;
    // Round 6: the sweep STOPS on a plateau -- when the best cell of the placeholder count just measured is within 1 % of the best of all
    // earlier counts, that cell is kept and the lanes stay where they are: a mapping that was measured and never left.  Every further
    // demon_lanes_apply re-creates every stream of the group, and a winner that has to be re-applied later may not come back (under
    // torch.distributed.run, 16 hardware queues: cells 4@0 .. 4@3 measured 4 822 - 4 845 pairs/s, 4@4 / 4@5 4 371 / 4 310, and pad 0 applied
    // again afterwards ran 4 357 in every one of 17 attempts, profiles/r06_forcedist_rccl_1rank.json of the first collection).  In every
    // table on record the plateau is reached at one placeholder.
    float best_before = -1.0f;
    for (int pad = 0; pad <= max_placeholders; ++pad) {
        if (pad != current || pad == 0) {   // (pad 0 is applied too: the measurement must not depend on a previous call's placeholders)
            if ((r = demon_lanes_apply(ctxs, nctx, pad))) return r;
            current = pad;
        }
        float pad_best = -1.0f;
        int pad_best_k = 1;
        for (int k = 1; k <= nctx; ++k) {
            if (pad && k == 1) continue;
            float v = 0.0f;
            if ((r = rate(k, &v))) return r;
            if (res->ntable < DEMON_LANES_TABLE_CAP) {
                res->table[res->ntable].lanes = k;
                res->table[res->ntable].placeholder_streams = pad;
                res->table[res->ntable].pairs_per_s = v;
                ++res->ntable;
            }
            const bool allowed = !lanes_mask || ((lanes_mask >> k) & 1u);   // (the whole table is measured; the winner comes from the allowed lane counts)
            if (allowed && v > res->pairs_per_s) { res->pairs_per_s = v; res->lanes = k; res->placeholder_streams = pad; }
            if (allowed && v > pad_best) { pad_best = v; pad_best_k = k; }
        }
        if (pad >= 1 && nctx > 1 && pad_best_k > 1 && pad_best >= 0.99f * best_before && !getenv("DEMON_LANES_FULL_SWEEP")) {
            res->pairs_per_s = pad_best; res->lanes = pad_best_k; res->placeholder_streams = pad;
            break;
        }
        best_before = std::max(best_before, pad_best);
    }
    if (res->pairs_per_s < 0.0f) return fail(c0, DEMON_ERR_INVALID, "lanes_mask allows no lane count in [1, nctx]");
    // Back to the winner -- and MEASURE it again there.  Which hardware queue a new stream gets also depends on how many streams the
    // process created before (round 5, under torch.distributed.run: the cell that measured 4 718 pairs/s ran 4 119 after it was simply
    // applied again, gpurun_out/r5h_verify.txt), so the placeholder count alone does not reproduce a mapping.  Apply, measure, and when
    // the rate is not the winner's, shift the creation count by one throw-away stream and try again; the state that is left is one
    // that was measured.
    res->verified_pairs_per_s = res->pairs_per_s;
    if (nctx > 1) {
        bool reproduced = false;
        for (res->attempts = 1; res->attempts <= DEMON_LANES_MAX_ATTEMPTS; ++res->attempts) {
            if (res->attempts > 1 || res->placeholder_streams != current) {
                int pad = res->placeholder_streams;
                if (res->attempts > 1) {
                    hipStream_t burn = nullptr;
                    if (hipStreamCreateWithFlags(&burn, hipStreamNonBlocking) == hipSuccess) hipStreamDestroy(burn);
                    // (round 6) from the fourth attempt on the other placeholder counts take turns too: what is looked for is ANY mapping that
                    // runs the winner's lane count at the winner's rate, and seventeen attempts on one count have been seen to miss it
                    if (res->attempts > 3 && max_placeholders > 0) pad = (res->placeholder_streams + res->attempts - 3) % (max_placeholders + 1);
                }
                if ((r = demon_lanes_apply(ctxs, nctx, pad))) return r;
                current = pad;
            }
            float v = 0.0f;
            if ((r = rate(res->lanes, &v))) return r;
            res->verified_pairs_per_s = v;
            if (v >= 0.975f * res->pairs_per_s) { reproduced = true; res->placeholder_streams = current; break; }
        }
        // every attempt missed the bar: the lanes stay on the LAST mapping tried, `verified_pairs_per_s` is its measured rate, and
        // attempts = DEMON_LANES_MAX_ATTEMPTS + 1 says that the sweep's winner was never reproduced (callers must not cache it)
        if (!reproduced) res->attempts = DEMON_LANES_MAX_ATTEMPTS + 1;
    } else {
        res->attempts = 1;
    }
    return DEMON_OK;
}

int demon_download_outputs(demon_ctx *c, int n, const demon_outputs *o, float *depth0)
{
    if (!c || n < 1 || n > c->max_batch) return fail(c, DEMON_ERR_INVALID, "bad batch");
    hipSetDevice(c->device);
    int r = download_outputs(c, n, o);
    if (r) return r;
    if ((r = d2h(c, depth0, c->depth0, n))) return r;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DEMON_OK;
}

int demon_download_normal0(demon_ctx *c, int n, float *normal0)
{
    if (!c || n < 1 || n > c->max_batch) return fail(c, DEMON_ERR_INVALID, "bad batch");
    if (c->variant != 2) return fail(c, DEMON_ERR_INVALID, "predict_normal0 exists only in v2 contexts (demon_create_v2)");
    hipSetDevice(c->device);
    int r = d2h(c, normal0, c->normal0, n);
    if (r) return r;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DEMON_OK;
}

int demon_bootstrap(demon_ctx *c, int n, const float *image_pair, const float *image2_2, const demon_outputs *o)
{
    int r = demon_upload_inputs(c, n, image_pair, image2_2);
    if (r) return r;
    if ((r = run_sequence(c, SEQ_BOOT, n, 0))) return r;
    return demon_download_outputs(c, n, o, nullptr);
}

int demon_iterative(demon_ctx *c, int n, const float *image_pair, const float *image2_2, const float *depth2,
                    const float *normal2, const float *rotation, const float *translation, const demon_outputs *o)
{
    int r = demon_upload_inputs(c, n, image_pair, image2_2);
    if (r) return r;
    if (!rotation || !translation) return fail(c, DEMON_ERR_INVALID, "null input pointer");
    if ((r = h2d(c, c->depth2, depth2, n))) return r;
    if ((r = h2d(c, c->normal2, normal2, n))) return r;
    HIP_TRY(c, hipMemcpyAsync(c->d_rot, rotation, sizeof(float) * 3 * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_trans, translation, sizeof(float) * 3 * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if ((r = run_sequence(c, SEQ_ITER, n, 0))) return r;
    return demon_download_outputs(c, n, o, nullptr);
}

int demon_refine(demon_ctx *c, int n, const float *image1, const float *depth2, float *depth0)
{
    int r = check_batch(c, n);
    if (r) return r;
    hipSetDevice(c->device);
    if ((r = h2d(c, c->image_pair.slice(0, 3), image1, n))) return r;
    if ((r = h2d(c, c->depth2, depth2, n))) return r;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if ((r = run_sequence(c, SEQ_REFINE, n, 0))) return r;
    return demon_download_outputs(c, n, nullptr, depth0);
}

int demon_full(demon_ctx *c, int n, const float *image_pair, const float *image2_2, int iterations,
               const demon_outputs *o, float *depth0)
{
    int r = demon_upload_inputs(c, n, image_pair, image2_2);
    if (r) return r;
    if (iterations < 0 || iterations > 64) return fail(c, DEMON_ERR_INVALID, "iterations out of range");
    if ((r = run_sequence(c, SEQ_FULL, n, iterations))) return r;
    return demon_download_outputs(c, n, o, depth0);
}

int demon_time_full(demon_ctx *c, int n, int iterations, int steps, float *total_ms)
{
    int r = check_batch(c, n);
    if (r) return r;
    if (steps < 1 || !total_ms) return fail(c, DEMON_ERR_INVALID, "bad steps");
    hipSetDevice(c->device);
    hipEvent_t e0, e1;
    HIP_TRY(c, hipEventCreate(&e0));
    HIP_TRY(c, hipEventCreate(&e1));
    HIP_TRY(c, hipEventRecord(e0, c->stream));
    for (int i = 0; i < steps; ++i)
        if ((r = run_sequence(c, SEQ_FULL, n, iterations))) return r;
    HIP_TRY(c, hipEventRecord(e1, c->stream));
    HIP_TRY(c, hipEventSynchronize(e1));
    HIP_TRY(c, hipEventElapsedTime(total_ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return DEMON_OK;
}

int demon_profile_full(demon_ctx *c, int n, int iterations, int repeats, demon_launch_record *rec, int cap, int *count)
{
    int r = check_batch(c, n);
    if (r) return r;
    if (repeats < 1 || !rec || !count) return fail(c, DEMON_ERR_INVALID, "bad arguments");
    hipSetDevice(c->device);
    // exactly the steps a default forward pass launches (run_steps, mode 0): of the k x 1 / 1 x k pairs either the fused step or
    // the two layer steps, never both; the image-feature cache steps only exist with reuse_image_features
    std::vector<const Step *> seq;
    auto active = [c, n](const Step &s) {
        return s.image_only < 2 && !pair_step_skipped(c, s, n) &&
               !((s.fused == 1 && !c->opt_fused_inputs) || (s.fused == 2 && c->opt_fused_inputs));
    };
    for (auto &s : c->net_boot) if (active(s)) seq.push_back(&s);
    for (int i = 0; i < iterations; ++i)
        for (auto &s : c->net_iter) if (active(s)) seq.push_back(&s);
    for (auto &s : c->net_refine) if (active(s)) seq.push_back(&s);
    std::vector<std::string> tags(seq.size());
    std::vector<hipEvent_t> ev(3 * seq.size());  // per step: start, in front of the split-K reduce launch (if any), end
    for (auto &e : ev) HIP_TRY(c, hipEventCreate(&e));
    std::vector<double> ms(seq.size(), 0.0), red_ms(seq.size(), 0.0);
    std::vector<char> marked(seq.size(), 0);
    if (c->opt_tune_lanes > 1) {
        // throughput mode (option tune_lanes = L): what a launch costs while others run beside it -- L concurrent replays of a graph
        // of 5 launches of the step, on L streams; a launch is charged 1 / (5 L) of the time until all are done
        for (auto &e : ev) hipEventDestroy(e);
        for (size_t i = 0; i < seq.size(); ++i) {
            bool failed = false;
            g_last_kernel = nullptr;
            double t = 0;
            for (int rep = 0; rep < repeats && !failed; ++rep) t += time_replay(c, [&] { seq[i]->fn(n, c->stream); }, failed);
            if (failed) return fail(c, DEMON_ERR_HIP, "throughput-mode profile failed at step " + seq[i]->name);
            tags[i] = g_last_kernel ? g_last_kernel : "";
            ms[i] = t / (5.0 * c->opt_tune_lanes);
        }
        *count = (int)seq.size();
        for (size_t i = 0; i < seq.size() && (int)i < cap; ++i) {
            memset(&rec[i], 0, sizeof rec[i]);
            strncpy(rec[i].name, seq[i]->name.c_str(), sizeof rec[i].name - 1);
            strncpy(rec[i].kernel, !tags[i].empty() ? tags[i].c_str() : seq[i]->kernel.c_str(), sizeof rec[i].kernel - 1);
            rec[i].flops = seq[i]->flops_per_sample * n;
            rec[i].bytes = seq[i]->bytes_per_sample * n + seq[i]->bytes_fixed;
            rec[i].ms = (float)(ms[i] / repeats);
        }
        return DEMON_OK;
    }
    for (int rep = 0; rep < repeats + 1; ++rep) {  // first pass = warm-up
        for (size_t i = 0; i < seq.size(); ++i) {
            HIP_TRY(c, hipEventRecord(ev[3 * i], c->stream));
            g_last_kernel = nullptr;
            g_reduce_mark = ev[3 * i + 1];
            g_reduce_marked = false;
            seq[i]->fn(n, c->stream);
            g_reduce_mark = nullptr;
            marked[i] = g_reduce_marked ? 1 : 0;
            tags[i] = g_last_kernel ? g_last_kernel : "";
            HIP_TRY(c, hipEventRecord(ev[3 * i + 2], c->stream));
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (rep == 0) continue;
        for (size_t i = 0; i < seq.size(); ++i) {
            float t = 0;
            HIP_TRY(c, hipEventElapsedTime(&t, ev[3 * i], ev[3 * i + 2]));
            ms[i] += t;
            if (marked[i] && hipEventElapsedTime(&t, ev[3 * i + 1], ev[3 * i + 2]) == hipSuccess) red_ms[i] += t;
        }
    }
    for (auto &e : ev) hipEventDestroy(e);
    *count = (int)seq.size();
    for (size_t i = 0; i < seq.size() && (int)i < cap; ++i) {
        memset(&rec[i], 0, sizeof rec[i]);
        strncpy(rec[i].name, seq[i]->name.c_str(), sizeof rec[i].name - 1);
        strncpy(rec[i].kernel, !tags[i].empty() ? tags[i].c_str() : seq[i]->kernel.c_str(), sizeof rec[i].kernel - 1);
        rec[i].flops = seq[i]->flops_per_sample * n;
        rec[i].bytes = seq[i]->bytes_per_sample * n + seq[i]->bytes_fixed;
        rec[i].ms = (float)(ms[i] / repeats);
        rec[i].reduce_ms = (float)(red_ms[i] / repeats);
    }
    return DEMON_OK;
}

// ---- op-level entry points -----------------------------------------------------------------------------
#define OP_PROLOGUE(c)                                   \
    if (!(c)) return DEMON_ERR_INVALID;                  \
    if (!(c)->stream) return fail(c, DEMON_ERR_NOT_READY, kNoStream); \
    hipSetDevice((c)->device);                           \
    TmpDev tmp;

#define OP_FINISH(c, dptr, hptr, nfloats)                                                                         \
    HIP_TRY(c, hipGetLastError());                                                                                \
    HIP_TRY(c, hipStreamSynchronize((c)->stream));                                                                \
    HIP_TRY(c, hipMemcpy(hptr, dptr, sizeof(float) * (size_t)(nfloats), hipMemcpyDeviceToHost));                  \
    return DEMON_OK;

int demon_op_depth_to_flow(demon_ctx *c, float *out, const float *depth, const float *intrinsics, const float *rotation,
                           const float *translation, int n, int h, int w, int inverse_depth, int normalize_flow, int gate)
{
    OP_PROLOGUE(c);
    if (!out || !depth || !intrinsics || !rotation || !translation || n < 1 || h < 1 || w < 1)
        return fail(c, DEMON_ERR_INVALID, "bad argument");
    const size_t hw = (size_t)h * w;
    float *d_depth = tmp.upload(depth, n * hw), *d_k = tmp.upload(intrinsics, 4 * n), *d_r = tmp.upload(rotation, 3 * n),
          *d_t = tmp.upload(translation, 3 * n), *d_out = tmp.alloc(2 * n * hw);
    if (!d_depth || !d_k || !d_r || !d_t || !d_out) return fail(c, DEMON_ERR_HIP, "temporary device allocation failed");
    launch_depth_to_flow(d_out, d_depth, hw, d_k, d_r, d_t, n, h, w, 2 * hw, inverse_depth, normalize_flow, gate, c->stream);
    OP_FINISH(c, d_out, out, 2 * n * hw);
}

int demon_op_flow_to_depth(demon_ctx *c, float *out, const float *flow, const float *intrinsics, const float *rotation,
                           const float *translation, int n, int h, int w, int inverse_depth, int normalized_flow, int method)
{
    OP_PROLOGUE(c);
    if (!out || !flow || !intrinsics || !rotation || !translation || n < 1 || h < 1 || w < 1 || method < 0 || method > 1)
        return fail(c, DEMON_ERR_INVALID, "bad argument");
    const size_t hw = (size_t)h * w;
    float *d_flow = tmp.upload(flow, 2 * n * hw), *d_k = tmp.upload(intrinsics, 4 * n), *d_r = tmp.upload(rotation, 3 * n),
          *d_t = tmp.upload(translation, 3 * n), *d_out = tmp.alloc(n * hw);
    if (!d_flow || !d_k || !d_r || !d_t || !d_out) return fail(c, DEMON_ERR_HIP, "temporary device allocation failed");
    launch_flow_to_depth(d_out, hw, d_flow, 2 * hw, d_k, d_r, d_t, n, h, w, inverse_depth, normalized_flow, method, 0.0f, c->stream);
    OP_FINISH(c, d_out, out, n * hw);
}

int demon_op_warp2d(demon_ctx *c, float *out, const float *input, const float *disp, int n, int ch, int h, int w,
                    int normalized, int border_mode, float border_value)
{
    OP_PROLOGUE(c);
    if (!out || !input || !disp || n < 1 || ch < 1 || h < 1 || w < 1) return fail(c, DEMON_ERR_INVALID, "bad argument");
    const size_t hw = (size_t)h * w;
    float *d_in = tmp.upload(input, n * ch * hw), *d_disp = tmp.upload(disp, 2 * n * hw), *d_out = tmp.alloc(n * ch * hw);
    if (!d_in || !d_disp || !d_out) return fail(c, DEMON_ERR_HIP, "temporary device allocation failed");
    launch_warp2d(d_out, ch * hw, d_in, ch * hw, d_disp, 2 * hw, n, ch, h, w, normalized, border_mode, border_value, c->stream);
    OP_FINISH(c, d_out, out, n * ch * hw);
}

int demon_op_leaky_relu(demon_ctx *c, float *out, const float *in, int64_t count, float leak)
{
    OP_PROLOGUE(c);
    if (!out || !in || count < 0) return fail(c, DEMON_ERR_INVALID, "bad argument");
    if (count == 0) return DEMON_OK;
    float *d_in = tmp.upload(in, count), *d_out = tmp.alloc(count);
    if (!d_in || !d_out) return fail(c, DEMON_ERR_HIP, "temporary device allocation failed");
    launch_leaky_relu(d_out, d_in, count, leak, c->stream);
    OP_FINISH(c, d_out, out, count);
}

int demon_op_replace_nonfinite(demon_ctx *c, float *out, const float *in, int64_t count, float value)
{
    OP_PROLOGUE(c);
    if (!out || !in || count < 0) return fail(c, DEMON_ERR_INVALID, "bad argument");
    if (count == 0) return DEMON_OK;
    float *d_in = tmp.upload(in, count), *d_out = tmp.alloc(count);
    if (!d_in || !d_out) return fail(c, DEMON_ERR_HIP, "temporary device allocation failed");
    launch_replace_nonfinite(d_out, d_in, count, value, c->stream);
    OP_FINISH(c, d_out, out, count);
}

int demon_op_scale_invariant_gradient(demon_ctx *c, float *out, const float *in, int nc, int h, int w, const int *deltas,
                                      const float *weights, int ndeltas, float epsilon)
{
    OP_PROLOGUE(c);
    if (!out || !in || !deltas || !weights || nc < 1 || h < 1 || w < 1 || ndeltas < 1 || ndeltas > 8)
        return fail(c, DEMON_ERR_INVALID, "bad argument (1..8 deltas)");
    const size_t hw = (size_t)h * w;
    float *d_in = tmp.upload(in, nc * hw), *d_out = tmp.alloc(2 * nc * hw);
    if (!d_in || !d_out) return fail(c, DEMON_ERR_HIP, "temporary device allocation failed");
    launch_sig(d_out, d_in, nc, h, w, deltas, weights, ndeltas, epsilon, c->stream);
    OP_FINISH(c, d_out, out, 2 * nc * hw);
}

int demon_op_depth_to_normals(demon_ctx *c, float *out, const float *depth, const float *intrinsics, int n, int h, int w,
                              int inverse_depth)
{
    OP_PROLOGUE(c);
    if (!out || !depth || !intrinsics || n < 1 || h < 1 || w < 1) return fail(c, DEMON_ERR_INVALID, "bad argument");
    const size_t hw = (size_t)h * w;
    float *d_depth = tmp.upload(depth, n * hw), *d_k = tmp.upload(intrinsics, 4 * n), *d_out = tmp.alloc(3 * n * hw);
    if (!d_depth || !d_k || !d_out) return fail(c, DEMON_ERR_HIP, "temporary device allocation failed");
    launch_depth_to_normals(d_out, d_depth, d_k, n, h, w, inverse_depth, c->stream);
    OP_FINISH(c, d_out, out, 3 * n * hw);
}

int demon_op_median3x3_downsample(demon_ctx *c, float *out, const float *in, int nc, int h, int w)
{
    OP_PROLOGUE(c);
    if (!out || !in || nc < 1 || h < 1 || w < 1) return fail(c, DEMON_ERR_INVALID, "bad argument");
    const size_t ho = (h + 1) / 2, wo = (w + 1) / 2;
    float *d_in = tmp.upload(in, (size_t)nc * h * w), *d_out = tmp.alloc(nc * ho * wo);
    if (!d_in || !d_out) return fail(c, DEMON_ERR_HIP, "temporary device allocation failed");
    launch_median3x3_downsample(d_out, d_in, nc, h, w, c->stream);
    OP_FINISH(c, d_out, out, nc * ho * wo);
}

int demon_op_pointwise_l2_loss(demon_ctx *c, float *loss, const float *inp, const float *gt, int n, int ch, int h, int w, float epsilon)
{
    OP_PROLOGUE(c);
    if (!loss || !inp || !gt || n < 1 || ch < 1 || h < 1 || w < 1) return fail(c, DEMON_ERR_INVALID, "bad argument");
    const size_t count = (size_t)n * ch * h * w, blocks = ((size_t)n * h * w + 255) / 256;
    float *d_inp = tmp.upload(inp, count), *d_gt = tmp.upload(gt, count), *d_part = tmp.alloc(blocks);
    if (!d_inp || !d_gt || !d_part) return fail(c, DEMON_ERR_HIP, "temporary device allocation failed");
    launch_pointwise_l2_partial(d_part, d_inp, d_gt, n, ch, h * w, epsilon, c->stream);
    std::vector<float> part(blocks);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(part.data(), d_part, sizeof(float) * blocks, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double sum = 0.0;
    for (float v : part) sum += v;
    *loss = (float)(sum / ((double)n * h * w));
    return DEMON_OK;
}

// one stand-alone contraction layer through the same packing + kernel path the networks use
static int run_single_layer(demon_ctx *c, Layer::Kind kind, float *out, const float *in, const float *w, const float *bias,
                            int n, int cin, int h, int wd, int cout, int kh, int kw, int sh, int sw, int lrelu, bool same = false)
{
    if (!c) return DEMON_ERR_INVALID;
    hipSetDevice(c->device);
    if (!out || !in || !w || !bias || n < 1 || cin < 1 || cout < 1 || h < 1 || wd < 1 || kh < 1 || kw < 1 || sh < 1 || sw < 1)
        return fail(c, DEMON_ERR_INVALID, "bad argument");
    if (!c->stream) return fail(c, DEMON_ERR_NOT_READY, kNoStream);   // (never the null stream: include/demon_hip.h, demon_release_streams)
    demon_ctx scratch;  // owns the temporary device allocations of this call
    scratch.device = c->device;
    scratch.max_batch = n;
    scratch.guard_bytes = guard_bytes_from_env();   // poison harness: input, output, every weight form between NaN canaries
    Layer L;
    L.kind = kind; L.Cin = cin; L.Cout = cout; L.kh = kh; L.kw = kw; L.sh = sh; L.sw = sw; L.ph = kh / 2; L.pw = kw / 2;
    L.act = lrelu;
    int ho, wo;
    if (kind == Layer::DECONV) { ho = 2 * h; wo = 2 * wd; }
    else if (kind == Layer::DENSE) { ho = 1; wo = 1; }
    else if (same) {
        // tf.layers.conv2d(padding='same') (v2/helpers.py:24-35): out = ceil(n / s), zeros in front = pad_total // 2
        ho = (h + sh - 1) / sh; wo = (wd + sw - 1) / sw;
        const int th = (ho - 1) * sh + kh - h, tw = (wo - 1) * sw + kw - wd;
        L.ph = th > 0 ? th / 2 : 0; L.pw = tw > 0 ? tw / 2 : 0;
    } else { ho = (h + 2 * L.ph - kh) / sh + 1; wo = (wd + 2 * L.pw - kw) / sw + 1; }
    L.in = buffer(&scratch, "in", cin, h, wd);
    L.out = buffer(&scratch, "out", cout, ho, wo);
    int rc = DEMON_OK;
    if (!L.in.base || !L.out.base || !plan_layer(&scratch, &L))   // (buffer() says why when it refused a size: 2^31-element limit)
        rc = fail(c, DEMON_ERR_HIP, scratch.err.empty() ? std::string("temporary device allocation failed") : scratch.err);
    if (!rc && upload_kernel(&scratch, &L, w)) rc = fail(c, DEMON_ERR_HIP, scratch.err);
    if (!rc && hipMemcpy(L.d_bias, bias, sizeof(float) * cout, hipMemcpyHostToDevice) != hipSuccess) rc = fail(c, DEMON_ERR_HIP, "bias upload failed");
    if (!rc && hipMemcpy(L.in.base, in, sizeof(float) * (size_t)n * cin * h * wd, hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(c, DEMON_ERR_HIP, "input upload failed");
    if (!rc) {
        run_layer(&L, n, c->stream, c->d_ws);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess ||
            hipMemcpy(out, L.out.base, sizeof(float) * (size_t)n * cout * ho * wo, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(c, DEMON_ERR_HIP, "layer execution failed");
    }
    if (!rc && scratch.guard_bytes) {
        std::string where;
        if (check_guards(&scratch, &where)) rc = fail(c, DEMON_ERR_HIP, std::string("poison harness: a kernel wrote outside its tensors (") + (g_last_kernel ? g_last_kernel : "?") + "): " + where);
    }
    for (void *p : scratch.allocations) hipFree(p);
    return rc;
}

int demon_debug_check_guards(demon_ctx *c, int *violations)
{
    if (!c || !violations) return DEMON_ERR_INVALID;
    hipSetDevice(c->device);
    if (!c->guard_bytes) return fail(c, DEMON_ERR_INVALID, "the context was not created under DEMON_POISON_GUARD=1");
    if (c->stream) HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::string where;
    *violations = check_guards(c, &where);
    if (*violations) c->err = "poison harness: " + where;
    return DEMON_OK;
}

int demon_last_kernel(char *tag, int tag_cap)
{
    const char *k = g_last_kernel ? g_last_kernel : "";
    if (tag && tag_cap > 0) { strncpy(tag, k, tag_cap - 1); tag[tag_cap - 1] = 0; }
    return (int)strlen(k);
}

// Times one contraction layer on device-resident random data (tuning / roofline diagnostics).
// kind: 0 conv (kh x kw, stride sh x sw, pad k/2), 1 transposed conv k4 s2, 2 dense.  tile < 0 / ksplit <= 0: automatic plan.
int demon_bench_layer(demon_ctx *c, int kind, int n, int cin, int h, int wd, int cout, int kh, int kw, int sh, int sw,
                      int tile, int ksplit, int iters, float *avg_ms, double *flops)
{
    if (!c || !avg_ms || iters < 1 || n < 1 || kind < 0 || kind > 2 || (tile >= TILE_COUNT && tile < 100) || (tile >= 100 + PTILE_COUNT && tile < 200) || (tile >= 200 + STREAM_VARIANTS && tile < 300) || (tile >= 300 + FRAG_VARIANTS && tile < 400) || (tile >= 400 + WINO1D_VARIANTS && tile != 500)) return fail(c, DEMON_ERR_INVALID, "bad argument");
    if (!c->stream) return fail(c, DEMON_ERR_NOT_READY, kNoStream);
    hipSetDevice(c->device);
    demon_ctx scratch;
    scratch.device = c->device;
    scratch.max_batch = n;
    Layer L;
    L.kind = kind == 0 ? Layer::CONV : (kind == 1 ? Layer::DECONV : Layer::DENSE);
    if (kind == 1) { kh = kw = 4; sh = sw = 2; }
    if (kind == 2) { kh = kw = sh = sw = 1; h = wd = 1; }
    L.Cin = cin; L.Cout = cout; L.kh = kh; L.kw = kw; L.sh = sh; L.sw = sw; L.ph = kh / 2; L.pw = kw / 2; L.act = 1;
    int ho, wo;
    if (kind == 1) { ho = 2 * h; wo = 2 * wd; }
    else if (kind == 2) { ho = wo = 1; }
    else { ho = (h + 2 * L.ph - kh) / sh + 1; wo = (wd + 2 * L.pw - kw) / sw + 1; }
    L.in = buffer(&scratch, "in", cin, h, wd);
    L.out = buffer(&scratch, "out", cout, ho, wo);
    int rc = DEMON_OK;
    if (!L.in.base || !L.out.base || !plan_layer(&scratch, &L))   // (buffer() says why when it refused a size: 2^31-element limit)
        rc = fail(c, DEMON_ERR_HIP, scratch.err.empty() ? std::string("temporary device allocation failed") : scratch.err);
    if (!rc) {
        // deterministic pseudo-random fill (full-range values: zero-filled operands would clock higher)
        const size_t nin = (size_t)n * cin * h * wd, nw = (size_t)L.ncls * L.Krows * L.Mpad;
        std::vector<float> hin(nin), hw(nw);
        unsigned st = 12345u;
        auto rnd = [&st]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
        for (auto &v : hin) v = rnd();
        for (auto &v : hw) v = rnd() * 0.05f;
        if (hipMemcpy(L.in.base, hin.data(), nin * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(L.d_wp, hw.data(), nw * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(c, DEMON_ERR_HIP, "upload failed");
    }
    if (!rc) {
        L.force_tile = tile;
        L.force_split = ksplit;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) run_layer(&L, n, c->stream, c->d_ws);
        hipEventRecord(e0, c->stream);
        for (int i = 0; i < iters; ++i) run_layer(&L, n, c->stream, c->d_ws);
        hipEventRecord(e1, c->stream);
        if (hipEventSynchronize(e1) != hipSuccess || hipGetLastError() != hipSuccess) rc = fail(c, DEMON_ERR_HIP, "layer execution failed");
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        *avg_ms = ms / iters;
        hipEventDestroy(e0);
        hipEventDestroy(e1);
        const double pix = kind == 1 ? 4.0 * h * wd : (double)ho * wo;
        const double kreal = kind == 1 ? 4.0 * cin : (double)L.K;
        if (flops) *flops = 2.0 * cout * kreal * pix * n;
    }
    for (void *p : scratch.allocations) hipFree(p);
    return rc;
}

// Diagnostic: per-workgroup wall-clock records of ONE launch of a network layer (only in builds with -DDEMON_TIMELINE).
int demon_debug_timeline(demon_ctx *c, const char *layer_name, int n, uint64_t *records, int cap, int *count, float *ms, char *kernel, int kernel_cap)
{
#ifndef DEMON_TIMELINE
    (void)layer_name; (void)n; (void)records; (void)cap; (void)count; (void)ms; (void)kernel; (void)kernel_cap;
    return fail(c, DEMON_ERR_INVALID, "libdemon_hip.so was built without -DDEMON_TIMELINE (see tools/timeline.py)");
#else
    int r = check_batch(c, n);
    if (r) return r;
    if (!layer_name || !records || !count || cap < 1) return fail(c, DEMON_ERR_INVALID, "bad argument");
    hipSetDevice(c->device);
    Layer *L = nullptr;
    for (auto &l : c->layers) if (l->name == layer_name) L = l.get();
    if (!L) return fail(c, DEMON_ERR_NOT_FOUND, std::string("unknown layer ") + layer_name);
    const size_t max_wg = 1u << 16;
    unsigned long long *buf = nullptr;
    HIP_TRY(c, hipMalloc((void **)&buf, max_wg * 64));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float *ws = c->d_ws;
    for (int i = 0; i < 3; ++i) run_layer(L, n, c->stream, ws);   // warm
    hipMemsetAsync(buf, 0, max_wg * 64, c->stream);
    hipStreamSynchronize(c->stream);
    g_timeline_dev = buf;
    hipEventRecord(e0, c->stream);
    run_layer(L, n, c->stream, ws);
    hipEventRecord(e1, c->stream);
    g_timeline_dev = nullptr;
    hipError_t err = hipStreamSynchronize(c->stream);
    if (kernel && kernel_cap > 0 && g_last_kernel) { strncpy(kernel, g_last_kernel, kernel_cap - 1); kernel[kernel_cap - 1] = 0; }
    float t = 0;
    hipEventElapsedTime(&t, e0, e1);
    if (ms) *ms = t;
    std::vector<unsigned long long> host(max_wg * 8);
    if (err == hipSuccess) err = hipMemcpy(host.data(), buf, max_wg * 64, hipMemcpyDeviceToHost);
    hipFree(buf);
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (err != hipSuccess) return fail(c, DEMON_ERR_HIP, "timeline run failed");
    int k = 0;
    for (size_t w = 0; w < max_wg && k < cap; ++w)
        if (host[8 * w]) { memcpy(records + 8 * (size_t)k, &host[8 * w], 64); ++k; }
    *count = k;
    return DEMON_OK;
#endif
}

int demon_op_conv2d(demon_ctx *c, float *out, const float *in, const float *w_hwio, const float *bias, int n, int cin, int h,
                    int w, int cout, int kh, int kw, int sh, int sw, int ph, int pw, int lrelu)
{
    const bool same = ph == -1 && pw == -1;
    if (c && !same && (ph != kh / 2 || pw != kw / 2))
        return fail(c, DEMON_ERR_INVALID, "padding must be k/2 (helpers.py:78-79) or -1,-1 for TF 'same' (v2/helpers.py:24-35)");
    return run_single_layer(c, Layer::CONV, out, in, w_hwio, bias, n, cin, h, w, cout, kh, kw, sh, sw, lrelu, same);
}

int demon_op_deconv4x4s2(demon_ctx *c, float *out, const float *in, const float *w_hwoi, const float *bias, int n, int cin,
                         int h, int w, int cout, int lrelu)
{
    return run_single_layer(c, Layer::DECONV, out, in, w_hwoi, bias, n, cin, h, w, cout, 4, 4, 2, 2, lrelu);
}

int demon_op_dense(demon_ctx *c, float *out, const float *in, const float *w_io, const float *bias, int n, int cin, int cout,
                   int lrelu)
{
    return run_single_layer(c, Layer::DENSE, out, in, w_io, bias, n, cin, 1, 1, cout, 1, 1, 1, 1, lrelu);
}

}  // extern "C"
