// Network-level engine of libawr_hip.so: the AWR backbones (ResNet18-deconv, stacked hourglass) as static execution plans
// built and replayed natively (include/awr_hip.h, "Network-level API").
//
//   awr_net   checkpoint layout of one backbone (model/resnet_deconv.py:19-136, model/hourglass.py:105-165): state_dict keys,
//             shapes and arena offsets, conv / BatchNorm layers bound to caller-owned parameter / gradient / buffer arenas,
//             packed GEMM copies of the weights.
//   awr_plan  one (batch, image size, mode) instance: every activation / gradient buffer allocated once, two flat lists of
//             pre-bound kernel launches (forward, backward).  The backward list is derived at build time by a small static
//             autograd: every forward node registers an emitter for its own gradient kernels; `gtarget` decides
//             write-vs-accumulate per tensor; identity skips alias the upstream gradient buffer.  Replaying a list allocates
//             nothing and never synchronises the host: a whole step can be captured in one hipGraph.
//
// Host code only (no kernels): it calls the operator-level entry points of this library (awr_conv_gemm, awr_conv_wgrad,
// awr_bn_*, awr_stem_*, ...) and the HIP runtime for memory, streams and events.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "awr_common.h"

extern "C" unsigned long long awr_dp_generation(const awr_dp* d);      // csrc/awr_dp.hip: generation of a communicator that has not been destroyed, else 0

namespace awrnet {

using awr::set_error;

constexpr float BN_EPS = 1e-5f;
constexpr float BN_MOMENTUM = 0.1f;
constexpr int N_ALIGN = 128;   // packed weight rows are padded to a multiple of the widest GEMM tile


static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

#define NET_CHECK(call)                 \
    do {                                \
        if (int e__ = (call)) return e__; \
    } while (0)

#define HIP_TRY(call)                                              \
    do {                                                           \
        hipError_t e__ = (call);                                   \
        if (e__ != hipSuccess) {                                   \
            set_error("%s: %s", #call, hipGetErrorString(e__));    \
            return AWR_ERR_HIP;                                    \
        }                                                          \
    } while (0)

// ------------------------------------------------------------------------------------------
// checkpoint layout: ordered state_dict entries
// ------------------------------------------------------------------------------------------
enum Kind { CONV_W = 0, DECONV_W = 1, CONV_B = 2, BN_W = 3, BN_B = 4, BN_MEAN = 5, BN_VAR = 6, COUNTER = 7 };
static inline bool is_param(int k) { return k <= BN_B; }

struct Entry {
    std::string key;
    int ndim;
    int64_t shape[4];
    int kind;
    int64_t off = 0;     // floats into the parameter arena (params) / buffer arena (running stats); index for counters
    int64_t numel = 1;
    bool unused = false;  // parameters that never receive a gradient (hourglass skip_layers with inp == out)
};

struct Layout {
    std::vector<Entry> e;
    int64_t n_params = 0, n_active = 0, n_buffers = 0;
    int n_counters = 0;
    std::map<std::string, int> index;

    void add(const std::string& key, std::vector<int64_t> shape, int kind) {
        Entry en;
        en.key = key;
        en.kind = kind;
        en.ndim = (int)shape.size();
        for (int i = 0; i < 4; ++i) en.shape[i] = i < en.ndim ? shape[i] : 1;
        for (int i = 0; i < en.ndim; ++i) en.numel *= shape[i];
        index[key] = (int)e.size();
        e.push_back(en);
    }
    void bn(const std::string& p, int c) {
        add(p + ".weight", {c}, BN_W);
        add(p + ".bias", {c}, BN_B);
        add(p + ".running_mean", {c}, BN_MEAN);
        add(p + ".running_var", {c}, BN_VAR);
        add(p + ".num_batches_tracked", {}, COUNTER);
    }
    const Entry& at(const std::string& key) const { return e[index.at(key)]; }
    bool has(const std::string& key) const { return index.count(key) != 0; }

    // arena offsets: live parameters first, never-trained ones at the tail (the optimiser / all-reduce cover [0, n_active)
    // only -- exactly torch's `p.grad is None` skip); every view 16-byte aligned
    void assign() {
        int64_t off = 0;
        for (int pass = 0; pass < 2; ++pass) {
            for (auto& en : e) {
                if (!is_param(en.kind) || en.unused != (pass == 1)) continue;
                en.off = off;
                off = round_up(off + en.numel, 4);
            }
            if (pass == 0) n_active = off;
        }
        n_params = off;
        int64_t boff = 0;
        int ci = 0;
        for (auto& en : e) {
            if (en.kind == BN_MEAN || en.kind == BN_VAR) {
                en.off = boff;
                boff += round_up(en.shape[0], 4);
            } else if (en.kind == COUNTER) {
                en.off = ci++;
            }
        }
        n_buffers = boff;
        n_counters = ci;
    }
};

static std::string fmt(const char* f, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof buf, f, ap);
    va_end(ap);
    return buf;
}

// get_deconv_net(depth, J, downsample).state_dict() (resnet_deconv.py:8-16, :31-53): BasicBlock [2,2,2,2] for depth 18, Bottleneck
// (expansion 4) [3,4,6,3] / [3,4,23,3] / [3,8,36,3] for 50 / 101 / 152.  Registration order inside a block: conv1, bn1, conv2, bn2,
// (conv3, bn3,) downsample (resnet_deconv.py:148-156, :182-195).
static const int* resnet_blocks(int depth) {
    static const int b18[4] = {2, 2, 2, 2}, b50[4] = {3, 4, 6, 3}, b101[4] = {3, 4, 23, 3}, b152[4] = {3, 8, 36, 3};
    return depth == 50 ? b50 : depth == 101 ? b101 : depth == 152 ? b152 : b18;
}

static void resnet_layout(Layout& L, int depth, int J, int downsample) {
    const bool bott = depth != 18;
    const int exp = bott ? 4 : 1;
    const int* nb = resnet_blocks(depth);
    L.add("pre.0.weight", {64, 1, 5, 5}, CONV_W);
    L.bn("pre.1", 64);
    int cin = 64;
    const int planes_[4] = {64, 128, 256, 512};
    for (int li = 1; li <= 4; ++li) {
        const int planes = planes_[li - 1];
        for (int bi = 0; bi < nb[li - 1]; ++bi) {
            const std::string p = fmt("layer%d.%d", li, bi);
            if (!bott) {
                L.add(p + ".conv1.weight", {planes, cin, 3, 3}, CONV_W);
                L.bn(p + ".bn1", planes);
                L.add(p + ".conv2.weight", {planes, planes, 3, 3}, CONV_W);
                L.bn(p + ".bn2", planes);
            } else {
                L.add(p + ".conv1.weight", {planes, cin, 1, 1}, CONV_W);
                L.bn(p + ".bn1", planes);
                L.add(p + ".conv2.weight", {planes, planes, 3, 3}, CONV_W);
                L.bn(p + ".bn2", planes);
                L.add(p + ".conv3.weight", {planes * exp, planes, 1, 1}, CONV_W);
                L.bn(p + ".bn3", planes * exp);
            }
            if (bi == 0 && (li > 1 || cin != planes * exp)) {      // _make_layer: stride != 1 or inplanes != planes * expansion
                L.add(p + ".downsample.0.weight", {planes * exp, cin, 1, 1}, CONV_W);
                L.bn(p + ".downsample.1", planes * exp);
            }
            cin = planes * exp;
        }
    }
    int lg = 0;
    while ((1 << lg) < downsample) ++lg;
    for (int i = 0; i < 4 - lg; ++i) {
        L.add(fmt("deconv_layers.%d.weight", 3 * i), {cin, 256, 4, 4}, DECONV_W);
        L.bn(fmt("deconv_layers.%d", 3 * i + 1), 256);
        cin = 256;
    }
    L.add("final1.weight", {3 * J, 256, 1, 1}, CONV_W);
    L.add("final1.bias", {3 * J}, CONV_B);
    L.add("final2.weight", {J, 256, 1, 1}, CONV_W);
    L.add("final2.bias", {J}, CONV_B);
}

static void hgconv_keys(Layout& L, const std::string& p, int cin, int cout, int k, bool bn) {
    L.add(p + ".conv.weight", {cout, cin, k, k}, CONV_W);
    L.add(p + ".conv.bias", {cout}, CONV_B);
    if (bn) L.bn(p + ".bn", cout);
}

// hourglass.py:28-42 -- registration order bn1, conv1, bn2, conv2, bn3, conv3, skip_layer (present even when unused)
static void residual_keys(Layout& L, const std::string& p, int cin, int cout) {
    const int h = cout / 2;
    L.bn(p + ".bn1", cin);
    hgconv_keys(L, p + ".conv1", cin, h, 1, false);
    L.bn(p + ".bn2", h);
    hgconv_keys(L, p + ".conv2", h, h, 3, false);
    L.bn(p + ".bn3", h);
    hgconv_keys(L, p + ".conv3", h, cout, 1, false);
    hgconv_keys(L, p + ".skip_layer", cin, cout, 1, false);
}

static void hourglass_keys(Layout& L, const std::string& p, int depth, int f) {
    residual_keys(L, p + ".up1", f, f);
    residual_keys(L, p + ".low1", f, f);
    if (depth > 1) hourglass_keys(L, p + ".low2", depth - 1, f);
    else residual_keys(L, p + ".low2", f, f);
    residual_keys(L, p + ".low3", f, f);
}

// PoseNet('hourglass_<nstack>', J).state_dict() (hourglass.py:105-142)
static void hourglass_layout(Layout& L, int nstack, int J, int f) {
    hgconv_keys(L, "pre.0", 1, 64, 5, true);
    residual_keys(L, "pre.1", 64, 128);
    residual_keys(L, "pre.3", 128, 256);
    residual_keys(L, "pre.4", 256, f);
    for (int i = 0; i < nstack; ++i) hourglass_keys(L, fmt("hgs.%d.0", i), 4, f);
    for (int i = 0; i < nstack; ++i) {
        residual_keys(L, fmt("features.%d.0", i), f, f);
        hgconv_keys(L, fmt("features.%d.1", i), f, f, 1, true);
    }
    for (int i = 0; i < nstack; ++i) {
        L.add(fmt("outs_1.%d.weight", i), {3 * J, f, 1, 1}, CONV_W);
        L.add(fmt("outs_1.%d.bias", i), {3 * J}, CONV_B);
    }
    for (int i = 0; i < nstack; ++i) {
        L.add(fmt("outs_2.%d.weight", i), {J, f, 1, 1}, CONV_W);
        L.add(fmt("outs_2.%d.bias", i), {J}, CONV_B);
    }
    for (int i = 0; i < nstack - 1; ++i) hgconv_keys(L, fmt("merge_features.%d.conv", i), f, f, 1, false);
    for (int i = 0; i < nstack - 1; ++i) hgconv_keys(L, fmt("merge_preds.%d.conv", i), 4 * J, f, 1, false);
    // Residual.skip_layer exists in every block but only runs when inp_dim != out_dim (hourglass.py:38-47)
    for (auto& en : L.e) {
        const size_t pos = en.key.find(".skip_layer.");
        if (pos == std::string::npos) continue;
        const Entry& w = L.e[L.index.at(en.key.substr(0, pos) + ".skip_layer.conv.weight")];
        if (w.shape[0] == w.shape[1]) en.unused = true;
    }
}

// ------------------------------------------------------------------------------------------
// conv geometry: the three GEMM problems of one nn.Conv2d / nn.ConvTranspose2d (k4 s2 p1)
// ------------------------------------------------------------------------------------------
struct Spec {
    bool deconv = false;
    int cin = 0, cout = 0, k = 1, stride = 1, pad = 0, cin_pad = 0, cout_pad = 0;
    int T() const { return k * k; }
    void out_hw(int h, int w, int& ho, int& wo) const {
        if (!deconv) {
            ho = (h + 2 * pad - k) / stride + 1;
            wo = (w + 2 * pad - k) / stride + 1;
        } else {
            ho = (h - 1) * stride - 2 * pad + k;
            wo = (w - 1) * stride - 2 * pad + k;
        }
    }
};

static Spec make_spec(bool deconv, int cin, int cout, int k, int stride, int pad, int cin_pad = 0, int cout_pad = 0) {
    Spec s;
    s.deconv = deconv;
    s.cin = cin; s.cout = cout; s.k = k; s.stride = stride; s.pad = pad;
    s.cin_pad = cin_pad ? cin_pad : cin;
    s.cout_pad = cout_pad ? cout_pad : cout;
    return s;
}

struct Tap { int dy, dx, wt; };
struct PhaseT { int py, px; std::vector<Tap> taps; };
struct Prob {
    int Hin, Win, Cin, Hout, Wout, N, Hq, Wq, so, si;
    std::vector<PhaseT> phases;
    bool full;
};

static std::vector<PhaseT> gather_taps(const Spec& s) {       // conv-type gather: one phase, taps (ky-p, kx-p, ky*k+kx)
    PhaseT ph{0, 0, {}};
    for (int ky = 0; ky < s.k; ++ky)
        for (int kx = 0; kx < s.k; ++kx) ph.taps.push_back({ky - s.pad, kx - s.pad, ky * s.k + kx});
    return {ph};
}
static std::vector<PhaseT> mirror_taps(const Spec& s) {       // stride-1 data gradient: dx[y] = sum_k dy[y + p - k] w[k]
    PhaseT ph{0, 0, {}};
    for (int ky = 0; ky < s.k; ++ky)
        for (int kx = 0; kx < s.k; ++kx) ph.taps.push_back({s.pad - ky, s.pad - kx, ky * s.k + kx});
    return {ph};
}
static int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
static int pymod(int a, int b) { return a - floordiv(a, b) * b; }
static std::vector<PhaseT> scatter_phases(const Spec& sp) {   // transposed-type (output stride 2): phase (py,px) takes the taps with (py+p-ky) even
    const int s = sp.stride, p = sp.pad, k = sp.k;
    std::vector<PhaseT> out;
    for (int py = 0; py < s; ++py)
        for (int px = 0; px < s; ++px) {
            PhaseT ph{py, px, {}};
            for (int ky = 0; ky < k; ++ky) {
                if (pymod(py + p - ky, s) != 0) continue;
                for (int kx = 0; kx < k; ++kx) {
                    if (pymod(px + p - kx, s) != 0) continue;
                    ph.taps.push_back({floordiv(py + p - ky, s), floordiv(px + p - kx, s), ky * k + kx});
                }
            }
            if (!ph.taps.empty()) out.push_back(ph);
        }
    return out;
}

static Prob fwd_problem(const Spec& s, int hin, int win) {
    int ho, wo;
    s.out_hw(hin, win, ho, wo);
    if (!s.deconv) return Prob{hin, win, s.cin_pad, ho, wo, s.cout_pad, ho, wo, 1, s.stride, gather_taps(s), true};
    auto ph = scatter_phases(s);
    const bool full = ph.size() == 4;
    return Prob{hin, win, s.cin_pad, ho, wo, s.cout_pad, ho / 2, wo / 2, 2, 1, ph, full};
}

// gradient w.r.t. the layer input: reads dY (B,Hout,Wout,cout_pad), writes dX (B,hin,win,cin_pad)
static Prob dgrad_problem(const Spec& s, int hin, int win) {
    int ho, wo;
    s.out_hw(hin, win, ho, wo);
    if (s.deconv) return Prob{ho, wo, s.cout_pad, hin, win, s.cin_pad, hin, win, 1, s.stride, gather_taps(s), true};   // adjoint of a transposed conv is a strided conv
    if (s.stride == 1) return Prob{ho, wo, s.cout_pad, hin, win, s.cin_pad, hin, win, 1, 1, mirror_taps(s), true};
    auto ph = scatter_phases(s);
    const bool full = ph.size() == 4;
    return Prob{ho, wo, s.cout_pad, hin, win, s.cin_pad, hin / 2, win / 2, 2, 1, ph, full};
}

static void fill_conv_args(awr_conv_args& a, const Prob& p, int B, const float* in, const float* w, const void* w_split, float* out, int T,
                           int kind = AWR_GEMM_OTHER) {
    memset(&a, 0, sizeof a);
    a.in = in; a.w = w; a.out = out; a.w_split = w_split;
    a.B = B; a.Hin = p.Hin; a.Win = p.Win; a.Cin = p.Cin;
    a.Hq = p.Hq; a.Wq = p.Wq; a.Hout = p.Hout; a.Wout = p.Wout; a.N = p.N;
    a.so = p.so; a.si = p.si; a.T = T;
    a.nphase = (int)p.phases.size();
    {   // the accumulation order is captured when the plan is built, like the deterministic mode; AUTO resolves per launch by its K extent
        size_t taps = 0;
        for (auto& ph : p.phases) taps = std::max(taps, ph.taps.size());
        a.accum = awr_resolve_gemm_accum((int)taps * p.Cin, kind);
    }
    for (int i = 0; i < a.nphase; ++i) {
        a.ph[i].py = p.phases[i].py;
        a.ph[i].px = p.phases[i].px;
        a.ph[i].ntaps = (int)p.phases[i].taps.size();
        for (int t = 0; t < a.ph[i].ntaps; ++t) {
            const Tap& tp = p.phases[i].taps[t];
            a.ph[i].tap[t] = (tp.dy & 0xff) | ((tp.dx & 0xff) << 8) | (tp.wt << 16);
        }
    }
}

// weight packing recipes for awr_pack_weight: (d0, d1, T, transpose, rows, ld)
struct PackRecipe { int d0, d1, T, transpose, rows, ld; };
static PackRecipe fwd_pack(const Spec& s) {
    if (!s.deconv) return {s.cout, s.cin, s.T(), 0, (int)round_up(s.cout_pad, N_ALIGN), s.cin_pad};
    return {s.cin, s.cout, s.T(), 1, (int)round_up(s.cout_pad, N_ALIGN), s.cin_pad};
}
static PackRecipe dgrad_pack(const Spec& s) {
    if (!s.deconv) return {s.cout, s.cin, s.T(), 1, (int)round_up(s.cin_pad, N_ALIGN), s.cout_pad};
    return {s.cin, s.cout, s.T(), 0, (int)round_up(s.cin_pad, N_ALIGN), s.cout_pad};
}

// R[d0][t][d1] in the weight's own (d0,d1) order.  conv: D = dY, G = X; deconv: D = X, G = dY.
struct WProb { bool d_is_dy; int Hd, Wd, Cd, Hg, Wg, Cg, sg, ntaps, d0, d1; int8_t dy[16], dx[16]; };
static WProb wgrad_problem(const Spec& s, int hin, int win) {
    int ho, wo;
    s.out_hw(hin, win, ho, wo);
    WProb w;
    memset(&w, 0, sizeof w);
    w.ntaps = s.T();
    for (int ky = 0; ky < s.k; ++ky)
        for (int kx = 0; kx < s.k; ++kx) {
            w.dy[ky * s.k + kx] = (int8_t)(ky - s.pad);
            w.dx[ky * s.k + kx] = (int8_t)(kx - s.pad);
        }
    w.sg = s.stride;
    if (!s.deconv) {
        w.d_is_dy = true; w.Hd = ho; w.Wd = wo; w.Cd = s.cout_pad; w.Hg = hin; w.Wg = win; w.Cg = s.cin_pad; w.d0 = s.cout; w.d1 = s.cin;
    } else {
        w.d_is_dy = false; w.Hd = hin; w.Wd = win; w.Cd = s.cin_pad; w.Hg = ho; w.Wg = wo; w.Cg = s.cout_pad; w.d0 = s.cin; w.d1 = s.cout;
    }
    return w;
}
static void fill_wgrad_args(awr_wgrad_args& a, const WProb& w, int B, const float* D, const float* G, float* R, int ld) {
    memset(&a, 0, sizeof a);
    a.D = D; a.G = G; a.R = R;
    a.B = B; a.Hd = w.Hd; a.Wd = w.Wd; a.Cd = w.Cd; a.Hg = w.Hg; a.Wg = w.Wg; a.Cg = w.Cg; a.sg = w.sg; a.T = w.ntaps; a.ld = ld;
    memcpy(a.dy, w.dy, 16);
    memcpy(a.dx, w.dx, 16);
}

// ------------------------------------------------------------------------------------------
// layers bound to the arenas
// ------------------------------------------------------------------------------------------
struct Packed {
    float* p = nullptr;
    void* split = nullptr;
    int rows = 0, T = 0, ld = 0;
    int64_t numel() const { return (int64_t)rows * T * ld; }
};

struct ConvLayer {
    Spec spec;
    float *w = nullptr, *gw = nullptr, *bias = nullptr, *gbias = nullptr;
    Packed p_fwd, p_dgrad;
    std::string name;
    // fused head: final1 (256->3J) and final2 (256->J) 1x1 convs as one 256->Cp GEMM (resnet_deconv.py:52-53,:133-136 /
    // hourglass.py:137-138,:153-157); Cp = 4J rounded up to 32, extra rows zero
    bool head = false;
    int J = 0, cp = 0;
    float *w1 = nullptr, *gw1 = nullptr, *b1 = nullptr, *gb1 = nullptr, *w2 = nullptr, *gw2 = nullptr, *b2 = nullptr, *gb2 = nullptr, *bias_cat = nullptr;
    const float* bias_ptr() const { return head ? bias_cat : bias; }
};

// conv3 + skip_layer of a hourglass residual whose skip path is a 1x1 conv (hourglass.py:38-47, :44-59), forward-fused: one GEMM
// over K = [conv3's input channels | the block input's channels], packed weight rows [W3 row | Wskip row], bias b3 + bskip
struct DualLayer {
    ConvLayer *c3 = nullptr, *sk = nullptr;
    Packed p;
    float* bias_sum = nullptr;
    std::string name;
};

struct BNLayer {
    int C = 0;
    float *gamma = nullptr, *beta = nullptr, *ggamma = nullptr, *gbeta = nullptr, *rmean = nullptr, *rvar = nullptr;
    std::string name;
};

}  // namespace awrnet

using namespace awrnet;

struct awr_plan;

struct awr_net {
    int kind = 0, nstack = 1, J = 14, downsample = 2, f = 256;
    int depth = 18;                      // kind 0: ResNet depth (18 BasicBlock; 50 / 101 / 152 Bottleneck)
    int nstage = 1, ndeconv = 3;
    Layout layout;
    float *params = nullptr, *grads = nullptr, *buffers = nullptr;
    std::map<std::string, ConvLayer> convs;
    std::map<std::string, BNLayer> bns;
    std::map<std::string, DualLayer> duals;
    std::vector<void*> owned;            // packed weights (device), freed with the net
    std::vector<awr_plan*> plans;
    std::vector<std::string> key_storage;

    float* P(const std::string& key) const { return params + layout.at(key).off; }
    float* G(const std::string& key) const { return grads + layout.at(key).off; }
    float* Bf(const std::string& key) const { return buffers + layout.at(key).off; }
};

namespace awrnet {

static int dev_alloc(std::vector<void*>& owner, size_t bytes, bool zero, void** out) {
    void* p = nullptr;
    if (bytes == 0) bytes = 16;
    HIP_TRY(hipMalloc(&p, bytes));
    if (zero) HIP_TRY(hipMemset(p, 0, bytes));
    owner.push_back(p);
    *out = p;
    return AWR_OK;
}

static int alloc_packed(awr_net* net, Packed& pk, const PackRecipe& r) {
    if (pk.p) return AWR_OK;
    pk.rows = r.rows; pk.T = r.T; pk.ld = r.ld;
    void* p;
    NET_CHECK(dev_alloc(net->owned, (size_t)pk.numel() * 4, true, &p));
    pk.p = (float*)p;
    NET_CHECK(dev_alloc(net->owned, (size_t)pk.numel() * 6, true, &p));      // split image: 3 bf16 pieces per element
    pk.split = p;
    return AWR_OK;
}

static ConvLayer conv_layer(awr_net* n, const std::string& wkey, const Spec& spec, const std::string& bkey = "") {
    ConvLayer c;
    c.spec = spec;
    c.w = n->P(wkey);
    c.gw = n->G(wkey);
    if (!bkey.empty()) {
        c.bias = n->P(bkey);
        c.gbias = n->G(bkey);
    }
    c.name = wkey.substr(0, wkey.rfind('.'));
    return c;
}

static BNLayer bn_layer(awr_net* n, const std::string& prefix) {
    BNLayer b;
    b.C = (int)n->layout.at(prefix + ".weight").numel;
    b.gamma = n->P(prefix + ".weight");
    b.beta = n->P(prefix + ".bias");
    b.ggamma = n->G(prefix + ".weight");
    b.gbeta = n->G(prefix + ".bias");
    b.rmean = n->Bf(prefix + ".running_mean");
    b.rvar = n->Bf(prefix + ".running_var");
    b.name = prefix;
    return b;
}

static int head_layer(awr_net* n, ConvLayer& h, int cin, const std::string& a, const std::string& b, const std::string& name) {
    h = ConvLayer();
    h.head = true;
    h.J = n->J;
    h.cp = (int)round_up(4 * n->J, 32);
    h.spec = make_spec(false, cin, 4 * n->J, 1, 1, 0, 0, h.cp);
    h.w1 = n->P(a + ".weight"); h.gw1 = n->G(a + ".weight"); h.b1 = n->P(a + ".bias"); h.gb1 = n->G(a + ".bias");
    h.w2 = n->P(b + ".weight"); h.gw2 = n->G(b + ".weight"); h.b2 = n->P(b + ".bias"); h.gb2 = n->G(b + ".bias");
    h.w = h.w1; h.gw = h.gw1;
    h.name = name;
    void* p;
    NET_CHECK(dev_alloc(n->owned, (size_t)h.cp * 4, true, &p));
    h.bias_cat = (float*)p;
    return AWR_OK;
}

// name -> layer objects bound to the current arenas (nets.py: _make_layers)
static int make_layers(awr_net* n) {
    n->convs.clear();
    n->bns.clear();
    n->duals.clear();
    if (n->kind == 0) {
        n->convs["pre.0"] = conv_layer(n, "pre.0.weight", make_spec(false, 25, 64, 1, 1, 0, 32));
        n->bns["pre.1"] = bn_layer(n, "pre.1");
        int cin = 64;
        const int planes_[4] = {64, 128, 256, 512}, stride_[4] = {1, 2, 2, 2};
        const bool bott = n->depth != 18;
        const int exp = bott ? 4 : 1;
        const int* nb = resnet_blocks(n->depth);
        for (int li = 1; li <= 4; ++li)
            for (int bi = 0; bi < nb[li - 1]; ++bi) {
                const std::string p = fmt("layer%d.%d", li, bi);
                const int planes = planes_[li - 1], s = bi == 0 ? stride_[li - 1] : 1;
                if (!bott) {
                    n->convs[p + ".conv1"] = conv_layer(n, p + ".conv1.weight", make_spec(false, cin, planes, 3, s, 1));
                    n->bns[p + ".bn1"] = bn_layer(n, p + ".bn1");
                    n->convs[p + ".conv2"] = conv_layer(n, p + ".conv2.weight", make_spec(false, planes, planes, 3, 1, 1));
                    n->bns[p + ".bn2"] = bn_layer(n, p + ".bn2");
                } else {      // Bottleneck: 1x1 -> 3x3 (carries the stride) -> 1x1 x4 (resnet_deconv.py:177-215)
                    n->convs[p + ".conv1"] = conv_layer(n, p + ".conv1.weight", make_spec(false, cin, planes, 1, 1, 0));
                    n->bns[p + ".bn1"] = bn_layer(n, p + ".bn1");
                    n->convs[p + ".conv2"] = conv_layer(n, p + ".conv2.weight", make_spec(false, planes, planes, 3, s, 1));
                    n->bns[p + ".bn2"] = bn_layer(n, p + ".bn2");
                    n->convs[p + ".conv3"] = conv_layer(n, p + ".conv3.weight", make_spec(false, planes, planes * exp, 1, 1, 0));
                    n->bns[p + ".bn3"] = bn_layer(n, p + ".bn3");
                }
                if (n->layout.has(p + ".downsample.0.weight")) {
                    n->convs[p + ".downsample.0"] = conv_layer(n, p + ".downsample.0.weight", make_spec(false, cin, planes * exp, 1, s, 0));
                    n->bns[p + ".downsample.1"] = bn_layer(n, p + ".downsample.1");
                }
                cin = planes * exp;
            }
        for (int i = 0; i < n->ndeconv; ++i) {
            n->convs[fmt("deconv_layers.%d", 3 * i)] = conv_layer(n, fmt("deconv_layers.%d.weight", 3 * i), make_spec(true, cin, 256, 4, 2, 1));
            n->bns[fmt("deconv_layers.%d", 3 * i + 1)] = bn_layer(n, fmt("deconv_layers.%d", 3 * i + 1));
            cin = 256;
        }
        NET_CHECK(head_layer(n, n->convs["head"], 256, "final1", "final2", "final"));
        return AWR_OK;
    }
    const int cp = (int)round_up(4 * n->J, 32);
    for (const auto& en : n->layout.e) {
        const std::string& key = en.key;
        if (en.kind == CONV_W && key.size() > 12 && key.compare(key.size() - 12, 12, ".conv.weight") == 0) {
            if (en.unused) continue;
            const std::string pfx = key.substr(0, key.size() - 12);
            const int cout = (int)en.shape[0], cin = (int)en.shape[1], k = (int)en.shape[2];
            Spec spec;
            if (pfx == "pre.0") spec = make_spec(false, 25, 64, 1, 1, 0, 32);
            else if (pfx.compare(0, 11, "merge_preds") == 0) spec = make_spec(false, cin, cout, 1, 1, 0, cp);
            else spec = make_spec(false, cin, cout, k, 1, (k - 1) / 2);
            n->convs[pfx] = conv_layer(n, key, spec, pfx + ".conv.bias");
        } else if (en.kind == BN_W) {
            const std::string pfx = key.substr(0, key.size() - 7);
            n->bns[pfx] = bn_layer(n, pfx);
        }
    }
    for (int i = 0; i < n->nstack; ++i)
        NET_CHECK(head_layer(n, n->convs[fmt("head.%d", i)], n->f, fmt("outs_1.%d", i), fmt("outs_2.%d", i), fmt("outs.%d", i)));
    return AWR_OK;
}

// ------------------------------------------------------------------------------------------
// plan-time tensor handle: an NHWC fp32 buffer plus (later) its gradient buffer
// ------------------------------------------------------------------------------------------
struct StatBuf {
    double* p = nullptr;
    int nslots = 0;
};

struct FusedStats { double* sp = nullptr; int ns = 0; };      // set by bn_act when the BatchNorm takes the producer's statistics

struct Tn {
    float* buf = nullptr;
    float* grad = nullptr;
    int B = 0, H = 0, W = 0, C = 0;
    bool needs_grad = true;
    StatBuf stats;
    // lazy: the tensor this handle stands for is relu(buf*scale+shift) -- a BatchNorm(+ReLU) output that is never written to
    // HBM; its consumers (conv / wgrad / maxpool loaders) apply the affine on the fly
    bool lazy = false, lz_relu = false;
    const float *lz_scale = nullptr, *lz_shift = nullptr;
    // a materialised BatchNorm output WITHOUT ReLU or residual (the ResNet downsample projection's): what a consumer that adds it to
    // another BatchNorm's output before a ReLU needs to reduce this BatchNorm's backward sums together with its own
    const float *bn_y = nullptr, *bn_coef4 = nullptr;
    StatBuf pre_sums;      // set by that consumer's backward: sum g / sum g*xhat of THIS BatchNorm are already in here
    // a conv output y whose BatchNorm's backward is NOT on the critical chain (set by bn_bwd, consumed by the conv's conv_bwd): the conv's data
    // gradient reads g = d(loss)/d(bn(y)) (masked) and y itself and forms d(y) = a1 g + a2 (y - mean) + a3 on the fly (awr_conv_args.in_bnb_y);
    // d(y) is still written -- by an apply launch that travels with the weight gradient on its side stream
    bool conv_out = false;
    awr_conv_args* pair_args = nullptr;       // written by a fused inference pair (conv_pair): a 2x2 max-pool of it can ride in that launch (pool_out)
    struct FusedStats* fstats = nullptr;      // produced by a max-pool / up-sampling add that can accumulate the next BatchNorm's statistics itself
    bool half_ok = false;     // conv output whose data gradient may run as two half-batch parts (set by conv())
    bool half_dy = false;     // ... and whose BatchNorm backward writes d(y) half by half: the second half on the weight gradient's stream
    int conv_taps = 0;
    const float *lz_g = nullptr, *lz_lin4 = nullptr, *lz_mean = nullptr, *lz_invstd = nullptr, *lz_coef = nullptr;
    std::string name;
    int64_t npix() const { return (int64_t)B * H * W; }
    int64_t numel() const { return npix() * C; }
};

enum OpKind { OP_CALL = 0, OP_ZERO, OP_COPY, OP_BUCKET, OP_FORK, OP_ENDFORK, OP_JOIN, OP_HALFWAIT };

struct Op {
    int kind = OP_CALL;
    std::function<int(void*)> fn;
    std::string name;
    void* p = nullptr;        // OP_ZERO / OP_COPY destination
    const void* q = nullptr;  // OP_COPY source
    size_t bytes = 0;
    int64_t lo = 0, hi = 0;   // OP_BUCKET
    int sid = 0;              // fork / join stream id
    bool side_ok = false;     // weight-gradient launch that may run on a side stream
    bool pair_next = false;   // side_ok launch that shares the side stream of the NEXT side_ok launch (its producer: awr_bn_bwd_apply -> awr_conv_wgrad)
    bool half_sig = false;    // side_ok launch (the second half of a BatchNorm-backward apply): record "half B is written" behind it; OP_HALFWAIT makes the
                              // issuing chain wait for that record (NOT for the weight gradient queued behind it on the same side stream)
    bool gemm = false;        // conv / stem family (timed by run_timed)
    bool boundary = false;    // NCHW <-> NHWC bridge at the reference boundary: skipped while the plan's NHWC boundary is on
    double macs = 0.0;
};

struct GemmRef {              // what autotune iterates over
    awr_conv_args* ca = nullptr;
    awr_wgrad_args* wa = nullptr;
    std::string name;
    int tm = 0, tn = 0, tb = 0;
    float us = 0.f;
    bool tuned = false;
};

struct UnpackJobH {           // host copy of awr_unpack_job + where its gradient lives in the arena
    awr_unpack_job job;
};

struct GradWrite { int64_t lo, hi; int ready; int job; };

}  // namespace awrnet

typedef void (*awr_bucket_cb)(void* user, int64_t lo, int64_t hi, void* stream);

struct awr_plan {
    awr_net* net = nullptr;
    int B = 0, H = 0;
    bool training = false, det = false;
    int bn_repeat = 1, n_buckets = 1;
    unsigned supervised = 0;
    std::vector<Op> fwd, bwd, pack_ops;
    std::vector<std::function<int()>> nodes;       // backward emitters, in forward order
    std::vector<ConvLayer*> layers;                // layers whose packed copies this plan refreshes
    int n_wino = 0;      // forward (and weight-gradient) launches that run as Winograd F(2x2, 3x3)
    std::vector<std::pair<ConvLayer*, float*>> wino_d;      // (layer, mirrored U[16][cout_pad][cin_pad])
    std::vector<std::pair<const awr_conv_args*, double>> wino_dg;      // candidate data-gradient launches (argument block, MACs): Winograd if the COMPLETED block is supported
    double wino_macs = 0;            // algorithmic multiply-adds of those launches (they execute 16 / 36 of them)
    std::vector<std::pair<ConvLayer*, float*>> wino;      // ... and whose forward runs as Winograd F(2x2, 3x3): (layer, U[16][cin_pad][cout_pad])
    std::deque<awr_wino_args> wino_args;                  // argument blocks of those forward launches (stable addresses)
    std::vector<DualLayer*> dual_layers;
    std::deque<Tn> tensors;
    std::deque<awr_conv_args> cargs;
    std::deque<awrnet::FusedStats> fstats;      // (deque: stable addresses, the forward launches and bn_act share them)
    std::deque<awr_wgrad_args> wargs;
    std::vector<void*> owned;
    std::vector<std::pair<void*, size_t>> zero_init;   // atomic accumulators that must be zero before the first real step
    int64_t bytes = 0;
    float* img = nullptr;
    std::vector<float*> outputs, grad_outs;
    std::vector<Tn*> head_preds;               // per stage: the head GEMM's NHWC output
    std::vector<float*> head_grads;            // per stage: NHWC buffer the backward reads d(pred) from (nullptr: second producer)
    bool nhwc_boundary = false;
    // wgrad split-K scratch arena
    float* scratch = nullptr;
    int64_t scratch_used = 0, scratch_cap = 0;
    std::vector<awr_unpack_job> unpack_jobs;
    std::vector<GradWrite> grad_writes;
    std::vector<void*> job_tables;
    std::map<Tn*, std::vector<awr_conv_args*>> grad_writers;
    std::map<Tn*, int> fork_results;
    int fork_sid = 0;
    std::vector<GemmRef> gemms;
    struct Bucket { int64_t lo, hi; int ready; };
    std::vector<Bucket> buckets;
    bool built_bwd = false;
    // pack tables (device) for refresh_weights: [want_split][0 = forward layouts (+ dual layers), 1 = data-gradient layouts]
    void* pack_tab[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    int pack_njobs[2][2] = {{0, 0}, {0, 0}};
    int64_t pack_rows[2][2] = {{0, 0}, {0, 0}};
    bool pack_built[2] = {false, false};

    // streams
    std::vector<hipStream_t> side;
    std::vector<hipStream_t> branch;      // backward branches (an hourglass level's full-resolution skip residual)
    hipStream_t comm = nullptr;
    bool comm_owned = false;
    std::vector<hipEvent_t> events;
    size_t ev_next = 0;
    awr_bucket_cb bucket_cb = nullptr;
    void* bucket_user = nullptr;
    awr_dp* dp = nullptr;                 // native RCCL exchange of the buckets (awr_plan_set_dp)
    int dp_error = 0;
    unsigned long long dp_gen = 0;      // generation of the attached communicator (awr_dp_generation at awr_plan_set_dp)
    bool dp_lost = false;               // the attached communicator was destroyed under the plan: EVERY backward fails until awr_plan_set_dp is called again
    awr_bucket_cb saved_cb = nullptr;     // the host's callback while dp is attached
    void* saved_user = nullptr;
};

namespace awrnet {

// ---- plan building -------------------------------------------------------------------------------------------
struct Tn;
static int env_or(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

struct ConvOpt {
    const float *in_scale = nullptr, *in_shift = nullptr;
    bool relu_in = false;
    const float *out_scale = nullptr, *out_shift = nullptr;
    Tn* res = nullptr;
    bool relu_out = false, want_stats = false, use_bias = true;
};

struct Builder {
    awr_plan& P;
    awr_net& N;
    int err = AWR_OK;
    explicit Builder(awr_plan& p) : P(p), N(*p.net) {}

    template <typename Tp>
    Tp* alloc(int64_t count, bool zero = false) {
        void* p = nullptr;
        const size_t bytes = (size_t)count * sizeof(Tp);
        if (int e = dev_alloc(P.owned, bytes, zero, &p)) {
            err = e;
            return nullptr;
        }
        P.bytes += (int64_t)bytes;
        if (zero) P.zero_init.push_back({p, bytes});
        return (Tp*)p;
    }
    Tn* new_t(int B, int H, int W, int C, bool needs_grad = true, const std::string& name = "") {
        P.tensors.emplace_back();
        Tn* t = &P.tensors.back();
        t->buf = alloc<float>((int64_t)B * H * W * C);
        t->B = B; t->H = H; t->W = W; t->C = C;
        t->needs_grad = needs_grad;
        t->name = name;
        return t;
    }
    StatBuf stat_buf(int nslots, int C) {
        StatBuf s;
        s.p = alloc<double>((int64_t)nslots * 2 * C, true);
        s.nslots = nslots;
        return s;
    }
    int gemm_slots(int B, int Hq, int Wq, int Nn, int nphase) const {
        if (!P.det) return AWR_STAT_SLOTS;
        return (int)(((int64_t)B * Hq * Wq + 63) / 64) * ((Nn + 63) / 64) * nphase;
    }
    int reduce_slots() const { return P.det ? AWR_REDUCE_MAX_BLOCKS : AWR_STAT_SLOTS; }

    float* scratch(int64_t n) {        // slice of the split-K scratch arena (zeroed by ONE fill per step)
        n = round_up(n, 4);
        const int64_t off = P.scratch_used;
        P.scratch_used += n;
        if (!P.scratch) P.scratch = alloc<float>(P.scratch_cap);
        if (P.scratch_used > P.scratch_cap) {
            set_error("plan: wgrad scratch arena too small");
            err = AWR_ERR_ARG;
            return nullptr;
        }
        return P.scratch + off;
    }

    // the forward of this stride-1 3x3 layer may run as Winograd F(2x2, 3x3) (mode, kernel constraints, launch-size rule)
    bool wino_fwd_ok(const Spec& spec, int B, int H, int W) const {
        return awr_get_conv_winograd() && !P.det && awr_get_gemm_products() == 1 && !spec.deconv && spec.k == 3 && spec.stride == 1 && spec.pad == 1 &&
               awr_wino_eligible(B, H, W, spec.cin_pad, spec.cout_pad);
    }

    Op& push(std::vector<Op>& list, const std::string& name, std::function<int(void*)> fn) {
        list.emplace_back();
        Op& o = list.back();
        o.fn = std::move(fn);
        o.name = name;
        return o;
    }
    Op& f(const std::string& name, std::function<int(void*)> fn) { return push(P.fwd, name, std::move(fn)); }
    Op& b(const std::string& name, std::function<int(void*)> fn) { return push(P.bwd, name, std::move(fn)); }

    void note_grad(const float* tensor, int64_t numel, int job = -1) {
        if (!N.grads || !tensor) return;
        const int64_t lo = tensor - N.grads;
        P.grad_writes.push_back({lo, lo + numel, (int)P.bwd.size() - 1, job});
    }

    // -> (gradient buffer, accumulate?) and marks the gradient as live
    std::pair<float*, bool> gtarget(Tn* t, bool dgrad_writer = false) {
        if (!dgrad_writer) P.grad_writers[t].push_back(nullptr);
        if (!t->grad) {
            t->grad = alloc<float>(t->numel());
            return {t->grad, false};
        }
        return {t->grad, true};
    }
    // grad(t) += src_grad where src_grad is a finished gradient buffer: alias when first
    void contribute_identity(Tn* t, float* src_grad) {
        if (!t->needs_grad) return;
        P.grad_writers[t].push_back(nullptr);
        if (!t->grad) {
            t->grad = src_grad;
        } else {
            float* g = t->grad;
            const int64_t n = t->numel();
            b("awr_add", [=](void* s) { return awr_add(g, src_grad, g, n, s); });
        }
    }
    void use_layer(ConvLayer* l) {
        for (auto* x : P.layers)
            if (x == l) return;
        if (int e = alloc_packed(&N, l->p_fwd, fwd_pack(l->spec))) err = e;
        if (P.training)
            if (int e = alloc_packed(&N, l->p_dgrad, dgrad_pack(l->spec))) err = e;
        P.layers.push_back(l);
    }

    static double gemm_macs(const Prob& fp, int B, const Spec& s) {
        double taps = 0;
        for (auto& ph : fp.phases) taps += ph.taps.size();
        taps /= (double)(fp.so * fp.so);
        return (double)B * fp.Hout * fp.Wout * s.cout * taps * s.cin;
    }

    // ---- forward branches that may run beside the main chain ----
    void fork(int sid) {
        P.fork_sid = sid;
        Op& o = f("__fork__", nullptr);
        o.kind = OP_FORK;
        o.sid = sid;
    }
    void end_fork(Tn* result) {
        Op& o = f("__endfork__", nullptr);
        o.kind = OP_ENDFORK;
        o.sid = P.fork_sid;
        P.fork_results[result] = P.fork_sid;
    }
    // Backward branches.  The emitters run in reverse node order, so a marker node pushed AFTER a group of nodes is emitted
    // BEFORE that group's backward ops.
    void bwd_marker_node(int kind, int sid) {
        P.nodes.push_back([=]() {
            Op& o = b(kind == OP_FORK ? "__fork__" : kind == OP_ENDFORK ? "__endfork__" : "__join__", nullptr);
            o.kind = kind;
            o.sid = sid;
            return err;
        });
    }
    void join_if(Tn* t) {
        if (!t) return;
        auto it = P.fork_results.find(t);
        if (it == P.fork_results.end()) return;
        Op& o = f("__join__", nullptr);
        o.kind = OP_JOIN;
        o.sid = it->second;
        P.fork_results.erase(it);
    }

    // ---- ops ----
    // y = conv(x) [+bias] [*s+t] [+res] [relu]
    Tn* conv(Tn* x, ConvLayer* layer, ConvOpt o = ConvOpt()) {
        use_layer(layer);
        const Spec& spec = layer->spec;
        const int B = x->B;
        const Prob prob = fwd_problem(spec, x->H, x->W);
        Tn* y = new_t(B, prob.Hout, prob.Wout, prob.N, true, layer->name + ".out");
        y->conv_out = o.res == nullptr;      // (its gradient has exactly one reader pair: this conv's weight and data gradient)
        y->conv_taps = spec.T();
        {   // half-batch BatchNorm-backward wavefront (bn_bwd / conv_bwd).  OPT-IN: built for VERDICT r4 item 3, parity-green, and measured SLOWER on every
            // shape (ResNet18 12.8 -> 13.4 ms, Hourglass-1 23.5 -> 24.7 ms at 32 768 rows per half: two half-batch data gradients lose more than the
            // hidden apply pass wins; profiles/r05_half_batch_wavefront.txt).  AWR_HALF_BNB_MIN_ROWS = rows of the data-gradient GEMM per half from
            // which a layer takes part (read per plan); 0 = off, the default.
#ifdef AWR_STUDY
            const int64_t half_min = env_or("AWR_HALF_BNB_MIN_ROWS", 0);
#else
            const int64_t half_min = 0;      // (the wavefront's scheduling states exist in study builds only: ADVICE r5)
#endif
            y->half_ok = P.training && half_min > 0 && !o.res && x->needs_grad && B % 2 == 0 && (int64_t)(B / 2) * x->H * x->W >= half_min;
        }
        if (o.want_stats) y->stats = stat_buf(gemm_slots(B, prob.Hq, prob.Wq, prob.N, (int)prob.phases.size()), prob.N);
        const float* bias = o.use_bias ? layer->bias_ptr() : nullptr;
        join_if(o.res);
        if (x->lazy) {
            o.in_scale = x->lz_scale;
            o.in_shift = x->lz_shift;
            o.relu_in = x->lz_relu;
        }
        // Winograd F(2x2, 3x3) forward (round 6, awr_set_conv_winograd; csrc/awr_wino.hip): stride-1 3x3 convolutions whose epilogue is bias /
        // ReLU / statistics.  2.25x fewer multiplies, chains of Cin terms instead of 9 Cin (so no blocked accumulation is needed).  Weight and
        // data gradients stay direct: they only need x and d(y).
        // Inference plans (round 6, last step): the folded eval-mode BatchNorm (out_scale / out_shift) and the residual add of a BasicBlock are epilogue
        // forms of the kernel too.
        if (wino_fwd_ok(spec, B, x->H, x->W) && (!P.training || (!o.res && !o.out_scale)) && !(o.out_scale && o.want_stats) &&
            (o.in_scale == nullptr) == (o.in_shift == nullptr) && (o.out_scale == nullptr) == (o.out_shift == nullptr)) {
            float* U = nullptr;
            for (auto& wl : P.wino)
                if (wl.first == layer) U = wl.second;
            if (!U) {
                U = alloc<float>((int64_t)16 * spec.cin_pad * spec.cout_pad);
                P.wino.push_back({layer, U});
            }
            P.wino_args.emplace_back();
            awr_wino_args* wa = &P.wino_args.back();
            memset(wa, 0, sizeof *wa);
            wa->in = x->buf; wa->U = U; wa->bias = bias; wa->in_scale = o.in_scale; wa->in_shift = o.in_shift; wa->out = y->buf;
            wa->stats = y->stats.p; wa->nslots = y->stats.p ? y->stats.nslots : 0;
            wa->res = o.res ? o.res->buf : nullptr;
            wa->out_scale = o.out_scale; wa->out_shift = o.out_shift;
            wa->B = B; wa->H = x->H; wa->W = x->W; wa->C = spec.cin_pad; wa->N = spec.cout_pad; wa->relu = o.relu_out; wa->relu_in = o.relu_in;
            const std::string name = "awr_conv_gemm:" + layer->name;
            Op& op = f(name, [wa](void* s) { return awr_wino_conv(wa, s); });
            op.gemm = true;
            op.macs = gemm_macs(prob, B, spec);
            P.n_wino++;
            P.wino_macs += op.macs;
            if (P.training) {
                const bool has_bias = bias != nullptr;
                P.nodes.push_back([=]() { return conv_bwd(x, y, layer, nullptr, has_bias); });
            }
            return y;
        }
        P.cargs.emplace_back();
        awr_conv_args* a = &P.cargs.back();
        fill_conv_args(*a, prob, B, x->buf, layer->p_fwd.p, layer->p_fwd.split, y->buf, spec.T(), P.training ? AWR_GEMM_FORWARD : AWR_GEMM_OTHER);
        a->in_scale = o.in_scale; a->in_shift = o.in_shift; a->bias = bias;
        a->out_scale = o.out_scale; a->out_shift = o.out_shift;
        a->res = o.res ? o.res->buf : nullptr;
        a->stats = y->stats.p;
        a->stat_slots = y->stats.p ? y->stats.nslots : 0;
        a->relu_in = o.relu_in; a->relu_out = o.relu_out;
        if (!P.training && !a->stats && a->accum == 0) {
            // low-batch inference: a launch of a few workgroups, each walking a long K loop, is latency-bound -> split-K scratch
            // (awr_conv_gemm picks the depth from the workgroup count, the tuner refines it).  Not in a parity plan (blocked accumulation):
            // the split-K kernel and the fused pair below accumulate in their own order -- a plan built in that mode only contains launch
            // forms that honour it
            const int64_t wgs = ((int64_t)B * prob.Hq * prob.Wq + 63) / 64 * ((prob.N + 63) / 64) * (int64_t)prob.phases.size();
            int minsteps = 1 << 30;
            for (auto& ph : prob.phases) minsteps = std::min(minsteps, (int)ph.taps.size() * (a->Cin / 32));
            int smax = 1;
            while (smax < 16 && wgs * smax < 2048 && minsteps / (smax * 2) >= 4) smax *= 2;
            if (smax > 1) {
                a->partial = alloc<float>((int64_t)smax * y->numel());
                a->split_max = smax;
            }
        }
        const std::string name = "awr_conv_gemm:" + layer->name;
        Op& op = f(name, [a](void* s) { return awr_conv_gemm(a, s); });
        op.gemm = true;
        op.macs = gemm_macs(prob, B, spec);
        P.gemms.push_back({a, nullptr, name});
        if (P.training) {
            Tn* res = o.res;
            const bool has_bias = bias != nullptr;
            P.nodes.push_back([=]() { return conv_bwd(x, y, layer, res, has_bias); });
        }
        return y;
    }

    // Inference: y = conv3(relu(bn3(conv2(x)))) + res as ONE launch (awr_conv_args.w2): conv2 has 128 output channels, so a 64x128 tile holds
    // every channel of its pixels and feeds the 1x1 conv3 from LDS -- the intermediate never goes to HBM.  o = conv2's epilogue (folded bn3, ReLU).
    // nullptr when the pair does not qualify (shape, product mode, too few workgroups to fill the chip without split-K).
    Tn* conv_pair(Tn* x, ConvLayer* c2, const ConvOpt& o, ConvLayer* c3, Tn* res, DualLayer* dual = nullptr, Tn* xin = nullptr) {
        const bool off = getenv("AWR_NO_FUSE2") != nullptr;                    // same-box A/B hook (read per plan)
        const int min_wgs = env_or("AWR_FUSE2_MIN_WGS", 1024);                 // (tests force the fused form on small batches)
        const bool no_dual = getenv("AWR_NO_FUSE2_DUAL") != nullptr;
        if (dual) c3 = dual->c3;
        const Spec &s2 = c2->spec, &s3 = c3->spec;
        const int64_t wgs = ((int64_t)x->B * x->H * x->W + 63) / 64;
        const int n1 = s2.cout;
        if (off || P.training || awr_resolve_gemm_accum(s2.T() * s2.cin_pad, AWR_GEMM_OTHER) != 0 || awr_resolve_gemm_accum(s3.cin_pad, AWR_GEMM_OTHER) != 0 || awr_get_gemm_products() != 1 || s2.deconv || s3.deconv || s2.stride != 1 || s3.k != 1 || s3.stride != 1 ||
            (n1 != 128 && n1 != 64) || s3.cin != n1 || s3.cout != 2 * n1 || wgs < min_wgs || x->lazy || (dual && no_dual))
            return nullptr;
        if (dual && (dual->sk->spec.k != 1 || dual->sk->spec.stride != 1 || dual->sk->spec.cout != s3.cout || dual->sk->spec.cin_pad % 32 != 0 ||
                     xin->lazy || xin->H != x->H || xin->W != x->W || xin->C != dual->sk->spec.cin_pad))
            return nullptr;
        use_layer(c2);
        const int B = x->B;
        const Prob prob = fwd_problem(s2, x->H, x->W);
        if (prob.so != 1 || prob.phases.size() != 1 || prob.N != n1) return nullptr;
        const float *w2, *bias2;
        int n1x = 0;
        if (dual) {
            (void)conv_dual_prepare(dual);      // [W3 | Wskip] side by side, biases summed (the packed buffer conv_dual launches with)
            w2 = dual->p.p;
            bias2 = dual->bias_sum;
            n1x = dual->sk->spec.cin_pad;
        } else {
            use_layer(c3);
            w2 = c3->p_fwd.p;
            bias2 = c3->bias_ptr();
        }
        if (err) return nullptr;
        Tn* y = new_t(B, prob.Hout, prob.Wout, 2 * n1, true, (dual ? dual->name : c3->name) + ".out");
        join_if(res);
        P.cargs.emplace_back();
        awr_conv_args* a = &P.cargs.back();
        fill_conv_args(*a, prob, B, x->buf, c2->p_fwd.p, c2->p_fwd.split, y->buf, s2.T());
        a->N1 = n1;
        a->N = 2 * n1;
        a->w2 = w2;
        a->bias = c2->bias_ptr();
        a->bias2 = bias2;
        a->in_scale = o.in_scale; a->in_shift = o.in_shift; a->relu_in = o.relu_in;
        a->out_scale = o.out_scale; a->out_shift = o.out_shift; a->relu_out = o.relu_out;
        a->res = res ? res->buf : nullptr;
        if (dual) { a->in2 = xin->buf; a->N1x = n1x; }
        if (n1 == 64) a->tile_m = env_or("AWR_FUSE2_TM64", 2), a->tile_n = 1;      // 128x64 tile (what the tuner picks for this conv on its own)
        y->pair_args = a;
        const std::string tail = dual ? "conv3+skip_layer" : c3->name.substr(c3->name.rfind('.', c3->name.rfind('.') - 1) + 1);
        const std::string name = "awr_conv_gemm:" + c2->name + "+" + tail;
        Op& op = f(name, [a](void* s) { return awr_conv_gemm(a, s); });
        op.gemm = true;
        op.macs = gemm_macs(prob, B, s2) + gemm_macs(fwd_problem(s3, prob.Hout, prob.Wout), B, s3) +
                  (dual ? gemm_macs(fwd_problem(dual->sk->spec, prob.Hout, prob.Wout), B, dual->sk->spec) : 0.0);
        GemmRef g{a, nullptr, name};
        g.tm = n1 == 64 ? a->tile_m : 1; g.tn = n1 == 64 ? 1 : 2; g.tuned = true;      // one geometry: nothing for the tuner to choose
        P.gemms.push_back(g);
        return y;
    }

    // registers a dual layer with the plan (packed [W3 | Wskip] buffer + summed bias, refreshed with the weights); idempotent
    int conv_dual_prepare(DualLayer* d) {
        use_layer(d->c3);
        use_layer(d->sk);
        bool seen = false;
        for (auto* q : P.dual_layers) seen = seen || q == d;
        const int cin1 = d->c3->spec.cin_pad, cin2 = d->sk->spec.cin_pad, cout = d->c3->spec.cout;
        if (!seen) {
            if (!d->p.p) {
                PackRecipe r{cout, cin1 + cin2, 1, 0, (int)round_up(cout, N_ALIGN), cin1 + cin2};
                if (int e = alloc_packed(&N, d->p, r)) err = e;
                void* bs = nullptr;
                if (int e = dev_alloc(N.owned, (size_t)cout * 4, true, &bs)) err = e;
                d->bias_sum = (float*)bs;
            }
            P.dual_layers.push_back(d);
        }
        return err;
    }

    // y = conv3(a) + skip(x) (+ both biases) as ONE launch (FP32-MFMA mode): the skip branch's output is never written or re-read.
    // Backward: the two layers' own weight / data gradients, both reading d(y).
    Tn* conv_dual(Tn* a, Tn* x, DualLayer* d, bool want_stats) {
        (void)conv_dual_prepare(d);
        const int cin1 = d->c3->spec.cin_pad, cin2 = d->sk->spec.cin_pad, cout = d->c3->spec.cout;
        const int B = a->B;
        Spec spec = make_spec(false, cin1 + cin2, cout, 1, 1, 0);
        const Prob prob = fwd_problem(spec, a->H, a->W);
        Tn* y = new_t(B, prob.Hout, prob.Wout, prob.N, true, d->name + ".out");
        if (want_stats) y->stats = stat_buf(gemm_slots(B, prob.Hq, prob.Wq, prob.N, 1), prob.N);
        P.cargs.emplace_back();
        awr_conv_args* ca = &P.cargs.back();
        fill_conv_args(*ca, prob, B, a->buf, d->p.p, nullptr, y->buf, 1, P.training ? AWR_GEMM_FORWARD : AWR_GEMM_OTHER);
        ca->in2 = x->buf;
        ca->Cin1 = cin1;
        if (a->lazy) { ca->in_scale = a->lz_scale; ca->in_shift = a->lz_shift; ca->relu_in = a->lz_relu; }
        ca->bias = d->bias_sum;
        ca->stats = y->stats.p;
        ca->stat_slots = y->stats.p ? y->stats.nslots : 0;
        const std::string name = "awr_conv_gemm:" + d->name;
        Op& op = f(name, [ca](void* s) { return awr_conv_gemm(ca, s); });
        op.gemm = true;
        op.macs = gemm_macs(fwd_problem(d->c3->spec, a->H, a->W), B, d->c3->spec) + gemm_macs(fwd_problem(d->sk->spec, x->H, x->W), B, d->sk->spec);
        P.gemms.push_back({ca, nullptr, name});
        if (P.training) {
            ConvLayer *c3 = d->c3, *sk = d->sk;
            P.nodes.push_back([=]() {
                NET_CHECK(conv_bwd(a, y, c3, nullptr, true));
                return conv_bwd(x, y, sk, nullptr, true);
            });
        }
        return y;
    }

    int conv_bwd(Tn* x, Tn* y, ConvLayer* layer, Tn* res, bool has_bias) {
        const Spec& spec = layer->spec;
        const int B = x->B, H = x->H, W = x->W;
        float* dy = y->grad;
        if (!dy) {
            set_error("plan: no gradient reached %s", y->name.c_str());
            return AWR_ERR_ARG;
        }
        // the bias gradient (column sums of dY) falls out of the slices a conv's wgrad kernel stages anyway; only a biased
        // TRANSPOSED conv (none in the reference nets) needs the stand-alone reduction
        const bool fused_bias = has_bias && !spec.deconv;
        if (has_bias && !fused_bias) {
            float* tgt = layer->gbias;
            const int64_t npix = y->npix();
            const int C = y->C;
            b("awr_bias_grad", [=](void* s) { return awr_bias_grad(dy, npix, C, tgt, 0, s); });
            note_grad(tgt, C);
        }
        // weight gradient: split-K partial sums into a packed buffer, then scatter to checkpoint layout
        const WProb wp = wgrad_problem(spec, H, W);
        const int ld = wp.Cg;
        const int64_t rsize = (int64_t)wp.Cd * wp.ntaps * ld;
        const float* D = wp.d_is_dy ? dy : x->buf;
        const float* G = wp.d_is_dy ? x->buf : dy;
        P.wargs.emplace_back();
        awr_wgrad_args* wa = &P.wargs.back();
        fill_wgrad_args(*wa, wp, B, D, G, nullptr, ld);
        if (x->lazy) {
            if (wp.d_is_dy) { wa->g_scale = x->lz_scale; wa->g_shift = x->lz_shift; wa->g_relu = x->lz_relu; }
            else { wa->d_scale = x->lz_scale; wa->d_shift = x->lz_shift; wa->d_relu = x->lz_relu; }
        }
        int nsum = 1, rstride = 0, bslots = AWR_STAT_SLOTS, bstride = y->C;
        float *R = nullptr, *bsum = nullptr;
        if (P.det) {
            // every K-chunk stores its own copy of the packed gradient (+ bias column sums); the batched scatter sums them in order
            wa->split_stride = rsize;
            const int tiles64 = ((wp.Cd + 63) / 64) * ((wp.Cg + 63) / 64) * wp.ntaps;
            int ms = (2048 + tiles64 - 1) / tiles64;
            wa->max_split = ms < 1 ? 1 : (ms > 256 ? 256 : ms);
            wa->R = const_cast<float*>(D);          // placeholder for the query
            int ns = 0;
            NET_CHECK(awr_conv_wgrad_splits(wa, &ns));
            nsum = ns; rstride = (int)rsize; bslots = ns; bstride = wp.Cd;
            wa->max_split = ns;      // the copies that exist: a later tile / target change can never make the kernel write beyond them
            R = alloc<float>((int64_t)nsum * rsize);
            if (fused_bias) bsum = alloc<float>((int64_t)nsum * wp.Cd);
        } else {
            R = scratch(rsize);
            if (fused_bias) bsum = scratch((int64_t)AWR_STAT_SLOTS * y->C);
        }
        if (err) return err;
        wa->R = R;
        wa->d_colsum = bsum;
        const std::string wname = "awr_conv_wgrad:" + layer->name;
        const double layer_macs = gemm_macs(fwd_problem(spec, H, W), B, spec);      // forward, data gradient and weight gradient cost the same
        if (y->lz_g) {      // d(y) for the weight gradient: written on ITS stream, right in front of it (the data gradient below does not wait for it)
            const float *g = y->lz_g, *yb = y->buf, *mean = y->lz_mean, *invstd = y->lz_invstd, *coef = y->lz_coef;
            const int64_t npix = y->npix();
            const int C = y->C;
            Op& aop = b("awr_bn_bwd_apply", [=](void* s) { return awr_bn_bwd_apply_only(g, nullptr, yb, mean, invstd, nullptr, nullptr, coef, npix, C, dy, nullptr, nullptr, s); });
            aop.side_ok = (res == nullptr);
            aop.pair_next = true;
        }
        // Winograd-domain weight gradient (winograd = "full"; csrc/awr_wino.hip): stride-1 3x3 layers with channel counts in multiples of 64 whose K loop
        // is long enough to pay for the per-split copies; writes the same packed R (and slot 0 of the bias column sums) the direct kernel accumulates into
        const bool wino_w = (awr_get_conv_winograd() & 3) >= 2 && !P.det && awr_get_wgrad_products() == 1 && spec.k == 3 && spec.stride == 1 && spec.pad == 1 &&
                            !spec.deconv && !layer->head && wp.d_is_dy && awr_wino_wgrad_eligible(B, H, W, spec.cin_pad, spec.cout_pad);
        float* wscratch = nullptr;
        if (wino_w) {
            wscratch = alloc<float>(awr_wino_wgrad_scratch(B, H, W, spec.cin_pad, spec.cout_pad));
            if (err) return err;
            P.n_wino++;
            P.wino_macs += layer_macs;
        }
        const float *wx = x->buf, *wsc = x->lazy ? x->lz_scale : nullptr, *wsh = x->lazy ? x->lz_shift : nullptr;
        const int wrelu = x->lazy ? x->lz_relu : 0, wC = spec.cin_pad, wN = spec.cout_pad;
        Op& wop = wino_w ? b(wname, [=](void* s) { return awr_wino_wgrad(wx, dy, wsc, wsh, wrelu, B, H, W, wC, wN, wscratch, R, ld, bsum, s); })
                         : b(wname, [wa](void* s) { return awr_conv_wgrad(wa, s); });      // (reference into the op vector: do not use after the next push)
        wop.gemm = true;
        wop.macs = layer_macs;
        // safe to run beside the main chain when dY is written once before this node and nobody touches it again.  With a fused
        // residual, d(res) ALIASES dY and later nodes accumulate into it in place -> stays on the main stream.
        wop.side_ok = (res == nullptr);
        // 128 -> 128 3x3 layers on >= 64x64 maps (the Hourglass residuals at full resolution): the one-wave-per-tap-row kernel stages
        // D and the halo'd G patch once for all nine taps (measured: 124 vs 114 TF in isolation, Hourglass-1 step 26.44 -> 25.79 ms; writing relu(bn2(.)) out for them on top: 25.87); on
        // every other shape the workgroup-per-tap kernel is equal or faster (profiles/r02_microbench_wgrad_algos.txt)
        if (!P.det && awr_get_wgrad_products() == 1 && spec.k == 3 && spec.stride == 1 && !spec.deconv && spec.cin == 128 && spec.cout == 128 &&
            H * W >= 4096 && (int64_t)B * H * W >= (1 << 17) && x->lazy)      // (a plain input takes the kernel-row kernel: awr_conv_wgrad's default)
            wa->algo = 2;
        if (!wino_w) P.gemms.push_back({nullptr, wa, wname});      // (tunable launches: tiles / algorithm of the direct kernel)
        // scattered back to checkpoint layout by a batched launch (end of backward / end of its bucket)
        auto add_job = [&](const float* packed, float* grad, int d0, int d1, int T_, int ld_, int slots, int stride, int64_t numel) {
            awr_unpack_job j;
            memset(&j, 0, sizeof j);
            j.packed = packed; j.grad = grad; j.d0 = d0; j.d1 = d1; j.T = T_; j.ld = ld_; j.slots = slots; j.slot_stride = stride;
            P.unpack_jobs.push_back(j);
            note_grad(grad, numel, (int)P.unpack_jobs.size() - 1);
        };
        if (!layer->head) {
            const int d0 = spec.deconv ? spec.cin : spec.cout, d1 = spec.deconv ? spec.cout : spec.cin;
            add_job(R, layer->gw, d0, d1, spec.T(), ld, nsum, rstride, (int64_t)d0 * d1 * spec.T());
            if (bsum) add_job(bsum, layer->gbias, 1, spec.cout, 1, spec.cout, bslots, bstride, spec.cout);
        } else {
            const int J = layer->J, cin = spec.cin;
            add_job(R, layer->gw1, 3 * J, cin, 1, ld, nsum, rstride, (int64_t)3 * J * cin);
            add_job(R + (int64_t)3 * J * ld, layer->gw2, J, cin, 1, ld, nsum, rstride, (int64_t)J * cin);
            if (bsum) {
                add_job(bsum, layer->gb1, 1, 3 * J, 1, 3 * J, bslots, bstride, 3 * J);
                add_job(bsum + 3 * J, layer->gb2, 1, J, 1, J, bslots, bstride, J);
            }
        }
        // data gradient
        if (x->needs_grad) {
            const Prob dp = dgrad_problem(spec, H, W);
            auto tgt = gtarget(x, true);
            float* gx = tgt.first;
            bool acc = tgt.second;
            if (!dp.full && !acc) {
                Op& z = b("__zero__", nullptr);
                z.kind = OP_ZERO;
                z.p = gx;
                z.bytes = (size_t)x->numel() * 4;
                acc = true;
            }
            P.cargs.emplace_back();
            awr_conv_args* da = &P.cargs.back();
            fill_conv_args(*da, dp, B, y->lz_g ? y->lz_g : dy, layer->p_dgrad.p, layer->p_dgrad.split, gx, spec.T(), AWR_GEMM_DGRAD);
            if (y->lz_g) { da->in_bnb_y = y->buf; da->in_bnb_coef = y->lz_lin4; }
            da->res = acc ? gx : nullptr;
            // remember who wrote d(x), in order: a full-coverage dgrad that is the LAST producer can host the fused BN-backward reduction
            P.grad_writers[x].push_back(dp.full ? da : nullptr);
            const std::string dname = "awr_conv_dgrad:" + layer->name;
#ifdef AWR_STUDY
            if (y->half_dy && res == nullptr) {      // d(y) arrives half by half (bn_bwd): part 0 now, part 1 behind the record of half B
                Op& d0 = b(dname, [da](void* s) { return awr_conv_gemm_part(da, 2, 0, s); });
                d0.gemm = true;
                d0.macs = 0.5 * layer_macs;
                Op& hw = b("__halfwait__", nullptr);
                hw.kind = OP_HALFWAIT;
                Op& d1 = b(dname + "/b", [da](void* s) { return awr_conv_gemm_part(da, 2, 1, s); });
                d1.gemm = true;
                d1.macs = 0.5 * layer_macs;
            } else
#endif
            {
                if (y->half_dy) {      // (cannot happen: half_ok excludes a fused residual) -- still correct: wait for half B first
                    Op& hw = b("__halfwait__", nullptr);
                    hw.kind = OP_HALFWAIT;
                }
                // Winograd data gradient (round 6): same eligibility as the forward; the launch decides from the COMPLETED argument block whether it
                // implements its epilogue (plain, accumulate, BatchNorm-backward reduction) and runs the direct kernel otherwise
                float* Ud = nullptr;
                if ((awr_get_conv_winograd() & 3) == 2 && !P.det && awr_get_gemm_products() == 1 && !spec.deconv && spec.k == 3 && spec.stride == 1 && spec.pad == 1 &&
                    dp.full && !y->lz_g && awr_wino_eligible(B, dp.Hin, dp.Win, spec.cout_pad, spec.cin_pad)) {
                    for (auto& wl : P.wino_d)
                        if (wl.first == layer) Ud = wl.second;
                    if (!Ud) {
                        Ud = alloc<float>((int64_t)16 * spec.cout_pad * spec.cin_pad);
                        P.wino_d.push_back({layer, Ud});
                    }
                    P.wino_dg.push_back({da, layer_macs});
                }
                Op& dop = Ud ? b(dname, [da, Ud](void* s) { return awr_wino_dgrad_or_direct(da, Ud, s); }) : b(dname, [da](void* s) { return awr_conv_gemm(da, s); });
                dop.gemm = true;
                dop.macs = layer_macs;
            }
            P.gemms.push_back({da, nullptr, dname});
        }
        if (res) contribute_identity(res, dy);
        return err;
    }

    // inference: per-channel (scale, shift) of an eval-mode BatchNorm, refreshed with the weights
    std::pair<const float*, const float*> fold_bn(BNLayer* bn) {
        float *sc = alloc<float>(bn->C), *sh = alloc<float>(bn->C);
        const int C = bn->C;
        const float *g = bn->gamma, *bt = bn->beta, *rm = bn->rmean, *rv = bn->rvar;
        push(P.pack_ops, "awr_bn_fold_eval", [=](void* s) { return awr_bn_fold_eval(C, g, bt, rm, rv, BN_EPS, sc, sh, s); });
        return {sc, sh};
    }

    float bn_momentum() const { return (float)(1.0 - pow(1.0 - (double)BN_MOMENTUM, (double)P.bn_repeat)); }

    // training-mode BatchNorm (+residual) (+ReLU): a = [relu](bn(y) [+ res]).  lazy (no residual): the normalised tensor is NOT
    // written; the returned handle carries (scale, shift, relu) for its consumers
    Tn* bn_act(Tn* y, BNLayer* bn, bool relu, Tn* res = nullptr, bool lazy = false) {
        const int B = y->B, H = y->H, W = y->W, C = y->C;
        const int64_t npix = y->npix();
        StatBuf own = y->stats;
        if (!own.p) {
            own = stat_buf(reduce_slots(), C);
            const float* xb = y->buf;
            double* sp = own.p;
            const int ns = own.nslots;
            if (y->fstats && !y->fstats->sp) {      // the max-pool / up-sampling add that wrote y accumulates them on the way (round 5): no pass of its own
                y->fstats->sp = sp;
                y->fstats->ns = ns;
            } else {
                f("awr_channel_stats", [=](void* s) { return awr_channel_stats(xb, npix, C, sp, ns, s); });
            }
        }
        // several BNs may normalise the same tensor (hourglass): finalize zeroes the accumulator, so every BN gets its own
        y->stats = StatBuf();
        float* coef4 = alloc<float>(4 * (int64_t)C);     // [scale | shift | mean | invstd][C]: one buffer so fused consumers take one pointer
        float *sc = coef4, *sh = coef4 + C, *mean = coef4 + 2 * C, *invstd = coef4 + 3 * C;
        const float mom = bn_momentum();
        {
            double* sp = own.p;
            const int ns = own.nslots;
            const float *g = bn->gamma, *bt = bn->beta;
            float *rm = bn->rmean, *rv = bn->rvar;
            f("awr_bn_finalize", [=](void* s) { return awr_bn_finalize(sp, C, npix, g, bt, rm, rv, mom, BN_EPS, sc, sh, mean, invstd, ns, s); });
        }
        Tn* a;
        if (lazy && !res) {
            P.tensors.emplace_back();
            a = &P.tensors.back();
            *a = Tn();
            a->buf = y->buf; a->B = B; a->H = H; a->W = W; a->C = C;
            a->name = bn->name + ".act(lazy)";
            a->lazy = true; a->lz_scale = sc; a->lz_shift = sh; a->lz_relu = relu;
        } else {
            a = new_t(B, H, W, C, true, bn->name + ".act");
            join_if(res);
            const float* xb = y->buf;
            const float* rb = res ? res->buf : nullptr;
            float* ob = a->buf;
            f("awr_bn_apply", [=](void* s) { return awr_bn_apply(xb, sc, sh, rb, relu ? 1 : 0, ob, npix, C, s); });
            if (!relu && !res) { a->bn_y = y->buf; a->bn_coef4 = coef4; }
        }
        P.nodes.push_back([=]() { return bn_bwd(y, a, bn, relu, res, mean, invstd, sc, sh, coef4); });
        return a;
    }

    int bn_bwd(Tn* y, Tn* a, BNLayer* bn, bool relu, Tn* res, float* mean, float* invstd, float* sc, float* sh, float* coef4) {
        float* da = a->grad;
        if (!da) {
            set_error("plan: no gradient reached %s", a->name.c_str());
            return AWR_ERR_ARG;
        }
        const int C = y->C;
        const int64_t npix = y->npix();
        float* coef = alloc<float>(3 * (int64_t)C);
        // Fused reduction: when the ONLY producer of d(a) is one full-coverage data-gradient GEMM (a BN+ReLU output read by a
        // single conv, materialised or not), that GEMM's epilogue masks with the re-derived ReLU and accumulates sum g /
        // sum g*xhat itself -- the separate reduction pass over d(a) and y disappears and the apply pass needs no mask.
        // With a residual added before the ReLU (ResNet block outputs: d(a) = the next block's skip gradient + its conv1 data gradient)
        // the LAST producer qualifies: it adds its tile onto the earlier contributions in place, takes the mask from the stored
        // activation, reduces, and leaves the MASKED gradient behind -- which is also d(res).
        auto& writers = P.grad_writers[a];
        awr_conv_args* ga = (!writers.empty() && relu) ? writers.back() : nullptr;
        if (ga && ga->res && (ga->res != ga->out || !res)) ga = nullptr;      // accumulating producers only for the residual form, in place
        if (ga && !ga->res && writers.size() != 1) ga = nullptr;
        const bool fused = ga != nullptr;
        StatBuf sums;
        if (fused) {
            sums = stat_buf(gemm_slots(ga->B, ga->Hq, ga->Wq, ga->N, ga->nphase), C);
            ga->bnr_y = y->buf;
            ga->bnr_coef = coef4;
            ga->bnr_act = res ? a->buf : nullptr;
            ga->stats = sums.p;
            ga->stat_slots = sums.nslots;
            // the residual is itself a BatchNorm output (no ReLU of its own: the downsample projection): its backward needs sum g and
            // sum g * xhat over the SAME masked gradient -- reduced here too, its stand-alone reduction pass disappears
            static const bool no_bnr2 = getenv("AWR_NO_BNR2") != nullptr;      // same-box A/B hook
            if (res && res->bn_y && res->needs_grad && res->C == C && !no_bnr2) {
                res->pre_sums = stat_buf(sums.nslots, C);
                ga->bnr2_y = res->bn_y;
                ga->bnr2_coef = res->bn_coef4;
                ga->stats2 = res->pre_sums.p;
            }
        } else if (a->pre_sums.p && !relu && !res) {
            sums = a->pre_sums;
        } else {
            sums = stat_buf(reduce_slots(), C);
        }
        // ReLU mask: without a residual the activation is re-derived from y (no read of `a`); with one it needs `a`; a fused producer
        // has applied it already
        const float* act = (relu && res && !fused) ? a->buf : nullptr;
        const float* msc = (relu && !res && !fused) ? sc : nullptr;
        const float* msh = (relu && !res && !fused) ? sh : nullptr;
        const float* yb = y->buf;
        double* sp = sums.p;
        const int ns = sums.nslots;
        const bool pre_reduced = !fused && a->pre_sums.p && !relu && !res;
        if (!fused && !pre_reduced) b("awr_bn_bwd_reduce", [=](void* s) { return awr_bn_bwd_reduce(da, act, yb, mean, invstd, msc, msh, npix, C, sp, ns, s); });
        float* gy;
        bool acc = false;
        if (y->needs_grad) {
            auto t = gtarget(y);
            gy = t.first;
            acc = t.second;
        } else {
            gy = alloc<float>(y->numel());
        }
        float *g_out = nullptr, *post_add = nullptr;
        if (res && res->needs_grad && relu && !fused) {      // the masked gradient is d(res): written out by the apply kernel
            P.grad_writers[res].push_back(nullptr);
            if (!res->grad) {
                res->grad = alloc<float>(res->numel());
                g_out = res->grad;
            } else {
                g_out = alloc<float>(res->numel());
                post_add = g_out;
            }
        }
        // Off the critical chain (round 4): when the masked gradient g is already in memory (fused reduction, no residual: nobody accumulates
        // into that buffer later) and y is a plain conv output, the data gradient of that conv evaluates the BatchNorm backward itself from
        // (g, y) and four coefficient vectors -- only the small finalize stays between the two dependent data-gradient GEMMs; the apply pass
        // that writes d(y) for the weight gradient is emitted by conv_bwd on the weight gradient's side stream.
        // Measured (profiles/r04_bn_bwd_lazy.txt) and NOT the default: the implicit GEMM re-stages every input element once per tap, so a 3x3
        // data gradient pays the arithmetic and the second tensor's traffic NINE times (ResNet18 13.28 -> 14.2 ms, Hourglass-1 24.1 -> 25.9 ms
        // with every eligible BatchNorm); restricted to single-tap (1x1) convolutions it is a wash (Hourglass-1 23.9 vs 24.1 ms).  The upper
        // bound -- the step with the apply passes simply left out -- is 12.76 / 21.7 ms, so the pass IS exposed, but the consumer is the
        // wrong place to hide it.  AWR_LAZY_BNB: 0 = never (default), 1 = 1x1 convolutions, 2 = every eligible BatchNorm.
        const int lazy_mode = env_or("AWR_LAZY_BNB", 0);
        const bool lazy_bnb = lazy_mode > 0 && fused && relu && !res && !acc && !g_out && y->conv_out && (lazy_mode > 1 || y->conv_taps == 1) && y->needs_grad &&
                              C <= 512 && awr_get_gemm_products() == 1 && awr_get_gemm_staging() == 2;
        {
            const float* gam = bn->gamma;
            float *gg = bn->ggamma, *gb = bn->gbeta;
            const float* dy_add = acc ? gy : nullptr;
            if (lazy_bnb) {
                float* lin4 = alloc<float>(4 * (int64_t)C);
                b("awr_bn_bwd_finalize", [=](void* s) { return awr_bn_bwd_finalize_lin(sp, C, npix, gam, mean, invstd, coef, lin4, gg, gb, 0, ns, s); });
                y->lz_g = da; y->lz_lin4 = lin4; y->lz_mean = mean; y->lz_invstd = invstd; y->lz_coef = coef;
            } else if (y->half_ok && y->needs_grad) {
                // Half-batch wavefront (round 5): the reduction needs the whole batch, the apply and the data gradient behind it do
                // not.  finalize -> apply(half A) on the issuing chain -> apply(half B) on the stream the layer's weight gradient is about to take
                // (it waits for the chain, i.e. starts when half A is done, and the weight gradient queues behind it: it needs both halves anyway)
                // -> conv_bwd issues dgrad(half A) beside it, waits for half B's record, then dgrad(half B).  The HBM-bound pass runs beside an
                // MFMA-bound one instead of alone; no extra stream (the runtime's four hardware queues are taken).
                b("awr_bn_bwd_finalize", [=](void* s) { return awr_bn_bwd_finalize(sp, C, npix, gam, invstd, coef, gg, gb, 0, ns, s); });
                const int64_t hp = npix / 2, off = hp * C;
                b("awr_bn_bwd_apply", [=](void* s) {
                    return awr_bn_bwd_apply_only(da, act, yb, mean, invstd, msc, msh, coef, hp, C, gy, dy_add, g_out, s);
                });
                Op& hb = b("awr_bn_bwd_apply", [=](void* s) {
                    return awr_bn_bwd_apply_only(da + off, act ? act + off : nullptr, yb + off, mean, invstd, msc, msh, coef, npix - hp, C, gy + off,
                                                 dy_add ? dy_add + off : nullptr, g_out ? g_out + off : nullptr, s);
                });
                hb.side_ok = true;
                hb.pair_next = true;
                hb.half_sig = true;
                y->half_dy = true;
            } else {
                b("awr_bn_bwd_apply", [=](void* s) {
                    return awr_bn_bwd_apply(da, act, yb, mean, invstd, gam, msc, msh, sp, coef, npix, C, gy, dy_add, g_out, gg, gb, 0, ns, s);
                });
            }
        }
        note_grad(bn->ggamma, C);
        note_grad(bn->gbeta, C);
        if (post_add) {
            float* rg = res->grad;
            const int64_t n = res->numel();
            b("awr_add", [=](void* s) { return awr_add(rg, post_add, rg, n, s); });
        }
        if (res && res->needs_grad && (!relu || fused)) contribute_identity(res, da);      // d(a) itself (already masked when fused) is d(res)
        return err;
    }

    Tn* maxpool(Tn* x, int k, int s_, int p) {
        const int B = x->B, H = x->H, W = x->W, C = x->C;
        const int Ho = (H + 2 * p - k) / s_ + 1, Wo = (W + 2 * p - k) / s_ + 1;
        Tn* y = new_t(B, Ho, Wo, C, true, x->name + ".pool");
        uint8_t* arg = P.training ? alloc<uint8_t>((int64_t)B * Ho * Wo * C) : nullptr;
        {
            const float* xb = x->buf;
            const float *ls = x->lazy ? x->lz_scale : nullptr, *lt = x->lazy ? x->lz_shift : nullptr;
            const int lr = x->lazy && x->lz_relu ? 1 : 0;
            float* ob = y->buf;
            FusedStats* fs = nullptr;
            if (P.training && env_or("AWR_FUSED_POOL_STATS", 1) && C % 4 == 0 && (C <= 1024 || C % 1024 == 0)) {
                P.fstats.emplace_back();
                fs = &P.fstats.back();
                y->fstats = fs;
            }
            // inference: a 2x2 / stride-2 pool of a fused pair's output is written by the pair itself (awr_conv_args.pool_out: 2D workgroup tiles, the
            // windows reduced in the epilogue) -- no pass that re-reads the full-resolution tensor.  AWR_PAIR_POOL=0: the separate pass (A/B hook)
            awr_conv_args* pa = x->pair_args;
            const int pbm = pa && pa->N1 == 64 && pa->tile_m == 2 ? 128 : 64;
            if (!P.training && pa && !pa->pool_out && k == 2 && s_ == 2 && p == 0 && !x->lazy && H % 2 == 0 && W % (pbm / 2) == 0 && env_or("AWR_PAIR_POOL", 1) &&
                awr_get_gemm_staging() != 0 && env_or("AWR_FUSE2_DMA", 1) && !pa->in_scale && !pa->relu_in) {
                pa->pool_out = ob;
            } else {
                f("awr_maxpool_fwd", [=](void* s) {
                    return fs && fs->sp ? awr_maxpool_fwd_stats(xb, ls, lt, lr, B, H, W, C, k, s_, p, ob, arg, fs->sp, fs->ns, s)
                                        : awr_maxpool_fwd(xb, ls, lt, lr, B, H, W, C, k, s_, p, ob, arg, s);
                });
            }
        }
        if (P.training) {
            P.nodes.push_back([=]() {
                if (!x->needs_grad) return (int)AWR_OK;
                auto t = gtarget(x);
                float* gx = t.first;
                const int acc = t.second ? 1 : 0;
                float* gyb = y->grad;
                b("awr_maxpool_bwd", [=](void* s) { return awr_maxpool_bwd(gyb, arg, B, H, W, C, k, s_, p, gx, acc, s); });
                return err;
            });
        }
        return y;
    }

    // out = up1 + nearest_upsample_x2(low)   (hourglass.py:77,:88)
    Tn* upsample_add(Tn* up1, Tn* low) {
        const int B = low->B, Hl = low->H, Wl = low->W, C = low->C;
        join_if(up1);
        Tn* y = new_t(B, 2 * Hl, 2 * Wl, C, true, up1->name + ".upadd");
        {
            const float *ub = up1->buf, *lb = low->buf;
            float* ob = y->buf;
            FusedStats* fs = nullptr;
            if (P.training && env_or("AWR_FUSED_POOL_STATS", 1) && C % 4 == 0 && (C <= 1024 || C % 1024 == 0)) {
                P.fstats.emplace_back();
                fs = &P.fstats.back();
                y->fstats = fs;
            }
            f("awr_upsample2_add", [=](void* s) {
                return fs && fs->sp ? awr_upsample2_add_stats(ub, lb, B, Hl, Wl, C, ob, fs->sp, fs->ns, s) : awr_upsample2_add(ub, lb, B, Hl, Wl, C, ob, s);
            });
        }
        if (P.training) {
            P.nodes.push_back([=]() {
                auto t = gtarget(low);
                float* gl = t.first;
                const int acc = t.second ? 1 : 0;
                float* gyb = y->grad;
                b("awr_upsample2_bwd", [=](void* s) { return awr_upsample2_bwd(gyb, B, Hl, Wl, C, gl, acc, s); });
                contribute_identity(up1, y->grad);
                return err;
            });
        }
        return y;
    }

    // NHWC (B,F,F,Cp) dense map -> the reference's NCHW (B,4J,F,F) tensor (+ gradient bridge)
    void head_out(Tn* pred, int J, float* out, float* gout) {
        const int B = pred->B, F = pred->H, Cp = pred->C;
        {
            const float* pb = pred->buf;
            f("awr_nhwc_to_nchw", [=](void* s) { return awr_nhwc_to_nchw(pb, B, F * F, Cp, 4 * J, out, s); }).boundary = true;
        }
        const int stage = (int)P.outputs.size();
        P.outputs.push_back(out);
        P.grad_outs.push_back(gout);
        P.head_preds.push_back(pred);
        P.head_grads.push_back(nullptr);
        if (P.training) {
            P.nodes.push_back([=]() {
                if (!(P.supervised & (1u << stage))) return (int)AWR_OK;   // no loss on this stage (hourglass: only the last stage, train.py:116-121)
                auto t = gtarget(pred);
                float* g = t.first;
                if (t.second) {
                    float* tmp = alloc<float>(pred->numel());
                    const int64_t n = pred->numel();
                    b("awr_nchw_to_nhwc", [=](void* s) { return awr_nchw_to_nhwc(gout, B, F * F, Cp, 4 * J, tmp, s); });
                    b("awr_add", [=](void* s) { return awr_add(g, tmp, g, n, s); });
                } else {
                    b("awr_nchw_to_nhwc", [=](void* s) { return awr_nchw_to_nhwc(gout, B, F * F, Cp, 4 * J, g, s); }).boundary = true;
                    P.head_grads[stage] = g;      // the only producer: a caller on the NHWC boundary writes d(pred) here itself
                }
                return err;
            });
        }
    }

    // Stems as the fused kernels of awr_stem.hip -- the un-normalised full-resolution conv output is never written, forward or
    // backward.  pool: ResNet (resnet_deconv.py:31-36, :118-121): conv 5x5 (1 -> 64, no bias) -> BatchNorm -> ReLU -> MaxPool(3,2,1),
    // the result is the pooled map.  !pool: hourglass (hourglass.py:112): conv 5x5 (bias) -> BatchNorm -> ReLU at full resolution.
    Tn* stem(float* img, ConvLayer* conv, BNLayer* bn, int H, int W, bool pool) {
        const int B = P.B;
        Tn* y = pool ? new_t(B, H / 2, W / 2, 64, true, conv->name + ".pool") : new_t(B, H, W, 64, true, conv->name + ".act");
        const float *w = conv->w, *bias = conv->bias;
        const std::string tag = ":" + conv->name;
        const int64_t npix = (int64_t)B * H * W;
        float* yb = y->buf;
        if (!P.training) {
            auto ss = fold_bn(bn);
            const float *sc = ss.first, *sh = ss.second;
            Op& o = pool ? f("awr_stem_pool" + tag, [=](void* s) { return awr_stem_pool(img, w, sc, sh, B, H, W, yb, nullptr, s); })
                         : f("awr_stem_conv" + tag, [=](void* s) { return awr_stem_conv(img, w, bias, sc, sh, 1, B, H, W, yb, s); });
            o.gemm = true;
            o.macs = (double)npix * 64 * 25;
            return y;
        }
        int ns_stats = AWR_STAT_SLOTS, ns_dw = AWR_STAT_SLOTS;
        if (P.det && awr_stem_slots(B, H, W, &ns_stats, &ns_dw)) err = AWR_ERR_ARG;
        StatBuf stats = stat_buf(ns_stats, 64);
        float* coef4 = alloc<float>(4 * 64);
        uint8_t* arg = pool ? alloc<uint8_t>((int64_t)B * (H / 2) * (W / 2) * 64) : nullptr;
        const float mom = bn_momentum();
        {
            double* sp = stats.p;
            Op& o = f("awr_stem_stats" + tag, [=](void* s) { return awr_stem_stats(img, w, bias, B, H, W, sp, ns_stats, s); });
            o.gemm = true;
            const float *g = bn->gamma, *bt = bn->beta;
            float *rm = bn->rmean, *rv = bn->rvar;
            f("awr_bn_finalize", [=](void* s) {
                return awr_bn_finalize(sp, 64, npix, g, bt, rm, rv, mom, BN_EPS, coef4, coef4 + 64, coef4 + 128, coef4 + 192, ns_stats, s);
            });
            Op& p = pool ? f("awr_stem_pool" + tag, [=](void* s) { return awr_stem_pool(img, w, coef4, coef4 + 64, B, H, W, yb, arg, s); })
                         : f("awr_stem_conv" + tag, [=](void* s) { return awr_stem_conv(img, w, bias, coef4, coef4 + 64, 1, B, H, W, yb, s); });
            p.gemm = true;
            p.macs = (double)npix * 64 * 25;     // algorithmic work: the conv once forward, its weight gradient once backward
        }
        P.nodes.push_back([=]() {
            if (!y->grad) {
                set_error("plan: no gradient reached the stem");
                return (int)AWR_ERR_ARG;
            }
            StatBuf sums = stat_buf(ns_stats, 64);
            float* coef = alloc<float>(3 * 64);
            float* slots = alloc<float>((int64_t)ns_dw * 64 * 26, true);
            float* dp = y->grad;
            double* sp = sums.p;
            Op& r = b("awr_stem_bwd_reduce" + tag, [=](void* s) { return awr_stem_bwd_reduce(img, w, bias, coef4, dp, arg, B, H, W, sp, ns_stats, s); });
            r.gemm = true;
            const float* gam = bn->gamma;
            float *gg = bn->ggamma, *gb = bn->gbeta;
            b("awr_bn_bwd_finalize", [=](void* s) { return awr_bn_bwd_finalize(sp, 64, npix, gam, coef4 + 192, coef, gg, gb, 0, ns_stats, s); });
            note_grad(bn->ggamma, 64);
            note_grad(bn->gbeta, 64);
            float *gw = conv->gw, *gbias = bias ? conv->gbias : nullptr;
            Op& wg = b("awr_stem_bwd_wgrad" + tag,
                       [=](void* s) { return awr_stem_bwd_wgrad(img, w, bias, coef4, coef, dp, arg, B, H, W, slots, gw, gbias, ns_dw, s); });
            wg.gemm = true;
            wg.macs = (double)npix * 64 * 25;
            note_grad(conv->gw, 64 * 25);
            if (gbias) note_grad(gbias, 64);
            return err;
        });
        return y;
    }
};

}  // namespace awrnet

namespace awrnet {

// Group gradient tensors into contiguous arena ranges that become final in backward order: the ranges tile [0, n_active) from
// the arena END downwards (the backward pass finishes the last layers first), ready_op non-decreasing, so bucket k can be
// all-reduced while the backward of the earlier layers still runs.
static std::vector<awr_plan::Bucket> plan_buckets(std::vector<GradWrite> ws, int64_t n_active, int n_buckets) {
    std::vector<awr_plan::Bucket> out;
    if (ws.empty()) return out;
    std::stable_sort(ws.begin(), ws.end(), [](const GradWrite& a, const GradWrite& b) { return a.lo > b.lo; });
    int64_t total = 0;
    for (auto& w : ws) total += w.hi - w.lo;
    // The LAST bucket (the arena's first layers: final only when the backward ends) has nothing left to hide its exchange behind, so it is
    // kept small -- the leading tensors up to 1 % of the parameters (ResNet18: the stem and layer1, 0.6 MB of 61.5 MB) -- and the others
    // share the rest equally.  (Equal quarters left 15 MB = a quarter of the all-reduce exposed after the last data-gradient GEMM.)
    int64_t tail = 0;
    if (n_buckets > 2)
        for (size_t i = ws.size(); i-- > 0;) {
            if (tail + (ws[i].hi - ws[i].lo) > total / 100) break;
            tail += ws[i].hi - ws[i].lo;
        }
    const double target = (double)(total - tail) / (double)(n_buckets > 1 ? n_buckets - (tail > 0 ? 1 : 0) : 1);
    int64_t acc = 0, hi_edge = n_active, left = total;
    int ready = -1;
    for (size_t i = 0; i < ws.size(); ++i) {
        acc += ws[i].hi - ws[i].lo;
        left -= ws[i].hi - ws[i].lo;
        if (ws[i].ready > ready) ready = ws[i].ready;
        const bool last = i + 1 == ws.size();
        const bool cut_tail = tail > 0 && left == tail;      // everything below is the small last bucket
        if (last || cut_tail || ((double)acc >= target && (int)out.size() < n_buckets - 1 - (tail > 0 ? 1 : 0))) {
            const int64_t edge = last ? 0 : ws[i].lo;
            out.push_back({edge, hi_edge, ready});
            hi_edge = edge;
            acc = 0;
        }
    }
    for (size_t k = 1; k < out.size(); ++k)      // a later bucket is never launched before an earlier one
        if (out[k].ready < out[k - 1].ready) out[k].ready = out[k - 1].ready;
    return out;
}

static int upload_table(awr_plan& P, const void* host, size_t bytes, void** dev) {
    void* p = nullptr;
    NET_CHECK(dev_alloc(P.owned, bytes, false, &p));
    HIP_TRY(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice));
    *dev = p;
    return AWR_OK;
}

static int make_unpack_op(awr_plan& P, const std::vector<int>& job_ids, Op& out) {
    std::vector<awr_unpack_job> jobs;
    int64_t total = 0;
    for (int id : job_ids) {
        awr_unpack_job j = P.unpack_jobs[id];
        j.first = total;                 // one workgroup per gradient row
        total += j.d0;
        jobs.push_back(j);
    }
    void* tab = nullptr;
    NET_CHECK(upload_table(P, jobs.data(), jobs.size() * sizeof(awr_unpack_job), &tab));
    const int n = (int)jobs.size();
    out = Op();
    out.name = "awr_unpack_wgrads_batched";
    out.fn = [=](void* s) { return awr_unpack_wgrads_batched((const awr_unpack_job*)tab, n, total, s); };
    return AWR_OK;
}

static int build_backward(Builder& bld) {
    awr_plan& P = bld.P;
    // split-K scratch for every weight gradient, as one arena (capacity = all packed gradients of the used layers)
    P.scratch_cap = 64;
    for (auto* l : P.layers)
        P.scratch_cap += round_up(!l->spec.deconv ? l->p_fwd.numel() : l->p_dgrad.numel(), 4) + AWR_STAT_SLOTS * round_up(l->spec.cout_pad, 4);
    const size_t first = P.bwd.size();
    bld.b("__zero__", nullptr).kind = OP_ZERO;       // placeholder for the scratch fill (kept in place so recorded op indices stay valid)
    for (size_t i = P.nodes.size(); i-- > 0;) NET_CHECK(P.nodes[i]());
    if (bld.err) return bld.err;
    if (P.scratch) {
        P.bwd[first].p = P.scratch;
        P.bwd[first].bytes = (size_t)P.scratch_used * 4;
    } else {
        P.bwd[first].p = bld.alloc<float>(4);
        P.bwd[first].bytes = 16;
    }
    const int64_t n_active = P.net->layout.n_active;
    if (P.n_buckets <= 1 || P.grad_writes.empty()) {
        if (!P.unpack_jobs.empty()) {       // ONE launch scatters every packed weight gradient back to checkpoint layout
            std::vector<int> ids(P.unpack_jobs.size());
            for (size_t i = 0; i < ids.size(); ++i) ids[i] = (int)i;
            Op op;
            NET_CHECK(make_unpack_op(P, ids, op));
            P.bwd.push_back(op);
        }
        P.buckets = {{0, n_active, (int)P.bwd.size() - 1}};
    } else {
        // data-parallel overlap: each bucket's weight gradients are scattered into the arena as soon as the backward has passed
        // its layers, then a marker lets the host start that bucket's all-reduce while the backward continues
        P.buckets = plan_buckets(P.grad_writes, n_active, P.n_buckets);
        struct Ins { int pos; std::vector<Op> ops; };
        std::vector<Ins> inserts;
        for (auto& bk : P.buckets) {
            std::vector<int> ids;
            for (auto& w : P.grad_writes)
                if (w.job >= 0 && bk.lo <= w.lo && w.lo < bk.hi) ids.push_back(w.job);
            Ins in;
            in.pos = bk.ready + 1;
            if (!ids.empty()) {
                Op op;
                NET_CHECK(make_unpack_op(P, ids, op));
                in.ops.push_back(op);
            }
            Op m;
            m.kind = OP_BUCKET;
            m.name = "__bucket__";
            m.lo = bk.lo;
            m.hi = bk.hi;
            in.ops.push_back(m);
            inserts.push_back(in);
        }
        std::stable_sort(inserts.begin(), inserts.end(), [](const Ins& a, const Ins& b) { return a.pos > b.pos; });
        for (auto& in : inserts) P.bwd.insert(P.bwd.begin() + in.pos, in.ops.begin(), in.ops.end());
    }
    P.built_bwd = true;
    return AWR_OK;
}

// ---- the two backbones (nets.py: ResNet18Deconv.build / HourglassNet.build) ----------------------------------
struct NetBuilder {
    Builder& b;
    awr_plan& P;
    awr_net& N;
    explicit NetBuilder(Builder& bb) : b(bb), P(bb.P), N(bb.N) {}
    ConvLayer* C(const std::string& k) { return &N.convs.at(k); }
    BNLayer* BN(const std::string& k) { return &N.bns.at(k); }

    // conv -> BatchNorm [-> +res] [-> ReLU]; fused into one GEMM epilogue in inference.  lazy (training): the BN+ReLU output is
    // not materialised -- legal when every consumer is a conv / max-pool loader
    Tn* cbr(Tn* x, const std::string& conv, const std::string& bn, bool relu, Tn* res = nullptr, bool lazy = false) {
        ConvLayer* cl = C(conv);
        ConvOpt o;
        o.use_bias = cl->bias_ptr() != nullptr;
        if (P.training) {
            o.want_stats = true;
            Tn* y = b.conv(x, cl, o);
            return b.bn_act(y, BN(bn), relu, res, lazy && !res);
        }
        auto ss = b.fold_bn(BN(bn));
        o.out_scale = ss.first; o.out_shift = ss.second; o.res = res; o.relu_out = relu;
        return b.conv(x, cl, o);
    }

    void resnet(float* img, int H, float* out, float* gout) {
        Tn* c = b.stem(img, C("pre.0"), BN("pre.1"), H, H, true);      // conv 5x5 -> BN -> ReLU -> MaxPool(3,2,1), one fused kernel family
        const bool bott = N.depth != 18;
        const int* nb = resnet_blocks(N.depth);
        for (int li = 1; li <= 4; ++li)
            for (int bi = 0; bi < nb[li - 1]; ++bi) {
                const std::string p = fmt("layer%d.%d", li, bi);
                Tn* r;
                const size_t n0 = P.nodes.size();
                size_t n1 = n0;
                if (N.convs.count(p + ".downsample.0")) {   // 1x1 projection (+ stride) + its BatchNorm: independent of the main branch until the residual add
                    b.fork(0);
                    r = cbr(c, p + ".downsample.0", p + ".downsample.1", false);
                    b.end_fork(r);
                    n1 = P.nodes.size();
                } else {
                    r = c;
                }
                // bn1 (+ bn2 of a Bottleneck) + ReLU feed ONE conv only: un-materialised (applied by that conv's loaders) while that is
                // cheaper than one write + read of the tensor -- the loader arithmetic costs 8-15 % of a GEMM whose K grows with the
                // channel count, the tensor pass does not: beyond 128 channels the activation is written out
                Tn* o = cbr(c, p + ".conv1", p + ".bn1", true, nullptr, C(p + ".conv1")->spec.cout <= 128);
                const size_t n2 = P.nodes.size();
                if (!bott) {
                    c = cbr(o, p + ".conv2", p + ".bn2", true, r);
                } else {
                    o = cbr(o, p + ".conv2", p + ".bn2", true, nullptr, C(p + ".conv2")->spec.cout <= 128);
                    c = cbr(o, p + ".conv3", p + ".bn3", true, r);
                }
                // Backward of a block with a projection (the emitters run in reverse node order).  Two things are arranged here:
                // (i) the projection's data gradient covers only one of the four stride phases of d(c) (zero fill + accumulate), conv1's
                // covers all of it: with the projection's backward emitted FIRST, conv1's data gradient is the LAST producer of d(c) -- a
                // full-coverage GEMM that adds in place and can host the fused BatchNorm-backward reduction of the previous block's
                // output (with the projection's own BatchNorm reduced in this block's bn-output GEMM, bn_bwd: no stand-alone reduction
                // pass is left in a ResNet step);  (ii) the projection's backward (BatchNorm apply, a small 1x1 weight and data gradient)
                // only needs the masked d(out) and only meets the main chain again at conv1's data gradient: it runs on a branch stream
                // beside the block's conv backward.  Emission: bn_last, FORK, projection, ENDFORK, ..., bn1, JOIN, conv1.
                // The forward launch order is untouched.
                static const bool no_reorder = getenv("AWR_NO_DS_REORDER") != nullptr;      // same-box A/B hooks
                static const bool no_branch = getenv("AWR_NO_DS_BRANCH") != nullptr;
                if (P.training && n1 > n0 && !no_reorder) {
                    std::vector<std::function<int()>> ds(P.nodes.begin() + n0, P.nodes.begin() + n1), mn(P.nodes.begin() + n1, P.nodes.end());
                    Builder* bp = &b;
                    auto marker = [bp](int kind) {
                        return std::function<int()>([bp, kind]() {
                            Op& o = bp->b(kind == OP_FORK ? "__fork__" : kind == OP_ENDFORK ? "__endfork__" : "__join__", nullptr);
                            o.kind = kind;
                            o.sid = 0;
                            return bp->err;
                        });
                    };
                    std::vector<std::function<int()>> order;
                    order.push_back(mn.front());                                   // conv1 (its data gradient: last producer of d(c))
                    if (!no_branch) order.push_back(marker(OP_JOIN));
                    for (size_t i = 1; i + 1 < mn.size(); ++i) order.push_back(mn[i]);
                    if (!no_branch) order.push_back(marker(OP_ENDFORK));
                    for (auto& f : ds) order.push_back(f);
                    if (!no_branch) order.push_back(marker(OP_FORK));
                    order.push_back(mn.back());                                    // the block output's BatchNorm
                    P.nodes.erase(P.nodes.begin() + n0, P.nodes.end());
                    for (auto& f : order) P.nodes.push_back(f);
                }
                (void)n2;
            }
        for (int i = 0; i < N.ndeconv; ++i)   // feeds the next transposed conv (K = 4 x 256 per phase: materialised) or the 1x1 head GEMM (lazy)
            c = cbr(c, fmt("deconv_layers.%d", 3 * i), fmt("deconv_layers.%d", 3 * i + 1), true, nullptr, i == N.ndeconv - 1);
        Tn* pred = b.conv(c, C("head"));
        b.head_out(pred, N.J, out, gout);
    }

    DualLayer* dual_for(const std::string& p) {
        // conv3 + skip_layer as one launch: FP32-MFMA mode only (the split-operand kernels take one input tensor)
        static const bool no_dual = getenv("AWR_NO_DUAL") != nullptr;      // bisecting hook (tools/diag_parity.py)
        if (no_dual || awr_get_gemm_products() != 1 || !N.convs.count(p + ".skip_layer")) return nullptr;
        DualLayer& d = N.duals[p];
        if (!d.c3) {
            d.c3 = C(p + ".conv3");
            d.sk = C(p + ".skip_layer");
            d.name = p + ".conv3+skip_layer";
        }
        return &d;
    }

    Tn* residual(Tn* x, const std::string& p) {
        ConvLayer* skip = N.convs.count(p + ".skip_layer") ? C(p + ".skip_layer") : nullptr;
        DualLayer* dual = dual_for(p);
        ConvOpt o;
        if (P.training) {
            // the three pre-activations feed exactly one conv each: never written to HBM
            Tn* a = b.bn_act(x, BN(p + ".bn1"), true, nullptr, true);
            o.want_stats = true;
            // (bn2's output feeds the 3x3 conv2, whose nine taps each re-apply the loader arithmetic; writing it out instead puts conv2's
            // forward on the pure-DMA GEMM and its weight gradient on the kernel-row kernel -- re-measured in round 4 with those kernels:
            // Hourglass-1 23.19-23.34 vs 23.12-23.23 ms, config 5 281.1 vs 279.4 ms and +10.7 GB of plan: still not worth the pass.
            // AWR_HG_WRITE_BN2=1 is the A/B hook; profiles/r04_hg_materialised_bn2.txt)
            static const bool write_bn2 = getenv("AWR_HG_WRITE_BN2") != nullptr;
            a = b.bn_act(b.conv(a, C(p + ".conv1"), o), BN(p + ".bn2"), true, nullptr, !write_bn2);
            a = b.bn_act(b.conv(a, C(p + ".conv2"), o), BN(p + ".bn3"), true, nullptr, true);
            if (dual) return b.conv_dual(a, x, dual, true);
            Tn* r = skip ? b.conv(x, skip) : x;
            o.res = r;
            return b.conv(a, C(p + ".conv3"), o);
        }
        // inference: bn1 stays a loader affine (x also feeds the skip path un-normalised); bn2 / bn3 + ReLU normalise tensors that only
        // conv2 / conv3 read, so they fold into the EPILOGUE of the conv that produces them
        auto s1 = b.fold_bn(BN(p + ".bn1")), s2 = b.fold_bn(BN(p + ".bn2")), s3 = b.fold_bn(BN(p + ".bn3"));
        o.in_scale = s1.first; o.in_shift = s1.second; o.relu_in = true;
        o.out_scale = s2.first; o.out_shift = s2.second; o.relu_out = true;
        Tn* y = b.conv(x, C(p + ".conv1"), o);
        ConvOpt o2;
        o2.out_scale = s3.first; o2.out_shift = s3.second; o2.relu_out = true;
        // conv2 and conv3 (+ the skip conv, as extra K of the second GEMM) in one launch when the batch fills the chip -- unless the Winograd mode takes
        // conv2 (2.25x fewer multiplies beat the saved round trip of its output: profiles/r06_winograd.txt)
        if (b.wino_fwd_ok(C(p + ".conv2")->spec, y->B, y->H, y->W)) {
        } else if (dual) {
            if (Tn* fused = b.conv_pair(y, C(p + ".conv2"), o2, nullptr, nullptr, dual, x)) return fused;
        } else if (!skip) {
            if (Tn* fused = b.conv_pair(y, C(p + ".conv2"), o2, C(p + ".conv3"), x)) return fused;
        }
        y = b.conv(y, C(p + ".conv2"), o2);
        if (dual) return b.conv_dual(y, x, dual, false);
        Tn* r = skip ? b.conv(x, skip) : x;
        ConvOpt o3;
        o3.res = r;
        return b.conv(y, C(p + ".conv3"), o3);
    }

    Tn* hg(Tn* x, const std::string& p, int depth) {
        // the skip branch of a level only meets the low-resolution path again at the up-sampling add: issued on its own side
        // stream, its full-resolution GEMMs fill the chip while the main stream walks the small (<= 16x16) levels
        const size_t n0 = P.nodes.size();
        b.fork(depth);
        Tn* up1 = residual(x, p + ".up1");
        b.end_fork(up1);
        const size_t n1 = P.nodes.size();
        Tn* pooled = b.maxpool(x, 2, 2, 0);
        // backward branches for the two outer levels only (64x64 / 32x32 skip residuals): one stream each, no false dependencies
        // (measured: branches for the inner levels too change nothing, 25.7 ms either way vs 26.8-27.3 without any)
        const bool bwd_branch = P.training && depth >= 3;
        if (bwd_branch) b.bwd_marker_node(OP_JOIN, depth);      // emitted right before maxpool's backward: d(x) += ... waits for the branch
        Tn* low = residual(pooled, p + ".low1");
        low = depth > 1 ? hg(low, p + ".low2", depth - 1) : residual(low, p + ".low2");
        low = residual(low, p + ".low3");
        if (bwd_branch) {
            // Backward of the level: d(out) feeds the up1 residual (large full-resolution GEMMs) and, through the up-sampling, the
            // whole low-resolution path (hundreds of small launches): independent until both add into d(x).  The up1 emitters are
            // moved behind the low path's, i.e. emitted FIRST and bracketed by fork / end-fork markers: up1's backward is the
            // first writer of d(x) on a branch stream, the low path runs beside it on the main stream, and the join above orders
            // the max-pool backward's accumulation into d(x) after the branch.  (d(out) itself is read by the up-sampling backward
            // BEFORE the fork and may be accumulated into in place by the branch afterwards: the identity skip aliases it.)
            std::rotate(P.nodes.begin() + n0, P.nodes.begin() + n1, P.nodes.end());
            const size_t nlow = P.nodes.size() - (n1 - n0);       // up1's emitters now start here
            P.nodes.insert(P.nodes.begin() + nlow, nullptr);      // placeholder: end-fork marker (emitted after up1's backward)
            const int sid = depth;
            Builder* bp = &b;
            P.nodes[nlow] = [bp, sid]() {
                Op& o = bp->b("__endfork__", nullptr);
                o.kind = OP_ENDFORK;
                o.sid = sid;
                return bp->err;
            };
            b.bwd_marker_node(OP_FORK, depth);                    // emitted right after the up-sampling add's backward
        }
        return b.upsample_add(up1, low);
    }

    void hourglass(float* img, int H, float* const* outs, float* const* gouts) {
        Tn* c = b.stem(img, C("pre.0"), BN("pre.0.bn"), H, H, false);      // conv 5x5 + bias -> BN -> ReLU at full resolution
        c = residual(c, "pre.1");
        c = b.maxpool(c, 2, 2, 0);
        c = residual(c, "pre.3");
        c = residual(c, "pre.4");
        for (int i = 0; i < N.nstack; ++i) {
            Tn* h = hg(c, fmt("hgs.%d.0", i), 4);
            Tn* ft = residual(h, fmt("features.%d.0", i));
            ft = cbr(ft, fmt("features.%d.1", i), fmt("features.%d.1.bn", i), true, nullptr, true);      // head / merge GEMMs
            Tn* pred = b.conv(ft, C(fmt("head.%d", i)));
            b.head_out(pred, N.J, outs[i], gouts ? gouts[i] : nullptr);
            if (i < N.nstack - 1) {
                ConvOpt o;
                o.res = c;
                Tn* m = b.conv(pred, C(fmt("merge_preds.%d.conv", i)), o);
                ConvOpt o2;
                o2.res = m;
                o2.want_stats = true;
                c = b.conv(ft, C(fmt("merge_features.%d.conv", i)), o2);
            }
        }
    }
};

}  // namespace awrnet

namespace awrnet {

// ---- replay ----------------------------------------------------------------------------------------------------
// Re-pack every conv weight (and re-fold eval BNs) from the parameter arena: one batched launch for the plain layers, a few
// extra calls for the fused heads.
static int refresh_weights(awr_plan& P, void* stream) {
    const int ws = awr_get_gemm_products() != 1 ? 1 : 0;      // split images are only written for the mode that reads them
    if (!P.pack_built[ws]) {
        std::vector<awr_pack_job> jobs[2];
        int64_t total[2] = {0, 0};
        for (auto* l : P.layers) {
            if (l->head) continue;
            const PackRecipe rc[2] = {fwd_pack(l->spec), dgrad_pack(l->spec)};
            Packed* dst[2] = {&l->p_fwd, &l->p_dgrad};
            for (int k = 0; k < 2; ++k) {
                if (!dst[k]->p) continue;
                awr_pack_job j;
                memset(&j, 0, sizeof j);
                j.src = l->w; j.dst = dst[k]->p; j.split = ws ? dst[k]->split : nullptr;
                j.d0 = rc[k].d0; j.d1 = rc[k].d1; j.T = rc[k].T; j.transpose = rc[k].transpose; j.rows = rc[k].rows; j.ld = rc[k].ld;
                j.first = total[k];
                total[k] += rc[k].rows;
                jobs[k].push_back(j);
            }
        }
        for (auto* d : P.dual_layers) {       // [W3 row | Wskip row] side by side in one K-contiguous buffer (FP32 mode only: no split image)
            const int cin1 = d->c3->spec.cin_pad, cin2 = d->sk->spec.cin_pad;
            const ConvLayer* src[2] = {d->c3, d->sk};
            const int colsv[2] = {cin1, cin2}, offs[2] = {0, cin1};
            for (int k = 0; k < 2; ++k) {
                awr_pack_job j;
                memset(&j, 0, sizeof j);
                j.src = src[k]->w; j.dst = d->p.p + offs[k]; j.split = nullptr;
                j.d0 = src[k]->spec.cout; j.d1 = src[k]->spec.cin; j.T = 1; j.transpose = 0; j.rows = d->p.rows; j.ld = d->p.ld; j.cols = colsv[k];
                j.first = total[0];
                total[0] += d->p.rows;
                jobs[0].push_back(j);
            }
        }
        for (int k = 0; k < 2; ++k) {
            if (!jobs[k].empty()) NET_CHECK(upload_table(P, jobs[k].data(), jobs[k].size() * sizeof(awr_pack_job), &P.pack_tab[ws][k]));
            P.pack_njobs[ws][k] = (int)jobs[k].size();
            P.pack_rows[ws][k] = total[k];
        }
        P.pack_built[ws] = true;
    }
    if (P.pack_njobs[ws][0])
        NET_CHECK(awr_pack_weights_batched((const awr_pack_job*)P.pack_tab[ws][0], P.pack_njobs[ws][0], P.pack_rows[ws][0], stream));
    // (packing the data-gradient layouts on a side stream under the forward's GEMMs was measured in rounds 2 and 3: no gain, 14.04 vs
    // 14.04 ms -- profiles/r03_summary.md -- so both tables go to the caller's stream)
    if (P.pack_njobs[ws][1])
        NET_CHECK(awr_pack_weights_batched((const awr_pack_job*)P.pack_tab[ws][1], P.pack_njobs[ws][1], P.pack_rows[ws][1], stream));
    for (auto& wl : P.wino)
        NET_CHECK(awr_wino_weights(wl.first->w, wl.first->spec.cout, wl.first->spec.cin, wl.first->spec.cout_pad, wl.first->spec.cin_pad, 0, wl.second, stream));
    for (auto& wl : P.wino_d)        // data-gradient form: output channels = the layer's Cin, contraction over its Cout, taps mirrored
        NET_CHECK(awr_wino_weights(wl.first->w, wl.first->spec.cin, wl.first->spec.cout, wl.first->spec.cin_pad, wl.first->spec.cout_pad, 1, wl.second, stream));
    hipStream_t st = awr::as_stream(stream);
    for (auto* l : P.layers) {
        if (!l->head) continue;
        const int J = l->J, cin = l->spec.cin, rows = l->p_fwd.rows;
        NET_CHECK(awr_pack_weight(l->w1, 3 * J, cin, 1, 0, 3 * J, cin, l->p_fwd.p, stream));
        NET_CHECK(awr_pack_weight(l->w2, J, cin, 1, 0, rows - 3 * J, cin, l->p_fwd.p + (int64_t)3 * J * cin, stream));
        if (l->p_dgrad.p)    // P[cin][1][cp]: columns [0,3J) from w1, [3J,4J) from w2 = the transpose of the forward pack
            NET_CHECK(awr_pack_weight(l->p_fwd.p, l->cp, cin, 1, 1, l->p_dgrad.rows, l->cp, l->p_dgrad.p, stream));
        if (ws) {
            NET_CHECK(awr_split_weight(l->p_fwd.p, l->p_fwd.split, l->p_fwd.numel(), stream));
            if (l->p_dgrad.p) NET_CHECK(awr_split_weight(l->p_dgrad.p, l->p_dgrad.split, l->p_dgrad.numel(), stream));
        }
        HIP_TRY(hipMemcpyAsync(l->bias_cat, l->b1, (size_t)3 * J * 4, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(l->bias_cat + 3 * J, l->b2, (size_t)J * 4, hipMemcpyDeviceToDevice, st));
    }
    for (auto* d : P.dual_layers) NET_CHECK(awr_add(d->c3->bias, d->sk->bias, d->bias_sum, d->c3->spec.cout, stream));
    for (auto& op : P.pack_ops) NET_CHECK(op.fn(stream));
    return AWR_OK;
}

// waiter waits for everything enqueued on signaler so far (capturable: event record + stream wait)
static int stream_wait(awr_plan& P, hipStream_t waiter, hipStream_t signaler) {
    if (P.events.size() < 2048) {      // (large enough that an event is never re-recorded within one pass: hundreds of waits per Hourglass backward)
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        P.events.push_back(e);
    }
    hipEvent_t e = P.events[P.ev_next++ % P.events.size()];
    HIP_TRY(hipEventRecord(e, signaler));
    HIP_TRY(hipStreamWaitEvent(waiter, e, 0));
    return AWR_OK;
}

// Replays one op list on `stream`.  Backward: weight-gradient GEMMs that qualify are issued round-robin on the side streams and
// run concurrently with the data-gradient chain (two DIFFERENT kernels co-resident on a CU are never in lock-step: each fills
// the other's prologue / epilogue / barrier bubbles); with a bucket callback, a bucket's scatter and its collective go to the
// comm stream so the main stream never waits for the side streams in the middle of the backward.  Forward: fork / join
// pseudo-ops route independent branches to the side streams.
static int run_list(awr_plan& P, std::vector<Op>& ops, void* stream, bool is_bwd) {
    hipStream_t main = awr::as_stream(stream);
    const bool use_side = !P.side.empty();
    const bool wside = is_bwd && use_side;
    // (no callback needed: a single-GPU plan built with several buckets uses the hand-off just to scatter its weight gradients early,
    // off the tail of the step)
    hipStream_t comm = (wside && P.n_buckets > 1) ? P.comm : nullptr;
    bool pending = false, handed = false;
    size_t nside = 0;
    void* cur = stream;
    bool after_join = false;
    hipEvent_t half_ev = nullptr;      // "half B of the last split BatchNorm-backward apply is written" (OP_HALFWAIT)
    auto hand_off = [&]() -> int {       // everything the bucket needs (main chain so far + weight gradients) -> comm stream
        NET_CHECK(stream_wait(P, comm, main));
        for (auto st : P.side)
            if (st != comm) NET_CHECK(stream_wait(P, comm, st));
        for (auto st : P.branch) NET_CHECK(stream_wait(P, comm, st));
        return AWR_OK;
    };
    for (auto& op : ops) {
        if (op.boundary && P.nhwc_boundary) continue;
        if (comm && op.kind == OP_CALL && op.name == "awr_unpack_wgrads_batched") {
            NET_CHECK(hand_off());
            handed = true;
            NET_CHECK(op.fn((void*)comm));
            continue;
        }
        if (comm && op.kind == OP_BUCKET) {
            if (!handed) NET_CHECK(hand_off());
            handed = false;
            if (P.bucket_cb) P.bucket_cb(P.bucket_user, op.lo, op.hi, (void*)comm);
            continue;
        }
        if (wside && pending && (op.kind == OP_CALL || op.kind == OP_BUCKET)) {
            const std::string& n = op.name;
            const bool no_join = op.kind == OP_CALL &&
                                 (n.compare(0, 9, "awr_conv_") == 0 || n.compare(0, 9, "awr_stem_") == 0 || n == "awr_bn_bwd_reduce" || n == "awr_bn_bwd_apply" ||
                                  n == "awr_bn_bwd_finalize" || n == "awr_maxpool_bwd" || n == "awr_upsample2_bwd" || n == "awr_add");
            if (!no_join) {      // join before anything that consumes the weight-gradient scratch (scatter, buckets)
                for (auto st : P.side) NET_CHECK(stream_wait(P, main, st));
                for (auto st : P.branch) NET_CHECK(stream_wait(P, main, st));
                pending = false;
                after_join = true;      // ... and issue it on MAIN: inside an open fork region `cur` is a branch stream that did not wait
            }
        }
        switch (op.kind) {
            case OP_ZERO:      // (on the issuing chain: a zero fill inside a branch belongs to the branch's data gradient)
                HIP_TRY(hipMemsetAsync(op.p, 0, op.bytes, awr::as_stream(cur)));
                continue;
            case OP_COPY:
                HIP_TRY(hipMemcpyAsync(op.p, op.q, op.bytes, hipMemcpyDeviceToDevice, awr::as_stream(cur)));
                continue;
            case OP_BUCKET:
                if (P.bucket_cb) P.bucket_cb(P.bucket_user, op.lo, op.hi, stream);
                continue;
            case OP_FORK:
                if (use_side) {
                    hipStream_t st = is_bwd ? P.branch[op.sid % P.branch.size()] : P.side[op.sid % P.side.size()];
                    NET_CHECK(stream_wait(P, st, main));
                    cur = (void*)st;
                    if (is_bwd) pending = true;      // the end-of-list join has to cover the branch streams
                }
                continue;
            case OP_ENDFORK:
                cur = stream;
                continue;
            case OP_JOIN:
                if (use_side) NET_CHECK(stream_wait(P, main, is_bwd ? P.branch[op.sid % P.branch.size()] : P.side[op.sid % P.side.size()]));
                continue;
            case OP_HALFWAIT:      // the second half of a BatchNorm-backward apply went to a side stream: its record, not the stream's tail
                if (half_ev) {
                    HIP_TRY(hipStreamWaitEvent(awr::as_stream(cur), half_ev, 0));
                    half_ev = nullptr;
                }
                continue;
            default:
                break;
        }
        int rc;
        if (wside && op.side_ok) {
            hipStream_t st = P.side[nside % P.side.size()];
            if (!op.pair_next) ++nside;
            // (issued BESIDE its layer's data gradient; behind it -- i.e. beside the memory-bound BatchNorm-backward passes that follow -- was re-measured in
            // round 4 with the LDS-DMA kernels: ResNet18 12.8 -> 13.9-14.0 ms, Hourglass-1 23.5 -> 23.9 ms, profiles/r04_wgrad_late.txt)
            NET_CHECK(stream_wait(P, st, awr::as_stream(cur)));      // its operands (dY, x) are final at this point of the issuing chain
            rc = op.fn((void*)st);
            pending = true;
            if (op.half_sig && rc == AWR_OK) {
                if (P.events.size() < 2048) {
                    hipEvent_t e;
                    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    P.events.push_back(e);
                }
                half_ev = P.events[P.ev_next++ % P.events.size()];
                HIP_TRY(hipEventRecord(half_ev, st));
            }
        } else {
            rc = op.fn(after_join ? stream : cur);
        }
        after_join = false;
        if (rc) return rc;
    }
    if (wside && (pending || comm)) {
        for (auto st : P.side) NET_CHECK(stream_wait(P, main, st));
        for (auto st : P.branch) NET_CHECK(stream_wait(P, main, st));
        if (comm) NET_CHECK(stream_wait(P, main, comm));      // next step's scratch fill / optimiser must see the scatters
    }
    if (is_bwd && P.dp && !P.dp_error && awr_dp_generation(P.dp) != P.dp_gen) {
        P.dp = nullptr;
        P.dp_lost = true;
    }
    if (is_bwd && P.dp_lost) {      // sticky: replicas that skipped an exchange have diverged -- no later backward may report success
        P.dp_error = 0;
        set_error("plan: the data-parallel communicator attached with awr_plan_set_dp has been destroyed (detach it with awr_plan_set_dp(plan, NULL) first); "
                  "gradients were NOT exchanged -- attach a communicator (or NULL) with awr_plan_set_dp to continue");
        return AWR_ERR_ARG;
    }
    if (is_bwd && P.dp && P.n_buckets <= 1 && !P.dp_error)        // a one-bucket plan has no markers: one exchange after the backward
        P.dp_error = awr_dp_allreduce(P.dp, P.net->grads, P.net->layout.n_active, stream);
    if (is_bwd && P.dp) NET_CHECK(awr_dp_wait(P.dp, stream));      // ... and the reduced gradients
    if (is_bwd && P.dp_error) {
        const int e = P.dp_error;
        P.dp_error = 0;
        return e;
    }
    return AWR_OK;
}

// serial replay with a HIP-event pair around every conv / stem launch: ms[i] per op of the list (0 for the others)
static int run_timed(awr_plan& P, std::vector<Op>& ops, void* stream, float* ms) {
    hipStream_t main = awr::as_stream(stream);
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev(ops.size(), {nullptr, nullptr});
    int rc = AWR_OK;
    for (size_t i = 0; i < ops.size() && rc == AWR_OK; ++i) {
        Op& op = ops[i];
        if (op.kind == OP_ZERO) { if (hipMemsetAsync(op.p, 0, op.bytes, main) != hipSuccess) rc = AWR_ERR_HIP; continue; }
        if (op.kind == OP_COPY) { if (hipMemcpyAsync(op.p, op.q, op.bytes, hipMemcpyDeviceToDevice, main) != hipSuccess) rc = AWR_ERR_HIP; continue; }
        if (op.kind != OP_CALL || (op.boundary && P.nhwc_boundary)) continue;
        // (every launch gets its event pair; callers that only want the GEMM family filter by the op flags)
        (void)hipEventCreate(&ev[i].first);
        (void)hipEventCreate(&ev[i].second);
        (void)hipEventRecord(ev[i].first, main);
        rc = op.fn(stream);
        (void)hipEventRecord(ev[i].second, main);
    }
    if (hipStreamSynchronize(main) != hipSuccess && rc == AWR_OK) {
        set_error("run_timed: hipStreamSynchronize failed");
        rc = AWR_ERR_HIP;
    }
    for (size_t i = 0; i < ops.size(); ++i) {
        ms[i] = 0.f;
        if (ev[i].first) {
            if (rc == AWR_OK) (void)hipEventElapsedTime(&ms[i], ev[i].first, ev[i].second);
            (void)hipEventDestroy(ev[i].first);
            (void)hipEventDestroy(ev[i].second);
        }
    }
    return rc;
}

// Pick the fastest workgroup tile (and split-K depth) for every GEMM launch of this static plan by timing the candidates in
// place with HIP events.  Runs right after a first replay (buffers hold real data; the next step rebuilds whatever the tuner
// scribbles on) and re-zeroes every atomic accumulator the launches touched.
static int autotune(awr_plan& P, int reps, void* stream) {
    if (P.det) return AWR_OK;      // a timing-dependent tile / split-K choice would change summation orders from run to run
    hipStream_t main = awr::as_stream(stream);
    NET_CHECK(refresh_weights(P, stream));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    auto launch = [&](GemmRef& g) { return g.ca ? awr_conv_gemm(g.ca, stream) : awr_conv_wgrad(g.wa, stream); };
    int rc = AWR_OK;
    for (auto& g : P.gemms) {
        if (g.ca && g.ca->w2) continue;             // the fused conv pair has one geometry
        struct Cand { int tm, tn, tb, algo = 0; };
        std::vector<Cand> cands;
        if (g.wa && g.wa->algo == 2) {
            // The wave-per-tap kernel has one geometry and stays.  AWR_TUNE_TAPS=1 (study hook) times it against the kernel-row kernel's split-K
            // depths: isolated, the row kernel with its in-LDS BatchNorm loader wins on these layers (127 vs 122.5 TF) and the tuner takes it --
            // and the Hourglass-1 step gets 0.3 ms SLOWER (23.6 -> 23.9 ms, three interleaved repetitions, profiles/r04_loop_exits.txt): a
            // weight gradient is a side-stream kernel, what counts is how it shares the CUs with the data-gradient chain it runs beside, and
            // 256 twelve-wave workgroups share better than 768-1280 four-wave ones that can fill every SIMD's register file.
            static const bool tune_taps = getenv("AWR_TUNE_TAPS") != nullptr;
            if (!tune_taps || !awr_conv_wgrad_algo_ok(g.wa, 3)) continue;
            cands = {{1, 1, 0, 2}};
            for (int tb : {768, 1024, 1280, 1536, 2048}) cands.push_back({1, 1, tb, 3});
        } else if (g.ca) {
            cands = {{1, 1, 0}, {2, 1, 0}};
            if (g.ca->N > 64) { cands.push_back({1, 2, 0}); cands.push_back({2, 2, 0}); }
            if (g.ca->partial && g.ca->split_max > 1 && awr_get_gemm_products() == 1) {      // (tile, split-K depth) pairs; tb = depth
                std::vector<Cand> withk;
                for (auto& c : cands)
                    for (int sk = 1; sk <= g.ca->split_max; sk *= 2) withk.push_back({c.tm, c.tn, sk});
                cands.swap(withk);
            }
        } else {
            cands = {{1, 1, 2048}, {1, 1, 3072}, {1, 1, 4096}};
            if (g.wa->Cd > 64) { cands.push_back({2, 1, 1536}); cands.push_back({2, 1, 2048}); }
            if (g.wa->Cg > 64) cands.push_back({1, 2, 2048});
            if (g.wa->Cd > 64 && g.wa->Cg > 64 && awr_get_wgrad_products() != 1) { cands.push_back({2, 2, 1024}); cands.push_back({2, 2, 2048}); }
            // one workgroup per kernel row (3x3 stride 1): its own split-K depths; the per-tap candidates above then run as algo 1
            if ((g.wa->algo == 0 || g.wa->algo == 3) && awr_conv_wgrad_algo_ok(g.wa, 3)) {
                for (auto& c : cands) c.algo = 1;
                for (int tb : {768, 1024, 1280, 1536, 2048}) cands.push_back({1, 1, tb, 3});
            }
        }
        float best_t = 1e30f;
        Cand best = cands[0];
        for (auto& c : cands) {
            if (g.ca) { g.ca->tile_m = c.tm; g.ca->tile_n = c.tn; if (c.tb) g.ca->split_k = c.tb; }
            else { g.wa->tile_m = c.tm; g.wa->tile_n = c.tn; g.wa->target_blocks = c.tb; if (c.algo) g.wa->algo = c.algo; }
            if ((rc = launch(g))) break;      // warm-up
            (void)hipEventRecord(e0, main);
            for (int r = 0; r < reps && rc == AWR_OK; ++r) rc = launch(g);
            (void)hipEventRecord(e1, main);
            (void)hipEventSynchronize(e1);
            float t = 0.f;
            (void)hipEventElapsedTime(&t, e0, e1);
            t /= (float)reps;
            if (t < best_t) { best_t = t; best = c; }
        }
        if (rc) break;
        if (g.ca) { g.ca->tile_m = best.tm; g.ca->tile_n = best.tn; if (best.tb) g.ca->split_k = best.tb; }
        else { g.wa->tile_m = best.tm; g.wa->tile_n = best.tn; g.wa->target_blocks = best.tb; if (best.algo) g.wa->algo = best.algo; }
        g.tm = best.tm; g.tn = best.tn; g.tb = best.tb; g.us = best_t * 1e3f; g.tuned = true;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    for (auto& z : P.zero_init) HIP_TRY(hipMemsetAsync(z.first, 0, z.second, main));
    if (P.scratch) HIP_TRY(hipMemsetAsync(P.scratch, 0, (size_t)P.scratch_cap * 4, main));
    HIP_TRY(hipStreamSynchronize(main));
    return AWR_OK;
}

static void free_all(std::vector<void*>& v) {
    for (void* p : v) (void)hipFree(p);
    v.clear();
}

static void destroy_plan(awr_plan* p) {
    for (auto e : p->events) (void)hipEventDestroy(e);
    if (p->comm && p->comm_owned) (void)hipStreamDestroy(p->comm);      // (side / branch streams belong to the process-wide pool)
    free_all(p->owned);
    delete p;
}

}  // namespace awrnet

extern "C" {

int awr_net_create(int kind, int nstack, int J, int downsample, awr_net** out) {
    AWR_REQUIRE(out, "net_create: null pointer");
    AWR_REQUIRE(kind == 0 || kind == 1, "net_create: kind must be 0 (ResNet18-deconv) or 1 (stacked hourglass)");
    AWR_REQUIRE(J >= 1 && J <= 64, "net_create: J=%d joints", J);
    AWR_REQUIRE(kind == 1 || downsample == 1 || downsample == 2 || downsample == 4, "net_create: downsample must be 1, 2 or 4 (config.py:31)");
    AWR_REQUIRE(kind == 0 || (nstack >= 1 && nstack <= 8), "net_create: nstack=%d", nstack);
    AWR_REQUIRE(kind == 1 || nstack == 0 || nstack == 1 || nstack == 18 || nstack == 50 || nstack == 101 || nstack == 152,
                "net_create: kind 0 takes the ResNet depth in `nstack`: 18 (also 0 / 1), 50, 101 or 152 (resnet_deconv.py:9-13), got %d", nstack);
    awr_net* n = new awr_net();
    n->kind = kind;
    n->J = J;
    if (kind == 0) {
        n->depth = nstack <= 1 ? 18 : nstack;
        n->nstack = n->nstage = 1;
        n->downsample = downsample;
        int lg = 0;
        while ((1 << lg) < downsample) ++lg;
        n->ndeconv = 4 - lg;
        resnet_layout(n->layout, n->depth, J, downsample);
    } else {
        n->nstack = n->nstage = nstack;
        n->downsample = 2;
        hourglass_layout(n->layout, nstack, J, n->f);
    }
    n->layout.assign();
    *out = n;
    return AWR_OK;
}

int awr_net_destroy(awr_net* n) {
    if (!n) return AWR_OK;
    for (auto* p : n->plans) destroy_plan(p);
    free_all(n->owned);
    delete n;
    return AWR_OK;
}

int awr_net_sizes(const awr_net* n, int64_t* n_tensors, int64_t* n_params, int64_t* n_active, int64_t* n_buffers, int* n_counters, int* nstage) {
    AWR_REQUIRE(n, "net_sizes: null pointer");
    if (n_tensors) *n_tensors = (int64_t)n->layout.e.size();
    if (n_params) *n_params = n->layout.n_params;
    if (n_active) *n_active = n->layout.n_active;
    if (n_buffers) *n_buffers = n->layout.n_buffers;
    if (n_counters) *n_counters = n->layout.n_counters;
    if (nstage) *nstage = n->nstage;
    return AWR_OK;
}

int awr_net_tensor_info(const awr_net* n, int64_t i, const char** key, int* kind, int* ndim, int64_t shape[4], int64_t* offset, int* unused) {
    AWR_REQUIRE(n && i >= 0 && i < (int64_t)n->layout.e.size(), "net_tensor_info: index out of range");
    const Entry& e = n->layout.e[i];
    if (key) *key = e.key.c_str();
    if (kind) *kind = e.kind;
    if (ndim) *ndim = e.ndim;
    if (shape) memcpy(shape, e.shape, sizeof e.shape);
    if (offset) *offset = e.off;
    if (unused) *unused = e.unused ? 1 : 0;
    return AWR_OK;
}

int awr_net_bind(awr_net* n, float* params, float* grads, float* buffers) {
    AWR_REQUIRE(n && params && grads && buffers, "net_bind: null pointer");
    AWR_REQUIRE((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)buffers) & 15) == 0, "net_bind: arenas must be 16-byte aligned");
    for (auto* p : n->plans) destroy_plan(p);      // plans hold raw pointers into the old arenas
    n->plans.clear();
    free_all(n->owned);
    n->params = params;
    n->grads = grads;
    n->buffers = buffers;
    return make_layers(n);
}

int awr_plan_create(awr_net* n, int B, int H, int training, unsigned supervised_mask, int bn_repeat, int n_buckets, float* img, float* const* outs,
                    float* const* grad_outs, awr_plan** out) {
    AWR_REQUIRE(n && out && img && outs, "plan_create: null pointer");
    AWR_REQUIRE(n->params, "plan_create: bind the parameter arenas first (awr_net_bind)");
    AWR_REQUIRE(B >= 1 && H >= 16 && H % 16 == 0, "plan_create: B=%d, H=%d (multiples of 16)", B, H);
    AWR_REQUIRE(!training || grad_outs, "plan_create: a training plan needs the gradient inputs");
    AWR_REQUIRE(bn_repeat >= 1 && n_buckets >= 1, "plan_create: bn_repeat / n_buckets must be >= 1");
    awr_plan* p = new awr_plan();
    p->net = n;
    p->B = B;
    p->H = H;
    p->training = training != 0;
    p->det = awr_get_deterministic() != 0;
    p->bn_repeat = bn_repeat;
    p->n_buckets = n_buckets;
    p->supervised = supervised_mask;
    p->img = img;
    Builder bld(*p);
    NetBuilder nb(bld);
    if (n->kind == 0) nb.resnet(img, H, outs[0], grad_outs ? grad_outs[0] : nullptr);
    else nb.hourglass(img, H, outs, grad_outs);
    int rc = bld.err;
    if (!rc && p->training) rc = build_backward(bld);
    if (!rc) rc = bld.err;
    if (rc) {
        destroy_plan(p);
        return rc;
    }
    n->plans.push_back(p);
    *out = p;
    return AWR_OK;
}

int awr_plan_destroy(awr_plan* p) {
    if (!p) return AWR_OK;
    auto& v = p->net->plans;
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i] == p) {
            v.erase(v.begin() + i);
            break;
        }
    destroy_plan(p);
    return AWR_OK;
}

int awr_plan_info(const awr_plan* p, int64_t* bytes, int* deterministic, int* n_fwd, int* n_bwd, int* n_buckets, int* n_gemm, int* n_bn) {
    AWR_REQUIRE(p, "plan_info: null pointer");
    if (bytes) *bytes = p->bytes;
    if (deterministic) *deterministic = p->det ? 1 : 0;
    if (n_fwd) *n_fwd = (int)p->fwd.size();
    if (n_bwd) *n_bwd = (int)p->bwd.size();
    if (n_buckets) *n_buckets = (int)p->buckets.size();
    if (n_gemm) *n_gemm = (int)p->gemms.size();
    if (n_bn) {
        int c = 0;
        for (auto& o : p->fwd) c += o.name == "awr_bn_finalize";
        *n_bn = c;
    }
    return AWR_OK;
}

int awr_plan_winograd(const awr_plan* p, int* n, double* macs) {
    AWR_REQUIRE(p, "plan_winograd: null pointer");
    int nd = 0;
    double md = 0;
    for (auto& c : p->wino_dg)
        if (awr_wino_dgrad_supported(c.first)) { ++nd; md += c.second; }
    if (n) *n = p->n_wino + nd;
    if (macs) *macs = p->wino_macs + md;
    return AWR_OK;
}

int awr_plan_bucket(const awr_plan* p, int i, int64_t* lo, int64_t* hi, int* ready_op) {
    AWR_REQUIRE(p && i >= 0 && i < (int)p->buckets.size(), "plan_bucket: index out of range");
    if (lo) *lo = p->buckets[i].lo;
    if (hi) *hi = p->buckets[i].hi;
    if (ready_op) *ready_op = p->buckets[i].ready;
    return AWR_OK;
}

int awr_plan_head_nhwc(const awr_plan* p, int stage, const float** pred, float** grad, int* Cp) {
    AWR_REQUIRE(p && stage >= 0 && stage < (int)p->head_preds.size(), "plan_head_nhwc: stage out of range");
    if (pred) *pred = p->head_preds[stage]->buf;
    if (grad) *grad = p->head_grads[stage];
    if (Cp) *Cp = p->head_preds[stage]->C;
    return AWR_OK;
}

int awr_plan_set_nhwc_boundary(awr_plan* p, int on) {
    AWR_REQUIRE(p, "plan_set_nhwc_boundary: null pointer");
    if (on && p->net->J > 56) {      // Cp = 4J rounded up to 32 > 224: the NHWC head kernels' LDS tile would exceed 64 KB (awr_head.hip: nhwc_geometry)
        set_error("plan_set_nhwc_boundary: the NHWC head / loss kernels serve at most 56 joints (got %d): use the NCHW boundary", p->net->J);
        return AWR_ERR_UNSUPPORTED;
    }
    if (on && p->training) {
        for (size_t s = 0; s < p->head_grads.size(); ++s)
            if ((p->supervised & (1u << s)) && !p->head_grads[s]) {
                set_error("plan_set_nhwc_boundary: the gradient of stage %d has a second producer inside the network (a later stack reads the prediction): "
                          "it has to be accumulated through grad_outs", (int)s);
                return AWR_ERR_UNSUPPORTED;
            }
    }
    p->nhwc_boundary = on != 0;
    return AWR_OK;
}

int awr_plan_tensor(const awr_plan* p, int i, const char** name, int dims[4], float** buf, float** grad, int* lazy, const float** lz_scale,
                    const float** lz_shift) {
    AWR_REQUIRE(p, "plan_tensor: null pointer");
    if (i < 0 || i >= (int)p->tensors.size()) return AWR_ERR_ARG;      // (no error string: callers iterate until this)
    const Tn& t = p->tensors[i];
    if (name) *name = t.name.c_str();
    if (dims) { dims[0] = t.B; dims[1] = t.H; dims[2] = t.W; dims[3] = t.C; }
    if (buf) *buf = t.buf;
    if (grad) *grad = t.grad;
    if (lazy) *lazy = t.lazy ? (t.lz_relu ? 2 : 1) : 0;
    if (lz_scale) *lz_scale = t.lz_scale;
    if (lz_shift) *lz_shift = t.lz_shift;
    return AWR_OK;
}

int awr_plan_op(const awr_plan* p, int list, int i, const char** name, double* macs, int* flags) {
    AWR_REQUIRE(p && (list == 0 || list == 1), "plan_op: list must be 0 (forward) or 1 (backward)");
    const auto& v = list ? p->bwd : p->fwd;
    AWR_REQUIRE(i >= 0 && i < (int)v.size(), "plan_op: index out of range");
    if (name) *name = v[i].name.c_str();
    if (macs) *macs = v[i].macs;
    if (flags) *flags = (v[i].side_ok ? 1 : 0) | (v[i].gemm ? 2 : 0);
    return AWR_OK;
}

}  // extern "C"

// ---- library streams --------------------------------------------------------------------------------------------------------
// HIP multiplexes streams onto a few hardware queues (4 by default); two streams on one queue run their kernels one after the other,
// and which streams share depends on how many streams the process created before (a communication library's, the framework's).
// Side streams only help when they sit on a queue of their own, so the library keeps ONE process-wide pool and orders it by a probe:
// a 100 us spin kernel on two streams at once takes 100 us when they are independent, 200 us when they share a queue.  pool[0..k) are
// mutually independent and independent of the null stream; the rest follow.  Plans borrow from the pool (never destroy).
namespace awrnet {

__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
}

static float probe_pair_ms(hipStream_t a, hipStream_t b, hipEvent_t e0, hipEvent_t e1, hipEvent_t eb) {
    const long long ticks = 10000;      // 100 us of the 100 MHz wall clock
    (void)hipEventRecord(e0, a);
    (void)hipStreamWaitEvent(b, e0, 0);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, ticks);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, ticks);
    (void)hipEventRecord(eb, b);
    (void)hipStreamWaitEvent(a, eb, 0);
    (void)hipEventRecord(e1, a);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

struct StreamPool {
    std::vector<hipStream_t> ordered;      // independent ones first
    int n_independent = 0;
};

static StreamPool& stream_pool() {
    // one pool per HIP device (streams belong to the device that was current when they were created), built once under a lock
    static std::map<int, StreamPool> pools;
    static std::mutex mtx;
    std::lock_guard<std::mutex> lock(mtx);
    int devid = 0;
    (void)hipGetDevice(&devid);
    auto it = pools.find(devid);
    if (it != pools.end()) return it->second;
    StreamPool& pool = pools[devid];
    const int NC = 8;
    std::vector<hipStream_t> cand;
    for (int i = 0; i < NC; ++i) {
        hipStream_t s;
        // AWR_SIDE_PRIORITY=low: the side / branch streams (weight gradients, forked branches) yield to the caller's stream, so that the
        // dependent chain on it runs at solo speed and the side work fills what it leaves (same-box A/B hook)
        static const bool low_prio = []() { const char* e = getenv("AWR_SIDE_PRIORITY"); return e && e[0] == 'l'; }();
        int least = 0, greatest = 0;
        if (low_prio) (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if ((low_prio ? hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least) : hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess) break;
        cand.push_back(s);
    }
    hipEvent_t e0, e1, eb;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventCreate(&eb);
    hipStream_t null_stream = nullptr;
    if (!cand.empty()) (void)probe_pair_ms(null_stream, cand[0], e0, e1, eb);      // warm-up (code load)
    std::vector<hipStream_t> indep, rest;
    for (auto c : cand) {
        bool ok = probe_pair_ms(null_stream, c, e0, e1, eb) < 0.15f;
        for (size_t k = 0; ok && k < indep.size(); ++k) ok = probe_pair_ms(indep[k], c, e0, e1, eb) < 0.15f;
        (ok ? indep : rest).push_back(c);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipEventDestroy(eb);
    pool.n_independent = (int)indep.size();
    pool.ordered = indep;
    pool.ordered.insert(pool.ordered.end(), rest.begin(), rest.end());
    return pool;
}

}  // namespace awrnet

extern "C" {

int awr_stream_pool_info(int* n_streams, int* n_independent) {
    StreamPool& sp = stream_pool();
    if (n_streams) *n_streams = (int)sp.ordered.size();
    if (n_independent) *n_independent = sp.n_independent;
    return AWR_OK;
}

int awr_plan_set_streams(awr_plan* p, int n_side, int comm) {
    AWR_REQUIRE(p && n_side >= 0 && n_side <= 4, "plan_set_streams: 0..4 side streams");
    p->side.clear();
    p->branch.clear();
    if (p->comm) {
        if (p->comm_owned) (void)hipStreamDestroy(p->comm);
        p->comm = nullptr;
    }
    StreamPool& sp = stream_pool();      // (the pool of the CURRENT device: call with the plan's device current, like every other entry point)
    // branch streams only for plans whose backward forks (every stream beyond the device's few hardware queues shares one with another
    // stream and serialises with it: measured on the data-parallel ResNet18 step, whose comm stream is the fourth)
    int nfork = 0;
    for (auto& op : p->bwd) nfork += op.kind == OP_FORK ? 1 : 0;
    const int nbranch = n_side > 0 && nfork > 0 ? 2 : 0;
    AWR_REQUIRE((int)sp.ordered.size() >= n_side + nbranch, "plan_set_streams: could not create the library's streams");
    for (int i = 0; i < n_side; ++i) p->side.push_back(sp.ordered[i]);
    for (int i = 0; i < nbranch; ++i) p->branch.push_back(sp.ordered[n_side + i]);
    // the "comm stream" buckets are handed to is the LAST side stream when there is one: a bucket's scatter has to wait for the weight
    // gradients on it anyway, and a stream of its own would be one too many for the hardware queues -- the collective itself runs on
    // the communication library's stream behind an event
    if (comm) {
        if (!p->side.empty()) {
            p->comm = p->side.back();
            p->comm_owned = false;
        } else {
            HIP_TRY(hipStreamCreateWithFlags(&p->comm, hipStreamNonBlocking));
            p->comm_owned = true;
        }
    }
    return AWR_OK;
}

int awr_plan_set_bucket_callback(awr_plan* p, awr_bucket_cb cb, void* user) {
    AWR_REQUIRE(p, "plan_set_bucket_callback: null pointer");
    if (p->dp) {      // takes effect when the communicator is detached
        p->saved_cb = cb;
        p->saved_user = user;
        return AWR_OK;
    }
    p->bucket_cb = cb;
    p->bucket_user = user;
    return AWR_OK;
}

// the plan's own bucket callback while a communicator is attached: gradient arena [lo, hi) is final in `stream` order
static int dp_dead(awr_plan* p) {      // the host destroyed the communicator before detaching it (awr_plan_set_dp(plan, NULL)): fail, do not dereference
    if (awr_dp_generation(p->dp) == p->dp_gen) return 0;
    awr::set_error("plan: the data-parallel communicator attached with awr_plan_set_dp has been destroyed (detach it with awr_plan_set_dp(plan, NULL) first)");
    p->dp = nullptr;
    p->dp_lost = true;
    return AWR_ERR_ARG;
}
static void dp_bucket(void* user, int64_t lo, int64_t hi, void* stream) {
    awr_plan* p = static_cast<awr_plan*>(user);
    if (p->dp_error || !p->dp) return;
    if ((p->dp_error = dp_dead(p))) return;
    p->dp_error = awr_dp_allreduce(p->dp, p->net->grads + lo, hi - lo, stream);
}

int awr_plan_set_dp(awr_plan* p, awr_dp* dp) {
    AWR_REQUIRE(p && p->built_bwd, "plan_set_dp: needs a training plan");
    AWR_REQUIRE(!dp || awr_dp_generation(dp) != 0, "plan_set_dp: not a live communicator");      // (before any state changes)
    // the host's own callback is saved when the plan's takes its place -- not when it is ALREADY in place (a re-attach after a lost
    // communicator, p->dp == NULL with bucket_cb == dp_bucket, must not overwrite the saved callback with dp_bucket itself)
    if (dp && p->bucket_cb != dp_bucket) {
        p->saved_cb = p->bucket_cb;
        p->saved_user = p->bucket_user;
    }
    p->dp = dp;
    p->dp_gen = dp ? awr_dp_generation(dp) : 0;
    p->dp_error = 0;
    p->dp_lost = false;
    if (dp) {
        p->bucket_cb = dp_bucket;
        p->bucket_user = p;
    } else {
        p->bucket_cb = p->saved_cb;
        p->bucket_user = p->saved_user;
    }
    return AWR_OK;
}

int awr_plan_refresh_weights(awr_plan* p, void* stream) {
    AWR_REQUIRE(p, "plan_refresh_weights: null pointer");
    return refresh_weights(*p, stream);
}

int awr_plan_forward(awr_plan* p, void* stream) {
    AWR_REQUIRE(p, "plan_forward: null pointer");
    return run_list(*p, p->fwd, stream, false);
}

int awr_plan_backward(awr_plan* p, void* stream) {
    AWR_REQUIRE(p && p->built_bwd, "plan_backward: not a training plan");
    return run_list(*p, p->bwd, stream, true);
}

int awr_plan_run_timed(awr_plan* p, int list, void* stream, float* ms) {
    AWR_REQUIRE(p && ms && (list == 0 || list == 1), "plan_run_timed: bad arguments");
    return run_timed(*p, list ? p->bwd : p->fwd, stream, ms);
}

int awr_plan_autotune(awr_plan* p, int reps, void* stream) {
    AWR_REQUIRE(p && reps >= 1, "plan_autotune: bad arguments");
    return autotune(*p, reps, stream);
}

int awr_plan_gemm(const awr_plan* p, int i, const char** name, int* tile_m, int* tile_n, int* target_blocks, float* us, int* tuned) {
    AWR_REQUIRE(p && i >= 0 && i < (int)p->gemms.size(), "plan_gemm: index out of range");
    const GemmRef& g = p->gemms[i];
    if (name) *name = g.name.c_str();
    if (tile_m) *tile_m = g.tm;
    if (tile_n) *tile_n = g.tn;
    if (target_blocks) *target_blocks = g.tb;
    if (us) *us = g.us;
    if (tuned) *tuned = g.tuned ? 1 : 0;
    return AWR_OK;
}

int awr_plan_set_gemm(awr_plan* p, int i, int tile_m, int tile_n, int target_blocks, float us) {
    AWR_REQUIRE(p && i >= 0 && i < (int)p->gemms.size(), "plan_set_gemm: index out of range");
    AWR_REQUIRE((tile_m == 1 || tile_m == 2) && (tile_n == 1 || tile_n == 2) && target_blocks >= 0, "plan_set_gemm: tiles in {1,2}");
    GemmRef& g = p->gemms[i];
    AWR_REQUIRE(!(p->det && g.wa), "plan_set_gemm: a deterministic plan sizes the per-chunk copies of its weight gradients for the default geometry");
    if (g.ca) {
        g.ca->tile_m = tile_m; g.ca->tile_n = tile_n;
        if (target_blocks && g.ca->partial && target_blocks <= g.ca->split_max) g.ca->split_k = target_blocks;      // conv launches: split-K depth
    } else {
        g.wa->tile_m = tile_m; g.wa->tile_n = tile_n; if (target_blocks) g.wa->target_blocks = target_blocks;
    }
    g.tm = tile_m; g.tn = tile_n; g.tb = target_blocks; g.us = us; g.tuned = true;
    return AWR_OK;
}

int awr_plan_gemm_algo(const awr_plan* p, int i, int* algo) {
    AWR_REQUIRE(p && algo && i >= 0 && i < (int)p->gemms.size(), "plan_gemm_algo: index out of range");
    *algo = p->gemms[i].wa ? p->gemms[i].wa->algo : 0;
    return AWR_OK;
}

int awr_plan_set_gemm_algo(awr_plan* p, int i, int algo) {
    AWR_REQUIRE(p && i >= 0 && i < (int)p->gemms.size(), "plan_set_gemm_algo: index out of range");
    GemmRef& g = p->gemms[i];
    if (!g.wa) {
        AWR_REQUIRE(algo == 0, "plan_set_gemm_algo: launch %d is not a weight gradient", i);
        return AWR_OK;
    }
    AWR_REQUIRE(!p->det, "plan_set_gemm_algo: a deterministic plan keeps the default geometry of its weight gradients");
    AWR_REQUIRE(algo >= 0 && algo <= 3 && awr_conv_wgrad_algo_ok(g.wa, algo), "plan_set_gemm_algo: algo %d does not serve launch %d", algo, i);
    g.wa->algo = algo;
    return AWR_OK;
}

}  // extern "C"
