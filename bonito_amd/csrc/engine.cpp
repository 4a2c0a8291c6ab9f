// Host side of libbonito_hip.so: error plumbing, weight packing, the encoder engine (a linear chain
// of layers executed as hand-written HIP kernels on one stream) and the thin extern "C" shells
// declared in include/bonito_hip.h.
//
// The engine is the MI355X replacement for what the reference obtains from
// koi.lstm.update_graph + cuDNN/cuBLAS under SeqdistModel.forward
// (/root/reference bonito/crf/model.py:193-194,240-246): it owns fp16 copies of the weights,
// all activation workspace (sized once for max_batch x max_chunk; HBM is 288 GB so nothing is
// re-allocated per batch) and runs conv -> [permute folded] -> LSTM x L -> LinearCRFEncoder
// [-> clamp folded] writing NTC fp16 scores straight into the caller's buffer.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/bonito_hip.h"
#include "common.h"
#include "kernels.h"

// ------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
void bh_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* bh_last_error(void) { return g_err; }
extern "C" int bh_abi_version(void) { return BH_ABI_VERSION; }
extern "C" size_t bh_sizeof_layer(void) { return sizeof(bh_layer_t); }
extern "C" int bh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        bh_set_error("hipGetDeviceCount failed");
        return -1;
    }
    return n;
}

// ------------------------------------------------------------------------------------------------
// fp32 -> fp16 bits, round-to-nearest-even (host)
static inline uint16_t f2h(float f) {
    _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

extern "C" int bh_rotary_table(int T, int dim, float* out);
static int g_q8_variant = 0;      // process-wide: geometry of the 8-bit recurrent kernel picked at engine creation ("lstm_q8_variant")
extern "C" size_t bh_conv1d_packed_halves(int Cin, int Cout, int K) {
    size_t kp = ((size_t)K * Cin + 31) / 32 * 32;
    size_t c16 = ((size_t)Cout + 15) / 16 * 16;
    return kp * c16;
}
// torch conv weight [Cout][Cin][K] -> [Cout16][Kp], column index = k*Cin + c (channel-minor taps)
extern "C" int bh_conv1d_pack(const float* w, int Cin, int Cout, int K, uint16_t* packed) {
    BH_REQUIRE(w && packed && Cin > 0 && Cout > 0 && K > 0, "conv1d_pack: bad arguments");
    size_t kp = ((size_t)K * Cin + 31) / 32 * 32;
    size_t c16 = ((size_t)Cout + 15) / 16 * 16;
    memset(packed, 0, kp * c16 * 2);
    for (int f = 0; f < Cout; ++f)
        for (int c = 0; c < Cin; ++c)
            for (int k = 0; k < K; ++k)
                packed[(size_t)f * kp + (size_t)k * Cin + c] = f2h(w[((size_t)f * Cin + c) * K + k]);
    return 0;
}
// W_hh [4H][H] (torch gate order i,f,g,o) -> [slice][gate][kstep][lane][8]: the A fragment of
// mfma 16x16x32 for rows gate*H + slice*16 + (lane&15), k = kstep*32 + (lane>>4)*8 + j.
extern "C" int bh_lstm_pack_whh(const float* whh, int H, uint16_t* packed) {
    BH_REQUIRE(whh && packed && H % 32 == 0 && H > 0, "lstm_pack_whh: H must be a positive multiple of 32");
    const int nks = H / 32, nsl = H / 16;
    for (int s = 0; s < nsl; ++s)
        for (int g = 0; g < 4; ++g)
            for (int ks = 0; ks < nks; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        int row = g * H + s * 16 + (lane & 15);
                        int col = ks * 32 + (lane >> 4) * 8 + j;
                        packed[((((size_t)s * 4 + g) * nks + ks) * 64 + lane) * 8 + j] =
                            f2h(whh[(size_t)row * H + col]);
                    }
    return 0;
}

// Tile packing for the workgroup-shared LSTM kernel: [slice][tile m][kstep][lane][8] with U = 4*MT units per slice;
// row r of tile m is (unit slice*U + (r>>2)*MT + m, gate r&3), so the MFMA result leaves all four gate
// pre-activations of MT consecutive units in one lane. w is [4H][H] in torch gate order (W_hh, or W_ih when
// insize == H).
static int lstm_pack_tiles(const float* w, int H, int MT, uint16_t* packed) {
    const int U = 4 * MT, nks = H / 32, nsl = H / U;
    for (int s = 0; s < nsl; ++s)
        for (int m = 0; m < MT; ++m)
            for (int ks = 0; ks < nks; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int r = lane & 15;
                        const int row = (r & 3) * H + s * U + (r >> 2) * MT + m;
                        const int col = ks * 32 + (lane >> 4) * 8 + j;
                        packed[((((size_t)s * MT + m) * nks + ks) * 64 + lane) * 8 + j] = f2h(w[(size_t)row * H + col]);
                    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int alloc(size_t n) {
        bytes = n;
        if (n == 0) return 0;
        BH_CHECK_HIP(hipMalloc(&p, n));
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
};

static int upload(DevBuf& b, const void* host, size_t bytes) {
    if (b.alloc(bytes)) return -1;
    BH_CHECK_HIP(hipMemcpy(b.p, host, bytes, hipMemcpyHostToDevice));
    return 0;
}
static int upload_f16(DevBuf& b, const float* w, size_t n) {
    std::vector<uint16_t> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = f2h(w[i]);
    return upload(b, h.data(), n * 2);
}
static int upload_f32(DevBuf& b, const float* w, size_t n) { return upload(b, w, n * 4); }

struct Layer {
    bh_layer_t d;        // descriptor (host pointers are not kept)
    DevBuf w0, w1, w2, w3, w4, w5, b0, b1;
    bool fused_clamp = false;   // a following CLAMP was folded into this layer
    float clamp_lo = -INFINITY, clamp_hi = INFINITY;
    // convolution: channel counts as laid out in memory. Channel-minor activations between two
    // convolutions are padded to a multiple of 8 channels (zero weights / zero bias -> act(0) = 0),
    // so e.g. the old-style 1 -> 4 -> 16 front end (crf/model.py:153-154) still runs on MFMA.
    int cin_eff = 0, cout_eff = 0;
    bool pointwise = false;     // K=1, stride 1 convolution executed by the GEMM kernel (supports the residual add)
    // 8-bit recurrent path (Q8-1, lstm_q8.hip): int8 weight tiles, per-row scales (s_ih * bound/127, s_hh/127), bound of the input
    bool q8 = false;
    int q_variant = 0;
    float q_bound = 1.0f;
    DevBuf q_wih, q_whh, q_sx, q_sh;
};

enum Layout { L_SIGNAL, L_NLC, L_TNC };

// RAII span: records a pair of events around a group of launches when profiling is on.
struct ProfSpan {
    bh_encoder* e; hipStream_t st; int cls; hipEvent_t a = nullptr, b = nullptr;
    ProfSpan(bh_encoder* e_, hipStream_t st_, int cls_);
    ~ProfSpan();
};

static inline int pad8(int n) { return (n + 7) / 8 * 8; }
static inline int conv_out_len(int L, int K, int stride, int pad) { return (L + 2 * pad - K) / stride + 1; }

}  // namespace

// ---- launch helpers shared by the kernel files (declared in common.h) ------------------------------------------------------------------
#include <mutex>
#include <unordered_map>
hipError_t bh_max_lds(const void* fn, int bytes) {
    static std::mutex mu;
    static std::unordered_map<unsigned long long, int> raised;       // (function, device) -> bytes granted so far
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long key = (unsigned long long)(uintptr_t)fn * 64ull + (unsigned)dev % 64u;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = raised.find(key);
        if (it != raised.end() && it->second >= bytes) return hipSuccess;
    }
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> g(mu);
        int& have = raised[key];
        if (have < bytes) have = bytes;
    }
    return e;
}
int bh_cu_count() {
    static std::atomic<int> cached[64];
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    cus = cached[dev].load(std::memory_order_relaxed);
    if (cus > 0) return cus;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
    cached[dev].store(cus, std::memory_order_relaxed);
    return cus;
}

struct bh_encoder {
    int device = 0;
    int max_batch = 0, max_chunk = 0;
    int n_cus = 0;
    std::vector<Layer> layers;
    DevBuf act[3], gates, sig, err, lstm_ws;
    // Timeout flags are PER FORWARD: forward number n (its "ticket") owns slot n % ERR_SLOTS of `err`; the slot is zeroed on the
    // stream in front of the forward, the persistent recurrent kernels raise it, and a 4-byte copy behind the forward mirrors it
    // into the same slot of the pinned host array. A caller that has observed the completion of forward n reads ITS flag
    // (bh_encoder_error_flag_at) - flags of other forwards in flight are neither consumed nor cleared (advisor finding, round 3:
    // the one sticky flag made a retry of batch i erase the evidence against batches i+1, i+2). At most ERR_SLOTS forwards of an
    // engine may be in flight. `sticky` collects the slots as they are recycled, for bh_encoder_check / bh_encoder_error_flag
    // ("has anything timed out since the last check").
    static constexpr int ERR_SLOTS = 64;
    int* err_host = nullptr;     // pinned [ERR_SLOTS] mirror of `err`
    // (atomics: bh_encoder_forward runs on the encoder thread of the product pipeline while bh_encoder_error_flag_at / _ack / _error_flag
    //  run on its decode thread - ctypes releases the GIL; advisor finding, round 4)
    std::atomic<long> ticket{-1};    // number of the most recent forward
    std::atomic<long> checked{-1};   // forwards <= checked have been reported by bh_encoder_check
    std::atomic<int> sticky{0};      // flags harvested from recycled slots (forwards in (checked, ticket - ERR_SLOTS])
    int* cur_err = nullptr;      // device slot of the forward being issued
    int n_act = 2;               // activation buffers in rotation: 3 when recurrent layers pre-fill their exchange sentinel
    // sentinel pre-fill of the NEXT recurrent layer's output buffer, on a side stream under the current layer's kernel
    hipStream_t fill_stream = nullptr;
    hipEvent_t fill_ready = nullptr, fill_done = nullptr;
    void* prefilled = nullptr;
    int lstm_prefill = 1;
    DevBuf q_act[2], q_ex;                 // 8-bit recurrent path: int8 activations in fragment order, exchange ring buffer
    DevBuf ex16;                           // fp16 workgroup-shared kernel: exchange ring buffer (lstm_layer_wgx_kernel)
    int lstm_pair = 1;                     // batches of more rings than one launch holds: two rings per workgroup instead of two launches
    int norm_fuse = 0;                     // transformer: 1 = alpha * residual added in the out_proj / fc2 epilogues instead of the norm kernel (measured: no gain, 67.7 vs 67.4 ms per sup step - the residual read costs the GEMM epilogue what it saves the norm kernel)
    int lstm_exchange = 1;                 // 1: hand-off through the ring buffer (no sentinel fill of the output tensor), 0: through the output
    int lstm_q8 = 1;                       // 0: run quantised layers through the fp16 kernels (A/B)
    DevBuf res;                            // pending residual projection of a QuartzNet block
    DevBuf t_qkv, t_mid, t_a, t_b, rot;   // transformer workspace + rotary cos/sin table [Tmax][32][2]
    int rot_len = 0;
    int out_features = 0;
    int lstm_force_slow = 0;
    int batch_pad = 16;          // chunks per LSTM ring: 16, or 32 when a wide (H > 512) layer uses two column tiles per ring
    int lstm_wide = 1;           // H > 512: stationary-W_hh kernel with 32-chunk rings (0: weight-streaming kernel)
    int attn_ring = 1;           // transformer: rotary in the Wqkv epilogue + persistent ring-buffer attention kernel
    int lstm_fused = 3;          // insize == hidden: 3 = + ring-in-a-workgroup kernel for narrow layers, 2 = workgroup-shared fused
                                 // kernel where it covers H, 1 = per-wave fused
                                 // kernel (input projection inside the recurrence), 0 = projection by a GEMM beforehand
    // optional per-kernel-class timing with HIP events on the caller's stream (bench.py roofline leg)
    bool profiling = false;
    struct Span { int cls; hipEvent_t a, b; };
    std::vector<Span> spans;
    ~bh_encoder() {
        for (auto& s : spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
        if (fill_ready) (void)hipEventDestroy(fill_ready);
        if (fill_done) (void)hipEventDestroy(fill_done);
        if (fill_stream) (void)hipStreamDestroy(fill_stream);
        if (err_host) (void)hipHostFree(err_host);
        for (auto& l : layers) {
            l.w0.release(); l.w1.release(); l.w2.release(); l.w3.release();
            l.w4.release(); l.w5.release(); l.b0.release(); l.b1.release();
            l.q_wih.release(); l.q_whh.release(); l.q_sx.release(); l.q_sh.release();
        }
        q_act[0].release(); q_act[1].release(); q_ex.release(); ex16.release();
        act[0].release(); act[1].release(); act[2].release(); gates.release(); sig.release(); err.release(); lstm_ws.release();
        res.release(); t_qkv.release(); t_mid.release(); t_a.release(); t_b.release(); rot.release();
    }
};

namespace {

ProfSpan::ProfSpan(bh_encoder* e_, hipStream_t st_, int cls_) : e(e_), st(st_), cls(cls_) {
    if (!e->profiling) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
    (void)hipEventRecord(a, st);
}
ProfSpan::~ProfSpan() {
    if (!a) return;
    (void)hipEventRecord(b, st);
    e->spans.push_back({cls, a, b});
}

// Walk the chain for chunks of L samples and batch N (padded): returns T, C, and optionally the
// largest activation / gate buffer needed.
static int walk(const bh_encoder* e, int N, int L, int* T_out, int* C_out, size_t* act_bytes, size_t* gate_bytes) {
    long len = L;
    int C = 1;
    size_t amax = 0, gmax = 0;
    for (const auto& l : e->layers) {
        switch (l.d.kind) {
            case BH_LAYER_CONV:
                len = conv_out_len((int)len, l.d.winlen, l.d.stride, l.d.padding);
                BH_REQUIRE(len > 0, "encoder: chunk of %d samples is too short for the convolution stack", L);
                C = l.d.out_size;
                amax = std::max(amax, (size_t)N * len * std::max(l.cout_eff, C) * 2);
                break;
            case BH_LAYER_LSTM:
                C = l.d.out_size;
                amax = std::max(amax, (size_t)N * len * C * 2);
                gmax = std::max(gmax, (size_t)N * len * 4 * C * 2);
                break;
            case BH_LAYER_LINEAR_CRF:
                C = l.d.out_size;
                break;
            case BH_LAYER_LINEAR:
                C = l.d.out_size;
                amax = std::max(amax, (size_t)N * len * C * 2);
                break;
            case BH_LAYER_CLAMP:
                break;
            case BH_LAYER_TRANSFORMER:
                amax = std::max(amax, (size_t)N * len * C * 2);
                break;
            case BH_LAYER_DWCONV:
                len = conv_out_len((int)len, l.d.winlen, l.d.stride, l.d.padding);
                BH_REQUIRE(len > 0, "encoder: chunk of %d samples is too short for the convolution stack", L);
                amax = std::max(amax, (size_t)N * len * C * 2);
                break;
            case BH_LAYER_RESIDUAL_PROJ:
                amax = std::max(amax, (size_t)N * len * l.d.out_size * 2);   // res buffer uses the same bound
                break;
            case BH_LAYER_CTC_DECODER:
                C = l.d.out_size;
                break;
            case BH_LAYER_UPSAMPLE:
                len *= l.d.scale_factor;
                amax = std::max(amax, (size_t)N * len * C * 2);
                break;
            default:
                BH_REQUIRE(false, "encoder: layer kind %d is not supported by this build", l.d.kind);
        }
    }
    *T_out = (int)len;
    *C_out = C;
    if (act_bytes) *act_bytes = amax;
    if (gate_bytes) *gate_bytes = gmax;
    return 0;
}

}  // namespace

extern "C" int bh_encoder_create(const bh_layer_t* layers, int n_layers, int device, int max_batch,
                                 int max_chunk, bh_encoder_t** out) {
    BH_REQUIRE(layers && n_layers > 0 && out, "encoder_create: bad arguments");
    BH_REQUIRE(max_batch > 0 && max_chunk > 0, "encoder_create: max_batch/max_chunk must be positive");
    int ndev = 0;
    BH_CHECK_HIP(hipGetDeviceCount(&ndev));
    BH_REQUIRE(ndev > 0, "encoder_create: no HIP device visible -- the MI355X engine has no CPU fallback");
    BH_REQUIRE(device >= 0 && device < ndev, "encoder_create: device %d out of range (%d visible)", device, ndev);
    int prev = 0;
    BH_CHECK_HIP(hipGetDevice(&prev));
    BH_CHECK_HIP(hipSetDevice(device));
    auto* e = new bh_encoder();
    e->device = device;
    e->max_batch = max_batch;
    e->max_chunk = max_chunk;
    int rc = 0;
    auto fail = [&](int code) {
        delete e;
        (void)hipSetDevice(prev);
        return code;
    };
    if (hipDeviceGetAttribute(&e->n_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) {
        bh_set_error("encoder_create: cannot query CU count");
        return fail(-1);
    }
    e->layers.resize(n_layers);
    int cur_channels = 1, cur_channels_eff = 1;
    float cur_bound = 4.0f;      // magnitude bound of the current activations, for the static int8 input scale of a Q8-1 layer
    bool any_q8 = false;
    for (int i = 0; i < n_layers; ++i) {
        Layer& L = e->layers[i];
        L.d = layers[i];
        const bh_layer_t& d = layers[i];
        switch (d.kind) {
            case BH_LAYER_CONV: {
                if (!(d.w0 && d.in_size > 0 && d.out_size > 0 && d.winlen > 0 && d.stride > 0)) {
                    bh_set_error("encoder_create: layer %d: malformed convolution", i);
                    return fail(-2);
                }
                if (d.groups > 1) { bh_set_error("encoder_create: layer %d: grouped conv not supported here", i); return fail(-2); }
                // does another convolution consume this output (possibly through a clamp)?
                bool next_is_conv = false;
                for (int j = i + 1; j < n_layers; ++j) {
                    if (layers[j].kind == BH_LAYER_CLAMP) continue;
                    next_is_conv = layers[j].kind == BH_LAYER_CONV;
                    break;
                }
                const int K = d.winlen;
                L.cin_eff = d.in_size == 1 ? 1 : cur_channels_eff;
                L.cout_eff = next_is_conv ? pad8(d.out_size) : d.out_size;
                if (d.in_size != 1 && (cur_channels != d.in_size || L.cin_eff < d.in_size)) {
                    bh_set_error("encoder_create: layer %d: convolution expects %d input channels, chain provides %d", i, d.in_size, cur_channels);
                    return fail(-2);
                }
                std::vector<float> bpad((size_t)L.cout_eff, 0.0f);
                if (d.b0) for (int f = 0; f < d.out_size; ++f) bpad[f] = d.b0[f];
                L.pointwise = d.in_size != 1 && K == 1 && d.stride == 1 && d.padding == 0 && L.cin_eff == d.in_size &&
                              d.in_size % 8 == 0 && L.cout_eff == d.out_size && d.out_size % 8 == 0;
                if (d.add_residual && !L.pointwise) {
                    bh_set_error("encoder_create: layer %d: only pointwise convolutions can add a residual", i);
                    return fail(-2);
                }
                if (L.pointwise) {
                    rc = upload_f16(L.w0, d.w0, (size_t)d.out_size * d.in_size);
                } else if (d.in_size == 1) {
                    std::vector<float> wpad((size_t)L.cout_eff * K, 0.0f);
                    for (int f = 0; f < d.out_size; ++f)
                        for (int k = 0; k < K; ++k) wpad[(size_t)f * K + k] = d.w0[(size_t)f * K + k];
                    rc = upload_f32(L.w0, wpad.data(), wpad.size());
                } else {
                    std::vector<float> wpad((size_t)L.cout_eff * L.cin_eff * K, 0.0f);
                    for (int f = 0; f < d.out_size; ++f)
                        for (int c = 0; c < d.in_size; ++c)
                            for (int k = 0; k < K; ++k)
                                wpad[((size_t)f * L.cin_eff + c) * K + k] = d.w0[((size_t)f * d.in_size + c) * K + k];
                    std::vector<uint16_t> pk(bh_conv1d_packed_halves(L.cin_eff, L.cout_eff, K));
                    rc = bh_conv1d_pack(wpad.data(), L.cin_eff, L.cout_eff, K, pk.data());
                    if (!rc) rc = upload(L.w0, pk.data(), pk.size() * 2);
                }
                if (!rc) rc = upload_f32(L.b0, bpad.data(), bpad.size());
                cur_channels = d.out_size;
                cur_channels_eff = L.cout_eff;
                cur_bound = d.activation == BH_ACT_TANH ? 1.0f : 4.0f;       // oracle/lstm_q8_ref.py: SWISH_BOUND = 4
                break;
            }
            case BH_LAYER_LSTM: {
                const int H = d.out_size, I = d.in_size;
                if (!(d.w0 && d.w1 && H > 0 && I > 0)) { bh_set_error("encoder_create: layer %d: malformed lstm", i); return fail(-2); }
                const bool reg_ok = H % 32 == 0 && H <= 512, stream_ok = H % 64 == 0 && H <= 1024;
                if (!(reg_ok || stream_ok) || I % 8 != 0) {
                    bh_set_error("encoder_create: layer %d: lstm needs hidden %% 32 == 0 (<= 512) or %% 64 == 0 (<= 1024), insize %% 8 == 0 (got %d, %d)", i, H, I);
                    return fail(-2);
                }
                rc = upload_f16(L.w0, d.w0, (size_t)4 * H * I);
                if (!rc) {
                    std::vector<uint16_t> pk((size_t)4 * H * H);
                    rc = bh_lstm_pack_whh(d.w1, H, pk.data());
                    if (!rc) rc = upload(L.w1, pk.data(), pk.size() * 2);
                }
                if (!rc) {
                    std::vector<float> b((size_t)4 * H, 0.0f);
                    for (int j = 0; j < 4 * H; ++j) b[j] = (d.b0 ? d.b0[j] : 0.0f) + (d.b1 ? d.b1[j] : 0.0f);
                    rc = upload_f32(L.b0, b.data(), b.size());
                }
                if (!rc && I == H && reg_ok) {      // fragment-packed W_ih for the fused kernel
                    std::vector<uint16_t> pk((size_t)4 * H * H);
                    rc = bh_lstm_pack_whh(d.w0, H, pk.data());
                    if (!rc) rc = upload(L.w2, pk.data(), pk.size() * 2);
                }
                if (!rc && bh_k_lstm_wide_ok(H)) {   // wide layer: W_hh tiles of 8 units + W_ih / bias with permuted rows, so that
                    const int MT = 2;                  // the GEMM writes G[t][n][(slice*4 + q)*8 + gate*2 + m]
                    std::vector<uint16_t> pk((size_t)4 * H * H);
                    rc = lstm_pack_tiles(d.w1, H, MT, pk.data());
                    if (!rc) rc = upload(L.w3, pk.data(), pk.size() * 2);
                    std::vector<float> wp((size_t)4 * H * I), bp((size_t)4 * H);
                    for (int s8 = 0; s8 < H / 8; ++s8)
                        for (int q = 0; q < 4; ++q)
                            for (int g = 0; g < 4; ++g)
                                for (int m = 0; m < MT; ++m) {
                                    const size_t dst = (((size_t)s8 * 4 + q) * 4 + g) * MT + m;
                                    const size_t src = (size_t)g * H + s8 * 8 + q * MT + m;
                                    memcpy(&wp[dst * I], d.w0 + src * I, sizeof(float) * I);
                                    bp[dst] = (d.b0 ? d.b0[src] : 0.0f) + (d.b1 ? d.b1[src] : 0.0f);
                                }
                    if (!rc) rc = upload_f16(L.w4, wp.data(), wp.size());
                    if (!rc) rc = upload_f32(L.b1, bp.data(), bp.size());
                    e->batch_pad = 32;
                }
                if (!rc && I == H && bh_k_lstm_wg_units(H) != 0) {    // tile-packed pair for the workgroup-shared kernel
                    const int MT = bh_k_lstm_wg_units(H) / 4;
                    std::vector<uint16_t> pk((size_t)4 * H * H);
                    rc = lstm_pack_tiles(d.w1, H, MT, pk.data());
                    if (!rc) rc = upload(L.w3, pk.data(), pk.size() * 2);
                    if (!rc) rc = lstm_pack_tiles(d.w0, H, MT, pk.data());
                    if (!rc) rc = upload(L.w4, pk.data(), pk.size() * 2);
                }
                if (!rc && d.quantize && I == H && bh_k_lstm_q8_units(H, g_q8_variant) != 0) {       // Q8-1 tiles and scales
                    const int U = bh_k_lstm_q8_units(H, g_q8_variant);
                    const size_t tile_bytes = (size_t)4 * H * ((H + 63) / 64 * 64);
                    std::vector<int8_t> pk(tile_bytes);
                    std::vector<float> s_ih((size_t)4 * H), s_hh((size_t)4 * H);
                    rc = bh_k_lstm_q8_pack(d.w0, H, U, pk.data(), s_ih.data());
                    if (!rc) rc = upload(L.q_wih, pk.data(), pk.size());
                    if (!rc) rc = bh_k_lstm_q8_pack(d.w1, H, U, pk.data(), s_hh.data());
                    if (!rc) rc = upload(L.q_whh, pk.data(), pk.size());
                    const float xs = (float)((double)cur_bound / 127.0);
                    for (int j = 0; j < 4 * H; ++j) { s_ih[j] = s_ih[j] * xs; s_hh[j] = s_hh[j] / 127.0f; }
                    if (!rc) rc = upload_f32(L.q_sx, s_ih.data(), s_ih.size());
                    if (!rc) rc = upload_f32(L.q_sh, s_hh.data(), s_hh.size());
                    L.q8 = true;
                    L.q_variant = g_q8_variant;
                    L.q_bound = cur_bound;
                    any_q8 = true;
                }
                cur_channels = cur_channels_eff = H;
                cur_bound = 1.0f;
                break;
            }
            case BH_LAYER_LINEAR_CRF: {
                if (!(d.w0 && d.in_size > 0 && d.out_size > 0 && d.in_size % 8 == 0)) {
                    bh_set_error("encoder_create: layer %d: malformed linearcrfencoder", i);
                    return fail(-2);
                }
                rc = upload_f16(L.w0, d.w0, (size_t)d.out_size * d.in_size);
                if (!rc && d.b0) rc = upload_f32(L.b0, d.b0, d.out_size);
                e->out_features = d.out_size;
                break;
            }
            case BH_LAYER_LINEAR: {
                if (!(d.w0 && d.in_size > 0 && d.out_size > 0 && d.in_size % 8 == 0 && d.out_size % 8 == 0) || cur_channels != d.in_size) {
                    bh_set_error("encoder_create: layer %d: linear needs in/out features %% 8 == 0 and %d input features (chain provides %d)",
                                 i, d.in_size, cur_channels);
                    return fail(-2);
                }
                rc = upload_f16(L.w0, d.w0, (size_t)d.out_size * d.in_size);
                if (!rc && d.b0) rc = upload_f32(L.b0, d.b0, d.out_size);
                cur_channels = cur_channels_eff = d.out_size;
                break;
            }
            case BH_LAYER_TRANSFORMER: {
                const int D = d.in_size, F = d.dim_ff;
                if (!(d.w0 && d.w1 && d.w2 && d.w3 && d.w4 && d.w5 && D > 0 && F > 0 && d.nhead > 0) ||
                    D % d.nhead != 0 || D / d.nhead != 64 || D % 8 != 0 || F % 8 != 0) {
                    bh_set_error("encoder_create: layer %d: transformer layer needs head_dim 64 and all weights", i);
                    return fail(-2);
                }
                rc = upload_f16(L.w0, d.w0, (size_t)3 * D * D);
                if (!rc && d.b0) rc = upload_f32(L.b0, d.b0, (size_t)3 * D);
                if (!rc) rc = upload_f16(L.w1, d.w1, (size_t)D * D);
                if (!rc && d.b1) rc = upload_f32(L.b1, d.b1, D);
                if (!rc) {   // fc1 rows interleaved (y_j, gate_j) for the SwiGLU epilogue of the GEMM
                    std::vector<float> wi((size_t)2 * F * D);
                    for (int j = 0; j < F; ++j) {
                        memcpy(&wi[(size_t)(2 * j) * D], d.w2 + (size_t)j * D, sizeof(float) * D);
                        memcpy(&wi[(size_t)(2 * j + 1) * D], d.w2 + (size_t)(F + j) * D, sizeof(float) * D);
                    }
                    rc = upload_f16(L.w2, wi.data(), wi.size());
                }
                if (!rc) rc = upload_f16(L.w3, d.w3, (size_t)D * F);
                if (!rc) rc = upload_f32(L.w4, d.w4, D);
                if (!rc) rc = upload_f32(L.w5, d.w5, D);
                break;
            }
            case BH_LAYER_DWCONV: {
                if (!(d.w0 && d.in_size > 0 && d.in_size % 8 == 0 && d.winlen > 0 && d.stride > 0) || cur_channels != d.in_size ||
                    cur_channels_eff != d.in_size) {
                    bh_set_error("encoder_create: layer %d: depthwise conv needs %d (multiple of 8) input channels, chain provides %d", i, d.in_size, cur_channels);
                    return fail(-2);
                }
                rc = upload_f32(L.w0, d.w0, (size_t)d.in_size * d.winlen);
                break;
            }
            case BH_LAYER_RESIDUAL_PROJ: {
                if (!(d.w0 && d.in_size % 8 == 0 && d.out_size % 8 == 0 && d.in_size > 0 && d.out_size > 0) ||
                    cur_channels != d.in_size || cur_channels_eff != d.in_size) {
                    bh_set_error("encoder_create: layer %d: residual projection shape mismatch", i);
                    return fail(-2);
                }
                rc = upload_f16(L.w0, d.w0, (size_t)d.out_size * d.in_size);
                if (!rc) {
                    std::vector<float> b((size_t)d.out_size, 0.0f);
                    if (d.b0) memcpy(b.data(), d.b0, sizeof(float) * d.out_size);
                    rc = upload_f32(L.b0, b.data(), b.size());
                }
                break;
            }
            case BH_LAYER_CTC_DECODER: {
                if (!(d.w0 && d.in_size % 8 == 0 && d.out_size >= 1 && d.out_size <= 8) || cur_channels != d.in_size) {
                    bh_set_error("encoder_create: layer %d: ctc decoder needs features %% 8 == 0 and <= 8 classes", i);
                    return fail(-2);
                }
                rc = upload_f32(L.w0, d.w0, (size_t)d.out_size * d.in_size);
                if (!rc) {
                    std::vector<float> b((size_t)d.out_size, 0.0f);
                    if (d.b0) memcpy(b.data(), d.b0, sizeof(float) * d.out_size);
                    rc = upload_f32(L.b0, b.data(), b.size());
                }
                e->out_features = d.out_size;
                break;
            }
            case BH_LAYER_UPSAMPLE: {
                const int D = d.in_size, sf = d.scale_factor;
                if (!(d.w0 && D > 0 && sf > 0 && D % 8 == 0)) { bh_set_error("encoder_create: layer %d: malformed upsample", i); return fail(-2); }
                rc = upload_f16(L.w0, d.w0, (size_t)sf * D * D);
                if (!rc && d.b0) rc = upload_f32(L.b0, d.b0, (size_t)sf * D);
                break;
            }
            case BH_LAYER_CLAMP: {
                // fold into the producing layer's epilogue
                int j = i - 1;
                if (j < 0 || (e->layers[j].d.kind != BH_LAYER_CONV && e->layers[j].d.kind != BH_LAYER_LINEAR_CRF &&
                              e->layers[j].d.kind != BH_LAYER_LINEAR)) {
                    bh_set_error("encoder_create: layer %d: clamp must follow a convolution or linear layer", i);
                    return fail(-2);
                }
                e->layers[j].fused_clamp = true;
                e->layers[j].clamp_lo = d.clamp_lo;
                e->layers[j].clamp_hi = d.clamp_hi;
                cur_bound = std::max(fabsf(d.clamp_lo), fabsf(d.clamp_hi));
                break;
            }
            default:
                bh_set_error("encoder_create: layer %d: kind %d is not supported by this build", i, d.kind);
                return fail(-2);
        }
        if (rc) return fail(rc);
        L.d.w0 = L.d.w1 = L.d.w2 = L.d.w3 = L.d.w4 = L.d.w5 = L.d.b0 = L.d.b1 = nullptr;
    }
    if (e->layers.back().d.kind != BH_LAYER_LINEAR_CRF && e->layers.back().d.kind != BH_LAYER_CTC_DECODER &&
        !(n_layers >= 2 && e->layers.back().d.kind == BH_LAYER_CLAMP &&
          e->layers[n_layers - 2].d.kind == BH_LAYER_LINEAR_CRF)) {
        bh_set_error("encoder_create: the chain must end in a linearcrfencoder (optionally followed by clamp) or a ctc decoder");
        return fail(-2);
    }
    int T = 0, C = 0;
    size_t ab = 0, gb = 0;
    const int Np = (max_batch + e->batch_pad - 1) / e->batch_pad * e->batch_pad;
    if (walk(e, Np, max_chunk, &T, &C, &ab, &gb)) return fail(-2);
    bool has_res = false;
    for (const auto& l : e->layers) has_res |= l.d.kind == BH_LAYER_RESIDUAL_PROJ;
    if (has_res && e->res.alloc(ab + 256)) return fail(-1);
    {   // two consecutive recurrent layers that exchange through sentinel-filled output: rotate three buffers so the
        // next layer's sentinel fill can run beside the current layer's kernel
        int prev_kind = 0;
        for (const auto& l : e->layers) {
            if (l.d.kind == BH_LAYER_CLAMP) continue;
            if (l.d.kind == BH_LAYER_LSTM && prev_kind == BH_LAYER_LSTM) e->n_act = 3;
            prev_kind = l.d.kind;
        }
        if (e->n_act == 3) {
            // (the side stream itself is created on first use: a stream that exists but idles still takes a slot in the
            // round-robin mapping of streams onto hardware queues, which multi-lane runs of narrow models notice)
            if (e->act[2].alloc(ab + 256) || hipMemset(e->act[2].p, 0, e->act[2].bytes) != hipSuccess ||
                hipEventCreateWithFlags(&e->fill_ready, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&e->fill_done, hipEventDisableTiming) != hipSuccess) {
                bh_set_error("encoder_create: sentinel pre-fill resources");
                return fail(-1);
            }
        }
    }
    if (any_q8) {      // int8 activations (fragment order, hidden size padded to 64) and the exchange ring buffer
        if (e->q_act[0].alloc(ab + 256) || e->q_act[1].alloc(ab + 256) || e->q_ex.alloc((size_t)4 * (Np / 16) * 16 * 1024 + 256))
            return fail(-1);
    }
    {   // exchange ring buffer of the fp16 workgroup-shared recurrent kernel: 4 slots x (H/32) KiB per ring
        int hmax = 0;
        for (const auto& l : e->layers)
            if (l.d.kind == BH_LAYER_LSTM && l.d.in_size == l.d.out_size && bh_k_lstm_wg_units(l.d.out_size) != 0) hmax = std::max(hmax, l.d.out_size);
        size_t exb = hmax ? bh_k_lstm_wgx_ex_bytes(Np, hmax) : 0;
        for (const auto& l : e->layers)
            if (l.d.kind == BH_LAYER_LSTM && bh_k_lstm_wide_ok(l.d.out_size)) exb = std::max(exb, bh_k_lstm_wide_ex_bytes(Np, l.d.out_size));
        if (exb && e->ex16.alloc(exb + 256)) return fail(-1);
    }
    if (e->act[0].alloc(ab + 256) || e->act[1].alloc(ab + 256) || e->gates.alloc(gb + 256) ||
        e->sig.alloc((size_t)Np * max_chunk * 2) || e->err.alloc(sizeof(int) * bh_encoder::ERR_SLOTS) || e->lstm_ws.alloc(bh_k_lstm_ws_bytes(Np, 1024)))
        return fail(-1);
    {   // transformer workspace: sized by walking to each transformer layer's token count
        long len = max_chunk;
        size_t m_qkv = 0, m_mid = 0, m_d = 0;
        int tmax = 0;
        for (const auto& l : e->layers) {
            if (l.d.kind == BH_LAYER_CONV) len = conv_out_len((int)len, l.d.winlen, l.d.stride, l.d.padding);
            else if (l.d.kind == BH_LAYER_UPSAMPLE) len *= l.d.scale_factor;
            else if (l.d.kind == BH_LAYER_TRANSFORMER) {
                const size_t M = (size_t)Np * len;
                m_qkv = std::max(m_qkv, M * 3 * l.d.in_size * 2);
                m_mid = std::max(m_mid, M * l.d.dim_ff * 2);
                m_d = std::max(m_d, M * l.d.in_size * 2);
                tmax = std::max(tmax, (int)len);
            }
        }
        if (m_qkv) {
            if (e->t_qkv.alloc(m_qkv + 256) || e->t_mid.alloc(m_mid + 256) || e->t_a.alloc(m_d + 256) || e->t_b.alloc(m_d + 256))
                return fail(-1);
            // rotary table: angle = t * 10000^(-i/32), fp32 like flash_attn.layers.rotary
            std::vector<float> cs((size_t)tmax * 32 * 2);
            if (bh_rotary_table(tmax, 64, cs.data())) return fail(-2);
            if (upload_f32(e->rot, cs.data(), cs.size())) return fail(-1);
            e->rot_len = tmax;
        }
    }
    if (hipHostMalloc((void**)&e->err_host, sizeof(int) * bh_encoder::ERR_SLOTS, hipHostMallocDefault) != hipSuccess) {
        bh_set_error("encoder_create: hipHostMalloc failed");
        return fail(-1);
    }
    for (int i = 0; i < bh_encoder::ERR_SLOTS; ++i) e->err_host[i] = 0;
    e->cur_err = (int*)e->err.p;
    if (hipMemset(e->err.p, 0, sizeof(int) * bh_encoder::ERR_SLOTS) != hipSuccess || hipMemset(e->act[0].p, 0, e->act[0].bytes) != hipSuccess ||
        hipMemset(e->act[1].p, 0, e->act[1].bytes) != hipSuccess) {
        bh_set_error("encoder_create: hipMemset failed");
        return fail(-1);
    }
    (void)hipSetDevice(prev);
    *out = e;
    return 0;
}

extern "C" void bh_encoder_destroy(bh_encoder_t* enc) {
    if (!enc) return;
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(enc->device);
    delete enc;
    (void)hipSetDevice(prev);
}

extern "C" int bh_encoder_output_shape(const bh_encoder_t* enc, int L, int* T, int* C, int* stride) {
    BH_REQUIRE(enc && L > 0, "encoder_output_shape: bad arguments");
    int t = 0, c = 0;
    if (walk(enc, 16, L, &t, &c, nullptr, nullptr)) return -2;
    if (T) *T = t;
    if (C) *C = c;
    if (stride) {
        int s = 1;
        for (const auto& l : enc->layers) {
            if (l.d.kind == BH_LAYER_CONV || l.d.kind == BH_LAYER_DWCONV) s *= l.d.stride;
            else if (l.d.kind == BH_LAYER_UPSAMPLE && l.d.scale_factor > 0) s /= l.d.scale_factor;
        }
        *stride = s;
    }
    return 0;
}

namespace {
// Do the convolutions at layers i (raw signal in), j1, j2 form the front end conv_front3_kernel serves (conv.hip)? Clamp layers in
// between are folded into the convolutions (fused_clamp) and skipped.
static bool conv_front3_at(const bh_encoder* e, size_t i, size_t* j1, size_t* j2) {
    const size_t nl = e->layers.size();
    size_t idx[2];
    int found = 0;
    for (size_t j = i + 1; j < nl && found < 2; ++j) {
        if (e->layers[j].d.kind == BH_LAYER_CLAMP) continue;
        if (e->layers[j].d.kind != BH_LAYER_CONV) return false;
        idx[found++] = j;
    }
    if (found < 2) return false;
    const Layer &a = e->layers[i], &b = e->layers[idx[0]], &c = e->layers[idx[1]];
    if (a.d.in_size != 1 || a.pointwise || b.pointwise || c.pointwise || b.d.add_residual || c.d.add_residual) return false;
    // the layer behind conv3 must be the recurrent stack (time-major store), as for conv_ws_kernel's use today
    int next_kind = 0;
    for (size_t j = idx[1] + 1; j < nl; ++j)
        if (e->layers[j].d.kind != BH_LAYER_CLAMP) { next_kind = e->layers[j].d.kind; break; }
    if (next_kind != BH_LAYER_LSTM) return false;
    if (!bh_k_conv_front3_ok(a.cout_eff, a.d.winlen, a.d.stride, b.cin_eff, b.cout_eff, b.d.winlen, b.d.stride, c.cin_eff, c.cout_eff,
                             c.d.winlen, c.d.stride))
        return false;
    if (a.d.out_size != b.d.in_size || b.d.out_size != c.d.in_size || c.cout_eff != c.d.out_size) return false;
    *j1 = idx[0];
    *j2 = idx[1];
    return true;
}

// which recurrence kernel serves a layer (see lstm.hip)
struct LstmPath { bool reg_path, wide, fused, wg, cta, q8, wgx, widex; };
static LstmPath lstm_path(const bh_encoder* e, const Layer& l) {
    const int H = l.d.out_size;
    LstmPath p;
    p.reg_path = H <= 512 && H % 32 == 0;
    p.wide = !p.reg_path && e->lstm_wide && bh_k_lstm_wide_ok(H) && l.w3.p != nullptr && l.w4.p != nullptr;
    p.fused = p.reg_path && e->lstm_fused && l.d.in_size == H && l.w2.p != nullptr;
    p.wg = p.fused && e->lstm_fused >= 2 && l.w3.p != nullptr && l.w4.p != nullptr;
    p.cta = p.wg && e->lstm_fused >= 3 && bh_k_lstm_cta_units(H) != 0 && bh_k_lstm_cta_units(H) == bh_k_lstm_wg_units(H);
    p.q8 = l.q8 && e->lstm_q8 && l.d.in_size == H;
    if (p.q8) p.fused = p.wg = p.cta = p.wide = false;
    p.wgx = p.wg && !p.cta && e->lstm_exchange && e->ex16.p != nullptr;
    p.widex = p.wide && e->lstm_exchange && e->ex16.p != nullptr;
    return p;
}
}  // namespace

// Human-readable list of the kernels the engine will launch per layer (one line each), e.g. for bench.py's roofline label.
extern "C" int bh_encoder_describe(const bh_encoder_t* e, char* buf, size_t n) {
    BH_REQUIRE(e && buf && n > 0, "encoder_describe: bad arguments");
    std::string out;
    char line[256];
    int li = 0;
    for (const auto& l : e->layers) {
        const bh_layer_t& d = l.d;
        switch (d.kind) {
            case BH_LAYER_CONV: {
                size_t fj1 = 0, fj2 = 0;
                const bool front = d.in_size == 1 && conv_front3_at(e, (size_t)li, &fj1, &fj2);
                snprintf(line, sizeof(line), "%d conv %d->%d k%d s%d: %s\n", li, d.in_size, d.out_size, d.winlen, d.stride,
                         l.pointwise ? "gemm (pointwise)" : front ? "conv_front3_kernel (this and the next two convolutions in one kernel)"
                         : d.in_size == 1 ? "conv_first_kernel" : "conv_igemm_kernel / conv_ws_kernel");
                break;
            }
            case BH_LAYER_LSTM: {
                const LstmPath p = lstm_path(e, l);
                const int H = d.out_size, U = bh_k_lstm_wg_units(H);
                if (p.q8) snprintf(line, sizeof(line), "%d lstm %d%s: lstm_layer_q8_kernel<%d,%d> (int8 W/x/h, i32 MFMA 16x16x64)\n", li, H, d.reverse ? " rev" : "", (H + 63) / 64, bh_k_lstm_q8_units(H, l.q_variant) / 4);
                else if (p.cta) snprintf(line, sizeof(line), "%d lstm %d%s: lstm_layer_cta_kernel<%d,%d>\n", li, H, d.reverse ? " rev" : "", H / 32, U / 4);
                else if (p.wgx) {
                    // (at the batch the engine was created for: more rings than one launch holds are served two per workgroup)
                    const int Np = (e->max_batch + e->batch_pad - 1) / e->batch_pad * e->batch_pad;
                    const int fit = U ? (e->n_cus / (8 * ((H / U) / 4))) * 8 : 0;
                    const bool pair = e->lstm_pair && fit > 0 && Np / 16 > fit;
                    snprintf(line, sizeof(line), "%d lstm %d%s: lstm_layer_%s_kernel<%d,%d>\n", li, H, d.reverse ? " rev" : "", pair ? "wgx2" : "wgx", H / 32, U / 4);
                }
                else if (p.fused) snprintf(line, sizeof(line), "%d lstm %d%s: lstm_layer_fused_kernel<%d>\n", li, H, d.reverse ? " rev" : "", H / 32);
                else if (p.wide) snprintf(line, sizeof(line), "%d lstm %d%s: gemm + lstm_layer_wide_kernel<%d,%s>\n", li, H, d.reverse ? " rev" : "", H / 32, p.widex ? "true" : "false");
                else if (p.reg_path) snprintf(line, sizeof(line), "%d lstm %d%s: gemm + lstm_layer_kernel<%d,false>\n", li, H, d.reverse ? " rev" : "", H / 32);
                else snprintf(line, sizeof(line), "%d lstm %d%s: gemm + lstm_layer_kernel<%d,true> (weight streaming)\n", li, H, d.reverse ? " rev" : "", H / 32);
                break;
            }
            case BH_LAYER_LINEAR_CRF: snprintf(line, sizeof(line), "%d linearcrfencoder %d->%d: gemm\n", li, d.in_size, d.out_size); break;
            case BH_LAYER_LINEAR: snprintf(line, sizeof(line), "%d linear %d->%d: gemm\n", li, d.in_size, d.out_size); break;
            case BH_LAYER_TRANSFORMER:
                snprintf(line, sizeof(line), "%d transformer d%d h%d ff%d: gemm (Wqkv%s, out_proj, fc1 SwiGLU, fc2) + %s + rmsnorm_residual_kernel\n",
                         li, d.in_size, d.nhead, d.dim_ff, e->attn_ring ? " + rotary" : "", e->attn_ring ? "attention_ring_kernel" : "attention_kernel");
                break;
            case BH_LAYER_UPSAMPLE: snprintf(line, sizeof(line), "%d linearupsample x%d: gemm\n", li, d.scale_factor); break;
            case BH_LAYER_CLAMP: snprintf(line, sizeof(line), "%d clamp: fused into the previous layer's epilogue\n", li); break;
            case BH_LAYER_DWCONV: snprintf(line, sizeof(line), "%d dwconv k%d: dwconv_kernel\n", li, d.winlen); break;
            case BH_LAYER_RESIDUAL_PROJ: snprintf(line, sizeof(line), "%d residual projection: gemm\n", li); break;
            case BH_LAYER_CTC_DECODER: snprintf(line, sizeof(line), "%d ctc decoder: ctc_head_kernel\n", li); break;
            default: snprintf(line, sizeof(line), "%d kind %d\n", li, d.kind);
        }
        out += line;
        ++li;
    }
    snprintf(buf, n, "%s", out.c_str());
    return 0;
}

extern "C" int bh_encoder_forward(bh_encoder_t* e, const void* signal, int N, int L, void* scores, void* stream_) {
    BH_REQUIRE(e && signal && scores, "encoder_forward: null argument");
    BH_REQUIRE(N > 0 && N <= e->max_batch, "encoder_forward: batch %d outside 1..%d", N, e->max_batch);
    BH_REQUIRE(L > 0 && L <= e->max_chunk, "encoder_forward: chunk %d outside 1..%d", L, e->max_chunk);
    hipStream_t st = (hipStream_t)stream_;
    int prev = 0;
    BH_CHECK_HIP(hipGetDevice(&prev));
    if (prev != e->device) BH_CHECK_HIP(hipSetDevice(e->device));
    struct Restore {
        int prev, dev;
        ~Restore() { if (prev != dev) (void)hipSetDevice(prev); }
    } restore{prev, e->device};

    // this forward's timeout slot (see bh_encoder::ERR_SLOTS): harvest the flag of the forward that used it last, zero it on the stream
    {
        const long n = ++e->ticket;
        const int slot = (int)(n % bh_encoder::ERR_SLOTS);
        if (n - bh_encoder::ERR_SLOTS > e->checked.load()) e->sticky.fetch_or(((volatile int*)e->err_host)[slot]);
        ((volatile int*)e->err_host)[slot] = 0;
        e->cur_err = (int*)e->err.p + slot;
        BH_CHECK_HIP(hipMemsetAsync(e->cur_err, 0, sizeof(int), st));
    }
    const int Np = (N + e->batch_pad - 1) / e->batch_pad * e->batch_pad;
    // stage the batch into an engine-owned [Np][L] buffer whose padding rows are zero
    if (Np != N) BH_CHECK_HIP(hipMemsetAsync((char*)e->sig.p + (size_t)N * L * 2, 0, (size_t)(Np - N) * L * 2, st));
    BH_CHECK_HIP(hipMemcpyAsync(e->sig.p, signal, (size_t)N * L * 2, hipMemcpyDeviceToDevice, st));

    e->prefilled = nullptr;
    const void* cur_q = nullptr;     // output of a Q8-1 layer feeding the next one (int8, fragment order)
    int qi = 0;
    const void* cur = e->sig.p;
    Layout lay = L_SIGNAL;
    int len = L, C = 1, which = 0;
    bool res_ready = false;
    const size_t nl = e->layers.size();
    // Sentinel pre-fill for the recurrent layer that follows layer i (whose output buffer is act[which]): the buffer after
    // it in the rotation is free once everything queued so far has run, so it is filled on the side stream while layer i's
    // own kernels run. The next layer writes H_next features per (t, n) row into it.
    auto prefill_next = [&](size_t i, int which, size_t rows) -> int {
        if (e->n_act != 3 || !e->lstm_prefill || e->profiling) return 0;
        const Layer* nx = nullptr;
        for (size_t j = i + 1; j < nl && !nx; ++j)
            if (e->layers[j].d.kind == BH_LAYER_LSTM) nx = &e->layers[j];
        if (!nx || lstm_path(e, *nx).cta || lstm_path(e, *nx).q8 || lstm_path(e, *nx).wgx || lstm_path(e, *nx).widex) return 0;   // those exchange elsewhere: nothing to pre-fill
        if (!e->fill_stream) BH_CHECK_HIP(hipStreamCreateWithFlags(&e->fill_stream, hipStreamNonBlocking));
        void* spare = e->act[(which + 1) % 3].p;
        BH_CHECK_HIP(hipEventRecord(e->fill_ready, st));
        BH_CHECK_HIP(hipStreamWaitEvent(e->fill_stream, e->fill_ready, 0));
        const int rc = bh_k_fill_u16(spare, 0xFFFFu, rows * nx->d.out_size, e->fill_stream);
        if (rc) return rc;
        BH_CHECK_HIP(hipEventRecord(e->fill_done, e->fill_stream));
        e->prefilled = spare;
        return 0;
    };
    for (size_t i = 0; i < nl; ++i) {
        Layer& l = e->layers[i];
        const bh_layer_t& d = l.d;
        if (d.kind == BH_LAYER_CLAMP) continue;
        // next non-clamp layer decides the output layout of a convolution
        int next_kind = 0;
        for (size_t j = i + 1; j < nl; ++j)
            if (e->layers[j].d.kind != BH_LAYER_CLAMP) { next_kind = e->layers[j].d.kind; break; }
        const float lo = l.fused_clamp ? l.clamp_lo : -INFINITY, hi = l.fused_clamp ? l.clamp_hi : INFINITY;
        switch (d.kind) {
            case BH_LAYER_CONV: {
                BH_REQUIRE(lay == L_SIGNAL || lay == L_NLC, "encoder_forward: convolution after a time-major layer");
                BH_REQUIRE(C == d.in_size, "encoder_forward: layer %zu expects %d channels, got %d", i, d.in_size, C);
                size_t fj1 = 0, fj2 = 0;
                if (lay == L_SIGNAL && conv_front3_at(e, i, &fj1, &fj2)) {
                    // conv1 -> conv2 -> conv3 in one kernel, the 16-channel intermediates stay in LDS (conv_front3_kernel)
                    const Layer &l2 = e->layers[fj1], &l3 = e->layers[fj2];
                    const int len1 = conv_out_len(len, d.winlen, d.stride, d.padding);
                    const int len2 = conv_out_len(len1, l2.d.winlen, l2.d.stride, l2.d.padding);
                    const int len3 = conv_out_len(len2, l3.d.winlen, l3.d.stride, l3.d.padding);
                    BH_REQUIRE(len1 > 0 && len2 > 0 && len3 > 0, "encoder_forward: chunk too short for the convolution stack");
                    void* dst3 = e->act[which].p;
                    int rc3 = prefill_next(fj2, which, (size_t)len3 * Np);
                    if (rc3) return rc3;
                    ProfSpan span(e, st, BH_PROF_CONV);
                    auto lo_of = [](const Layer& x) { return x.fused_clamp ? x.clamp_lo : -INFINITY; };
                    auto hi_of = [](const Layer& x) { return x.fused_clamp ? x.clamp_hi : INFINITY; };
                    rc3 = bh_k_conv_front3(cur, Np, len, (const float*)l.w0.p, (const float*)l.b0.p, d.winlen, d.padding, d.activation, lo_of(l),
                                           hi_of(l), l2.w0.p, (const float*)l2.b0.p, l2.d.winlen, l2.d.padding, l2.d.activation, lo_of(l2),
                                           hi_of(l2), l3.w0.p, (const float*)l3.b0.p, l3.cout_eff, l3.d.winlen, l3.d.stride, l3.d.padding,
                                           l3.d.activation, lo_of(l3), hi_of(l3), dst3, (long)l3.cout_eff, (long)Np * l3.cout_eff, st);
                    if (rc3) return rc3;
                    cur = dst3; which = (which + 1) % e->n_act; len = len3; C = l3.d.out_size; lay = L_TNC;
                    i = fj2;                        // (the two convolutions and the clamps between them are done)
                    break;
                }
                const int lout = conv_out_len(len, d.winlen, d.stride, d.padding);
                void* dst = e->act[which].p;
                const bool tnc = next_kind == BH_LAYER_LSTM;
                const int co = l.cout_eff;
                const long os_n = tnc ? co : (long)lout * co;
                const long os_t = tnc ? (long)Np * co : co;
                int rc;
                if (tnc) { rc = prefill_next(i, which, (size_t)lout * Np); if (rc) return rc; }
                ProfSpan span(e, st, BH_PROF_CONV);
                if (l.pointwise && !tnc) {
                    const void* rsd = d.add_residual ? e->res.p : nullptr;
                    BH_REQUIRE(!d.add_residual || res_ready, "encoder_forward: layer %zu adds a residual that was never projected", i);
                    rc = bh_k_linear(cur, l.w0.p, (const float*)l.b0.p, dst, Np * len, d.out_size, d.in_size, d.in_size,
                                     d.in_size, d.out_size, d.activation, 1.0f, lo, hi, 0, 0, 0, 0, 0, st, rsd, d.out_size);
                    if (d.add_residual) res_ready = false;
                } else if (lay == L_SIGNAL)
                    rc = bh_k_conv_first(cur, (const float*)l.w0.p, (const float*)l.b0.p, dst, Np, len, lout,
                                         co, d.winlen, d.stride, d.padding, d.activation, lo, hi, os_n, os_t, st);
                else
                    rc = bh_k_conv_igemm(cur, l.w0.p, (const float*)l.b0.p, dst, Np, len, lout, l.cin_eff,
                                         co, d.winlen, d.stride, d.padding, d.activation, lo, hi, os_n, os_t, st);
                if (rc) return rc;
                cur = dst; which = (which + 1) % e->n_act; len = lout; C = d.out_size; lay = tnc ? L_TNC : L_NLC;
                break;
            }
            case BH_LAYER_LSTM: {
                BH_REQUIRE(lay == L_TNC, "encoder_forward: lstm needs time-major input");
                BH_REQUIRE(C == d.in_size, "encoder_forward: layer %zu expects %d features, got %d", i, d.in_size, C);
                const int H = d.out_size;
                const int M = len * Np;
                int rc;
                void* dst = e->act[which].p;
                const LstmPath lp = lstm_path(e, l);
                const bool reg_path = lp.reg_path, wide = lp.wide, fused = lp.fused, cta = lp.cta;
                if (lp.q8) {
                    const int R = Np / 16;
                    const size_t tile = bh_k_lstm_q8_tile_bytes(H);
                    bool next_q8 = false;
                    if (next_kind == BH_LAYER_LSTM)
                        for (size_t j = i + 1; j < nl; ++j)
                            if (e->layers[j].d.kind == BH_LAYER_LSTM) { next_q8 = lstm_path(e, e->layers[j]).q8; break; }
                    const void* xq = cur_q;
                    if (!xq) {       // first quantised layer: fp16 rows -> int8 fragments with the static input scale
                        ProfSpan span(e, st, BH_PROF_OTHER);
                        rc = bh_k_quantise_rows(cur, e->q_act[qi].p, len, Np, H, R, l.q_bound, st);
                        if (rc) return rc;
                        xq = e->q_act[qi].p;
                        qi ^= 1;
                    }
                    void* hq_out = next_q8 ? e->q_act[qi].p : nullptr;
                    void* h16_out = next_q8 ? nullptr : dst;
                    ProfSpan span(e, st, BH_PROF_LSTM_REC);
                    rc = bh_k_lstm_q8_arm(e->q_ex.p, R, H, st);
                    if (rc) return rc;
                    const int U = bh_k_lstm_q8_units(H, l.q_variant);
                    const int wpr = (H / U) / 4, per_cu = U == 4 ? 3 : (l.q_variant == 2 && H == 384) ? 2 : 1;
                    const int fit = (e->n_cus * per_cu) / (8 * wpr);
                    BH_REQUIRE(fit >= 1, "encoder_forward: device has too few CUs (%d) for hidden size %d", e->n_cus, H);
                    for (int r0 = 0; r0 < R; r0 += fit * 8) {
                        const int nr = std::min(fit * 8, R - r0);
                        rc = bh_k_lstm_layer_q8((const char*)xq + (size_t)r0 * tile, l.q_wih.p, l.q_whh.p, (const float*)l.q_sx.p,
                                                (const float*)l.q_sh.p, (const float*)l.b0.p,
                                                hq_out ? (char*)hq_out + (size_t)r0 * tile : nullptr,
                                                h16_out ? (char*)h16_out + (size_t)r0 * 16 * H * 2 : nullptr,
                                                (char*)e->q_ex.p + (size_t)r0 * tile, len, Np, H, R, nr, d.reverse, e->cur_err, st,
                                                (int*)e->lstm_ws.p, e->lstm_force_slow, l.q_variant, nullptr, bh_k_lstm_max_spins());
                        if (rc) return rc;
                    }
                    if (next_q8) { cur_q = hq_out; qi ^= 1; }
                    else { cur_q = nullptr; cur = dst; which = (which + 1) % e->n_act; }
                    C = H;
                    break;
                }
                BH_REQUIRE(cur_q == nullptr, "encoder_forward: layer %zu would read int8 activations it cannot consume", i);
                if (!fused) {
                    ProfSpan span(e, st, BH_PROF_LSTM_GEMM);
                    rc = bh_k_linear(cur, wide ? l.w4.p : l.w0.p, (const float*)(wide ? l.b1.p : l.b0.p), e->gates.p, M, 4 * H,
                                     d.in_size, d.in_size, d.in_size, 4 * H, bh::ACT_NONE, 1.0f, -INFINITY, INFINITY, 0, 0, 0, 0, 0, st);
                    if (rc) return rc;
                }
                if (!cta && !lp.wgx && !lp.widex) {      // exchange sentinel in the output tensor (the ring-in-a-workgroup kernel exchanges through
                                            // LDS only, the ring-buffer kernel through its own armed buffer)
                    if (e->prefilled == dst) {          // filled beside the previous layer's kernel
                        BH_CHECK_HIP(hipStreamWaitEvent(st, e->fill_done, 0));
                    } else {
                        ProfSpan span(e, st, BH_PROF_FILL);
                        rc = bh_k_fill_u16(dst, 0xFFFFu, (size_t)M * H, st);
                        if (rc) return rc;
                    }
                }
                e->prefilled = nullptr;
                if (next_kind == BH_LAYER_LSTM) { rc = prefill_next(i, which, (size_t)M); if (rc) return rc; }
                ProfSpan span(e, st, BH_PROF_LSTM_REC);
                // co-residency: one launch serves at most (CUs / (8 * H/16)) * 32 rings
                const int nsl = H / 16;
                const bool wgk = lp.wgx || cta;                                  // a workgroup-shared kernel serves the layer (else: per-wave kernels)
                const int wg_wpr = wgk ? (H / bh_k_lstm_wg_units(H)) / 4 : 1;    // workgroups per ring
                const int groups_fit = wgk ? e->n_cus / (8 * wg_wpr) : reg_path ? e->n_cus / (8 * nsl) : e->n_cus / (8 * (nsl / 4));
                BH_REQUIRE(groups_fit >= 1, "encoder_forward: device has too few CUs (%d) for hidden size %d", e->n_cus, H);
                const int wide_fit = wide ? e->n_cus / (8 * (H / 32)) : 0;          // ring groups (of 8 rings) that are co-resident
                BH_REQUIRE(!wide || wide_fit >= 1, "encoder_forward: device has too few CUs (%d) for hidden size %d", e->n_cus, H);
                const int rings_per_launch = wide ? wide_fit * 8 : cta ? (1 << 20) : wgk ? groups_fit * 8 : reg_path ? groups_fit * 32 : groups_fit * 8;
                const int ring_chunks = wide ? 32 : 16;
                const int n_rings = Np / ring_chunks;
                for (int r0 = 0; r0 < n_rings;) {
                    // more rings than one launch holds: the ring-buffer kernel carries two rings per workgroup (lstm_layer_wgx2_kernel)
                    const bool pair = lp.wgx && e->lstm_pair && n_rings - r0 > rings_per_launch;
                    const int nr = std::min(pair ? 2 * rings_per_launch : rings_per_launch, n_rings - r0);
                    const size_t col = (size_t)r0 * ring_chunks;
                    if (pair)
                        rc = bh_k_lstm_layer_wgx2((const char*)cur + col * H * 2, l.w4.p, (const float*)l.b0.p, l.w3.p,
                                                  (char*)dst + col * H * 2, (char*)e->ex16.p + (size_t)r0 * (H / 32) * 1024, len, Np, H, n_rings,
                                                  d.reverse, e->cur_err, st, nr, (int*)e->lstm_ws.p, e->lstm_force_slow, r0 == 0);
                    else if (wide)
                        rc = bh_k_lstm_layer_wide((const char*)e->gates.p + col * 4 * H * 2, l.w3.p, (char*)dst + col * H * 2, len, Np, H,
                                                  d.reverse, e->cur_err, st, nr, (int*)e->lstm_ws.p, e->lstm_force_slow,
                                                  lp.widex ? (char*)e->ex16.p + (size_t)r0 * 2 * (H / 32) * 1024 : nullptr, n_rings, r0 == 0);
                    else if (cta)
                        rc = bh_k_lstm_layer_cta((const char*)cur + col * H * 2, l.w4.p, (const float*)l.b0.p, l.w3.p,
                                                 (char*)dst + col * H * 2, len, Np, H, d.reverse, st, nr);
                    else if (lp.wgx)
                        rc = bh_k_lstm_layer_wgx((const char*)cur + col * H * 2, l.w4.p, (const float*)l.b0.p, l.w3.p,
                                                 (char*)dst + col * H * 2, (char*)e->ex16.p + (size_t)r0 * (H / 32) * 1024, len, Np, H, n_rings,
                                                 d.reverse, e->cur_err, st, nr, (int*)e->lstm_ws.p, e->lstm_force_slow, r0 == 0);
                    else if (fused)
                        rc = bh_k_lstm_layer_fused((const char*)cur + col * H * 2, l.w2.p, (const float*)l.b0.p, l.w1.p,
                                                   (char*)dst + col * H * 2, len, Np, H, d.reverse, e->cur_err, st, nr,
                                                   (int*)e->lstm_ws.p, e->lstm_force_slow);
                    else if (reg_path)
                        rc = bh_k_lstm_layer((const char*)e->gates.p + col * 4 * H * 2, l.w1.p,
                                             (char*)dst + col * H * 2, len, Np, H, d.reverse, e->cur_err, st, nr,
                                             (int*)e->lstm_ws.p, e->lstm_force_slow);
                    else
                        rc = bh_k_lstm_layer_stream((const char*)e->gates.p + col * 4 * H * 2, l.w1.p,
                                                    (char*)dst + col * H * 2, len, Np, H, d.reverse, e->cur_err, st, nr,
                                                    (int*)e->lstm_ws.p, e->lstm_force_slow);
                    if (rc) return rc;
                    r0 += nr;
                }
                cur = dst; which = (which + 1) % e->n_act; C = H;
                break;
            }
            case BH_LAYER_LINEAR_CRF: {
                BH_REQUIRE(lay == L_TNC || lay == L_NLC, "encoder_forward: linearcrfencoder needs encoded input");
                BH_REQUIRE(C == d.in_size, "encoder_forward: layer %zu expects %d features, got %d", i, d.in_size, C);
                const int M = len * Np;
                const float sc = d.scale != 0.0f ? d.scale : 1.0f;
                int rc;
                ProfSpan span(e, st, BH_PROF_CRF_LINEAR);
                if (lay == L_TNC)   // rows are (t, n): remap to the caller's [N][T][C], drop padding rows
                    rc = bh_k_linear(cur, l.w0.p, (const float*)l.b0.p, scores, M, d.out_size, d.in_size, d.in_size,
                                     d.in_size, d.out_size, d.activation, sc, lo, hi, 0, Np, 1, len, N, st);
                else
                    rc = bh_k_linear(cur, l.w0.p, (const float*)l.b0.p, scores, len * N, d.out_size, d.in_size, d.in_size,
                                     d.in_size, d.out_size, d.activation, sc, lo, hi, 0, 0, 0, 0, 0, st);
                if (rc) return rc;
                C = d.out_size;
                break;
            }
            case BH_LAYER_LINEAR: {     // feature-axis linear layer; rows keep their layout ((t, n) or (n, t))
                BH_REQUIRE(lay == L_TNC || lay == L_NLC, "encoder_forward: linear needs encoded input");
                BH_REQUIRE(C == d.in_size, "encoder_forward: layer %zu expects %d features, got %d", i, d.in_size, C);
                void* dst = e->act[which].p;
                ProfSpan span(e, st, BH_PROF_OTHER);
                int rc = bh_k_linear(cur, l.w0.p, (const float*)l.b0.p, dst, len * Np, d.out_size, d.in_size, d.in_size, d.in_size,
                                     d.out_size, bh::ACT_NONE, 1.0f, lo, hi, 0, 0, 0, 0, 0, st);
                if (rc) return rc;
                cur = dst; which = (which + 1) % e->n_act; C = d.out_size;
                break;
            }
            case BH_LAYER_TRANSFORMER: {
                BH_REQUIRE(lay == L_NLC, "encoder_forward: transformer layer needs [N][T][D] input");
                BH_REQUIRE(C == d.in_size, "encoder_forward: layer %zu expects d_model %d, got %d", i, d.in_size, C);
                BH_REQUIRE(len <= e->rot_len, "encoder_forward: %d tokens exceed the rotary table (%d)", len, e->rot_len);
                const int D = d.in_size, F = d.dim_ff;
                const int M = N * len;            // batch-major: padded chunks sit behind the valid rows
                const float eps = d.eps > 0.0f ? d.eps : 1e-5f;
                int rc;
                // profile spans (measurement only): the attention kernel and fc1 each get a span of their own - ONE launch per span, so that a
                // roofline can name a kernel - the projections and norms around them share the two older classes
                // default: rotary + softmax scale in the Wqkv epilogue, persistent ring-buffer attention kernel; the
                // block-per-workgroup kernel (rotation applied while staging) serves wider windows and "attn_ring" = 0
                const bool ring = e->attn_ring && d.win_left <= 128 && d.win_left + d.win_right <= 256;
                {
                    ProfSpan span(e, st, BH_PROF_ATTENTION);
                    if (ring)
                        rc = bh_k_linear_qkv_rotary(cur, l.w0.p, (const float*)l.b0.p, e->t_qkv.p, M, D, D, (const float*)e->rot.p,
                                                    len, 0.125f * 1.4426950408889634f, st);     // scores in log2 units
                    else
                        rc = bh_k_linear(cur, l.w0.p, (const float*)l.b0.p, e->t_qkv.p, M, 3 * D, D, D, D, 3 * D, bh::ACT_NONE,
                                         1.0f, -INFINITY, INFINITY, 0, 0, 0, 0, 0, st);
                    if (rc) return rc;
                }
                {
                    ProfSpan span(e, st, BH_PROF_ATTENTION_CORE);
                    if (ring)
                        rc = bh_k_attention_prerotated(e->t_qkv.p, e->t_a.p, N, len, d.nhead, D / d.nhead, d.win_left, d.win_right, st);
                    else
                        rc = bh_k_attention(e->t_qkv.p, e->t_a.p, (const float*)e->rot.p, N, len, d.nhead, D / d.nhead,
                                            d.win_left, d.win_right, st);
                    if (rc) return rc;
                }
                {
                    ProfSpan span(e, st, BH_PROF_ATTENTION);
                    // DeepNorm residual alpha * x fused into the projection's epilogue (fp32 accumulator + alpha * x, one rounding); the norm
                    // kernel then reads one tensor instead of two ("norm_fuse" 1; default 0 = the separate residual read in the norm kernel)
                    if (!rc && e->norm_fuse)
                        rc = bh_k_linear(e->t_a.p, l.w1.p, (const float*)l.b1.p, e->t_b.p, M, D, D, D, D, D, bh::ACT_NONE,
                                         1.0f, -INFINITY, INFINITY, 0, 0, 0, 0, 0, st, cur, D, d.alpha);
                    else if (!rc)
                        rc = bh_k_linear(e->t_a.p, l.w1.p, (const float*)l.b1.p, e->t_b.p, M, D, D, D, D, D, bh::ACT_NONE,
                                         1.0f, -INFINITY, INFINITY, 0, 0, 0, 0, 0, st);
                    if (!rc) rc = bh_k_rmsnorm_residual(e->t_b.p, e->norm_fuse ? nullptr : cur, (const float*)l.w4.p, e->t_a.p, M, D, d.alpha, eps, st);
                    if (rc) return rc;
                }
                void* dst = e->act[which].p;
                {
                    ProfSpan span(e, st, BH_PROF_MLP_FC1);
                    rc = bh_k_linear(e->t_a.p, l.w2.p, nullptr, e->t_mid.p, M, 2 * F, D, D, D, F, bh::ACT_NONE, 1.0f,
                                     -INFINITY, INFINITY, 1, 0, 0, 0, 0, st);
                    if (rc) return rc;
                }
                {
                    ProfSpan span(e, st, BH_PROF_MLP);
                    if (e->norm_fuse)
                        rc = bh_k_linear(e->t_mid.p, l.w3.p, nullptr, e->t_b.p, M, D, F, F, F, D, bh::ACT_NONE, 1.0f,
                                         -INFINITY, INFINITY, 0, 0, 0, 0, 0, st, e->t_a.p, D, d.alpha);
                    else if (!rc)
                        rc = bh_k_linear(e->t_mid.p, l.w3.p, nullptr, e->t_b.p, M, D, F, F, F, D, bh::ACT_NONE, 1.0f,
                                         -INFINITY, INFINITY, 0, 0, 0, 0, 0, st);
                    if (!rc) rc = bh_k_rmsnorm_residual(e->t_b.p, e->norm_fuse ? nullptr : e->t_a.p, (const float*)l.w5.p, dst, M, D, d.alpha, eps, st);
                    if (rc) return rc;
                }
                cur = dst; which = (which + 1) % e->n_act;
                break;
            }
            case BH_LAYER_DWCONV: {
                BH_REQUIRE(lay == L_NLC && C == d.in_size, "encoder_forward: depthwise conv needs [N][L][%d] input", d.in_size);
                const int lout = conv_out_len(len, d.winlen, d.stride, d.padding);
                void* dst = e->act[which].p;
                ProfSpan span(e, st, BH_PROF_CONV);
                int rc = bh_k_dwconv(cur, (const float*)l.w0.p, dst, Np, len, lout, C, d.winlen, d.stride, d.padding, st);
                if (rc) return rc;
                cur = dst; which = (which + 1) % e->n_act; len = lout;
                break;
            }
            case BH_LAYER_RESIDUAL_PROJ: {
                BH_REQUIRE(lay == L_NLC && C == d.in_size, "encoder_forward: residual projection needs [N][L][%d] input", d.in_size);
                ProfSpan span(e, st, BH_PROF_CONV);
                int rc = bh_k_linear(cur, l.w0.p, (const float*)l.b0.p, e->res.p, Np * len, d.out_size, d.in_size, d.in_size,
                                     d.in_size, d.out_size, bh::ACT_NONE, 1.0f, -INFINITY, INFINITY, 0, 0, 0, 0, 0, st);
                if (rc) return rc;
                res_ready = true;
                break;
            }
            case BH_LAYER_CTC_DECODER: {
                BH_REQUIRE(lay == L_NLC && C == d.in_size, "encoder_forward: ctc decoder needs [N][T][%d] input", d.in_size);
                ProfSpan span(e, st, BH_PROF_CRF_LINEAR);
                int rc = bh_k_ctc_head(cur, (const float*)l.w0.p, (const float*)l.b0.p, scores, (long)N * len, d.in_size,
                                       d.out_size, st);
                if (rc) return rc;
                C = d.out_size;
                break;
            }
            case BH_LAYER_UPSAMPLE: {
                BH_REQUIRE(lay == L_NLC && C == d.in_size, "encoder_forward: upsample needs [N][T][%d] input", d.in_size);
                const int D = d.in_size, sf = d.scale_factor;
                void* dst = e->act[which].p;
                ProfSpan span(e, st, BH_PROF_OTHER);
                int rc = bh_k_linear(cur, l.w0.p, (const float*)l.b0.p, dst, N * len, sf * D, D, D, D, sf * D, bh::ACT_NONE,
                                     1.0f, -INFINITY, INFINITY, 0, 0, 0, 0, 0, st);
                if (rc) return rc;
                cur = dst; which = (which + 1) % e->n_act; len *= sf;     // [N][T][s*D] viewed as [N][s*T][D]
                break;
            }
            default:
                BH_REQUIRE(false, "encoder_forward: unsupported layer kind %d", d.kind);
        }
    }
    // The persistent recurrent kernels raise this forward's slot on a spin timeout and then finish with invalid output. Mirror it
    // into pinned host memory behind this forward: whoever has observed the completion of this call on `st` (an event, a D2H copy
    // of decoded outputs, a synchronise) reads it without another round trip -- bh_encoder_error_flag_at(ticket).
    BH_CHECK_HIP(hipMemcpyAsync(e->err_host + (e->cur_err - (int*)e->err.p), e->cur_err, sizeof(int), hipMemcpyDeviceToHost, st));
    return 0;
}

extern "C" long bh_encoder_last_ticket(const bh_encoder_t* e) { return e ? e->ticket.load() : -1; }

extern "C" int bh_encoder_error_flag_at(const bh_encoder_t* e, long ticket) {
    if (!e || !e->err_host) return 0;
    const long cur = e->ticket.load();
    if (ticket < 0 || ticket > cur) return 0;
    if (cur - ticket >= bh_encoder::ERR_SLOTS) {
        bh_set_error("forward %ld is more than %d forwards old: its timeout flag has been recycled", ticket, bh_encoder::ERR_SLOTS);
        return -1;
    }
    const int flag = ((volatile const int*)e->err_host)[ticket % bh_encoder::ERR_SLOTS];
    if (flag) bh_set_error("device-side timeout in a persistent kernel (flag=%d, forward %ld): the scores of that forward are invalid", flag, ticket);
    return flag;
}

// The caller has dealt with the timeout of forward `ticket` (re-ran the batch): drop its flag, so that it is neither harvested into the
// engine-wide flag when the slot is recycled nor reported by bh_encoder_error_flag / bh_encoder_check (advisor finding, round 4: a
// successful retry left poll() poisoned for the life of the engine).
extern "C" int bh_encoder_ack(bh_encoder_t* e, long ticket) {
    BH_REQUIRE(e && e->err_host, "encoder_ack: null engine");
    const long cur = e->ticket.load();
    BH_REQUIRE(ticket >= 0 && ticket <= cur, "encoder_ack: forward %ld has not been issued (last %ld)", ticket, cur);
    if (cur - ticket >= bh_encoder::ERR_SLOTS) {
        bh_set_error("forward %ld is more than %d forwards old: its timeout flag has been recycled", ticket, bh_encoder::ERR_SLOTS);
        return -1;
    }
    ((volatile int*)e->err_host)[ticket % bh_encoder::ERR_SLOTS] = 0;
    return 0;
}

// flags of the forwards that bh_encoder_check has not reported yet (host side only: completed forwards whose copy has landed)
static int pending_flags(const bh_encoder_t* e) {
    int flag = e->sticky.load();
    const long cur = e->ticket.load();
    long lo = e->checked.load() + 1;
    if (lo < cur - bh_encoder::ERR_SLOTS + 1) lo = cur - bh_encoder::ERR_SLOTS + 1;
    if (lo < 0) lo = 0;
    for (long n = lo; n <= cur; ++n) flag |= ((volatile const int*)e->err_host)[n % bh_encoder::ERR_SLOTS];
    return flag;
}

extern "C" int bh_encoder_error_flag(const bh_encoder_t* e) {
    if (!e || !e->err_host) return 0;
    const int flag = pending_flags(e);
    if (flag) bh_set_error("device-side timeout in a persistent kernel (flag=%d): the scores of that forward are invalid", flag);
    return flag;
}

extern "C" int bh_encoder_check(bh_encoder_t* e, void* stream_) {
    BH_REQUIRE(e, "encoder_check: null engine");
    BH_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream_));
    const int flag = pending_flags(e);
    // reported: forget everything up to the most recent forward. (A forward still in flight on ANOTHER stream than `stream_` is
    // past this check; its flag stays readable through its ticket.)
    e->sticky.store(0);
    if (flag) bh_set_error("device-side timeout in a persistent kernel (flag=%d)", flag);
    e->checked.store(e->ticket.load());
    return flag;
}

extern "C" int bh_encoder_profile(bh_encoder_t* e, int enable) {
    BH_REQUIRE(e, "encoder_profile: null engine");
    e->profiling = enable != 0;
    return 0;
}
extern "C" int bh_encoder_profile_read(bh_encoder_t* e, float* ms, int* launches) {
    BH_REQUIRE(e && ms && launches, "encoder_profile_read: null argument");
    for (int i = 0; i < BH_PROF_CLASSES; ++i) { ms[i] = 0.0f; launches[i] = 0; }
    for (auto& s : e->spans) {
        BH_CHECK_HIP(hipEventSynchronize(s.b));
        float t = 0.0f;
        BH_CHECK_HIP(hipEventElapsedTime(&t, s.a, s.b));
        if (s.cls >= 0 && s.cls < BH_PROF_CLASSES) { ms[s.cls] += t; launches[s.cls] += 1; }
        (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b);
    }
    e->spans.clear();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// operator-level shells
extern "C" int bh_linear(const void* X, const void* W, const float* bias, void* out, int M, int N, int K, int ldx,
                         int ldw, int ldo, int act, float scale, float clamp_lo, float clamp_hi, int gated,
                         int row_div, long row_s_hi, long row_s_lo, int row_lim, void* stream) {
    BH_REQUIRE(X && W && out, "linear: null pointer");
    return bh_k_linear(X, W, bias, out, M, N, K, ldx, ldw, ldo, act, scale, clamp_lo, clamp_hi, gated, row_div,
                       row_s_hi, row_s_lo, row_lim, (hipStream_t)stream);
}
extern "C" int bh_conv1d_first(const void* signal, const float* w, const float* bias, void* out, int N, int Lin,
                               int Cout, int K, int stride, int pad, int act, float clamp_lo, float clamp_hi,
                               long os_n, long os_t, void* stream) {
    BH_REQUIRE(signal && w && out && stride > 0, "conv1d_first: bad arguments");
    const int lout = conv_out_len(Lin, K, stride, pad);
    BH_REQUIRE(lout > 0, "conv1d_first: input too short");
    return bh_k_conv_first(signal, w, bias, out, N, Lin, lout, Cout, K, stride, pad, act, clamp_lo, clamp_hi, os_n,
                           os_t, (hipStream_t)stream);
}
extern "C" int bh_conv1d(const void* in, const void* wpacked, const float* bias, void* out, int N, int Lin, int Cin,
                         int Cout, int K, int stride, int pad, int act, float clamp_lo, float clamp_hi, long os_n,
                         long os_t, void* stream) {
    BH_REQUIRE(in && wpacked && out && stride > 0, "conv1d: bad arguments");
    const int lout = conv_out_len(Lin, K, stride, pad);
    BH_REQUIRE(lout > 0, "conv1d: input too short");
    return bh_k_conv_igemm(in, wpacked, bias, out, N, Lin, lout, Cin, Cout, K, stride, pad, act, clamp_lo, clamp_hi,
                           os_n, os_t, (hipStream_t)stream);
}
// cos/sin of position * 10000^(-2i/dim), interleaved [T][dim/2][2], fp32 products like flash_attn's rotary
extern "C" int bh_rotary_table(int T, int dim, float* out) {
    BH_REQUIRE(out && T > 0 && dim > 0 && dim % 2 == 0, "rotary_table: bad arguments");
    const int half = dim / 2;
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < half; ++i) {
            const float inv = 1.0f / powf(10000.0f, (float)(2 * i) / (float)dim);
            const float ang = (float)t * inv;
            out[((size_t)t * half + i) * 2] = cosf(ang);
            out[((size_t)t * half + i) * 2 + 1] = sinf(ang);
        }
    return 0;
}
extern "C" int bh_attention(const void* qkv, void* out, const float* cos_sin, int N, int T, int nhead, int head_dim,
                            int win_left, int win_right, void* stream) {
    BH_REQUIRE(qkv && out && cos_sin, "attention: null pointer");
    return bh_k_attention(qkv, out, cos_sin, N, T, nhead, head_dim, win_left, win_right, (hipStream_t)stream);
}
extern "C" int bh_attention_prerotated(const void* qkv, void* out, int N, int T, int nhead, int head_dim, int win_left, int win_right,
                                       void* stream) {
    BH_REQUIRE(qkv && out, "attention_prerotated: null pointer");
    return bh_k_attention_prerotated(qkv, out, N, T, nhead, head_dim, win_left, win_right, (hipStream_t)stream);
}
extern "C" int bh_rmsnorm_residual(const void* a, const void* x, const float* w, void* out, long M, int D, float alpha,
                                   float eps, void* stream) {
    BH_REQUIRE(a && x && w && out && M > 0, "rmsnorm_residual: bad arguments");
    return bh_k_rmsnorm_residual(a, x, w, out, M, D, alpha, eps, (hipStream_t)stream);
}
extern "C" int bh_ctc_greedy_decode(const float* logp, const long* offsets, int R, int classes, float qscale, float qbias,
                                    int8_t* labels, int8_t* qual, int* path, int* count, void* stream) {
    BH_REQUIRE(logp && offsets && labels && qual && path && count, "ctc_greedy_decode: null pointer");
    return bh_k_ctc_greedy(logp, offsets, R, classes, qscale, qbias, labels, qual, path, count, (hipStream_t)stream);
}
extern "C" size_t bh_ctc_beam_search_workspace(long total_steps, int R, int classes, int beam_size) {
    return bh_k_ctc_beam_workspace(total_steps, R, classes, beam_size);
}
extern "C" int bh_ctc_beam_search(const float* logp, const long* offsets, int R, int classes, int beam_size, float threshold,
                                  void* workspace, int8_t* labels, int* path, int* count, void* stream) {
    BH_REQUIRE(logp && offsets && workspace && labels && path && count, "ctc_beam_search: null pointer");
    return bh_k_ctc_prefix_beam(logp, offsets, R, classes, beam_size, threshold, workspace, labels, path, count,
                                (hipStream_t)stream);
}
extern "C" int bh_dwconv1d(const void* in, const float* w, void* out, int N, int Lin, int C, int K, int stride, int pad,
                           void* stream) {
    BH_REQUIRE(in && w && out && stride > 0, "dwconv1d: bad arguments");
    const int lout = conv_out_len(Lin, K, stride, pad);
    BH_REQUIRE(lout > 0, "dwconv1d: input too short");
    return bh_k_dwconv(in, w, out, N, Lin, lout, C, K, stride, pad, (hipStream_t)stream);
}
extern "C" size_t bh_lstm_workspace(int N, int H) { return bh_k_lstm_ws_bytes(N, H); }
extern "C" int bh_lstm_layer(const void* gates_in, const void* whh_packed, void* h_out, int T, int N, int H,
                             int reverse, void* workspace, int* err_flag, int flags, void* stream) {
    BH_REQUIRE(gates_in && whh_packed && h_out && err_flag && workspace, "lstm_layer: null pointer");
    BH_REQUIRE(T > 0, "lstm_layer: T must be positive");
    int rc = bh_k_fill_u16(h_out, 0xFFFFu, (size_t)T * N * H, (hipStream_t)stream);
    if (rc) return rc;
    if (H > 512 || (flags & 2)) {     // flags bit 1: force the weight-streaming kernel
        int dev = 0, cus = 0;
        BH_CHECK_HIP(hipGetDevice(&dev));
        BH_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        const int per = std::max(1, cus / (8 * (H / 64))) * 8;
        for (int r0 = 0; r0 < N / 16; r0 += per) {
            const int nr = std::min(per, N / 16 - r0);
            rc = bh_k_lstm_layer_stream((const char*)gates_in + (size_t)r0 * 16 * 4 * H * 2, whh_packed,
                                        (char*)h_out + (size_t)r0 * 16 * H * 2, T, N, H, reverse, err_flag,
                                        (hipStream_t)stream, nr, (int*)workspace, flags & 1);
            if (rc) return rc;
        }
        return 0;
    }
    return bh_k_lstm_layer(gates_in, whh_packed, h_out, T, N, H, reverse, err_flag, (hipStream_t)stream, N / 16,
                           (int*)workspace, flags & 1);
}
// Operator level (parity tests): one Q8-1 recurrent layer straight from fp32 host weights. Packs, uploads, quantises x with
// the static scale 127 / bound, runs the 8-bit kernel and synchronises. `sums` (optional) receives the exact int32 partial sums
// [T][N][4H][2] (input part, recurrent part) the gate arithmetic started from; `hq_frag` (optional) the int8 output in
// fragment order [T][N/16][ceil(H/64)][64][16].
extern "C" int bh_lstm_q8_layer(const void* x, float bound, const float* w_ih, const float* w_hh, const float* bias, int T, int N,
                                int H, int reverse, int variant, void* h16_out, int8_t* hq_frag, int32_t* sums, void* stream_) {
    BH_REQUIRE(x && w_ih && w_hh && h16_out && T > 0 && N > 0 && N % 16 == 0, "lstm_q8_layer: bad arguments");
    const int U = bh_k_lstm_q8_units(H, variant);
    BH_REQUIRE(U != 0, "lstm_q8_layer: hidden size %d is not covered by the 8-bit kernel", H);
    hipStream_t st = (hipStream_t)stream_;
    const int R = N / 16;
    const size_t tile = bh_k_lstm_q8_tile_bytes(H), wbytes = (size_t)4 * H * ((H + 63) / 64 * 64);
    std::vector<int8_t> pk(wbytes);
    std::vector<float> s_ih((size_t)4 * H), s_hh((size_t)4 * H), b((size_t)4 * H, 0.0f);
    DevBuf q_wih, q_whh, q_sx, q_sh, q_b, xq, ex, ws, err;
    struct Free { std::vector<DevBuf*> v; ~Free() { for (auto* d : v) d->release(); } } guard{{&q_wih, &q_whh, &q_sx, &q_sh, &q_b, &xq, &ex, &ws, &err}};
    if (bh_k_lstm_q8_pack(w_ih, H, U, pk.data(), s_ih.data()) || upload(q_wih, pk.data(), pk.size())) return -1;
    if (bh_k_lstm_q8_pack(w_hh, H, U, pk.data(), s_hh.data()) || upload(q_whh, pk.data(), pk.size())) return -1;
    const float xs = (float)((double)bound / 127.0);
    for (int j = 0; j < 4 * H; ++j) { s_ih[j] *= xs; s_hh[j] /= 127.0f; if (bias) b[j] = bias[j]; }
    if (upload_f32(q_sx, s_ih.data(), s_ih.size()) || upload_f32(q_sh, s_hh.data(), s_hh.size()) || upload_f32(q_b, b.data(), b.size())) return -1;
    if (xq.alloc((size_t)T * R * tile) || ex.alloc(4 * (size_t)R * tile) || ws.alloc(bh_k_lstm_ws_bytes(N, 1024)) || err.alloc(sizeof(int))) return -1;
    BH_CHECK_HIP(hipMemsetAsync(err.p, 0, sizeof(int), st));
    int rc = bh_k_quantise_rows(x, xq.p, T, N, H, R, bound, st);
    if (!rc) rc = bh_k_lstm_q8_arm(ex.p, R, H, st);
    if (rc) return rc;
    int dev = 0, cus = 0;
    BH_CHECK_HIP(hipGetDevice(&dev));
    BH_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int wpr = (H / U) / 4, fit = (cus * (U == 4 ? 3 : (variant == 2 && H == 384) ? 2 : 1)) / (8 * wpr);
    BH_REQUIRE(fit >= 1, "lstm_q8_layer: device has too few CUs for hidden size %d", H);
    for (int r0 = 0; r0 < R; r0 += fit * 8) {
        const int nr = std::min(fit * 8, R - r0);
        rc = bh_k_lstm_layer_q8((const char*)xq.p + (size_t)r0 * tile, q_wih.p, q_whh.p, (const float*)q_sx.p, (const float*)q_sh.p,
                                (const float*)q_b.p, hq_frag ? (char*)hq_frag + (size_t)r0 * tile : nullptr,
                                (char*)h16_out + (size_t)r0 * 16 * H * 2, (char*)ex.p + (size_t)r0 * tile, T, N, H, R, nr, reverse,
                                (int*)err.p, st, (int*)ws.p, 0, variant, sums ? sums + (size_t)r0 * 16 * 4 * H * 2 : nullptr,
                                bh_k_lstm_max_spins());
        if (rc) return rc;
    }
    int flag = 0;
    BH_CHECK_HIP(hipMemcpyAsync(&flag, err.p, sizeof(int), hipMemcpyDeviceToHost, st));
    BH_CHECK_HIP(hipStreamSynchronize(st));
    BH_REQUIRE(flag == 0, "lstm_q8_layer: exchange timeout in the recurrent kernel");
    return 0;
}

// debug: copy the LSTM statistics block (tune bit 4) of the last launch to the host
extern "C" int bh_encoder_debug_read(bh_encoder_t* e, void* host, size_t bytes, size_t offset) {
    BH_REQUIRE(e && host && offset + bytes <= e->lstm_ws.bytes, "encoder_debug_read: out of range");
    BH_CHECK_HIP(hipMemcpy(host, (char*)e->lstm_ws.p + offset, bytes, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int bh_encoder_set_option(bh_encoder_t* e, const char* name, int value) {
    BH_REQUIRE(e && name, "encoder_set_option: null argument");
    if (!strcmp(name, "lstm_force_slow")) { e->lstm_force_slow = (e->lstm_force_slow & ~1) | (value & 1); return 0; }
    if (!strcmp(name, "lstm_fused")) { e->lstm_fused = value; return 0; }
    if (!strcmp(name, "attn_ring")) { e->attn_ring = value; return 0; }
    if (!strcmp(name, "lstm_wide")) { e->lstm_wide = value; return 0; }
    if (!strcmp(name, "lstm_prefill")) { e->lstm_prefill = value; return 0; }
    if (!strcmp(name, "lstm_q8")) { e->lstm_q8 = value; return 0; }
    if (!strcmp(name, "lstm_exchange")) { e->lstm_exchange = value; return 0; }
    if (!strcmp(name, "lstm_pair")) { e->lstm_pair = value; return 0; }
    if (!strcmp(name, "norm_fuse")) { e->norm_fuse = value; return 0; }
    if (!strcmp(name, "gemm_v1")) { bh_k_linear_force_v1(value); return 0; }   // process-wide A/B switch
    if (!strcmp(name, "lstm_tune")) { e->lstm_force_slow = (e->lstm_force_slow & 1) | (value << 8); return 0; }
    BH_REQUIRE(false, "encoder_set_option: unknown option '%s'", name);
}
extern "C" size_t bh_beam_search_workspace(int N, int T, int state_len) { return bh_k_beam_workspace(N, T, state_len); }
extern "C" int bh_beam_search(const void* scores, int N, int T, int state_len, int beam_width, float beam_cut,
                              float blank_score, float q_scale, float q_offset, void* workspace, int8_t* sequence,
                              int8_t* qstring, int8_t* moves, float* qfloat, void* stream) {
    BH_REQUIRE(scores && workspace && sequence && qstring && moves, "beam_search: null pointer");
    return bh_k_beam_search(scores, N, T, state_len, beam_width, beam_cut, blank_score, q_scale, q_offset, workspace,
                            sequence, qstring, moves, qfloat, (hipStream_t)stream);
}
extern "C" int bh_crf_reverse_complement(const void* in, void* out, int N, int T, int state_len, int layout_5s,
                                         long stride_n, long stride_t, void* stream) {
    BH_REQUIRE(in && out, "crf_reverse_complement: null pointer");
    return bh_k_crf_revcomp(in, out, N, T, state_len, layout_5s, stride_n, stride_t, (hipStream_t)stream);
}
extern "C" int bh_crf_logz(const void* scores, int N, int T, int state_len, float blank_score, void* workspace,
                           double* logz, void* stream) {
    BH_REQUIRE(scores && workspace && logz, "crf_logz: null pointer");
    return bh_k_crf_logz(scores, N, T, state_len, blank_score, workspace, logz, (hipStream_t)stream);
}
extern "C" int bh_signal_normalise(const int16_t* raw, const long* offsets, const float* cal_scale, const float* cal_offset, int n_reads,
                                   int strategy, double quantile_a, double quantile_b, double shift_mult, double scale_mult,
                                   double fixed_shift, double fixed_scale, int do_trim, double* shift, double* scale, int* weak,
                                   int* trim, void* stream) {
    return bh_k_signal_normalise(raw, offsets, cal_scale, cal_offset, n_reads, strategy, quantile_a, quantile_b, shift_mult,
                                 scale_mult, fixed_shift, fixed_scale, do_trim, shift, scale, weak, trim, (hipStream_t)stream);
}
extern "C" int bh_signal_chunks(const int16_t* raw, const long* offsets, const float* cal_scale, const float* cal_offset,
                                const double* shift, const double* scale, const int* weak, const int* chunk_read,
                                const long* chunk_start, const long* chunk_len, int n_chunks, int chunk_samples, void* out,
                                void* stream) {
    BH_REQUIRE(raw && offsets && cal_scale && cal_offset && shift && scale && weak && chunk_read && chunk_start && chunk_len && out,
               "signal_chunks: null pointer");
    return bh_k_signal_chunks(raw, offsets, cal_scale, cal_offset, shift, scale, weak, chunk_read, chunk_start, chunk_len,
                              n_chunks, chunk_samples, out, (hipStream_t)stream);
}
extern "C" int bh_set_option(const char* name, int value) {
    BH_REQUIRE(name != nullptr, "set_option: null name");
    if (bh_k_decode_set_option(name, value) == 0) return 0;
    if (bh_k_conv_set_option(name, value) == 0) return 0;
    if (bh_k_lstm_set_option(name, value) == 0) return 0;
    if (!strcmp(name, "gemm_path")) { bh_k_linear_force_v1(value); return 0; }
    if (!strcmp(name, "attn_waves")) { extern int g_attn_waves; g_attn_waves = value; return 0; }
    if (!strcmp(name, "attn_version")) { extern int g_attn_version; g_attn_version = value == 1 ? 1 : 2; return 0; }
    if (!strcmp(name, "attn_expt")) { extern int g_attn_expt; g_attn_expt = value; return 0; }
    if (!strcmp(name, "gemm_stagger")) { bh_k_linear_stagger(value); return 0; }
    if (!strcmp(name, "gemm_order")) { bh_k_linear_order(value); return 0; }
    if (!strcmp(name, "gemm_gf")) { bh_k_linear_gf(value); return 0; }
    if (!strcmp(name, "gemm_tile16")) { bh_k_linear_tile16(value); return 0; }     // process-wide A/B switch: the four-wave GEMM's MFMA shape
    if (!strcmp(name, "lstm_q8_variant")) { g_q8_variant = value; return 0; }
    BH_REQUIRE(false, "set_option: unknown option '%s'", name);
    return -1;
}
extern "C" size_t bh_crf_posterior_viterbi_workspace(int N, int T, int state_len) {
    return bh_k_posterior_viterbi_workspace(N, T, state_len);
}
extern "C" int bh_crf_posterior_viterbi(const void* scores, int N, int T, int state_len, float blank_score, void* workspace,
                                        int8_t* moves, int8_t* path, void* stream) {
    BH_REQUIRE(scores && workspace && moves && path, "crf_posterior_viterbi: null pointer");
    return bh_k_posterior_viterbi(scores, N, T, state_len, blank_score, workspace, moves, path, (hipStream_t)stream);
}
extern "C" size_t bh_crf_viterbi_workspace(int N, int T, int state_len) {
    size_t S = 1;
    for (int i = 0; i < state_len; ++i) S *= 4;
    return (size_t)N * T * S + 256;
}
extern "C" int bh_crf_viterbi(const void* scores, int N, int T, int state_len, int layout_5s, float blank_score,
                              long stride_n, long stride_t, void* workspace, int8_t* moves, int8_t* path,
                              float* best, void* stream) {
    BH_REQUIRE(scores && workspace && moves && path, "crf_viterbi: null pointer");
    return bh_k_crf_viterbi(scores, N, T, state_len, layout_5s, blank_score, stride_n, stride_t, workspace, nullptr,
                            moves, path, best, (hipStream_t)stream);
}
