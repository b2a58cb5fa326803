// fp32 implicit-GEMM conv1d on the gfx950 matrix cores: shared by the decoder and the exported
// rvc_conv1d_forward unit-test entry point.
#pragma once
#include <vector>

#include "common.h"

namespace rvc {

// One launch computes, for every batch item b, GEMM rows m (output channels, or (phase, channel) pairs of a
// transposed conv) and GEMM columns c (time):
//   acc[m][c] = sum_{tap, ci} W[tap][ci][m] * act(X[ci][c + tap*dil - padl])        (zero outside [0, l_in))
// where the input channels are the concatenation of two sources x1 (c1 channels, leaky slope1) and
// x2 (c2 channels, leaky slope2), both [batch][c][l_in].
// regular mode  (up_stride == 0): y[b][m][c]              = out_scale*(acc + bias[m] + res + accin)
// scatter mode  (up_stride  > 0): y[b][m % c_out][c*s + m / c_out - up_pad] = acc + bias[m % c_out]
struct ConvParams {
    const float *x1 = nullptr; int c1 = 0; float slope1 = 1.f; int64_t x1_bstride = 0;
    const float *x2 = nullptr; int c2 = 0; float slope2 = 1.f; int64_t x2_bstride = 0;
    int64_t l_in = 0;                // row length of x1 (and of x2 unless l_in2 is set)
    int64_t l_in2 = 0;               // row length of x2 (0 = same as l_in)
    const float *w = nullptr;        // [KW][c1 + c2][m_total]
    const uint16_t *w16 = nullptr;   // the same slab stored as bf16 (BASELINE cfg 4); used instead of w when set
                                     // (regular mode, 3 / 7 / 11 taps: the ResBlock / MRF-layer convs)
    const float *bias = nullptr;     // [c_out] (+ b * bias_bstride)
    int64_t bias_bstride = 0;
    const float *res = nullptr;      // [batch][c_out][l_out], regular mode only
    const float *accin = nullptr;    // [batch][c_out][l_out], regular mode only
    float *y = nullptr;
    int64_t y_bstride = 0;           // = c_out * l_out
    int m_total = 0;
    int c_out = 0;
    int64_t n_cols = 0;
    int64_t l_out = 0;
    int kw = 1, dil = 1, padl = 0;
    int up_stride = 0, up_pad = 0;
    int up_interleave = 0;   // polyphase rows ordered (channel, phase) instead of (phase, channel): even up_stride only -- a lane's neighbouring phases of one channel are 8 (rate % 4: 16) contiguous bytes of the output
    float out_scale = 1.f;
    int batch = 1;
    const uint32_t *w_wino16 = nullptr;   // ... stored as bf16 pairs (with w16)
    const void *w_direct1 = nullptr; // bf16-VALUED taps as one-term direct-form fragments (convbf1.hip, K3d): taken first where it applies
    const void *w_winobf = nullptr;  // the transformed taps as bf16x3 matrix-instruction fragments (winobf.hip): taken next
    const float *w_wino = nullptr;   // the same taps in wino.hip's layout: launch_conv may take the fast (Winograd) form for
                                     // plain 3 / 7 / 11-tap layers where it is the faster one
    int debug = 0;   // experiments only (RVC_CONV_DEBUG): 1 = skip x loads, 2 = skip y stores, 4 = skip res loads
};

// While alive, launch_conv calls made by THIS thread use `hint` utterances-in-flight instead of the process default
// (a decoder handle's own setting, rvc_decoder_set_concurrency_hint); hint <= 0 changes nothing.
struct ConcurrencyScope {
    explicit ConcurrencyScope(int hint);
    ~ConcurrencyScope();
    int saved;
};

// picks the tile configuration from m_total; returns non-zero and sets the error on unsupported shapes
int launch_conv(const ConvParams &p, hipStream_t stream);

// fused ResBlock layer (resblock.hip): y = conv2(leaky(conv1_d(leaky(x)))) + x [+ accin], * out_scale; x != y
bool resblock_layer_supported(int c, int k);
int launch_resblock_layer(const float *x, const float *w1, const float *b1, const float *w2, const float *b2, const float *accin,
                          float *y, int batch, int c, int64_t L, int k, int dil, float slope, float out_scale, hipStream_t stream);

// fused ResBlock PAIR in direct form on the bf16 matrix cores, fp32 operands as exact bf16x3 splits (resblock_bf.hip): the 32- and
// 64-channel stages; same semantics as launch_resblock_layer with the pair's taps packed as matrix-instruction fragments
// tap_splits: 3 = fp32 taps as exact bf16 triples (six products per multiply-add), 1 = bf16-VALUED taps (three products)
bool resblock_bf_enabled();   // rvc_resblock_bf16x3_set_enabled(0) (ablation build: RVC_RBF=0) puts NEW handles' stages back on the unfused kernels
void resblock_bf_set_enabled(bool on);
bool resblock_bf_supported(int c, int k, int dil);
bool resblock_bf_preferred(int c, int k, int tap_splits = 3);   // the shapes where it beats the two launches it replaces (measured)
bool resblock_bf_fits(int c, int64_t L);
size_t resblock_bf_weight_bytes(int c, int k, int tap_splits = 3);
void resblock_bf_pack_host(const float *w1, const float *w2, int c, int k, std::vector<uint16_t> *out, int tap_splits = 3);   // w: [c][c][k]
int launch_resblock_bf(const float *x, const void *u, const float *b1, const float *b2, const float *accin, float *y, int batch, int c,
                       int64_t L, int k, int dil, float slope, float out_scale, hipStream_t stream, int tap_splits = 3);

// one square conv with bf16-VALUED taps in direct form on the bf16 matrix cores, one-term taps x exact bf16x3 activations (convbf1.hip,
// K3d): C = 128 / 256; same semantics as launch_winobf2_conv (y must not alias x)
bool convbf1_supported(int c, int k, int dil);
bool convbf1_preferred(int c, int k);       // the layers of a bf16-storage handle that take it (measured): the ones K3f cannot hold
bool convbf1_fits(int c, int64_t L);
size_t convbf1_weight_bytes(int c, int k);
void convbf1_pack_host(const float *w, int c, int k, std::vector<uint16_t> *out);   // w: [c][c][k], rounded to bf16 here
int launch_convbf1(const float *x, const void *u, const float *bias, const float *res, const float *accin, float *y, int batch, int c,
                   int64_t L, int k, int dil, float slope, float out_scale, hipStream_t stream);

// the upsampling step (ConvTranspose1d in polyphase form + the folded noise conv) on the bf16 matrix cores, exact bf16x3 operands
// (upsbf.hip, K3u); even rates 2 / 8 / 10 / 12, c_in % 64 == 0, at most 64 folded noise rows
bool upsbf_supported(int c_in, int c_out, int rate, int ksize, int vk);
bool upsbf_fold_noise(int nc_k);       // the noise conv rides the GEMM as extra rows (vk = (rate - 1) stride + nc_k, + a ones row for the bias) whenever there is one
size_t upsbf_weight_bytes(int c_in, int c_out, int rate, int vk);
void upsbf_pack_host(const float *uw, const float *nw, const float *bias, int c_in, int c_out, int rate, int ksize, int vk, int nc_k,
                     int nc_stride, std::vector<uint16_t> *out);   // uw [c_in][c_out][ksize], nw [c_out][nc_k] or null
int launch_upsbf(const float *x, const float *har, int64_t Lh, const void *u, const float *bias, float *y, int batch, int c_in, int c_out,
                 int64_t L_in, int64_t L_out, int rate, int ksize, int pad, int vk, int64_t S, int64_t P, float slope, hipStream_t stream);

// K10b (conv2dbf.hip): the U-Net's 3x3 convs as exact bf16x3 products; the K-split levels share K10's fixed-order finish pass
bool conv2dbf_supported(int c_in, int c_out, int H, int W, int taps);
size_t conv2dbf_weight_bytes(int c_out, int c_in);
size_t conv2dbf_workspace_bytes(int batch, int c_in, int c_out, int H, int W);
void conv2dbf_pack_host(const float *w, int c_out, int c_in, std::vector<uint16_t> *out);   // w [c_out][c_in][3][3]
int launch_conv2dbf(const float *x, const void *u, const float *bias, const float *res, float *y, int batch, int c_in, int c_out, int H, int W,
                    int relu, float *ws, size_t ws_bytes, hipStream_t stream);
int launch_conv2d_finish(const float *partial, int split, int batch, int c_out, int H, int W, const float *bias, const float *res, int relu,
                         float *y, hipStream_t stream);   // conv2d.hip: y = act(sum_s partial[s] + bias) + res, s in order

// fp32 -> bf16, round to nearest even (what torch's .bfloat16() does); NaN stays NaN
static inline uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// fast (Winograd F(4,3), grouped) form of the 3 / 7 / 11-tap dilated convs (wino.hip): y = out_scale * (conv(act(x)) + bias + res + accin)
bool wino_enabled();   // RVC_WINO != 0 (conv.hip)
bool wino_supported(int k, int dil);
bool wino_fits(int c_in, int c_out, int64_t L);
int launch_wino_conv(const float *x, const void *u, bool u_bf16, const float *bias, const float *res, const float *accin, float *y,
                     int batch, int c_in, int c_out, int64_t L, int k, int dil, float slope, float out_scale, hipStream_t stream);
// regular conv weight [c_out][c_in][k] -> [3 ceil(k/3)][c_in / 2][c_out][2] (zero taps appended, channel pairs interleaved)
void wino_pack_host(const float *w_host, int c_out, int c_in, int k, std::vector<float> *out);
void wino_pack_host_bf16(const float *w_host, int c_out, int c_in, int k, std::vector<uint32_t> *out);   // one word per channel pair
int wino_pack_weight(const float *w_host, int c_out, int c_in, int k, float **out_dev);

// the same 7- / 11-tap layers on the bf16 matrix cores with fp32 operands split exactly into three bf16 (winobf.hip)
bool winobf_enabled();   // RVC_WINOBF != 0 (conv.hip)
bool winobf_supported(int c_in, int c_out, int k, int dil);
bool winobf_fits(int c_in, int c_out, int64_t L);
size_t winobf_weight_bytes(int c_out, int c_in, int k);
void winobf_pack_host(const float *w_host, int c_out, int c_in, int k, std::vector<uint16_t> *out);
int winobf_pack_weight(const float *w_host, int c_out, int c_in, int k, void **out_dev);
int launch_winobf_conv(const float *x, const void *u, const float *bias, const float *res, const float *accin, float *y, int batch,
                       int c_in, int c_out, int64_t L, int k, int dil, float slope, float out_scale, hipStream_t stream);
// ... with one transform point per wave (winobf2.hip): the form launch_winobf_conv takes unless the ablation build's
// RVC_WBF_V2=0 asks for winobf.hip's own kernel; takes c_out % 128 == 0 layers, fragments packed point-major
bool winobf2_enabled();
bool winobf2_supported(int c_in, int c_out, int k, int dil);
int launch_winobf2_conv(const float *x, const void *u, const float *bias, const float *res, const float *accin, float *y, int batch,
                        int c_in, int c_out, int64_t L, int k, int dil, float slope, float out_scale, hipStream_t stream);

// host-side repacks (return freshly hipMalloc'ed device buffers)
// regular conv weight [c_out][c_in][k] -> [k][c_in][c_out]
int pack_conv_weight(const float *w_host, int c_out, int c_in, int k, float **out_dev);

}  // namespace rvc
