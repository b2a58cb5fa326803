// Shared host-side structures of the vocoder handle (decoder.hip: NSF / MRF schedule; refine.hip: RefineGAN).
#pragma once
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "conv.h"

namespace rvc {

constexpr int MRF_MAX_DIM = 9;

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
    int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

struct DevBuf {
    float *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int upload(const std::vector<float> &h) {
        if (p) { (void)hipFree(p); p = nullptr; }
        hipError_t e = hipMalloc((void **)&p, h.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
        return e == hipSuccess ? 0 : fail("device upload of %zu floats failed: %s", h.size(), hipGetErrorString(e));
    }
};

struct DevBuf16 {
    uint16_t *p = nullptr;
    ~DevBuf16() { if (p) (void)hipFree(p); }
    int upload(const std::vector<uint16_t> &h) {
        if (p) { (void)hipFree(p); p = nullptr; }
        hipError_t e = hipMalloc((void **)&p, h.size() * sizeof(uint16_t));
        if (e == hipSuccess) e = hipMemcpy(p, h.data(), h.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
        return e == hipSuccess ? 0 : fail("device upload of %zu bf16 values failed: %s", h.size(), hipGetErrorString(e));
    }
};

struct ConvW {
    DevBuf w, b;
    DevBuf16 w16;                 // set instead of w when the handle stores its ResBlock weights as bf16
    DevBuf wu;                    // square 3 / 7 / 11-tap layers: the taps again in wino.hip's layout
    DevBuf16 wu16;                // ... as bf16 pairs when the handle stores bf16 (two uint16 per word)
    DevBuf16 wx;                  // square 7 / 11-tap layers at >= 64 channels (fp32 storage): transformed taps split in three bf16 (winobf.hip)
    DevBuf16 wd;                  // bf16 storage, 128 / 256 channels where the fused pair does not apply: one-term direct-form fragments (convbf1.hip)
    int c_in = 0, c_out = 0, k = 0;
};

struct Stage {
    int c_in = 0, c_out = 0, rate = 0, ksize = 0, pad = 0, opad = 0;
    int taps = 0;                 // polyphase taps J = ceil(k / rate)
    bool interleave = false;      // GEMM rows ordered (channel, phase): even rates
    int nc_stride = 0, nc_k = 0, nc_pad = 0;
    int vk = 0, vk_rows = 0;      // folded noise-conv rows: valid, padded to a multiple of 8
    int64_t S = 0, P = 0;         // V[k][q] = har[q*S + k - P]
    DevBuf w, b;                  // [taps][c_in + vk_rows][rate * c_out], [c_out]
    // nc_rows > 0: the noise conv is NOT folded into the upsampler's GEMM (w has c_in rows per tap) but added by its own
    // launch, y += W_nc V2 with V2[k][t] = har[t * nc_stride + k - nc_pad].  Folding it costs (rate - 1) * nc_stride + nc_k
    // GEMM rows per tap for nc_k useful ones: 520 against 512 real input channels in the first stage of the 48 k vocoder
    // (half of that launch's matrix instructions multiplied zeros), 8-48 rows in the later stages.
    DevBuf16 wub;                 // the upsampler GEMM (+ the folded noise rows) as bf16x3 fragments (upsbf.hip, K3u); taken instead of w when set
    int ub_vk = 0;                // noise rows folded into K3u's GEMM (0: the stage's noise conv is a separate launch)
    int nc_rows = 0;              // nc_k padded to a multiple of 8
    DevBuf nc_w;                  // [1][nc_rows][c_out]
    std::vector<ConvW> c1, c2;    // [n_res_kernels * n_res_dilations]
    std::vector<DevBuf16> pair;   // 32- / 64-channel stages: (c1, c2) of each entry as one slab of bf16x3 matrix-instruction fragments (resblock_bf.hip)
    int pair_splits = 3;          // ... with this many terms per tap: 1 when the handle stores bf16 weights (the taps are bf16-valued)
};

// One forward call's side streams: the ResBlock branches of a stage (resblocks[i * nk + m], m = 0..nk-1: hifigan_nsf.py:195-203 sums
// them) depend on the stage's input only, so branch m > 0 runs on side stream m - 1 next to branch 0 on the caller's stream.  A
// whole-CU kernel whose grid is 1.17 rounds of the 256 CUs (stage 0 of the 48 k vocoder: 300 blocks) leaves most of the chip idle
// through its second round; the next branch's blocks take those CUs.
constexpr int MAX_BRANCH_LANES = 7;
struct BranchLanes {
    hipStream_t side[MAX_BRANCH_LANES] = {};
    hipEvent_t fork = nullptr;
    hipEvent_t last[MAX_BRANCH_LANES + 1] = {};   // the branch's last launch (it adds into the running sum after its predecessor's)
    int n_side = 0;
    int create(int n);
    ~BranchLanes();
};

// RefineGAN-only state (refine.hip)
struct RefineStage {
    int ch_in = 0, ch_out = 0, rate = 0, down_c = 0, down_k = 0, down_stride = 0, down_pad = 0;
    DevBuf down_w, down_b;             // [down_c][down_k], [down_c]
    ConvW input_conv;                  // k7, (ch_in + down_c) -> ch_out
    DevBuf adain1[4], adain2[4];       // per branch [ch_out]
    std::vector<ConvW> c1, c2;         // [n_res_kernels * n_res_dilations]
    std::vector<DevBuf16> pair;        // narrow stages: (c1, c2) of each entry as one slab of bf16x3 fragments (resblock_bf.hip)
};

}  // namespace rvc

struct rvc_decoder {
    rvc_decoder_config cfg;
    std::map<std::string, rvc::HostTensor> host;
    bool finalized = false;
    int upp = 1;
    int dim = 1;              // sine components (NSF 1, MRF 9)
    float lin_w[rvc::MRF_MAX_DIM] = {0};
    float lin_b = 0.f;
    rvc::ConvW pre;                // conv_pre packed
    rvc::DevBuf cond_w, cond_b;
    std::vector<rvc::Stage> stages;
    rvc::DevBuf post_w;
    float post_b = 0.f;
    int post_cin = 0;
    // RefineGAN (kind == RVC_DEC_REFINE)
    float merge_w = 1.f;
    rvc::DevBuf pre_w, pre_b;      // pre_conv [256][7], [256]
    rvc::ConvW mel;                // mel_conv k7 192 -> 256
    std::vector<rvc::RefineStage> rstages;
    std::atomic<int> concurrency{0};   // rvc_decoder_set_concurrency_hint (0 = process default)
    // rvc_decoder_set_branch_parallel: ResBlock branches of a short stage on this many side streams (NSF / MRF schedules); -1 = one per branch after the first
    std::atomic<int> branch_parallel{0};
    std::mutex lanes_mu;
    std::map<hipStream_t, rvc::BranchLanes *> lanes;  // per caller stream (utterances in flight on different streams must not queue behind each other's branches)
    ~rvc_decoder() { for (auto &kv : lanes) delete kv.second; }
    // debug tap
    int tap_stage = -2;
    float *tap_dev = nullptr;
};


namespace rvc {
const HostTensor *find(const rvc_decoder *d, const std::string &name);
int need(const rvc_decoder *d, const std::string &name, const HostTensor **out, std::vector<int64_t> shape);
int build_conv(const rvc_decoder *d, const std::string &prefix, int c_out, int c_in, int k, bool bias, ConvW *out, bool as_bf16 = false);

// small kernels shared by both schedules (launch wrappers defined in decoder.hip)
int launch_unfold_src(const float *har, int batch, int64_t L, int64_t S, int64_t P, int k_valid, int k_rows, int64_t nq, float *V,
                      hipStream_t stream);
int launch_cond_bias(const float *pre_b, const float *cond_w, const float *cond_b, const float *g, int batch, int gin, int c0,
                     float *out, hipStream_t stream);
int launch_conv_post(const float *x, const float *w, float bias, int batch, int c_in, int64_t L, float slope, float *out,
                     hipStream_t stream);

// refine.hip
int refine_finalize(rvc_decoder *d);
size_t refine_workspace_bytes(const rvc_decoder *d, int batch, int64_t T);
int refine_forward(rvc_decoder *d, const float *z_dev, const float *f0_dev, const float *g_dev, const rvc_decoder_noise *noise,
                   int batch, int64_t T, float *out_dev, void *workspace_dev, size_t workspace_bytes, hipStream_t stream);
}  // namespace rvc
