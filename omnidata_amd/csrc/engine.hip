// engine.hip -- libdptx.so host side: the C ABI of include/dptx.h, weight folding/packing, the
// activation arena and the fixed kernel schedule of one DPT-Hybrid-384 forward.
//
// Reference being replaced (paths under omnidata_tools/torch/modules/midas/):
//   dpt_depth.py:67-85  DPT.forward          -> Engine::forward
//   vit.py:119-155      forward_flex         -> stem / stages / tokens / 12 blocks
//   vit.py:61-99        forward_vit          -> readout + reassemble (act_postprocess3/4)
//   blocks.py:263-341   RCU / FeatureFusion  -> fusion()
//   dpt_depth.py:91-99  head                 -> head convs + head_out
// timm 0.4.12 vit_base_resnet50_384 (vit.py:483) is restated per SURVEY.md A.2.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/dptx.h"
#include "kernels.h"

using namespace dptx;

namespace {

constexpr int IMG = 384;
constexpr int S_TOK = 577;
constexpr int FEAT = 256;
const int STAGE_DEPTH[3] = {3, 4, 9};
const int STAGE_OUT[3] = {256, 512, 1024};
const int STAGE_STRIDE[3] = {1, 2, 2};

// ---------------------------------------------------------------- host 16-bit conversions
inline uint16_t f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                            // RNE
  return (uint16_t)(u >> 16);
}
inline uint16_t f32_to_fp16(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));
  if (x >= 0x477ff000u) {  // >= 65520 rounds to inf
    return (uint16_t)(sign | 0x7c00u);
  }
  if (x < 0x38800000u) {  // subnormal half (|f| < 2^-14)
    if (x < 0x33000000u) return (uint16_t)sign;  // < 2^-25 -> 0
    const int e = (int)(x >> 23);                // biased exponent
    uint32_t m = (x & 0x7fffffu) | 0x800000u;    // 24-bit significand
    const int shift = 126 - e;                   // 14..24: result = m >> shift with RNE (units of 2^-24)
    const uint32_t q = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u);
    const uint32_t half = 1u << (shift - 1);
    uint32_t r = q;
    if (rem > half || (rem == half && (q & 1u))) r++;
    return (uint16_t)(sign | r);
  }
  uint32_t r = x - 0x38000000u;  // rebias exponent (127-15)<<23
  const uint32_t rem = r & 0x1fffu;
  r >>= 13;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
  return (uint16_t)(sign | r);
}

inline float fp16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu, u;
  if (e == 0) {
    if (m == 0) u = sign;
    else {  // subnormal half: normalise
      int sh = 0;
      while (!(m & 0x400u)) { m <<= 1; ++sh; }
      u = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((m & 0x3ffu) << 13);
    }
  } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
  else u = sign | ((e + 112u) << 23) | (m << 13);
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// OCP e4m3 (fp8 "fn": bias 7, 3 mantissa bits, max 448, no infinities), round to nearest even, saturating
inline uint8_t f32_to_e4m3(float f) {
  if (f != f) return 0x7f;
  const uint8_t sign = std::signbit(f) ? 0x80 : 0;
  double a = std::fabs((double)f);
  if (a >= 448.0) return sign | 0x7e;
  if (a < std::ldexp(1.0, -10)) return sign;  // below half of the smallest subnormal (2^-9)
  int e;
  (void)std::frexp(a, &e);  // a = m * 2^e, m in [0.5, 1)
  e -= 1;                   // a = (1.xxx) * 2^e
  if (e < -6) {             // subnormal: multiples of 2^-9
    const int q = (int)std::nearbyint(std::ldexp(a, 9));
    return sign | (uint8_t)q;  // q == 8 is the smallest normal (exp field 1, mantissa 0): the same bit pattern
  }
  int m = (int)std::nearbyint((std::ldexp(a, -e) - 1.0) * 8.0);
  if (m == 8) { m = 0; ++e; }
  if (e > 8) return sign | 0x7e;
  return sign | (uint8_t)(((e + 7) << 3) | m);
}

inline float e4m3_to_f32(uint8_t v) {
  const int e = (v >> 3) & 0xf, m = v & 7;
  const double a = e == 0 ? std::ldexp((double)m, -9) : std::ldexp(1.0 + m / 8.0, e - 7);
  return (float)((v & 0x80) ? -a : a);
}

// --------------------------------------------------------------------------- weight spec
// R_DECONV: ConvTranspose2d [Cin, Cout, k, k] with kernel == stride, packed as the GEMM operand [(dy*k+dx)*Cout + co][ci];
// R_DECONV_BIAS: its bias, tiled k*k times so that the GEMM epilogue can add it per column
// R_DERIVED: fp32 vector that is not part of the state_dict but computed at pack time (LayerNorm fold: column sums of the
// folded qkv / fc1 weights)
enum Role { R_STDCONV, R_CONV, R_LINEAR, R_VEC, R_HEAD4, R_UNUSED, R_DECONV, R_DECONV_BIAS, R_DERIVED };
struct Spec {
  std::string key;
  std::vector<int64_t> shape;
  Role role;
};

void add(std::vector<Spec>& v, const std::string& k, std::vector<int64_t> s, Role r) { v.push_back({k, std::move(s), r}); }

// Mirrors omnidata_amd/weights.py:state_dict_spec (reference key names).
// `dual`: a second decoder (depth, 1 channel) under "depth.scratch.*" next to the normal one (3 channels) under
// "scratch.*"; both read the one shared encoder "pretrained.*" (SURVEY.md 8d config 5).
// backbone 0: vitb_rn50_384 (DPT-Hybrid); 1: vitl16_384 (DPT-Large; omnidata_amd/weights.py
// vitl16_state_dict_spec)
// fp8 dtype, DPTX_FLAG_FP8_VIT (round 6): per-output-channel inverse scales of the e4m3 copies of qkv / fc1 / fc2 of one block, and
// the column sums of the DEQUANTISED folded e4m3 weights (the LayerNorm fold's mean term must cancel against what the MFMA
// actually multiplies)
void add_vit_fp8(std::vector<Spec>& v, const std::string& p, int D_VIT, int D_MLP) {
  add(v, p + "attn.qkv.f8scale", {3 * D_VIT}, R_DERIVED);
  add(v, p + "mlp.fc1.f8scale", {D_MLP}, R_DERIVED);
  add(v, p + "mlp.fc2.f8scale", {D_VIT}, R_DERIVED);
  add(v, p + "attn.qkv.lnsum8", {3 * D_VIT}, R_DERIVED);
  add(v, p + "mlp.fc1.lnsum8", {D_MLP}, R_DERIVED);
}

std::vector<Spec> build_spec(int C, bool dual, int backbone) {
  std::vector<Spec> v;
  const std::string vp = "pretrained.model.";
  const int D_VIT = backbone == 1 ? 1024 : 768, D_MLP = 4 * D_VIT, depth = backbone == 1 ? 24 : 12;
  if (backbone == 1) {
    add(v, vp + "cls_token", {1, 1, D_VIT}, R_VEC);
    add(v, vp + "pos_embed", {1, S_TOK, D_VIT}, R_VEC);
    add(v, vp + "patch_embed.proj.weight", {D_VIT, 3, 16, 16}, R_LINEAR);  // flattened (c, ky, kx) = the patch matrix's k order
    add(v, vp + "patch_embed.proj.bias", {D_VIT}, R_VEC);
    for (int l = 0; l < depth; ++l) {
      const std::string p = vp + "blocks." + std::to_string(l) + ".";
      add(v, p + "norm1.weight", {D_VIT}, R_VEC);
      add(v, p + "norm1.bias", {D_VIT}, R_VEC);
      add(v, p + "attn.qkv.weight", {3 * D_VIT, D_VIT}, R_LINEAR);
      add(v, p + "attn.qkv.bias", {3 * D_VIT}, R_VEC);
      add(v, p + "attn.proj.weight", {D_VIT, D_VIT}, R_LINEAR);
      add(v, p + "attn.proj.bias", {D_VIT}, R_VEC);
      add(v, p + "norm2.weight", {D_VIT}, R_VEC);
      add(v, p + "norm2.bias", {D_VIT}, R_VEC);
      add(v, p + "mlp.fc1.weight", {D_MLP, D_VIT}, R_LINEAR);
      add(v, p + "mlp.fc1.bias", {D_MLP}, R_VEC);
      add(v, p + "mlp.fc2.weight", {D_VIT, D_MLP}, R_LINEAR);
      add(v, p + "mlp.fc2.bias", {D_VIT}, R_VEC);
      add(v, p + "attn.qkv.lnsum", {3 * D_VIT}, R_DERIVED);
      add(v, p + "mlp.fc1.lnsum", {D_MLP}, R_DERIVED);
      add_vit_fp8(v, p, D_VIT, D_MLP);
    }
    add(v, vp + "norm.weight", {D_VIT}, R_UNUSED);
    add(v, vp + "norm.bias", {D_VIT}, R_UNUSED);
    add(v, vp + "head.weight", {1000, D_VIT}, R_UNUSED);
    add(v, vp + "head.bias", {1000}, R_UNUSED);
    const int feats[4] = {256, 512, 1024, 1024};
    for (int n = 1; n <= 4; ++n) {
      const std::string p = "pretrained.act_postprocess" + std::to_string(n) + ".";
      const int f = feats[n - 1];
      add(v, p + "0.project.0.weight", {D_VIT, 2 * D_VIT}, R_LINEAR);
      add(v, p + "0.project.0.bias", {D_VIT}, R_VEC);
      add(v, p + "3.weight", {f, D_VIT, 1, 1}, R_CONV);
      add(v, p + "3.bias", {f}, R_VEC);
      if (n == 1) { add(v, p + "4.weight", {f, f, 4, 4}, R_DECONV); add(v, p + "4.bias", {f}, R_DECONV_BIAS); }
      if (n == 2) { add(v, p + "4.weight", {f, f, 2, 2}, R_DECONV); add(v, p + "4.bias", {f}, R_DECONV_BIAS); }
      if (n == 4) { add(v, p + "4.weight", {f, f, 3, 3}, R_CONV); add(v, p + "4.bias", {f}, R_VEC); }
    }
  } else {
  add(v, vp + "cls_token", {1, 1, D_VIT}, R_VEC);
  add(v, vp + "pos_embed", {1, S_TOK, D_VIT}, R_VEC);
  const std::string bp = vp + "patch_embed.backbone.";
  add(v, bp + "stem.conv.weight", {64, 3, 7, 7}, R_STDCONV);
  add(v, bp + "stem.norm.weight", {64}, R_VEC);
  add(v, bp + "stem.norm.bias", {64}, R_VEC);
  int cin = 64;
  for (int s = 0; s < 3; ++s) {
    const int cout = STAGE_OUT[s], mid = cout / 4;
    for (int b = 0; b < STAGE_DEPTH[s]; ++b) {
      const std::string p = bp + "stages." + std::to_string(s) + ".blocks." + std::to_string(b) + ".";
      if (b == 0) {
        add(v, p + "downsample.conv.weight", {cout, cin, 1, 1}, R_STDCONV);
        add(v, p + "downsample.norm.weight", {cout}, R_VEC);
        add(v, p + "downsample.norm.bias", {cout}, R_VEC);
      }
      add(v, p + "conv1.weight", {mid, cin, 1, 1}, R_STDCONV);
      add(v, p + "norm1.weight", {mid}, R_VEC);
      add(v, p + "norm1.bias", {mid}, R_VEC);
      add(v, p + "conv2.weight", {mid, mid, 3, 3}, R_STDCONV);
      add(v, p + "norm2.weight", {mid}, R_VEC);
      add(v, p + "norm2.bias", {mid}, R_VEC);
      add(v, p + "conv3.weight", {cout, mid, 1, 1}, R_STDCONV);
      add(v, p + "norm3.weight", {cout}, R_VEC);
      add(v, p + "norm3.bias", {cout}, R_VEC);
      cin = cout;
    }
  }
  add(v, vp + "patch_embed.proj.weight", {D_VIT, 1024, 1, 1}, R_CONV);
  add(v, vp + "patch_embed.proj.bias", {D_VIT}, R_VEC);
  for (int l = 0; l < depth; ++l) {
    const std::string p = vp + "blocks." + std::to_string(l) + ".";
    add(v, p + "norm1.weight", {D_VIT}, R_VEC);
    add(v, p + "norm1.bias", {D_VIT}, R_VEC);
    add(v, p + "attn.qkv.weight", {3 * D_VIT, D_VIT}, R_LINEAR);
    add(v, p + "attn.qkv.bias", {3 * D_VIT}, R_VEC);
    add(v, p + "attn.proj.weight", {D_VIT, D_VIT}, R_LINEAR);
    add(v, p + "attn.proj.bias", {D_VIT}, R_VEC);
    add(v, p + "norm2.weight", {D_VIT}, R_VEC);
    add(v, p + "norm2.bias", {D_VIT}, R_VEC);
    add(v, p + "mlp.fc1.weight", {D_MLP, D_VIT}, R_LINEAR);
    add(v, p + "mlp.fc1.bias", {D_MLP}, R_VEC);
    add(v, p + "mlp.fc2.weight", {D_VIT, D_MLP}, R_LINEAR);
    add(v, p + "mlp.fc2.bias", {D_VIT}, R_VEC);
    add(v, p + "attn.qkv.lnsum", {3 * D_VIT}, R_DERIVED);
    add(v, p + "mlp.fc1.lnsum", {D_MLP}, R_DERIVED);
    add_vit_fp8(v, p, D_VIT, D_MLP);
  }
  add(v, vp + "norm.weight", {D_VIT}, R_UNUSED);
  add(v, vp + "norm.bias", {D_VIT}, R_UNUSED);
  add(v, vp + "head.weight", {1000, D_VIT}, R_UNUSED);
  add(v, vp + "head.bias", {1000}, R_UNUSED);
  for (int n = 3; n <= 4; ++n) {
    const std::string p = "pretrained.act_postprocess" + std::to_string(n) + ".";
    add(v, p + "0.project.0.weight", {D_VIT, 2 * D_VIT}, R_LINEAR);
    add(v, p + "0.project.0.bias", {D_VIT}, R_VEC);
    add(v, p + "3.weight", {D_VIT, D_VIT, 1, 1}, R_CONV);
    add(v, p + "3.bias", {D_VIT}, R_VEC);
    if (n == 4) {
      add(v, p + "4.weight", {D_VIT, D_VIT, 3, 3}, R_CONV);
      add(v, p + "4.bias", {D_VIT}, R_VEC);
    }
  }
  }  // backbone
  // "<conv>.f8scale": per-output-channel inverse of the power-of-two scale the e4m3 copy of the weight was quantised with
  // (fp8 dtype; derived at pack time, zero otherwise)
  auto decoder = [&](const std::string& pre, int ch) {
    const int rn_in[4] = {256, 512, D_VIT, D_VIT};
    for (int i = 1; i <= 4; ++i) add(v, pre + "scratch.layer" + std::to_string(i) + "_rn.weight", {FEAT, rn_in[i - 1], 3, 3}, R_CONV);
    for (int i = 1; i <= 4; ++i) {
      const std::string p = pre + "scratch.refinenet" + std::to_string(i) + ".";
      add(v, p + "out_conv.weight", {FEAT, FEAT, 1, 1}, R_CONV);
      add(v, p + "out_conv.bias", {FEAT}, R_VEC);
      add(v, p + "out_conv.f8scale", {FEAT}, R_DERIVED);
      for (int u = 1; u <= 2; ++u)
        for (int c = 1; c <= 2; ++c) {
          const bool unused = (i == 4 && u == 1);  // blocks.py:329-333: resConfUnit1 needs two inputs
          const std::string q = p + "resConfUnit" + std::to_string(u) + ".conv" + std::to_string(c) + ".";
          add(v, q + "weight", {FEAT, FEAT, 3, 3}, unused ? R_UNUSED : R_CONV);
          add(v, q + "bias", {FEAT}, unused ? R_UNUSED : R_VEC);
          if (!unused) add(v, q + "f8scale", {FEAT}, R_DERIVED);
        }
    }
    add(v, pre + "scratch.output_conv.0.weight", {FEAT / 2, FEAT, 3, 3}, R_CONV);
    add(v, pre + "scratch.output_conv.0.bias", {FEAT / 2}, R_VEC);
    add(v, pre + "scratch.output_conv.0.f8scale", {FEAT / 2}, R_DERIVED);
    add(v, pre + "scratch.output_conv.2.weight", {32, FEAT / 2, 3, 3}, R_CONV);
    add(v, pre + "scratch.output_conv.2.bias", {32}, R_VEC);
    add(v, pre + "scratch.output_conv.4.weight", {ch, 32, 1, 1}, R_HEAD4);
    add(v, pre + "scratch.output_conv.4.bias", {ch}, R_VEC);
  };
  decoder("", dual ? 3 : C);
  if (dual) decoder("depth.", 1);
  return v;
}

inline size_t numel(const std::vector<int64_t>& s) {
  size_t n = 1;
  for (auto d : s) n *= (size_t)d;
  return n;
}
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

constexpr int STEM_K = 176;  // stem K axis ordered (c, ky, kx) with kx padded 7 -> 8 (168) and rounded up to 11 MFMA k-steps

size_t packed_entry_bytes(const Spec& s) {
  switch (s.role) {
    case R_STDCONV:
      if (s.shape[3] == 7) return (size_t)s.shape[0] * STEM_K * 2;
      return numel(s.shape) * 2;
    case R_CONV:
    case R_DECONV:
    case R_LINEAR: return numel(s.shape) * 2;
    case R_VEC:
    case R_DERIVED:
    case R_HEAD4: return numel(s.shape) * 4;
    case R_DECONV_BIAS: return numel(s.shape) * 4 * 16;  // room for the 4x4 case (the 2x2 one uses a quarter)
    default: return 0;
  }
}

// The packed blob starts with a 256-byte header that names the layout it was packed for: ranks (dptx_import_packed_device) and
// the on-disk cache refuse a blob whose packing differs from what this handle would produce itself -- e.g. a rank whose
// environment switched the LayerNorm fold off would otherwise multiply folded qkv / fc1 weights as if they were plain.
constexpr size_t BLOB_HEADER = 256;
struct BlobHeader {
  char magic[8];          // "DPTXBLOB"
  uint32_t version;       // layout version of this header / of the packing code
  int32_t dtype, backbone, dual_task, num_channels, ln_fold, ws_form;
  float ws_eps;
  uint64_t packed_single, packed_bytes;
};
static_assert(sizeof(BlobHeader) <= BLOB_HEADER, "header fits its slot");
constexpr uint32_t BLOB_VERSION = 5;   // 5: fp8 entries of the ViT linears (round 6)

struct Buf {  // arena slice: `off` in the whole-batch plan, `off2` in the half-batch plan (two sub-batches on two streams)
  size_t off = 0, bytes = 0, off2 = 0;
};

struct TapInfo {
  const void* ptr;
  int64_t shape[4];
  bool fp32;
  int mode;  // kernel mode of the group that produced it (single-pass groups of a MIXED engine write the hi plane only)
};

}  // namespace

struct dptx_engine {
  dptx_config cfg;
  int backbone = 0;                              // 0 vitb_rn50_384, 1 vitl16_384
  int dv = 768, dm = 3072, nh = 12, depth = 12;  // ViT width, MLP width, heads, blocks
  std::vector<Spec> spec;
  std::unordered_map<std::string, size_t> spec_index;
  std::unordered_map<std::string, size_t> packed_off;  // byte offset inside the blob
  size_t packed_bytes = 0;    // total blob bytes (hi blob [+ lo blob in bf16x3 mode])
  size_t packed_single = 0;   // bytes of one plane of the blob
  size_t arena_single = 0;    // bytes of one plane of the arena
  Planes pl{0, 0};
  std::map<std::string, std::vector<float>> staged;  // fp32 tensors loaded so far
  std::vector<uint8_t> host_blob;
  bool finalized = false;  // host blob valid
  bool device_ready = false;
  char* d_blob = nullptr;
  // The packed blob's allocation is reference-counted: dptx_share_packed hands a second handle the same memory, and whichever
  // handle is destroyed LAST frees it -- destroying the handle that loaded the weights while others still read them is legal.
  std::shared_ptr<void> blob_hold;
  bool blob_shared = false;   // this handle did not load the blob it reads (dptx_share_packed): it never writes it
  char* d_arena = nullptr;
  size_t arena_bytes = 0;
  std::string err;
  std::map<std::string, TapInfo> taps;
  bool taps_on = false;
  // kernel mode (common.h MODE_*) of a layer group: the engine dtype, or -- MIXED -- fp16x3 / fp16 per dptx_config.x3_groups
  int mode_of(int group) const {
    if (cfg.dtype == DPTX_DTYPE_FP8) return MODE_BF16;  // fp8 GEMMs are launched explicitly (Run::conv), the rest is bf16
    if (cfg.dtype != DPTX_DTYPE_MIXED) return cfg.dtype;
    return (cfg.x3_groups & group) ? MODE_FP16X3 : MODE_FP16;
  }
  bool fp8() const { return cfg.dtype == DPTX_DTYPE_FP8; }
  // Per-LAYER precision inside the decoder groups (RN / FUSION / HEAD) of the MIXED dtype: conv key (reference name without
  // a "depth." prefix) -> 1 (one MFMA per product on the hi planes), 3 (hi/lo planes, three MFMAs) or 2 (hi/lo planes of the
  // weights, hi plane of the activations: a_hi w_hi + a_hi w_lo); keys that are not listed follow their group's bit in x3_groups.  The default table (dptx_create) is the cheapest assignment found by
  // oracle/precision_layers.py that keeps the emulated deviation from the fp32 forward where the all-3-MFMA decoder has it.
  std::unordered_map<std::string, int> layer_prec;
  bool mixed() const { return cfg.dtype == DPTX_DTYPE_MIXED; }
  int layer_mfmas(const std::string& wkey, int group) const {
    const std::string k = wkey.compare(0, 6, "depth.") == 0 ? wkey.substr(6) : wkey;
    auto it = layer_prec.find(k);
    if (it != layer_prec.end()) return it->second;
    return (cfg.x3_groups & group) ? 3 : 1;
  }
  // hi/lo-plane kernel (2 or 3 MFMAs per product: the weights' lo plane is used) / the layer reads the lo plane of its INPUT
  bool layer_x3(const std::string& wkey, int group) const { return layer_mfmas(wkey, group) >= 2; }
  bool layer_reads_lo(const std::string& wkey, int group) const { return layer_mfmas(wkey, group) == 3; }
  // LayerNorm of the ViT blocks folded into the qkv / fc1 GEMMs (include/dptx.h DPTX_FLAG_NO_LN_FOLD): the packed qkv / fc1
  // weights and biases are then the folded ones, so this is fixed at dptx_create
  bool ln_fold = false;
  // 16-bit token stream (include/dptx.h DPTX_FLAG_FP32_STREAM): with the fold in place the single-pass dtypes need the
  // residual stream of the ViT blocks only as the 16-bit tensor the qkv / fc1 GEMMs multiply -- Hn IS the stream, the
  // proj / fc2 epilogues read and write 2 + 2 bytes per element instead of 4 + 4 + 2, and no fp32 copy exists
  bool stream16 = false;
  // fp8: the second plane of the arena / blob holds the e4m3 copies (byte offset off / 2 inside it) instead of lo planes
  bool two_planes() const {
    return cfg.dtype == DPTX_DTYPE_BF16X3 || cfg.dtype == DPTX_DTYPE_FP16X3 || cfg.dtype == DPTX_DTYPE_MIXED || fp8();
  }
  bool bf16_storage() const { return cfg.dtype == DPTX_DTYPE_BF16 || cfg.dtype == DPTX_DTYPE_BF16X3 || fp8(); }
  // convolutions that CAN run on e4m3 operands in the fp8 dtype (their weights get an e4m3 copy at pack time): the decoder's
  // RCU convs, out_conv and the first head conv ...
  static bool fp8_weight(const std::string& key) {
    return key.find("scratch.") != std::string::npos && key.size() > 7 && key.compare(key.size() - 7, 7, ".weight") == 0 &&
           (key.find("resConfUnit") != std::string::npos || key.find("out_conv") != std::string::npos ||
            key.find("output_conv.0.") != std::string::npos);
  }
  // ... and the ones that actually do (include/dptx.h DPTX_FLAG_FP8_ALL): by default only the six convolutions of the
  // resConfUnit1 blocks of refinenet1..3 -- 14.3 GMAC per decoder whose e4m3 rounding moves the output least
  // (oracle/fp8_layers.py: the set costs <= 1.5 deg of mean angular error on either synthetic weight family, so that the
  // mode stays within 2 x the bf16 engine's error); with the flag all 19 (round 3's mode: 7.5 - 9 deg, a lossy throughput mode)
  bool fp8_all = false;
  // DPTX_FLAG_FP8_VIT (round 6): qkv / fc1 / fc2 of every transformer block on e4m3 operands as well -- 45 of the forward's
  // 127.6 GMAC (proj stays bf16: its operand is the attention kernel's output, which has no e4m3 copy).  oracle/fp8_vit.py: with
  // per-output-channel weight scales and ONE calibrated scale per activation tensor these layers cost 2.1-2.7 deg of mean
  // angular error on the default weight family and 1.2-1.9 deg on the trained-like one, about what the six default decoder
  // convolutions cost -- not the "destroyed output" that rounds 3-5 assumed.
  bool fp8_vit = false;
  static bool fp8_vit_weight(const std::string& key) {
    auto ends = [&](const char* suf) { const size_t n = strlen(suf); return key.size() >= n && key.compare(key.size() - n, n, suf) == 0; };
    return key.find("pretrained.model.blocks.") != std::string::npos &&
           (ends("attn.qkv.weight") || ends("mlp.fc1.weight") || ends("mlp.fc2.weight"));
  }
  bool fp8_use(const std::string& key) const {
    if (!fp8()) return false;
    if (fp8_vit_weight(key)) return fp8_vit && ln_fold && stream16;
    if (!fp8_weight(key)) return false;
    return fp8_all || key.find("resConfUnit1.") != std::string::npos;
  }
  const void* w8(const std::string& key) const { return d_blob + packed_single + packed_off.at(key) / 2; }
  // per-output-channel inverse weight scales of an fp8 conv ("....weight" -> "....f8scale" entry of the blob)
  const float* wscale(const std::string& key) const { return f(key.substr(0, key.size() - 6) + "f8scale"); }
  // fp8 activation scales: every tensor that has an e4m3 copy gets a power-of-two scale s (the copy holds e4m3(x * s), the
  // consuming conv multiplies its accumulators by 1 / s), indexed by the order in which a forward produces the copies.
  // 1.0 until dptx_calibrate_fp8 has measured the tensors' max |x| on a calibration batch (slot -> max |x| in act_amax).
  static constexpr int MAX_Q8 = 128;
  std::vector<float> act_scale = std::vector<float>(MAX_Q8, 1.0f);
  std::vector<float> act_amax = std::vector<float>(MAX_Q8, 0.0f);
  int n_q8 = 0;                  // slots used by the last forward
  bool calibrating = false, calibrated = false;
  unsigned* d_amax = nullptr;    // [MAX_Q8] float bits (calibration forwards only)
  // range check of the fp16-plane dtypes (include/dptx.h dptx_range_status): sticky device flag, set by a scan of H0
  unsigned* d_range = nullptr;
  bool range_check() const {
    return (cfg.dtype == DPTX_DTYPE_FP16 || cfg.dtype == DPTX_DTYPE_FP16X3 || cfg.dtype == DPTX_DTYPE_MIXED) &&
           !(cfg.flags & DPTX_FLAG_NO_RANGE_CHECK);
  }
  void* q8(const void* p) const { return d_arena + arena_single + ((const char*)p - d_arena) / 2; }  // e4m3 copy of an arena tensor
  // fused head tail (head.hip): no stage taps wanted; a 3-MFMA second head conv has its fused form for fp16 planes only
  // (head_tail_x3_kernel: "mixed", fp16x3); DPTX_HEAD_FUSED=0 keeps the three launches (DPTX_HEAD_FUSED_X3=0: for the x3 form only)
  bool head_fused(bool conv2_x3) const {
    static int env = -1, env3 = -1;
    if (env < 0) { const char* t = getenv("DPTX_HEAD_FUSED"); env = (t && t[0] == '0') ? 0 : 1; }
    if (env3 < 0) { const char* t = getenv("DPTX_HEAD_FUSED_X3"); env3 = (t && t[0] == '0') ? 0 : 1; }
    if (env != 1 || taps_on) return false;
    if (!conv2_x3) return true;
    return env3 == 1 && (cfg.dtype == DPTX_DTYPE_MIXED || cfg.dtype == DPTX_DTYPE_FP16X3);
  }
  size_t tok_tap_stride = 0;      // floats per token-stream snapshot
  float* d_tok_taps = nullptr;  // [13][max_batch*(max tokens)*768] fp32 copies of the token stream (taps_on)
  int64_t launches = 0;
  double exec_macs = 0.0;
  int last_batch = 0;
  int last_regions = 1;  // sub-batch regions the last forward used (1: the whole-batch plan)
  // debug (dptx_debug_set_launch_sums): word sums of the ViT buffers after every launch of the ViT blocks, [launch][5] uint64
  unsigned long long* d_lsums = nullptr;
  int lsum_cap = 0, lsum_n = 0;
  // optional per-launch timing (one event after every launch; kernels are serialized on the stream)
  bool profiling = false;
  std::vector<hipEvent_t> events;
  std::vector<int> event_cat;
  std::vector<std::string> event_name;
  double cat_ms[4] = {0, 0, 0, 0};
  int64_t cat_launches[4] = {0, 0, 0, 0};
  double cat_macs[4] = {0, 0, 0, 0};

  // two sub-batches on two internal streams (MFMA-bound and HBM-bound launches of the two halves overlap, tails fill)
  static constexpr int MAX_STREAMS = 4;
  int n_streams = 2;              // cfg.streams (0 = auto: 2 streams exist, used only when measured faster); 1 = caller's stream
  // cfg.streams == 0 and no DPTX_STREAMS: the two-half schedule is used only after dptx_tune_schedule measured it >= 3 % faster
  bool auto_streams = false, split_tuned = false, split_on = false;
  float tune_ms_single = 0.f, tune_ms_split = 0.f;
  int force_schedule = -1;        // dptx_tune_schedule: 0 / 1 while it times a candidate
  int half_batch = 0;             // images per sub-batch region (ceil(max_batch / n_streams))
  size_t half_region = 0;         // bytes of one sub-batch region (n_streams of them fit in one arena plane)
  hipStream_t sub_stream[MAX_STREAMS] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[MAX_STREAMS] = {nullptr, nullptr, nullptr, nullptr};
  // arena slices
  int max_h = 384, max_w = 384;   // largest supported input (cfg.max_height/max_width; 0 = 384)
  Buf pos_alt;
  Buf sraw, stem, S[3], T1, T2, PA, PB, DS, part[4], X, Hn, QKV, AO, F1, R3, R4, L3, T4, L4, clsb, lrn[4], tA, tB,
      tC, P[4], H0, H0U, H1, lnst;

  int fail(int code, const std::string& m) {
    err = m;
    return code;
  }
  const void* w(const std::string& key) const { return d_blob + packed_off.at(key); }
  const float* f(const std::string& key) const { return (const float*)(d_blob + packed_off.at(key)); }
  BlobHeader blob_header() const {
    BlobHeader hd;
    memset(&hd, 0, sizeof hd);
    memcpy(hd.magic, "DPTXBLOB", 8);
    hd.version = BLOB_VERSION;
    hd.dtype = cfg.dtype; hd.backbone = backbone; hd.dual_task = cfg.dual_task; hd.num_channels = cfg.num_channels;
    hd.ln_fold = ln_fold ? 1 : 0; hd.ws_form = cfg.ws_form; hd.ws_eps = cfg.ws_eps;
    hd.packed_single = packed_single; hd.packed_bytes = packed_bytes;
    return hd;
  }
};

namespace {

// Makes cfg.device_id current for the duration of a C entry point and restores the caller's device afterwards (a process
// that drives several GPUs keeps its own current device; torch.cuda.current_device() is not switched behind its back).
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) err = hipSetDevice(dev);
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};

#define HIPCHK(e, call)                                                                   \
  do {                                                                                    \
    hipError_t _r = (call);                                                               \
    if (_r != hipSuccess) return (e)->fail(DPTX_E_HIP, std::string(#call) + ": " + hipGetErrorString(_r)); \
  } while (0)

// Bump-allocates every activation buffer for Bn images; half = false fills Buf::off (one run over the whole batch),
// half = true fills Buf::off2 (layout of ONE of the two half-batch regions).  Returns the bytes used.
size_t plan_arena_for(dptx_engine* e, size_t B, bool half) {
  // every buffer scales with the pixel count of the largest supported input (H, W multiples of 32); P = H*W
  const size_t P = (size_t)e->max_h * e->max_w;
  const size_t p2 = P / 4, p4 = P / 16, p8 = P / 64, p16 = P / 256, p32 = P / 1024, S = p16 + 1;
  const size_t DV = (size_t)e->dv;
  size_t off = 0;
  auto take = [&](Buf& b, size_t elems, size_t esz) {
    const size_t bytes = align_up(elems * esz, 256);
    if (half) b.off2 = off; else { b.off = off; b.bytes = bytes; }
    off += bytes;
  };
  take(e->sraw, B * p2 * 64, 2);
  take(e->stem, B * p4 * 64, 2);
  take(e->S[0], B * p4 * 256, 2);
  take(e->S[1], B * p8 * 512, 2);
  take(e->S[2], B * p16 * 1024, 2);
  take(e->T1, B * p4 * 128, 2);  // largest conv1 output: stage1 block0 (128 ch @ 1/4 resolution)
  take(e->T2, B * p4 * 64, 2);   // largest conv2 output: stage0 (64 ch @ 1/4 resolution)
  take(e->PA, B * p4 * 256, 2);
  take(e->PB, B * p4 * 256, 2);
  take(e->DS, B * p4 * 256, 2);
  // GroupNorm partial records (32 groups x float2): p2/256 chunks for the stem, p4/32 MFMA row blocks for a stage conv
  for (int i = 0; i < 4; ++i) take(e->part[i], B * (std::max(p2 / 256, p4 / 32) + 64) * 64, 4);
  take(e->X, B * S * DV, 4);
  take(e->lnst, B * S * 8 * 2, 4);  // LayerNorm fold: (sum, sum of squares) per token row and 128-column block (<= 8 blocks)
  take(e->Hn, B * S * DV, 2);
  take(e->QKV, B * S * 3 * DV, 2);
  take(e->AO, B * S * DV, 2);
  take(e->F1, B * S * (size_t)e->dm, 2);
  take(e->R3, B * p16 * DV, 2);
  take(e->R4, B * p16 * DV, 2);
  take(e->L3, B * p16 * DV, 2);
  take(e->T4, B * p16 * DV, 2);
  take(e->L4, B * p32 * DV, 2);
  take(e->clsb, B * DV, 4);
  take(e->pos_alt, S * DV, 4);  // bilinearly resized pos_embed for inputs other than 384x384
  const size_t rn_px[4] = {p4, p8, p16, p32};
  for (int i = 0; i < 4; ++i) take(e->lrn[i], B * rn_px[i] * FEAT, 2);
  take(e->tA, B * p4 * FEAT, 2);
  take(e->tB, B * p4 * FEAT, 2);
  take(e->tC, B * p4 * FEAT, 2);
  const size_t p_px[4] = {p2, p4, p8, p16};  // P[0]=path_1 (1/2 res) ... P[3]=path_4 (1/16 res)
  for (int i = 0; i < 4; ++i) take(e->P[i], B * p_px[i] * FEAT, 2);
  take(e->H0, B * p2 * 128, 2);
  take(e->H0U, B * P * 128, 2);
  take(e->H1, B * P * 32, 2);
  return off;
}

void plan_arena(dptx_engine* e) {
  const size_t full = plan_arena_for(e, (size_t)e->cfg.max_batch, false);
  const int ns = e->n_streams > 1 ? e->n_streams : 2;
  e->half_batch = (e->cfg.max_batch + ns - 1) / ns;
  e->half_region = align_up(plan_arena_for(e, (size_t)e->half_batch, true), 256);
  const size_t off = std::max(full, (size_t)ns * e->half_region);
  e->arena_single = off;
  const int npl = e->two_planes() ? 2 : 1;
  e->arena_bytes = off * npl;
  e->pl.act = npl == 2 ? (long long)(off / 2) : 0;
}

// ------------------------------------------------------------------------------- packing
int pack_host(dptx_engine* e) {
  std::string missing;
  for (const auto& s : e->spec)
    if (s.role != R_UNUSED && s.role != R_DERIVED && !e->staged.count(s.key)) missing += (missing.empty() ? "" : ", ") + s.key;
  if (!missing.empty()) return e->fail(DPTX_E_KEY, "missing tensors (strict load): " + missing);
  e->host_blob.assign(e->packed_bytes, 0);
  {
    const BlobHeader hd = e->blob_header();
    memcpy(e->host_blob.data(), &hd, sizeof hd);
  }
  const bool bf = e->bf16_storage();
  const bool x3 = e->two_planes() && !e->fp8();
  const size_t lo_elems = e->packed_single / 2;  // uint16 distance hi -> lo plane
  auto bf16_to_f32 = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
  // writes element i of a 16-bit tensor (and its lo plane in bf16x3 mode)
  auto put = [&](uint16_t* d16, size_t i, float x) {
    if (!bf) {
      const uint16_t hi = f32_to_fp16(x);
      d16[i] = hi;
      if (x3) d16[i + lo_elems] = f32_to_fp16(x - fp16_to_f32(hi));
      return;
    }
    const uint16_t hi = f32_to_bf16(x);
    d16[i] = hi;
    if (x3) d16[i + lo_elems] = f32_to_bf16(x - bf16_to_f32(hi));
  };
  // value the MFMA sees for a packed single-plane operand element
  auto r16 = [&](float x) { return bf ? bf16_to_f32(f32_to_bf16(x)) : fp16_to_f32(f32_to_fp16(x)); };
  auto ends_with = [](const std::string& k, const char* suf) {
    const size_t n = strlen(suf);
    return k.size() >= n && k.compare(k.size() - n, n, suf) == 0;
  };
  for (const auto& s : e->spec) {
    if (s.role == R_UNUSED || s.role == R_DERIVED) continue;
    const std::vector<float>& src = e->staged.at(s.key);
    uint8_t* dst = e->host_blob.data() + e->packed_off.at(s.key);
    // LayerNorm fold (kernels.h GemmParams::ln_stats): qkv / fc1 of every block become W' = W diag(gamma), b' = b + W beta
    // and the column sums of the ROUNDED W' go to the derived "...lnsum" entry
    const bool fold_w = e->ln_fold && s.role == R_LINEAR && (ends_with(s.key, "attn.qkv.weight") || ends_with(s.key, "mlp.fc1.weight"));
    const bool fold_b = e->ln_fold && s.role == R_VEC && (ends_with(s.key, "attn.qkv.bias") || ends_with(s.key, "mlp.fc1.bias"));
    if (fold_w || fold_b) {
      const bool qkv = s.key.find("attn.qkv") != std::string::npos;
      const std::string blk = s.key.substr(0, s.key.find(qkv ? "attn.qkv" : "mlp.fc1"));
      const std::string wkey = blk + (qkv ? "attn.qkv.weight" : "mlp.fc1.weight");
      const std::vector<float>& W = e->staged.at(wkey);
      const std::vector<float>& gamma = e->staged.at(blk + (qkv ? "norm1.weight" : "norm2.weight"));
      const std::vector<float>& beta = e->staged.at(blk + (qkv ? "norm1.bias" : "norm2.bias"));
      const size_t K = gamma.size(), N = W.size() / K;
      if (fold_b) {
        float* d32 = (float*)dst;
        for (size_t n = 0; n < N; ++n) {
          double acc = src[n];
          for (size_t k = 0; k < K; ++k) acc += (double)W[n * K + k] * (double)beta[k];
          d32[n] = (float)acc;
        }
      } else {
        uint16_t* d16 = (uint16_t*)dst;
        float* cs = (float*)(e->host_blob.data() + e->packed_off.at(blk + (qkv ? "attn.qkv.lnsum" : "mlp.fc1.lnsum")));
        for (size_t n = 0; n < N; ++n) {
          double acc = 0.0;
          for (size_t k = 0; k < K; ++k) {
            const float wf = W[n * K + k] * gamma[k];
            put(d16, n * K + k, wf);
            acc += (double)r16(wf);
          }
          cs[n] = (float)acc;
        }
      }
      continue;
    }
    if (s.role == R_VEC || s.role == R_HEAD4) {
      memcpy(dst, src.data(), src.size() * 4);
      continue;
    }
    if (s.role == R_DECONV_BIAS) {  // bias[co] for every one of the (up to 16) taps: column (tap*Cout + co) of the GEMM
      for (int t = 0; t < 16; ++t) memcpy(dst + (size_t)t * src.size() * 4, src.data(), src.size() * 4);
      continue;
    }
    uint16_t* d16 = (uint16_t*)dst;
    if (s.role == R_LINEAR) {
      for (size_t i = 0; i < src.size(); ++i) put(d16, i, src[i]);
      continue;
    }
    if (s.role == R_DECONV) {  // [Cin][Cout][k][k] -> GEMM operand W[(dy*k + dx)*Cout + co][ci]
      const int CI = (int)s.shape[0], CO = (int)s.shape[1], K = (int)s.shape[2];
      for (int ci = 0; ci < CI; ++ci)
        for (int co = 0; co < CO; ++co)
          for (int dy = 0; dy < K; ++dy)
            for (int dx = 0; dx < K; ++dx)
              put(d16, ((size_t)(dy * K + dx) * CO + co) * CI + ci, src[(((size_t)ci * CO + co) * K + dy) * K + dx]);
      continue;
    }
    // convolution OIHW -> [O][kh][kw][I]; StdConv2dSame weights are standardised first
    const int O = (int)s.shape[0], I = (int)s.shape[1], KH = (int)s.shape[2], KW = (int)s.shape[3];
    const size_t per_o = (size_t)I * KH * KW;
    const bool stem = (KH == 7);
    const size_t out_k = stem ? STEM_K : per_o;
    for (int o = 0; o < O; ++o) {
      const float* wo = src.data() + (size_t)o * per_o;
      double mean = 0.0, scale = 1.0;
      if (s.role == R_STDCONV) {
        double sum = 0.0;
        for (size_t i = 0; i < per_o; ++i) sum += wo[i];
        mean = sum / (double)per_o;
        double var = 0.0;
        for (size_t i = 0; i < per_o; ++i) { const double t = wo[i] - mean; var += t * t; }
        var /= (double)per_o;  // biased, as torch.std_mean(unbiased=False)
        scale = e->cfg.ws_form == 0 ? 1.0 / (std::sqrt(var) + (double)e->cfg.ws_eps)
                                    : 1.0 / std::sqrt(var + (double)e->cfg.ws_eps);
      }
      for (int ky = 0; ky < KH; ++ky)
        for (int kx = 0; kx < KW; ++kx)
          for (int i = 0; i < I; ++i) {
            const double wv = ((double)wo[((size_t)i * KH + ky) * KW + kx] - mean) * scale;
            // stem (stem.hip): k = (c*7 + ky)*8 + kx; every other conv: k = (ky*KW + kx)*I + c
            const size_t kidx = stem ? ((size_t)i * 7 + ky) * 8 + kx : ((size_t)ky * KW + kx) * I + i;
            put(d16, (size_t)o * out_k + kidx, (float)wv);
          }
    }
  }
  if (e->fp8()) {
    // e4m3 copies of the fp8 layers' weights (same [O][kh][kw][I] order as the bf16 copy, one byte per element, second
    // plane of the blob), quantised per OUTPUT CHANNEL after a power-of-two scale that puts the channel's max |w| into
    // (224, 448] (e4m3 tops out at 448); the inverses go to the layer's "f8scale" vector, which the
    // GEMM epilogue multiplies the accumulators with.
    for (size_t si = 0; si < e->spec.size(); ++si) {
      const Spec& sp = e->spec[si];
      if (sp.role != R_CONV || !dptx_engine::fp8_weight(sp.key)) continue;
      const size_t off = e->packed_off.at(sp.key);
      const std::vector<float>& src = e->staged.at(sp.key);
      const int O = (int)sp.shape[0], I = (int)sp.shape[1], KH = (int)sp.shape[2], KW = (int)sp.shape[3];
      const size_t per_o = (size_t)I * KH * KW;
      uint8_t* d8 = e->host_blob.data() + e->packed_single + off / 2;
      float* inv = (float*)(e->host_blob.data() + e->packed_off.at(sp.key.substr(0, sp.key.size() - 6) + "f8scale"));
      for (int o = 0; o < O; ++o) {
        float mx = 0.f;
        for (size_t i = 0; i < per_o; ++i) mx = std::max(mx, std::fabs(src[(size_t)o * per_o + i]));
        const int k = mx > 0.f ? (int)std::floor(std::log2(448.0 / (double)mx)) : 0;
        const float sc = std::ldexp(1.0f, k);
        for (int ky = 0; ky < KH; ++ky)
          for (int kx = 0; kx < KW; ++kx)
            for (int i = 0; i < I; ++i)
              d8[(size_t)o * per_o + ((size_t)ky * KW + kx) * I + i] = f32_to_e4m3(src[(((size_t)o * I + i) * KH + ky) * KW + kx] * sc);
        inv[o] = 1.0f / sc;
      }
    }
  }
  if (e->fp8()) {
    // e4m3 copies of qkv / fc1 / fc2 (DPTX_FLAG_FP8_VIT decides whether they are USED; they are always packed, so that the
    // blob's layout does not depend on the flag): the weight the bf16 GEMM multiplies -- folded with the LayerNorm's gamma for
    // qkv / fc1 -- per OUTPUT CHANNEL after a power-of-two scale into (224, 448]; "lnsum8" = column sums of the dequantised copy
    for (size_t si = 0; si < e->spec.size(); ++si) {
      const Spec& sp = e->spec[si];
      if (sp.role != R_LINEAR || !dptx_engine::fp8_vit_weight(sp.key)) continue;
      const std::vector<float>& W = e->staged.at(sp.key);
      const size_t N = (size_t)sp.shape[0], K = (size_t)sp.shape[1];
      const bool fc2 = ends_with(sp.key, "mlp.fc2.weight"), qkv = ends_with(sp.key, "attn.qkv.weight");
      const std::string stem_key = sp.key.substr(0, sp.key.size() - 6);   // "....attn.qkv." / "....mlp.fc1." / "....mlp.fc2."
      const std::string blk = sp.key.substr(0, sp.key.find(qkv ? "attn.qkv" : (fc2 ? "mlp.fc2" : "mlp.fc1")));
      const std::vector<float>* gamma = (!fc2 && e->ln_fold) ? &e->staged.at(blk + (qkv ? "norm1.weight" : "norm2.weight")) : nullptr;
      uint8_t* d8 = e->host_blob.data() + e->packed_single + e->packed_off.at(sp.key) / 2;
      float* inv = (float*)(e->host_blob.data() + e->packed_off.at(stem_key + "f8scale"));
      float* cs8 = fc2 ? nullptr : (float*)(e->host_blob.data() + e->packed_off.at(stem_key + "lnsum8"));
      for (size_t n = 0; n < N; ++n) {
        float mx = 0.f;
        for (size_t k = 0; k < K; ++k) mx = std::max(mx, std::fabs(W[n * K + k] * (gamma ? (*gamma)[k] : 1.0f)));
        const int sh = mx > 0.f ? (int)std::floor(std::log2(448.0 / (double)mx)) : 0;
        const float sc = std::ldexp(1.0f, sh);
        double acc = 0.0;
        for (size_t k = 0; k < K; ++k) {
          const uint8_t q = f32_to_e4m3(W[n * K + k] * (gamma ? (*gamma)[k] : 1.0f) * sc);
          d8[n * K + k] = q;
          acc += (double)e4m3_to_f32(q);
        }
        inv[n] = 1.0f / sc;
        if (cs8) cs8[n] = (float)(acc / (double)sc);
      }
    }
  }
  e->finalized = true;
  return DPTX_OK;
}

// ------------------------------------------------------------------------------ schedule
struct Run {
  dptx_engine* e;
  int B;
  hipStream_t st;
  int dt;
  int Hi = 384, Wi = 384;  // input size (multiples of 32)
  int io = 0;              // element type of the caller's x / y buffers (DPTX_IO_*)
  size_t abase = 0;        // byte offset of this run's arena region
  bool half = false;       // half-batch plan (Buf::off2) instead of the whole-batch plan
  hipError_t err = hipSuccess;
  const char* where = "";
  int64_t launches = 0;    // accounting of this run (copied to the engine by the caller)
  double exec_macs = 0.0, cat_macs[4] = {0, 0, 0, 0};
  // fp8: slot (dptx_engine::act_scale index) of the latest e4m3 copy of an arena buffer; slots count up in launch order
  std::unordered_map<const void*, int> q8_slot;
  int q8_count = 0;
  int q8_produce(const void* out) {
    const int slot = q8_count < dptx_engine::MAX_Q8 ? q8_count : dptx_engine::MAX_Q8 - 1;
    ++q8_count;
    q8_slot[out] = slot;
    return slot;
  }
  float q8_scale_of(const void* in) const {
    auto it = q8_slot.find(in);
    return it == q8_slot.end() ? 1.0f : e->act_scale[it->second];
  }
  // calibration forwards (dptx_calibrate_fp8) measure max |x| of every tensor that gets an e4m3 copy
  void q8_measure(const void* out, size_t n, int relu, int slot) {
    if (e->calibrating && e->d_amax) chk(launch_amax(MODE_BF16, out, n, relu, e->d_amax + slot, st), "fp8.amax");
  }

  char* A(const Buf& b) const { return e->d_arena + abase + (half ? b.off2 : b.off); }

  void chk(hipError_t r, const char* w, int cat = 3) {
    if (err == hipSuccess && r != hipSuccess) { err = r; where = w; }
    launches++;
    if (e->profiling) {
      const size_t i = e->event_cat.size() + 1;  // events[0] marks the start of the forward
      if (i >= e->events.size()) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return;
        e->events.push_back(ev);
      }
      (void)hipEventRecord(e->events[i], st);
      e->event_cat.push_back(cat);
      e->event_name.emplace_back(w);
    }
  }
  // MIXED dtype, decoder groups: which arena tensors currently have a valid lo plane (set by their producers)
  std::unordered_map<const void*, bool> lo_valid_;
  bool has_lo(const void* p) const {
    auto it = lo_valid_.find(p);
    return it != lo_valid_.end() && it->second;
  }
  void mark_lo(const void* p, bool v) { lo_valid_[p] = v; }
  int cur_group = 0;
  void tap(const char* name, const void* p, int64_t h, int64_t w, int64_t c, bool fp32 = false) {
    int m = dt;
    if (e->mixed() && lo_valid_.count(p)) m = has_lo(p) ? MODE_FP16X3 : MODE_FP16;  // decoder tensors: by their own planes
    e->taps[name] = TapInfo{p, {B, h, w, c}, fp32, m};
  }
  void group(int g) { dt = e->mode_of(g); cur_group = g; }  // the launches that follow belong to layer group g

  // NHWC convolution as implicit GEMM
  // gn_part != nullptr: the epilogue also writes the GroupNorm(32) statistics of the output (per 32-row block records,
  // kernels.h GemmParams::gn_part); the caller checked gn_fusable(Hout * Wout)
  // out_lo (MIXED dtype, decoder groups only; -1 elsewhere): 1 = some consumer multiplies this tensor with 3 MFMAs and needs
  // its lo plane, 0 = nobody reads it.  The layer's own arithmetic follows dptx_engine::layer_x3.
  void conv(const void* in, int Hin, int Win, int Cin, const std::string& wkey, int ksz, int stride, int pad_t, int pad_l,
            int Hout, int Wout, int Cout, void* out, const float* bias, int act, int a_relu, const void* R1 = nullptr,
            const void* R2 = nullptr, float* gn_part = nullptr, int q = 0, int out_lo = -1) {
    // fp8 dtype: q = 1 / 2 also writes the e4m3 copy of the output (2: ReLU'd, for consumers that pre-activate); a conv
    // whose weight has an e4m3 copy runs on the fp8 MFMA, reading the e4m3 copy of `in` (ReLU'd by its producer)
    // (a calibration forward runs these convs on their bf16 operands: a saturated e4m3 copy upstream must not distort the
    // max |x| measured downstream)
    const bool f8 = e->fp8_use(wkey) && !e->calibrating;
    GemmParams p{};
    p.A = in; p.W = e->w(wkey); p.C = out; p.bias = bias; p.R1 = R1; p.R2 = R2;
    int slot = -1;
    if (e->fp8() && q) {
      slot = q8_produce(out);
      p.C8 = e->q8(out); p.q_relu = q == 2; p.q_scale = e->act_scale[slot];
    }
    p.M = B * Hout * Wout; p.N = Cout; p.K = ksz * ksz * Cin; p.ldw = p.K;
    p.a_rpi = Hout * Wout; p.Wout = Wout; p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.a_pix_stride = Cin;
    p.a_img_stride = (long long)Hin * Win * Cin; p.a_off = 0;
    p.a_bytes = (long long)B * Hin * Win * Cin * 2;
    p.ksz = ksz; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
    p.c_rpi = 0x7fffffff; p.c_img_rows = 0; p.c_row_off = 0; p.ldc = Cout;
    p.act = act; p.a_relu = a_relu; p.planes = e->pl;
    p.k_tap_fast = (ksz == 3 && Cin >= 512) ? 1 : 0;  // measured per layer: profiles/r01_experiments.md
    if (gn_part) { p.gn_part = gn_part; p.gn_hw = Hout * Wout; p.gn_blocks = Hout * Wout / 32; p.gn_cpg = Cout / 32; }
    if (f8) {
      p.A = e->q8(in); p.W = e->w8(wkey); p.a_bytes /= 2; p.a_relu = 0;
      p.out_scale = 1.0f / q8_scale_of(in); p.out_scale_v = e->wscale(wkey);
    }
    exec_macs += (double)p.M / B * p.N * p.K;
    cat_macs[0] += (double)p.M / B * p.N * p.K;
    int mode = f8 ? MODE_FP8 : dt;
    if (out_lo >= 0 && e->mixed()) {  // per-layer policy
      const bool x3 = e->layer_x3(wkey, cur_group), a_lo = e->layer_reads_lo(wkey, cur_group);
      p.a_hi_only = x3 && !a_lo;
      if (a_lo && !has_lo(in)) {  // cannot happen with the schedule below: every producer honours its consumers' needs
        if (err == hipSuccess) { err = hipErrorInvalidValue; where = "precision policy: a 3-MFMA layer reads a tensor without lo plane"; }
        return;
      }
      const bool r1lo = R1 != nullptr && has_lo(R1), r2lo = R2 != nullptr && has_lo(R2);
      mode = x3 ? MODE_FP16X3 : MODE_FP16;
      p.r1_hi_only = !r1lo; p.r2_hi_only = !r2lo; p.c_hi_only = !out_lo;
      p.epi2 = (!x3 && (out_lo || r1lo || r2lo)) ? 1 : 0;
      mark_lo(out, out_lo != 0);
    }
    chk(launch_gemm(mode, p, st), wkey.c_str(), 0);
    if (slot >= 0) q8_measure(out, (size_t)p.M * p.N, q == 2, slot);
  }

  // GroupNorm statistics come out of the producing conv's epilogue when an image's rows are whole 32-row MFMA blocks
  // (always at 384x384; DPTX_GN_FUSED=0 keeps the separate statistics launch for A/B runs)
  static bool gn_fusable(int HW) {
    static int env = -1;
    if (env < 0) { const char* t = getenv("DPTX_GN_FUSED"); env = (t && t[0] == '0') ? 0 : 1; }
    return env == 1 && HW % 32 == 0;
  }
  void gn_stats(const void* X, float* part, int HW, int C) { chk(launch_gn_stats(dt, X, part, B, HW, C, e->pl, st), "gn_stats", 2); }
  void gn_apply(void* X, const std::string& nkey, float* part, int HW, int C, int relu, const void* R = nullptr,
                const std::string& rkey = "", const float* rpart = nullptr) {
    GnParams g{};
    g.nrec = gn_fusable(HW) ? HW / 32 : 0;
    g.X = X; g.Y = X; g.gamma = e->f(nkey + ".weight"); g.beta = e->f(nkey + ".bias"); g.partial = part;
    g.R = R;
    if (!rkey.empty()) { g.r_gamma = e->f(rkey + ".weight"); g.r_beta = e->f(rkey + ".bias"); g.r_partial = rpart; }
    g.B = B; g.HW = HW; g.C = C; g.relu = relu; g.eps = 1e-5f;
    chk(launch_gn_apply(dt, g, e->pl, st), nkey.c_str(), 2);
  }

  // RCU (blocks.py:263-286): out = conv2(relu(conv1(relu(x)))) + x (+ extra)
  // q_out (fp8 dtype): e4m3 copy of the unit's output -- 2 when its consumer pre-activates (another RCU), 1 otherwise
  // out_lo: MIXED per-layer policy (conv()): does a consumer of the unit's output need its lo plane
  void rcu(const std::string& p, const void* x, int H, int W, void* tmp, void* out, const void* extra, int q_out, int out_lo) {
    const int mid_lo = e->mixed() ? (int)e->layer_reads_lo(p + "conv2.weight", cur_group) : -1;
    conv(x, H, W, FEAT, p + "conv1.weight", 3, 1, 1, 1, H, W, FEAT, tmp, e->f(p + "conv1.bias"), /*act*/ 1, /*a_relu*/ 1, nullptr,
         nullptr, nullptr, e->fp8_use(p + "conv2.weight") ? 1 : 0, mid_lo);
    conv(tmp, H, W, FEAT, p + "conv2.weight", 3, 1, 1, 1, H, W, FEAT, out, e->f(p + "conv2.bias"), 0, 0, x, extra, nullptr, q_out,
         e->mixed() ? out_lo : -1);
  }

  int forward(const void* x, void* y, void* y2);
};

int Run::forward(const void* x, void* y, void* y2) {
  dptx_engine* E = e;
  E->taps.clear();
  E->event_cat.clear();
  E->event_name.clear();
  if (E->profiling) {
    if (E->events.empty()) {
      hipEvent_t ev;
      if (hipEventCreate(&ev) == hipSuccess) E->events.push_back(ev);
    }
    if (!E->events.empty()) (void)hipEventRecord(E->events[0], st);
  }
  const std::string vp = "pretrained.model.";
  const std::string bp = vp + "patch_embed.backbone.";
  const int D_VIT = E->dv, D_MLP = E->dm, N_HEADS = E->nh;
  const bool large = E->backbone == DPTX_BACKBONE_VITL16_384;
  float* part0 = (float*)A(E->part[0]);
  float* part1 = (float*)A(E->part[1]);
  float* part2 = (float*)A(E->part[2]);
  float* part3 = (float*)A(E->part[3]);
  const int h2 = Hi / 2, w2 = Wi / 2, h4 = Hi / 4, w4 = Wi / 4, gh = Hi / 16, gw = Wi / 16, h32 = Hi / 32, w32 = Wi / 32;
  const int NP = gh * gw;   // patch tokens per image (576 at 384x384)
  const int S = NP + 1;     // + cls
  const bool native = (gh == 24 && gw == 24);
  const float* pos = E->f("pretrained.model.pos_embed");
  if (!native) {  // vit.py:119-125: forward_flex resizes pos_embed to the input's patch grid on every call
    chk(launch_pos_resize(pos, (float*)A(E->pos_alt), 24, gh, gw, D_VIT, st), "pos_resize");
    pos = (const float*)A(E->pos_alt);
  }

  if (!large) {
  // ---- stem: fused conv7x7 s2 SAME (stem.hip, no im2col) -> GN+ReLU -> MaxPool2dSame(3,2) ---------
  group(DPTX_GROUP_RESNET);
  chk(launch_stem_conv(dt, x, io, E->w(bp + "stem.conv.weight"), A(E->sraw), B, Hi, Wi, E->pl, st), "stem.conv", 0);
  exec_macs += (double)h2 * w2 * 64 * STEM_K;
  cat_macs[0] += (double)h2 * w2 * 64 * STEM_K;
  gn_stats(A(E->sraw), part0, h2 * w2, 64);
  chk(launch_gn_relu_maxpool(dt, A(E->sraw), A(E->stem), E->f(bp + "stem.norm.weight"), E->f(bp + "stem.norm.bias"),
                             part0, B, h2, w2, 64, 1e-5f, E->pl, st),
      "stem.pool", 2);
  tap("stem", A(E->stem), h4, w4, 64);

  // ---- ResNetV2 stages (3,4,9) non-preact bottlenecks -----------------------------------
  const void* cur = A(E->stem);
  int H = h4, Wd = w4, cin = 64;
  for (int s = 0; s < 3; ++s) {
    const int cout = STAGE_OUT[s], mid = cout / 4;
    for (int b = 0; b < STAGE_DEPTH[s]; ++b) {
      const std::string p = bp + "stages." + std::to_string(s) + ".blocks." + std::to_string(b) + ".";
      const int stride = (b == 0) ? STAGE_STRIDE[s] : 1;
      const int Ho = H / stride, Wo = Wd / stride;
      void* out = (b == STAGE_DEPTH[s] - 1) ? (void*)A(E->S[s]) : (void*)((b & 1) ? A(E->PB) : A(E->PA));
      const bool f_in = gn_fusable(H * Wd), f_out = gn_fusable(Ho * Wo);  // statistics from the conv epilogues
      if (b == 0) {
        conv(cur, H, Wd, cin, p + "downsample.conv.weight", 1, stride, 0, 0, Ho, Wo, cout, A(E->DS), nullptr, 0, 0, nullptr, nullptr,
             f_out ? part3 : nullptr);
        if (!f_out) gn_stats(A(E->DS), part3, Ho * Wo, cout);
      }
      conv(cur, H, Wd, cin, p + "conv1.weight", 1, 1, 0, 0, H, Wd, mid, A(E->T1), nullptr, 0, 0, nullptr, nullptr, f_in ? part0 : nullptr);
      if (!f_in) gn_stats(A(E->T1), part0, H * Wd, mid);
      gn_apply(A(E->T1), p + "norm1", part0, H * Wd, mid, 1);
      // 3x3, stride on conv2 (V1.5); TF-SAME: s1 -> pad (1,1); s2 on even H -> pad (0,1)
      const int pad = (stride == 1) ? 1 : 0;
      conv(A(E->T1), H, Wd, mid, p + "conv2.weight", 3, stride, pad, pad, Ho, Wo, mid, A(E->T2), nullptr, 0, 0, nullptr, nullptr,
           f_out ? part1 : nullptr);
      if (!f_out) gn_stats(A(E->T2), part1, Ho * Wo, mid);
      gn_apply(A(E->T2), p + "norm2", part1, Ho * Wo, mid, 1);
      conv(A(E->T2), Ho, Wo, mid, p + "conv3.weight", 1, 1, 0, 0, Ho, Wo, cout, out, nullptr, 0, 0, nullptr, nullptr,
           f_out ? part2 : nullptr);
      if (!f_out) gn_stats(out, part2, Ho * Wo, cout);
      if (b == 0)
        gn_apply(out, p + "norm3", part2, Ho * Wo, cout, 1, A(E->DS), p + "downsample.norm", part3);
      else
        gn_apply(out, p + "norm3", part2, Ho * Wo, cout, 1, cur);
      cur = out;
      H = Ho;
      Wd = Wo;
      cin = cout;
    }
    const char* names[3] = {"s0", "s1", "s2"};
    tap(names[s], cur, H, Wd, cout);
  }
  }  // !large

  // ---- tokens: 1x1 proj + bias + pos_embed -> fp32 stream X[b*S + 1 + p] (S = 577 at 384x384); cls rows ----
  float* X = (float*)A(E->X);
  // LayerNorm fold (kernels.h GemmParams::ln_stats): every launch that writes the fp32 token stream also writes its 16-bit
  // copy into Hn and the per-row (sum, sum of squares) records into lnst; qkv / fc1 read Hn and normalise in the epilogue
  float* lnst = (float*)A(E->lnst);
  const int ln_nblk = D_VIT / 128;  // records per token row (6 or 8; the row stride is always 8)
  const bool s16 = E->stream16;     // Hn is the (16-bit) token stream; X is not written
  group(DPTX_GROUP_EMBED);
  int hn_slot = -1;   // fp8 ViT: calibration slot of the token stream's latest e4m3 copy
  {
    // hybrid: the 1x1 projection of the ResNet's 1/16-resolution map (K = 1024); DPT-Large: timm PatchEmbed, a 16x16
    // stride-16 convolution = a dense GEMM over the patch matrix (K = 3*16*16 = 768, misc.hip patchify16)
    const int Kp = large ? 768 : 1024;
    // (the patch matrix lives in F1, which is free here: Hn receives the 16-bit copy of the token stream below)
    if (large) chk(launch_patchify16(dt, x, io, A(E->F1), B, Hi, Wi, E->pl, st), "patchify");
    GemmParams p;
    gemm_params_dense(p, B * NP, D_VIT, Kp);
    p.A = large ? A(E->F1) : A(E->S[2]); p.W = E->w(vp + "patch_embed.proj.weight"); p.C = X;
    p.bias = E->f(vp + "patch_embed.proj.bias");
    p.c_rpi = NP; p.c_img_rows = S; p.c_row_off = 1; p.ldc = D_VIT; p.c_fp32 = 1;
    p.R2 = pos; p.r2_bcast = 1; p.r2_fp32 = 1; p.planes = E->pl;
    if (E->ln_fold) { p.C16 = A(E->Hn); p.row_stats = lnst; p.stats_nblk = 8; }
    if (s16) { p.C = A(E->Hn); p.c_fp32 = 0; p.C16 = nullptr; }  // the 16-bit tensor is the stream itself
    // fp8 ViT (DPTX_FLAG_FP8_VIT): block 0's qkv multiplies the e4m3 copy of the stream
    hn_slot = E->fp8_use(vp + "blocks.0.attn.qkv.weight") ? q8_produce(A(E->Hn)) : -1;
    if (hn_slot >= 0) { p.C8 = E->q8(A(E->Hn)); p.q_relu = 0; p.q_scale = E->act_scale[hn_slot]; }
    exec_macs += (double)NP * D_VIT * Kp;
    cat_macs[0] += (double)NP * D_VIT * Kp;
    chk(launch_gemm(dt, p, st), "patch_embed.proj", 0);
  }
  chk(launch_cls_rows(E->mode_of(DPTX_GROUP_VIT), E->f(vp + "cls_token"), pos, s16 ? nullptr : X, B, S, D_VIT,
                      E->ln_fold ? A(E->Hn) : nullptr, E->ln_fold ? lnst : nullptr, st,
                      hn_slot >= 0 ? E->q8(A(E->Hn)) : nullptr, hn_slot >= 0 ? E->act_scale[hn_slot] : 1.0f),
      "cls_rows");
  if (hn_slot >= 0) q8_measure(A(E->Hn), (size_t)B * S * D_VIT, 0, hn_slot);
  const int M = B * S;
  const size_t tok_elems = (size_t)M * D_VIT;
  auto tok_tap = [&](int idx, const char* name) {
    if (!E->taps_on) return;
    float* dst = E->d_tok_taps + (size_t)idx * E->tok_tap_stride;
    if (s16) {  // debug copy: not a launch of the schedule (not counted, not timed)
      const hipError_t r = launch_to_f32(E->mode_of(DPTX_GROUP_VIT), A(E->Hn), dst, tok_elems, E->pl, st);
      if (err == hipSuccess && r != hipSuccess) { err = r; where = "tok_tap"; }
    } else if (err == hipSuccess) {
      err = hipMemcpyAsync(dst, X, tok_elems * 4, hipMemcpyDeviceToDevice, st);
    }
    E->taps[name] = TapInfo{dst, {B, S, D_VIT, 1}, true, dt};
  };
  tok_tap(0, "tok0");
  // debug: word sums of {lnst, Hn, QKV, AO, F1} after a launch (single-stream forwards only; tools/gpu/r4_hunt3.py)
  auto vit_sums = [&]() {
    if (!E->d_lsums || half) return;
    const Buf* bufs[5] = {&E->lnst, &E->Hn, &E->QKV, &E->AO, &E->F1};
    const size_t bytes[5] = {(size_t)M * 8 * 8, tok_elems * 2, (size_t)M * 3 * D_VIT * 2, tok_elems * 2, (size_t)M * D_MLP * 2};
    if (E->lsum_n + 5 > E->lsum_cap) return;
    for (int i = 0; i < 5; ++i) {
      const hipError_t r = launch_checksum(A(*bufs[i]), bytes[i], E->d_lsums + E->lsum_n + i, st);
      if (err == hipSuccess && r != hipSuccess) { err = r; where = "vit_sums"; }
    }
    E->lsum_n += 5;
  };
  if (E->d_lsums) { E->lsum_n = 0; (void)hipMemsetAsync(E->d_lsums, 0, (size_t)E->lsum_cap * 8, st); }
  vit_sums();

  // ln: 0 plain; 1 consumer of the folded LayerNorm (qkv, fc1); 2 producer (proj, fc2: 16-bit copy + row statistics)
  // q8_out (fp8 ViT): the consumer of C multiplies on e4m3 operands -- the epilogue also writes C's e4m3 copy (one calibrated
  // power-of-two scale per tensor, like the decoder's: Run::conv)
  auto dense = [&](const void* A, int a_fp32, const std::string& wkey, int N, int K, void* C, int c_fp32, const float* bias,
                   int act, const void* R1, int r1_fp32, int ln = 0, const float* ln_colsum = nullptr, bool q8_out = false) {
    GemmParams p;
    gemm_params_dense(p, M, N, K);
    p.A = A; p.a_fp32 = a_fp32; p.W = E->w(wkey); p.C = C; p.c_fp32 = c_fp32; p.bias = bias; p.act = act;
    p.R1 = R1; p.r1_fp32 = r1_fp32; p.planes = E->pl;
    if (ln == 1) { p.ln_stats = lnst; p.ln_colsum = ln_colsum; p.ln_nblk = ln_nblk; p.ln_eps = 1e-6f; p.ln_inv_dim = 1.0f / (float)K; }
    if (ln == 2) { p.C16 = this->A(E->Hn); p.row_stats = lnst; p.stats_nblk = 8; }
    if (ln == 2 && s16) {  // in place on the 16-bit stream: every thread reads exactly the elements it then writes
      p.C = this->A(E->Hn); p.c_fp32 = 0; p.R1 = this->A(E->Hn); p.r1_fp32 = 0; p.C16 = nullptr;
    }
    int mode = dt;
    if (E->fp8_use(wkey) && !E->calibrating) {  // e4m3 operands: the e4m3 copies of A and of the (folded) weight
      p.A = E->q8(A); p.W = E->w8(wkey); p.a_bytes /= 2;
      p.out_scale = 1.0f / q8_scale_of(A); p.out_scale_v = E->wscale(wkey);
      if (ln == 1) p.ln_colsum = E->f(wkey.substr(0, wkey.size() - 6) + "lnsum8");
      mode = MODE_FP8;
    }
    int slot = -1;
    if (q8_out) {
      slot = q8_produce(p.C);
      p.C8 = E->q8(p.C); p.q_relu = 0; p.q_scale = E->act_scale[slot];
    }
    exec_macs += (double)S * N * K;
    cat_macs[0] += (double)S * N * K;
    chk(launch_gemm(mode, p, st), wkey.c_str(), 0);
    if (slot >= 0) q8_measure(p.C, (size_t)M * N, 0, slot);
  };

  // ProjectReadout + reassemble for hook n (hybrid: 3 -> block 8, 4 -> block 11; DPT-Large: 1..4 -> blocks 5, 11, 17, 23,
  // n = 1, 2 followed by a ConvTranspose2d with kernel == stride: a GEMM whose N enumerates (dy, dx, co) + a re-layout)
  auto readout = [&](int n) {
    group(DPTX_GROUP_REASSEMBLE);
    const std::string pp = "pretrained.act_postprocess" + std::to_string(n) + ".";
    float* clsb = (float*)A(E->clsb);
    chk(launch_readout_cls(dt, s16 ? (const float*)A(E->Hn) : X, (long long)S * D_VIT, E->w(pp + "0.project.0.weight"), 2 * D_VIT,
                           D_VIT, E->f(pp + "0.project.0.bias"), clsb, B, D_VIT, D_VIT, E->pl, st, s16 ? 1 : 0),
        "readout_cls");
    // the token GEMM reads a 16-bit image of the fp32 stream: with the LayerNorm fold the last fc2 epilogue has already
    // written it (single-plane: the fold implies single-pass ViT blocks; a 3-MFMA reassemble group then still needs the
    // lo plane, so it casts); otherwise Hn is free between blocks
    if (!(E->ln_fold && !mode_is_x3(dt))) chk(launch_cast_f32(dt, X, A(E->Hn), (size_t)B * S * D_VIT, E->pl, st), "readout_cast");
    exec_macs += (double)D_VIT * D_VIT;  // per image
    void* R = (n & 1) ? A(E->R3) : A(E->R4);
    GemmParams p{};
    p.A = A(E->Hn); p.a_bytes = (long long)B * S * D_VIT * 2; p.planes = E->pl;
    p.W = E->w(pp + "0.project.0.weight"); p.ldw = 2 * D_VIT; p.C = R;
    p.M = B * NP; p.N = D_VIT; p.K = D_VIT;
    p.a_rpi = NP; p.Wout = NP; p.Hin = 1; p.Win = NP; p.Cin = D_VIT; p.a_pix_stride = D_VIT;
    p.a_img_stride = (long long)S * D_VIT; p.a_off = D_VIT;  // skip the cls row
    p.ksz = 1; p.stride = 1;
    p.c_rpi = NP; p.c_img_rows = NP; p.c_row_off = 0; p.ldc = D_VIT;
    p.bias = clsb; p.bias_per_img = 1; p.act = 2;
    exec_macs += (double)NP * D_VIT * D_VIT;
    cat_macs[0] += (double)NP * D_VIT * D_VIT;
    chk(launch_gemm(dt, p, st), "readout", 0);
    if (n <= 2) {
      const int f = n == 1 ? 256 : 512, k = n == 1 ? 4 : 2;
      conv(R, gh, gw, D_VIT, pp + "3.weight", 1, 1, 0, 0, gh, gw, f, A(E->T4), E->f(pp + "3.bias"), 0, 0);
      GemmParams d;
      gemm_params_dense(d, B * NP, k * k * f, f);
      d.A = A(E->T4); d.W = E->w(pp + "4.weight"); d.C = A(E->F1); d.bias = E->f(pp + "4.bias"); d.planes = E->pl;
      exec_macs += (double)NP * k * k * f * f;
      cat_macs[0] += (double)NP * k * k * f * f;
      chk(launch_gemm(dt, d, st), "reassemble.deconv", 0);
      chk(launch_depth_to_space(dt, A(E->F1), A(E->S[n - 1]), B, gh, gw, k, f, E->pl, st), "reassemble.d2s");
      tap(n == 1 ? "l1" : "l2", A(E->S[n - 1]), gh * k, gw * k, f);
    } else if (n == 3) {
      conv(R, gh, gw, D_VIT, pp + "3.weight", 1, 1, 0, 0, gh, gw, D_VIT, A(E->L3), E->f(pp + "3.bias"), 0, 0);
      tap("l3", A(E->L3), gh, gw, D_VIT);
    } else {
      conv(R, gh, gw, D_VIT, pp + "3.weight", 1, 1, 0, 0, gh, gw, D_VIT, A(E->T4), E->f(pp + "3.bias"), 0, 0);
      conv(A(E->T4), gh, gw, D_VIT, pp + "4.weight", 3, 2, 1, 1, h32, w32, D_VIT, A(E->L4), E->f(pp + "4.bias"), 0, 0);
      tap("l4", A(E->L4), h32, w32, D_VIT);
    }
  };

  // ---- 12 (24) transformer blocks (timm Block; LN eps 1e-6) -------------------------------
  for (int l = 0; l < E->depth; ++l) {
    group(DPTX_GROUP_VIT);
    const std::string p = vp + "blocks." + std::to_string(l) + ".";
    const bool lf = E->ln_fold;
    if (!lf) chk(launch_layernorm(dt, X, E->f(p + "norm1.weight"), E->f(p + "norm1.bias"), A(E->Hn), M, D_VIT, 1e-6f, E->pl, st), "ln1", 2);
    // fp8 ViT: the tensors that an e4m3 GEMM reads get their e4m3 copy from the epilogue that produces them -- the stream
    // after proj (fc1's operand), the GELU output (fc2's), the stream after fc2 (the next block's qkv)
    const std::string pn = vp + "blocks." + std::to_string(l + 1) + ".";
    const bool q_fc1 = E->fp8_use(p + "mlp.fc1.weight"), q_fc2 = E->fp8_use(p + "mlp.fc2.weight");
    const bool q_next = l + 1 < E->depth && E->fp8_use(pn + "attn.qkv.weight");
    dense(A(E->Hn), 0, p + "attn.qkv.weight", 3 * D_VIT, D_VIT, A(E->QKV), 0, E->f(p + "attn.qkv.bias"), 0, nullptr, 0, lf ? 1 : 0,
          lf ? E->f(p + "attn.qkv.lnsum") : nullptr);
    vit_sums();
    chk(launch_attention(dt, A(E->QKV), A(E->AO), B, S, N_HEADS, E->pl, st), "attention", 1);
    vit_sums();
    exec_macs += 2.0 * N_HEADS * (double)S * S * 64;
    cat_macs[1] += 2.0 * N_HEADS * (double)S * S * 64;
    dense(A(E->AO), 0, p + "attn.proj.weight", D_VIT, D_VIT, X, 1, E->f(p + "attn.proj.bias"), 0, X, 1, lf ? 2 : 0, nullptr, q_fc1);
    vit_sums();
    if (!lf) chk(launch_layernorm(dt, X, E->f(p + "norm2.weight"), E->f(p + "norm2.bias"), A(E->Hn), M, D_VIT, 1e-6f, E->pl, st), "ln2", 2);
    dense(A(E->Hn), 0, p + "mlp.fc1.weight", D_MLP, D_VIT, A(E->F1), 0, E->f(p + "mlp.fc1.bias"), 2, nullptr, 0, lf ? 1 : 0,
          lf ? E->f(p + "mlp.fc1.lnsum") : nullptr, q_fc2);
    vit_sums();
    dense(A(E->F1), 0, p + "mlp.fc2.weight", D_VIT, D_MLP, X, 1, E->f(p + "mlp.fc2.bias"), 0, X, 1, lf ? 2 : 0, nullptr, q_next);
    vit_sums();
    {
      char nm[16];
      snprintf(nm, sizeof nm, "blk%d", l);
      tok_tap(l + 1, nm);
    }
    if (large) {
      if (l % 6 == 5) readout(l / 6 + 1);  // dpt_depth.py:41-45 hooks [5, 11, 17, 23]
    } else {
      if (l == 8) readout(3);  // hooks [0, 1, 8, 11]: the first two are the ResNet stages
      if (l == 11) readout(4);
    }
  }
  // timm's final model.norm is dead compute in the reference (vit.py:153, result discarded :64)

  // one decoder = scratch.* of one task ("" -> scratch.*, "depth." -> depth.scratch.*); the dual-task engine runs two
  // on the same encoder outputs (S[0], S[1], L3, L4 are not modified by a decoder)
  // MIXED dtype: the decoder groups follow the per-LAYER precision table (dptx_engine::layer_prec); a producer writes the lo
  // plane of its output exactly when a consumer multiplies it with 3 MFMAs (or, for the path tensors, when it was computed
  // with 3 MFMAs itself), whatever its own arithmetic is.  The encoder's outputs carry lo planes when their groups are 3-MFMA.
  const bool mx = E->mixed();
  if (mx) {
    const bool enc_lo = mode_is_x3(E->mode_of(large ? DPTX_GROUP_REASSEMBLE : DPTX_GROUP_RESNET));
    const bool re_lo = mode_is_x3(E->mode_of(DPTX_GROUP_REASSEMBLE));
    mark_lo(A(E->S[0]), enc_lo); mark_lo(A(E->S[1]), enc_lo); mark_lo(A(E->L3), re_lo); mark_lo(A(E->L4), re_lo);
  }
  auto X3 = [&](const std::string& key, int grp) { return mx && E->layer_x3(key, grp); };            // computes with lo planes
  auto RLO = [&](const std::string& key, int grp) { return mx && E->layer_reads_lo(key, grp); };   // reads its input's lo plane
  auto lo_of = [&](bool need) { return mx ? (int)need : -1; };
  auto planes_mode = [&](const void* t) { return mx ? (has_lo(t) ? MODE_FP16X3 : MODE_FP16) : dt; };  // elementwise kernels

  auto decode = [&](const std::string& pre, int ch, void* yout) {
  // ---- scratch.layerN_rn (3x3, no bias) ---------------------------------------------------
  const void* rn_in[4] = {A(E->S[0]), A(E->S[1]), A(E->L3), A(E->L4)};
  const int rn_h[4] = {h4, Hi / 8, gh, h32};
  const int rn_w[4] = {w4, Wi / 8, gw, w32};
  const int rn_c[4] = {256, 512, D_VIT, D_VIT};
  const char* rn_names[4] = {"l1_rn", "l2_rn", "l3_rn", "l4_rn"};
  group(DPTX_GROUP_RN);
  for (int i = 0; i < 4; ++i) {
    // consumer of lrn[i] as a GEMM operand: refinenet4.resConfUnit2.conv1 (i = 3), refinenet{i+1}.resConfUnit1.conv1 (else)
    const std::string cons = pre + "scratch.refinenet" + std::to_string(i + 1) + (i == 3 ? ".resConfUnit2.conv1.weight" : ".resConfUnit1.conv1.weight");
    conv(rn_in[i], rn_h[i], rn_w[i], rn_c[i], pre + "scratch.layer" + std::to_string(i + 1) + "_rn.weight", 3, 1, 1, 1, rn_h[i],
         rn_w[i], FEAT, A(E->lrn[i]), nullptr, 0, 0, nullptr, nullptr, nullptr, /*q: every consumer pre-activates*/ E->fp8_use(cons) ? 2 : 0,
         lo_of(RLO(cons, DPTX_GROUP_FUSION)));
    tap((pre + rn_names[i]).c_str(), A(E->lrn[i]), rn_h[i], rn_w[i], FEAT);
  }

  // ---- RefineNet fusion 4 -> 1 (blocks.py:320-341).  out_conv (1x1) is applied BEFORE the x2
  // bilinear upsample: both are linear and the interpolation weights sum to 1, so
  // out_conv(up(x)) == up(out_conv(x)) exactly in real arithmetic, at a quarter of the MACs.
  group(DPTX_GROUP_FUSION);
  const void* path = nullptr;
  const char* p_names[4] = {"p1", "p2", "p3", "p4"};
  const std::string oc = pre + "scratch.output_conv.";
  for (int i = 4; i >= 1; --i) {
    const std::string p = pre + "scratch.refinenet" + std::to_string(i) + ".";
    const int h = rn_h[i - 1], w = rn_w[i - 1];
    const void* sum;
    if (i == 4) {
      sum = A(E->lrn[3]);
    } else {
      rcu(p + "resConfUnit1.", A(E->lrn[i - 1]), h, w, A(E->tA), A(E->tB), path, E->fp8_use(p + "resConfUnit2.conv1.weight") ? 2 : 0,
          lo_of(RLO(p + "resConfUnit2.conv1.weight", DPTX_GROUP_FUSION)));  // tB = path + RCU1(lrn)
      sum = A(E->tB);
    }
    rcu(p + "resConfUnit2.", sum, h, w, A(E->tA), A(E->tC), nullptr, E->fp8_use(p + "out_conv.weight") ? 1 : 0,
        lo_of(RLO(p + "out_conv.weight", DPTX_GROUP_FUSION)));
    // the path tensor keeps a lo plane when out_conv computed one (it is added to the next stage's sum in fp32) and when
    // the first head conv multiplies it with 3 MFMAs.  A 2-MFMA head conv reads the hi plane only: path_1 is then interpolated
    // from both planes of the low-resolution tensor and rounded ONCE, into the hi plane
    conv(A(E->tC), h, w, FEAT, p + "out_conv.weight", 1, 1, 0, 0, h, w, FEAT, A(E->tA), E->f(p + "out_conv.bias"), 0, 0, nullptr,
         nullptr, nullptr, 0, lo_of(X3(p + "out_conv.weight", DPTX_GROUP_FUSION) || (i == 1 && X3(oc + "0.weight", DPTX_GROUP_HEAD))));
    // path_1 feeds the first head conv: in the fp8 dtype the up-sampling also writes its e4m3 copy
    {
      const bool up8 = i == 1 && E->fp8_use(oc + "0.weight");
      const int slot = up8 ? q8_produce(A(E->P[0])) : -1;
      const bool hi_only = mx && i == 1 && has_lo(A(E->tA)) && !RLO(oc + "0.weight", DPTX_GROUP_HEAD);
      chk(launch_upsample2x(planes_mode(A(E->tA)), A(E->tA), A(E->P[i - 1]), B, h, w, FEAT, E->pl, st,
                            up8 ? E->q8(A(E->P[0])) : nullptr, up8 ? E->act_scale[slot] : 1.0f, hi_only),
          "fusion.up");
      if (up8) q8_measure(A(E->P[0]), (size_t)B * 4 * h * w * FEAT, 0, slot);
      if (mx) mark_lo(A(E->P[i - 1]), has_lo(A(E->tA)) && !hi_only);
    }
    path = A(E->P[i - 1]);
    tap((pre + p_names[i - 1]).c_str(), path, 2 * h, 2 * w, FEAT);
  }

  // ---- head (dpt_depth.py:91-99) ----------------------------------------------------------
  group(DPTX_GROUP_HEAD);
  const bool conv2_x3 = mx ? X3(oc + "2.weight", DPTX_GROUP_HEAD) : mode_is_x3(dt);
  conv(path, h2, w2, FEAT, oc + "0.weight", 3, 1, 1, 1, h2, w2, 128, A(E->H0), E->f(oc + "0.bias"), 0, 0, nullptr, nullptr, nullptr, 0,
       lo_of(conv2_x3));
  tap((pre + "h0").c_str(), A(E->H0), h2, w2, 128);
  // fp16-plane dtypes: did anything upstream leave the fp16 range?  (one pass over the hi plane of H0: every decoder path and,
  // through them, the ViT blocks reach it by residual adds; the ResNetV2 stages cannot overflow -- standardised weights and
  // a GroupNorm behind every convolution bound them whatever the checkpoint)
  if (E->range_check() && E->d_range)
    chk(launch_nonfinite_scan(mx ? MODE_FP16 : dt, A(E->H0), (size_t)B * h2 * w2 * 128, E->d_range, st), "range.scan");
  // (a 2-MFMA second head conv -- dptx_set_layer_precision(..., 2) -- has no fused form: head_tail_x3_kernel always spends three
  //  MFMAs and reads H0's lo plane, so the fused and the tapped forward would differ; it takes the three-launch tail.  ADVICE r4)
  const bool conv2_two = mx && E->layer_mfmas(oc + "2.weight", DPTX_GROUP_HEAD) == 2;
  if (E->head_fused(conv2_x3) && !conv2_two) {
    // x2 upsample + conv 128->32 + ReLU + conv 1x1 + ReLU in one launch (head.hip): the 37.7 MB/image up-sampled map
    // and the 32-channel map never reach memory.  Not with a 3-MFMA conv2, and not while stage taps are recorded ("h1").
    chk(launch_head_tail(conv2_x3 ? MODE_FP16X3 : (mx ? MODE_FP16 : dt), A(E->H0), E->w(oc + "2.weight"), E->f(oc + "2.bias"),
                         E->f(oc + "4.weight"), E->f(oc + "4.bias"), yout, io, B, h2, w2, ch, E->cfg.non_negative, st, E->pl),
        "head.tail", 0);
    exec_macs += (double)Hi * Wi * 32 * 1152;
    cat_macs[0] += (double)Hi * Wi * 32 * (1152 + ch);
  } else {
    chk(launch_upsample2x(planes_mode(A(E->H0)), A(E->H0), A(E->H0U), B, h2, w2, 128, E->pl, st), "head.up");
    if (mx) mark_lo(A(E->H0U), has_lo(A(E->H0)));
    conv(A(E->H0U), Hi, Wi, 128, oc + "2.weight", 3, 1, 1, 1, Hi, Wi, 32, A(E->H1), E->f(oc + "2.bias"), 1, 0, nullptr, nullptr,
         nullptr, 0, lo_of(conv2_x3));
    tap((pre + "h1").c_str(), A(E->H1), Hi, Wi, 32);
    chk(launch_head_out(planes_mode(A(E->H1)), A(E->H1), E->f(oc + "4.weight"), E->f(oc + "4.bias"), yout, io, B, Hi * Wi, ch,
                        E->cfg.non_negative, E->pl, st),
        "head.out");
  }
  exec_macs += (double)Hi * Wi * 32 * ch;
  };
  decode("", E->cfg.num_channels, y);
  if (E->cfg.dual_task) {
    // the second decoder re-uses the first one's arena buffers: the first one's stage taps are gone (debug a decoder
    // with a single-task handle -- its results are bit-identical)
    for (const char* n : {"l1_rn", "l2_rn", "l3_rn", "l4_rn", "p1", "p2", "p3", "p4", "h0", "h1"}) E->taps.erase(n);
    decode("depth.", 1, y2);
  }
  E->n_q8 = q8_count < dptx_engine::MAX_Q8 ? q8_count : dptx_engine::MAX_Q8;
  if (err != hipSuccess) return E->fail(DPTX_E_HIP, std::string("launch failed at ") + where + ": " + hipGetErrorString(err));
  return DPTX_OK;
}

}  // namespace

// =================================================================================== C ABI
extern "C" {

#ifndef DPTX_SRC_HASH
#define DPTX_SRC_HASH "unknown"
#endif
#define DPTX_BUILD_KIND ""
// "src=<hash>": sha256 of the sources this binary was built from (omnidata_amd/build.py source_hash); the Python loader
// compares it with the sources next to the .so and refuses a stale binary
const char* dptx_version(void) { return "dptx 0.3.0 (gfx950) src=" DPTX_SRC_HASH; }

void dptx_default_config(dptx_config* cfg) {
  memset(cfg, 0, sizeof *cfg);
  cfg->num_channels = 3;
  cfg->max_batch = 32;
  cfg->dtype = DPTX_DTYPE_MIXED;  // the mode that matches the reference's fp32 forward within 1e-3; BF16 is the throughput mode
  cfg->device_id = 0;
  cfg->non_negative = 1;
  cfg->ws_form = 0;
  cfg->ws_eps = 1e-8f;
}

int dptx_create(dptx_handle* out, const dptx_config* cfg) {
  if (!out || !cfg) return DPTX_E_INVALID;
  *out = nullptr;
  // max_batch <= 48: the largest activation (up-sampled head input, 37.7 MB/image) must stay below the 2 GB that a
  // 32-bit buffer offset of the direct-to-LDS loads can address; callers chunk larger batches (model.py does).
  const int max_h = cfg->max_height ? cfg->max_height : IMG, max_w = cfg->max_width ? cfg->max_width : IMG;
  if (max_h < 64 || max_w < 64 || max_h % 32 != 0 || max_w % 32 != 0 || max_h > 4096 || max_w > 4096) return DPTX_E_INVALID;
  if ((long long)cfg->max_batch * max_h * max_w * 256 >= (1ll << 31)) return DPTX_E_INVALID;  // = max_batch <= 56 at 384x384
  if (cfg->streams < 0 || cfg->streams > 4) return DPTX_E_INVALID;
  if (cfg->backbone != DPTX_BACKBONE_VITB_RN50_384 && cfg->backbone != DPTX_BACKBONE_VITL16_384) return DPTX_E_INVALID;
  // (the attention kernel's 32-bit LDS-DMA offsets over its qkv rows [B * S][3 * width] need B * S * 3 * width * 2 < 2^31: with
  //  S = h * w / 256 + 1 and width <= 1024 that is implied by the activation bound above for both backbones, so there is no
  //  separate check here; attention.hip launch_attention keeps the backstop.  ADVICE r5)
  if (cfg->backbone == DPTX_BACKBONE_VITL16_384 && cfg->dual_task) return DPTX_E_INVALID;  // the dual-task model is the hybrid
  if ((cfg->dual_task != 0 && (cfg->dual_task != 1 || cfg->num_channels != 3)) ||
      (cfg->num_channels != 1 && cfg->num_channels != 3) || cfg->max_batch < 1 || cfg->max_batch > 48 ||
      cfg->dtype < DPTX_DTYPE_BF16 || cfg->dtype > DPTX_DTYPE_FP8 || (cfg->ws_form != 0 && cfg->ws_form != 1) ||
      (cfg->flags & ~(DPTX_FLAG_NO_LN_FOLD | DPTX_FLAG_GROUP_POLICY | DPTX_FLAG_FP32_STREAM | DPTX_FLAG_NO_RANGE_CHECK | DPTX_FLAG_FP8_ALL |
                      DPTX_FLAG_FP8_VIT)) ||
      cfg->reserved != 0)
    return DPTX_E_INVALID;
  int x3_groups = 0;
  if (cfg->dtype == DPTX_DTYPE_MIXED) {
    x3_groups = cfg->x3_groups ? cfg->x3_groups : (DPTX_GROUP_ALL & ~DPTX_GROUP_VIT);
    if (x3_groups & ~DPTX_GROUP_ALL) return DPTX_E_INVALID;
    if (cfg->backbone == DPTX_BACKBONE_VITL16_384) x3_groups |= DPTX_GROUP_RESNET;  // no such layers: nothing depends on them
    // a 3-MFMA group reads the lo planes of its inputs: every producer of those must be a 3-MFMA group as well
    auto needs = [&](int g, int producers) { return !(x3_groups & g) || (x3_groups & producers) == producers; };
    if (!needs(DPTX_GROUP_EMBED, DPTX_GROUP_RESNET) || !needs(DPTX_GROUP_RN, DPTX_GROUP_RESNET | DPTX_GROUP_REASSEMBLE) ||
        !needs(DPTX_GROUP_FUSION, DPTX_GROUP_RN) || !needs(DPTX_GROUP_HEAD, DPTX_GROUP_FUSION))
      return DPTX_E_INVALID;
  }
  dptx_engine* e = new (std::nothrow) dptx_engine();
  if (!e) return DPTX_E_ALLOC;
  e->cfg = *cfg;
  e->cfg.x3_groups = x3_groups;
  e->backbone = cfg->backbone;
  if (e->backbone == DPTX_BACKBONE_VITL16_384) { e->dv = 1024; e->dm = 4096; e->nh = 16; e->depth = 24; }
  e->max_h = max_h;
  e->fp8_all = (cfg->flags & DPTX_FLAG_FP8_ALL) != 0;
  e->fp8_vit = (cfg->flags & DPTX_FLAG_FP8_VIT) != 0;
  if (cfg->dtype == DPTX_DTYPE_MIXED && cfg->x3_groups == 0 && !(cfg->flags & DPTX_FLAG_GROUP_POLICY)) {
    // default per-layer table of the decoder (oracle/precision_layers.py: per-layer sensitivities on both synthetic weight
    // families; "policy A" of profiles/r03_precision_layers.md): these nine 3x3 convolutions -- 30.6 of the decoder's 57.7
    // GMAC -- run one MFMA per product, because their operand rounding moves the output 10x less per MAC than the 1x1
    // out_convs, the deep (low-resolution) fusion stages and the head do
    for (const char* k : {"scratch.layer1_rn.weight",
                          "scratch.refinenet1.resConfUnit1.conv1.weight", "scratch.refinenet1.resConfUnit1.conv2.weight",
                          "scratch.refinenet1.resConfUnit2.conv1.weight", "scratch.refinenet1.resConfUnit2.conv2.weight",
                          "scratch.refinenet2.resConfUnit1.conv1.weight", "scratch.refinenet2.resConfUnit1.conv2.weight",
                          "scratch.refinenet3.resConfUnit1.conv1.weight", "scratch.refinenet3.resConfUnit1.conv2.weight"})
      e->layer_prec[k] = 1;
    // round 4, NOT in the default table: the first head convolution (10.9 of the head's 16.3 GMAC, 2.2 of the 19.5 ms) on TWO MFMAs
    // (dptx_set_layer_precision(h, "scratch.output_conv.0.weight", 2): the weights keep their lo plane, the input is rounded
    // once) is +3.9 % (1636 -> 1700 img/s) for rms 1.07e-4 -> 1.19e-4; the worst of the 14.2 M outputs of a B = 32 batch moves
    // from 6.3e-4 to 7.6e-4 -- inside north_star's 1e-3, but with 1.3x instead of 1.6x margin (profiles/r04_experiments.md).
    // DPTX_HEAD0_MFMAS = 1 / 2 / 3 sets the layer for A/B runs of one binary.
    const char* t2 = getenv("DPTX_HEAD0_MFMAS");
    if (t2 && atoi(t2) >= 1 && atoi(t2) <= 3) e->layer_prec["scratch.output_conv.0.weight"] = atoi(t2);
  }
  {
    // fused schedules (include/dptx.h DPTX_FLAG_*; the environment variables are for A/B runs of one binary)
    const char* t = getenv("DPTX_LN_FOLD");
    e->ln_fold = !(cfg->flags & DPTX_FLAG_NO_LN_FOLD) && !(t && t[0] == '0') && !mode_is_x3(e->mode_of(DPTX_GROUP_VIT));
    t = getenv("DPTX_STREAM16");
    e->stream16 = e->ln_fold && !(cfg->flags & DPTX_FLAG_FP32_STREAM) && !(t && t[0] == '0') &&
                  (cfg->dtype == DPTX_DTYPE_BF16 || cfg->dtype == DPTX_DTYPE_FP16 || cfg->dtype == DPTX_DTYPE_FP8);
  }
  {
    const char* t = getenv("DPTX_STREAMS");  // experiments: overrides cfg.streams
    const int ns = t ? atoi(t) : cfg->streams;
    e->n_streams = ns == 0 ? 2 : (ns < 1 ? 1 : (ns > dptx_engine::MAX_STREAMS ? dptx_engine::MAX_STREAMS : ns));
    e->auto_streams = ns == 0;
  }
  e->max_w = max_w;
  e->tok_tap_stride = (size_t)cfg->max_batch * ((size_t)max_h * max_w / 256 + 1) * (size_t)e->dv;
  e->spec = build_spec(cfg->num_channels, cfg->dual_task != 0, e->backbone);
  size_t off = BLOB_HEADER;  // the blob's layout tag (BlobHeader)
  for (size_t i = 0; i < e->spec.size(); ++i) {
    e->spec_index[e->spec[i].key] = i;
    const size_t b = packed_entry_bytes(e->spec[i]);
    if (b) {
      e->packed_off[e->spec[i].key] = off;
      off += align_up(b, 256);
    }
  }
  e->packed_single = off;
  e->packed_bytes = off * (e->two_planes() ? 2 : 1);
  e->pl.w = e->two_planes() ? (long long)(off / 2) : 0;
  plan_arena(e);
  if (cfg->device_id >= 0) {
    int n = 0;
    hipError_t r = hipGetDeviceCount(&n);
    if (r != hipSuccess || cfg->device_id >= n) {
      delete e;
      return DPTX_E_NODEVICE;
    }
  }
  *out = e;
  return DPTX_OK;
}

void dptx_destroy(dptx_handle h) {
  if (!h) return;
  if (h->cfg.device_id >= 0) {
    DeviceGuard guard(h->cfg.device_id);
    h->blob_hold.reset();   // frees the packed blob if this was its last user (hipFree in the deleter)
    if (h->d_arena) (void)hipFree(h->d_arena);
    if (h->d_tok_taps) (void)hipFree(h->d_tok_taps);
    if (h->d_amax) (void)hipFree(h->d_amax);
    if (h->d_range) (void)hipFree(h->d_range);
    for (auto ev : h->events) (void)hipEventDestroy(ev);
    for (int r = 0; r < dptx_engine::MAX_STREAMS; ++r) {
      if (h->sub_stream[r]) (void)hipStreamDestroy(h->sub_stream[r]);
      if (h->ev_join[r]) (void)hipEventDestroy(h->ev_join[r]);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  }
  delete h;
}

const char* dptx_last_error(dptx_handle h) { return h ? h->err.c_str() : "null handle"; }

int dptx_load_tensor(dptx_handle h, const char* ref_key, const float* host_fp32, const int64_t* shape, int32_t ndim) {
  if (!h || !ref_key || !host_fp32 || !shape || ndim < 1) return DPTX_E_INVALID;
  auto it = h->spec_index.find(ref_key);
  if (it == h->spec_index.end()) return h->fail(DPTX_E_KEY, std::string("unexpected key: ") + ref_key);
  const Spec& s = h->spec[it->second];
  bool ok = (size_t)ndim == s.shape.size();
  for (int i = 0; ok && i < ndim; ++i) ok = shape[i] == s.shape[i];
  if (!ok) return h->fail(DPTX_E_KEY, std::string("shape mismatch for ") + ref_key);
  if (s.role == R_UNUSED) return DPTX_OK;
  h->staged[s.key].assign(host_fp32, host_fp32 + numel(s.shape));
  h->finalized = false;
  h->device_ready = false;
  return DPTX_OK;
}

static int ensure_sub_streams(dptx_handle h);
// own_blob: the caller is about to WRITE the blob (finalize / import): a handle that shares another's gets one of its own first
static int ensure_device_memory(dptx_handle h, bool own_blob = true) {
  DeviceGuard guard(h->cfg.device_id);
  HIPCHK(h, guard.err);
  // a handle about to WRITE a blob that others read too (it shares somebody's, or somebody shares its own) takes a fresh one
  if (own_blob && h->d_blob && (h->blob_shared || h->blob_hold.use_count() > 1)) {
    h->blob_hold.reset();
    h->d_blob = nullptr;
    h->blob_shared = false;
    h->device_ready = false;
  }
  if (!h->d_blob && own_blob) {
    void* p = nullptr;
    HIPCHK(h, hipMalloc(&p, h->packed_bytes));
    const int dev = h->cfg.device_id;
    h->blob_hold = std::shared_ptr<void>(p, [dev](void* q) {
      DeviceGuard g(dev);
      (void)hipFree(q);
    });
    h->d_blob = (char*)p;
  }
  if (!h->d_arena) HIPCHK(h, hipMalloc((void**)&h->d_arena, h->arena_bytes));
  if (!h->d_range) {
    HIPCHK(h, hipMalloc((void**)&h->d_range, 256));
    HIPCHK(h, hipMemset(h->d_range, 0, 256));
  }
  return ensure_sub_streams(h);
}

int dptx_finalize_weights(dptx_handle h) {
  if (!h) return DPTX_E_INVALID;
  int r = pack_host(h);
  if (r != DPTX_OK) return r;
  if (h->cfg.device_id < 0) return DPTX_OK;
  r = ensure_device_memory(h);
  if (r != DPTX_OK) return r;
  DeviceGuard guard(h->cfg.device_id);
  HIPCHK(h, guard.err);
  HIPCHK(h, hipMemcpy(h->d_blob, h->host_blob.data(), h->packed_bytes, hipMemcpyHostToDevice));
  h->device_ready = true;
  h->staged.clear();  // fp32 staging copies are no longer needed
  return DPTX_OK;
}

size_t dptx_packed_bytes(dptx_handle h) { return h ? h->packed_bytes : 0; }

int dptx_export_packed_host(dptx_handle h, void* dst_host, size_t bytes) {
  if (!h || !dst_host) return DPTX_E_INVALID;
  if (!h->finalized) return h->fail(DPTX_E_INVALID, "export before dptx_finalize_weights");
  if (bytes < h->packed_bytes) return h->fail(DPTX_E_INVALID, "export buffer too small");
  memcpy(dst_host, h->host_blob.data(), h->packed_bytes);
  return DPTX_OK;
}

int dptx_export_packed_device(dptx_handle h, void* dst_dev, size_t bytes, void* stream) {
  if (!h || !dst_dev) return DPTX_E_INVALID;
  if (h->cfg.device_id < 0) return h->fail(DPTX_E_NODEVICE, "host-only handle");
  if (!h->device_ready) return h->fail(DPTX_E_INVALID, "export before weights are on the device");
  if (bytes < h->packed_bytes) return h->fail(DPTX_E_INVALID, "export buffer too small");
  DeviceGuard guard(h->cfg.device_id);
  HIPCHK(h, guard.err);
  HIPCHK(h, hipMemcpyAsync(dst_dev, h->d_blob, h->packed_bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return DPTX_OK;
}

int dptx_import_packed_device(dptx_handle h, const void* src_dev, size_t bytes, void* stream) {
  if (!h || !src_dev) return DPTX_E_INVALID;
  if (h->cfg.device_id < 0) return h->fail(DPTX_E_NODEVICE, "host-only handle");
  if (bytes != h->packed_bytes) return h->fail(DPTX_E_INVALID, "packed blob size mismatch (different config/build?)");
  DeviceGuard guard(h->cfg.device_id);
  HIPCHK(h, guard.err);
  {
    // the blob's layout tag must be the one this handle packs itself (dtype, backbone, LayerNorm fold, weight-std form ...).
    // Checked BEFORE the handle's state is touched: a rejected import leaves a working handle -- its own blob,
    // or the one it shares -- exactly as it was
    BlobHeader got;
    HIPCHK(h, hipMemcpyAsync(&got, src_dev, sizeof got, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
    const BlobHeader want = h->blob_header();
    if (memcmp(&got, &want, sizeof got) != 0) {
      char msg[320];
      snprintf(msg, sizeof msg,
               "packed blob was packed for another layout: magic %.8s version %u dtype %d backbone %d dual %d channels %d ln_fold %d "
               "ws_form %d (this handle: version %u dtype %d backbone %d dual %d channels %d ln_fold %d ws_form %d)",
               got.magic, got.version, got.dtype, got.backbone, got.dual_task, got.num_channels, got.ln_fold, got.ws_form,
               want.version, want.dtype, want.backbone, want.dual_task, want.num_channels, want.ln_fold, want.ws_form);
      return h->fail(DPTX_E_INVALID, msg);
    }
  }
  const int r = ensure_device_memory(h);   // un-shares / allocates: only now that the source is known to be importable
  if (r != DPTX_OK) return r;
  HIPCHK(h, hipMemcpyAsync(h->d_blob, src_dev, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  h->device_ready = true;
  return DPTX_OK;
}

int dptx_share_packed(dptx_handle dst, dptx_handle src) {
  if (!dst || !src || dst == src) return DPTX_E_INVALID;
  if (dst->cfg.device_id < 0 || src->cfg.device_id < 0) return dst->fail(DPTX_E_NODEVICE, "host-only handle");
  if (dst->cfg.device_id != src->cfg.device_id) return dst->fail(DPTX_E_INVALID, "dptx_share_packed: the handles live on different devices");
  if (!src->device_ready || !src->d_blob) return dst->fail(DPTX_E_INVALID, "dptx_share_packed: the source handle has no weights on the device");
  const BlobHeader a = dst->blob_header(), b = src->blob_header();
  if (dst->packed_bytes != src->packed_bytes || memcmp(&a, &b, sizeof a) != 0)
    return dst->fail(DPTX_E_INVALID, "dptx_share_packed: the handles pack different blobs (dtype / backbone / channels / dual task / fold)");
  DeviceGuard guard(dst->cfg.device_id);
  HIPCHK(dst, guard.err);
  dst->blob_hold = src->blob_hold;   // (a blob of dst's own is released here if nobody else holds it)
  dst->d_blob = src->d_blob;
  dst->blob_shared = true;
  const int r = ensure_device_memory(dst, false);   // arena + range flag
  if (r != DPTX_OK) return r;
  dst->device_ready = true;
  return DPTX_OK;
}

size_t dptx_workspace_bytes(dptx_handle h) { return h ? h->packed_bytes + h->arena_bytes : 0; }

int dptx_enable_taps(dptx_handle h, int on) {
  if (!h) return DPTX_E_INVALID;
  if (h->cfg.device_id < 0) return h->fail(DPTX_E_NODEVICE, "host-only handle");
  if (on && !h->d_tok_taps) {
    DeviceGuard guard(h->cfg.device_id);
  HIPCHK(h, guard.err);
    HIPCHK(h, hipMalloc((void**)&h->d_tok_taps, (size_t)(h->depth + 1) * h->tok_tap_stride * 4));
  }
  h->taps_on = on != 0;
  return DPTX_OK;
}

// Runs one forward: either one Run over the whole batch on the caller's stream, or -- two streams, no taps, no per-launch
// timing, batch >= 2 -- two Runs over the two halves of the batch on the handle's two internal streams, forked from and
// joined to the caller's stream with events.  Images are independent and every kernel is batch-invariant bit for bit, so
// both schedules return the same bits; the second one lets the MFMA-bound launches of one half overlap the HBM-bound
// launches and the tails of the other.
// The internal streams of the sub-batch schedule.  Created when the handle gets its device memory (round 5), not at the first
// forward: HIP maps streams onto hardware queues as they are created, and a handle whose streams came into being AFTER a
// process had created many others (torch's stream pool) was measured with its two half-batch runs no longer overlapping --
// 1990 instead of 2600 images/s (profiles/r05_experiments.md).
static int ensure_sub_streams(dptx_handle h) {
  if (h->ev_fork || h->n_streams < 2) return DPTX_OK;
  // One stream PRIORITY per sub-batch stream, cycling through the device's range: the runtime keeps a separate pool of
  // hardware queues per priority, so streams of different priorities can never be mapped onto the same queue -- which is what
  // happens to same-priority streams once a process holds more streams than queues (GPU_MAX_HW_QUEUES, 4), and then the two
  // half-batch runs execute one after the other (1974 instead of 2600 images/s, lease r5l4).  The priorities only order the
  // dispatch of ready work between the halves; both halves are needed before the forward is complete.
  int pri_least = 0, pri_greatest = 0;
  HIPCHK(h, hipDeviceGetStreamPriorityRange(&pri_least, &pri_greatest));   // numerically: least >= greatest
  const int levels = pri_least - pri_greatest + 1;
  for (int r = 0; r < h->n_streams && r < dptx_engine::MAX_STREAMS; ++r) {
    const int pri = levels > 1 ? pri_least - (r % levels) : pri_least;
    HIPCHK(h, hipStreamCreateWithPriority(&h->sub_stream[r], hipStreamNonBlocking, pri));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_join[r], hipEventDisableTiming));
  }
  HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
  return DPTX_OK;
}

static int run_forward(dptx_handle h, const void* x, int io, void* y, void* y2, int batch, int height, int width,
                       hipStream_t stream) {
  const int C = h->cfg.num_channels;
  const size_t esz = io == DPTX_IO_FP32 ? 4 : 2;  // bytes per element of the caller's buffers
  const bool want_split = h->force_schedule >= 0 ? h->force_schedule == 1 : (!h->auto_streams || (h->split_tuned && h->split_on));
  const bool split = want_split && h->n_streams >= 2 && batch >= 2 && !h->taps_on && !h->profiling && !h->calibrating;
  // tile selection of the GEMMs (kernels.h gemm_set_cu_share): a sub-batch run shares the chip with the other streams' runs.
  // Measured (profiles/r03_experiments.md, two streams): the 256x256 rule judged against 1 / streams of the chip (the
  // half-batch qkv, proj and fc2 GEMMs then take the ping-pong kernel) -- with the lockstep-epilogue kernel of the first half
  // of round 3 0.7 was the optimum (+1.2 %) and 0.5 lost; with the persistent, register-direct form 0.5 wins on boxes that
  // are not power-bound (+1.3 ... 2.9 % together with the 1.5 x per-tile advantage in launch_dt) and is neutral on those
  // that are.  Scaled narrow-tile thresholds lose.  DPTX_CU_SHARE / DPTX_CU_SHARE_SMALL override (1 = tile every launch for
  // the whole chip, as rounds 1-2 did)
  static float share_env = -1.f, share_small_env = -1.f;
  if (share_env < 0.f) { const char* t = getenv("DPTX_CU_SHARE"); share_env = t ? (float)atof(t) : 0.f; }
  if (share_small_env < 0.f) { const char* t = getenv("DPTX_CU_SHARE_SMALL"); share_small_env = t ? (float)atof(t) : 0.f; }
  // whatever path leaves this function (HIPCHK returns included), the thread's tile-selection state is back at "whole chip"
  struct ShareReset { ~ShareReset() { gemm_set_cu_share(1.0f); } } share_reset;
  if (!split) {
    // DPTX_CU_SHARE_WHOLE (A/B runs): tile selection of a whole-batch run that shares the GPU with another handle's forward
    // (omnidata_amd/pipeline.py) -- measured neutral, profiles/r05_experiments.md
    static float whole_env = -1.f;
    if (whole_env < 0.f) { const char* t = getenv("DPTX_CU_SHARE_WHOLE"); whole_env = t ? (float)atof(t) : 0.f; }
    gemm_set_cu_share(whole_env > 0.f ? whole_env : 1.0f);
    Run run{h, batch, stream, h->cfg.dtype, height, width, io};
    const int rc = run.forward(x, y, y2);
    h->launches = run.launches;
    h->exec_macs = run.exec_macs;
    for (int c = 0; c < 4; ++c) h->cat_macs[c] = run.cat_macs[c];
    h->last_batch = batch;
    h->last_regions = 1;
    return rc;
  }
  const int nr = batch < h->n_streams ? batch : h->n_streams;  // sub-batches: the first (batch % nr) get one image more
  gemm_set_cu_share(share_env > 0.f ? share_env : 1.0f / (float)nr, share_small_env > 0.f ? share_small_env : 1.0f);
  {
    const int rs = ensure_sub_streams(h);
    if (rs != DPTX_OK) return rs;
  }
  HIPCHK(h, hipEventRecord(h->ev_fork, stream));
  const size_t px = (size_t)height * width;
  int rc = DPTX_OK;
  int64_t launches = 0;
  size_t first = 0;
  for (int r = 0; r < nr; ++r) {
    const int nb = batch / nr + (r < batch % nr ? 1 : 0);
    HIPCHK(h, hipStreamWaitEvent(h->sub_stream[r], h->ev_fork, 0));
    Run run{h, nb, h->sub_stream[r], h->cfg.dtype, height, width, io};
    run.abase = (size_t)r * h->half_region;
    run.half = true;
    const int rr = run.forward((const char*)x + first * 3 * px * esz, (char*)y + first * C * px * esz,
                               y2 ? (char*)y2 + first * px * esz : nullptr);
    if (rc == DPTX_OK) rc = rr;
    launches += run.launches;
    if (r == 0) {
      h->exec_macs = run.exec_macs;
      for (int c = 0; c < 4; ++c) h->cat_macs[c] = run.cat_macs[c];
    }
    HIPCHK(h, hipEventRecord(h->ev_join[r], h->sub_stream[r]));
    HIPCHK(h, hipStreamWaitEvent(stream, h->ev_join[r], 0));
    first += (size_t)nb;
  }
  gemm_set_cu_share(1.0f);
  h->taps.clear();  // stage taps describe whole-batch runs only (dptx_enable_taps makes the forward single-pass)
  h->launches = launches;
  h->last_batch = batch;
  h->last_regions = nr;
  return rc;
}

int dptx_forward(dptx_handle h, const void* x_dev, int32_t x_dtype, void* y_dev, int32_t batch, void* stream) {
  return dptx_forward_hw(h, x_dev, x_dtype, y_dev, batch, IMG, IMG, stream);
}

int dptx_forward_hw(dptx_handle h, const void* x_dev, int32_t x_dtype, void* y_dev, int32_t batch, int32_t height,
                    int32_t width, void* stream) {
  if (!h || !x_dev || !y_dev) return DPTX_E_INVALID;
  if (h->cfg.device_id < 0) return h->fail(DPTX_E_NODEVICE, "dptx_forward on a host-only handle");
  if (!h->device_ready) return h->fail(DPTX_E_INVALID, "dptx_forward before weights were finalized/imported");
  if (batch < 1 || batch > h->cfg.max_batch) return h->fail(DPTX_E_INVALID, "batch out of range [1, max_batch]");
  if (x_dtype != DPTX_IO_FP32 && x_dtype != DPTX_IO_BF16 && x_dtype != DPTX_IO_FP16)
    return h->fail(DPTX_E_INVALID, "unsupported x_dtype (DPTX_IO_FP32 / DPTX_IO_BF16 / DPTX_IO_FP16)");
  DeviceGuard guard(h->cfg.device_id);
  HIPCHK(h, guard.err);
  if (height < 64 || width < 64 || height % 32 != 0 || width % 32 != 0)
    return h->fail(DPTX_E_INVALID, "input height/width must be multiples of 32, >= 64");
  if ((long long)height * width > (long long)h->max_h * h->max_w)
    return h->fail(DPTX_E_INVALID, "input larger than the engine was planned for (dptx_config.max_height/max_width)");
  if (h->cfg.dual_task) return h->fail(DPTX_E_INVALID, "dual-task handle: call dptx_forward_dual");
  return run_forward(h, x_dev, x_dtype, y_dev, nullptr, batch, height, width, (hipStream_t)stream);
}

int dptx_forward_dual(dptx_handle h, const void* x_dev, int32_t x_dtype, void* y_normal_dev, void* y_depth_dev, int32_t batch,
                      int32_t height, int32_t width, void* stream) {
  if (!h || !x_dev || !y_normal_dev || !y_depth_dev) return DPTX_E_INVALID;
  if (h->cfg.device_id < 0) return h->fail(DPTX_E_NODEVICE, "dptx_forward_dual on a host-only handle");
  if (!h->cfg.dual_task) return h->fail(DPTX_E_INVALID, "dptx_forward_dual needs a handle created with dual_task = 1");
  if (!h->device_ready) return h->fail(DPTX_E_INVALID, "dptx_forward_dual before weights were finalized/imported");
  if (batch < 1 || batch > h->cfg.max_batch) return h->fail(DPTX_E_INVALID, "batch out of range [1, max_batch]");
  if (x_dtype != DPTX_IO_FP32 && x_dtype != DPTX_IO_BF16 && x_dtype != DPTX_IO_FP16)
    return h->fail(DPTX_E_INVALID, "unsupported x_dtype (DPTX_IO_FP32 / DPTX_IO_BF16 / DPTX_IO_FP16)");
  if (height < 64 || width < 64 || height % 32 != 0 || width % 32 != 0)
    return h->fail(DPTX_E_INVALID, "input height/width must be multiples of 32, >= 64");
  if ((long long)height * width > (long long)h->max_h * h->max_w)
    return h->fail(DPTX_E_INVALID, "input larger than the engine was planned for (dptx_config.max_height/max_width)");
  DeviceGuard guard(h->cfg.device_id);
  HIPCHK(h, guard.err);
  return run_forward(h, x_dev, x_dtype, y_normal_dev, y_depth_dev, batch, height, width, (hipStream_t)stream);
}

int dptx_tune_schedule(dptx_handle h, const void* x_dev, int32_t x_dtype, void* y_dev, void* y_depth_dev, int32_t batch,
                       int32_t height, int32_t width, int32_t reps, void* stream) {
  if (!h || !x_dev || !y_dev) return DPTX_E_INVALID;
  if (h->cfg.device_id < 0) return h->fail(DPTX_E_NODEVICE, "dptx_tune_schedule on a host-only handle");
  if (!h->device_ready) return h->fail(DPTX_E_INVALID, "dptx_tune_schedule before weights were finalized/imported");
  if ((h->cfg.dual_task != 0) != (y_depth_dev != nullptr)) return h->fail(DPTX_E_INVALID, "dptx_tune_schedule: y_depth_dev is for dual-task handles");
  if (!h->auto_streams || h->n_streams < 2 || batch < 2) return DPTX_OK;   // nothing to choose
  if (reps < 1) reps = 2;
  DeviceGuard guard(h->cfg.device_id);
  HIPCHK(h, guard.err);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIPCHK(h, hipEventCreate(&e0));
  HIPCHK(h, hipEventCreate(&e1));
  struct Cleanup { dptx_handle h; hipEvent_t a, b; ~Cleanup() { h->force_schedule = -1; (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } cleanup{h, e0, e1};
  float ms[2] = {0.f, 0.f};
  auto fwd = [&]() {
    return h->cfg.dual_task ? dptx_forward_dual(h, x_dev, x_dtype, y_dev, y_depth_dev, batch, height, width, stream)
                            : dptx_forward_hw(h, x_dev, x_dtype, y_dev, batch, height, width, stream);
  };
  for (int cand = 0; cand < 2; ++cand) {
    h->force_schedule = cand;
    int rc = fwd();   // untimed: first-use costs (kernel attributes, stream wake-up)
    if (rc != DPTX_OK) return rc;
    HIPCHK(h, hipEventRecord(e0, (hipStream_t)stream));
    for (int r = 0; r < reps; ++r) {
      rc = fwd();
      if (rc != DPTX_OK) return rc;
    }
    HIPCHK(h, hipEventRecord(e1, (hipStream_t)stream));
    HIPCHK(h, hipEventSynchronize(e1));
    HIPCHK(h, hipEventElapsedTime(&ms[cand], e0, e1));
    ms[cand] /= (float)reps;
  }
  h->tune_ms_single = ms[0];
  h->tune_ms_split = ms[1];
  h->split_on = ms[1] < 0.97f * ms[0];   // the two-stream schedule has to EARN its place: ties go to the simpler one
  h->split_tuned = true;
  return DPTX_OK;
}

int dptx_schedule_info(dptx_handle h, int32_t* split, int32_t* tuned, float* ms_single, float* ms_split) {
  if (!h) return DPTX_E_INVALID;
  if (split) *split = h->auto_streams ? (int32_t)(h->split_tuned && h->split_on) : (int32_t)(h->n_streams >= 2);
  if (tuned) *tuned = (int32_t)(h->auto_streams && h->split_tuned);
  if (ms_single) *ms_single = h->tune_ms_single;
  if (ms_split) *ms_split = h->tune_ms_split;
  return DPTX_OK;
}

int dptx_probe_stream_overlap(int32_t device_id, void* stream_a, void* stream_b, int32_t spin_us, float* ratio) {
  if (!ratio || device_id < 0) return DPTX_E_INVALID;
  DeviceGuard guard(device_id);
  if (guard.err != hipSuccess) return DPTX_E_HIP;
  hipStream_t a = (hipStream_t)stream_a, b = (hipStream_t)stream_b;
  const long long ticks = (long long)(spin_us > 0 ? spin_us : 2000) * 100;   // s_memrealtime: 100 MHz
  auto run = [&](bool both, double& sec) -> bool {
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return false;
    const auto t0 = std::chrono::steady_clock::now();
    if (launch_spin(ticks, a) != hipSuccess) return false;
    if (both && launch_spin(ticks, b) != hipSuccess) return false;
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return false;
    sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return true;
  };
  double warm = 0, alone = 0, pair = 0;
  if (!run(true, warm) || !run(false, alone) || !run(true, pair) || alone <= 0) return DPTX_E_HIP;
  *ratio = (float)(pair / alone);
  return DPTX_OK;
}

int dptx_set_layer_precision(dptx_handle h, const char* conv_weight_key, int32_t mfmas) {
  if (!h || !conv_weight_key || mfmas < 1 || mfmas > 3) return DPTX_E_INVALID;
  if (!h->mixed()) return h->fail(DPTX_E_INVALID, "per-layer precision applies to dtype = DPTX_DTYPE_MIXED");
  std::string k = conv_weight_key;
  if (k.compare(0, 6, "depth.") == 0) k = k.substr(6);
  const bool decoder = k.compare(0, 8, "scratch.") == 0 && k.size() > 7 && k.compare(k.size() - 7, 7, ".weight") == 0 &&
                       k.find("output_conv.4") == std::string::npos && h->spec_index.count(k) && h->spec[h->spec_index[k]].role == R_CONV;
  if (!decoder) return h->fail(DPTX_E_KEY, std::string("not a decoder convolution (scratch.*): ") + conv_weight_key);
  h->layer_prec[k] = mfmas;
  return DPTX_OK;
}

int dptx_calibrate_fp8(dptx_handle h, const void* x_dev, int32_t x_dtype, void* y_dev, void* y2_dev, int32_t batch, int32_t height,
                       int32_t width, void* stream) {
  if (!h || !x_dev || !y_dev) return DPTX_E_INVALID;
  if (!h->fp8()) return h->fail(DPTX_E_INVALID, "dptx_calibrate_fp8 needs a handle created with dtype = DPTX_DTYPE_FP8");
  if (h->cfg.device_id < 0) return h->fail(DPTX_E_NODEVICE, "dptx_calibrate_fp8 on a host-only handle");
  if (!h->device_ready) return h->fail(DPTX_E_INVALID, "dptx_calibrate_fp8 before weights were finalized/imported");
  if (batch < 1 || batch > h->cfg.max_batch) return h->fail(DPTX_E_INVALID, "batch out of range [1, max_batch]");
  if (x_dtype != DPTX_IO_FP32 && x_dtype != DPTX_IO_BF16 && x_dtype != DPTX_IO_FP16) return h->fail(DPTX_E_INVALID, "unsupported x_dtype");
  if (height < 64 || width < 64 || height % 32 != 0 || width % 32 != 0 || (long long)height * width > (long long)h->max_h * h->max_w)
    return h->fail(DPTX_E_INVALID, "input height/width must be multiples of 32, >= 64, within the planned size");
  if ((h->cfg.dual_task != 0) != (y2_dev != nullptr)) return h->fail(DPTX_E_INVALID, "y2_dev: the depth output of a dual-task handle");
  DeviceGuard guard(h->cfg.device_id);
  HIPCHK(h, guard.err);
  hipStream_t st = (hipStream_t)stream;
  if (!h->d_amax) HIPCHK(h, hipMalloc((void**)&h->d_amax, dptx_engine::MAX_Q8 * sizeof(unsigned)));
  HIPCHK(h, hipMemsetAsync(h->d_amax, 0, dptx_engine::MAX_Q8 * sizeof(unsigned), st));
  h->calibrating = true;
  const int rc = run_forward(h, x_dev, x_dtype, y_dev, y2_dev, batch, height, width, st);
  h->calibrating = false;
  if (rc != DPTX_OK) return rc;
  unsigned bits[dptx_engine::MAX_Q8];
  HIPCHK(h, hipMemcpyAsync(bits, h->d_amax, sizeof bits, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  for (int i = 0; i < dptx_engine::MAX_Q8; ++i) {
    float a;
    memcpy(&a, &bits[i], 4);
    h->act_amax[i] = a;
    // power of two that puts the calibration maximum into (112, 224]: one binade of headroom below e4m3's 448
    int k = (a > 0.f && std::isfinite(a)) ? (int)std::floor(std::log2(224.0 / (double)a)) : 0;
    k = k < -24 ? -24 : (k > 24 ? 24 : k);
    h->act_scale[i] = std::ldexp(1.0f, k);
  }
  h->calibrated = true;
  return DPTX_OK;
}

int dptx_fp8_get_calibration(dptx_handle h, float* scales, float* amax, int32_t capacity) {
  if (!h) return DPTX_E_INVALID;
  if (!h->fp8()) return h->fail(DPTX_E_INVALID, "not an fp8 handle");
  const int n = h->n_q8;
  if (capacity < n) return h->fail(DPTX_E_INVALID, "capacity < number of e4m3 tensors of the last forward");
  for (int i = 0; i < n; ++i) {
    if (scales) scales[i] = h->act_scale[i];
    if (amax) amax[i] = h->act_amax[i];
  }
  return n;
}

int dptx_fp8_set_calibration(dptx_handle h, const float* scales, int32_t n) {
  if (!h || !scales || n < 0 || n > dptx_engine::MAX_Q8) return DPTX_E_INVALID;
  if (!h->fp8()) return h->fail(DPTX_E_INVALID, "not an fp8 handle");
  for (int i = 0; i < n; ++i) {
    int e2 = 0;
    const float m = std::frexp(scales[i], &e2);
    if (!(scales[i] > 0.f) || m != 0.5f) return h->fail(DPTX_E_INVALID, "activation scales must be positive powers of two");
    h->act_scale[i] = scales[i];
  }
  h->calibrated = true;
  return DPTX_OK;
}

int dptx_range_status(dptx_handle h, int32_t* nonfinite, int32_t reset, void* stream) {
  if (!h || !nonfinite) return DPTX_E_INVALID;
  *nonfinite = 0;
  if (h->cfg.device_id < 0) return h->fail(DPTX_E_NODEVICE, "host-only handle");
  if (!h->range_check() || !h->d_range) return DPTX_OK;  // bf16-plane dtypes have fp32's range: nothing to report
  DeviceGuard guard(h->cfg.device_id);
  HIPCHK(h, guard.err);
  unsigned v = 0;
  HIPCHK(h, hipMemcpyAsync(&v, h->d_range, sizeof v, hipMemcpyDeviceToHost, (hipStream_t)stream));
  if (reset) HIPCHK(h, hipMemsetAsync(h->d_range, 0, sizeof v, (hipStream_t)stream));
  HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
  *nonfinite = v != 0;
  return DPTX_OK;
}

int dptx_tap(dptx_handle h, const char* name, float* dst_host, size_t capacity_floats, int64_t shape4[4]) {
  if (!h || !name || !dst_host || !shape4) return DPTX_E_INVALID;
  auto it = h->taps.find(name);
  if (it == h->taps.end()) return h->fail(DPTX_E_KEY, std::string("unknown or unavailable tap: ") + name);
  const TapInfo& t = it->second;
  size_t n = 1;
  for (int i = 0; i < 4; ++i) { shape4[i] = t.shape[i]; n *= (size_t)t.shape[i]; }
  if (capacity_floats < n) return h->fail(DPTX_E_INVALID, "tap buffer too small");
  DeviceGuard guard(h->cfg.device_id);
  HIPCHK(h, guard.err);
  HIPCHK(h, hipDeviceSynchronize());
  if (t.fp32) {
    HIPCHK(h, hipMemcpy(dst_host, t.ptr, n * 4, hipMemcpyDeviceToHost));
  } else {
    float* tmp = nullptr;
    HIPCHK(h, hipMalloc((void**)&tmp, n * 4));
    hipError_t r = launch_to_f32(t.mode, t.ptr, tmp, n, h->pl, nullptr);
    if (r == hipSuccess) r = hipMemcpy(dst_host, tmp, n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(tmp);
    if (r != hipSuccess) return h->fail(DPTX_E_HIP, hipGetErrorString(r));
  }
  return DPTX_OK;
}

int dptx_forward_info(dptx_handle h, int64_t* launches, double* algorithmic_macs, double* executed_macs) {
  if (!h) return DPTX_E_INVALID;
  if (launches) *launches = h->launches;
  if (algorithmic_macs)  // SURVEY.md 8d (dual: 69.96 shared + 2 x 57.67 per image pair)
    *algorithmic_macs = h->backbone == DPTX_BACKBONE_VITL16_384
                            ? (h->cfg.num_channels == 3 ? 258.216e9 : 258.206e9)  // DPT-Large: 200.12 encoder + 58.08 decoder
                            : (h->cfg.dual_task ? 185.29e9 : (h->cfg.num_channels == 3 ? 127.624e9 : 127.615e9));
  if (executed_macs) *executed_macs = h->exec_macs;
  return DPTX_OK;
}

int dptx_set_profiling(dptx_handle h, int on) {
  if (!h) return DPTX_E_INVALID;
  if (h->cfg.device_id < 0) return h->fail(DPTX_E_NODEVICE, "host-only handle");
  h->profiling = on != 0;
  return DPTX_OK;
}

int dptx_profile_get(dptx_handle h, int32_t category, double* ms, int64_t* launches, double* macs_per_image) {
  if (!h || category < 0 || category > 3) return DPTX_E_INVALID;
  if (h->event_cat.empty()) return h->fail(DPTX_E_INVALID, "no profiled forward recorded");
  DeviceGuard guard(h->cfg.device_id);
  HIPCHK(h, guard.err);
  HIPCHK(h, hipEventSynchronize(h->events[h->event_cat.size()]));
  double t = 0.0;
  int64_t n = 0;
  for (size_t i = 0; i < h->event_cat.size(); ++i) {
    if (h->event_cat[i] != category) continue;
    float dt = 0.f;
    HIPCHK(h, hipEventElapsedTime(&dt, h->events[i], h->events[i + 1]));
    t += dt;
    ++n;
  }
  if (ms) *ms = t;
  if (launches) *launches = n;
  if (macs_per_image) *macs_per_image = h->cat_macs[category];
  return DPTX_OK;
}

int dptx_profile_dump(dptx_handle h, const char* path) {
  if (!h || !path) return DPTX_E_INVALID;
  if (h->event_cat.empty()) return h->fail(DPTX_E_INVALID, "no profiled forward recorded");
  DeviceGuard guard(h->cfg.device_id);
  HIPCHK(h, guard.err);
  HIPCHK(h, hipEventSynchronize(h->events[h->event_cat.size()]));
  FILE* fp = fopen(path, "w");
  if (!fp) return h->fail(DPTX_E_INVALID, std::string("cannot open ") + path);
  fprintf(fp, "idx,category,name,ms\n");
  for (size_t i = 0; i < h->event_cat.size(); ++i) {
    float dt = 0.f;
    (void)hipEventElapsedTime(&dt, h->events[i], h->events[i + 1]);
    fprintf(fp, "%zu,%d,%s,%.5f\n", i, h->event_cat[i], h->event_name[i].c_str(), dt);
  }
  fclose(fp);
  return DPTX_OK;
}

// --------------------------------------------------------------------- op-level entry points
static Planes g_op_planes{0, 0};
int dptx_op_set_planes(int64_t act_plane_elems, int64_t w_plane_elems) {
  g_op_planes.act = act_plane_elems;
  g_op_planes.w = w_plane_elems;
  return DPTX_OK;
}

int dptx_op_gemm(int32_t dtype, const void* A, const void* W, const float* bias, const void* R, void* C, int32_t M, int32_t N,
                 int32_t K, int32_t act, int32_t a_fp32, int32_t c_fp32, int32_t r_fp32, void* stream) {
  GemmParams p;
  gemm_params_dense(p, M, N, K);
  p.A = A; p.W = W; p.C = C; p.bias = bias; p.R1 = R; p.act = act; p.a_fp32 = a_fp32; p.c_fp32 = c_fp32; p.r1_fp32 = r_fp32;
  p.planes = g_op_planes;
  return launch_gemm(dtype, p, (hipStream_t)stream) == hipSuccess ? DPTX_OK : DPTX_E_HIP;
}

// dense GEMM with the LayerNorm fold's CONSUMER epilogue (tools/gpu/r4_micro.py, tests): C = act((A W^T - mu colsum) rstd + bias),
// (mu, rstd) of row m from the (sum, sum of squares) records ln_stats[m][0 .. ln_nblk) (row stride 8 records)
int dptx_op_gemm_ln(int32_t dtype, const void* A, const void* W, const float* bias, void* C, int32_t M, int32_t N, int32_t K,
                    int32_t act, const float* ln_stats, const float* ln_colsum, int32_t ln_nblk, float ln_eps, void* stream) {
  GemmParams p;
  gemm_params_dense(p, M, N, K);
  p.A = A; p.W = W; p.C = C; p.bias = bias; p.act = act; p.planes = g_op_planes;
  p.ln_stats = ln_stats; p.ln_colsum = ln_colsum; p.ln_nblk = ln_nblk; p.ln_eps = ln_eps; p.ln_inv_dim = 1.0f / (float)K;
  return launch_gemm(dtype, p, (hipStream_t)stream) == hipSuccess ? DPTX_OK : DPTX_E_HIP;
}

// dense GEMM with the LayerNorm fold's PRODUCER epilogue on the 16-bit token stream (what the proj / fc2 launches run):
// C <- C + A W^T + bias in place, and (sum, sum of squares) of every new row per 128-column block into row_stats[m][0 .. N / 128)
int dptx_op_gemm_stream(int32_t dtype, const void* A, const void* W, const float* bias, void* C, float* row_stats, int32_t M,
                        int32_t N, int32_t K, void* stream) {
  GemmParams p;
  gemm_params_dense(p, M, N, K);
  p.A = A; p.W = W; p.C = C; p.R1 = C; p.bias = bias; p.planes = g_op_planes;
  p.row_stats = row_stats; p.stats_nblk = 8;
  return launch_gemm(dtype, p, (hipStream_t)stream) == hipSuccess ? DPTX_OK : DPTX_E_HIP;
}

// the same on the fp32 token stream of the parity mode: X (fp32) <- X + A W^T + bias in place, its 16-bit image into C16
int dptx_op_gemm_stream32(int32_t dtype, const void* A, const void* W, const float* bias, float* X, void* C16, float* row_stats,
                          int32_t M, int32_t N, int32_t K, void* stream) {
  GemmParams p;
  gemm_params_dense(p, M, N, K);
  p.A = A; p.W = W; p.C = X; p.c_fp32 = 1; p.R1 = X; p.r1_fp32 = 1; p.C16 = C16; p.bias = bias; p.planes = g_op_planes;
  p.row_stats = row_stats; p.stats_nblk = 8;
  return launch_gemm(dtype, p, (hipStream_t)stream) == hipSuccess ? DPTX_OK : DPTX_E_HIP;
}

int dptx_op_head_tail(int32_t dtype, const void* H0, const void* W2, const float* b2, const float* w4, const float* b4, float* y,
                      int32_t B, int32_t Hs, int32_t Ws, int32_t C, int32_t relu_out, void* stream) {
  return launch_head_tail(dtype, H0, W2, b2, w4, b4, y, DPTX_IO_FP32, B, Hs, Ws, C, relu_out, (hipStream_t)stream, g_op_planes) == hipSuccess
             ? DPTX_OK
             : DPTX_E_HIP;
}

int dptx_op_conv(int32_t dtype, const void* X, const void* Wt, const float* bias, const void* R, void* Y, int32_t B, int32_t H,
                 int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t stride, int32_t pad_t, int32_t pad_l, int32_t Ho,
                 int32_t Wo, int32_t a_relu, int32_t act, void* stream) {
  GemmParams p{};
  p.A = X; p.W = Wt; p.C = Y; p.bias = bias; p.R1 = R;
  p.M = B * Ho * Wo; p.N = Cout; p.K = ksize * ksize * Cin; p.ldw = p.K;
  p.a_rpi = Ho * Wo; p.Wout = Wo; p.Hin = H; p.Win = W; p.Cin = Cin; p.a_pix_stride = Cin;
  p.a_img_stride = (long long)H * W * Cin;
  p.a_bytes = (long long)B * H * W * Cin * 2;
  p.ksz = ksize; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
  p.c_rpi = 0x7fffffff; p.ldc = Cout; p.act = act; p.a_relu = a_relu; p.planes = g_op_planes;
  p.k_tap_fast = (ksize == 3 && Cin >= 512) ? 1 : 0;
  return launch_gemm(dtype, p, (hipStream_t)stream) == hipSuccess ? DPTX_OK : DPTX_E_HIP;
}

int dptx_op_conv_planes(int32_t dtype, const void* X, const void* Wt, const float* bias, const void* R, void* Y, int32_t B, int32_t H,
                        int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t stride, int32_t pad_t, int32_t pad_l, int32_t Ho,
                        int32_t Wo, int32_t a_relu, int32_t act, int32_t epi2, int32_t c_hi_only, int32_t r1_hi_only, void* stream) {
  GemmParams p{};
  p.A = X; p.W = Wt; p.C = Y; p.bias = bias; p.R1 = R;
  p.M = B * Ho * Wo; p.N = Cout; p.K = ksize * ksize * Cin; p.ldw = p.K;
  p.a_rpi = Ho * Wo; p.Wout = Wo; p.Hin = H; p.Win = W; p.Cin = Cin; p.a_pix_stride = Cin;
  p.a_img_stride = (long long)H * W * Cin;
  p.a_bytes = (long long)B * H * W * Cin * 2;
  p.ksz = ksize; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
  p.c_rpi = 0x7fffffff; p.ldc = Cout; p.act = act; p.a_relu = a_relu; p.planes = g_op_planes;
  p.k_tap_fast = (ksize == 3 && Cin >= 512) ? 1 : 0;
  p.epi2 = epi2 & 1; p.a_hi_only = (epi2 >> 1) & 1; p.c_hi_only = c_hi_only; p.r1_hi_only = r1_hi_only;
  return launch_gemm(dtype, p, (hipStream_t)stream) == hipSuccess ? DPTX_OK : DPTX_E_HIP;
}

int dptx_op_stem_conv(int32_t dtype, const float* x, const void* Wt, void* y, int32_t B, int32_t H, int32_t W, void* stream) {
  return launch_stem_conv(dtype, x, DPTX_IO_FP32, Wt, y, B, H, W, g_op_planes, (hipStream_t)stream) == hipSuccess ? DPTX_OK : DPTX_E_HIP;
}

int dptx_op_attention(int32_t dtype, const void* qkv, void* out, int32_t B, int32_t S, int32_t heads, void* stream) {
  return launch_attention(dtype, qkv, out, B, S, heads, g_op_planes, (hipStream_t)stream) == hipSuccess ? DPTX_OK : DPTX_E_HIP;
}

int dptx_op_layernorm(int32_t dtype, const float* x, const float* gamma, const float* beta, void* y, int32_t M, int32_t C,
                      float eps, void* stream) {
  return launch_layernorm(dtype, x, gamma, beta, y, M, C, eps, g_op_planes, (hipStream_t)stream) == hipSuccess ? DPTX_OK : DPTX_E_HIP;
}

int dptx_op_groupnorm(int32_t dtype, const void* X, const float* gamma, const float* beta, const void* R, void* Y, int32_t B,
                      int32_t HW, int32_t C, int32_t relu, float eps, void* scratch_f32, void* stream) {
  if (scratch_f32 == nullptr) return DPTX_E_INVALID;
  hipError_t r = launch_gn_stats(dtype, X, (float*)scratch_f32, B, HW, C, g_op_planes, (hipStream_t)stream);
  if (r != hipSuccess) return DPTX_E_HIP;
  GnParams g{};
  g.X = X; g.Y = Y; g.gamma = gamma; g.beta = beta; g.partial = (float*)scratch_f32; g.R = R;
  g.B = B; g.HW = HW; g.C = C; g.relu = relu; g.eps = eps;
  return launch_gn_apply(dtype, g, g_op_planes, (hipStream_t)stream) == hipSuccess ? DPTX_OK : DPTX_E_HIP;
}

// ---- arena debugging (tests/test_gpu_poison.py): a forward must not read an arena byte it did not write itself
int dptx_debug_arena_fill(dptx_handle h, int32_t byte_value) {
  if (!h) return DPTX_E_INVALID;
  if (h->cfg.device_id < 0) return h->fail(DPTX_E_NODEVICE, "host-only handle");
  if (!h->d_arena) return h->fail(DPTX_E_INVALID, "no arena yet (weights not finalized / imported)");
  DeviceGuard guard(h->cfg.device_id);
  HIPCHK(h, guard.err);
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemset(h->d_arena, byte_value & 0xff, h->arena_bytes));
  HIPCHK(h, hipDeviceSynchronize());
  return DPTX_OK;
}

int dptx_debug_arena_read(dptx_handle h, void* dst_host, size_t offset, size_t bytes) {
  if (!h || !dst_host) return DPTX_E_INVALID;
  if (h->cfg.device_id < 0) return h->fail(DPTX_E_NODEVICE, "host-only handle");
  if (!h->d_arena || offset > h->arena_bytes || bytes > h->arena_bytes - offset) return h->fail(DPTX_E_INVALID, "arena range");
  DeviceGuard guard(h->cfg.device_id);
  HIPCHK(h, guard.err);
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemcpy(dst_host, h->d_arena + offset, bytes, hipMemcpyDeviceToHost));
  return DPTX_OK;
}

static std::vector<std::pair<const char*, const Buf*>> arena_buf_list(dptx_handle h) {
  return {{"sraw", &h->sraw}, {"stem", &h->stem}, {"S0", &h->S[0]}, {"S1", &h->S[1]}, {"S2", &h->S[2]}, {"T1", &h->T1}, {"T2", &h->T2},
          {"PA", &h->PA}, {"PB", &h->PB}, {"DS", &h->DS}, {"part0", &h->part[0]}, {"part1", &h->part[1]}, {"part2", &h->part[2]},
          {"part3", &h->part[3]}, {"X", &h->X}, {"lnst", &h->lnst}, {"Hn", &h->Hn}, {"QKV", &h->QKV}, {"AO", &h->AO}, {"F1", &h->F1},
          {"R3", &h->R3}, {"R4", &h->R4}, {"L3", &h->L3}, {"T4", &h->T4}, {"L4", &h->L4}, {"clsb", &h->clsb}, {"pos_alt", &h->pos_alt},
          {"lrn0", &h->lrn[0]}, {"lrn1", &h->lrn[1]}, {"lrn2", &h->lrn[2]}, {"lrn3", &h->lrn[3]}, {"tA", &h->tA}, {"tB", &h->tB},
          {"tC", &h->tC}, {"P0", &h->P[0]}, {"P1", &h->P[1]}, {"P2", &h->P[2]}, {"P3", &h->P[3]}, {"H0", &h->H0}, {"H0U", &h->H0U},
          {"H1", &h->H1}};
}

int dptx_debug_arena_layout(dptx_handle h, char* dst, size_t capacity) {
  if (!h) return DPTX_E_INVALID;
  std::string s;
  char line[256];
  snprintf(line, sizeof line, "arena_bytes %zu\narena_single %zu\nhalf_region %zu\nhalf_batch %d\nmax_batch %d\nn_streams %d\nplanes %d\n",
           h->arena_bytes, h->arena_single, h->half_region, h->half_batch, h->cfg.max_batch, h->n_streams, h->two_planes() ? 2 : 1);
  s += line;
  for (const auto& nb : arena_buf_list(h)) {
    snprintf(line, sizeof line, "buf %s %zu %zu %zu\n", nb.first, nb.second->off, nb.second->bytes, nb.second->off2);
    s += line;
  }
  if (dst && capacity > 0) {
    const size_t n = s.size() < capacity - 1 ? s.size() : capacity - 1;
    memcpy(dst, s.data(), n);
    dst[n] = 0;
  }
  return (int)s.size() + 1;
}

// One 64-bit word sum (order-independent) per arena buffer, sub-batch region and plane of the layout the LAST forward used, on
// `stream` (i.e. behind that forward): out_dev[(plane * regions + region) * nbuf + buf], buffers in dptx_debug_arena_layout
// order.  Returns the number of sums (or the capacity needed when out_dev is null).  Comparing the vectors of two forwards of
// the same input names the first tensor that differs (tools/gpu/r4_hunt.py).
int dptx_debug_arena_checksums(dptx_handle h, void* out_dev, int32_t capacity, void* stream) {
  if (!h) return DPTX_E_INVALID;
  const auto bufs = arena_buf_list(h);
  const int nbuf = (int)bufs.size();
  const int regions = h->last_regions > 1 ? h->last_regions : 1;
  const int planes = h->two_planes() ? 2 : 1;
  const int total = planes * regions * nbuf;
  if (!out_dev) return total;
  if (capacity < total) return h->fail(DPTX_E_INVALID, "checksum buffer too small");
  if (h->cfg.device_id < 0 || !h->d_arena) return h->fail(DPTX_E_INVALID, "no arena");
  DeviceGuard guard(h->cfg.device_id);
  HIPCHK(h, guard.err);
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(h, hipMemsetAsync(out_dev, 0, (size_t)total * 8, st));
  // extent of a buffer in the plan in use: up to the next buffer's offset
  std::vector<std::pair<size_t, int>> order;
  for (int i = 0; i < nbuf; ++i) order.push_back({regions > 1 ? bufs[i].second->off2 : bufs[i].second->off, i});
  std::sort(order.begin(), order.end());
  const size_t plan_end = regions > 1 ? h->half_region : h->arena_single;
  for (int pl = 0; pl < planes; ++pl)
    for (int r = 0; r < regions; ++r)
      for (int k = 0; k < nbuf; ++k) {
        const size_t off = order[k].first, end = k + 1 < nbuf ? order[k + 1].first : plan_end;
        if (end <= off) continue;
        const char* base = h->d_arena + (size_t)pl * h->arena_single + (regions > 1 ? (size_t)r * h->half_region : 0) + off;
        HIPCHK(h, launch_checksum(base, end - off, (unsigned long long*)out_dev + ((size_t)pl * regions + r) * nbuf + order[k].second, st));
      }
  return total;
}

// debug: word sums of {lnst, Hn, QKV, AO, F1} after every launch of the ViT blocks of the following single-stream forwards go to
// dev_buf[launch * 5 + buffer] (uint64; launch 0 = after the cls rows, then qkv / attention / proj / fc1 / fc2 per block);
// NULL switches it off.  The first entry that differs between two forwards of one input names the launch whose output differs.
int dptx_debug_set_launch_sums(dptx_handle h, void* dev_buf, int32_t capacity) {
  if (!h) return DPTX_E_INVALID;
  h->d_lsums = (unsigned long long*)dev_buf;
  h->lsum_cap = dev_buf ? capacity : 0;
  h->lsum_n = 0;
  return DPTX_OK;
}

int dptx_debug_set_trace(void* dev_buf) {
  gemm_set_trace((long long*)dev_buf);
  return DPTX_OK;
}

int dptx_debug_set_gemm_flags(int32_t flags) {
  gemm_set_debug_flags(flags);
  return DPTX_OK;
}

int dptx_op_conv_fp8(const void* X8, const void* Wt8, const float* bias, const void* R, void* Y, void* Y8, int32_t B, int32_t H,
                     int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t stride, int32_t pad_t, int32_t pad_l, int32_t Ho,
                     int32_t Wo, int32_t act, int32_t q_relu, float out_scale, void* stream) {
  GemmParams p{};
  p.A = X8; p.W = Wt8; p.C = Y; p.bias = bias; p.R1 = R; p.C8 = Y8; p.q_relu = q_relu; p.out_scale = out_scale;
  p.M = B * Ho * Wo; p.N = Cout; p.K = ksize * ksize * Cin; p.ldw = p.K;
  p.a_rpi = Ho * Wo; p.Wout = Wo; p.Hin = H; p.Win = W; p.Cin = Cin; p.a_pix_stride = Cin;
  p.a_img_stride = (long long)H * W * Cin;
  p.a_bytes = (long long)B * H * W * Cin;
  p.ksz = ksize; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
  p.c_rpi = 0x7fffffff; p.ldc = Cout; p.act = act;
  p.k_tap_fast = (ksize == 3 && Cin >= 512) ? 1 : 0;
  return launch_gemm(MODE_FP8, p, (hipStream_t)stream) == hipSuccess ? DPTX_OK : DPTX_E_HIP;
}

int dptx_op_conv_groupnorm(int32_t dtype, const void* X, const void* Wt, void* Yraw, const float* gamma, const float* beta,
                           const void* R, void* Y, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize,
                           int32_t stride, int32_t pad_t, int32_t pad_l, int32_t Ho, int32_t Wo, int32_t relu, float eps,
                           void* scratch_f32, void* stream) {
  if (scratch_f32 == nullptr || (Ho * Wo) % 32 != 0) return DPTX_E_INVALID;
  GemmParams p{};
  p.A = X; p.W = Wt; p.C = Yraw;
  p.M = B * Ho * Wo; p.N = Cout; p.K = ksize * ksize * Cin; p.ldw = p.K;
  p.a_rpi = Ho * Wo; p.Wout = Wo; p.Hin = H; p.Win = W; p.Cin = Cin; p.a_pix_stride = Cin;
  p.a_img_stride = (long long)H * W * Cin;
  p.a_bytes = (long long)B * H * W * Cin * 2;
  p.ksz = ksize; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
  p.c_rpi = 0x7fffffff; p.ldc = Cout; p.planes = g_op_planes;
  p.k_tap_fast = (ksize == 3 && Cin >= 512) ? 1 : 0;
  p.gn_part = (float*)scratch_f32; p.gn_hw = Ho * Wo; p.gn_blocks = Ho * Wo / 32; p.gn_cpg = Cout / 32;
  if (launch_gemm(dtype, p, (hipStream_t)stream) != hipSuccess) return DPTX_E_HIP;
  GnParams g{};
  g.X = Yraw; g.Y = Y; g.gamma = gamma; g.beta = beta; g.partial = (float*)scratch_f32; g.R = R;
  g.B = B; g.HW = Ho * Wo; g.C = Cout; g.relu = relu; g.eps = eps; g.nrec = Ho * Wo / 32;
  return launch_gn_apply(dtype, g, g_op_planes, (hipStream_t)stream) == hipSuccess ? DPTX_OK : DPTX_E_HIP;
}

int dptx_op_upsample2x(int32_t dtype, const void* X, void* Y, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
  return launch_upsample2x(dtype, X, Y, B, H, W, C, g_op_planes, (hipStream_t)stream) == hipSuccess ? DPTX_OK : DPTX_E_HIP;
}

}  // extern "C"
