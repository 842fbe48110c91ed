// torch_ops.cpp — TORCH_LIBRARY(mmamd, ...) over the C-ABI of libmmamd.so ("one shared object ... registered via
// TORCH_LIBRARY: gemm_bf16, layernorm, attn_fwd, patch_embed, embed_tokens, pool_proj_normalize, contrastive_fwd ...").
//
// (ROCm builds of torch present HIP devices as DeviceType::CUDA: guards and streams are the *MasqueradingAsCUDA forms.)
// Host C++ only (no kernels): every op checks its tensors, allocates the outputs with ATen, and calls the extern "C" entry point of
// include/mmamd.h with raw device pointers and the CURRENT HIP stream of the tensors' device.  Registered for the dispatch keys
//   CUDA (= HIP on ROCm builds of torch): the kernels;   Meta: shape / dtype inference, so FakeTensor tracing (torch.compile) works.
// What this buys over the ctypes binding (multimodal_amd/_lib.py, kept as the no-torch binding of INTEGRATION.md): the ops are real
// dispatcher ops — torch.jit.script(CLIPTextEncoder(...)), torch.jit.script(MultiHeadSelfAttention(...)) and torch.compile(model)
// see `torch.ops.mmamd.*` calls instead of opaque Python (reference tests: tests/models/clip/test_text_encoder.py:162-174,
// tests/modules/layers/test_multi_head_attention.py:50-57).
//
// Parameters are passed as they are kept by the nn.Module (fp32 or bf16 nn.Parameters); `packed()` below keeps one kernel-ready
// copy per (parameter, dtype) — bf16 for GEMM weights, fp32 for vectors — keyed on the parameter's TensorImpl, storage pointer and
// version counter, so optimizer steps / load_state_dict / .to() refresh it (the C++ twin of multimodal_amd/_packing.PackedCache).
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <ATen/core/dispatch/Dispatcher.h>
#include <torch/csrc/autograd/autograd_not_implemented_fallback.h>
#include <torch/library.h>

#include <cmath>
#include <mutex>
#include <tuple>
#include <unordered_map>

#include "../../include/mmamd.h"

namespace {

using at::Tensor;
using c10::optional;

void check_status(int rc, const char* what) {
  TORCH_CHECK(rc == 0, what, " failed (status ", rc, "): ", mmamd_last_error());
}

// evaluated as the last argument of every C-ABI call: also drops a stale per-thread HIP status (mmamd_clear_last_hip_error)
mmamd_stream_t cur_stream(const Tensor& t) {
  (void)mmamd_clear_last_hip_error();
  return (mmamd_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
}

int dt_code(at::ScalarType t) {
  if (t == at::kFloat) return MMAMD_F32;
  if (t == at::kBFloat16) return MMAMD_BF16;
  TORCH_CHECK(false, "mmamd: unsupported dtype ", t, " (float32 / bfloat16 only)");
}
at::ScalarType code_dt(int64_t code) {
  TORCH_CHECK(code == MMAMD_F32 || code == MMAMD_BF16, "mmamd: bad dtype code ", code);
  return code == MMAMD_F32 ? at::kFloat : at::kBFloat16;
}

void chk(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), "mmamd: ", name, " is on ", t.device(), ": the MI355X path needs HIP device tensors (no CPU fallback)");
  TORCH_CHECK(t.is_contiguous(), "mmamd: ", name, " must be contiguous");
}
void chk(const Tensor& t, const char* name, at::ScalarType dt) {
  chk(t, name);
  TORCH_CHECK(t.scalar_type() == dt, "mmamd: ", name, " must be ", dt, ", got ", t.scalar_type());
}

// ---- kernel-ready parameter copies ---------------------------------------------------------------------------------------------
struct PackedEntry {
  c10::weak_intrusive_ptr<c10::TensorImpl> src;
  const void* data;
  int64_t version;
  Tensor packed;
};
std::mutex g_pack_mu;
std::unordered_map<uint64_t, PackedEntry> g_pack;  // key: TensorImpl address ^ dtype code

Tensor convert_impl(const Tensor& x, int64_t dtype) {
  chk(x, "x");
  const at::ScalarType want = code_dt(dtype);
  if (x.scalar_type() == want) return x;
  c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
  Tensor out = at::empty(x.sizes(), x.options().dtype(want));
  check_status(mmamd_convert(x.data_ptr(), dt_code(x.scalar_type()), out.data_ptr(), (int)dtype, x.numel(), cur_stream(x)), "mmamd_convert");
  return out;
}

Tensor packed_impl(const Tensor& p, int64_t dtype) {
  const at::ScalarType want = code_dt(dtype);
  Tensor t = p.detach();
  TORCH_CHECK(t.is_cuda(), "mmamd: parameter lives on ", t.device(), ": move the module to a HIP device (.to('cuda')); there is no CPU path");
  if (t.scalar_type() == want && t.is_contiguous()) return t;
  c10::TensorImpl* impl = p.unsafeGetTensorImpl();
  const uint64_t key = (reinterpret_cast<uint64_t>(impl) << 1) ^ (uint64_t)dtype;
  {
    std::lock_guard<std::mutex> lk(g_pack_mu);
    auto it = g_pack.find(key);
    if (it != g_pack.end()) {
      auto alive = it->second.src.lock();
      if (alive && alive.get() == impl && it->second.data == t.data_ptr() && it->second.version == (int64_t)p._version()) return it->second.packed;
      g_pack.erase(it);
    }
  }
  Tensor conv = convert_impl(t.is_contiguous() ? t : t.contiguous(), dtype);
  {
    std::lock_guard<std::mutex> lk(g_pack_mu);
    if (g_pack.size() > 4096) {  // drop entries whose parameter died
      for (auto it = g_pack.begin(); it != g_pack.end();) it = it->second.src.expired() ? g_pack.erase(it) : std::next(it);
    }
    g_pack.erase(key);
    g_pack.emplace(key, PackedEntry{c10::weak_intrusive_ptr<c10::TensorImpl>(p.getIntrusivePtr()), t.data_ptr(), (int64_t)p._version(), conv});
  }
  return conv;
}
Tensor packed_meta(const Tensor& p, int64_t dtype) { return at::empty(p.sizes(), p.options().dtype(code_dt(dtype))); }

Tensor f32v(const Tensor& p) { return packed_impl(p, MMAMD_F32); }
Tensor bf16w(const Tensor& p) { return packed_impl(p, MMAMD_BF16); }

// ---- ops ------------------------------------------------------------------------------------------------------------------------
Tensor layernorm_impl(const Tensor& x, const Tensor& gamma, const Tensor& beta, double eps, int64_t out_dtype) {
  chk(x, "x");
  const Tensor g = f32v(gamma), b = f32v(beta);
  const int64_t d = x.size(-1), rows = x.numel() / d;
  c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
  Tensor y = at::empty(x.sizes(), x.options().dtype(code_dt(out_dtype)));
  check_status(mmamd_layernorm(x.data_ptr(), dt_code(x.scalar_type()), g.data_ptr<float>(), b.data_ptr<float>(), y.data_ptr(), (int)out_dtype,
                               (int)rows, (int)d, (float)eps, cur_stream(x)), "mmamd_layernorm");
  return y;
}
Tensor layernorm_meta(const Tensor& x, const Tensor&, const Tensor&, double, int64_t out_dtype) {
  return at::empty(x.sizes(), x.options().dtype(code_dt(out_dtype)));
}

// out[M,N] = act(a[M,K] @ w[N,K]^T + bias) (+ residual).  a: bf16 activations; w / bias: the module's parameters (packed here).
Tensor gemm_impl(const Tensor& a, const Tensor& w, const optional<Tensor>& bias, const optional<Tensor>& residual, int64_t act, int64_t out_dtype) {
  chk(a, "a", at::kBFloat16);
  TORCH_CHECK(a.dim() == 2 && w.dim() == 2 && a.size(1) == w.size(1), "mmamd::gemm_bf16: a [M,K] and w [N,K] expected");
  const Tensor wp = bf16w(w);
  Tensor bp;
  if (bias.has_value()) bp = f32v(*bias);
  const int64_t M = a.size(0), K = a.size(1), N = w.size(0);
  const at::ScalarType odt = code_dt(out_dtype);
  if (residual.has_value()) {
    chk(*residual, "residual", odt);
    TORCH_CHECK(residual->size(0) == M && residual->size(1) == N, "mmamd::gemm_bf16: residual shape");
  }
  c10::hip::HIPGuardMasqueradingAsCUDA guard(a.device());
  Tensor out = at::empty({M, N}, a.options().dtype(odt));
  check_status(mmamd_gemm_bf16(a.data_ptr(), (int)K, wp.data_ptr(), (int)K, bp.defined() ? bp.data_ptr<float>() : nullptr,
                               residual.has_value() ? residual->data_ptr() : nullptr, (int)N, out.data_ptr(), (int)N, (int)out_dtype, (int)M,
                               (int)N, (int)K, (int)act, cur_stream(a)), "mmamd_gemm_bf16");
  return out;
}
Tensor gemm_meta(const Tensor& a, const Tensor& w, const optional<Tensor>&, const optional<Tensor>&, int64_t, int64_t out_dtype) {
  return at::empty({a.size(0), w.size(0)}, a.options().dtype(code_dt(out_dtype)));
}

Tensor attn_fwd_impl(const Tensor& qkv, int64_t B, int64_t S, int64_t H, bool causal) {
  chk(qkv, "qkv", at::kBFloat16);
  TORCH_CHECK(qkv.dim() == 2 && qkv.size(0) == B * S && qkv.size(1) == 3 * H * 64, "mmamd::attn_fwd: qkv must be [B*S, 3*H*64]");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(qkv.device());
  Tensor out = at::empty({B * S, H * 64}, qkv.options());
  check_status(mmamd_attention_fwd(qkv.data_ptr(), out.data_ptr(), (int)B, (int)S, (int)H, causal ? 1 : 0, 1.0f / std::sqrt(64.0f), cur_stream(qkv)),
               "mmamd_attention_fwd");
  return out;
}
Tensor attn_fwd_meta(const Tensor& qkv, int64_t B, int64_t S, int64_t H, bool) { return at::empty({B * S, H * 64}, qkv.options()); }

// conv patch embedding (no bias) + CLS + positional embedding + ln_pre -> fp32 residual stream [B*(G2+1), w]
Tensor patch_embed_impl(const Tensor& img, const Tensor& conv_w, const Tensor& cls, const Tensor& pos, const Tensor& ln_w, const Tensor& ln_b,
                        double eps, int64_t patch) {
  chk(img, "image");
  TORCH_CHECK(img.dim() == 4 && conv_w.dim() == 4, "mmamd::patch_embed: image [B,C,H,W] and conv weight [w,C,p,p] expected");
  const int64_t B = img.size(0), C = img.size(1), HW = img.size(2), g = HW / patch, G2 = g * g, w = conv_w.size(0);
  const int64_t K = C * patch * patch, kpad = (K + 63) / 64 * 64;
  c10::hip::HIPGuardMasqueradingAsCUDA guard(img.device());
  mmamd_stream_t st = cur_stream(img);
  Tensor cols = at::empty({B * G2, kpad}, img.options().dtype(at::kBFloat16));
  check_status(mmamd_patchify(img.data_ptr(), dt_code(img.scalar_type()), cols.data_ptr(), (int)B, (int)C, (int)HW, (int)patch, (int)kpad, st),
               "mmamd_patchify");
  Tensor wk = bf16w(conv_w).view({w, K});
  if (kpad != K) {  // K = 588 (patch 14): zero-padded copy; rebuilt per call here (the eager path caches it in the module)
    Tensor padded = at::zeros({w, kpad}, wk.options());
    padded.narrow(1, 0, K).copy_(wk);
    wk = padded;
  }
  Tensor pe = at::empty({B * G2, w}, img.options().dtype(at::kFloat));
  check_status(mmamd_gemm_bf16(cols.data_ptr(), (int)kpad, wk.data_ptr(), (int)kpad, nullptr, nullptr, 0, pe.data_ptr(), (int)w, MMAMD_F32,
                               (int)(B * G2), (int)w, (int)kpad, MMAMD_ACT_NONE, st), "mmamd_gemm_bf16");
  const Tensor c = f32v(cls), p = f32v(pos), gw = f32v(ln_w), gb = f32v(ln_b);
  Tensor x = at::empty({B * (G2 + 1), w}, pe.options());
  check_status(mmamd_vit_assemble_ln(pe.data_ptr(), MMAMD_F32, c.data_ptr<float>(), p.data_ptr<float>(), gw.data_ptr<float>(),
                                     gb.data_ptr<float>(), (float)eps, x.data_ptr<float>(), (int)B, (int)G2, (int)w, st), "mmamd_vit_assemble_ln");
  return x;
}
Tensor patch_embed_meta(const Tensor& img, const Tensor& conv_w, const Tensor&, const Tensor&, const Tensor&, const Tensor&, double, int64_t patch) {
  const int64_t g = img.size(2) / patch;
  return at::empty({img.size(0) * (g * g + 1), conv_w.size(0)}, img.options().dtype(at::kFloat));
}

Tensor embed_tokens_impl(const Tensor& ids, const Tensor& table, const Tensor& pos) {
  chk(ids, "ids", at::kLong);
  TORCH_CHECK(ids.dim() == 2 && table.dim() == 2, "mmamd::embed_tokens: ids [B,S] and table [vocab,d] expected");
  Tensor tb = table.detach();
  TORCH_CHECK(tb.is_cuda() && (tb.scalar_type() == at::kFloat || tb.scalar_type() == at::kBFloat16), "mmamd::embed_tokens: fp32 / bf16 HIP table");
  if (!tb.is_contiguous()) tb = tb.contiguous();
  const Tensor p = f32v(pos);
  const int64_t B = ids.size(0), S = ids.size(1), vocab = tb.size(0), d = tb.size(1);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(ids.device());
  Tensor x = at::empty({B * S, d}, tb.options().dtype(at::kFloat));
  check_status(mmamd_embed_tokens(ids.data_ptr<int64_t>(), tb.data_ptr(), dt_code(tb.scalar_type()), p.data_ptr<float>(), x.data_ptr<float>(),
                                  (int)B, (int)S, (int)d, (int)vocab, cur_stream(ids)), "mmamd_embed_tokens");
  return x;
}
Tensor embed_tokens_meta(const Tensor& ids, const Tensor& table, const Tensor&) {
  return at::empty({ids.size(0) * ids.size(1), table.size(1)}, table.options().dtype(at::kFloat));
}

// pooled row (argmax of ids, or row 0) -> LayerNorm -> projection (-> L2 normalise); proj [d,E] (x @ proj) or a Linear weight [E,d]
Tensor pool_proj_normalize_impl(const Tensor& h, int64_t B, int64_t S, const optional<Tensor>& ids, const Tensor& ln_w, const Tensor& ln_b,
                                double eps, const Tensor& proj, bool proj_is_linear_weight, bool normalize) {
  chk(h, "h", at::kFloat);
  const Tensor gw = f32v(ln_w), gb = f32v(ln_b), P = f32v(proj);
  const int64_t d = h.size(-1);
  const int64_t E = proj_is_linear_weight ? P.size(0) : P.size(1);
  const int sk = proj_is_linear_weight ? 1 : (int)P.size(1), se = proj_is_linear_weight ? (int)d : 1;
  if (ids.has_value()) chk(*ids, "ids", at::kLong);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(h.device());
  Tensor out = at::empty({B, E}, h.options()), ws = at::empty({B, d}, h.options());
  check_status(mmamd_pool_ln_proj(h.data_ptr<float>(), (int)S, (int)d, ids.has_value() ? ids->data_ptr<int64_t>() : nullptr, gw.data_ptr<float>(),
                                  gb.data_ptr<float>(), (float)eps, P.data_ptr<float>(), sk, se, out.data_ptr<float>(), (int)B, (int)E,
                                  normalize ? 1 : 0, ws.data_ptr<float>(), cur_stream(h)), "mmamd_pool_ln_proj");
  return out;
}
Tensor pool_proj_normalize_meta(const Tensor& h, int64_t B, int64_t, const optional<Tensor>&, const Tensor&, const Tensor&, double, const Tensor& proj,
                                bool proj_is_linear_weight, bool) {
  return at::empty({B, proj_is_linear_weight ? proj.size(0) : proj.size(1)}, h.options());
}

Tensor l2_normalize_impl(const Tensor& x, double eps) {
  chk(x, "x");
  TORCH_CHECK(x.dim() == 2, "mmamd::l2_normalize expects [rows, d]");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
  Tensor y = at::empty_like(x);
  check_status(mmamd_l2_normalize(x.data_ptr(), dt_code(x.scalar_type()), y.data_ptr(), dt_code(x.scalar_type()), (int)x.size(0), (int)x.size(1),
                                  (float)eps, cur_stream(x)), "mmamd_l2_normalize");
  return y;
}
Tensor same_meta(const Tensor& x, double) { return at::empty_like(x); }

void clamp_scalar_impl(Tensor p, optional<double> lo, optional<double> hi) {
  chk(p, "scalar", at::kFloat);
  TORCH_CHECK(p.numel() == 1, "mmamd::clamp_scalar_ expects a 1-element tensor");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(p.device());
  check_status(mmamd_clamp_scalar(p.data_ptr<float>(), lo.has_value(), (float)lo.value_or(0.0), hi.has_value(), (float)hi.value_or(0.0), cur_stream(p)),
               "mmamd_clamp_scalar");
}
void clamp_scalar_meta(Tensor, optional<double>, optional<double>) {}

Tensor activation_impl(const Tensor& x, int64_t act) {
  chk(x, "x");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
  Tensor y = at::empty_like(x);
  check_status(mmamd_activation(x.data_ptr(), nullptr, y.data_ptr(), dt_code(x.scalar_type()), x.numel(), (int)act, cur_stream(x)), "mmamd_activation");
  return y;
}
Tensor activation_meta(const Tensor& x, int64_t) { return at::empty_like(x); }

// (out3 = [loss, loss_a, loss_b], logits_a [B,WB], logits_b [B,WB]); a_all / b_all may be column slices of one packed [WB, 2E] buffer
std::tuple<Tensor, Tensor, Tensor> contrastive_fwd_impl(const Tensor& a, const Tensor& b, const Tensor& a_all, const Tensor& b_all,
                                                        const Tensor& logit_scale, int64_t label_offset, const optional<Tensor>& mask,
                                                        double label_smoothing, int64_t reduction) {
  chk(a, "a", at::kFloat);
  chk(b, "b", at::kFloat);
  TORCH_CHECK(a_all.is_cuda() && b_all.is_cuda() && a_all.scalar_type() == at::kFloat && b_all.scalar_type() == at::kFloat &&
                  a_all.stride(-1) == 1 && b_all.stride(-1) == 1 && a_all.stride(0) == b_all.stride(0),
              "mmamd::contrastive_fwd: gathered features must be fp32 HIP tensors with unit inner stride and a common row stride");
  Tensor ls = logit_scale.detach();
  chk(ls, "logit_scale", at::kFloat);
  const int64_t B = a.size(0), E = a.size(1), WB = a_all.size(0);
  Tensor m8;
  if (mask.has_value()) {
    TORCH_CHECK(mask->scalar_type() == at::kBool && mask->numel() == B, "mmamd::contrastive_fwd: mask must be a boolean tensor of shape (batch,)");
    m8 = mask->contiguous().view(at::kByte);
  }
  c10::hip::HIPGuardMasqueradingAsCUDA guard(a.device());
  Tensor la = at::empty({B, WB}, a.options()), lb = at::empty({B, WB}, a.options()), out3 = at::empty({3}, a.options()), ws = at::empty({2 * B}, a.options());
  check_status(mmamd_contrastive_fwd(a.data_ptr<float>(), b.data_ptr<float>(), a_all.data_ptr<float>(), b_all.data_ptr<float>(), (int)a_all.stride(0),
                                     ls.data_ptr<float>(), (int)B, (int)WB, (int)E, (int)label_offset, m8.defined() ? m8.data_ptr<uint8_t>() : nullptr,
                                     (float)label_smoothing, (int)reduction, la.data_ptr<float>(), lb.data_ptr<float>(), out3.data_ptr<float>(),
                                     ws.data_ptr<float>(), cur_stream(a)), "mmamd_contrastive_fwd");
  return std::make_tuple(out3, la, lb);
}
std::tuple<Tensor, Tensor, Tensor> contrastive_fwd_meta(const Tensor& a, const Tensor&, const Tensor& a_all, const Tensor&, const Tensor&, int64_t,
                                                        const optional<Tensor>&, double, int64_t) {
  return std::make_tuple(at::empty({3}, a.options()), at::empty({a.size(0), a_all.size(0)}, a.options()), at::empty({a.size(0), a_all.size(0)}, a.options()));
}

// ---- CoCa / encoder-decoder building blocks (modules/layers/{patch_embedding,multi_head_attention,attention_pooler}.py,
//      models/coca/text_decoder.py of the reference) ------------------------------------------------------------------------------

// Conv2d(kernel = stride = patch, with bias) + optional CLS row + position embeddings -> fp32 [B*(G2 (+1)), d]
Tensor image_embed_impl(const Tensor& img, const Tensor& conv_w, const Tensor& conv_b, const optional<Tensor>& cls, const Tensor& pos, int64_t patch) {
  chk(img, "image");
  TORCH_CHECK(img.dim() == 4 && conv_w.dim() == 4 && img.size(2) == img.size(3), "mmamd::image_embed: square image [B,C,H,H] and conv weight [d,C,p,p] expected");
  const int64_t B = img.size(0), C = img.size(1), HW = img.size(2), g = HW / patch, G2 = g * g, d = conv_w.size(0);
  const int64_t K = C * patch * patch, kpad = (K + 63) / 64 * 64;
  c10::hip::HIPGuardMasqueradingAsCUDA guard(img.device());
  mmamd_stream_t st = cur_stream(img);
  Tensor cols = at::empty({B * G2, kpad}, img.options().dtype(at::kBFloat16));
  check_status(mmamd_patchify(img.data_ptr(), dt_code(img.scalar_type()), cols.data_ptr(), (int)B, (int)C, (int)HW, (int)patch, (int)kpad, st),
               "mmamd_patchify");
  Tensor wk = bf16w(conv_w).view({d, K});
  if (kpad != K) {
    Tensor padded = at::zeros({d, kpad}, wk.options());
    padded.narrow(1, 0, K).copy_(wk);
    wk = padded;
  }
  const Tensor bias = f32v(conv_b), p = f32v(pos);
  Tensor pe = at::empty({B * G2, d}, img.options().dtype(at::kFloat));
  check_status(mmamd_gemm_bf16(cols.data_ptr(), (int)kpad, wk.data_ptr(), (int)kpad, bias.data_ptr<float>(), nullptr, 0, pe.data_ptr(), (int)d,
                               MMAMD_F32, (int)(B * G2), (int)d, (int)kpad, MMAMD_ACT_NONE, st), "mmamd_gemm_bf16");
  Tensor c;
  if (cls.has_value()) c = f32v(*cls);
  Tensor x = at::empty({B * (G2 + (c.defined() ? 1 : 0)), d}, pe.options());
  check_status(mmamd_flava_image_embed(pe.data_ptr<float>(), c.defined() ? c.data_ptr<float>() : nullptr, p.data_ptr<float>(), nullptr, nullptr,
                                       x.data_ptr<float>(), (int)B, (int)G2, (int)d, st), "mmamd_flava_image_embed");
  return x;
}
Tensor image_embed_meta(const Tensor& img, const Tensor& conv_w, const Tensor&, const optional<Tensor>& cls, const Tensor&, int64_t patch) {
  const int64_t g = img.size(2) / patch;
  return at::empty({img.size(0) * (g * g + (cls.has_value() ? 1 : 0)), conv_w.size(0)}, img.options().dtype(at::kFloat));
}

// general attention: q [B*Sq (or Sq when shared_q), H*hd], k / v [B*Sk, H*hd] bf16, possibly column slices of wider matrices;
// key_mask uint8 [B,Sk], full_mask uint8 [B or 1, Sq, Sk] (0 = masked) -> bf16 [B*Sq, H*hd]
Tensor attn_x_impl(const Tensor& q, const Tensor& k, const Tensor& v, int64_t B, int64_t Sq, int64_t Sk, int64_t H, int64_t hd, bool causal,
                   const optional<Tensor>& key_mask, const optional<Tensor>& full_mask, bool shared_q) {
  for (const Tensor* t : {&q, &k, &v})
    TORCH_CHECK(t->is_cuda() && t->scalar_type() == at::kBFloat16 && t->dim() == 2 && t->stride(1) == 1 && t->size(1) == H * hd,
                "mmamd::attn_x: q / k / v must be bf16 HIP matrices (views) with unit inner stride and H*head_dim columns");
  TORCH_CHECK(q.size(0) == (shared_q ? Sq : B * Sq) && k.size(0) == B * Sk && v.size(0) == B * Sk, "mmamd::attn_x: row counts do not match B, Sq, Sk");
  const uint8_t *km = nullptr, *fm = nullptr;
  int64_t fm_bs = 0;
  if (key_mask.has_value()) {
    chk(*key_mask, "key_mask", at::kByte);
    TORCH_CHECK(key_mask->numel() == B * Sk, "mmamd::attn_x: key_mask must be [B, Sk]");
    km = key_mask->data_ptr<uint8_t>();
  }
  if (full_mask.has_value()) {
    chk(*full_mask, "full mask", at::kByte);
    TORCH_CHECK(full_mask->numel() == Sq * Sk || full_mask->numel() == B * Sq * Sk, "mmamd::attn_x: full mask must be [B or 1, Sq, Sk]");
    fm = full_mask->data_ptr<uint8_t>();
    fm_bs = (full_mask->numel() == B * Sq * Sk && B > 1) ? Sq * Sk : 0;
  }
  c10::hip::HIPGuardMasqueradingAsCUDA guard(q.device());
  Tensor out = at::empty({B * Sq, H * hd}, q.options());
  check_status(mmamd_attention_x_fwd(q.data_ptr(), (int)q.stride(0), shared_q ? 0 : Sq * q.stride(0), k.data_ptr(), v.data_ptr(), (int)k.stride(0),
                                     (int)v.stride(0), Sk * k.stride(0), km, fm, fm_bs, causal ? 1 : 0, out.data_ptr(), (int)(H * hd), nullptr, MMAMD_F32,
                                     nullptr, (int)B, (int)Sq, (int)Sk, (int)H, (int)hd, 1.0f / std::sqrt((float)hd), cur_stream(q)),
               "mmamd_attention_x_fwd");
  return out;
}
Tensor attn_x_meta(const Tensor& q, const Tensor&, const Tensor&, int64_t B, int64_t Sq, int64_t, int64_t H, int64_t hd, bool, const optional<Tensor>&,
                   const optional<Tensor>&, bool) {
  return at::empty({B * Sq, H * hd}, q.options());
}

// CoCaTextEmbeddings: table[ids] + pos (+ the CLS row cls + pos[S]) -> fp32 [B*(S (+1)), d]
Tensor coca_text_embed_impl(const Tensor& ids, const Tensor& table, const Tensor& pos, const optional<Tensor>& cls) {
  chk(ids, "ids", at::kLong);
  TORCH_CHECK(ids.dim() == 2 && table.dim() == 2, "mmamd::coca_text_embed: ids [B,S] and table [vocab,d] expected");
  const Tensor tb = f32v(table), p = f32v(pos);
  Tensor c;
  if (cls.has_value()) c = f32v(*cls);
  const int64_t B = ids.size(0), S = ids.size(1), d = tb.size(1);
  TORCH_CHECK(p.numel() >= (S + (c.defined() ? 1 : 0)) * d, "mmamd::coca_text_embed: fewer position rows than tokens");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(ids.device());
  Tensor x = at::empty({B * (S + (c.defined() ? 1 : 0)), d}, tb.options());
  check_status(mmamd_coca_text_embed(ids.data_ptr<int64_t>(), tb.data_ptr<float>(), p.data_ptr<float>(), c.defined() ? c.data_ptr<float>() : nullptr,
                                     x.data_ptr<float>(), (int)B, (int)S, (int)d, (int)tb.size(0), cur_stream(ids)), "mmamd_coca_text_embed");
  return x;
}
Tensor coca_text_embed_meta(const Tensor& ids, const Tensor& table, const Tensor&, const optional<Tensor>& cls) {
  return at::empty({ids.size(0) * (ids.size(1) + (cls.has_value() ? 1 : 0)), table.size(1)}, table.options().dtype(at::kFloat));
}

// CoCaTextDecoder.build_mask as uint8 [B, S+1, S+1]: src = int64 token ids (use_pad_id: keep = id != pad_id) or a padding mask (keep = value != 0)
Tensor coca_text_mask_impl(const Tensor& src, bool use_pad_id, int64_t pad_id) {
  chk(src, "mask source");
  TORCH_CHECK(src.dim() == 2, "mmamd::coca_text_mask: [B, S] ids or padding mask expected");
  int kind = 0;
  if (use_pad_id) {
    TORCH_CHECK(src.scalar_type() == at::kLong, "mmamd::coca_text_mask: token ids must be int64");
  } else {
    const auto t = src.scalar_type();
    kind = t == at::kFloat ? 1 : t == at::kLong ? 2 : (t == at::kByte || t == at::kBool) ? 3 : -1;
    TORCH_CHECK(kind > 0, "mmamd::coca_text_mask: unsupported mask dtype ", t);
  }
  const int64_t B = src.size(0), S = src.size(1);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(src.device());
  Tensor out = at::empty({B, S + 1, S + 1}, src.options().dtype(at::kByte));
  check_status(mmamd_coca_text_mask(src.data_ptr(), kind, use_pad_id ? pad_id : 0, out.data_ptr<uint8_t>(), (int)B, (int)S, cur_stream(src)),
               "mmamd_coca_text_mask");
  return out;
}
Tensor coca_text_mask_meta(const Tensor& src, bool, int64_t) {
  return at::empty({src.size(0), src.size(1) + 1, src.size(1) + 1}, src.options().dtype(at::kByte));
}

// exact-fp32 rows @ W^T + bias (pooled projections: one row per sample)
Tensor rows_linear_f32_impl(const Tensor& x, const Tensor& weight, const optional<Tensor>& bias) {
  chk(x, "x", at::kFloat);
  TORCH_CHECK(x.dim() == 2 && weight.dim() == 2 && x.size(1) == weight.size(1), "mmamd::rows_linear_f32: x [B,d] and weight [E,d] expected");
  const Tensor w = f32v(weight);
  Tensor b;
  if (bias.has_value()) b = f32v(*bias);
  const int64_t B = x.size(0), d = x.size(1), E = w.size(0);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
  Tensor out = at::empty({B, E}, x.options());
  check_status(mmamd_rows_linear_f32(x.data_ptr<float>(), d, w.data_ptr<float>(), b.defined() ? b.data_ptr<float>() : nullptr, 0, out.data_ptr<float>(),
                                     (int)B, (int)d, (int)E, cur_stream(x)), "mmamd_rows_linear_f32");
  return out;
}
Tensor rows_linear_f32_meta(const Tensor& x, const Tensor& weight, const optional<Tensor>&) { return at::empty({x.size(0), weight.size(0)}, x.options()); }

// FLAVA's attention (modules/layers/attention.py:185-241): output + the normalised probabilities [B,H,S,S], key-padding mask honoured
std::tuple<Tensor, Tensor> attn_probs_impl(const Tensor& qkv, int64_t B, int64_t S, int64_t H, const optional<Tensor>& key_mask, bool write_probs,
                                           int64_t probs_dtype) {
  chk(qkv, "qkv", at::kBFloat16);
  TORCH_CHECK(qkv.dim() == 2 && qkv.size(0) == B * S && qkv.size(1) == 3 * H * 64, "mmamd::attn_probs: qkv must be [B*S, 3*H*64]");
  TORCH_CHECK(probs_dtype == MMAMD_F32 || probs_dtype == MMAMD_BF16, "mmamd::attn_probs: probs_dtype must be 0 (fp32) or 1 (bf16)");
  const uint8_t* km = nullptr;
  if (key_mask.has_value()) {
    chk(*key_mask, "key_mask", at::kByte);
    TORCH_CHECK(key_mask->dim() == 2 && key_mask->size(0) == B && key_mask->size(1) == S, "mmamd::attn_probs: key_mask must be uint8 [B, S]");
    km = key_mask->data_ptr<uint8_t>();
  }
  c10::hip::HIPGuardMasqueradingAsCUDA guard(qkv.device());
  Tensor out = at::empty({B * S, H * 64}, qkv.options());
  Tensor probs = write_probs ? at::empty({B, H, S, S}, qkv.options().dtype(probs_dtype == MMAMD_F32 ? at::kFloat : at::kBFloat16))
                             : at::empty({0}, qkv.options().dtype(at::kFloat));
  check_status(mmamd_attention_probs_fwd(qkv.data_ptr(), km, out.data_ptr(), write_probs ? probs.data_ptr() : nullptr, (int)probs_dtype, (int)B, (int)S,
                                         (int)H, 1.0f / std::sqrt(64.0f), cur_stream(qkv)), "mmamd_attention_probs_fwd");
  return std::make_tuple(out, probs);
}
std::tuple<Tensor, Tensor> attn_probs_meta(const Tensor& qkv, int64_t B, int64_t S, int64_t H, const optional<Tensor>&, bool write_probs,
                                           int64_t probs_dtype) {
  return std::make_tuple(at::empty({B * S, H * 64}, qkv.options()),
                         write_probs ? at::empty({B, H, S, S}, qkv.options().dtype(probs_dtype == MMAMD_F32 ? at::kFloat : at::kBFloat16))
                                     : at::empty({0}, qkv.options().dtype(at::kFloat)));
}

// The loss's ONE collective as a dispatcher op (modules/losses/contrastive_loss_with_temperature.py:26-47 does two list all-gathers + two
// concats): the packed [B, 2E] block of a rank -> [W*B, 2E].  Backend-agnostic (it only calls c10d's functional collective through the
// dispatcher: RCCL for HIP tensors, gloo for CPU tensors), traceable; group_size <= 1 returns a copy.
Tensor allgather_packed_impl(const Tensor& buf, std::string group_name, int64_t group_size) {
  TORCH_CHECK(buf.dim() == 2, "mmamd::allgather_packed: [B, 2E] block expected");
  if (group_size <= 1) return buf.clone();
  static auto gather = c10::Dispatcher::singleton()
                           .findSchemaOrThrow("_c10d_functional::all_gather_into_tensor", "")
                           .typed<Tensor(const Tensor&, int64_t, std::string)>();
  static auto wait = c10::Dispatcher::singleton().findSchemaOrThrow("_c10d_functional::wait_tensor", "").typed<Tensor(const Tensor&)>();
  return wait.call(gather.call(buf.contiguous(), group_size, std::move(group_name)));
}

int64_t abi_version_impl() { return mmamd_abi_version(); }

// drops every kernel-ready parameter copy this shim holds (multimodal_amd._packing.invalidate_packed() calls it: `.data` writes do not
// bump torch's version counter, so a stale copy could otherwise survive in scripted / compiled forwards)
void clear_packed_impl() {
  std::lock_guard<std::mutex> lk(g_pack_mu);
  g_pack.clear();
}

}  // namespace

TORCH_LIBRARY(mmamd, m) {
  m.def("abi_version() -> int", abi_version_impl);
  m.def("clear_packed() -> ()", clear_packed_impl);
  m.def("packed(Tensor p, int dtype) -> Tensor");
  m.def("convert(Tensor x, int dtype) -> Tensor");
  m.def("layernorm(Tensor x, Tensor gamma, Tensor beta, float eps, int out_dtype) -> Tensor");
  m.def("gemm_bf16(Tensor a, Tensor w, Tensor? bias, Tensor? residual, int act, int out_dtype) -> Tensor");
  m.def("attn_fwd(Tensor qkv, int B, int S, int H, bool causal) -> Tensor");
  m.def("attn_probs(Tensor qkv, int B, int S, int H, Tensor? key_mask, bool write_probs, int probs_dtype) -> (Tensor, Tensor)");
  m.def("allgather_packed(Tensor buf, str group_name, int group_size) -> Tensor");
  m.def("patch_embed(Tensor img, Tensor conv_w, Tensor cls, Tensor pos, Tensor ln_w, Tensor ln_b, float eps, int patch) -> Tensor");
  m.def("embed_tokens(Tensor ids, Tensor table, Tensor pos) -> Tensor");
  m.def("pool_proj_normalize(Tensor h, int B, int S, Tensor? ids, Tensor ln_w, Tensor ln_b, float eps, Tensor proj, bool proj_is_linear_weight, "
        "bool normalize) -> Tensor");
  m.def("l2_normalize(Tensor x, float eps) -> Tensor");
  m.def("clamp_scalar_(Tensor(a!) p, float? lo, float? hi) -> ()");
  m.def("activation(Tensor x, int act) -> Tensor");
  m.def("image_embed(Tensor img, Tensor conv_w, Tensor conv_b, Tensor? cls, Tensor pos, int patch) -> Tensor");
  m.def("attn_x(Tensor q, Tensor k, Tensor v, int B, int Sq, int Sk, int H, int head_dim, bool causal, Tensor? key_mask, Tensor? full_mask, "
        "bool shared_q) -> Tensor");
  m.def("coca_text_embed(Tensor ids, Tensor table, Tensor pos, Tensor? cls) -> Tensor");
  m.def("coca_text_mask(Tensor src, bool use_pad_id, int pad_id) -> Tensor");
  m.def("rows_linear_f32(Tensor x, Tensor weight, Tensor? bias) -> Tensor");
  m.def("contrastive_fwd(Tensor a, Tensor b, Tensor a_all, Tensor b_all, Tensor logit_scale, int label_offset, Tensor? mask, "
        "float label_smoothing, int reduction) -> (Tensor, Tensor, Tensor)");
}

TORCH_LIBRARY_IMPL(mmamd, CUDA, m) {  // the "CUDA" dispatch key is the HIP device on ROCm builds of torch
  m.impl("packed", packed_impl);
  m.impl("convert", convert_impl);
  m.impl("layernorm", layernorm_impl);
  m.impl("gemm_bf16", gemm_impl);
  m.impl("attn_fwd", attn_fwd_impl);
  m.impl("attn_probs", attn_probs_impl);
  m.impl("patch_embed", patch_embed_impl);
  m.impl("embed_tokens", embed_tokens_impl);
  m.impl("pool_proj_normalize", pool_proj_normalize_impl);
  m.impl("l2_normalize", l2_normalize_impl);
  m.impl("clamp_scalar_", clamp_scalar_impl);
  m.impl("activation", activation_impl);
  m.impl("image_embed", image_embed_impl);
  m.impl("attn_x", attn_x_impl);
  m.impl("coca_text_embed", coca_text_embed_impl);
  m.impl("coca_text_mask", coca_text_mask_impl);
  m.impl("rows_linear_f32", rows_linear_f32_impl);
  m.impl("contrastive_fwd", contrastive_fwd_impl);
}

TORCH_LIBRARY_IMPL(mmamd, Meta, m) {
  m.impl("packed", packed_meta);
  m.impl("convert", packed_meta);
  m.impl("layernorm", layernorm_meta);
  m.impl("gemm_bf16", gemm_meta);
  m.impl("attn_fwd", attn_fwd_meta);
  m.impl("attn_probs", attn_probs_meta);
  m.impl("patch_embed", patch_embed_meta);
  m.impl("embed_tokens", embed_tokens_meta);
  m.impl("pool_proj_normalize", pool_proj_normalize_meta);
  m.impl("l2_normalize", same_meta);
  m.impl("clamp_scalar_", clamp_scalar_meta);
  m.impl("activation", activation_meta);
  m.impl("image_embed", image_embed_meta);
  m.impl("attn_x", attn_x_meta);
  m.impl("coca_text_embed", coca_text_embed_meta);
  m.impl("coca_text_mask", coca_text_mask_meta);
  m.impl("rows_linear_f32", rows_linear_f32_meta);
  m.impl("contrastive_fwd", contrastive_fwd_meta);
}

TORCH_LIBRARY_IMPL(mmamd, CompositeExplicitAutograd, m) { m.impl("allgather_packed", allgather_packed_impl); }

// No derivative formulas are registered for these ops (training runs through the autograd.Function nodes of multimodal_amd/_autograd.py,
// whose forward AND backward are C-ABI kernels).  Without an Autograd kernel a scripted / compiled module called with grad mode on would
// return outputs that are silently cut from the graph; with this fallback, differentiating through any mmamd op RAISES
// ("derivative for mmamd::... is not implemented"), and inference under no_grad / with frozen parameters is unaffected.
TORCH_LIBRARY_IMPL(mmamd, Autograd, m) {
  for (const char* name : {"packed", "convert", "layernorm", "gemm_bf16", "attn_fwd", "attn_probs", "patch_embed", "embed_tokens", "pool_proj_normalize",
                           "l2_normalize", "activation", "image_embed", "attn_x", "coca_text_embed", "rows_linear_f32", "contrastive_fwd"})
    m.impl(name, torch::autograd::autogradNotImplementedFallback());
}
