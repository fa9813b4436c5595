/*
 * tante_hip.h -- C ABI of libtante_hip.so, the MI355X (gfx950) kernel library behind the TANTE
 * Taylor-rollout hot path.
 *
 * The reference (zwu88/TANTE) is 100 % Python and has NO existing FFI: every "kernel" on its path is
 * an ATen op reached through nn.Module.forward().  This header therefore declares the boundary a
 * maintainer would bind instead of those ATen calls; each entry point cites the reference lines whose
 * arithmetic it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - plain C: pointers are DEVICE pointers owned by the caller (torch tensors' data_ptr()); the
 *    library never allocates device memory, frees or synchronises.  All launches go to the caller's
 *    hipStream_t (passed as void*), so the calls are HIP-graph capturable.
 *  - process state, all of it: (1) the thread-local last-error string; (2) one pointer per device set by
 *    tante_set_seed_mix() (the device word the fused training kernels XOR into their dropout seeds);
 *    (3) a thread-local cache of hipFFT plans keyed by (device, shape) behind tante_spectral_layer*;
 *    (4) the table of launch-heuristic overrides written ONLY by tante_set_option() -- the library never
 *    reads the environment (a -DTANTE_ABLATE diagnostic build, which is not the product, does).
 *  - return value: 0 on success, negative on error (-1 bad argument, -2 unsupported shape,
 *    -3 HIP launch error); tante_last_error() returns a description.  No exceptions cross the ABI.
 *  - dtype codes: TANTE_F32 = 0, TANTE_BF16 = 1.  "compute" selects the matrix-core path:
 *    TANTE_F32 -> v_mfma_f32_16x16x4_f32 (exact fp32), TANTE_BF16 -> v_mfma_f32_16x16x32_bf16
 *    (fp32 accumulate).  LayerNorm statistics, softmax and GELU are always evaluated in fp32.
 *  - token tensors are (B,T,Hp,Wp,C) row-major, C innermost ("tokens x C").
 */
#ifndef TANTE_HIP_H
#define TANTE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TANTE_F32 0
#define TANTE_BF16 1

/* activation codes */
#define TANTE_ACT_NONE 0
#define TANTE_ACT_GELU_ERF 1  /* nn.GELU()                      enc_dec_cnn.py:215, attn_backbone.py:113 */
#define TANTE_ACT_GELU_TANH 2 /* nn.GELU(approximate="tanh")    attn_backbone.py:54 */
#define TANTE_ACT_RELU 3      /* nn.ReLU()                      tante.py:185,208 */

/* how the rows of the left operand are gathered */
#define TANTE_A_LINEAR 0     /* row r at ((r / n0) * s1 + (r % n0) * s0 + off) elements, K contiguous */
#define TANTE_A_PATCH_NHWC 1 /* row = (img, ho, wo): P x P patch of a channels-last image, k = (kh, kw, ci) */
#define TANTE_A_PATCH_NCHW 2 /* row = (img, ho, wo): P x P patch of a channels-first image, k = (ci, kh, kw) */

/* what the epilogue does with a finished (row, 4 consecutive columns) group */
#define TANTE_E_LINEAR 0      /* out[r * ld + n] = act(v + bias[n]) (+ residual[r * res_ld + n]) */
#define TANTE_E_FILM 1        /* out = (v + bias) * film_a[t][n] + film_b[t][n] + s_emb[hw][n],  r = (b, t, hw) */
#define TANTE_E_DECONV_NHWC 2 /* r = (img, hi, wi), n = (kh, kw, co): out[img][hi*P+kh][wi*P+kw][co] */
#define TANTE_E_DECONV_NCHW 3 /* r = (img, hi, wi), n = (co, kh, kw): out[img][co][hi*P+kh][wi*P+kw]  (fp32) */

/* source layouts understood by tante_pack_weight */
#define TANTE_W_LINEAR 0      /* nn.Linear weight (N, K);  also Conv2d (Cout, Cin, P, P) with k = (ci, kh, kw) */
#define TANTE_W_CONV_NHWC 1   /* Conv2d weight (Cout, Cin, P, P), k re-ordered to (kh, kw, ci) */
#define TANTE_W_DECONV_NHWC 2 /* ConvTranspose2d weight (Cin, Cout, P, P): k = ci, n = (kh, kw, co) */
#define TANTE_W_DECONV_NCHW 3 /* ConvTranspose2d weight (Cin, Cout, P, P): k = ci, n = (co, kh, kw) */
/* transposed packings: the data-gradient GEMM of each forward layout (dX = dY . W), N and K are those of the dgrad GEMM */
#define TANTE_W_LINEAR_T 4      /* source (K, N) row-major: Linear weight (N_fwd = K, K_fwd = N); Conv2d read as (ci,kh,kw) */
#define TANTE_W_CONV_NHWC_T 5   /* Conv2d (Cout, Cin, P, P): n = (kh, kw, ci), k = co;  C_other = Cin */
#define TANTE_W_DECONV_NHWC_T 6 /* ConvTranspose2d (Cin, Cout, P, P): n = ci, k = (kh, kw, co);  C_other = Cout */
#define TANTE_W_DECONV_NCHW_T 7 /* ConvTranspose2d (Cin, Cout, P, P): n = ci, k = (co, kh, kw) */

/* Geometry of a packed weight for (N, K) under a compute dtype.  The packed image is
 * [n_pad / nt tiles][nt rows][k_pad / chunk 16-byte chunks, XOR-swizzled], i.e. exactly the LDS
 * tile image the GEMM streams.  bytes = n_pad * k_pad * sizeof(compute dtype). */
typedef struct TantePackGeom {
  int32_t n_pad, k_pad, nt, cb; /* cb = k_pad / (16 for f32 | 32 for bf16) "chunk blocks" */
  int64_t bytes;
} TantePackGeom;

int tante_pack_geom(int N, int K, int compute, TantePackGeom* out);

/* Pack (and optionally LayerNorm-fold) a weight.  If gamma != NULL the packed matrix is
 * W * diag(gamma) and bias_out[n] = bias[n] + sum_k W[n][k] * beta[k], which turns
 * LayerNorm(x) @ W^T + b (attn_backbone.py:68,74 and :82) into normalise(x) @ W'^T + b'.
 * bias_out has n_pad floats; bias may be NULL (zeros).  P / C_other describe conv layouts:
 * CONV_NHWC: C_other = Cin; DECONV_*: C_other = Cout; bias is indexed by co. */
int tante_pack_weight(const float* w, const float* bias, const float* gamma, const float* beta, int layout, int N,
                      int K, int P, int C_other, int compute, void* w_out, float* bias_out, void* stream);

typedef struct TanteGemm {
  /* left operand */
  const void* a;
  int32_t a_dtype, a_mode;
  int32_t M, K;            /* rows, logical K (<= k_pad) */
  int64_t a_s1, a_s0, a_off; /* LINEAR addressing (elements) */
  int32_t a_n0;
  int32_t Hin, Win, Cin, P; /* PATCH_*: input image (per img) and patch size; rows = imgs*(Hin/P)*(Win/P);
                             * image i starts at (i / a_n0) * a_s1 + (i % a_n0) * Cin*Hin*Win + a_off  (a sliding
                             * window over a longer (B, T_total, ...) buffer needs no copy) */
  int32_t ln;               /* 1: normalise each row over K (biased variance, eps) before the product */
  float ln_eps;
  /* packed right operand */
  const void* w;
  const float* bias; /* n_pad floats */
  int32_t N;
  int32_t compute; /* TANTE_F32 | TANTE_BF16, must match the packing */
  /* epilogue */
  int32_t act, e_mode;
  void* out;
  int32_t out_dtype;
  int64_t out_ld;
  const float* residual; /* LINEAR only, may alias out */
  int64_t res_ld;
  const float *film_a, *film_b, *s_emb; /* FILM: (T, N), (T, N), (HW, N) */
  int32_t T, HW;
  int32_t Hi, Wi, Po, Cout; /* DECONV_*: input grid per img, upsampling factor, output channels */
  /* training epilogues (LINEAR, dense bf16 rows, M >= 4096; zero = off):
   *   drop_p > 0: out = residual + keep(drop_seed, row * N + n) * x / (1 - drop_p), the residual-branch dropout of
   *               attn_backbone.py:57,81-82 applied to the product before the skip is added (same mask as tante_dropout_add);
   *   dact != NULL: out = x * act'(dact[row * N + n]) with act = dact_kind -- the activation backward folded into the data-gradient GEMM.
 *               Also with e_mode = TANTE_E_DECONV_NHWC (no activation, Cout % 4 == 0, any M / K): dact is indexed like `out`, i.e. the
 *               scatter writes d(pre) of the activation that fed a k = s patch conv (enc_dec_cnn.py:221-225 backwards). */
  float drop_p;
  uint64_t drop_seed;
  const void* dact;
  int32_t dact_dtype, dact_kind;
  /* PATCH_NCHW only (ABI 9): 'same' padding of a kernel-P stride-P conv (enc_dec_cnn.py:66-81: (P - 1) / 2): patch (ho, wo) starts at pixel
   * (ho P - a_pad, wo P - a_pad), zero outside the image.  0, or 1 with P = 4 on the bf16 register-stationary path (K = 256 / 512,
   * M >= 4096, linear epilogue, act none / exact GELU); anything else is refused (-2). */
  int32_t a_pad;
} TanteGemm;

/* out = epilogue(gather(a) @ W^T).  Replaces, depending on the descriptor:
 *   F.linear inside nn.MultiheadAttention in/out projection          attn_backbone.py:74-80
 *   LayerNorm + Linear (+GELU) of the block MLP                      attn_backbone.py:50-56,82
 *   RealConv2d with kernel = stride (patch embed) + GELU             enc_dec_cnn.py:97-110,221-225
 *   film(t_seq) + s_emb + t_emb epilogue                             tante.py:136-141,218-230
 *   RealTransConv2d with kernel = stride (derivative head) + GELU    enc_dec_cnn.py:164-184,269-273 */
int tante_gemm(const TanteGemm* g, void* stream);

/* Sequence regrouping of the (B,T,Hp,Wp) token grid for one axis letter (attn_backbone.py:148-182):
 * token(s, l) = (s / n_s0) * S1 + (s % n_s0) * S0 + (l / n_l0) * P1 + (l % n_l0) * P0. */
typedef struct TanteSeq {
  int32_t nseq, L;
  int32_t n_s0;
  int64_t S1, S0;
  int32_t n_l0;
  int64_t P1, P0;
} TanteSeq;

/* o[token, h*d : (h+1)*d] = softmax(q k^T / sqrt(d) [+ causal mask]) v  per (sequence, head);
 * qkv is (tokens, 3C) as produced by the packed in-projection.  Replaces the scaled-dot-product
 * core of nn.MultiheadAttention (attn_backbone.py:74-80) and causal_mask (l.35-36). */
int tante_attention(const void* qkv, void* o, int dtype, int C, int n_head, const TanteSeq* seq, int causal,
                    void* stream);

/* x += W2 gelu_erf(W1 x_line + b1) + b2 along one axis of an fp32 (outer, n, inner) tensor, in place
 * (the vertical / horizontal / temporal propagators, attn_backbone.py:111-119,140-146). */
int tante_axis_mlp(float* x, int64_t outer, int n, int64_t inner, const float* w1, const float* b1, const float* w2,
                   const float* b2, void* stream);

/* The same with the compute mode: TANTE_BF16 allows the polynomial GELU of the bf16 path, TANTE_F32 keeps erff; axes of up to 8
 * positions with inner % 4 == 0 (the temporal propagator) take a float4-vectorised kernel. */
int tante_axis_mlp_c(float* x, int64_t outer, int n, int64_t inner, const float* w1, const float* b1, const float* w2,
                     const float* b2, int compute, void* stream);
/* The same out of place (dst = src + MLP(src) along the axis; src stays intact for the backward pass) for short axes: n <= 8 and
 * inner % 4 == 0 (the temporal propagator); -2 otherwise. */
int tante_axis_mlp_oop(const float* src, float* dst, int64_t outer, int n, int64_t inner, const float* w1, const float* b1, const float* w2,
                       const float* b2, int compute, void* stream);

/* Vertical then horizontal propagator in ONE pass over x (BT, nH, nW, C) fp32, in place (attn_backbone.py:140-143):
 * x += MLP_H(x) along h; x += MLP_W(x) along w.  The n x n contractions run on MFMA in the compute dtype
 * (bf16: operands rounded to bf16, fp32 accumulate, A&S erf; fp32: exact fp32 MFMA, erff).  Needs nH, nW <= 64,
 * C % 16 == 0 and nH * (nW * 16 + 2) * 4 bytes of LDS <= 160 KiB; otherwise call tante_axis_mlp twice. */
int tante_axis_hw(float* x, int64_t BT, int nH, int nW, int C, const float* wh1, const float* bh1, const float* wh2,
                  const float* bh2, const float* ww1, const float* bw1, const float* ww2, const float* bw2, int compute,
                  void* stream);
/* The same, reading every (b, t) plane from a frame-major cache of encoder outputs taken BEFORE FiLM (src[t][b] at
 * src + t * src_t_stride + b * src_b_stride, (nH * nW, C) fp32 each) and applying film(t) + s_emb + t_emb while it loads
 * (tante.py:136-141): x[b, t] = src[t][b] * film_a[t] + film_b[t] + s_emb, then the two propagators; x (B, T, nH, nW, C) is written
 * only.  Lets a rollout loop encode each frame once instead of once per window that contains it. */
int tante_axis_hw_film(float* x, const float* src, int64_t src_t_stride, int64_t src_b_stride, const float* film_a, const float* film_b,
                       const float* s_emb, int T, int64_t BT, int nH, int nW, int C, const float* wh1, const float* bh1, const float* wh2,
                       const float* bh2, const float* ww1, const float* bw1, const float* ww2, const float* bw2, int compute, void* stream);

/* tante_axis_hw out of place: xout = H- then W-propagator of xin, xin intact (a rollout keeps every Taylor order's stream for the one
 * head launch without copying rows aside).  Whole-tile bf16 form only (as tante_axis_hw_train): -2 otherwise, and the caller copies. */
int tante_axis_hw_oop(const float* xin, float* xout, int64_t BT, int nH, int nW, int C, const float* wh1, const float* bh1, const float* wh2,
                      const float* bh2, const float* ww1, const float* bw1, const float* ww2, const float* bw2, int compute, void* stream);

/* The training forward of the two propagators in one launch, out of place: xout = H- then W-propagator of xin (xin stays intact for the
 * backward pass) and xmid = the planes between the two (the W propagator's input, which tante_axis_mlp_bwd needs).  bf16 compute,
 * whole 16-row tiles only (nH, nW multiples of 16 and the plane within the LDS): -2 otherwise, and the caller runs tante_axis_mlp twice. */
int tante_axis_hw_train(const float* xin, float* xout, float* xmid, int64_t BT, int nH, int nW, int C, const float* wh1, const float* bh1,
                        const float* wh2, const float* bh2, const float* ww1, const float* bw1, const float* ww2, const float* bw2, int compute,
                        void* stream);

/* film tables (tante.py:218-230): a[r][c] = 1 + scale(t[r])[c], b[r][c] = shift(t[r])[c] (+ add[r][c]).
 * scale/shift = Linear(1, C/2) -> ReLU -> Linear(C/2, C).  rows = len(t). */
int tante_film_table(const float* t, int rows, int C, const float* sc_w0, const float* sc_b0, const float* sc_w2,
                     const float* sc_b2, const float* sh_w0, const float* sh_b0, const float* sh_w2,
                     const float* sh_b2, const float* add, float* a_out, float* b_out, void* stream);

/* Backward of tante_film_table: dA, dB (rows, C) -> the gradients of the eight MLP parameters, written or (accumulate) added onto
 * existing buffers; d(add) = dB is the caller's.  One launch instead of torch's ten small GEMMs + their glue per train step. */
int tante_film_table_bwd(const float* t, int rows, int C, const float* sc_w0, const float* sc_b0, const float* sc_w2, const float* sh_w0,
                         const float* sh_b0, const float* sh_w2, const float* dA, const float* dB, float* g_sc_w0, float* g_sc_b0,
                         float* g_sc_w2, float* g_sc_b2, float* g_sh_w0, float* g_sh_b0, float* g_sh_w2, float* g_sh_b2, int accumulate,
                         void* stream);

/* y[r][c] = x_row(r)[c] * a[g][c] + b[g][c], g = r / rows_per, x_row(r) = x + g * x_bstride + (r % rows_per) * C
 * (film on (B, L, C) tokens, tante.py:222-224,229-230; tables hold 1+scale and shift). */
int tante_film_apply(const float* x, int64_t x_bstride, float* y, int64_t rows, int C, int64_t rows_per, const float* a,
                     const float* b, void* stream);

/* out[i] = z[i * E + E - 1]: the "[..., -1]" of the channel-attention letter 'C' (attn_backbone.py:188). */
int tante_gather_last(const float* z, int64_t n, int E, float* out, void* stream);

/* DefaultChannelsFirstFormatter.process_input for the model input (data/datamodule.py:184-192): x (n_img, HW, D) channels-last fp32 ->
 * nan_to_num -> channels-first images, image i = (b, t) at out + b * out_bstride + t * D * HW (e.g. the head of a rollout buffer). */
int tante_format_input(const float* x, int64_t n_img, int T, int64_t HW, int D, float* out, int64_t out_bstride, void* stream);
/* torch.nan_to_num over a dense fp32 tensor (NaN -> 0, +-inf -> +-FLT_MAX): the formatter's pass over the reference frames
 * `y_ref` (data/datamodule.py:187, `torch.nan_to_num(data["output"])`).  x, y 16-byte aligned, n elements; y may equal x. */
int tante_nan_to_num(const float* x, float* y, int64_t n, void* stream);

/* Taylor sum (tante.py:165-171): out[b][i-1] = last[b] + sum_k derivs[k][b] * (i * dt)^k / k!,
 * i = 1..n_out.  last = input[:, -1] given as base pointer + batch stride (elements);
 * derivs = n_order device pointers, each (B, frame) contiguous; out[b] starts at out + b * out_bstride (elements),
 * so the prediction can be written straight into the next window of a rollout buffer. */
int tante_taylor(const float* last, int64_t last_bstride, const float* const* derivs, int n_order, double dt,
                 int n_out, float* out, int64_t out_bstride, int64_t B, int64_t frame, void* stream);

/* interprator head reduction (tante.py:194-201): t (B, L) raw per-token scalars ->
 * rt[b] = mean_l clamp(t[b][l], 0, out_T - 1) + ep. */
int tante_rt_reduce(const float* t, int B, int L, float out_T, float ep, float* rt, void* stream);

/* ---- fused TransformerBlock (bf16 MFMA path) ----------------------------------------------------
 * tante_block_fused: x (tokens, C) fp32, in place, ONE launch for the whole block
 *     x += out_proj(MHA(LayerNorm1(x)));  x += W2 gelu_tanh(W1 LayerNorm2(x) + b1) + b2   attn_backbone.py:59-83
 * The residual rows are read once, stay in registers and are written once; q, k, v, scores, per-head
 * outputs and the MLP hidden layer never leave the CU.  It consumes the "weight stream" built by
 * tante_pack_block from the block's 12 fp32 parameters (LayerNorm affine folded, 1/sqrt(d) folded into q,
 * bf16, k-permuted LDS tile images in consumption order).
 * Supported: head dim 32, C in {64,128,256}, hidden in {C, 2C} (C = 256: hidden = 256), sequence length L
 * dividing 32 (tante_block_fused_supported); anything else goes through tante_gemm + tante_attention. */
int tante_block_fused_supported(int C, int n_head, int hidden, int L);
int64_t tante_block_stream_bytes(int C, int hidden);
int tante_pack_block(const float* ln1_w, const float* ln1_b, const float* in_w, const float* in_b, const float* out_w,
                     const float* out_b, const float* ln2_w, const float* ln2_b, const float* fc1_w, const float* fc1_b,
                     const float* fc2_w, const float* fc2_b, int C, int hidden, void* block_stream, void* stream);
int tante_block_fused(float* x, const void* block_stream, int C, int n_head, int hidden, const TanteSeq* seq, int causal,
                      float eps, void* stream);
/* The same at L = 4 (the T letter) WITH the temporal propagator of attn_backbone.py:144-145 applied to the rows first, inside the launch:
 * tprop = w1 (4 x 4, row-major), b1 (4), w2 (4 x 4), b2 (4) -- 40 floats in device memory.  Equals tante_axis_mlp_c (bf16 compute) along the
 * T axis followed by tante_block_fused, bit for bit, in one pass over x. */
int tante_block_fused_tprop(float* x, const void* block_stream, int C, int n_head, int hidden, const TanteSeq* seq, int causal, float eps,
                            const float* tprop, void* stream);

/* Training forward of a whole TransformerBlock in ONE launch (C = 256, 8 heads, hidden 256, sequences up to 64 tokens): the same
 * arithmetic as tante_block_fused with dropout (attn_backbone.py:47-83 in train() mode: attention-probability dropout inside
 * nn.MultiheadAttention, nn.Dropout on both residual branches) and every tensor the backward pass of the unfused operators reads,
 * stored in THEIR layouts, so that tante_layernorm_bwd / tante_attention_bwd / the data- and weight-gradient GEMMs run unchanged:
 *   xh1, xh2 (tokens, 256) bf16  LayerNorm outputs (no affine: gamma / beta are folded into the consumer weights)
 *   st1, st2 (tokens, 2) fp32    (mean, rstd) per token
 *   qkv (tokens, 768) bf16       packed projection, q NOT scaled, biases included (may be NULL: tante_block_bwd_fused recomputes it)
 *   o (tokens, 256) bf16         attention output (after probability dropout)
 *   x1 (tokens, 256) fp32        x + dropout(out_proj(o))   (may be NULL: tante_block_tail_bwd works from xh2 / st2 and never reads it)
 *   hpre, act (tokens, 256) bf16 fc1 pre-activation and its tanh-GELU
 *   out (tokens, 256) fp32       x1 + dropout(fc2(act))   (x itself is left untouched)
 * Masks are dropout_keep(seed, index, p) with tante_attention_dropout's index for seed_attn and row * 256 + column (the GEMM
 * epilogue's, tante_dropout_bwd's) for seed_out / seed_mlp. */
/* A 64-bit word in DEVICE memory that the fused training kernels (tante_block_fused_train, tante_block_tail_bwd, the MFMA form of
 * tante_attention_bwd) XOR into every dropout seed they are given, read when the kernel runs: a HIP graph of a train step replays its
 * by-value seeds unchanged, so the host rewrites this word before each replay and the step draws fresh masks.  Per device (the
 * current one); NULL (the default) switches it off.  The other dropout entry points (tante_dropout_*, tante_attention_dropout, the
 * tante_gemm epilogue) take their seeds as given. */
int tante_set_seed_mix(const uint64_t* device_word);

typedef struct TanteBlockTrain {
  float* out;
  void *xh1, *qkv, *o, *xh2, *hpre, *act;
  float *st1, *x1, *st2;
  float p_drop;
  uint64_t seed_attn, seed_out, seed_mlp;
} TanteBlockTrain;
/* Packs ONLY the stream tante_block_fused_train reads (into the buffer tante_block_stream_bytes sizes), from weights whose LayerNorm
 * affines are already folded in (W diag(gamma), b + W beta: the training path keeps those as differentiable tensors). */
int tante_pack_block_train(const float* in_w_folded, const float* in_b_folded, const float* out_w, const float* out_b,
                           const float* fc1_w_folded, const float* fc1_b_folded, const float* fc2_w, const float* fc2_b, int C, int hidden,
                           void* block_stream, void* stream);
int tante_block_fused_train(const float* x, const void* block_stream, int C, int n_head, int hidden, const TanteSeq* seq, int causal,
                            float eps, const TanteBlockTrain* tr, void* stream);

/* Backward of the block's tail in ONE launch (C = 256, hidden 256; block_bwd.hip): from the gradient of the block output to the gradient
 * of the attention output, i.e. the backward of  out = x1 + drop(fc2(gelu_tanh(fc1(LayerNorm2(x1))))),  x1 = xs + drop(out_proj(o))
 * (attn_backbone.py:81-82; six launches unfused).  Reads dout (M, 256) fp32 and what tante_block_fused_train saved (hpre, xh2, st2) with
 * its seeds; writes dx1 (M, 256) fp32 = the gradient reaching x1 (skip path + LayerNorm2 path; it is also the gradient of the skip
 * operand xs), d_o (M, 256) bf16 = the gradient of the attention output, and the bf16 row operands of the three weight gradients:
 * dy2 (fc2: dW2 = dy2^T act), dhpre (fc1: dW1' = dhpre^T xh2), dy1 (out-proj: dWo = dy1^T o).  The transposed weights come pre-packed:
 * tante_pack_block_tail_bwd(fc2.weight, folded fc1 weight, out_proj.weight) into tante_block_tail_bwd_stream_bytes bytes. */
int64_t tante_block_tail_bwd_stream_bytes(int C, int hidden);
int tante_pack_block_tail_bwd(const float* fc2_w, const float* fc1_w_folded, const float* out_w, int C, int hidden, void* bwd_stream, void* stream);
int tante_block_tail_bwd(const float* dout, const void* hpre, const void* xh2, const float* st2, const void* bwd_stream, int64_t M, int C,
                         int hidden, float p_drop, uint64_t seed_out, uint64_t seed_mlp, float* dx1, void* dy2, void* dhpre, void* dy1,
                         void* d_o, void* stream);

/* The FRONT of the block's backward in one launch: dxh = dqkv W_in' (the q | k | v data gradient, W_in' = the LayerNorm1-folded
 * in-projection weight (768, 256)) and LayerNorm1's backward with the skip gradient added:
 *   dx = dx1 + rstd1 (dxh - mean(dxh) - xh1 mean(dxh xh1))     (attn_backbone.py:79-80 backwards; replaces a data-gradient tante_gemm +
 * tante_layernorm_bwd).  dqkv (M, 768) bf16 = tante_attention_bwd's result, xh1 (M, 256) bf16 and st1 (M, 2) fp32 = the forward's saved
 * LayerNorm1 image and statistics, dx1 (M, 256) fp32 = the gradient reaching the block input through the skip path (tante_block_tail_bwd's
 * dx1), dx (M, 256) fp32 = the gradient of the block input.  head_bwd_stream = tante_pack_block_tail_bwd(W_in'[0:256], W_in'[256:512],
 * W_in'[512:768]): the three row blocks of the folded weight, transposed into fragments like the tail's three weights. */
int tante_block_head_bwd(const void* dqkv, const void* xh1, const float* st1, const float* dx1, const void* head_bwd_stream, int64_t M, int C,
                         float* dx, void* stream);

/* The WHOLE backward of a TransformerBlock in ONE launch (block_bwd_fs.hip; attn_backbone.py:59-83 backwards, trainer/trainer.py:191):
 * tante_block_tail_bwd + tante_attention_bwd + tante_block_head_bwd with nothing but LDS and registers in between, on the forward
 * kernel's partition (a workgroup owns whole sequences of `seq`).  Reads dout (tokens, 256) fp32 and what tante_block_fused_train saved:
 * xh1, xh2, hpre (tokens, 256) bf16 and st1, st2 (tokens, 2) fp32 -- NOT the packed projection: q | k | v are recomputed from xh1 with the
 * forward's own weights (block_stream = the buffer tante_pack_block_train filled), so the training forward may be called with qkv = NULL.
 * tail_bwd_stream = tante_pack_block_tail_bwd(fc2, folded fc1, out_proj), head_bwd_stream = the same packing of the three 256-row blocks
 * of the folded in-projection weight.  Dropout masks are regenerated from (p_drop, seeds) as the forward drew them.  Writes dx (tokens,
 * 256) fp32 and the bf16 row operands of the four weight gradients: dy2 (fc2: dW2 = dy2^T act), dhpre (fc1: dW1' = dhpre^T xh2), dy1
 * (out-proj: dWo = dy1^T o), dqkv (tokens, 768) (in-proj: dW_in' = dqkv^T xh1).  Shapes: C = 256, 8 heads, hidden 256, L | 16 (any
 * mask) or L in {32, 48, 64} non-causal (tante_block_bwd_fused_supported); others take the three launches above. */
int tante_block_bwd_fused_supported(int C, int n_head, int hidden, int L, int causal);
int tante_block_bwd_fused(const float* dout, const void* xh1, const float* st1, const void* hpre, const void* xh2, const float* st2,
                          const void* tail_bwd_stream, const void* block_stream, const void* head_bwd_stream, int C, int n_head, int hidden,
                          const TanteSeq* seq, int causal, float p_drop, uint64_t seed_attn, uint64_t seed_out, uint64_t seed_mlp, float* dx,
                          void* dy2, void* dhpre, void* dy1, void* dqkv, void* stream);

/* ---- fused derivative head (bf16 MFMA path) ----------------------------------------------------------
 * One launch per Taylor order: rows r = (img, hp, wp) of the token stream (gathered like TANTE_A_LINEAR: the last time slot by
 * stride) -> 3 x [ConvTranspose2d k = s = 2 (+GELU erf)] -> for i < n_out:  out_i (+)= coefs[i] * derivative, where out_i is frame i
 * of out[img] (out + img * out_bstride + i * D*H*W, channels-first) and, when `last` != NULL, the sum starts from the last input
 * frame (last + img * last_bstride) instead of accumulating.  Replaces dec_CNN.forward (enc_dec_cnn.py:263-277) and the Taylor sum
 * (tante.py:165-171); the three intermediate images and the derivative fields never exist.  C in {128, 256}, D <= 16, patch_scale 8. */
int tante_head_fused_supported(int C, int D);
int64_t tante_head_stream_bytes(int C);
int tante_pack_head(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, int C, int D,
                    void* head_stream, void* stream);
int tante_head_fused(const float* x, int32_t a_n0, int64_t a_s1, int64_t a_s0, int64_t a_off, int n_img, int Hp, int Wp, int C, int D,
                     const void* head_stream, float* out, int64_t out_bstride, int n_out, const float* coefs, const float* last,
                     int64_t last_bstride, void* stream);
/* Every Taylor order's derivative head in ONE launch, one prediction frame (tante.py:145-154 + 165-171 with output_length = 1):
 *   out = last + sum_k coefs[k] * head_k(rows_k),  k < n_ord <= 4.
 * rows[k], k < n_ord - 1: dense (n_img * Hp * Wp, C) fp32 copies of the last-slot token rows as backbone k left them (the stream is
 * updated in place by the later backbones); rows[n_ord - 1] is the stream itself, addressed through (a_n0, a_s1, a_s0, a_off) like
 * tante_head_fused.  head_streams[k]: tante_pack_head of decoder k.  The frame is read (`last`) and written once, not once per order. */
int tante_head_fused_multi(int n_ord, const float* const* rows, const void* const* head_streams, const float* coefs, int32_t a_n0,
                           int64_t a_s1, int64_t a_s0, int64_t a_off, int n_img, int Hp, int Wp, int C, int D, float* out,
                           int64_t out_bstride, const float* last, int64_t last_bstride, void* stream);
/* The same with every order's rows addressed by (a_n0, a_s1, a_s0, a_off): rows[k] = the whole residual stream as backbone k left it
 * (each backbone writes a buffer of its own through tante_axis_hw_oop; nothing is copied aside). */
int tante_head_fused_multi_streams(int n_ord, const float* const* rows, const void* const* head_streams, const float* coefs, int32_t a_n0,
                           int64_t a_s1, int64_t a_s0, int64_t a_off, int n_img, int Hp, int Wp, int C, int D, float* out,
                           int64_t out_bstride, const float* last, int64_t last_bstride, void* stream);

/* Attention over DENSE sequences (token b * L + l) with nn.MultiheadAttention's masks as additive fp32 tensors: attn_mask (L, L) shared
 * (mask_bstride 0) or (Bp * n_head, L, L) (mask_bstride L * L), key_padding_mask (Bp, L); either may be NULL; -inf blocks a key.
 * qkv / o layouts as tante_attention.  The reference's TransformerBlock.forward(x, key_padding_mask, attn_mask, causal) signature
 * (attn_backbone.py:59-72); the TANTE path passes `causal` only, so this is a completeness kernel (one lane per query). */
int tante_attention_masked(const void* qkv, void* o, int dtype, int C, int n_head, int Bp, int L, int causal, const float* attn_mask,
                           int64_t mask_bstride, const float* key_padding_mask, void* stream);
/* Its backward: dqkv (tokens, 3C) = (dq | dk | dv) from dO (tokens, C), probabilities recomputed from qkv under the same masks (no
 * attention dropout on this path).  stats: Bp * n_head * L * 3 floats of scratch (row max, 1 / row sum, dO . O).  Also the train
 * path's fallback for dense sequences longer than tante_attention_bwd takes (L > 128: the channel letter 'C' over 256 channels,
 * attn_backbone.py:176-184).  Head dims 4, 8, 16, 32, 64.  Deterministic (fixed-order sums, no atomics). */
int tante_attention_masked_bwd(const void* qkv, const void* dO, void* dqkv, int dtype, int C, int n_head, int Bp, int L, int causal,
                               const float* attn_mask, int64_t mask_bstride, const float* key_padding_mask, float* stats, void* stream);

/* ---- the token-local tail of a rollout call in ONE launch (head_enc.hip; bf16, C = 256, D <= 16, Hp Wp % 16 == 0) -----------------
 * tante_head_enc_fused: every Taylor order's derivative head + the Taylor sum (as tante_head_fused_multi_streams: rows[k] = the residual
 * stream backbone k left, addressed by (a_n0, a_s1, a_s0, a_off), a_n0 % 16 == 0) and, when enc_stream != NULL, the RE-ENCODING of the
 * predicted frame for the next call: with patch_scale 8 the 8 x 8 x D block a token's heads write (enc_dec_cnn.py:263-277,
 * tante.py:165-171) is exactly the block the three encoder stages reduce back to that token (enc_dec_cnn.py:217-229), so the frame
 * is stored (it is the output) and encoded from the same registers:  z (n_img Hp Wp, C) fp32 = enc_CNN(frame) before FiLM, the
 * rollout's frame-cache entry that tante_enc23_frames would otherwise produce from the stored frame (2 launches, 25 MB).
 * enc_stream = tante_pack_head_enc(conv1.w, conv1.b, conv2.w, conv2.b, conv3.w, conv3.b).  ws: tante_head_enc_ws_bytes(rows) bytes,
 * ZEROED ONCE by the caller before the first use (the four pixel workgroups of a token group hand their encoder stage-2 outputs over
 * through it as bf16 operand fragments; the last to arrive runs stage 3 over the whole K = 512 in a fixed tap order: deterministic --
 * followed by one arrival counter per group, which the kernel leaves at zero).  One workspace per stream of concurrent launches.
 * enc_stream == NULL: heads + Taylor sum only (z, ws unused). */
int tante_head_enc_supported(int C, int D);
int64_t tante_head_enc_stream_bytes(int C);
int64_t tante_head_enc_ws_bytes(int64_t rows);
int tante_pack_head_enc(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, int C, int D,
                        void* enc_stream, void* stream);
int tante_head_enc_fused(int n_ord, const float* const* rows, const void* const* head_streams, const float* coefs, int32_t a_n0,
                         int64_t a_s1, int64_t a_s0, int64_t a_off, int n_img, int Hp, int Wp, int C, int D, float* out,
                         int64_t out_bstride, const float* last, int64_t last_bstride, const void* enc_stream, float* z, void* ws,
                         int64_t ws_bytes, void* stream);

/* ---- general encoder / decoder stages, spectral operator path, CViT (operators.hip) ---------------------------------
 * tante_im2col: rows = output positions (img, oh, ow) of a convolution with kernel (kh, kw), stride (sh, sw), zero padding (ph, pw)
 *   over x (n_img, C, H, W) [nchw = 1] or (n_img, H, W, C) [nchw = 0]; columns ordered (c, kh, kw) [korder 0, the native
 *   nn.Conv2d weight flatten] or (kh, kw, c) [korder 1].  With tante_gemm it is RealConv2d's conv for every patch_size /
 *   overlap_ratio / 'same' padding (enc_dec_cnn.py:49-96) and CViT's Conv3d patch embed with kernel (1, p, p) (cvit.py:73-78). */
int tante_im2col(const void* x, int x_dtype, int nchw, int64_t n_img, int C, int H, int W, int kh, int kw, int sh, int sw, int ph,
                 int pw, int korder, void* cols, int cols_dtype, void* stream);
/* F.adaptive_avg_pool2d to (Ht, Wt) on a channels-last image, then `act` (RealConv2d.forward, enc_dec_cnn.py:104-110). */
int tante_avgpool_nhwc(const void* x, int x_dtype, int64_t n_img, int H, int W, int C, int Ht, int Wt, int act, void* y, int y_dtype,
                       void* stream);
/* Its backward without the activation (autograd of enc_dec_cnn.py:104-110 on the train path): dx (n_img, H, W, C) from dy (n_img, Ht, Wt, C);
 * every input pixel gathers dy / window area from the one or two cells per axis whose adaptive windows contain it. */
int tante_avgpool_nhwc_bwd(const void* dy, int dy_dtype, int64_t n_img, int H, int W, int C, int Ht, int Wt, void* dx, int dx_dtype,
                           void* stream);
/* Gather half of a ConvTranspose2d whose taps overlap (stride < kernel P, padding `pad`; RealTransConv2d with overlap_ratio > 0,
 * enc_dec_cnn.py:128-166): cols (n_img*Hi*Wi, P*P*Cout) with columns (kh, kw, co) is the tap matrix from one GEMM; out
 * (n_img, Hf, Wf, Cout) channels-last, Hf = (Hi - 1) stride - 2 pad + P, gets the overlapping taps summed plus the bias. */
int tante_col2im_nhwc(const void* cols, int cols_dtype, int64_t n_img, int Hi, int Wi, int P, int stride, int pad, int Cout,
                      const float* bias, void* out, int out_dtype, void* stream);
/* F.interpolate(mode="bilinear", align_corners=False) of the (Hi, Wi) window at (crop_y, crop_x) of `in` to (Ho, Wo), then `act`
 * (RealTransConv2d.forward, enc_dec_cnn.py:164-184: the padded transposed conv is the unpadded one cropped by the padding).
 * Element (img, c, y, x) of in / out lives at img*sn + c*sc + y*sh + x*sw (elements). */
int tante_resize_bilinear(const void* in, int in_dtype, int64_t n_img, int C, int Hi, int Wi, int crop_y, int crop_x, int64_t isn,
                          int64_t isc, int64_t ish, int64_t isw, int Ho, int Wo, int64_t osn, int64_t osc, int64_t osh, int64_t osw,
                          int act, void* out, int out_dtype, void* stream);
/* y = LayerNorm(x) * gamma + beta per row (nn.LayerNorm; cvit.py:124-127, 228, 269, 412-414).  gamma / beta may be NULL. */
int tante_layernorm_affine(const void* x, int x_dtype, int64_t M, int C, float eps, const float* gamma, const float* beta, void* y,
                           int y_dtype, void* stream);
/* SpectralLayer.forward (enc_dec_fno.py:184-222) on x (n, Cin, H, W) fp32:  out = act(irfft2(low-mode contraction of rfft2(x)) +
 * conv1x1(x)), norm "ortho".  w_re / w_im: the complex weight (Cin, Cout, wm1, wm2) split into planes; modes clip to
 * min(modes1, H) x min(modes2, W/2 + 1); the bottom band overwrites the top one where they overlap (l.203-210).  FFTs run through
 * hipFFT plans cached per shape on the caller's stream; `work` holds the two spectra (tante_spectral_workspace_bytes). */
int64_t tante_spectral_workspace_bytes(int64_t n, int Cin, int Cout, int H, int W);
int tante_spectral_layer(const float* x, int64_t n, int Cin, int H, int W, const float* w_re, const float* w_im, int wm1, int wm2,
                         int modes1, int modes2, const float* w0, const float* b0, int Cout, int act, float* out, void* work,
                         int64_t work_bytes, void* stream);
/* The same with a compute mode: TANTE_F32 = tante_spectral_layer (exact fp32 matrix products); TANTE_BF16 lets the inverse row transform
 * + 1x1 conv run as split-operand products on the bf16 matrix pipe (every fp32 operand as hi + lo bf16 parts, three products, fp32
 * accumulation: ~1e-5 relative to the fp32 result) where the shape allows -- the mode of a bf16 model, whose bar is 1e-2. */
int tante_spectral_layer_c(const float* x, int64_t n, int Cin, int H, int W, const float* w_re, const float* w_im, int wm1, int wm2,
                           int modes1, int modes2, const float* w0, const float* b0, int Cout, int act, float* out, void* work,
                           int64_t work_bytes, int compute, void* stream);
/* tante_spectral_layer_c in TANTE_BF16 mode writing act(layer(x)) as a bf16 (n, Cout, H, W) image (round to nearest even): for a layer whose
 * output only feeds a bf16 patch gather (enc_FNO: spectral -> GELU -> conv, enc_dec_fno.py:224-273) -- tante_im2col reads bf16 images --
 * with the same final bits and half the bytes both ways.  tante_spectral_bf16out_supported: the shape has this form (truncated-DFT
 * path, W % 128 == 0, Cout <= 32, Cout % 4 == 0, Cin <= 64).  work as tante_spectral_layer. */
int tante_spectral_bf16out_supported(int64_t n, int Cin, int Cout, int H, int W, int modes1, int modes2);
int tante_spectral_layer_bf16out(const float* x, int64_t n, int Cin, int H, int W, const float* w_re, const float* w_im, int wm1, int wm2,
                                 int modes1, int modes2, const float* w0, const float* b0, int Cout, int act, void* out, void* work,
                                 int64_t work_bytes, void* stream);
/* (ABI 12) tante_spectral_layer_c in TANTE_BF16 mode with the images of x `x_istride` elements apart (>= Cin H W: one frame of every batch
 * item inside a rollout buffer, trainer/trainer.py:144-159's window without the copy) and the output layout chosen by out_mode:
 * 0 = fp32 (n, Cout, H, W), 1 = bf16 (n, Cout, H, W), 2 = fp32 channels-last rows ((n h w), Cout) -- what the transposed-conv GEMM behind
 * dec_FNO's first spectral layer reads (enc_dec_fno.py:276-323).  Served by the split-bf16 kernels of the truncated-DFT path only:
 * _supported says whether (shape, strided, out_mode) is; -2 otherwise.  x, out, w0, b0 16-byte aligned; work as tante_spectral_layer. */
int tante_spectral_layer_x_supported(int64_t n, int Cin, int Cout, int H, int W, int modes1, int modes2, int strided, int out_mode);
int tante_spectral_layer_x(const float* x, int64_t x_istride, int64_t n, int Cin, int H, int W, const float* w_re, const float* w_im, int wm1,
                           int wm2, int modes1, int modes2, const float* w0, const float* b0, int Cout, int act, void* out, int out_mode,
                           void* work, int64_t work_bytes, void* stream);

/* Backward of tante_spectral_layer (act none): dx (n, Cin, H, W) = irfft2(M^H rfft2(dy)) + W0^T dy, and the complex weight gradient
 * dw_re / dw_im (Cin, Cout, wm1, wm2) in PyTorch's convention (dL/dRe + i dL/dIm).  w0t: the 1x1 weight transposed, (Cin, Cout).
 * The 1x1 conv's own weight / bias gradients are ordinary reductions (tante_wgrad lines, tante_colsum). */
int tante_spectral_layer_bwd(const float* x, const float* dy, int64_t n, int Cin, int H, int W, const float* w_re, const float* w_im, int wm1,
                             int wm2, int modes1, int modes2, const float* w0t, int Cout, float* dx, float* dw_re, float* dw_im, void* work,
                             int64_t work_bytes, void* stream);
/* tante_col2im_nhwc with an explicit output size: the input gradient of a (padded / strided) convolution whose patch-matrix gradient is
 * `cols` with columns (kh, kw, c) -- pixels no patch covers stay zero. */
int tante_col2im_nhwc_sized(const void* cols, int cols_dtype, int64_t n_img, int Hi, int Wi, int P, int stride, int pad, int Cout,
                            const float* bias, int Hf, int Wf, void* out, int out_dtype, void* stream);
/* Backward of tante_resize_bilinear: the gradient of every output pixel is ADDED (fp32 atomics) to its source pixels in `din`
 * (zero it first); same addressing convention as the forward. */
int tante_resize_bilinear_bwd(const void* dout, int d_dtype, int64_t n_img, int C, int Hi, int Wi, int crop_y, int crop_x, int64_t isn,
                              int64_t isc, int64_t ish, int64_t isw, int Ho, int Wo, int64_t osn, int64_t osc, int64_t osh, int64_t osw,
                              float* din, void* stream);
/* softmax(q k^T / sqrt(D)) v per (batch, head) with separate query and key/value sequences -- the core of
 * nn.MultiheadAttention(q, kv, kv) in CViT's CrossAttnBlock / TimeAggregation / SelfAttnBlock (cvit.py:125, 162, 199-204).
 * Rows: q (b, i) at (b*Lq + i)*ldq + h*D, k / v (b, j) at (b*Lk + j)*ldkv + h*D, o at (b*Lq + i)*ldo + h*D (elements). */
int tante_cross_attention(const void* q, const void* k, const void* v, void* o, int dtype, int64_t n_batch, int n_head, int D, int Lq,
                          int Lk, int64_t ldq, int64_t ldkv, int64_t ldo, void* stream);
/* CViT blocks at width 512 (8 heads x 64, mlp_ratio 1), bf16: everything behind the attention of a SelfAttnBlock / CrossAttnBlock
 * (models/cvit.py:112-169) in one launch, 64 tokens per workgroup (cvit_fused.hip).
 *   mode 0: x1 = out_proj(a) + resid; out (M, 512) fp32 = x1 + fc2(gelu(fc1(LN2(x1))))
 *   mode 1: ... then the model's tail (cvit.py:459-466, Mlp with one layer, cvit.py:213-242): z = norm2(.), y = z + gelu(dense(z)),
 *           out (M, out_dim <= 16) fp32 = output_layer(LN(y))
 * a: (M, 512) bf16 attention output; resid: (resid_period, 512) fp32, token t adds row t % resid_period (the decoder's queries are the
 * same coordinate embedding for every sample); M and resid_period multiples of 16 (64-token workgroups for long launches, 16-token
 * ones for short ones: TANTE_CVIT_CHAIN_TOKENS overrides).  w: the matrices as bf16 operand fragments,
 * 512 KiB each, out_proj | fc1 | fc2 [| dense], fragment (wave, k-step, row tile) at ((wave*16 + ks)*4 + j) KiB holding for lane
 * (l15, kk) the 8 values W[64*wave + 16*j + l15][32*ks + 8*kk ..]; LN2's affine folded into fc1 (and its bias).  bias: (n_mat, 512).
 * g2 / b2: norm2's affine.  wout: 16 fragments (wave, pair p), lane (row, kk) = W'[row][64*wave + 32*p + 4*kk + 0..3], then the same
 * columns + 16 -- the Mlp's LayerNorm folded in, rows >= out_dim zero; bout: 16 floats.  tante_amd/cvit.py builds all of them. */
int tante_cvit_chain512(const void* a, const float* resid, int64_t resid_period, const void* w, const float* bias, const float* g2,
                        const float* b2, float eps_ln2, float eps_norm2, float eps_mlp, const void* wout, const float* bout, int out_dim,
                        int64_t M, int mode, float* out, void* stream);
/* Mode 0 of tante_cvit_chain512 followed, in the same launch, by the NEXT SelfAttnBlock's input projection (cvit.py:129-134 of the block
 * that follows): qkv (M, 1536) bf16 = in_proj'(LN1_next(out)), LN1 folded into the weights.  w: six packed matrices
 * out_proj | fc1 | fc2 | Wq | Wk | Wv (the last three from the next block), bias (6, 512). */
int tante_cvit_chain512_qkv(const void* a, const float* resid, int64_t resid_period, const void* w, const float* bias, float eps_ln2,
                            float eps_ln1_next, int64_t M, float* out, void* qkv, void* stream);
/* The same with the query rows of sample b starting at row b * q_batch_rows (q_batch_rows = Lq: tante_cross_attention; 0: every sample
 * attends with the SAME Lq queries -- the decoder's coordinate queries, cvit.py:452, whose projection is then computed once). */
int tante_cross_attention_q(const void* q, const void* k, const void* v, void* o, int dtype, int64_t n_batch, int n_head, int D, int Lq,
                            int Lk, int64_t ldq, int64_t ldkv, int64_t ldo, int64_t q_batch_rows, void* stream);
/* Backward of tante_cross_attention.  o is the forward output; dq gets the query gradient in the layout of q; the key / value gradients
 * are ADDED (fp32 atomics) to dk / dv, rows (b, j) at (b*Lk + j)*ldg + h*D -- zero them first.  stats: n_batch*n_head*Lq*3 floats of
 * scratch (softmax max, 1 / sum, dO . O per query). */
int tante_cross_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* dO, void* dq, float* dk, float* dv,
                              float* stats, int dtype, int64_t n_batch, int n_head, int D, int Lq, int Lk, int64_t ldq, int64_t ldkv,
                              int64_t ldo, int64_t ldg, void* stream);
/* Backward of tante_layernorm_affine: dx (fp32), and dgamma / dbeta ADDED to the given buffers (may be NULL). */
int tante_layernorm_affine_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma, int64_t M, int C, float eps,
                               float* dx, float* dgamma, float* dbeta, void* stream);
/* Backward of tante_grid_embed for the latents and the (trainable) grid positions; `out` is the forward result, wsum_work N floats. */
int tante_grid_embed_bwd(const float* coords, const float* grid, const float* latents, const float* out, const float* dout, int64_t N,
                         int G, int LD, float eps, float* wsum_work, float* dlatents, float* dgrid, void* stream);
/* CViT grid embedding (cvit.py:434-438): out[n] = sum_g softmax_g(-eps |coords_n - grid_g|^2) latents[g];  coords (N, 2),
 * grid (G, 2), latents (G, LD), LD <= 1024.  Exact: every grid point is evaluated, exact-zero weights are skipped. */
int tante_grid_embed(const float* coords, const float* grid, const float* latents, int64_t N, int G, int LD, float eps, float* out,
                     void* stream);
/* FourierEmbs (cvit.py:308-331): out[n] = [cos(coords_n . K), sin(coords_n . K)], K (2, E/2). */
int tante_fourier_embed(const float* coords, const float* kernel, int64_t N, int E, float* out, void* stream);

/* ---- fused patch-embed stages 2 + 3 (bf16 MFMA path) -----------------------------------------------------
 * h1: the stage-1 image, channels-last bf16 (n_img, 4 Hp, 4 Wp, C/4) as tante_gemm writes it.  One launch computes
 * Conv2d(k = s = 2) -> GELU(erf) -> Conv2d(k = s = 2) -> x * film_a[t] + film_b[t] + s_emb[hw] into the fp32 token stream
 * out (n_img * Hp * Wp, C), t = img % T  (enc_dec_cnn.py:221-229 stages 2-3 + tante.py:136-141).  C = 256, patch_scale 8. */
int tante_enc23_supported(int C);
int64_t tante_enc23_stream_bytes(int C);
int tante_pack_enc23(const float* w2, const float* b2, const float* w3, const float* b3, int C, void* enc_stream, void* stream);
int tante_enc23_fused(const void* h1, int n_img, int Hp, int Wp, int C, const void* enc_stream, const float* film_a,
                      const float* film_b, const float* s_emb, int T, float* out, void* stream);
/* The same two stages WITHOUT the FiLM / positional epilogue, for a rollout loop that encodes every frame once: images are ordered
 * (b, f) with f < frames, and image (b, f) is written to the frame-major row block (f * (n_img / frames) + b) * Hp * Wp of out
 * (the cache tante_axis_hw_film reads). */
int tante_enc23_frames(const void* h1, int n_img, int frames, int Hp, int Wp, int C, const void* enc_stream, float* out, void* stream);

/* ---- losses / metrics / optimiser step of the harness ------------------------------------------------
 * pred is addressed as pred[b*pb + t*pt + s*ps + c*pc] (so the channels-first rollout buffer needs no permute copy),
 * ref and grad are contiguous channels-last (B, T, HW, C).
 * tante_metric_sums: sums[(b*T + t)*C + c][0..4] = { sum_s (pred-ref)^2, sum_s ref^2, sum_s ref, sum_s (ref-p)^2, sum_s (ref-p) } with
 *   the pivot p = ref[b, t, first pixel, c] -- every metric of trainer/metrics.py (MSE l.53-60, NMSE l.82-98, L2RE l.100-111,
 *   NNMSE l.114-130, VRMSE l.158-164) is a closed form of them; the 'std' normalisations use the shifted moments (entries 3, 4),
 *   which stay well conditioned where sum y^2 - (sum y)^2 / n cancels (torch.std in the reference is a stable two-pass form).
 * tante_mse_grad: grad = scale * (pred - ref), the gradient of  MSE(...).mean()  with scale = 2 / (B*T*HW*C)  (trainer.py:189). */
int tante_metric_sums(const float* pred, int64_t pb, int64_t pt, int64_t ps, int64_t pc, const float* ref, int B, int T, int64_t HW,
                      int C, float* sums, void* stream);
int tante_mse_grad(const float* pred, int64_t pb, int64_t pt, int64_t ps, int64_t pc, const float* ref, int B, int T, int64_t HW,
                   int C, float scale, float* grad, void* stream);
/* out (one double) = sum g[i]^2 over a flat fp32 gradient bucket: the square of clip_grad_norm_'s total norm (trainer.py:193). */
int tante_sumsq(const float* g, int64_t n, double* out, void* stream);
/* Fused clip + AdamW over flat fp32 buckets (params, exp_avg, exp_avg_sq, grads), torch.optim.AdamW semantics
 * (decoupled decay, bias correction with `step` counted from 1).  g is first multiplied by grad_scale (1/world for a
 * summed all-reduce) and, if max_norm > 0, by min(1, max_norm / (sqrt(*sumsq) * grad_scale + 1e-6)) -- computed on the
 * device, so the train step has no host synchronisation. */
int tante_adamw_step(float* p, float* m, float* v, const float* g, int64_t n, const double* sumsq, float max_norm, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);

/* g[i] = clamp(g[i], -clip, clip) over a flat gradient bucket: clip_grad_value_ of the adaptive-dt trainer (r_trainer.py:155) */
int tante_clip_value(float* g, int64_t n, float clip, void* stream);
/* backward of tante_rt_reduce: dt[b][l] = drt[b] / L (the clamp is straight-through in the reference, tante.py:195-198) */
int tante_rt_reduce_bwd(const float* drt, int B, int L, float* dt, void* stream);

/* ---- backward kernels of the train step (trainer/trainer.py:191 loss.backward() through the whole rollout) ----------
 * Data gradients of every dense / conv layer are tante_gemm calls with the TANTE_W_*_T packings (dX = dY . W, with the
 * scatter / gather modes of the forward layer swapped); the rest: */

/* xhat = (x - mean) * rstd per row (no affine: the host folds gamma/beta into the consumer's weight), stats[row] = {mean, rstd};
 * backward: dx = dskip + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = d xhat  (dskip = gradient of the residual branch, may be NULL) */
int tante_layernorm_fwd(const float* x, int64_t M, int C, float eps, void* xhat, int out_dtype, float* stats, void* stream);
int tante_layernorm_bwd(const void* dxhat, int g_dtype, const float* x, const float* stats, const float* dskip, int64_t M, int C,
                        float* dx, void* stream);
/* post = act(pre);  dpre = dpost * act'(pre)   (act codes above) */
int tante_act_fwd(const void* pre, int in_dtype, void* post, int out_dtype, int64_t n, int act, void* stream);
int tante_act_bwd(const void* dpost, int d_dtype, const void* pre, int pre_dtype, void* dpre, int out_dtype, int64_t n, int act,
                  void* stream);
/* out[c] (+)= sum over (o, i) of x[(o*C + c)*inner + i]: bias gradients of channels-last (inner = 1) / channels-first tensors */
int tante_colsum(const void* x, int dtype, int64_t outer, int C, int64_t inner, float* out, int accumulate, void* stream);
/* y = v * a[t] + b[t] + s_emb[hw] over rows r = (b, t, hw) (the FiLM + positional epilogue as an op of its own) and its backward:
 * dv = dy * a[t], da / db (T, C) and ds (HW, C) reduced over the other indices */
int tante_film_pos_fwd(const float* v, const float* a, const float* b, const float* s_emb, int64_t rows, int C, int T, int64_t HW,
                       float* y, void* stream);
/* CViT's encoder (models/cvit.py:291-297): y[(b, hw), t] = v[(b, t), hw] + t_emb[t] + s_emb[hw] -- the two positional sums and the
 * 'b t s d -> (b s) t d' regrouping in one pass.  v (B * T * HW, C) rows (b, t, hw); y (B * HW * T, C) rows (b, hw, t); C % 4 == 0. */
int tante_pos_embed_tmajor(const float* v, const float* t_emb, const float* s_emb, int64_t B, int T, int64_t HW, int C, float* y,
                           void* stream);
int tante_film_pos_bwd(const float* dy, const float* v, const float* a, int64_t BT, int64_t HW, int C, int T, float* dv, float* da,
                       float* db, float* ds, void* stream);
/* The same two with the window given as T <= 8 separate frame tensors (frame t of item b at f[t] + b * bstride[t] floats, rows (hw, c);
 * 16-byte aligned): a BPTT rollout keeps one encoding per frame and a window is any T of them (tante.py:136-141 applied to
 * trainer.py:144-159's sliding window).  bwd writes the frames' gradients to dv[t], contiguous (B, HW, C) each; C = 256. */
typedef struct TanteFrames {
  const float* f[8];
  int64_t bstride[8];
} TanteFrames;
int tante_film_pos_fwd_frames(const TanteFrames* frames, const float* a, const float* b, const float* s_emb, int64_t B, int T, int64_t HW, int C,
                              float* y, void* stream);
/* (round 6) tante_film_pos_fwd_frames + the vertical propagator (tante_axis_mlp_c over (B T, n = Hp, inner = Wp C)) in ONE launch: the
 * propagator's tile load applies a[t] v + b[t] + s_emb[hw] to the cached frames; x (B T, Hp Wp, C) is written, not read.  bf16
 * matrix-pipe propagator shapes with C = 256 only (tante_axis_mlp_film_supported).  attn_backbone.py:140-141 on tante.py:136-141's result. */
int tante_axis_mlp_film_supported(int64_t B, int T, int n, int64_t inner, int C);
int tante_axis_mlp_film(float* x, const TanteFrames* frames, const float* a, const float* b, const float* s_emb, int64_t B, int T, int n,
                        int64_t inner, int C, const float* w1, const float* b1, const float* w2, const float* b2, void* stream);
int tante_film_pos_bwd_frames(const float* dy, const TanteFrames* frames, const float* a, int64_t B, int64_t HW, int C, int T, float* const* dv,
                              float* da, float* db, float* ds, void* stream);
/* The same with accumulation, for a BPTT rollout: bit t of dv_acc_mask -- dv[t] is ADDED to (a frame encoding that sits in several
 * windows receives one gradient per window; the uses add in place instead of autograd summing fresh tensors); acc_flags bit 0: da / db
 * are added to (the FiLM tables are shared by every call of the rollout), bit 1: ds is added to (the parameter's gradient slot). */
int tante_film_pos_bwd_frames_acc(const float* dy, const TanteFrames* frames, const float* a, int64_t B, int64_t HW, int C, int T,
                                  float* const* dv, int dv_acc_mask, float* da, float* db, float* ds, int acc_flags, void* stream);
/* Taylor sum backward: dderivs[k] = sum_i (i dt)^k / k! * dout_i,  dlast (+)= sum_i dout_i (dlast may be the last frame of the
 * input-window gradient, addressed by base pointer + batch stride) */
int tante_taylor_bwd(const float* dout, int64_t dout_bstride, float* const* dderivs, int n_order, double dt, int n_out, float* dlast,
                     int64_t dlast_bstride, int accumulate, int64_t B, int64_t frame, void* stream);
/* dqkv from (qkv, dO) for per-(sequence, head) softmax attention, probabilities recomputed (sequences up to 128 tokens) */
int tante_attention_bwd(const void* qkv, const void* dO, void* dqkv, int dtype, int C, int n_head, const TanteSeq* seq, int causal,
                        float p_drop, uint64_t seed, void* stream);
/* Training-mode attention with dropout p on the softmax probabilities (nn.MultiheadAttention(dropout=p), attn_backbone.py:48).
 * The mask is a pure function of (seed, sequence, head, query, key), so tante_attention_bwd regenerates it from the same seed. */
int tante_attention_dropout(const void* qkv, void* o, int dtype, int C, int n_head, const TanteSeq* seq, int causal, float p_drop,
                            uint64_t seed, void* stream);
/* out = res + dropout(y) and dy = dropout'(dout): self.drop(.) on both residual branches (attn_backbone.py:57,81-82); the mask is
 * a pure function of (seed, element index). */
int tante_dropout_add(const void* y, int y_dtype, const float* res, float p, uint64_t seed, int64_t n, float* out, void* stream);
int tante_dropout_bwd(const float* dout, float p, uint64_t seed, int64_t n, void* dy, int y_dtype, void* stream);
/* axis propagator backward: dx = dy + W1^T (gelu'(pre) * (W2^T dy)); also writes h = gelu(pre) and dpre (same layout as x) for
 * the weight-gradient GEMMs */
int tante_axis_mlp_bwd(const float* x, const float* dy, int64_t outer, int n, int64_t inner, const float* w1, const float* b1,
                       const float* w2, float* dx, float* h, float* dpre, void* stream);
/* weight / bias gradients of one propagator layer from two (outer, n, inner) fp32 tensors in the residual stream's layout:
 * dW[a][j] (+)= sum_{o,i} U[o][a][i] V[o][j][i], db[a] (+)= sum_{o,i} U[o][a][i] (db may be NULL); n <= 64, inner % 16 == 0.
 * (dW2 = <dy, h>, dW1 = <dpre, x> of attn_backbone.py:111-119's two Linear layers.) */
/* LayerNorm's affine folded into the consuming Linear on the train path (attn_backbone.py:50-56 with nn.LayerNorm's weight / bias):
 * fwd: We = W diag(gamma), be = b + W beta (b may be NULL);  bwd, from the accumulated gradients GW (N, K), Gb (N) of (We, be), ADDS
 * dW += GW diag(gamma) + Gb beta^T, db += Gb (db may be NULL), dgamma[k] += sum_n GW[n][k] W[n][k], dbeta[k] += sum_n W[n][k] Gb[n]. */
int tante_fold_fwd(const float* W, const float* b, const float* gamma, const float* beta, int N, int K, float* We, float* be, void* stream);
int tante_fold_bwd(const float* GW, const float* Gb, const float* W, const float* gamma, const float* beta, int N, int K, float* dW, float* db,
                   float* dgamma, float* dbeta, void* stream);
/* The same, and GW / Gb are ZEROED as they are read (K <= 256): the accumulators of a folded pair can then live across train steps without a
 * fill per weight and step. */
int tante_fold_bwd_clear(float* GW, float* Gb, const float* W, const float* gamma, const float* beta, int N, int K, float* dW, float* db,
                         float* dgamma, float* dbeta, void* stream);
/* The propagator's backward WITH its parameter gradients in one launch: bf16 operands on the matrix cores for axis lengths 16 / 32 / 48
 * (inner % 64 == 0), fp32 on the vector units -- tante_axis_mlp_bwd's own expressions -- for 4 (inner % 4 == 0);
 * tante_axis_mlp_bwd_fused_supported says which shapes: dx = dy + W1^T (gelu'(pre) (W2^T dy)) with pre = W1 x + b1, and
 * dW1 += <dpre, x>, db1 += sum dpre, dW2 += <dy, gelu(pre)>, db2 += sum dy (all four ADDED into).  x, dy, dx: fp32 (outer, n, inner), 16-byte
 * aligned.  Replaces tante_axis_mlp_bwd + two tante_axis_wgrad on the bf16 train path (attn_backbone.py:111-119, 140-145). */
int tante_axis_mlp_bwd_fused_supported(int n, int64_t inner);
int tante_axis_mlp_bwd_fused(const float* x, const float* dy, int64_t outer, int n, int64_t inner, const float* w1, const float* b1, const float* w2,
                             float* dx, float* dW1, float* db1, float* dW2, float* db2, void* stream);
/* The same with a caller-owned workspace of tante_axis_wgrad_workspace_bytes() bytes (tante_axis_wgrad_ws's; one call at a time per
 * workspace): workgroups store their partial gradients there and a second small kernel sums them, instead of every workgroup adding its
 * partial atomically (512 same-address atomics per value). */
int tante_axis_mlp_bwd_fused_ws(const float* x, const float* dy, int64_t outer, int n, int64_t inner, const float* w1, const float* b1,
                                const float* w2, float* dx, float* dW1, float* db1, float* dW2, float* db2, void* workspace,
                                int64_t workspace_bytes, void* stream);
/* tante_pack_block_train for n blocks in one launch (weights with the LayerNorm affines folded in; block_stream of tante_block_stream_bytes). */
typedef struct TanteBlockWeights {
  const float* in_w; const float* in_b; const float* out_w; const float* out_b;
  const float* fc1_w; const float* fc1_b; const float* fc2_w; const float* fc2_b;
  void* block_stream;
} TanteBlockWeights;
int tante_pack_block_train_multi(const TanteBlockWeights* blocks, int n, int C, int hidden, void* stream);
/* n folds' forward in one launch (tante_fold_fwd's expressions per entry; b may be NULL) -- attn_backbone.py:50-56. */
typedef struct TanteFoldFwd {
  const float* W; const float* b; const float* gamma; const float* beta;
  float* We; float* be;
  int N, K;
} TanteFoldFwd;
int tante_fold_fwd_multi(const TanteFoldFwd* folds, int n, void* stream);
/* n transposed-fragment streams in one launch: entry e = tante_pack_block_tail_bwd(a, b, c) -> dst (the block-tail stream from
 * (fc2 weight, folded fc1 weight, out-proj weight), the block-head stream from the three 256-row blocks of the folded in-projection). */
typedef struct TanteMat3 {
  const float* a; const float* b; const float* c;
  void* dst;
} TanteMat3;
int tante_pack_block_tail_bwd_multi(const TanteMat3* mats, int n, int C, int hidden, void* stream);
/* n folds' backward in ONE launch (the train step has two per TransformerBlock, each 1 - 3 MB of work that costs 11 us as a launch of its
 * own); clear != 0: as tante_fold_bwd_clear (every K <= 256).  attn_backbone.py:50-56, as above. */
typedef struct TanteFold {
  float* GW; float* Gb;                         /* accumulated gradients of (We, be): (N, K), (N) */
  const float* W; const float* gamma; const float* beta;
  float* dW; float* db; float* dgamma; float* dbeta;     /* gradient slots, added into (db may be NULL) */
  int N, K;
} TanteFold;
int tante_fold_bwd_multi(const TanteFold* folds, int n, int clear, void* stream);
int tante_axis_wgrad(const float* U, const float* V, int64_t outer, int n, int64_t inner, float* dW, float* db, int accumulate, void* stream);
/* The same with a caller-owned workspace of tante_axis_wgrad_workspace_bytes() bytes (16-byte aligned; its LAST 256 bytes -- the arrival
 * counters -- zero on first use): every workgroup stores its partial there and the last of each group of 16 to finish adds the group's sum
 * into dW / db -- a sixteenth of the same-address atomics, no second launch; the counters are back at zero when the kernel ends.  One call
 * at a time per workspace (calls on one stream are).  Null or too small -> every workgroup adds its partial atomically. */
int64_t tante_axis_wgrad_workspace_bytes(void);
int tante_axis_wgrad_ws(const float* U, const float* V, int64_t outer, int n, int64_t inner, float* dW, float* db, int accumulate,
                        void* workspace, int64_t workspace_bytes, void* stream);

/* A gathered row matrix (rows r, `cols` columns): LINEAR rows at (r/n0)*s1 + (r%n0)*s0 + off with element stride es, or the
 * k = s patches of an image batch exactly as in TanteGemm (n0 images per batch item, s1 elements between items). */
typedef struct TanteRowMat {
  const void* p;
  int32_t dtype, mode;
  int64_t s1, s0, off, es;
  int32_t n0;
  int32_t Hin, Win, Cin, P;
} TanteRowMat;

/* dW[i][j] (+)= sum_r U[r][i] * V[r][j], written at the parameter's own index: layout is a TANTE_W_* source layout of
 * tante_pack_weight and (n, k) = (i, j), or (j, i) when swap != 0 (transposed-conv weights: U = input pixels, V = output-gradient
 * patches).  dbias (may be NULL): dbias[i] (+)= sum_r U[r][i], the bias gradient of a Linear / patch-embed layer, computed from
 * the tiles that are staged anyway.  MFMA in `compute`, fp32 atomics across the row split. */
int tante_wgrad(const TanteRowMat* U, const TanteRowMat* V, int64_t R, int I, int J, float* dW, float* dbias, int layout, int P,
                int C_other, int swap, int compute, int accumulate, void* stream);
/* The same with a caller-owned scratch buffer (16-byte aligned; one call at a time per buffer): a gradient that fits ONE output tile (I, J <= 64
 * -- the skinny gradients of the convolution stages) is then cut into many more row ranges whose partials are stored there and summed by a
 * second kernel, instead of a few hundred workgroups adding into the same I x J addresses.  Null: tante_wgrad. */
int tante_wgrad_ws(const TanteRowMat* U, const TanteRowMat* V, int64_t R, int I, int J, float* dW, float* dbias, int layout, int P, int C_other,
                   int swap, int compute, int accumulate, void* workspace, int64_t workspace_bytes, void* stream);
/* The same for n_seg operand pairs of R rows each, contracted as one row range (dW = sum_g U_g^T V_g): the uses of one weight in a
 * back-propagation through time share a launch and one atomic epilogue where the shapes allow (dense bf16 rows, I and J multiples of
 * 128, R % 32 == 0, at most 8 segments per launch); otherwise the segments run one by one. */
int tante_wgrad_multi(const TanteRowMat* U, const TanteRowMat* V, int n_seg, int64_t R, int I, int J, float* dW, float* dbias, int layout,
                      int P, int C_other, int swap, int compute, int accumulate, void* stream);
/* The same with a caller-owned scratch buffer (16-byte aligned; any size, 64 MiB covers every TANTE shape): when it is large enough the
 * split-R partial tiles are STORED there and summed by a second kernel instead of added with fp32 atomics (which run at a fifth of the
 * store rate and limited the launch to half the CUs); too small or null -> the atomic form.  The buffer is scratch for the duration of
 * the call's kernels on `stream` only. */
int tante_wgrad_multi_ws(const TanteRowMat* U, const TanteRowMat* V, int n_seg, int64_t R, int I, int J, float* dW, float* dbias, int layout,
                         int P, int C_other, int swap, int compute, int accumulate, void* workspace, int64_t workspace_bytes, void* stream);

/* Several weights' gradients in one launch: each job is a tante_wgrad_multi_ws call with accumulate = 1 (dW += sum_g U_g^T V_g, dbias
 * likewise).  Up to four jobs of dense bf16 rows (I, J multiples of 128, R % 32 == 0) share the launch's workgroups -- a quarter of the
 * row splits, partial tiles and ramp each (the four weights of a transformer block: trainer step, one launch per block instead of
 * four) -- and one reduce launch; anything else runs job by job. */
typedef struct TanteWgradJob {
  const TanteRowMat* U;
  const TanteRowMat* V;
  int32_t n_seg;
  int64_t R;
  int32_t I, J;
  float* dW;
  float* dbias;
  int32_t layout, P, C_other, swap;
} TanteWgradJob;
int tante_wgrad_jobs_ws(const TanteWgradJob* jobs, int n_jobs, int compute, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- the TRAINING form of the token-local tail of a rollout call (tail_chain.hip; bf16, C = 256, patch_scale 8, D <= 12, Wp % 16 == 0) ----
 * Replaces, on the differentiable path, the per-stage launches of dec_CNN.forward (enc_dec_cnn.py:263-277: three kernel = stride = 2
 * transposed convolutions with erf-GELU between them), the Taylor sum (tante.py:165-171), enc_CNN.forward on the predicted frame
 * (enc_dec_cnn.py:217-229) and the whole backward of that chain (~13 + ~33 launches per rollout call) by ONE forward and ONE backward launch.
 * All "rows" tensors are token-major dense bf16 matrices: a token's 8 x 8 pixel block in hierarchical order, pixel index
 * (kh1, kw1, kh2, kw2[, kh3, kw3]) = the taps of the three stages, so every weight gradient is dW = U^T V of two of them with NO gather:
 *   xl16 (T, 256) | pre1 / act1 (4T, 128) | pre2 / act2 (16T, 64) | f16 (16T, 64: (kh3, kw3, d) at k = sub * D + d, zero beyond 4 D)
 *   pre1e / act1e (16T, 64) | pre2e / act2e (4T, 128) | z (T, 256) fp32             (T = n_img * Hp * Wp tokens, 16 per workgroup)
 * Weights come pre-packed (tante_tail_pack_dec / _enc: conv weights in the reference layouts, (Cin, Cout, 2, 2) / (Cout, Cin, 2, 2)) into
 * buffers of tante_tail_stream_bytes(0 dec fwd | 1 dec bwd | 2 enc fwd | 3 enc bwd) bytes (4: see TanteTailBwd.bias_ws).  Token rows of the residual stream are
 * addressed like tante_head_fused: row r at (r / a_n0) * a_s1 + (r % a_n0) * a_s0 + a_off (elements; a_n0 % 16 == 0).
 * n_ord = 0 is the encoder alone on given frames (`base`; `out` may be NULL): the initial window of a rollout -- enc_CNN.forward with its
 * saves, and in tante_tail_bwd (dbase NULL) the two wide encoder stages backwards for the weight-gradient operands. */
#define TANTE_TAIL_MAX_ORD 3
typedef struct TanteTailOrdF {
  const float* x;        /* this order's residual stream */
  const void* w;         /* decoder forward stream */
  float coef;            /* Taylor coefficient (i dt)^k / k! */
  void *xl16, *pre1, *act1, *pre2, *act2;
} TanteTailOrdF;
typedef struct TanteTailFwd {
  TanteTailOrdF o[TANTE_TAIL_MAX_ORD];
  int32_t n_ord;
  int64_t a_s1, a_s0, a_off;
  int32_t a_n0;
  int32_t n_img, Hp, Wp, D;
  const float* base;     /* the window's last frame (D, H, W) per image */
  int64_t base_bstride;
  float* out;            /* the predicted frame */
  int64_t out_bstride;
  const void* we;        /* encoder forward stream, or NULL: no re-encoding (the rollout's last call) */
  void *f16, *pre1e, *act1e, *pre2e, *act2e;
  float* z;
} TanteTailFwd;
typedef struct TanteTailOrdB {
  const void* w;         /* decoder backward stream */
  float coef;
  const void *pre1, *pre2;
  void *dpre1, *dpre2, *dder;      /* (T, 512) | (4T, 256) | (16T, 64): the V operands of the three decoder weight gradients */
  float* dx;             /* gradient of this order's residual stream (rows addressed by a_*) */
  float *db1, *db2, *db3;          /* bias gradients, ADDED (through bias_ws); may be NULL */
  const void* act2;      /* with dw3 and bias_ws: the last stage's weight gradient (64, D, 2, 2) is computed inside the launch (per-workgroup */
  float* dw3;            /*   partials in bias_ws, ADDED by the reduce launch) and dder may be NULL */
} TanteTailOrdB;
typedef struct TanteTailBwd {
  TanteTailOrdB o[TANTE_TAIL_MAX_ORD];
  int32_t n_ord;
  int64_t a_s1, a_s0, a_off;
  int32_t a_n0;
  int32_t n_img, Hp, Wp, D;
  const float* dext;     /* gradient reaching the predicted frame from outside the chain (loss, the next call's Taylor base), or NULL */
  int64_t dext_bstride;
  float* dbase;          /* out: gradient of the window's last frame (= the frame's total gradient), or NULL */
  int64_t dbase_bstride;
  const float* dz;       /* gradient of the frame's encoding (T, 256), or NULL: the encoder part is skipped */
  const void* we;        /* encoder backward stream */
  const void *pre1e, *pre2e;
  void *dz16, *dpre2e, *dpre1e;    /* (T, 256) | (4T, 128) | (16T, 64): the U operands of the three encoder weight gradients */
  float* bias_ws;        /* scratch, (T / 16) * (n_ord + 1) * tante_tail_stream_bytes(4) bytes: per-workgroup partial sums of the decoder bias
                            gradients and of the two pixel-level weight gradients, summed into their destinations by a second small launch;
                            NULL: none of them */
  const void* f16;       /* with dwe1 (+ dbe1) and bias_ws: the first encoder stage's weight (64, D, 2, 2) and bias gradients are computed */
  float *dwe1, *dbe1;    /*   inside the launch (ADDED) and dpre1e may be NULL */
} TanteTailBwd;
int tante_tail_supported(int C, int D, int Hp, int Wp);
int64_t tante_tail_stream_bytes(int which);
int tante_tail_pack_dec(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, int D,
                        void* fwd_stream, void* bwd_stream, void* stream);
int tante_tail_pack_enc(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, int D,
                        void* fwd_stream, void* bwd_stream, void* stream);
int tante_tail_fwd(const TanteTailFwd* args, void* stream);
int tante_tail_bwd(const TanteTailBwd* args, void* stream);

const char* tante_last_error(void);
int tante_abi_version(void);

/* Launch-heuristic overrides (workgroup counts, kernel-variant choices).  Every option switches between forms that compute the same
 * function; none changes a result beyond the summation order the variant implies.  Names start with "TANTE_" (e.g. "TANTE_FS_GROUPS":
 * 0 = automatic, 1 = unpaired, 2 = paired form of the fused block kernel; "TANTE_FS_WAVES" = 8 forces its 8-wave form;
 * "TANTE_WGRAD_WGS", "TANTE_GEMM_WGS", ... = grid sizes).  Process-wide, thread-safe; an option that was never set keeps the built-in
 * default.  There is no counterpart in the reference (it has no native code to tune). */
int tante_set_option(const char* name, int value);
int tante_get_option(const char* name, int dflt);

#ifdef __cplusplus
}
#endif
#endif /* TANTE_HIP_H */
