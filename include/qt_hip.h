/*
 * qt_hip.h -- C ABI of libqt_hip.so, the MI355X (gfx950) fake-quantization engine.
 *
 * This is the drop-in boundary for the hot path of jeffreyyu0602/quantized-training.
 * The reference has no FFI of its own: its operator boundary is the torch.library
 * namespace `quantized_ops` (src/quantized_training/decomposed.py:16) plus the autograd
 * function FusedAmaxObsFakeQuantFunction (src/quantized_training/fake_quantize.py:197-252).
 * Each entry point below names the reference interface it replaces (file:line relative
 * to the upstream checkout).  INTEGRATION.md shows the ctypes binding a maintainer of the
 * reference would add to call them from those Python impls.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / C++ types cross the boundary
 *   - every function returns 0 (QT_OK) or a negative qt_status / positive hipError_t;
 *     nothing throws, nothing allocates, no ownership is transferred
 *   - `*_dev` pointers are device (HBM) addresses owned by the caller; `stream` is a
 *     hipStream_t passed as void* (NULL = the legacy default stream); launches are async
 *   - bf16 / fp16 tensors travel as uint16_t bit patterns
 *   - stateless and thread-safe (the only state is the caller's buffers)
 */
#ifndef QT_HIP_H
#define QT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 6): qt_fp8_gemm's last argument is the heuristic suggestion index (`algo`), no longer a 0 / 1 `tune` flag; the training
 * entry points (qt_attention_train_*, qt_grad_fanin_bf16, qt_embedding_backward_bf16, qt_fake_quant_chain_bf16) exist.  A caller built
 * against 1 must not bind this library, and the package refuses a library that reports anything else.
 * 3 (round 6): qt_train_gemm_backward_bf16 and the step-end entry points (qt_clip_adamw_plan / _ws_bytes / _bf16) exist. */
#define QT_ABI_VERSION 3
#define QT_MAP_ENTRIES 65536

typedef enum qt_status {
    QT_OK = 0,
    QT_ERR_BAD_DTYPE = -1,   /* reference: ValueError("Unsupported dtype") fake_quantize.py:95 */
    QT_ERR_BAD_ARG = -2,
    QT_ERR_UNALIGNED = -3,
    QT_ERR_NO_DEVICE = -4
} qt_status;

/* Closed-form rounding selector for kernels that do not need the LDS table.
 * kind QT_FMT_LUT: use the 65 536-entry table (any dtype).  The other kinds are proven equal
 * to the table on all 65 536 inputs (tests/test_capi_host.py) and skip the LDS staging. */
typedef enum qt_fmt_kind {
    QT_FMT_LUT = 0,
    QT_FMT_IDENTITY = 1,     /* dtype None / bfloat16 / float32        fake_quantize.py:34-40 */
    QT_FMT_FP_SAT = 2,       /* e4m3 / e5m2: p0 = mantissa bits, p1 = min normal exponent,
                                fmax = saturation value                fp8.py:10-67          */
    QT_FMT_INT = 3           /* intN / uintN: flo, fhi = clamp bounds   fake_quantize.py:43-52 */
} qt_fmt_kind;

typedef struct qt_format {
    int32_t kind;
    int32_t p0;
    int32_t p1;              /* kind QT_FMT_LUT: bit 0 set = the 512 x 4 row words of qt_build_rowparams FOLLOW the 65 536 map entries
                                in lut_dev (bit 1: rows by sign and exponent, bit 2: results take the input's sign, bit 3: zero
                                results of negative inputs are -0, bit 4: the input -0.0 is an exception to that and maps to the
                                value whose bits are in fhi); large aligned tensors then take the row form -- 8 vector
                                instructions and a 16-byte LDS read per element instead of a gather from a 128 KiB table */
    float flo;
    float fhi;
} qt_format;

int qt_abi_version(void);
const char *qt_status_string(int code);

/* *id = the identifier of the stream capture `stream` is recording into, 0 when it is not capturing.  The host side keys scratch that
 * launches share (the split-K workspace of qt_linear_fqt_ws_bf16, which must be used by ONE ordered sequence
 * of launches) by (device, stream, capture): every captured graph and every eager stream gets its own.  No reference counterpart:
 * the reference has no native scratch (modules/qat/linear.py:40-41 allocates through torch). */
int qt_stream_capture_id(void *stream, unsigned long long *id);

/* ---- A1: value-map builder (host) ------------------------------------------------------
 * Replaces get_quantization_map(dtype)                       fake_quantize.py:31-95
 * (and through it quantize_to_fp8_e4m3/_e5m2 fp8.py:10-67, _quantize_elemwise_core in bf16
 * arithmetic fp8.py:147-203, quantize_to_posit posit.py:6-67).
 * out_host[i] = bf16 bits of Q_dtype(bf16_from_bits(i)); NaN entries are 0x7FC0.
 * dtype NULL or "" = identity.  Unknown dtype -> QT_ERR_BAD_DTYPE. */
int qt_build_map(const char *dtype, uint16_t *out_host);

/* Closed-form descriptor for `dtype` (kind QT_FMT_LUT when there is none). */
int qt_format_for(const char *dtype, qt_format *out);

/* Host evaluation of a closed-form descriptor on one bf16 pattern (used by the CPU tests to
 * prove descriptor == table on all 65 536 inputs). */
uint16_t qt_format_apply_host(const qt_format *fmt, uint16_t bf16_bits);

/* ---- exported rounding functions on fp32 data (host) -----------------------------------
 * quantize_to_fp8_e4m3 / _e5m2 (fp8.py:10-67) and quantize_to_posit (posit.py:6-67) called
 * directly on tensors (package exports, __init__.py:54-56).  n elements, in-place allowed. */
int qt_round_fp8_host(const float *x, float *y, size_t n, int mbits, float fp8_max, float fp8_min);
int qt_round_posit_host(const float *x, float *y, size_t n, int nbits, int es);
/* Same on device buffers. */
int qt_round_fp8_f32(const float *x_dev, float *y_dev, size_t n, int mbits, float fp8_max, float fp8_min,
                     void *stream);
int qt_round_posit_f32(const float *x_dev, float *y_dev, size_t n, int nbits, int es, void *stream);

/* quantize_to_posit(input, nbits, es, round_to_even, return_pbits) with its two options (posit.py:6-67): round_to_even = 0
 * keeps values below the smallest representable step at minpos instead of flushing them (posit.py:50-53); pbits (nullable,
 * int32 per element) receives the posit bit pattern times the sign of the input (posit.py:60-65).  The pattern is the
 * mathematically intended one where upstream's int32 arithmetic overflows (2 + run + es + 23 > 33) and, for regime-dominated
 * inputs (whose pattern upstream leaves to platform-dependent shift counts), that of the value the input is clamped to. */
int qt_posit_quantize_host(const float *x, float *y, int32_t *pbits, size_t n, int nbits, int es, int round_to_even);
int qt_posit_quantize_f32(const float *x_dev, float *y_dev, int32_t *pbits_dev, size_t n, int nbits, int es, int round_to_even,
                          void *stream);

/* ---- A5: quantized_ops::vmap(Tensor self, Tensor other) -> Tensor   decomposed.py:143-163
 * y[i] = lut[idx(x[i])], idx = bf16 bits, or hi16(f32 bits) | (lo16 != 0) for fp32 / fp16
 * (fp16 goes through its fp32 image; the looked-up bf16 value is cast to the output dtype). */
int qt_vmap_bf16(const uint16_t *x_dev, uint16_t *y_dev, size_t n, const qt_format *fmt,
                 const uint16_t *lut_dev, void *stream);
int qt_vmap_f32(const float *x_dev, float *y_dev, size_t n, const qt_format *fmt,
                const uint16_t *lut_dev, void *stream);
int qt_vmap_f16(const uint16_t *x_dev, uint16_t *y_dev, size_t n, const qt_format *fmt,
                const uint16_t *lut_dev, void *stream);

/* ---- A6: quantized_ops::quantize / ::dequantize with a per-tensor scale
 *                                                        decomposed.py:166-210, :213-262
 * quantize:   y = vmap(x / s [+ zp], lut)
 * dequantize: y = vmap?((vmap?(x, in_lut) [- zp]) * s, out_lut)
 * scale_dev / zp_dev point at ONE element of the tensor's own dtype (bf16 bits or float).
 * zp_dev, in_lut_dev, out_lut_dev may be NULL. */
int qt_quantize_bf16(const uint16_t *x_dev, uint16_t *y_dev, size_t n, const qt_format *fmt,
                     const uint16_t *lut_dev, const uint16_t *scale_dev, const uint16_t *zp_dev, void *stream);
int qt_quantize_f32(const float *x_dev, float *y_dev, size_t n, const qt_format *fmt,
                    const uint16_t *lut_dev, const float *scale_dev, const float *zp_dev, void *stream);
int qt_dequantize_bf16(const uint16_t *x_dev, uint16_t *y_dev, size_t n, const uint16_t *scale_dev,
                       const uint16_t *zp_dev, const uint16_t *in_lut_dev, const uint16_t *out_lut_dev,
                       void *stream);
int qt_dequantize_f32(const float *x_dev, float *y_dev, size_t n, const float *scale_dev, const float *zp_dev,
                      const uint16_t *in_lut_dev, const uint16_t *out_lut_dev, void *stream);

/* ---- A7: FusedAmaxObsFakeQuantFunction.forward              fake_quantize.py:197-248
 * Split into the two device steps of one call (no host synchronisation anywhere):
 *
 * qt_scale_update  (fake_quantize.py:230-242): for each channel c < C
 *     amax = max_l history[l][c];  history <- roll(history, -1, 0) (when L > 1);
 *     history[0][c] <- 0 (the fused pass below max-accumulates the current amax into it);
 *     sf = amax / quant_max, kept at the old scale when amax <= 0 or non-finite,
 *     optionally 2^ceil(log2 sf);  scale[c] <- sf.
 * qt_fake_quant_*  (fake_quantize.py:218-223 + :245-246): ONE read of x that
 *     (a) max-accumulates |x| into amax_bits_dev (uint32 view of history[0]; NULL = observer off)
 *     (b) writes y = vmap(x / s, lut) * s with s = (input dtype)scale_f32_dev[0]
 *         (y_dev NULL = observe only).
 * Per-channel variants view x as [outer][C][inner] and use scale[c] / amax_bits[c]. */
int qt_scale_update(float *history_dev, int L, int C, float *scale_dev, float quant_max, int force_pow2,
                    void *stream);
/* qt_scale_update for `count` fake-quantizers in one launch: element i of each device array describes fake-quantizer i
 * (history_ptrs_dev[i] -> its [L_i, C_i] history, scale_ptrs_dev[i] -> its C_i scales).  A captured training step runs
 * this once up front instead of one small launch before every fake-quant pass; valid because the scale a call applies
 * depends only on the amaxes of earlier calls. */
int qt_scale_update_multi(float *const *history_ptrs_dev, const int *L_dev, const int *C_dev, float *const *scale_ptrs_dev,
                          const float *quant_max_dev, const int *force_pow2_dev, int count, void *stream);
int qt_fake_quant_bf16(const uint16_t *x_dev, uint16_t *y_dev, size_t n, const qt_format *fmt,
                       const uint16_t *lut_dev, const float *scale_f32_dev, uint32_t *amax_bits_dev,
                       void *stream);
int qt_fake_quant_f32(const float *x_dev, float *y_dev, size_t n, const qt_format *fmt,
                      const uint16_t *lut_dev, const float *scale_f32_dev, uint32_t *amax_bits_dev,
                      void *stream);
/* Same pass for e4m3 / e5m2 that ALSO (or only) emits the quantized code q = vmap(x / s) as one OCP
 * FP8 byte per element (y8_dev; exact, every e4m3/e5m2 value is an FP8 value), for FP8-MFMA GEMMs:
 * (q, s) is what the reference's converted graphs feed the GEMM (quantize -> GEMM ->
 * dequantize(s_x * s_w), quantize_pt2e.py:323-446).  y_dev (bf16 fake-quantized tensor) may be NULL.
 * n must be a multiple of 16; fmt must be the e4m3 or e5m2 descriptor of qt_format_for. */
int qt_fake_quant_bf16_fp8(const uint16_t *x_dev, uint16_t *y_dev, uint8_t *y8_dev, size_t n,
                           const qt_format *fmt, const float *scale_f32_dev, uint32_t *amax_bits_dev,
                           void *stream);
/* Same pass reading a permuted view: x is a logical [d0, d1, d2, inner] bf16 tensor with element strides
 * (s0, s1, s2, 1) -- e.g. attention's q / k / v, [B, S, H, D] storage viewed as [B, H, S, D] -- and y is
 * written contiguous, which is the layout the reference's vmap returns (decomposed.py:155).  Saves the
 * separate .contiguous() copy.  inner and the strides must be multiples of 8. */
/* FP8-only pass (stateless E4M3 / E5M2, unit scale) over up to four bf16 tensors in ONE launch, their codes written
 * back to back into y8 (tensor i at element offset ns[0] + ... + ns[i-1]): the weight passes of sibling Linears that
 * share an input (q / k / v projections), feeding one concatenated FP8 GEMM.  xs / ns are host arrays. */
int qt_fake_quant_bf16_fp8_multi(const uint16_t *const *xs_dev, const size_t *ns, int count, uint8_t *y8_dev,
                                 const qt_format *fmt, void *stream);

/* Many per-tensor fake-quant calls of ONE format as one launch: item i is `qt_fake_quant_bf16(x, y, 8 * nvec, fmt, lut, scale, amax)`
 * with its own scale and amax slot (fake_quantize.py:230-246 per tensor, unchanged).  For the weight fake-quantizers of a training
 * step (modules/qat/linear.py:40-41 issues one per Linear and forward; run_glue_no_trainer.py:647-667): the weights do not change
 * between the start of a step and its optimizer update, so the caller may run all of them first.  items_dev: DEVICE array of
 * `count` items, tensors 16-byte aligned bf16 with 8 | n; `first_tile` = sum over earlier items of ceil(nvec / 1024) and
 * total_tiles the sum over all (the launch geometry, computed by the caller once); amax_bits NULL = not observed.  Formats: intN,
 * e4m3 / e5m2 closed forms, and table formats whose map carries the row form (qt_format.p1 bit 0). */
typedef struct qt_fq_item {
    const void *x_bf16;
    void *y_bf16;
    const float *scale_f32;
    uint32_t *amax_bits;
    unsigned long long nvec;
    unsigned long long first_tile;
} qt_fq_item;
int qt_fake_quant_multi_bf16(const qt_fq_item *items_dev, int count, unsigned long long total_tiles, const qt_format *fmt,
                             const uint16_t *lut_dev, void *stream);

/* The FP8-codes-only weight pass (item 9 of INTEGRATION.md: `qt_fake_quant_bf16_fp8` with y = NULL, stateless E4M3 / E5M2 at unit scale)
 * of MANY tensors as one launch: every `weight_fake_quant(W)` of an evaluation forward whose GEMMs take the weight-pass + library-GEMM
 * route (modules/qat/linear.py:40-41 issues one per Linear; the weights are constants of the forward).  items_dev: DEVICE array; tensors
 * 16-byte aligned bf16 with 16 | n; `first_tile` = sum over earlier items of ceil(npair / 1024), total_tiles the sum over all. */
typedef struct qt_fq8_item {
    const void *x_bf16;
    void *y8;
    unsigned long long npair;     /* n / 16 */
    unsigned long long first_tile;
} qt_fq8_item;
int qt_fake_quant_multi_bf16_fp8(const qt_fq8_item *items_dev, int count, unsigned long long total_tiles, const qt_format *fmt, void *stream);

/* out[c] = bf16(sum_r x[r][c]) for a contiguous bf16 [rows][cols] matrix, fp32 sums in a fixed order (deterministic): the bias
 * gradient of nnqat.Linear's backward, grad_output.sum(0) of the fake-quantized gradient (modules/qat/linear.py:40-41 through
 * autograd; quantize.py:116-179 quantizes grad_output first).  cols % 8 == 0, x_dev 16-byte aligned. */
int qt_colsum_bf16(const uint16_t *x_dev, uint16_t *out_dev, long rows, long cols, void *stream);
int qt_fake_quant_rows_bf16(const uint16_t *x_dev, uint16_t *y_dev, long d0, long d1, long d2, long inner,
                            long s0, long s1, long s2, const qt_format *fmt, const uint16_t *lut_dev,
                            const float *scale_f32_dev, uint32_t *amax_bits_dev, void *stream);
/* Same layout change with a stateless E4M3 / E5M2 fake-quantizer (unit scale, no observer) that also emits the FP8
 * code of every element (y8 contiguous like y): one pass replaces ".contiguous()" + the consumer Linear's input pass.  y_dev may be
 * NULL when only the codes are wanted (the value operand of P.V as an FP8 GEMM). */
int qt_fake_quant_rows_bf16_fp8(const uint16_t *x_dev, uint16_t *y_dev, uint8_t *y8_dev, long d0, long d1, long d2,
                                long inner, long s0, long s1, long s2, const qt_format *fmt, void *stream);
int qt_fake_quant_pc_bf16(const uint16_t *x_dev, uint16_t *y_dev, size_t outer, size_t C, size_t inner,
                          const qt_format *fmt, const uint16_t *lut_dev, const float *scale_f32_dev,
                          uint32_t *amax_bits_dev, void *stream);
int qt_fake_quant_pc_f32(const float *x_dev, float *y_dev, size_t outer, size_t C, size_t inner,
                         const qt_format *fmt, const uint16_t *lut_dev, const float *scale_f32_dev,
                         uint32_t *amax_bits_dev, void *stream);

/* ---- block-scaled formats: MXFakeQuantFunction.forward            fake_quantize.py:98-133
 * (= calculate_mx_qparam + quantize + multiply, decomposed.py:365-448) for blocks of `block_size`
 * consecutive elements of the last axis of a contiguous [rows, cols] tensor:
 *   s = amax(block) / quant_max; s = scale_lut ? vmap(s, scale_lut) : s; s = s > 0 ? s : 1;
 *   y = vmap(x / s, lut) * s          (all in the tensor's dtype)
 * sf_dev receives the rows * cols / block_size block scales in the tensor's dtype.
 * block_size: power of two, 8..512 (bf16) / 4..256 (fp32); cols % block_size == 0. */
int qt_fake_quant_mx_bf16(const uint16_t *x_dev, uint16_t *y_dev, uint16_t *sf_dev, size_t rows, size_t cols,
                          int block_size, const qt_format *fmt, const uint16_t *lut_dev, float quant_max,
                          const uint16_t *scale_lut_dev, void *stream);
int qt_fake_quant_mx_f32(const float *x_dev, float *y_dev, float *sf_dev, size_t rows, size_t cols,
                         int block_size, const qt_format *fmt, const uint16_t *lut_dev, float quant_max,
                         const uint16_t *scale_lut_dev, void *stream);

/* ---- A9/A10: fake-quant GEMMs (bf16 in, fp32 accumulate on MFMA, bf16 out) ---------------
 * qt_linear_fq_bf16 replaces  F.linear(fq_a(x), weight_fake_quant(W), b)
 *     modules/qat/linear.py:40-41 + the activation pre-hook quantize.py:128-140
 *     x [M,K] row-major, W [N,K] row-major (nn.Linear layout), bias [N] or NULL, y [M,N].
 * qt_bmm_fq_bf16 replaces  MatmulFunctional.forward(fq(a), fq(b))
 *     modules/quantizable/functional_modules.py:22-26 (QK^T and AV)
 *     a [B,M,K] (row stride lda, batch stride sa), b is [B,K,N] addressed with strides
 *     (ldb_k, ldb_n, sb) so that K^T needs no copy; y [B,M,N] contiguous.
 * Operand quantization is applied while the tile is loaded (x/s -> table -> *s, each rounded to
 * bf16 exactly like the elementwise pass); fmt kind QT_FMT_IDENTITY = operand not quantized.
 * The amax observers of both operands are accumulated from the tiles as they stream
 * (amax_*_bits_dev NULL = off). */
typedef struct qt_operand_q {
    qt_format fmt;
    const uint16_t *lut_dev;       /* may be NULL unless fmt.kind == QT_FMT_LUT */
    const float *scale_f32_dev;    /* NULL = scale 1 */
    uint32_t *amax_bits_dev;       /* NULL = observer off */
} qt_operand_q;

int qt_linear_fq_bf16(const uint16_t *x_dev, const uint16_t *w_dev, const uint16_t *bias_dev, uint16_t *y_dev,
                      int M, int N, int K, const qt_operand_q *qx, const qt_operand_q *qw, void *stream);
int qt_bmm_fq_bf16(const uint16_t *a_dev, const uint16_t *b_dev, uint16_t *y_dev, int B, int M, int N, int K,
                   long lda, long sa, long ldb_k, long ldb_n, long sb, const qt_operand_q *qa,
                   const qt_operand_q *qb, void *stream);

/* ---- A7 / A11, several fake-quantizer calls of a training step over ONE tensor as one launch --------------------------------
 * The reference's hooks (quantize.py:116-179: activation_pre_process, error_pre_process, error_post_process) call fake-quantizers
 * back to back on the same bf16 tensor x [rows][cols]: stage i reads x (src = -1) or the RESULT of stage src < i, applies the
 * format `fmt` (one format for all stages: QT_FMT_INT, QT_FMT_FP_SAT, or a table format whose device map carries the row form,
 * p1 bit 0) with ITS scale (scale_f32_dev, NULL = 1: y = fq(x / s) * s, fake_quantize.py:243-246) and accumulates the amax of ITS
 * input into amax_bits_dev (NULL: not observed; the slot was zeroed by qt_scale_update, fake_quantize.py:230-242) -- i.e. every
 * stage is exactly one qt_fake_quant_bf16 call, bit for bit.  out_dev (nullable, 16-byte aligned): where the stage's result goes.
 * colsum_stage >= 0: also colsum_out_dev[c] = bf16(sum over rows of that stage's result[., c]) (grad_bias = grad_output.sum(0) of
 * the Linear behind the quantizer, run_glue_no_trainer.py:660-667): fp32 sums in a fixed order inside a workgroup, 64-bit
 * fixed-point (one unit = 2^-42 of the largest value the stage can produce, colsum_max x scale) across workgroups -- run-to-run
 * bit-identical.  colsum_max: the largest finite magnitude of the format's value map.  ws_dev: qt_fake_quant_chain_ws_bytes(rows,
 * cols) bytes, ZERO before the first launch, left zero by every launch, used by one ordered sequence of launches only.
 * `stages` is a HOST array.  cols % 8 == 0, nstage <= 4. */
typedef struct {
    const float *scale_f32_dev;
    uint32_t *amax_bits_dev;
    uint16_t *out_dev;
    int src;
} qt_chain_stage;
int qt_fake_quant_chain_bf16(const uint16_t *x_dev, long rows, long cols, const qt_chain_stage *stages, int nstage, const qt_format *fmt,
                             const uint16_t *lut_dev, int colsum_stage, float colsum_max, uint16_t *colsum_out_dev, void *ws_dev,
                             size_t ws_bytes, void *stream);
size_t qt_fake_quant_chain_ws_bytes(long rows, long cols);

/* ---- the gradients that meet at one tensor of a training step, added in one launch (round 5) ----------------------------------------
 * A LayerNorm's output feeds several consumers; in the backward pass every consuming Linear's grad_input passes the Linear's backward
 * quantizer (quantize.py:147-148: register_full_backward_hook, `--quantize_backprop ...,residual`) and the autograd engine adds the
 * arrivals one by one.  sum = (((first + y_0) + y_1) + ...), y_i = fq_i(x_i) (items[i].fq != 0: exactly qt_fake_quant_bf16 with that
 * scale and amax slot; y_i also written to out_dev when given) or x_i itself; every addition is torch's bf16 add (fp32, one rounding),
 * taken in the order given.  count <= 4, n % 8 == 0, pointers 16-byte aligned; the quantized items share `fmt`. */
typedef struct {
    const uint16_t *x_dev;
    int fq;
    const float *scale_f32_dev;
    uint32_t *amax_bits_dev;
    uint16_t *out_dev;
} qt_fanin_item;
int qt_grad_fanin_bf16(const uint16_t *first_dev, const qt_fanin_item *items, int count, uint16_t *sum_dev, size_t n, const qt_format *fmt,
                       const uint16_t *lut_dev, void *stream);

/* ---- nn.Embedding's weight gradient inside a training step (round 5; outside the reference package, inside the measured step) ------
 * Bit for bit torch's embedding_dense_backward for <= 3072 indices (embedding_backward_feature_kernel: per 16-row chunk the fp32 sum of
 * the rows of an index in row order, rounded to bf16 and added to the table row in bf16, chunks in order; rows naming padding_idx
 * skipped) as two launches instead of one workgroup's walk over all chunks.  grad [n][cols] bf16, ids [n] int64, partials_dev
 * [n][cols] bf16 scratch, grad_weight_dev [num_rows][cols] bf16 ZERO-FILLED by the caller (torch allocates it with at::zeros);
 * padding_idx < 0: none; no bounds check of the indices (torch has none either).  cols % 8 == 0, n <= 3072. */
int qt_embedding_backward_bf16(const uint16_t *grad_dev, const long *ids_dev, long n, long cols, long padding_idx, long num_rows,
                               uint16_t *partials_dev, uint16_t *grad_weight_dev, void *stream);

/* ---- the model's own elementwise kernels of a TRAINING step, with the fake-quantizer calls that follow them (round 5) -------------
 * The reference's examples run HF's blocks under autograd (run_glue_no_trainer.py:647-667) with the hooks of quantize.py:116-179
 * around every GEMM: a LayerNorm / GELU / softmax kernel of torch's, then one fake-quantizer launch per hook.  These entry points
 * produce the same tensors -- torch's arithmetic and rounding points, forward and backward -- and evaluate the stages of a chain
 * (qt_chain_stage: each exactly one qt_fake_quant_bf16 call on the value the kernel holds in registers) in the same launch.
 *
 * qt_gelu_chain_bf16: y = bf16(erf-GELU(x)) (BertIntermediate's activation), stages on y.
 * qt_gelu_backward_chain_bf16: grad_in = bf16(grad_out * (Phi(x) + x phi(x))) (torch's GeluBackward), stages on grad_in, optional
 *   column sums of one stage (see qt_fake_quant_chain_bf16).
 * qt_layernorm_train_bf16: y = LayerNorm(x) over the last dimension (fp32 statistics, biased variance, one rounding), mean / rstd
 *   [rows] fp32 kept for the backward, stages on y.  cols <= 1024.  residual_dev / sum_dev (both or neither): the residual add in
 *   front of the LayerNorm (modeling_bert.py:188, 212 upstream) in the same launch -- sum = bf16(x + residual) (torch's add) is
 *   written to sum_dev (what the backward needs as the LayerNorm's input) and normalised.
 * qt_layernorm_train_backward_bf16: grad_in = rstd (g - mean(g) - xhat mean(g xhat)), g = grad_out * weight (torch's
 *   layer_norm_grad_input), stages on grad_in; grad_weight = sum_rows grad_out * xhat, grad_bias = sum_rows grad_out and, with
 *   colsum_stage >= 0, the column sums of that stage's result, each as fp32 partial sums per workgroup (part_dev:
 *   qt_layernorm_train_backward_groups(rows) x 3 x cols floats, caller-owned scratch) added in workgroup order by a second small
 *   launch: deterministic.  fan_items (nullable, fan_count <= 3): further gradients that arrive for the LayerNorm's result --
 *   grad_out is replaced by qt_grad_fanin_bf16(grad_out, fan_items) as it is loaded (same sum, same bits, no launch of its own).
 * qt_softmax_fq_probs_bf16: qt_softmax_fq_bf16 that also writes the unquantized probabilities (probs_dev, nullable).
 * qt_softmax_backward_chain_bf16: grad_scores = bf16(bf16((dP - sum dP P) P) * scaling) (torch's _softmax_backward_data, then the
 *   scaling's backward; the additive mask's backward is the identity), stages on grad_scores (qk_matmul's backward-pre quantizer).
 * Every `stages` is a HOST array; nstage >= 1; pointers 16-byte aligned; cols % 8 == 0. */
int qt_gelu_chain_bf16(const uint16_t *x_dev, uint16_t *y_dev, long rows, long cols, const qt_chain_stage *stages, int nstage,
                       const qt_format *fmt, const uint16_t *lut_dev, void *stream);
int qt_gelu_backward_chain_bf16(const uint16_t *grad_out_dev, const uint16_t *x_dev, uint16_t *grad_in_dev, long rows, long cols,
                                const qt_chain_stage *stages, int nstage, const qt_format *fmt, const uint16_t *lut_dev, int colsum_stage,
                                float colsum_max, uint16_t *colsum_out_dev, void *ws_dev, size_t ws_bytes, void *stream);
int qt_layernorm_train_bf16(const uint16_t *x_dev, const uint16_t *weight_dev, const uint16_t *bias_dev, uint16_t *y_dev, float *mean_dev,
                            float *rstd_dev, long rows, long cols, float eps, const qt_chain_stage *stages, int nstage, const qt_format *fmt,
                            const uint16_t *lut_dev, const uint16_t *residual_dev, uint16_t *sum_dev, void *stream);
long qt_layernorm_train_backward_groups(long rows);
int qt_layernorm_train_backward_bf16(const uint16_t *grad_out_dev, const uint16_t *x_dev, const uint16_t *weight_dev, const float *mean_dev,
                                     const float *rstd_dev, uint16_t *grad_in_dev, long rows, long cols, const qt_chain_stage *stages, int nstage,
                                     const qt_format *fmt, const uint16_t *lut_dev, int colsum_stage, float *part_dev, size_t part_bytes,
                                     uint16_t *grad_weight_dev, uint16_t *grad_bias_dev, uint16_t *colsum_out_dev, const qt_fanin_item *fan_items,
                                     int fan_count, void *stream);
int qt_softmax_fq_probs_bf16(const uint16_t *scores_dev, const uint16_t *mask_dev, uint16_t *out_dev, uint16_t *probs_dev, long batch, int heads,
                             int q_len, long cols, long mask_sb, long mask_sh, long mask_sq, float scaling, const qt_format *fmt,
                             const uint16_t *lut_dev, const float *scale_f32_dev, uint32_t *amax_bits_dev, void *stream);
int qt_softmax_backward_chain_bf16(const uint16_t *grad_probs_dev, const uint16_t *probs_dev, uint16_t *grad_scores_dev, long rows, long cols,
                                   float scaling, const qt_chain_stage *stages, int nstage, const qt_format *fmt, const uint16_t *lut_dev,
                                   void *stream);

/* ---- the attention core of a TRAINING step, one launch forward and one backward (round 5) ------------------------------------------
 * Replaces everything between the query / key / value projections and the output projection of the reference's quantizable
 * attention block (modules/quantizable/modeling_bert.py:118-158, functional_modules.py:22-26, hooks: quantize.py:116-179) for
 * head_dim 64 and 32..128 positions (a multiple of 32) -- qt_attention_train_supported says whether a shape is covered:
 *   forward   q' = fq0(q), k' = fq1(k), v' = fq2(v);  S = bf16(q' k'^T);  P = softmax(bf16(bf16(S * scaling) + mask)) (fp32 inside,
 *             one rounding);  P' = fq3(P);  O = bf16(P' v');  optionally fq4(O) (the output projection's input quantizer)
 *   backward  g = e0(dO);  dP = bf16(g v'^T), dV = bf16(P'^T g);  dS = bf16(bf16((dP - sum dP P) P) * scaling);  dS' = e1(dS);
 *             dQ = bf16(dS' k'), dK = bf16(dS'^T q')
 * Every fq / e is one qt_fake_quant_bf16 call with its own scale and amax slot (fqs[i]: scale_f32_dev, amax_bits_dev -- nullable:
 * not observed --, out_dev; src ignored); all quantizers of a launch share `fmt`.  One workgroup per (batch, head), the products on
 * the matrix cores with fp32 accumulation; the softmax arithmetic is qt_softmax_fq_probs_bf16's / qt_softmax_backward_chain_bf16's.
 * q, k, v: [batch, heads, positions, 64] views given by the element strides of batch, position and head (the 64 values of a head
 *   contiguous, strides % 8 == 0); fqs[0..2].out_dev (q', k', v') are written with the SAME strides and are what the backward reads.
 * mask: additive bf16, nullable, element strides of (batch, head, query row), columns contiguous.
 * drop_keep_dev (nullable): attention-probability dropout (modeling_bert.py:152 upstream, between the softmax and av_matmul): a keep mask
 *   [batch, heads, positions, positions] of bytes (1 = keep) drawn by the caller; P_d = bf16(P * keep * drop_scale), drop_scale =
 *   1 / (1 - p) -- torch's dropout arithmetic; P' = fq3(P_d); the backward applies bf16(dP * keep * drop_scale) before the softmax's.
 * probs_dev, fqs[3].out_dev: P and P', [batch, heads, positions, positions].  out_dev, fqs[4].out_dev (nullable): O and fq4(O) in
 *   [batch, positions, heads, 64] -- the layout the output projection reads, so no permute copy follows.
 * backward: grad_out_dev and grad_q/k/v_dev in [batch, positions, heads, 64]; fqs[0] = e0 (av_matmul's backward-pre quantizer),
 *   fqs[1] = e1 (qk_matmul's).  g, dS and dS' stay on the chip unless asked for (a caller whose hooks want to see those calls):
 *   fqs[0].out_dev (nullable) receives g in [batch, positions, heads, 64], grad_scores_dev / fqs[1].out_dev (nullable) dS / dS' in
 *   [batch, heads, positions, positions].
 *   grad_fqs (nullable, 3 entries: dQ, dK, dV): the projections' own backward-pre quantizers (quantize.py:145-146 on the query / key /
 *   value Linears) applied to the gradients on their way out -- entry i with out_dev != NULL writes fq(gradient i) there
 *   ([batch, positions, heads, 64]) next to the unquantized gradient; colsum_out_devs (nullable HOST array of 3 nullable device
 *   pointers): [heads * 64] bf16 sums of that result over batch and position = the projection's bias gradient (fixed order inside a
 *   workgroup, 64-bit fixed-point atomics across the batch: deterministic; colsum_max as in qt_fake_quant_chain_bf16).  ws_dev:
 *   qt_attention_train_backward_ws_bytes(heads) bytes of scratch, zero before the first launch (every launch leaves it zero). */
int qt_attention_train_supported(long batch, int heads, int positions, int head_dim);
int qt_attention_train_bf16(const uint16_t *q_dev, const uint16_t *k_dev, const uint16_t *v_dev, long stride_b, long stride_s, long stride_h,
                            const uint16_t *mask_dev, long mask_sb, long mask_sh, long mask_sq, const qt_chain_stage *fqs, uint16_t *probs_dev,
                            uint16_t *out_dev, const uint8_t *drop_keep_dev, float drop_scale, long batch, int heads, int positions, int head_dim,
                            float scaling, const qt_format *fmt, const uint16_t *lut_dev, void *stream);
int qt_attention_train_backward_bf16(const uint16_t *grad_out_dev, const uint16_t *qq_dev, const uint16_t *kq_dev, const uint16_t *vq_dev,
                                     long stride_b, long stride_s, long stride_h, const uint16_t *probs_dev, const uint16_t *pq_dev,
                                     const qt_chain_stage *fqs, uint16_t *grad_scores_dev, uint16_t *grad_q_dev, uint16_t *grad_k_dev,
                                     uint16_t *grad_v_dev, const qt_chain_stage *grad_fqs, uint16_t *const *colsum_out_devs, float colsum_max,
                                     void *ws_dev, size_t ws_bytes, const uint8_t *drop_keep_dev, float drop_scale, long batch, int heads,
                                     int positions, int head_dim, float scaling, const qt_format *fmt, const uint16_t *lut_dev, void *stream);
size_t qt_attention_train_backward_ws_bytes(int heads);

/* ---- A9 on the FP8 matrix cores with the weight fake-quantizer fused into the GEMM (the default Linear route for
 * stateless E4M3 / E5M2 specs): y[M][sum n] = x . [fq(W_0); fq(W_1); ...]^T (+ bias_i), bf16 out, fp32 accumulation.
 *     modules/qat/linear.py:40-41   F.linear(input, self.weight_fake_quant(self.weight), self.bias)
 * x8: the FP8 codes of the already fake-quantized activation ([M][K], what qt_fake_quant_bf16_fp8 and the producer-fused
 * kernels emit; format 0 = E4M3, 1 = E5M2).  w_devs[i]: the UNQUANTIZED bf16 weight i, [ns[i]][K] row-major; the kernel
 * applies the e4m3 / e5m2 value map (w_format; unit scale, fp8.py:10-67) to every weight tile on its way from LDS to the
 * matrix core, so fq(W) is never written to HBM and W is read once.  Up to 4 weights that share the activation (q / k / v
 * projections) go in one launch, their outputs side by side in y (row stride sum n).  bias_devs (nullable array of
 * nullable bf16 [ns[i]] vectors).  w_devs / bias_devs / ns are HOST arrays.  K % 128 == 0, ns[i] % 16 == 0. */
int qt_linear_fq8_bf16(const uint8_t *x8_dev, int x_format, const uint16_t *const *w_devs, const uint16_t *const *bias_devs,
                       const int *ns, int count, int w_format, uint16_t *y_dev, int M, int K, void *stream);

/* ---- A9 under autograd: the three bf16 GEMMs of a QAT Linear in a TRAINING step (modules/qat/linear.py:40-41 called by
 * run_glue_no_trainer.py:655-668; the hooks of quantize.py:116-179 have already fake-quantized every operand):
 *     C[M][N] (bf16) = op(A) . op(B) (+ bias[N], bf16), fp32 accumulation, one rounding
 *         forward   y  = x  . Wq^T + b     trans_a 0: A = x  [M][K], row stride lda     trans_b 0: B = Wq [N][K], row stride ldb
 *         dgrad     gx = gy . Wq           trans_a 0: A = gy [M][K]                      trans_b 1: B = Wq [K][N]
 *         wgrad     gW = gy^T . x          trans_a 1: A = gy [K][M]                      trans_b 1: B = x  [K][N]
 * `count` (1..4) problems of ONE shape in one launch (query / key / value), each with its own pointers; `problems` is a HOST array.
 * K % 64 == 0, M % 8 == 0, N % 8 == 0, lda / ldb % 8 == 0, ldc % 4 == 0, a / b 16-byte and c / bias 8-byte aligned -- anything else returns
 * QT_ERR_BAD_ARG / QT_ERR_UNALIGNED and the caller keeps torch's GEMM.  Deterministic: the order of an element's additions is fixed. */
typedef struct qt_gemm_problem {
    const uint16_t *a;
    const uint16_t *b;
    const uint16_t *bias;     /* nullable */
    uint16_t *c;
} qt_gemm_problem;
int qt_train_gemm_bf16(const qt_gemm_problem *problems, int count, int trans_a, int trans_b, int M, int N, int K, long lda, long ldb, long ldc,
                       void *stream);
/* Both backward products of `count` (1..4) Linears of ONE shape -- query / key / value, or a single Linear -- in one launch:
 *     gx[i] [T][I] = gy[i] [T][O] . wq[i] [O][I]          gw[i] [O][I] = gy[i]^T . x[i] [T][I]
 * the same tiles in the same k order as qt_train_gemm_bf16(trans_a 0, trans_b 1) and (1, 1) compute: bit for bit their results.
 * `items` is a HOST array.  T, O multiples of 64 and >= 256, I % 8 == 0, ld_gy / ld_w / ld_x % 8 == 0, ld_gx / ld_gw % 4 == 0; gy / wq / x
 * 16-byte, gx / gw 8-byte aligned -- anything else returns QT_ERR_BAD_ARG / QT_ERR_UNALIGNED and the caller issues the two single launches. */
typedef struct qt_linear_backward {
    const uint16_t *gy, *wq, *x;
    uint16_t *gx, *gw;
} qt_linear_backward;
int qt_train_gemm_backward_bf16(const qt_linear_backward *items, int count, int T, int O, int I, long ld_gy, long ld_w, long ld_x, long ld_gx,
                                long ld_gw, void *stream);

/* ---- H3's step end: clip_grad_norm_(max_norm) and the AdamW update of every parameter tensor in four launches ------------------------
 *     run_glue_no_trainer.py:655-668   accelerator.clip_grad_norm_(model.parameters(), 1.0); optimizer.step()
 * The arithmetic is torch's, which the reference calls: torch.nn.utils.clip_grad_norm_ on bf16 gradients (per-tensor norms and the
 * coefficient rounded to bf16 where torch's tensors are bf16) and torch.optim.AdamW's fused kernel (torch 2.10, ATen/native/cuda/
 * fused_adam_utils.cuh:27-98: hyper-parameters in double, state in fp32, one rounding per stored value).  bf16 parameters, gradients,
 * exp_avg, exp_avg_sq; amsgrad / maximize / grad scaler are not covered (the caller keeps torch's optimizer for those).
 * qt_adamw_tensor: one parameter tensor.  step_dev (nullable): torch's capturable step count, incremented by the call; NULL: `step` is
 * the count AFTER this update, kept by the host.  lr_dev (nullable): the learning rate read on the device (a tensor lr), else `lr`.
 * qt_clip_adamw_plan (host): fills first_chunk of every entry and chunk_tensor_out[chunk] = tensor index (nullable; `capacity` entries);
 * returns the number of 8192-element chunks = workgroups of a launch.  The table and the chunk map are then copied to the device by the
 * caller (tensors_dev, chunk_tensor_dev).  ws_dev: qt_clip_adamw_ws_bytes(ntensors, nchunks) bytes, 16-byte aligned, caller-owned.
 * max_norm <= 0: no clipping.  total_norm_out_dev (nullable, fp32 holding the bf16 value torch returns).  phases: 1 = norm and
 * coefficient only (kept in ws_dev), 2 = step counts + update with the coefficient in ws_dev, 3 = both.  The gradients are NOT rewritten
 * (torch's clip scales them in place; the loop drops them right after the step).  Deterministic: every sum has a fixed order. */
typedef struct qt_adamw_tensor {
    void *param_dev, *grad_dev, *exp_avg_dev, *exp_avg_sq_dev;
    float *step_dev;
    const float *lr_dev;
    long numel;
    long first_chunk;
    double lr, beta1, beta2, eps, weight_decay;
    double step;
} qt_adamw_tensor;
long qt_clip_adamw_plan(qt_adamw_tensor *tensors_host, int ntensors, int32_t *chunk_tensor_out, long capacity);
size_t qt_clip_adamw_ws_bytes(int ntensors, long nchunks);
int qt_clip_adamw_bf16(qt_adamw_tensor *tensors_dev, const int32_t *chunk_tensor_dev, int ntensors, long nchunks, float max_norm,
                       float *total_norm_out_dev, void *ws_dev, size_t ws_bytes, int phases, void *stream);

/* Host-only query: how qt_linear_fq8_bf16 (pair 0, n_total = sum n) or qt_mlp_fq8_bf16 (pair 1, n_total = N: the gate / up pairs)
 * cuts a problem on the current device -- tiles_m x tiles_n workgroups of 256 rows x groups_lo..groups_hi 16-column groups (gate and
 * up groups both counted in pair mode) -- and which kernel variant runs it: 0 = two k tiles per step (narrow tiles), 2 / 4 = one
 * k tile per step with that many weight pieces per wave, 6 = six pieces with two register sets each (the widest tiles).  The parity
 * tests enumerate it over the package's route tables so that every (variant, tile width) a default route can reach is covered. */
int qt_linear_fq8_plan(int M, long n_total, int K, int pair, int *tiles_m, int *tiles_n, int *groups_lo, int *groups_hi, int *variant);

/* ---- The gated MLP front half as ONE launch (LLaMA: modeling_llama.LlamaMLP.forward, act_fn(gate_proj(x)) * up_proj(x), whose
 * result is the down projection's fake-quantized input): both fake-quant Linears of A9 (modules/qat/linear.py:40-41, stateless
 * E4M3 / E5M2 weight specs) on the same FP8-coded activation, then, per output element, in the module chain's arithmetic:
 *     g = bf16(acc_gate + bias_gate), u = bf16(acc_up + bias_up), p = bf16(bf16(g / (1 + exp(-g))) * u)   -- SiLU * up
 *     h = fq_out(p)                                              -- the consumer's input fake-quantizer (fake_quantize.py:217-248)
 * written as bf16 values (h_dev [M][N]; NULL: not written, a consumer that multiplies codes needs only those) AND as FP8 codes
 * (h8_dev [M][N]); out_format: an e4m3 / e5m2 closed-form format with unit
 * scale (qt_format_for).  w_gate_dev / w_up_dev: UNQUANTIZED bf16 [N][K]; the weight value map (w_format) is applied inside the
 * GEMM as in qt_linear_fq8_bf16.  K % 128 == 0, N % 16 == 0; pointers 16-byte aligned. */
int qt_mlp_fq8_bf16(const uint8_t *x8_dev, int x_format, const uint16_t *w_gate_dev, const uint16_t *w_up_dev,
                    const uint16_t *bias_gate_dev, const uint16_t *bias_up_dev, int N, int w_format, uint16_t *h_dev, uint8_t *h8_dev,
                    const qt_format *out_format, int M, int K, void *stream);

/* ---- A9 for every other value map (posit, intN, fp6 / fp4, ...): the weight fake-quantizer inside a bf16 GEMM ------------
 *     modules/qat/linear.py:40-41   F.linear(input, self.weight_fake_quant(self.weight), self.bias)
 *     fake_quantize.py:31-95        the 65 536-entry value map the weight fake-quantizer applies (stateless specs: scale 1)
 * qt_build_rowparams (host) turns a value map (qt_build_map) into 512 rows of four words, one row per (sign, exponent) of a
 * bf16 input: within a row the map is   t = f32(|x| bits + D);  y = clamp((t + C) - C, lo, hi)   -- a round-to-nearest-even
 * onto a power-of-two grid done by the fp32 adder.  Every row is verified against the map on all of its 128 inputs; rows
 * that cannot be written this way (non-finite inputs, a few rows at the far ends of some formats) are `flagged`.
 *   row[r] = {D (int32), C (fp32 bits; bit 0 set = flagged), lo, hi (fp32 bits)};  r = bits >> 7 (rows 256.. = negative inputs)
 *   a flagged row holds {0, 1, y0, y0} with y0 = |map| of the row's first input (mantissa 0; row 0: of zero itself), so the formula
 *   is still right for that one input -- the exact zeros of a flagged row 0 need no map lookup -- or {1, 1, 0, 0} if y0 is a NaN or has its sign bit set
 *   signed_rows 0: rows 256..511 equal rows 0..255 (|map(-x)| == |map(x)|), the kernel indexes by exponent only
 *   sign_mask 0x80008000: the result takes the input's sign; 0: it stays positive (unsigned formats such as fp8_e5m3)
 *   zero_sign: what a zero result of a negative non-zero input looks like in the map -- 0: +0, 1: -0, 2: both occur (a GEMM operand's
 *   zero has no sign that matters; the elementwise passes reproduce the map's bits and take the row form only for 0 and 1) */
typedef struct qt_rowparams {
    uint32_t row[512][4];
    int32_t signed_rows;
    uint32_t sign_mask;
    int32_t n_flagged;
    int32_t zero_sign;
    uint8_t flagged[512];
} qt_rowparams;
int qt_build_rowparams(const uint16_t *map_host, qt_rowparams *out_host);
/* Host evaluation of the row form on one bf16 pattern (tests: rows that are not flagged reproduce the map bit for bit). */
uint16_t qt_rowparams_apply_host(const qt_rowparams *rp, uint16_t bf16_bits, int *flagged);

/* y[M][sum n] = x . [fq(W_0); fq(W_1); ...]^T (+ bias_i), bf16 in / out, fp32 accumulation on v_mfma_f32_16x16x32_bf16.
 * x_dev: the bf16 VALUES of the already fake-quantized activation [M][K].  w_devs[i]: the UNQUANTIZED bf16 weight i [ns[i]][K];
 * every weight tile goes through the row form of its value map on its way to LDS (rows_dev: the 512 x 4 words of
 * qt_build_rowparams in device memory), so fq(W) never exists in HBM and W is read once.  A tile that meets a flagged row is
 * redone with map_dev (the 65 536-entry map in device memory) -- results are exact for every bf16 weight.  Products of
 * quantized values are the reference's bf16 products.  K % 32 == 0, ns[i] % 16 == 0, up to 4 weights per launch; w_devs /
 * bias_devs / ns are HOST arrays.  Alignment: x_dev, y_dev, w_devs[i], rows_dev 16 bytes (y rows are stored 16 bytes per lane),
 * bias_devs[i] 8 bytes.  This entry point never splits K (no workspace): narrow outputs run on part of the chip. */
int qt_linear_fqt_bf16(const uint16_t *x_dev, const uint16_t *const *w_devs, const uint16_t *const *bias_devs, const int *ns,
                       int count, const uint32_t *rows_dev, int signed_rows, uint32_t sign_mask, const uint16_t *map_dev,
                       uint16_t *y_dev, int M, int K, void *stream);

/* The same product with split-K for narrow outputs (modules/qat/linear.py:40-41 at LLaMA's o / down projections: N <= 5120 at
 * M = 1024 gives 80 tiles of 512 x 128 for 256 CUs).  qt_linear_fqt_plan says how the library cuts a problem: *ksplit workgroups
 * share a tile (1: no split, no workspace needed), each multiplies a contiguous range of k steps and writes fp32 partial sums to
 * ws_dev (*ws_bytes, 16-byte aligned, caller-owned scratch, contents irrelevant); the workgroup that draws the last of a tile's
 * tickets adds the partial sums in split order (deterministic: independent of arrival order), adds the bias and writes y.
 * tickets_dev: *n_tickets uint32, ZERO before the first launch; every launch leaves them zero again.  Launches that share
 * a workspace must be ordered on one stream.  qt_linear_fqt_ws_bf16 returns QT_ERR_BAD_ARG when the plan needs more workspace
 * or tickets than were passed.  Numerics: fp32 sums of the same exact products; with ksplit > 1 the summation order differs from
 * the unsplit kernel's (k ranges are summed separately, then added). */
int qt_linear_fqt_plan(int M, long n_total, int K, int *ksplit, size_t *ws_bytes, size_t *n_tickets);
int qt_linear_fqt_ws_bf16(const uint16_t *x_dev, const uint16_t *const *w_devs, const uint16_t *const *bias_devs, const int *ns,
                          int count, const uint32_t *rows_dev, int signed_rows, uint32_t sign_mask, const uint16_t *map_dev,
                          uint16_t *y_dev, int M, int K, float *ws_dev, size_t ws_bytes, uint32_t *tickets_dev, size_t n_tickets,
                          void *stream);


/* ---- A10: attention-score path between the two attention GEMMs ------------------------------
 * Replaces, for one attention block, the chain
 *     attn_scaling(scores, scaling) ; + attention_mask ; softmax(fp32) ; .to(bf16) ; fq(probs)
 *     modules/quantizable/modeling_bert.py:142-158 (MulFunctional, mask add, nn.Softmax, av_matmul's
 *     input hook quantize.py:128-140) -- six passes over the S x S tensor -- by one read and one write.
 * scores / out: bf16 [batch, heads, q_len, cols] contiguous; mask: additive bf16 addressed as
 * mask + b*mask_sb + h*mask_sh + q*mask_sq + col (NULL = none).  fmt / lut / scale / amax describe the
 * fake-quantizer applied to the probabilities (fmt kind IDENTITY = plain softmax).  cols <= 4096, cols % 8 == 0. */
int qt_softmax_fq_bf16(const uint16_t *scores_dev, const uint16_t *mask_dev, uint16_t *out_dev, long batch,
                       int heads, int q_len, long cols, long mask_sb, long mask_sh, long mask_sq, float scaling,
                       const qt_format *fmt, const uint16_t *lut_dev, const float *scale_f32_dev,
                       uint32_t *amax_bits_dev, void *stream);
/* Same pass for a stateless E4M3 / E5M2 fake-quantizer of the probabilities (unit scale, no observer) that writes their
 * FP8 code (out8, contiguous like scores; out_dev may be NULL): the P.V product then runs as an FP8 GEMM (qt_fp8_gemm)
 * and the S x S tensor costs 1 byte per element to write and read instead of 2. */
int qt_softmax_fq_bf16_fp8(const uint16_t *scores_dev, const uint16_t *mask_dev, uint16_t *out_dev, uint8_t *out8_dev,
                           long batch, int heads, int q_len, long cols, long mask_sb, long mask_sh, long mask_sq,
                           float scaling, const qt_format *fmt, void *stream);
/* The FP8-only variant with a per-row shortcut for masks that end in a masked run (causal, right padding): row_live_dev[r] = one
 * past the last column of mask row r whose entry is above -1e30 (qt_mask_row_live; r = b * live_sb + h * live_sh + q * live_sq, in rows
 * of the mask's own [b][h][q] extent, 0 strides for broadcast dimensions).  512-column pieces that lie entirely beyond it are
 * neither loaded nor evaluated -- their probabilities are exactly 0 -- unless the row has no unmasked column at all (then it is a
 * uniform distribution and is evaluated in full).  Same results as qt_softmax_fq_bf16_fp8, bit for bit. */
int qt_mask_row_live(const uint16_t *mask_dev, long rows, long cols, long row_stride, int *row_live_dev, void *stream);
/* The same, and *irregular_dev (device) = 0 when every row is exactly "zeros up to its extent, the bf16 minimum from there on" (causal
 * masks, right padding), else 1: qt_attention_fp8 then decides on the device whether it has to read the mask (no host read-back, so it
 * works inside a stream capture). */
int qt_mask_row_live_checked(const uint16_t *mask_dev, long rows, long cols, long row_stride, int *row_live_dev, int *irregular_dev,
                             void *stream);
/* ---- The attention core on FP8 codes, head_dim D = 128 or 64, ONE launch (modules/quantizable/modeling_llama.py:228-246, modeling_bert.py:118-158):
 *     out = av_matmul(fq_p(softmax(attn_scaling(qk_matmul(fq_q(q), fq_k(k)^T), scale) + mask)), fq_v(v))
 * for stateless E4M3 / E5M2 fake-quantizers (unit scale) of one format on all four matmul inputs, with the module chain's rounding
 * points (bf16 scores, bf16(score * scaling), bf16(. + mask), fp32 softmax over the whole row, bf16 probabilities, fp32 P.V, bf16 out).
 * q8 / k8: the FP8 codes of fq_q(q), fq_k(k), [B][H][Sq | Sk][D]; vt8: the codes of fq_v(v) TRANSPOSED to [B][H][D][Sk] with the keys
 * of every 128-block in the k-slot order of the P.V instruction -- written by qt_value_codes_t from the bf16 value tensor (element
 * strides for batch, head, key; head_dim contiguous), which IS the fq_v call.  mask: additive bf16 or NULL (element strides, columns
 * contiguous); row_live (optional, qt_mask_row_live): key blocks beyond every row's last unmasked column are skipped; mask_is_simple: the
 * caller has checked that every mask row is exactly 0 up to its row_live entry and the bf16 minimum from there on (causal, right padding),
 * and the kernel applies that without reading the mask (x + 0 = x; bf16(x + min) = min for finite x); mask_irregular_dev (optional): the
 * device-side verdict of qt_mask_row_live_checked, 0 meaning the same.  out: [B][Sq][H][D] bf16;
 * with out8 (+ out_format, an e4m3 / e5m2 closed-form format) the consumer's stateless input fake-quantizer -- the output projection's --
 * is applied on the way out: out = fq(result), out8 its FP8 codes.  Sk % 128 == 0, Sk <= 1024. */
int qt_value_codes_t(const uint16_t *v_dev, uint8_t *vt8_dev, long B, long H, long Sk, int head_dim, long stride_b, long stride_h, long stride_k,
                     const qt_format *fmt, void *stream);
int qt_attention_fp8(const uint8_t *q8_dev, const uint8_t *k8_dev, const uint8_t *vt8_dev, int operand_format, const uint16_t *mask_dev,
                     long mask_sb, long mask_sh, long mask_sq, const int *row_live_dev, long live_sb, long live_sh, long live_sq,
                     int mask_is_simple, const int *mask_irregular_dev, uint16_t *out_dev, uint8_t *out8_dev, const qt_format *out_format, long B,
                     int H, int Sq, int Sk, int head_dim, float scaling, void *stream);
int qt_softmax_fq_bf16_fp8_live(const uint16_t *scores_dev, const uint16_t *mask_dev, uint8_t *out8_dev, long batch, int heads, int q_len,
                                long cols, long mask_sb, long mask_sh, long mask_sq, float scaling, const qt_format *fmt,
                                const int *row_live_dev, long live_sb, long live_sh, long live_sq, void *stream);

/* Whole attention core for already fake-quantized q, k, v (bf16 [B, H, S, D] contiguous, D = 64 or 128):
 *     O = av_matmul( fq_P( softmax( attn_scaling(qk_matmul(q, k^T), scaling) + mask ) ), v )
 * with every bf16 rounding point of the reference chain kept (modules/quantizable/modeling_bert.py:118-158,
 * functional_modules.py:22-26) and the S x S tensor never written.  mask as in qt_softmax_fq_bf16 (strides % 4
 * == 0).  fmt / lut / scale / amax: fake-quantizer of the probabilities.  out: bf16 [B, Sq, H, D] contiguous
 * (the transposed layout the attention block needs next). */
int qt_attention_fq_bf16(const uint16_t *q_dev, const uint16_t *k_dev, const uint16_t *v_dev, const uint16_t *mask_dev,
                         uint16_t *out_dev, int B, int H, int Sq, int Sk, int D, long mask_sb, long mask_sh,
                         long mask_sq, float scaling, const qt_format *fmt, const uint16_t *lut_dev,
                         const float *scale_f32_dev, uint32_t *amax_bits_dev, void *stream);
/* The same at unit scale without an observer, with the CONSUMER's input fake-quantizer (the output projection's hook,
 * quantize.py:128-140) applied to the result as well when it is the probabilities' stateless format: out = fmt(attention output). */
/* Both of the above (out_fq = 0 / 1; scale_dev / amax_bits_dev as in qt_attention_fq_bf16, both NULL with out_fq) with the mask's row
 * extents (qt_mask_row_live_checked: row_live_dev, strides in rows for (b, h, q), and its device flag): when the flag says every mask
 * row is "zeros, then the bf16 minimum" (causal masks, right padding) the kernel derives the mask values and the dead key tiles from
 * the extents and does not read the mask at all.  Same results bit for bit. */
int qt_attention_fq_live_bf16(const uint16_t *q_dev, const uint16_t *k_dev, const uint16_t *v_dev, const uint16_t *mask_dev,
                              uint16_t *out_dev, int B, int H, int Sq, int Sk, int D, long mask_sb, long mask_sh, long mask_sq,
                              float scaling, const qt_format *fmt, const uint16_t *lut_dev, const float *scale_dev,
                              uint32_t *amax_bits_dev, int out_fq, const int *row_live_dev, long live_sb, long live_sh, long live_sq,
                              const int *mask_irregular_dev, void *stream);
int qt_attention_fq_out_bf16(const uint16_t *q_dev, const uint16_t *k_dev, const uint16_t *v_dev, const uint16_t *mask_dev,
                             uint16_t *out_dev, int B, int H, int Sq, int Sk, int D, long mask_sb, long mask_sh, long mask_sq,
                             float scaling, const qt_format *fmt, const uint16_t *lut_dev, void *stream);

/* The same attention core for stateless TABLE formats (posit, fpN, ...: BASELINE configs[3]) as ONE launch with the whole score strip
 * in registers (round 4; the design of qt_attention_fp8 on v_mfma_f32_16x16x32_bf16): head_dim 128, Sk a multiple of 128 up to 1024,
 * q / k [B][H][S][128] bf16 VALUES of fq(q), fq(k); the probabilities' fake-quantizer (and, with out_fq, the output projection's input
 * fake-quantizer: the same format) is `fmt` in its row form -- lut_dev is the device map with the row words behind it and fmt->p1 bit 0
 * set (fake_quantize._device_map).  vt_dev: fq(v) transposed to [B][H][128][Sk] with the keys of every 32-chunk in the k-slot order of
 * the P.V instruction, written by qt_value_t_rows from a [B, H, Sk, 128] view with element strides (this IS the `fq_v` call of
 * modeling_llama.py:244-246).  Mask: additive bf16 with element strides, or -- with row_live_dev / mask_irregular_dev from
 * qt_mask_row_live_checked -- never read when every row is "zeros, then the bf16 minimum".  Every rounding point of the module chain
 * is kept (S, S * scaling, + mask in bf16; fp32 softmax rounded to bf16; fq_P; bf16 output). */
int qt_value_t_rows(const uint16_t *v_dev, uint16_t *vt_dev, long B, long H, long Sk, int head_dim, long stride_b, long stride_h, long stride_k,
                    const qt_format *fmt, const uint16_t *lut_dev, void *stream);
int qt_attention_rows_bf16(const uint16_t *q_dev, const uint16_t *k_dev, const uint16_t *vt_dev, const uint16_t *mask_dev, long mask_sb, long mask_sh,
                           long mask_sq, const int *row_live_dev, long live_sb, long live_sh, long live_sq, const int *mask_irregular_dev,
                           uint16_t *out_dev, int out_fq, const qt_format *fmt, const uint16_t *lut_dev, long B, int H, int Sq, int Sk, int head_dim,
                           float scaling, void *stream);

/* ---- section 8(f).2: block-scaled (microscaling) GEMMs on the scaled matrix instruction ---------------------
 * Replaces linear_mx / matmul_mx (decomposed.py:304-363: operand * expand(block scale) twice, then F.linear /
 * torch.matmul) when both operands are in a format v_mfma_scale_f32_16x16x128_f8f6f4 takes -- element formats
 * QT_MX_E4M3 / E5M2 (the reference's fp8_e4m3 / fp8_e5m2), E2M3 / E3M2 (fp6_e2m3 / fp6_e3m2), E2M1 (fp4_e2m1) with
 * power-of-two scales (force_scale_power_of_two, "fp8_e8m0") over blocks of 32 k (or a multiple of 32).
 *
 * Packed operand: codes[rows][K * bits / 8] (element i of a row in bits [i*bits, (i+1)*bits), little-endian),
 * e8m0[rows][K / 32] (scale 2^(byte - 127)).
 *
 * qt_mx_pack: values + block scales, as the reference's quantize / quantize_mx return them (bf16 or fp32, addressed
 * with element strides so that a [K, N] operand of matmul_mx is packed as [N, K]), -> packed operand.  *bad_dev
 * (nullable, caller-zeroed) is set to 1 when a value is not exactly representable in elem_format or a scale is not
 * 2^e with -127 <= e <= 127: the caller must then keep the dequantize + GEMM path.
 * qt_mx_gemm: C[b][M][N] = A[b][M][K] . B[b][N][K]^T (+ bias[N]), fp32 accumulation, C / bias bf16 or fp32.  Batch
 * strides are given in rows (0 = operand shared by every batch).  Returns QT_ERR_BAD_DTYPE for a format pair
 * without a kernel, QT_ERR_UNALIGNED unless K * bits / 8 is a multiple of 16 and the code pointers are 16-byte aligned. */
enum { QT_MX_E4M3 = 0, QT_MX_E5M2 = 1, QT_MX_E2M3 = 2, QT_MX_E3M2 = 3, QT_MX_E2M1 = 4 };
int qt_mx_pack(const void *x_dev, const void *scale_dev, int is_f32, uint8_t *codes_dev, uint8_t *e8m0_dev, long batch,
               long rows, long K, long x_batch_stride, long x_row_stride, long x_k_stride, long s_batch_stride,
               long s_row_stride, long s_k_stride, int block_size, int elem_format, int *bad_dev, void *stream);
int qt_mx_gemm(const uint8_t *a_codes, const uint8_t *a_e8m0, int a_format, const uint8_t *b_codes, const uint8_t *b_e8m0,
               int b_format, void *c_dev, int c_is_f32, const void *bias_dev, long batch, int M, int N, int K,
               long a_batch_stride_rows, long b_batch_stride_rows, void *stream);

/* ---- section 8(f).1: converted per-tensor graphs, quantize -> GEMM -> dequantize(s_x * s_w)     quantize_pt2e.py:323-446
 * The int8 GEMM of such a graph on v_mfma_i32_16x16x64_i8 with the dequantize node in its epilogue:
 *     C[b][M][N] = round_C( round_C( A[b][M][K] . B[b][N][K]^T + bias[N] ) * out_scale )
 * A, B: int8 CODES (what `quantize` produced, narrowed; weights are stored as codes by convert_pt2e), exact int32
 * accumulation; bias (nullable) and out_scale (nullable; one element, or N elements with out_scale_per_col) in C's dtype
 * (bf16 / fp32); the two roundings are those of the graph's `aten.linear` output and of its `dequantize` multiply.
 * fold_f32 (fp32 C only): the dequantize node looks its input up in the identity value map first, which for an fp32 tensor
 * keeps the high 16 bits with a sticky bit (decomposed.py:151-153); applied to (acc + bias) before the multiply.
 * Batch strides in rows (0 = shared operand).  K % 16 == 0, code pointers 16-byte aligned. */
int qt_q8_gemm(const int8_t *a_codes, const int8_t *b_codes, void *c_dev, int c_is_f32, const void *bias_dev, const void *out_scale_dev,
               int out_scale_per_col, int fold_f32, long batch, int M, int N, int K, long a_batch_stride_rows, long b_batch_stride_rows, void *stream);

/* Fused quantize_mx for blocks along the last axis (decomposed.py:365-448 with axes = [-1]): x [rows][cols] in the
 * tensor dtype -> q (element values map[x / scale], nullable), scales [rows][cols / block_size] (same dtype), and, when
 * codes / e8m0 are given (requires force_pow2, block_size % 32 == 0, pack_format = the QT_MX_* format the map rounds
 * to), the packed operand for qt_mx_gemm.  force_pow2: scale = 2^(floor(log2 amax) - floor(log2 quant_max)) with the
 * reference's in-dtype logarithm; otherwise scale = amax / quant_max [-> scale_lut].  block_size: power of two,
 * 8..512 (bf16) / 4..256 (fp32), dividing cols. */
int qt_quantize_mx_bf16(const uint16_t *x_dev, uint16_t *q_dev, uint16_t *scales_dev, uint8_t *codes_dev, uint8_t *e8m0_dev,
                        size_t rows, size_t cols, int block_size, const qt_format *fmt, const uint16_t *lut_dev,
                        float quant_max, int force_pow2, const uint16_t *scale_lut_dev, int pack_format, void *stream);
int qt_quantize_mx_f32(const float *x_dev, float *q_dev, float *scales_dev, uint8_t *codes_dev, uint8_t *e8m0_dev,
                       size_t rows, size_t cols, int block_size, const qt_format *fmt, const uint16_t *lut_dev,
                       float quant_max, int force_pow2, const uint16_t *scale_lut_dev, int pack_format, void *stream);

/* ---- the model's own elementwise chains between the fake-quantized GEMMs (Hugging Face LLaMA block) -----------
 * Not part of the reference package: its examples run HF's modeling_llama as is, where each of these is 3-8 torch
 * kernels.  Same operation order and bf16 rounding points as those chains (transformers modeling_llama.py:
 * LlamaRMSNorm.forward, LlamaMLP.forward, apply_rotary_pos_emb / rotate_half).
 *   qt_rmsnorm_bf16:  y[r][c] = bf16(w[c] * bf16(x32 * rsqrt(mean_c(x32^2) + eps))); cols % 8 == 0, cols <= 16384
 *   qt_silu_mul_bf16: y[rows][cols] = bf16(bf16(silu(gate)) * up); gate / up rows start every *_row_stride elements
 *                     (cols when contiguous, more when they are column slices of one fused projection); cols % 8 == 0
 *   qt_rope_bf16:     q, k in [B][S][H][D] memory order (the transposed views HF passes) with a (b, s) row every
 *                     q_row_stride / k_row_stride elements (H * D when contiguous; larger when the projection is a
 *                     column slice of a wider GEMM output), cos / sin [B][S][D];
 *                     out (contiguous [B][S][H][D]) = bf16(bf16(x * cos) + bf16(rotate_half(x) * sin)); D % 16 == 0 */
int qt_rmsnorm_bf16(const uint16_t *x_dev, const uint16_t *weight_dev, uint16_t *y_dev, long rows, long cols, float eps,
                    void *stream);
/* qt_rmsnorm_bf16 with the FIRST consumer's stateless E4M3 / E5M2 fake-quantizer applied to the result (bf16 + FP8
 * code).  Its sibling consumers (k/v projections beside q, up beside gate) still run their own passes on the result;
 * those formats are idempotent, so what they compute is unchanged.  y_dev may be NULL: the codes only (consumers that multiply codes
 * need nothing else; the codes decode to exactly the values y would hold). */
int qt_rmsnorm_fq8_bf16(const uint16_t *x_dev, const uint16_t *weight_dev, uint16_t *y_dev, uint8_t *y8_dev, long rows,
                        long cols, float eps, const qt_format *fmt, void *stream);
/* BERT-style blocks (transformers modeling_bert.py BertSelfOutput / BertOutput / BertIntermediate, and the twins of
 * upstream modules/quantizable/modeling_bert.py:174-214): `LayerNorm(dense(x) + residual)` and the erf-form GELU.
 *   qt_layernorm_bf16: s = bf16(x + residual) (residual may be NULL: s = x); y = bf16(w * (rstd * (s - mean)) + b) with the
 *                      row mean / biased variance in fp32; cols % 8 == 0, cols <= 16384.  With y8 (and optionally yq) the
 *                      consumer's stateless E4M3 / E5M2 fake-quantizer `fmt` is applied as well: yq = fq(y) as bf16 (NULL: not
 *                      written -- a consumer that multiplies codes needs nothing else, and they decode to exactly fq(y)), y8 its
 *                      FP8 code; y itself stays unquantized (it also feeds the next residual connection).
 *   qt_gelu_bf16:      y = bf16((x * 0.5) * (1 + erf(x * sqrt(1/2)))); with y8 the consumer's fake-quantizer is applied on
 *                      the way out (y = fq(gelu(x)) as bf16 -- may then be NULL: codes only --, y8 its FP8 code); n % 8 == 0 */
int qt_layernorm_bf16(const uint16_t *x_dev, const uint16_t *residual_dev, const uint16_t *weight_dev, const uint16_t *bias_dev,
                      uint16_t *y_dev, uint16_t *yq_dev, uint8_t *y8_dev, long rows, long cols, float eps, const qt_format *fmt,
                      void *stream);
/* qt_layernorm_bf16 with the stateless E4M3 / E5M2 input fake-quantizers of ALL the Linears consuming the result (2 or 3: q, k, v)
 * evaluated in the same launch: yq = fq_0(y) as bf16, y8[i] = the FP8 codes of fq_i(y) (as qt_rmsnorm_consumers_bf16). */
int qt_layernorm_consumers_bf16(const uint16_t *x_dev, const uint16_t *residual_dev, const uint16_t *weight_dev, const uint16_t *bias_dev,
                                uint16_t *y_dev, uint16_t *yq_dev, long rows, long cols, float eps, int consumers, uint8_t *const *y8_dev,
                                const qt_format *const *fmt, void *stream);
int qt_gelu_bf16(const uint16_t *x_dev, uint16_t *y_dev, uint8_t *y8_dev, size_t n, const qt_format *fmt, void *stream);
/* The residual add of a LLaMA block (modeling_llama.py LlamaDecoderLayer.forward: `hidden = residual + hidden`) absorbed into
 * the RMSNorm behind it: sum = bf16(x + residual) (written out: it is the next residual), y = RMSNorm(sum) as
 * qt_rmsnorm_bf16 computes it; with y8 (non-NULL) the first consumer's stateless E4M3 / E5M2 fake-quantizer is applied to y
 * as in qt_rmsnorm_fq8_bf16 (y_dev may then be NULL: codes only). */
int qt_add_rmsnorm_bf16(const uint16_t *x_dev, const uint16_t *residual_dev, const uint16_t *weight_dev, uint16_t *sum_dev,
                        uint16_t *y_dev, uint8_t *y8_dev, long rows, long cols, float eps, const qt_format *fmt, void *stream);
/* qt_add_rmsnorm_bf16 for PT2E-prepared graphs, where the residual stream itself is fake-quantized in front of the NEXT add (the
 * annotator quantizes the earlier-defined operand of a same-shape add, xnnpack_quantizer_utils.py:232-282): the sum is normalised
 * unquantized, and written to sum_dev as sum_fmt(sum) (stateless E4M3 / E5M2) -- that fake-quantizer's call, evaluated where the
 * tensor is in registers.  sum_fmt == NULL: exactly qt_add_rmsnorm_bf16. */
int qt_add_rmsnorm_sumfq_bf16(const uint16_t *x_dev, const uint16_t *residual_dev, const uint16_t *weight_dev, uint16_t *sum_dev,
                              uint16_t *y_dev, uint8_t *y8_dev, long rows, long cols, float eps, const qt_format *fmt,
                              const qt_format *sum_fmt, void *stream);
/* RMSNorm (residual_dev / sum_dev both NULL) or residual add + RMSNorm with the stateless E4M3 / E5M2 input fake-quantizers of ALL the
 * Linears that consume the result (2 or 3: q, k, v -- gate, up) evaluated on it in the same launch: y = fq_0(result) as bf16, y8[i] =
 * the FP8 codes of fq_i(result).  Each consumer's hook then hands its codes through instead of launching its own pass over the tensor
 * (quantize.py:128-140 issues one call per consumer; this evaluates each of them where the tensor is in registers). */
int qt_rmsnorm_consumers_bf16(const uint16_t *x_dev, const uint16_t *residual_dev, const uint16_t *weight_dev, uint16_t *sum_dev,
                              uint16_t *y_dev, long rows, long cols, float eps, int consumers, uint8_t *const *y8_dev,
                              const qt_format *const *fmt, void *stream);
/* The window loss of the WikiText loop (examples/language_modeling/wikitext.py:146-158; transformers' ForCausalLMLoss) from the lm head's
 * bf16 logits [batch][seq_len][vocab] (row_stride elements between positions): label of position s = labels[b][s + 1] (int64; none
 * for the last position), cross entropy in fp32 with ignore_index, mean over the scored positions -> *loss_out (fp32, device).
 * row_loss_scratch: batch * seq_len floats.  Deterministic (fixed summation order, no atomics). */
int qt_causal_lm_loss_bf16(const uint16_t *logits_dev, const long long *labels_dev, long batch, long seq_len, long vocab, long row_stride,
                           long long ignore_index, float *row_loss_scratch_dev, float *loss_out_dev, void *stream);
int qt_silu_mul_bf16(const uint16_t *gate_dev, const uint16_t *up_dev, uint16_t *y_dev, size_t rows, size_t cols,
                     size_t gate_row_stride, size_t up_row_stride, void *stream);
/* qt_silu_mul_bf16 with the consumer's stateless E4M3 / E5M2 fake-quantizer (unit scale) applied on the way out:
 * y = fq(bf16(bf16(silu(gate)) * up)) as bf16 plus its FP8 code -- what the down-projection's input hook would compute
 * from the unfused result (quantize.py:128-140), bit for bit. */
int qt_silu_mul_fq8_bf16(const uint16_t *gate_dev, const uint16_t *up_dev, uint16_t *y_dev, uint8_t *y8_dev, size_t rows,
                         size_t cols, size_t gate_row_stride, size_t up_row_stride, const qt_format *fmt, void *stream);
int qt_rope_bf16(const uint16_t *q_dev, const uint16_t *k_dev, const uint16_t *cos_dev, const uint16_t *sin_dev,
                 uint16_t *q_out_dev, uint16_t *k_out_dev, long B, long S, long Hq, long Hk, long D, long q_row_stride,
                 long k_row_stride, void *stream);
/* qt_rope_bf16 followed by the two stateless E4M3 / E5M2 fake-quantizers of qk_matmul's inputs, outputs contiguous in
 * [B][H][S][D] order (what those hooks' permuted-view pass would write): three launches in one.  q_out8 / k_out8
 * (nullable) receive the FP8 codes in the same order, for Q.K^T as an FP8 GEMM. */
int qt_rope_fq_bf16(const uint16_t *q_dev, const uint16_t *k_dev, const uint16_t *cos_dev, const uint16_t *sin_dev,
                    uint16_t *q_out_dev, uint16_t *k_out_dev, uint8_t *q_out8_dev, uint8_t *k_out8_dev, long B, long S,
                    long Hq, long Hk, long D, long q_row_stride, long k_row_stride, const qt_format *fmt_q,
                    const qt_format *fmt_k, void *stream);
/* The same with the codes mandatory and the bf16 outputs optional (both or neither: the codes decode to exactly those values, so a
 * consumer that multiplies codes needs nothing else), and -- v_dev non-NULL -- qt_value_codes_t (below: the attention kernel's fq_v
 * call on the value projection, [B][Hk][S][D] by element strides, D = 64 or 128 contiguous, S % 128 == 0) in the SAME launch: the two
 * jobs are independent and each alone is too small to fill the chip.  cos_dev = sin_dev = NULL: no rotation -- q and k are only moved
 * to [B][H][S][D] order and fake-quantized (BERT's transpose_for_scores + qk_matmul's input hooks). */
int qt_rope_fq_value(const uint16_t *q_dev, const uint16_t *k_dev, const uint16_t *cos_dev, const uint16_t *sin_dev, uint16_t *q_out_dev,
                     uint16_t *k_out_dev, uint8_t *q_out8_dev, uint8_t *k_out8_dev, long B, long S, long Hq, long Hk, long D,
                     long q_row_stride, long k_row_stride, const qt_format *fmt_q, const qt_format *fmt_k, const uint16_t *v_dev,
                     uint8_t *vt8_dev, long v_stride_b, long v_stride_h, long v_stride_k, const qt_format *fmt_v, void *stream);
/* The producer kernels above for stateless TABLE formats (posit, fpN, ... without `qs`): the consumers' fake-quantizer
 * (quantize.py:128-140 hooks in front of the next Linear / matmul; one format for all consumers, which makes their calls idempotent
 * repeats of this one) is applied to the result in its ROW FORM -- `map_dev` is the 65 536-entry map with the 512 row words of
 * qt_build_rowparams behind it and fmt->p1 bit 0 set (see qt_format) -- and bf16 values are written; no separate elementwise pass.
 * qt_rmsnorm_map_bf16: RMSNorm (residual_dev / sum_dev NULL) or residual add + RMSNorm; qt_silu_mul_map_bf16: SiLU(gate) * up;
 * qt_rope_map_bf16: rotary embedding of q and k into contiguous [B][H][S][D] outputs (what qk_matmul's hooks would write).
 * PT2E-prepared graphs (see qt_add_rmsnorm_sumfq_bf16 / qt_rope_fq_inner_value): quantize_sum != 0 writes the sum through the same map,
 * inner_q / inner_k != 0 put bf16(x * cos) through it before the rotary's add. */
int qt_rmsnorm_map_bf16(const uint16_t *x_dev, const uint16_t *residual_dev, const uint16_t *weight_dev, uint16_t *sum_dev, uint16_t *y_dev,
                        long rows, long cols, float eps, const qt_format *fmt, const uint16_t *map_dev, int quantize_sum, void *stream);
int qt_silu_mul_map_bf16(const uint16_t *gate_dev, const uint16_t *up_dev, uint16_t *y_dev, size_t rows, size_t cols, size_t gate_row_stride,
                         size_t up_row_stride, const qt_format *fmt, const uint16_t *map_dev, void *stream);
int qt_rope_map_bf16(const uint16_t *q_dev, const uint16_t *k_dev, const uint16_t *cos_dev, const uint16_t *sin_dev, uint16_t *q_out_dev,
                     uint16_t *k_out_dev, long B, long S, long Hq, long Hk, long D, long q_row_stride, long k_row_stride,
                     const qt_format *fmt, const uint16_t *map_dev, int inner_q, int inner_k, void *stream);
/* qt_rope_map_bf16 and qt_value_t_rows (the table-format attention core's value pass, below) in ONE launch, as qt_rope_fq_value does
 * for FP8: both read slices of the q / k / v projections' product, neither depends on the other, each alone is too small to fill the
 * chip.  v_dev: [B][Hk][S][D] by element strides, D = 128 contiguous, S % 128 == 0; vt_dev: [B][Hk][128][S] bf16 (qt_value_t_rows'
 * layout); `fmt` / `map_dev` serve q, k and v -- the value's fake-quantizer must be that format too. */
int qt_rope_map_value(const uint16_t *q_dev, const uint16_t *k_dev, const uint16_t *cos_dev, const uint16_t *sin_dev, uint16_t *q_out_dev,
                      uint16_t *k_out_dev, long B, long S, long Hq, long Hk, long D, long q_row_stride, long k_row_stride,
                      const qt_format *fmt, const uint16_t *map_dev, int inner_q, int inner_k, const uint16_t *v_dev, uint16_t *vt_dev,
                      long v_stride_b, long v_stride_h, long v_stride_k, void *stream);
/* qt_rope_map_value with a third independent job of the same format in the launch: y = fq(W) for a weight tensor (w_elems bf16 elements,
 * w_elems % 8 == 0, both pointers 16-byte aligned) -- the `weight_fake_quant(self.weight)` call of a Linear that runs as weight pass +
 * library GEMM right behind the attention core (modules/qat/linear.py:40-41: LLaMA's o_proj), HBM-bound where the other two jobs wait
 * on latency.  w_elems == 0: qt_rope_map_value. */
int qt_rope_map_value_weight(const uint16_t *q_dev, const uint16_t *k_dev, const uint16_t *cos_dev, const uint16_t *sin_dev, uint16_t *q_out_dev,
                             uint16_t *k_out_dev, long B, long S, long Hq, long Hk, long D, long q_row_stride, long k_row_stride,
                             const qt_format *fmt, const uint16_t *map_dev, int inner_q, int inner_k, const uint16_t *v_dev, uint16_t *vt_dev,
                             long v_stride_b, long v_stride_h, long v_stride_k, const uint16_t *w_dev, uint16_t *wq_dev, size_t w_elems,
                             void *stream);
/* qt_rope_fq_value for PT2E-prepared graphs (wikitext.py:60-136 exports HF's apply_rotary_pos_emb as q * cos + rotate_half(q) * sin and
 * the annotator fake-quantizes the add's earlier operand): out = fmt(inner(bf16(x * cos)) + bf16(rotate_half(x) * sin)) with inner_q /
 * inner_k stateless closed-form FP formats (NULL: none -- exactly qt_rope_fq_value).  v_dev may be NULL (no value job). */
int qt_rope_fq_inner_value(const uint16_t *q_dev, const uint16_t *k_dev, const uint16_t *cos_dev, const uint16_t *sin_dev, uint16_t *q_out_dev,
                           uint16_t *k_out_dev, uint8_t *q_out8_dev, uint8_t *k_out8_dev, long B, long S, long Hq, long Hk, long D,
                           long q_row_stride, long k_row_stride, const qt_format *fmt_q, const qt_format *fmt_k, const qt_format *inner_q,
                           const qt_format *inner_k, const uint16_t *v_dev, uint8_t *vt8_dev, long v_stride_b, long v_stride_h,
                           long v_stride_k, const qt_format *fmt_v, void *stream);

/* ---- plain FP8 GEMM on already fake-quantized operands, through hipBLASLt with the algorithm NAMED by the caller ----------
 * C[b][M][N] (bf16) = A[b][M][K] . op(B) (+ bias[N], bf16); A, B are OCP FP8 bytes (format 0 = E4M3, 1 = E5M2) whose
 * values are exactly the fake-quantized bf16 values (unit scale), so the products are the reference's
 * (modules/qat/linear.py:40-41 after both fake-quantizers; modules/quantizable/modeling_bert.py:118-158 for the two
 * attention matmuls).  b_is_kn = 0: B is [N][K] row-major (a Linear weight, or K for Q.K^T); 1: B is [K][N] row-major (V
 * for P.V).  Batch strides in elements (0 = shared).  `workspace` is caller-owned scratch for the library (64 MiB is plenty;
 * may be NULL).  `algo`: index into the library's ordered list of suggestions for this problem (0 = its first; an index
 * it did not return, or one that needs more workspace than given, runs the first).  Nothing is ever timed here: the choice
 * -- and with it the order of the fp32 additions -- is the same in every process, rank and box (the package passes a committed
 * per-shape table, fused._LT_ALGO_TABLE).  Returns QT_ERR_NO_DEVICE when libhipblaslt cannot be resolved in the process and
 * QT_ERR_BAD_DTYPE when it has no kernel for the problem -- the caller then keeps its other route. */
int qt_fp8_gemm(const uint8_t *a8_dev, int a_format, const uint8_t *b8_dev, int b_format, int b_is_kn, void *c_bf16_dev,
                const void *bias_bf16_dev, long batch, int M, int N, int K, long a_batch_stride, long b_batch_stride,
                long c_batch_stride, void *workspace_dev, size_t workspace_bytes, int algo, void *stream);
/* Tools only (tools/tune_lt_algos.py fills the committed table with it): times every suggestion of the library for the problem
 * (20 launches each on `stream`, which must not be capturing), *best = index of the fastest, us[i] = microseconds per launch of
 * suggestion i (-1: not runnable; nullable, max_us entries).  Returns the number of suggestions (>= 1) or a negative error. */
int qt_fp8_gemm_tune(const uint8_t *a8_dev, int a_format, const uint8_t *b8_dev, int b_format, int b_is_kn, void *c_bf16_dev,
                     const void *bias_bf16_dev, long batch, int M, int N, int K, long a_batch_stride, long b_batch_stride,
                     long c_batch_stride, void *workspace_dev, size_t workspace_bytes, int *best, float *us, int max_us, void *stream);

/* hipblasLtGetVersion of the library qt_fp8_gemm resolved in this process (0: none, or it does not say).  The position of a kernel in
 * the library's suggestion list belongs to ONE build of the library: the package's committed algorithm table names the version it was
 * measured on and passes algo 0 (the library's first suggestion) under any other. */
int qt_fp8_gemm_library_version(void);

/* Bench helper: times `iters` back-to-back launches of the fused per-tensor pass with HIP events
 * on `stream` and returns the mean milliseconds per launch in *ms_out (bench.py roofline leg).
 * Launch i works on x_dev + (i % pool_count) * pool_stride and y_dev + (i % pool_count) * pool_stride
 * (elements), so a pool larger than the 256 MiB Infinity Cache measures HBM, not cache, traffic. */
int qt_bench_fake_quant_bf16(const uint16_t *x_dev, uint16_t *y_dev, size_t n, const qt_format *fmt,
                             const uint16_t *lut_dev, const float *scale_f32_dev, uint32_t *amax_bits_dev,
                             int iters, size_t pool_stride, int pool_count, void *stream, float *ms_out);
/* Same, for the pass that emits the quantized code as FP8 bytes (qt_fake_quant_bf16_fp8; y_dev NULL = FP8 output
 * only, which is how the weight pass of the FP8 GEMM route runs): y8 pool addressed like x. */
int qt_bench_fake_quant_bf16_fp8(const uint16_t *x_dev, uint16_t *y_dev, uint8_t *y8_dev, size_t n, const qt_format *fmt,
                                 const float *scale_f32_dev, uint32_t *amax_bits_dev, int iters, size_t pool_stride,
                                 int pool_count, void *stream, float *ms_out);

#ifdef __cplusplus
}
#endif
#endif /* QT_HIP_H */
