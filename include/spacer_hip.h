/* libspacer_hip.so -- C ABI of the MI355X-native SG-RLVR / GRPO hot path.
 *
 * The reference (OuyangKun10/SpaceR) has NO native boundary: every tensor op of its per-step hot path runs
 * inside third-party Python wheels (transformers / flash-attn / deepspeed) called from
 *   SpaceR-SG-RLVR/src/r1-v/src/open_r1/trainer/SG_RLVR_trainer.py   (abbrev. TR below).
 * This header is the boundary a maintainer binds instead (ctypes stub: INTEGRATION.md).  Each entry names
 * the reference call site whose arithmetic it replaces.
 *
 * Conventions (all entries):
 *   - return 0 (SPACER_OK) or a negative SPACER_E* code; spacer_last_error() gives a thread-local message
 *   - plain device pointers + sizes; no allocation, no ownership transfer, caller owns every buffer
 *   - asynchronous on `stream` (a hipStream_t passed as void*), no hidden synchronisation
 *   - bf16 = raw IEEE bfloat16 bits (uint16), row-major, innermost dimension contiguous unless an ld is given
 *   - no exceptions cross the boundary; re-entrant, no global mutable state, NO environment variables read: launch-plan
 *     switches (tests, A/B runs, CU budgets) travel in an explicit spacer_plan passed by the caller
 */
#ifndef SPACER_HIP_H
#define SPACER_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* spacer_stream_t; /* hipStream_t */

enum {
    SPACER_OK = 0,
    SPACER_EINVAL = -1,  /* bad argument / unsupported shape */
    SPACER_ELAUNCH = -2, /* HIP launch failure */
    SPACER_ENOMEM = -3,  /* caller workspace too small */
};

enum spacer_act { SPACER_ACT_NONE = 0, SPACER_ACT_QUICK_GELU = 1, SPACER_ACT_GELU_ERF = 2, SPACER_ACT_SILU = 3 };

const char* spacer_last_error(void);
int spacer_version(void);

/* ------------------------------------------------------------------------------------------------
 * GEMM.  C[M,N] = act(alpha * A[M,K] . B[N,K]^T + bias[N]) + residual[M,N]
 * Replaces every nn.Linear / Conv3d-as-GEMM the reference reaches through TR:357 (model forward) and
 * TR:463 (generate): ViT patch embed / qkv / proj / fc1 / fc2 / merger, LLM q,k,v,o,gate,up,down, lm_head.
 * K must be a multiple of 64 (pad the contraction dim with zeros); M, N arbitrary.
 * out_f32 selects the dtype of C and residual (0 = bf16, 1 = fp32).  residual == C gives C += ...
 * ---------------------------------------------------------------------------------------------- */
/* Launch-plan switches.  NULL (or a struct whose switches are all zero, with struct_bytes set) = the defaults the benchmark runs with.  The Python layer fills one from the
 * SPACER_* environment variables ONCE at import (spacer_amd/kernels.py:PLAN); the library itself never calls getenv. */
typedef struct spacer_plan {
    int struct_bytes;      /* = sizeof(spacer_plan) as the CALLER declares it.  The library reads a plan only when this equals its own
                            * sizeof(spacer_plan) and fails with SPACER_EINVAL otherwise: a binding that is a field short (round 4: the
                            * documented ctypes stub) is rejected instead of over-read.  tests/test_abi_layout.py pins the layouts */
    int gemm_tile;         /* 0 = cost model; 128 / 256 = force that tile kernel (tests run every shape through both) */
    int gemm_no_split;     /* 1 = no split-K tail in the 256-tile GEMM: one fp32 summation order per output (bit-exact comparisons) */
    int skinny_blocks;     /* decode GEMMs: target workgroups per launch; 0 = 2 x cus (one resident round); 1 = ONE K range per
                            * column group, i.e. no split-K atomics (bit-reproducible rollouts) */
    int skinny_no_balance; /* 1 = decode gate|up GEMM without the tail balance */
    int cus;               /* compute units the launch may count on; 0 = 256 (all of MI355X).  A caller that runs a second,
                            * CU-masked stream beside the decode loop passes the decode loop's share */
    int skinny_skew;       /* K-split decode GEMMs: 0 = the library's rule (equal K ranges, except a skew of alpha = 0.375 for launches that
                            * fill the resident slots with ranges of <= 2 slices); < 0 = always equal ranges; k + 1 = ranges with shares
                            * 1 + (k / 16) (2 r / (R - 1) - 1): k = 0 an even split, k > 0 skewed so that the ranges' atomic flushes do
                            * not all land when the weight stream ends */
} spacer_plan;

typedef struct spacer_gemm_epilogue {
    const void* bias;     /* bf16 [N] or NULL */
    const void* residual; /* [M, ldr] in the output dtype, or NULL */
    long ldr;
    int out_f32;
    int act;     /* enum spacer_act */
    float alpha; /* 0 is read as 1 */
    /* optional split-K workspace of spacer_gemm_workspace_bytes() bytes, 16-byte aligned, owned by the caller (scratch,
     * no initialisation needed); launches sharing a workspace must be ordered on one stream.  With it, the tiles of
     * the last partially filled round of CUs are cut along K and summed in a fixed order by a second small kernel.
     * NULL: the tail of the launch is not split. */
    void* workspace;
    long workspace_bytes;
    const spacer_plan* plan; /* NULL = defaults */
} spacer_gemm_epilogue;

int spacer_gemm_bf16_nt(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                        const spacer_gemm_epilogue* epi, spacer_stream_t stream);

/* The same GEMM with operands read in place from CONTRACTION-MAJOR arrays: trans_a -> A(m,k) = A[k*lda + m] (a [K, M] row-major
 * array), trans_b -> B(n,k) = B[k*ldb + n] ([K, N]).  What loss.backward() needs for every nn.Linear of the model the reference
 * trains (SG_RLVR_trainer.py:357 forward, HF Trainer.training_step backward):
 *     dX[T,in] = dY[T,out] . W[out,in]            trans_a = 0, trans_b = 1   (M = T,   N = in, K = out)
 *     dW[out,in] += dY[T,out]^T . X[T,in]         trans_a = 1, trans_b = 1   (M = out, N = in, K = T, any K >= 1: a ragged
 *                                                                             last K tile is masked in the kernel)
 * trans_a needs M % 8 == 0, trans_b needs N % 8 == 0; (trans_a = 1, trans_b = 0) is not instantiated.  Always the 256 tile. */
int spacer_gemm_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K, int trans_a,
                     int trans_b, const spacer_gemm_epilogue* epi, spacer_stream_t stream);

long spacer_gemm_workspace_bytes(void);

/* SwiGLU MLP input half in one launch (HF Qwen2MLP.forward: act_fn(gate_proj(x)) * up_proj(x), modeling_qwen2_vl.py Qwen2MLP;
 * Qwen2.5-VL vision MLP with biases): W bf16 [2*inter, K] = [gate_proj rows | up_proj rows], bias bf16 [2*inter] or NULL.
 *   act bf16 [M, inter] = silu(A.Wgate^T + bgate) * (A.Wup^T + bup);   gu bf16 [M, 2*inter] (or NULL) = the rounded gate|up.
 * Same bits as spacer_gemm_bf16_nt into gu + spacer_swiglu_fwd.  Always the 256-tile kernel: needs inter % 128 == 0 and K % 64 == 0
 * (SPACER_EINVAL otherwise); spacer_gemm_swiglu_fused(M, inter, K, plan) != 0 says whether the cost model WOULD put the problem on
 * that tile (large enough M) -- callers take the two-step path (GEMM + spacer_swiglu_fwd) when it returns 0.
 * The predicates (this one, spacer_gemm_pair_fused, spacer_gemm_pair_epilogue_fused, spacer_gemm_tile) never return a negative code:
 * a plan the library rejects (wrong struct_bytes) answers 0 -- "not fused" / "no tile" -- with spacer_last_error() set. */
int spacer_gemm_swiglu_fused(int M, int inter, int K, const spacer_plan* plan);
int spacer_gemm_swiglu_bf16(const void* A, long lda, const void* W, long ldb, const void* bias, void* act, long ld_act, void* gu,
                            long ld_gu, int M, int inter, int K, spacer_stream_t stream);

/* Which tile spacer_gemm_bf16_nt runs an [M,N,K] problem on: 256 (gemm_bf16_nt_256h_kernel) or 128 (gemm_bf16_nt_kernel).
 * Pure host function; profilers use it to attribute a launch to the kernel rocprof will name. */
int spacer_gemm_tile(int M, int N, int K, int have_workspace, const spacer_plan* plan);

/* Skinny GEMM for the decode loop (M <= 64 rows; <= 128 with packed weights; weights streamed once from HBM, split-K):
 * C32[M,N] += A[M,K] . B[N,K]^T  (fp32 atomics; C may be the fp32 residual stream itself).  K % 256 == 0.
 * epi must be NULL or {out_f32 = 1, residual = C} (its ``plan`` is honoured).  Replaces the per-token projections inside HF
 * generate's loop (TR:463). */
int spacer_gemm_skinny_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                            const spacer_gemm_epilogue* epi, spacer_stream_t stream);

/* Fragment-major weight copy for the decode loop: out = [N/16][K/32][64 lanes][8 bf16] so that every wave load of
 * the skinny GEMM is 1 KiB contiguous (N % 16 == 0, K % 32 == 0); rebuilt once per optimizer step. */
int spacer_pack_weight_frag(const void* W, long ld, void* out, int N, int K, spacer_stream_t stream);
int spacer_gemm_skinny_packed_bf16(const void* A, long lda, const void* Bpacked, void* C, long ldc, int M, int N, int K,
                                   const spacer_plan* plan, spacer_stream_t stream);
/* Store form, C32 = A . W^T (no accumulate, C needs no zero fill): only for wide N (>= 448 * 64 columns), where every workgroup
 * covers the whole K range -- the lm_head projection of the decode step (HF lm_head inside generate).  SPACER_EINVAL otherwise. */
int spacer_gemm_skinny_packed_store_bf16(const void* A, long lda, const void* Bpacked, void* C, long ldc, int M, int N, int K,
                                         const spacer_plan* plan, spacer_stream_t stream);

/* Decode-loop MLP front half in one launch:  Y[M, inter] (bf16) = silu(A . Wgate^T) * (A . Wup^T)   (HF Qwen2MLP's
 * act_fn(gate_proj(x)) * up_proj(x) inside generate, TR:463).  W = [gate (inter rows) | up (inter rows)] x K is packed
 * by spacer_pack_weight_frag_swiglu so each 16-column MFMA fragment holds 8 gate + 8 up columns of the same outputs;
 * every block owns the whole K, so there is no fp32 partial buffer and no separate SwiGLU kernel.  M <= 128. */
int spacer_pack_weight_frag_swiglu(const void* W, long ld, void* out, int inter, int K, spacer_stream_t stream);
int spacer_gemm_skinny_swiglu_bf16(const void* A, long lda, const void* Bpacked, void* Y, long ldy, int M, int inter, int K,
                                   spacer_stream_t stream);
/* The same with a caller-owned workspace of spacer_gemm_skinny_swiglu_workspace_bytes() bytes, ZERO-FILLED ONCE by the caller
 * (the kernel leaves it zeroed; launches sharing it must be ordered on one stream).  With it, when N/64 column groups leave a
 * short last round on the 512 resident workgroup slots (7B: 592), the tail groups are cut along K into small blocks that run
 * beside the whole-K blocks and meet through agent-scope atomics + a ticket (last arriver runs the SwiGLU epilogue).
 * Round 4: when the 2*inter/16 column fragments number between 4 and 5 per resident slot (2 x plan->cus; 7B: 2368 on 512) both
 * entries launch EXACTLY one resident round instead -- `fragments - 4 x slots` workgroups five fragments wide, the rest four; the
 * fifth fragment's K quarters are summed in LDS, no atomics, no workspace use, bit-reproducible (M <= 128). */
long spacer_gemm_skinny_swiglu_workspace_bytes(void);
int spacer_gemm_skinny_swiglu_bf16_ws(const void* A, long lda, const void* Bpacked, void* Y, long ldy, int M, int inter, int K,
                                      void* workspace, long workspace_bytes, const spacer_plan* plan, spacer_stream_t stream);

/* out[C, Rpad] = in[R, C]^T, zero-filling columns R..Rpad-1 (bf16).  Feeds the NT GEMM in backward. */
int spacer_transpose_bf16(const void* in, long ld_in, void* out, long ld_out, int R, int C, int Rpad,
                          spacer_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Normalisation (HF Qwen2VLRMSNorm / nn.LayerNorm reached through TR:357).
 * x may be bf16 or fp32 (x_f32); y is bf16.  rstd/mean are fp32 [rows] saved for backward (may be NULL).
 * ---------------------------------------------------------------------------------------------- */
int spacer_rmsnorm_fwd(const void* x, int x_f32, const void* w, void* y, float* rstd, int rows, int cols, float eps,
                       spacer_stream_t stream);
/* dx has the dtype of x (dx += ... when dx_accumulate); dw / db are fp32 [cols] and are ACCUMULATED into
 * (atomics), i.e. they can be the gradient buffers themselves. */
int spacer_rmsnorm_bwd(const void* x, int x_f32, const void* w, const void* dy, const float* rstd, void* dx,
                       int dx_accumulate, float* dw, int rows, int cols, spacer_stream_t stream);
int spacer_layernorm_fwd(const void* x, int x_f32, const void* w, const void* b, void* y, float* mean, float* rstd,
                         int rows, int cols, float eps, spacer_stream_t stream);
int spacer_layernorm_bwd(const void* x, int x_f32, const void* w, const void* dy, const float* mean,
                         const float* rstd, void* dx, int dx_accumulate, float* dw, float* db, int rows, int cols,
                         spacer_stream_t stream);
/* Same, with a caller-owned fp32 scratch (16-byte aligned; rows/16 * cols * 4 bytes, twice that for LayerNorm, is always
 * enough): dw / db are then reduced in two stages without atomics (deterministic, and faster: the atomic flush is a third of
 * the kernel at 5498 x 3584).  A workspace that is NULL or too small silently selects the atomic form. */
int spacer_rmsnorm_bwd_ws(const void* x, int x_f32, const void* w, const void* dy, const float* rstd, void* dx,
                          int dx_accumulate, float* dw, int rows, int cols, void* workspace, long workspace_bytes,
                          spacer_stream_t stream);
int spacer_layernorm_bwd_ws(const void* x, int x_f32, const void* w, const void* dy, const float* mean,
                            const float* rstd, void* dx, int dx_accumulate, float* dw, float* db, int rows, int cols,
                            void* workspace, long workspace_bytes, spacer_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Rotary embeddings.  cos/sin are fp32 tables [tokens, head_dim] (host builds them from the M-RoPE /
 * ViT position ids).  x is bf16 [tokens, heads, head_dim] in place: x = x*cos + rot_half(x)*sin in fp32.
 * inverse != 0 applies the transpose rotation (backward).
 * ---------------------------------------------------------------------------------------------- */
int spacer_rope_inplace(void* x, long token_stride, const float* cos_t, const float* sin_t, int tokens, int heads,
                        int head_dim, int inverse, spacer_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Attention.  Token-packed layout: q [T, Hq, D], k/v [T, Hkv, D] (bf16, strides in elements), o [T, Hq, D].
 * Work is described by "segments": segment s owns query rows [q_start, q_start+q_len) which attend
 *   (1) a shared prefix of keys [pre_start, pre_start+pre_len) with no mask, then
 *   (2) their own keys [q_start, q_start+q_len), causal when `causal` != 0 else full.
 * ViT per-frame attention (HF vision cu_seqlens) = segments with pre_len 0, causal 0, D = 80.
 * LLM prefill = one causal segment.  Shared-prefix scoring of K rollouts of one prompt = the prompt as a
 * causal segment + K segments with pre = the prompt (equal to K independent causal rows, TR:527).
 * lse (fp32 [Hq, T]) is written for backward.  D in {80, 128}.
 * ---------------------------------------------------------------------------------------------- */
typedef struct spacer_attn_segment {
    int q_start, q_len, pre_start, pre_len;
} spacer_attn_segment;

int spacer_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, long q_stride, long kv_stride,
                    long o_stride, const spacer_attn_segment* segs_dev, int num_segs, int max_q_len, int T, int Hq,
                    int Hkv, int D, int causal, float scale, spacer_stream_t stream);
/* Backward.  dq is bf16 in the layout of q (q_stride); dk/dv are fp32 [T, Hkv, D] contiguous accumulators
 * ZEROED BY THE CALLER (keys of a shared prompt collect contributions from several segments through
 * atomics); delta (fp32 [Hq, T]) is scratch.  Segments must partition the tokens by their own ranges. */
int spacer_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                    float* delta, void* dq, float* dk, float* dv, long q_stride, long kv_stride, long o_stride,
                    const spacer_attn_segment* segs_dev, int num_segs, int max_q_len, int T, int Hq, int Hkv, int D,
                    int causal, float scale, spacer_stream_t stream);

/* Decode attention: one new query token per sequence (B sequences), KV = shared prompt KV of the
 * sequence's prompt (prefix_k/v [n_prompts, Pmax, Hkv, D], length prefix_len[prompt_of[b]]) followed by the
 * sequence's own tail (tail_k/v [B, Cmax, Hkv, D], length tail_len_dev[0] + 1 -- read from device memory so
 * the step can be replayed from a hipGraph).  D = 128. */
int spacer_attn_decode(const void* q, const void* prefix_k, const void* prefix_v, const int* prefix_len,
                       const int* prompt_of, const void* tail_k, const void* tail_v, const int* tail_len_dev, void* o,
                       int B, int Pmax, int Cmax, int Hq, int Hkv, int D, float scale, spacer_stream_t stream);
/* Same result, but the prompt keys are scored ONCE per (prompt, kv head) for all Kn rollouts of the prompt (rows
 * b = prompt * Kn + k must be contiguous per prompt; Kn * Hq/Hkv <= 64) and merged with each rollout's tail keys:
 * decode attention is bound by per-CU load bandwidth, so not re-reading the prompt per rollout is ~3x faster.
 * workspace: spacer_attn_decode_workspace_bytes(n_prompts, Hkv) bytes of device memory. */
long spacer_attn_decode_workspace_bytes(int n_prompts, int Hkv);
int spacer_attn_decode_shared(const void* q, const void* prefix_k, const void* prefix_v, const int* prefix_len,
                              const int* prompt_of, const void* tail_k, const void* tail_v, const int* tail_len_dev, void* o,
                              void* workspace, int B, int Kn, int Pmax, int Cmax, int Hq, int Hkv, int D, float scale,
                              spacer_stream_t stream);
/* The same with PER-PROMPT rollout counts (round 6): prompt p owns the decode rows [row0[p], row0[p + 1]) (row0 int32 [n_prompts + 1] on the
 * device; at most Kmax rows per prompt, Kmax * Hq / Hkv <= 64).  The reference's T-GRPO branch generates G / 2 rollouts for the
 * frame-shuffled twin of a sample (TR:473 num_return_sequences = self.shuffled_num_generations): main + twin rollouts of 8 samples decode as
 * 8 x 8 + 8 x 4 = 96 rows instead of 128 (4.87 instead of 5.47 ms per token-step).  workspace as spacer_attn_decode_shared. */
int spacer_attn_decode_shared_rows(const void* q, const void* prefix_k, const void* prefix_v, const int* prefix_len,
                                   const int* prompt_of, const int* row0, const void* tail_k, const void* tail_v,
                                   const int* tail_len_dev, void* o, void* workspace, int B, int n_prompts, int Kmax, int Pmax,
                                   int Cmax, int Hq, int Hkv, int D, float scale, spacer_stream_t stream);

/* Decode-step helpers (all read the step / tail length from device memory so that one decode step can be
 * captured in a hipGraph and replayed):
 *   rope table: cos/sin fp32 [B, D] for text position pos_base[b] + *step_dev (M-RoPE rows coincide for text)
 *   qkv finish: acc32 [B,(Hq+2Hkv)*D] (+bias, rotary on q,k) -> q_out bf16 [B,Hq*D]; k,v appended to the tail
 *               cache [B,Cmax,Hkv,D] at position *tail_len_dev; acc32 is re-zeroed for the next layer
 *   swiglu_f32: y bf16 [B,inter] = silu(gate)*up from acc32 [B, 2*inter] (re-zeroed) */
/* Round 5 -- decode batches of <= 16 rows (one prompt group of K = 8 rollouts per GPU is the reference script's own launch shape,
 * run_SpaceR_SG_RLVR.sh:21,39): the gate|up + SwiGLU GEMM with the post-attention RMSNorm folded in.  (Whole-K workgroups for the
 * q|k|v / o / down projections of such a batch were built and measured slower than the K-split kernels: scripts/probes/
 * decode_rows16_probe.hip, profiles/r05_decode_small_rows.md.)
 *   normed SwiGLU: y bf16 [M, inter] = silu(rstd g) * (rstd u), [g | u] = bf16(x32) . Wp^T, rstd = rsqrt(mean_k x32^2 + eps): HF
 *                  post_attention_layernorm + gate_proj / up_proj + act_fn of one generate step for M <= 16 rows in ONE launch;
 *                  Wp = spacer_pack_weight_frag_swiglu of W diag(w_norm); workspace / plan as spacer_gemm_skinny_swiglu_bf16_ws. */
int spacer_gemm_skinny_swiglu_normed(const float* x32, long ldx, const void* Bpacked, void* Y, long ldy, int M, int inter, int K,
                                     float eps, void* workspace, long workspace_bytes, const spacer_plan* plan, spacer_stream_t stream);
int spacer_decode_rope_table(const int* pos_base, const int* step_dev, float theta, float* cos_t, float* sin_t, int B,
                             int D, spacer_stream_t stream);
int spacer_decode_qkv_finish(float* acc32, const void* bias, const float* cos_t, const float* sin_t, void* q_out,
                             void* tail_k, void* tail_v, const int* tail_len_dev, int B, int Hq, int Hkv, int D, int Cmax,
                             spacer_stream_t stream);
int spacer_swiglu_f32_fwd(float* acc32, void* y, int B, int inter, spacer_stream_t stream);
/* The decode q|k|v projection with the RMSNorm in front of it folded in (HF: input_layernorm + q/k/v_proj of one generate step):
 *   normed GEMM:   C32[M,N] += bf16(X32[M,K]) . Wp^T  and  rowss[m] += sum_k X32[m,k]^2   (M <= 64, K % 256 == 0, N % 16 == 0),
 *                  Wp = spacer_pack_weight_frag of W diag(w_norm) -- norm(x) W^T = rstd * (x (W diag(w))^T) exactly
 *   normed finish: spacer_decode_qkv_finish with every sum scaled by rstd[b] = rsqrt(rowss[b] / norm_cols + eps) first;
 *                  rowss_zero[0..B) (the sums of the NEXT normed GEMM) is cleared, B <= 256 */
int spacer_gemm_skinny_packed_normed(const float* X32, long ldx, const void* Bpacked, float* C32, long ldc, float* rowss, int M,
                                     int N, int K, const spacer_plan* plan, spacer_stream_t stream);
int spacer_decode_qkv_finish_normed(float* acc32, const void* bias, const float* cos_t, const float* sin_t, void* q_out,
                                    void* tail_k, void* tail_v, const int* tail_len_dev, const float* rowss, float* rowss_zero,
                                    int norm_cols, float eps, int B, int Hq, int Hkv, int D, int Cmax, spacer_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Element-wise pieces of the MLPs and residual stream.
 * ---------------------------------------------------------------------------------------------- */
/* y = silu(gate) * up ; gate/up are the two halves of gu [rows, 2*inter] (gate first) */
int spacer_swiglu_fwd(const void* gu, void* y, int rows, int inter, spacer_stream_t stream);
/* dgu from dy (bf16) and gu */
int spacer_swiglu_bwd(const void* gu, const void* dy, void* dgu, int rows, int inter, spacer_stream_t stream);
/* y = act(x) ; dx = dy * act'(x)   (act in {QUICK_GELU, GELU_ERF}) */
int spacer_act_fwd(const void* x, void* y, long n, int act, spacer_stream_t stream);
int spacer_act_bwd(const void* x, const void* dy, void* dx, long n, int act, spacer_stream_t stream);
/* db[cols] (fp32, +=) = column sums of dy [rows, cols] bf16 */
int spacer_bias_grad(const void* dy, long ld, float* db, int rows, int cols, spacer_stream_t stream);
/* out_bf16 = f32 (cast), and f32 accumulate/copy helpers for attention grads */
/* zero fill of `bytes` bytes on the stream (accumulators of the backward kernels, the flat gradient after the optimizer step) */
int spacer_zero(void* p, long bytes, spacer_stream_t stream);
int spacer_cast_f32_to_bf16(const float* in, void* out, long n, spacer_stream_t stream);
int spacer_cast_bf16_to_f32(const void* in, float* out, long n, spacer_stream_t stream);
/* out[i,:] = src[idx[i],:] (bf16) and dst[idx[i],:] += src[i,:] (bf16 -> fp32 atomics): select / scatter the hidden
 * rows whose logits the loss reads (TR:528 keeps only positions prompt_length-1 .. end). */
int spacer_gather_rows_bf16(const void* src, long ld, const int* idx, void* out, int n, int cols, spacer_stream_t stream);
int spacer_scatter_add_rows_f32(const void* src, const int* idx, float* dst, long ld, int n, int cols,
                                spacer_stream_t stream);
int spacer_cast_f32_to_bf16_strided(const float* in, long ld_in, void* out, long ld_out, int rows, int cols,
                                    spacer_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Embedding gather (+ video scatter) and its backward.  HF embed_tokens + masked_scatter (TR:357 path).
 * out fp32 [T, H] (the residual stream is fp32).  Tokens with ids[t] == video_token_id take row
 * video[vis_index++] instead (video rows are consumed in order).
 * ---------------------------------------------------------------------------------------------- */
int spacer_embed_fwd(const int64_t* ids, const void* table, const void* video, const int* video_row_of_token,
                     float* out, int T, int H, spacer_stream_t stream);
/* d_table (fp32 [V,H]) += rows of d_out for text tokens; d_video (fp32) = rows for video tokens */
int spacer_embed_bwd(const int64_t* ids, const int* video_row_of_token, const float* d_out, float* d_table,
                     float* d_video, int T, int H, spacer_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Frames -> normalised patch rows (HF Qwen2VLVideoProcessor rescale + CLIP normalise + patchify,
 * reached through TR:417-425).  frames uint8 [F, 3, Hpx, Wpx]; out bf16 [gt*gh*gw, Kpad] in
 * merge-block-major token order, feature order (c, tp, py, px), zero-padded from 1176 to Kpad.
 * ---------------------------------------------------------------------------------------------- */
int spacer_patchify(const uint8_t* frames, void* out, int F, int Hpx, int Wpx, int patch, int tpatch, int merge,
                    int Kpad, spacer_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Video front end before patchify (qwen_vl_utils/vision_process.py, reached through TR:406 process_vision_info):
 *   spacer_gather_frames_u8      out[f] = src[idx[f]]: the uniform frame sampling of :252 (idx = linspace().round()) on
 *                                decoded frames already in HBM; frame_bytes % 16 == 0, 16-byte aligned buffers.
 *   spacer_resize_bicubic_aa_u8  :310-315 `resize(video, [h, w], BICUBIC, antialias=True)` on uint8 [planes = F*3, H, W]
 *                                -> uint8 [planes, h, w]: torch's separable antialiased bicubic (horizontal pass, fp32
 *                                intermediate, vertical pass), rounded half-to-even to the uint8 grid and clamped, as
 *                                torchvision's tensor resize does for uint8 input.  Window / weight tables per axis
 *                                (xmin, xsize int32 [w]; wx fp32 [w, taps_x]; same for y) are the caller's
 *                                (vision_process.aa_tables restates aten's _compute_indices_weights_aa);
 *                                workspace = spacer_resize_workspace_bytes(planes, H, w) bytes of fp32 scratch.
 * ---------------------------------------------------------------------------------------------- */
int spacer_gather_frames_u8(const uint8_t* src, const int* idx, uint8_t* out, int n, long frame_bytes, spacer_stream_t stream);
long spacer_resize_workspace_bytes(int planes, int H, int w);
int spacer_resize_bicubic_aa_u8(const uint8_t* src, uint8_t* dst, int planes, int H, int W, int h, int w, const int* xmin,
                                const int* xsize, const float* wx, int taps_x, const int* ymin, const int* ysize,
                                const float* wy, int taps_y, float* workspace, spacer_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Per-token log-probs (TR:353-366) from fp32 logits rows: logp[r] = logits[r, tgt[r]] - logsumexp(logits[r,:]).
 * Also returns lse.  Backward writes dlogits = (onehot(tgt) - softmax) * g[r] (= g[r] * d logp / d logits) as bf16
 * into `dlogits` ([rows, ldd] bf16) for the lm_head backward GEMMs.
 * ---------------------------------------------------------------------------------------------- */
int spacer_logprob_fwd(const float* logits, long ld, const int64_t* targets, float* logp, float* lse, int rows,
                       int vocab, spacer_stream_t stream);
int spacer_logprob_bwd(const float* logits, long ld, const int64_t* targets, const float* lse, const float* g,
                       void* dlogits, long ldd, int rows, int vocab, spacer_stream_t stream);

/* The same over VOCABULARY CHUNKS: the lm_head GEMM runs over row ranges [col0, col0 + cols) of lm_head and each
 * chunk of fp32 logits [rows, cols] updates the running (max, sum-exp, target logit) of every row -- the [rows, vocab] logits of
 * TR:357-366 are never materialised.  first != 0 initialises the running state.  spacer_lse_finish: logp = target - lse.
 * spacer_logprob_bwd_chunk = spacer_logprob_bwd restricted to the chunk's columns (dlogits [rows, cols] bf16). */
int spacer_lse_chunk(const float* logits, long ld, const int64_t* targets, int col0, int cols, float* m_run, float* s_run,
                     float* t_run, int rows, int first, spacer_stream_t stream);
int spacer_lse_finish(const float* m_run, const float* s_run, const float* t_run, float* logp, float* lse, int rows,
                      spacer_stream_t stream);
int spacer_logprob_bwd_chunk(const float* logits, long ld, const int64_t* targets, int col0, const float* lse, const float* g,
                             void* dlogits, long ldd, int rows, int cols, spacer_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GRPO loss (TR:551-552, 640-643, 682): per-token k3 KL, clipped-free ratio loss, masked row means.
 * logp/ref_logp fp32 [G, C], mask int32 [G, C], adv fp32 [G].  Writes loss[0], mean_kl[0] and
 * dlogp fp32 [G, C] = d loss / d logp.  rows with an all-zero mask contribute NaN exactly as the reference.
 * ---------------------------------------------------------------------------------------------- */
int spacer_grpo_loss(const float* logp, const float* ref_logp, const float* adv, const int* mask, float beta,
                     float* loss, float* mean_kl, float* dlogp, int G, int C, spacer_stream_t stream);

/* First-EOS completion mask (TR:493-498). ids int64 [G, C] -> mask int32 [G, C], lengths int32 [G]. */
int spacer_completion_mask(const int64_t* ids, int eos_id, int* mask, int* lengths, int G, int C,
                           spacer_stream_t stream);
/* EOS-trimmed scoring (round 6; TR:493-498 builds the mask, TR:640-643 multiplies it into the loss: positions behind a rollout's
 * first EOS contribute exactly zero, so the scoring passes and the backward pack only the first lengths[i] tokens of rollout i).
 * idx int64 [n] = flat positions (i * C + t, t < lengths[i]) of the packed tokens inside the [G, C] rectangle:
 *   gather : dst[j] = src[idx[j]]   (d loss / d logp of the rectangle -> packed rows)
 *   scatter: dst[idx[j]] = src[j]   (packed log-probs -> the rectangle; the caller zeroes dst first) */
int spacer_gather_f32(const float* src, const int64_t* idx, float* dst, long n, spacer_stream_t stream);
int spacer_scatter_f32(const float* src, const int64_t* idx, float* dst, long n, spacer_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Sampling (HF generate with do_sample, temperature 1, top_k, top_p; TR:277-284).  logits fp32 [B, vocab].
 * One token per row into out_ids[b].  finished[b] != 0 rows emit pad_id.  Philox counter = (seed, step, b).
 * suppress_eos != 0 forbids eos_id (fixed-length throughput mode).
 * ---------------------------------------------------------------------------------------------- */
int spacer_sample_top_p(const float* logits, long ld, int B, int vocab, int top_k, float top_p, float temperature,
                        uint64_t seed, const int* step_dev, int eos_id, int pad_id, int suppress_eos, int* finished,
                        int64_t* out_ids, float* out_logp, void* workspace, long workspace_bytes,
                        spacer_stream_t stream);
long spacer_sample_workspace_bytes(int B, int vocab);
/* Decode-loop form: the Philox step is *step_dev + step_bias, and the token is also written to out_matrix[b * out_ld + step]
 * (the [B, C] completion matrix HF generate returns, TR:463) -- no per-step copy on the host side.  spacer_decode_embed is the
 * step's first launch: token-embedding gather of the previous step's tokens (fp32 rows) + advance of the device-side step
 * counters (*counter0 += 1, *counter1 += 1 when non-NULL), so a whole decode step is library launches only. */
int spacer_sample_top_p_step(const float* logits, long ld, int B, int vocab, int top_k, float top_p, float temperature, uint64_t seed,
                             const int* step_dev, int step_bias, int eos_id, int pad_id, int suppress_eos, int* finished,
                             int64_t* out_ids, int64_t* out_matrix, long out_ld, spacer_stream_t stream);
int spacer_decode_embed(const int64_t* ids, const void* table, float* out, int B, int H, int* counter0, int* counter1,
                        spacer_stream_t stream);
/* Round 6: spacer_sample_top_p / spacer_sample_top_p_step_ws with a workspace of spacer_sample_workspace_bytes(B, vocab) bytes (16-byte
 * aligned, scratch, no initialisation) run the two passes over a row's logits with several workgroups per row when the batch alone cannot
 * fill the chip (B < 256 rows of >= 64 K logits): 32.6 / 39.4 / 46.9 / 58.3 -> 30.4 / 30.7 / 34.3 / 47.2 us per call at 8 / 12 / 64 / 128 rows
 * (scripts/probes/sampler_time.py; what remains is the select / sort / draw chain of three dependent launches, not the passes).  Same token for the same logits and counter as the
 * one-workgroup-per-row form (the candidate list is sorted by (value, index) before the exact top-k trim); workspace NULL = that form. */
int spacer_sample_top_p_step_ws(const float* logits, long ld, int B, int vocab, int top_k, float top_p, float temperature, uint64_t seed,
                                const int* step_dev, int step_bias, int eos_id, int pad_id, int suppress_eos, int* finished,
                                int64_t* out_ids, int64_t* out_matrix, long out_ld, void* workspace, long workspace_bytes,
                                spacer_stream_t stream);
/* Synthetic completion lengths (the seeded variable-length mode of the benchmark / tests; SURVEY 8(d) "free-running mode"): on the
 * step's logits, right before spacer_sample_top_p_step with suppress_eos = 0 -- row b's EOS logit becomes -inf, except at token index
 * *step_dev + step_bias == eos_at[b] where it dominates the row, so rollout b ends with EOS as its eos_at[b]-th token (length
 * eos_at[b] + 1; eos_at[b] >= C: never).  Graph-capturable like the sampler (the step lives in device memory). */
int spacer_eos_schedule(float* logits, long ld, int B, int vocab, const int* step_dev, int step_bias, const int* eos_at, int eos_id,
                        spacer_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer: global grad-norm (sum of squares into acc[0]) and fused AdamW on fp32 master weights with
 * bf16 shadow refresh.  Replaces DeepSpeed's clip + AdamW step behind HF Trainer.training_step.
 * ---------------------------------------------------------------------------------------------- */
int spacer_sumsq_f32(const float* g, long n, float* acc, spacer_stream_t stream);
int spacer_adamw_step(float* master, void* shadow_bf16, float* m, float* v, const float* grad, long n, float lr,
                      float beta1, float beta2, float eps, float weight_decay, float bias_c1, float bias_c2,
                      const float* sumsq_dev, float max_norm, float grad_scale, spacer_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Precise scoring mode (csrc/precise.hip): per-token log-probs within 1e-3 of an fp32 evaluation at full depth (TR:353-366;
 * the north-star tolerance).  An activation is carried as a PAIR of bf16 arrays (hi = bf16(x), lo = bf16(x - hi)); a linear
 * layer is two accumulate passes of spacer_gemm_bf16_nt (A_hi, then A_lo with residual == C) into an fp32 C.  These entries
 * are the fp32 -> pair producers between the GEMMs and the attention on pair operands.
 * Round 4: the producers can also emit what the FAST path's backward kernels read -- norm statistics (mean / rstd), bf16
 * pre-activations (act / SwiGLU inputs), the attention log-sum-exp -- so that a training step (TR:527-552: the log-probs that
 * enter KL, loss and metrics) evaluates its forward in this mode and back-propagates through the production kernels on the hi
 * halves (spacer_attn_bwd, spacer_gemm_bf16 dX / dW, spacer_rmsnorm_bwd, ...).  All such outputs are optional (NULL).
 * ---------------------------------------------------------------------------------------------- */
/* A linear layer on a pair operand in ONE launch: C = (A_hi + A_lo) . B^T (+ bias, residual, alpha as spacer_gemm_bf16_nt; C fp32,
 * out_f32 = 1) over the K-concatenated operands [A_hi | A_lo] . [B | B]^T on the 256-tile kernel -- the fp32 output is written once
 * instead of written, re-read and re-written by a second accumulate pass.  A_hi / A_lo share lda.  Only problems the 256 tile takes
 * (spacer_gemm_pair_fused(M, N, K, have_workspace, plan) != 0); otherwise SPACER_EINVAL and the caller runs the two passes. */
int spacer_gemm_pair_fused(int M, int N, int K, int have_workspace, const spacer_plan* plan);
int spacer_gemm_bf16_pair_nt(const void* A_hi, const void* A_lo, long lda, const void* B, long ldb, void* C, long ldc, int M, int N,
                             int K, const spacer_gemm_epilogue* epi, spacer_stream_t stream);
/* Round 5: the pair GEMM with the PRODUCER that used to follow it in the epilogue -- the fp32 [M, N] tensor between them (1.67 GB per
 * decoder layer for gate|up at two cfg3 prompt groups, written by the GEMM and re-read by spacer_swiglu_f32_pair) is never written:
 *   SPACER_PAIR_SWIGLU  W = [gate rows | up rows] ([N = 2 inter, K]); (y_hi, y_lo) [M, inter] = pair(silu(g) * u) of
 *                       [g | u] = (A_hi + A_lo) . W^T + bias; tape_bf16 [M, 2 inter] (or NULL) = bf16(g | u)        HF Qwen2MLP.forward
 *   SPACER_PAIR_ROPE    (y_hi, y_lo) [M, N] = pair(x) with rotate_half rotary (fp32 tables rope_cos / rope_sin [M, 128]) applied to the
 *                       first rope_heads heads of x = (A_hi + A_lo) . W^T + bias; head_dim must be 128                HF apply_multimodal_rotary_pos_emb
 *   SPACER_PAIR_ACT     (y_hi, y_lo) [M, N] = pair(act(x)); tape_bf16 [M, N] (or NULL) = bf16(x)                       ViT fc1 + quick-GELU, merger GELU
 * Same sums as spacer_gemm_bf16_pair_nt and the same producer arithmetic as spacer_swiglu_f32_pair / spacer_rope_f32_pair /
 * spacer_act_f32_pair applied to them.  256-tile problems only: spacer_gemm_pair_epilogue_fused(kind, M, N, K, head_dim,
 * have_workspace, plan) != 0, else SPACER_EINVAL (callers then run the pair GEMM and the producer kernel).  workspace: the split-K scratch of
 * the spacer_gemm_epilogue struct, passed directly; may be NULL. */
enum spacer_pair_epilogue { SPACER_PAIR_SWIGLU = 2, SPACER_PAIR_ROPE = 3, SPACER_PAIR_ACT = 4 };
int spacer_gemm_pair_epilogue_fused(int kind, int M, int N, int K, int head_dim, int have_workspace, const spacer_plan* plan);
int spacer_gemm_bf16_pair_epilogue(int kind, const void* A_hi, const void* A_lo, long lda, const void* W, long ldb, const void* bias,
                                   void* y_hi, void* y_lo, long ld_y, void* tape_bf16, long ld_tape, const float* rope_cos,
                                   const float* rope_sin, int rope_heads, int head_dim, int act, int M, int N, int K, void* workspace,
                                   long workspace_bytes, const spacer_plan* plan, spacer_stream_t stream);
/* y_hi / y_lo [rows, ldy] bf16 <- x fp32 [rows, ldx] (cols, ldx, ldy multiples of 4) */
int spacer_split_f32_pair(const float* x, long ldx, void* y_hi, void* y_lo, long ldy, int rows, int cols, spacer_stream_t stream);
/* y = act(x) in fp32 (enum spacer_act), as a pair; pre_bf16 [rows, ldy] (or NULL) receives bf16(x), the point spacer_act_bwd
 * differentiates at */
int spacer_act_f32_pair(const float* x, long ldx, void* y_hi, void* y_lo, long ldy, int rows, int cols, int act, void* pre_bf16,
                        spacer_stream_t stream);
/* gu fp32 [rows, 2*inter] = [gate | up] -> silu(gate) * up as a pair [rows, inter]; gu_bf16 [rows, 2*inter] (or NULL) receives
 * bf16(gate | up) for spacer_swiglu_bwd */
int spacer_swiglu_f32_pair(const float* gu, void* y_hi, void* y_lo, int rows, int inter, void* gu_bf16, spacer_stream_t stream);
/* RMSNorm (layer == 0, b ignored) or LayerNorm (layer != 0) of fp32 rows, bf16 weights, output as a pair; mean_out (LayerNorm
 * only) / rstd_out fp32 [rows] (or NULL) receive the row statistics spacer_layernorm_bwd / spacer_rmsnorm_bwd take */
int spacer_norm_f32_pair(const float* x, const void* w, const void* b, void* y_hi, void* y_lo, int rows, int cols, float eps,
                         int layer, float* mean_out, float* rstd_out, spacer_stream_t stream);
/* rotary on the first rot_heads heads of fp32 rows [tokens, heads, head_dim] (ldx floats between tokens), the other heads copied;
 * output as a pair with token stride ldy */
int spacer_rope_f32_pair(const float* x, long ldx, const float* cos_t, const float* sin_t, void* y_hi, void* y_lo, long ldy,
                         int tokens, int rot_heads, int heads, int head_dim, spacer_stream_t stream);
/* spacer_embed_fwd with fp32 vision rows */
int spacer_embed_fwd_f32video(const int64_t* ids, const void* table, const float* video, const int* video_row_of_token, float* out,
                              int T, int H, spacer_stream_t stream);
/* spacer_attn_fwd on pair operands: S = Qh Kh + Qh Kl + Ql Kh, O = Ph Vh + Ph Vl + Pl Vh, fp32 softmax; O written as a pair.
 * lse fp32 [Hq, T] (or NULL): natural-log sum-exp of every query row, the layout spacer_attn_bwd reads. */
int spacer_attn_fwd_pair(const void* q_hi, const void* q_lo, const void* k_hi, const void* k_lo, const void* v_hi, const void* v_lo,
                         void* o_hi, void* o_lo, float* lse, long q_stride, long kv_stride, long o_stride,
                         const spacer_attn_segment* segs_dev, int num_segs, int max_q_len, int T, int Hq, int Hkv, int D, int causal,
                         float scale, int variant, spacer_stream_t stream);
/* variant: 0 = DMA-staged tiles, 256 query rows per workgroup (round 5); 1 = the register-staged round-3 kernel (A/B runs, tests).
 * Same MFMA order per wave: the two give the same bits. */

#ifdef __cplusplus
}
#endif
#endif /* SPACER_HIP_H */
