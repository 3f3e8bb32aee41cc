/*
 * vine_ppo.h — C ABI of the fused PPO-update ops (rows R1, R3 of SURVEY 8a) exported by libvine_hip.so.
 *
 * The reference runs these through rl_games 1.5.2 (absent; PY = cfg/train/Vine5LinkMovingBasePPO.yaml):
 *   - LSTM cell pointwise math of `rnn: lstm, units 256` (PY:31-37), stepped seq_len = 4 (PY:80) in the update and
 *     1 step at a time in the rollout, with the hidden state zeroed where `dones` is set;
 *   - the PPO loss of calc_gradients (in-tree text: isaacgymenvs/learning/common_agent.py:319-411, 427-435,
 *     482-516): clipped surrogate, clipped value loss, bound loss, entropy, KL.
 * GEMMs stay in hipBLASLt/rocBLAS (MFMA); these kernels fuse the surrounding pointwise work, which otherwise
 * costs ~300 tiny launches per optimiser step.
 *
 * All pointers are device pointers owned by the caller (PyTorch); calls enqueue on `stream` and never synchronise.
 * Returns 0 or a negative VineStatus (vine.h); message via vine_last_error().
 */
#ifndef VINE_PPO_H
#define VINE_PPO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One LSTM step for B sequences, hidden size H (gate order i, f, g, o as torch.nn.LSTM).
 *   gates = igates[b] + keep_b * hgates[b] + bias;  c' = f * (keep_b * c_prev) + i * g;  h' = o * tanh(c')
 * keep_b = 1 - done[b * done_stride] (done == NULL: keep = 1).
 * igates rows are `ig_stride` floats apart (a [B, T, 4H] tensor viewed at time t), h_out rows `h_stride` apart.
 * gates_act (nullable, [B,4H]) receives the activated gates for the backward pass.
 * hp_next (nullable, rows h_stride apart) receives (1 - done_next[b]) * h': the masked hidden state step t+1
 * consumes, i.e. the operand of the recurrent weight gradient, so the backward pass need not rebuild it; stored
 * as bfloat16 when hp_bf16 != 0 (mixed-precision update: GEMM operands in bf16, all arithmetic and state in fp32);
 * its rows are hp_stride elements apart (0 = h_stride).  gates_act uses the same storage type as hp_next (the stored
 * activations only feed the backward pass).
 * hgates == NULL: igates already holds x W_ih^T + h W_hh^T (the rollout runs ONE GEMM over a concatenated [x | h]
 * operand, whose h block is this call's hp_next). */
int vine_lstm_cell_forward(int64_t B, int64_t H, const float* igates, int64_t ig_stride, const float* hgates,
                           const float* bias, const float* c_prev, const uint8_t* done, int64_t done_stride,
                           float* h_out, int64_t h_stride, float* c_out, void* gates_act, void* hp_next,
                           const uint8_t* done_next, int64_t done_next_stride, int32_t hp_bf16, int64_t hp_stride,
                           void* stream);

/* The same LSTM step with the recurrent GEMM fused in (matrix cores, bfloat16 operands, fp32 accumulation):
 *   gates = A W^T + igates + bias   with A [B, K] bf16 (rows lda apart) and W [4H, K] bf16 (rows ldw apart),
 * then the pointwise update exactly as vine_lstm_cell_forward; the [B, 4H] pre-activations never reach HBM.
 * A2 (nullable): second operand source; columns [0, K1) of the product's K come from A, [K1, K) from A2 (the update
 * keeps the layer input x_t and the masked state h_{t-1} in separate buffers; W is then [w_ih | w_hh] and no input
 * projection is ever materialised).  Two-source form: K - K1 == 256, K1 in {32, 64, 96, 128}.
 * igates is nullable (no separate input projection when x is part of the operand).  gates_act and
 * hp_next are bfloat16.  Needs B % 64 == 0, H % 16 == 0, K in {128, 256, 288, 320, 352, 384, 512}; otherwise
 * VINE_ERR_UNSUPPORTED (callers fall back to GEMM + vine_lstm_cell_forward). */
int vine_lstm_step_mfma(int64_t B, int64_t H, int64_t K, const void* A, int64_t lda, const void* A2, int64_t lda2,
                        int64_t K1, const void* W, int64_t ldw, const float* igates, int64_t ig_stride,
                        const float* bias, const float* c_prev, const uint8_t* done, int64_t done_stride, float* h_out,
                        int64_t h_stride, float* c_out, void* gates_act, void* hp_next, const uint8_t* done_next,
                        int64_t done_next_stride, int64_t hp_stride, void* stream);

/* Persistent LSTM forward over a whole (short) sequence, mixed precision, ONE launch for all T steps: a workgroup owns
 * 32 sequences, keeps h_t (bf16, LDS) and c_t (fp32, registers) on chip between the steps and streams the weights from
 * a fragment-ordered copy (vine_lstm_tile_weights) straight into the MFMA operand registers.  Same arithmetic, same
 * accumulation order and same outputs as T calls of vine_lstm_step_mfma with the [x | h] operand.
 *   x        [B*T, ldx] bf16, row = seq * T + t; its first KX columns are the step's input block (zero-padded to a
 *            multiple of 32: KX in {32, 64, 96, 128})
 *   hp       [B, T, H] bf16 (rows hp_stride apart): slot 0 holds the masked initial state (read), slots 1 .. T-1
 *            receive the masked h_{t-1} each later step consumed (the recurrent weight gradient's operand)
 *   w_tiled  vine_lstm_tile_weights(H, KX + H, [w_ih | 0 | w_hh], ...) -- [4H, KX + H] bf16 in fragment order
 *   bias [4H] fp32 (b_ih + b_hh), c0 [B, H] fp32, done [B*T] uint8 (nullable)
 *   h_out [B*T, H] fp32, c_all [T+1, B, H] fp32 (slots 1 .. T written), gates [T, B, 4H] bf16 (nullable)
 *   c_bf16 != 0: the saved cell states are only ever read by the backward pass and are stored as bfloat16 (c_all
 *            [T+1, B, H] bf16, slots 1 .. T-1 written) -- the recurrence itself runs on fp32 registers either way -- and
 *            the final state c_T goes to c_last [B, H] fp32 (the LSTM state handed back to the caller)
 *   h0 (nullable) [B, H] fp32: the initial hidden state; the kernel then forms the masked bf16 copy itself
 *            (h0 * (1 - done[b, 0])) and WRITES it to slot 0 of hp instead of reading it from there
 *   c_bf16 bit 1 ("h once"; needs bit 0 and h0): h_out is [B, T + 1, H] in the 16-bit format, UNMASKED, and is the only
 *            copy of the hidden states -- slot 0 receives h0, slot t + 1 the state h_t; vine_ln_heads_loss reads slots
 *            1 .. T (dx_bf16 bit 1 + T in bits 8-15) and vine_weight_grad_cat_seq_mfma slots 0 .. T - 1 with done[k] applied
 *            as the recurrent weight gradient's operand; hp is not used (may be NULL)
 * Needs B % 32 == 0, H == 256, T <= 8; VINE_ERR_UNSUPPORTED otherwise (callers use the per-step kernels). */
int vine_lstm_seq_forward_mfma(int64_t B, int64_t T, int64_t H, int64_t KX, const void* x, int64_t ldx, void* hp,
                               int64_t hp_stride, const void* w_tiled, const float* bias, const float* c0,
                               const uint8_t* done, void* h_out, void* c_all, void* gates, int32_t c_bf16, float* c_last,
                               const float* h0, void* stream);

/* Fragment-ordered copy of an LSTM weight for the persistent kernels (H == 256; dst: H * K bf16 elements... K columns
 * of all 4H rows for the forward form).  transposed = 0: src [4H, ld] row-major with K = 32 * ksteps columns used
 * (forward operand [w_ih | 0 | w_hh]).  transposed = 1: src = w_hh [4H, ld >= H]; the copy serves the backward
 * product dG w_hh (reduction over K = 4H, output units H). */
int vine_lstm_tile_weights(int64_t H, int64_t K, const void* src, int64_t ld, int32_t transposed, void* dst, void* stream);

/* Backward twin of vine_lstm_seq_forward_mfma: all T steps of vine_lstm_step_backward_mfma in ONE launch (dG_{t+1} in
 * LDS, dc / c in registers).  g_out [B*T, H] fp32 (row = seq * T + t), w_hh_tiled = vine_lstm_tile_weights(H, 4H,
 * w_hh, ld, 1, ...), gates [T, B, 4H] bf16 and c_all [T+1, B, H] fp32 as the forward pass left them (slot 0 of c_all
 * is never read: c0 [B, H] is), done [B*T] uint8 (nullable); dgates [B*T, 4H] bf16 out; bias_partial (nullable)
 * [B / 32, 4H] fp32: its column sums are the bias gradient.  c_bf16 / c_last: as left by the forward kernel; g_bf16 != 0:
 * g_out holds bfloat16 (the gradient w.r.t. the hidden states as vine_ln_heads_loss writes it with dx_bf16).
 * Needs B % 32 == 0, H == 256, T <= 8. */
int vine_lstm_seq_backward_mfma(int64_t B, int64_t T, int64_t H, const void* g_out, const void* w_hh_tiled,
                                const void* gates, const void* c_all, const float* c0, const uint8_t* done,
                                void* dgates, float* bias_partial, int32_t c_bf16, const float* c_last, int32_t g_bf16,
                                void* stream);

/* vine_lstm_seq_backward_mfma and vine_mlp3_bwd_elu_mfma as two PHASES of one launch (round 4): workgroup b owns sequences
 * [32 b, 32 b + 32) in both -- rows [128 b, 128 b + 128) of the sequence-major sample order at T = 4 -- and the MLP phase
 * reads the gate gradients its own LSTM phase wrote; only a workgroup barrier separates them.  T == 4, H == 256, 16-bit
 * saved cell states (c_T in c_last) and hidden-state gradient; arguments as in the two functions (K0 = 4H, C3 / C2 / C1 =
 * 64 / 128 / 256, dgates rows 4H apart). */
int vine_lstm_seq_backward_mlp3_mfma(int64_t B, int64_t T, int64_t H, const void* g_out, const void* w_hh_tiled,
                                     const void* gates, const void* c_all, const float* c0, const uint8_t* done,
                                     void* dgates, float* bias_partial, const float* c_last, const void* wt0, int64_t ldw0,
                                     const void* wt1, int64_t ldw1, const void* wt2, int64_t ldw2, const void* a3,
                                     int64_t a3_stride, const void* a2, const void* a1, float alpha, void* gz3, void* gz2,
                                     void* gz1, float* part3, float* part2, float* part1, void* stream);

/* The row-local chain of one optimiser step as four PHASES of one launch (round 4): vine_lstm_seq_forward_mfma ("h once"
 * form) -> vine_ln_heads_loss (its loss rows left for vine_column_sums_batched_fin: VineLossFinalize) ->
 * vine_lstm_seq_backward_mfma -> vine_mlp3_bwd_elu_mfma.  Workgroup b owns sequences [32 b, 32 b + 32) = samples
 * [128 b, 128 b + 128) in all four and every phase reads from its predecessor only what the same workgroup wrote, so a
 * workgroup barrier separates them.  The update's default configuration only: T == 4, H == 256, x block 96 wide
 * (rows ldx >= 96 apart), NH == 3 heads, 16-bit hidden states / saved cell states / gate activations / dh / dG, B % 32 == 0; field
 * meanings as the arguments of the four functions (x: the LSTM operand rows written by vine_mlp3_elu_mfma*; h_out
 * [B, T + 1, 256]; d_out [B T, 256]; dgates [B T, 1024]; a3 = the MLP block of x, rows a3_stride apart). */
typedef struct VineTrunkArgs {
    int64_t B, T;
    const void* x; int64_t ldx; const void* w_tiled; const float* bias; const float* c0; const float* h0; const uint8_t* done;
    void* h_out; void* c_all; void* gates; float* c_last;
    const float* ln_gamma; const float* ln_beta; float ln_eps; int32_t clip_value;
    const float* w_heads; const float* b_heads; const float* logstd; const float* actions; const float* old_neglogp;
    const float* advantages; const float* old_values; const float* returns; const float* old_mu; const float* old_sigma;
    float e_clip, critic_coef, entropy_coef, bounds_coef, soft_bound, alpha;
    float* heads; void* d_out; float* ln_partial; float* loss_partial; float* stats; float* grad_logstd; float* grad_mu_bias;
    float* grad_value_bias; float* kl_out; float* logstd_grad_accum; float* mu_store; float* sigma_store;
    const float* loss_scale; float* found_inf;
    const void* w_hh_tiled; void* dgates; float* bias_partial;
    const void* wt0; int64_t ldw0; const void* wt1; int64_t ldw1; const void* wt2; int64_t ldw2; const void* a3; int64_t a3_stride;
    const void* a2; const void* a1;
    void* gz3; void* gz2; void* gz1; float* part3; float* part2; float* part1;
} VineTrunkArgs;
int vine_trunk_phases(const VineTrunkArgs* args, void* stream);
int64_t vine_trunk_args_size(void);      /* sizeof(VineTrunkArgs): checked against the host mirror */

/* Linear + bias + ELU on the matrix cores: out = elu(A W^T + bias) with A [n, K] bf16 (rows lda apart), W [N, K] bf16,
 * out [n, N] bf16 (rows out_stride apart, e.g. a column block of the LSTM operand buffer); the fp32 pre-activation is
 * never stored.  Needs n % 64 == 0, N % 64 == 0, K in {32, 64, 128, 256}; otherwise VINE_ERR_UNSUPPORTED (callers fall
 * back to GEMM + vine_bias_elu). */
int vine_linear_elu_mfma(int64_t n, int64_t N, int64_t K, const void* A, int64_t lda, const void* W, int64_t ldw,
                         const float* bias, float alpha, void* out, int64_t out_stride, void* stream);

/* The MLP [32 -> C1 -> C2 -> C3] (+ ELU after every layer) of the default network in ONE launch, activations carried in
 * registers from layer to layer: x [n, 32] bf16 rows ldx apart (normalised observations + zero pad columns), w1p [C1, 32]
 * (zero-padded), w2 [C2, C1], w3 [C3, C2] bf16, biases fp32; out [n, C3] bf16 rows out_stride apart; act1 [n, C1] /
 * act2 [n, C2] bf16 (packed, nullable): the intermediate activations for the backward pass.  With `raw` (fp32
 * observations [n, F_in], F_in <= 32) the kernel first forms x = clamp((raw - mean) / sqrt(var + eps), +-clip) from the
 * float64 RunningMeanStd statistics (vine_normalize_obs' arithmetic), zero-pads it to 32 columns and WRITES it to x.
 * Covers C1 = 256, C2 = 128, C3 = 64, n % 64 == 0 (VINE_ERR_UNSUPPORTED otherwise: use vine_linear_elu_mfma per layer). */
int vine_mlp3_elu_mfma(int64_t n, void* x, int64_t ldx, const float* raw, int64_t F_in, const double* mean, const double* var,
                       float eps, float clip, const void* w1p, const float* b1, int64_t C1, const void* w2, int64_t ldw2,
                       const float* b2, int64_t C2, const void* w3, int64_t ldw3, const float* b3, int64_t C3, float alpha,
                       void* act1, void* act2, void* out, int64_t out_stride, void* stream);

/* vine_mlp3_elu_mfma with the optimiser step's operand preparation riding in the same launch (round 4): `njobs` moves in
 * the vocabulary of vine_copy_batched (same arrays, same meaning) are executed by the workgroups of this kernel as a side
 * job -- they must build nothing this kernel reads; the consumers are the launches behind it.  w1 [C1, ldw1]: ldw1 == 32 =
 * the zero-padded w1p of vine_mlp3_elu_mfma; ldw1 < 32 (even) = the weight as stored, [C1, ldw1] packed, padded to 32
 * columns on its way into LDS.  njobs == 0: vine_mlp3_elu_mfma. */
int vine_mlp3_elu_mfma_prep(int64_t n, void* x, int64_t ldx, const float* raw, int64_t F_in, const double* mean,
                            const double* var, float eps, float clip, const void* w1p, int64_t ldw1, const float* b1,
                            int64_t C1, const void* w2, int64_t ldw2, const float* b2, int64_t C2, const void* w3, int64_t ldw3,
                            const float* b3, int64_t C3, float alpha, void* act1, void* act2, void* out, int64_t out_stride,
                            int32_t njobs, const int32_t* op, const int32_t* elem, const void* const* src,
                            const void* const* src2, void* const* dst, const int64_t* rows, const int64_t* cols,
                            const int64_t* src_stride, const int64_t* dst_stride, const int64_t* aux, void* stream);

/* Backward of that MLP in ONE launch:  gz3 = (dG wt0^T) * elu'(a3),  gz2 = (gz3 wt1^T) * elu'(a2),  gz1 = (gz2 wt2^T) * elu'(a1)
 * with dG [n, K0] (the LSTM gate gradients), wt0 [C3, K0] = the MLP block of w_ih transposed, wt1 [C2, C3] = W3^T,
 * wt2 [C1, C2] = W2^T, a3 [n, C3] (rows a3_stride apart) / a2 [n, C2] / a1 [n, C1] the stored ELU outputs, all bf16;
 * gz3 / gz2 / gz1 packed bf16 (the operands of the three weight gradients); part3 / part2 / part1
 * [n / rows_per_workgroup, C] fp32: one row of column sums of each gz per workgroup (partial bias gradients;
 * rows_per_workgroup = 128 when n % 128 == 0 and n >= 32768, else 64).  Covers C3 = 64, C2 = 128, C1 = 256, K0 = 1024,
 * n % 64 == 0 (VINE_ERR_UNSUPPORTED otherwise: vine_linear_bwd_elu_mfma per layer). */
int vine_mlp3_bwd_elu_mfma(int64_t n, const void* dG, int64_t lddg, int64_t K0, const void* wt0, int64_t ldw0, const void* wt1,
                           int64_t ldw1, const void* wt2, int64_t ldw2, const void* a3, int64_t a3_stride, const void* a2,
                           const void* a1, int64_t C3, int64_t C2, int64_t C1, float alpha, void* gz3, void* gz2, void* gz1,
                           float* part3, float* part2, float* part1, void* stream);

/* Backward of a Linear through the previous layer's ELU, on the matrix cores:
 *   gz = (G Wt^T) * elu'(a)   with G [n, K] bf16 (gradient w.r.t. this layer's pre-activation), Wt [N, K] bf16 = the
 *   layer's weight TRANSPOSED (N = its input width), a [n, N] bf16 = the previous layer's ELU output, gz [n, N] bf16;
 * partial (nullable, [n / 64, N] fp32): per-workgroup column sums of gz = partial bias gradient of the previous layer.
 * Needs n % 64 == 0, N % 64 == 0, K in {64, 128, 256} (weight slab resident in LDS) or {512, 1024} (both operands
 * streamed in 128-wide k chunks); otherwise VINE_ERR_UNSUPPORTED (GEMM + vine_elu_backward). */
int vine_linear_bwd_elu_mfma(int64_t n, int64_t N, int64_t K, const void* G, int64_t ldg, const void* Wt, int64_t ldw,
                             const void* a, int64_t a_stride, float alpha, void* gz, int64_t gz_stride, float* partial,
                             void* stream);

/* Backward of the step above.
 *   dh = g_out[b] (rows g_stride apart) + keep_next_b * g_rec[b];   dc = keep_next_b * dc_next[b] + dh * o * (1 - tanh(c)^2)
 * g_rec / dc_next = gradients w.r.t. the MASKED (h_t, c_t) consumed by step t+1 (NULL at the last step),
 * keep_next = 1 - done_next.  Writes the pre-activation gate gradients (rows dg_stride apart) and the gradient
 * w.r.t. the masked c_{t-1} (dc_prev = dc * f).
 * bias_partial (nullable, [VINE_PPO_PARTIAL_BLOCKS, 4H]) receives per-workgroup column sums of the gate gradients;
 * bias_partial_prev (nullable, same shape): the rows another time step's call wrote; they are added in, so that after
 * chaining the T calls the last buffer alone holds the partial sums of the whole sequence.  The column sums of
 * that buffer are the bias gradient -- deterministic, no atomics.
 * dgates_bf16 != 0: the gate gradients (only ever GEMM operands) are stored as bfloat16, and gates_act is read as
 * bfloat16 (as the forward call with hp_bf16 != 0 wrote it). */
#define VINE_PPO_PARTIAL_BLOCKS 512
int vine_lstm_cell_backward(int64_t B, int64_t H, const float* g_out, int64_t g_stride, const float* g_rec,
                            const float* dc_next, const uint8_t* done_next, int64_t done_next_stride,
                            const void* gates_act, const float* c_new, const float* c_prev, const uint8_t* done,
                            int64_t done_stride, void* dgates, int64_t dg_stride, float* dc_prev,
                            float* bias_partial, const float* bias_partial_prev, int32_t dgates_bf16, void* stream);

/* The same backward step with the recurrent input gradient computed inside (mixed precision only):
 *   g_rec[b] = dgates_next[b, :] (bf16, rows dgn_stride apart: the gate gradients the call for step t+1 wrote) times
 *              w_hh, given TRANSPOSED as w_hh_t [H, 4H] bf16 (rows ldw apart),
 * on the matrix cores, then exactly the arithmetic above; gates_act / dgates are bfloat16.  dgates_next == w_hh_t ==
 * NULL at the last time step.  bias_partial / bias_partial_prev: [B / 64, 4H] rows, chained as above.
 * Requires B % 64 == 0 and H in {128, 256}; VINE_ERR_UNSUPPORTED otherwise (callers then use the two-launch form:
 * GEMM + vine_lstm_cell_backward). */
int vine_lstm_step_backward_mfma(int64_t B, int64_t H, const float* g_out, int64_t g_stride, const void* dgates_next,
                                 int64_t dgn_stride, const void* w_hh_t, int64_t ldw, const float* dc_next,
                                 const uint8_t* done_next, int64_t done_next_stride, const void* gates_act,
                                 const float* c_new, const float* c_prev, const uint8_t* done, int64_t done_stride,
                                 void* dgates, int64_t dg_stride, float* dc_prev, float* bias_partial,
                                 const float* bias_partial_prev, void* stream);

/* Weight gradient of a linear layer on the matrix cores (mixed precision): for each of `slices` equal row slices
 *   part[s][m][n] = sum over the rows k of slice s of dy[k][m] * x[k][n],   m < M, n < Nv
 * dy [rows, M] and x [rows, Np] bfloat16 (rows ldy / ldx elements apart, 16-B aligned), part [slices, M, Nv] fp32.
 * Np = the tile width actually read from x (32, 96 or a multiple of 128; columns Nv .. Np-1 must be readable, their
 * products are not stored).  The sum over the slices (vine_column_sums over part viewed as [slices, M * Nv]) is the
 * gradient dy^T x -- fixed summation order, no atomics.  Requires M % 64 == 0 and rows % (32 * slices) == 0;
 * VINE_ERR_UNSUPPORTED otherwise (callers fall back to a library GEMM). */
int vine_weight_grad_mfma(int64_t rows, int64_t M, int64_t Np, int64_t Nv, const void* dy, int64_t ldy, const void* x,
                          int64_t ldx, int64_t slices, float* part, void* stream);

/* Weight gradient(s) dy^T [x1 | x2] in ONE pass over dy on the matrix cores, for each of `slices` equal row slices:
 *   part1[s][m][n] = sum_k dy[k][m] x1[k][n] (n < Nv1),   part2[s][m][n] = sum_k dy[k][m] x2[k][n] (n < Nv2)
 * dy [rows, M], x1 [rows, N1p] (N1p columns read, Nv1 <= N1p stored; N1p = 0: no first operand), x2 [rows, N2p] (Nv2 <= N2p
 * stored), all bfloat16 with 16-B aligned rows (ldy / ldx1 / ldx2 elements apart); part1 [slices, M, Nv1], part2
 * [slices, M, Nv2] fp32.  The sums over the slices (vine_column_sums) are the gradients: fixed order, no atomics.
 * The LSTM of the default network: dy = dG [n, 4H], x1 = the 96-column step-input block (92 stored), x2 = the masked hidden
 * states (256): dW_ih and dW_hh from one read of dG.  NT = output-tile width / 16, one of {11, 8, 2} (64-row tiles);
 * requires M % 64 == 0, N1p % 16 == 0, (N1p + N2p) % (16 NT) == 0, slices % 8 == 0, rows % (32 slices) == 0.  NT = 22
 * selects ONE 128 x 352 tile per workgroup for exactly N1p = 96, N2p = 256 (M % 128 == 0, rows % (64 slices) == 0), the
 * faster choice for the LSTM.  VINE_ERR_UNSUPPORTED otherwise (callers fall back to library GEMMs). */
int vine_weight_grad_cat_mfma(int64_t rows, int64_t M, const void* dy, int64_t ldy, const void* x1, int64_t ldx1, int64_t N1p,
                              int64_t Nv1, const void* x2, int64_t ldx2, int64_t N2p, int64_t Nv2, int64_t NT, int64_t slices,
                              float* part1, float* part2, void* stream);

/* The LSTM's two weight gradients with the recurrent operand formed on the fly from the ONE copy of the hidden states that
 * vine_lstm_seq_forward_mfma writes with c_bf16 bit 1: operand row k = seq * T + t of the second product is
 * (1 - done[k]) * h_{t-1} = slot t of sequence seq in h_all [rows / T, T + 1, ldh] (16-bit, unmasked; slot 0 = the
 * initial state).  x1 = the 96-column step-input block (Nv1 stored), 256 hidden units (Nv2 stored); NT = 22 | 21 as in
 * vine_weight_grad_cat_mfma (the wide tile only); needs 32 % T == 0 and done 16-B aligned.  Same partial sums, bit for
 * bit, as vine_weight_grad_cat_mfma on the masked [rows, 256] tensor. */
int vine_weight_grad_cat_seq_mfma(int64_t rows, int64_t M, const void* dy, int64_t ldy, const void* x1, int64_t ldx1,
                                  int64_t Nv1, const void* h_all, int64_t ldh, const uint8_t* done, int64_t T, int64_t Nv2,
                                  int64_t NT, int64_t slices, float* part1, float* part2, void* stream);

/* Up to VINE_WEIGHT_GRAD_MAX_GROUP products of the small-tile family above (NT in {11, 8, 2}) in ONE launch: problem k
 * takes element k of every array, with the argument meaning of vine_weight_grad_cat_mfma (x1[k] / part1[k] unused when
 * N1p[k] = 0).  The three MLP weight gradients of the update are ready at the same time and share one launch. */
#define VINE_WEIGHT_GRAD_MAX_GROUP 6
int vine_weight_grad_group(int32_t nprob, const int64_t* rows, const int64_t* M, const void* const* dy, const int64_t* ldy,
                           const void* const* x1, const int64_t* ldx1, const int64_t* N1p, const int64_t* Nv1,
                           const void* const* x2, const int64_t* ldx2, const int64_t* N2p, const int64_t* Nv2, const int64_t* NT,
                           const int64_t* slices, float* const* part1, float* const* part2, void* stream);

/* LayerNorm over the last dimension (rl_games `rnn.layer_norm: True`, PY:36; torch.nn.LayerNorm arithmetic: biased
 * variance, eps inside the square root).  H in {256, 512, 1024}; one 64-lane wave per row.
 * forward: y = (x - mean) * rstd * gamma + beta; mean/rstd [n] (both nullable) are kept for the backward pass.
 * backward: dx, and partial [VINE_PPO_PARTIAL_BLOCKS, 2H] whose row sum is {d gamma | d beta}. */
int vine_layernorm_forward(int64_t n, int64_t H, const float* x, const float* gamma, const float* beta, float eps,
                           float* y, float* mean, float* rstd, void* stream);
int vine_layernorm_backward(int64_t n, int64_t H, const float* dy, const float* x, const float* mean, const float* rstd,
                            const float* gamma, float* dx, float* partial, void* stream);

/* LayerNorm followed by NH output heads (rows of w [NH, H], bias wb [NH]; NH = actions + 1: [mu | value]) in one pass,
 * H == 256 and 2 <= NH <= 5 (else VINE_ERR_UNSUPPORTED: use the separate kernels and GEMMs):
 *   forward : heads[r] = w LN(x_r) + wb, LN(x) is never written; mean/rstd [n] kept for the backward pass;
 *   backward: from g = d loss / d heads [n, NH]: dx [n, H] and partial [VINE_PPO_PARTIAL_BLOCKS, (2 + NH) H] whose
 *             column sums are { d gamma | d beta | d w[0] | ... | d w[NH-1] }. */
int vine_layernorm_heads_forward(int64_t n, int64_t H, int64_t NH, const float* x, const float* gamma, const float* beta,
                                 float eps, const float* w, const float* wb, float* heads, float* mean, float* rstd,
                                 void* stream);
int vine_layernorm_heads_backward(int64_t n, int64_t H, int64_t NH, const float* g, const float* x, const float* mean,
                                  const float* rstd, const float* gamma, const float* beta, const float* w, float* dx,
                                  float* partial, void* stream);

/* ELU backward from the layer OUTPUT a = elu(z) (PY:19 `activation: elu`): out = g * (a > 0 ? 1 : a + alpha); rows of
 * g / a / out are *_stride floats apart (so a column block of a wider matrix works); out may alias g.
 * partial (nullable, [VINE_PPO_PARTIAL_BLOCKS, C]): per-workgroup column sums of out = the bias gradient of the
 * Linear in front of the activation.  C % 4 == 0 and C/4 divides 256.
 * a_bf16 / out_bf16: `a` / `out` hold bfloat16 (the mixed-precision update keeps activations only as GEMM operands). */
int vine_elu_backward(int64_t n, int64_t C, const float* g, int64_t g_stride, const void* a, int64_t a_stride,
                      float alpha, void* out, int64_t out_stride, float* partial, int32_t a_bf16, int32_t out_bf16,
                      void* stream);

/* Column sums of src [R, C] (rows row_stride floats apart), deterministic and without scratch memory: finishes the
 * per-workgroup partial sums of the kernels above and the slices of the split-K weight gradients.
 * out1 == NULL: out0[0:C] = sums.  out1 != NULL, dup == 0: columns [0, n0) -> out0, [n0, C) -> out1.
 * dup != 0: all C sums to both out0 and out1 (the two LSTM bias vectors share one gradient). */
int vine_column_sums(int64_t R, int64_t C, const float* src, int64_t row_stride, float* out0, int64_t n0, float* out1,
                     int32_t dup, void* stream);

/* Generalised advantage estimation over a rollout stored [T, N] (row R2; rl_games discount_values, next-nonterminal
 * form; the in-tree AMP variant is isaacgymenvs/learning/common_agent.py:413-425): one launch instead of T x 6.
 * dones[t] is the flag stored with step t (the env finished an episode at step t-1), last_dones the flags after the
 * last step; returns (nullable) = advs + values. */
int vine_gae(int32_t T, int64_t N, const float* rewards, const float* values, const uint8_t* dones,
             const float* last_values, const uint8_t* last_dones, float gamma, float tau, float* advs, float* returns,
             void* stream);

/* The rollout buffers [T, N, ...] -> the PPO dataset [N T, ...] in three launches (rows R2 + R5 and rl_games'
 * swap_and_flatten01 / prepare_dataset; the in-tree text is isaacgymenvs/learning/common_agent.py:318-411): GAE as vine_gae
 * with returns = A + V and advantages = returns - V; value_mean_std (float64 running statistics, training mode) merged
 * with the values and then with the returns, each series normalised with the statistics after its own update
 * (normalize_value) -- the running statistics themselves are READ ONLY here: the twice-updated {mean, var, count} go to
 * vms_pending [3] and the caller commits them (rl_games updates the module in prepare_dataset, not in the rollout);
 * advantages normalised by their mean and unbiased std (+1e-8) (normalize_advantage); the three series
 * land in ds_values / ds_returns / ds_advantages [N T].  Every job transposes one more rollout buffer src [T, N, width]
 * (elem_bytes 4, or 1 with width 1: done flags) into dst [N T, width].  Arrays of length njobs (<= 8) in host memory.
 * N % 64 == 0.  scratch: ceil(N / 256) * 6 + 4 doubles.  Fixed summation order, no atomics, no memsets. */
int vine_dataset_assemble(int32_t T, int64_t N, const float* rewards, const float* values, const uint8_t* dones,
                          const float* last_values, const uint8_t* last_dones, float gamma, float tau, const double* vms_mean,
                          const double* vms_var, const double* vms_count, float vms_eps, int32_t normalize_value,
                          int32_t normalize_advantage, float* ds_values, float* ds_returns, float* ds_advantages,
                          int32_t njobs, const void* const* job_src, void* const* job_dst, const int32_t* job_width,
                          const int32_t* job_elem_bytes, double* scratch, double* vms_pending, void* stream);

/* RunningMeanStd of rl_games in training mode: merge the batch moments of x [n,F] (fp32, packed, F <= 64) into the
 * float64 running mean / variance / count (Chan et al.; unbiased batch variance like torch's x.var(0)).  Two launches,
 * fixed summation order, no memsets: safe inside a captured hipGraph.  scratch: VINE_RMS_BLOCKS * 2 * F doubles. */
#define VINE_RMS_BLOCKS 128
int vine_rms_update(int64_t n, int64_t F, const float* x, double* running_mean, double* running_var, double* count,
                    double* scratch, void* stream);
/* The same for k consecutive batches of n rows each (x [k n, F]) in three launches instead of 2 k: the batch sums of all k
 * at once, then the k merges IN ORDER -- bit-identical to k calls of vine_rms_update on the slices.  snap_mean / snap_var
 * [k, F] receive the running moments as they stand after batch 0, 1, ..: the optimiser steps of PPO's first mini-epoch
 * (rl_games updates the observation normaliser in every training forward of it: the `running_mean_std` of
 * a2c_common / models.py) each normalise with the moments their own forward would have left.
 * scratch: k * (VINE_RMS_BLOCKS + 1) * 2 * F doubles. */
int vine_rms_update_multi(int32_t k, int64_t n, int64_t F, const float* x, double* running_mean, double* running_var,
                          double* count, double* scratch, double* snap_mean, double* snap_var, void* stream);

/* RunningMeanStd of rl_games in eval mode: out = clamp((x - mean) / sqrt(var + eps), +-clip) for x [n,F] packed,
 * float64 statistics; out rows out_stride elements apart (a column block of a wider buffer), fp32 or bfloat16. */
int vine_normalize_obs(int64_t n, int64_t F, const float* x, const double* mean, const double* var, float eps, float clip,
                       void* out, int64_t out_stride, int32_t out_bf16, void* stream);

/* Up to 16 vine_column_sums jobs in one launch (arrays of length njobs, host memory, same meaning per entry).
 * found_inf (nullable, device): set to 1.0 when any finished sum is not finite -- the weight gradients of a loss-scaled
 * fp16 backward pass end here, and an overflowed 16-bit gradient shows up in them as inf / NaN (vine_adam_step_amp). */
int vine_column_sums_batched(int32_t njobs, const int64_t* R, const int64_t* C, const float* const* src,
                             const int64_t* row_stride, float* const* out0, const int64_t* n0, float* const* out1,
                             const int32_t* dup, float* found_inf, void* stream);

/* Up to 24 small 2-D element moves in one launch (arrays of length njobs in host memory): dst[r, c] for r < rows,
 * c < cols, rows of dst / src dst_stride / src_stride ELEMENTS apart.  op: 0 copy, 1 zero, 2 transpose (dst[r, c] =
 * src[c, r]), 3 float32 -> bfloat16, 4 dst = src + src2 (float32), 5 dst = src * (1 - mask[r * aux]) with src float32,
 * mask = src2 (uint8, nullable) and dst float32 or bfloat16.  elem: element size of dst in bytes (2 or 4).
 * Used for the per-step operand preparation of the mixed-precision update (concatenated / padded / transposed weight
 * operands, merged head weights, observation casts). */
int vine_copy_batched(int32_t njobs, const int32_t* op, const int32_t* elem, const void* const* src,
                      const void* const* src2, void* const* dst, const int64_t* rows, const int64_t* cols,
                      const int64_t* src_stride, const int64_t* dst_stride, const int64_t* aux, void* stream);

/* out = elu(z + bias) for z [n,C] packed fp32 (a GEMM output without epilogue); out rows out_stride apart, fp32 or
 * bfloat16 (out_bf16). */
int vine_bias_elu(int64_t n, int64_t C, const float* z, const float* bias, float alpha, void* out, int64_t out_stride,
                  int32_t out_bf16, void* stream);

/* PPO loss of one minibatch of n samples with A action dims, forward AND backward in one pass:
 *   loss = mean(a_loss) + 0.5 * critic_coef * mean(c_loss) - entropy_coef * mean(entropy) + bounds_coef * mean(b_loss)
 * Outputs d(loss)/d(mu) [n,A], d(loss)/d(value) [n], d(loss)/d(logstd) [A] and
 * stats[8] = {mean a_loss, mean c_loss, mean b_loss, mean entropy, mean kl(old||new), loss, 0, 0}.
 * grad_logstd and stats are overwritten by the call.  Reductions are two-stage through `scratch`
 * (VINE_PPO_LOSS_SCRATCH_FLOATS floats, contents irrelevant): no float atomics, so gradients and statistics are
 * bit-reproducible.  Optional extra outputs (all nullable) that save the update one small launch each: kl_out[1]
 * receives the mean KL (the slot next to the gradients that rides in the all-reduce); logstd_grad_accum[A] gets
 * grad_logstd ADDED (the log-sigma parameter's gradient slot); mu_store / sigma_store [n, A] receive the new mu and
 * sigma of every sample (rl_games' dataset.update_mu_sigma; they may alias old_mu / old_sigma).  Rows of mu / grad_mu are mu_stride floats apart and elements of
 * value / grad_value value_stride apart (0 = packed: A and 1), so both heads can live in one [n, A+1] GEMM output.
 * grad_mu_bias [A] / grad_value_bias [1] (both or neither): the column sums of grad_mu / grad_value -- the gradients
 * of the two head biases -- are ADDED to them (the optimiser leaves its gradient block zeroed after every step). */
#define VINE_PPO_LOSS_BLOCKS 1024
#define VINE_PPO_LOSS_SCRATCH_FLOATS (VINE_PPO_LOSS_BLOCKS * 32)
int vine_ppo_loss(int64_t n, int32_t A, const float* mu, const float* logstd, const float* value, const float* actions,
                  const float* old_neglogp, const float* advantages, const float* old_values, const float* returns,
                  const float* old_mu, const float* old_sigma, float e_clip, int32_t clip_value, float critic_coef,
                  float entropy_coef, float bounds_coef, float soft_bound, float* grad_mu, float* grad_value,
                  float* grad_logstd, float* stats, int64_t mu_stride, int64_t value_stride, float* grad_mu_bias,
                  float* grad_value_bias, float* scratch, float* kl_out, float* logstd_grad_accum, float* mu_store,
                  float* sigma_store, const float* loss_scale, void* stream);
/* loss_scale (nullable, device scalar): every GRADIENT the call produces is multiplied by *loss_scale (the statistics are
 * not) -- torch.amp.GradScaler's `scaler.scale(loss).backward()` (rl_games drives the reference's `mixed_precision: True`
 * update through one); vine_adam_step_amp unscales. */

/* LayerNorm + heads + PPO loss + their backward in ONE launch (H = 256, NH = A + 1 in 2..5): from the LSTM output x [n, H] it forms heads = W LN(x) + wb ([n, NH], written out),
 * the loss terms and statistics of vine_ppo_loss on them, and d loss / d x ([n, H], dx: fp32, or bfloat16 with dx_bf16 --
 * the operand type vine_lstm_seq_backward_mfma takes with g_bf16) -- what
 * vine_layernorm_heads_forward + vine_ppo_loss + vine_layernorm_heads_backward produce in three launches, with the
 * same arithmetic.  ln_partial [n / R, (2 + NH) H] with R = vine_ln_heads_loss_rows() rows per workgroup (128 by default;
 * n % R == 0, n / R <= VINE_PPO_LOSS_BLOCKS): per-workgroup sums {d gamma | d beta | d W} (finish with vine_column_sums).  stats / grad_logstd / grad_mu_bias / grad_value_bias / scratch / kl_out / logstd_grad_accum /
 * mu_store / sigma_store: as in vine_ppo_loss. */
int vine_ln_heads_loss(int64_t n, int64_t H, int32_t NH, const void* x, const float* gamma, const float* beta, float eps,
                       const float* w, const float* wb, const float* logstd, const float* actions, const float* old_neglogp,
                       const float* advantages, const float* old_values, const float* returns, const float* old_mu,
                       const float* old_sigma, float e_clip, int32_t clip_value, float critic_coef, float entropy_coef,
                       float bounds_coef, float soft_bound, float* heads, void* dx, int32_t dx_bf16, float* ln_partial,
                       float* stats, float* grad_logstd, float* grad_mu_bias, float* grad_value_bias, float* scratch,
                       float* kl_out, float* logstd_grad_accum, float* mu_store, float* sigma_store, const float* loss_scale,
                       float* found_inf, void* stream);
/* dx_bf16 bit 0: dx is stored in the library's 16-bit format (vine_lp16_format()).  Bit 1 (needs bit 0): x is in that
 * format too -- what an autocast LSTM hands its LayerNorm; otherwise x is fp32.  Bits 8-15 = T > 0 (needs bit 1): x is the
 * [n / T, T + 1, H] tensor vine_lstm_seq_forward_mfma writes with c_bf16 bit 1 and sample seq * T + t is slot t + 1 of
 * sequence seq; 0: x is [n, H].  loss_scale: as in vine_ppo_loss.  found_inf (nullable, device): set to 1.0 when a
 * 16-bit dx element overflows the format or is NaN. */
int vine_ln_heads_loss_rows(void);

/* dx_bf16 bit 2 of vine_ln_heads_loss ("defer"): the kernel stops after writing its per-workgroup rows of loss sums to
 * `scratch` ([n / R, 32] floats) -- stats, grad_logstd, grad_mu_bias / grad_value_bias, kl_out and logstd_grad_accum are
 * NOT written by it.  The caller hands this record to vine_column_sums_batched_fin (the launch that ends the backward
 * pass), where one more workgroup folds the rows and writes them, beside the other jobs: the serial tail of the loss
 * kernel (a ticket behind its own stores + two dependent passes over the rows by one workgroup) disappears. */
typedef struct VineLossFinalize {
    const float* partial;      /* the `scratch` of the deferred vine_ln_heads_loss call */
    int32_t blocks;            /* n / vine_ln_heads_loss_rows() */
    int32_t A;                 /* NH - 1 */
    int64_t n;
    const float* logstd;
    float critic_coef, entropy_coef, bounds_coef;
    float* stats;
    float* grad_logstd;
    float* grad_mu_bias;       /* nullable (both or neither) */
    float* grad_value_bias;
    float* kl_out;             /* nullable */
    float* logstd_grad_accum;  /* nullable */
    const float* loss_scale;   /* nullable */
} VineLossFinalize;
/* vine_column_sums_batched + the deferred loss finalize (fin nullable: then exactly vine_column_sums_batched). */
int vine_column_sums_batched_fin(int32_t njobs, const int64_t* R, const int64_t* C, const float* const* src,
                                 const int64_t* row_stride, float* const* out0, const int64_t* n0, float* const* out1,
                                 const int32_t* dup, float* found_inf, const VineLossFinalize* fin, void* stream);

/* "fp16" (default build: IEEE half operands, the reference's autocast dtype; needs the loss scaling above) or "bf16"
 * (-DVINE_LP_BF16): the 16-bit storage format of every `bf16` / `lp16` operand, saved activation and parameter copy of
 * this header.  Parameter names containing "bf16" date from round 2 and mean "this 16-bit format". */
const char* vine_lp16_format(void);

/* Rollout inference at the reference's precision (rl_games play_steps runs fp32; only the update is autocast): the
 * policy trunk of one rollout step on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: f32 operands, f32 accumulate).
 * vine_mlp3_elu_f32: observation normalisation ((raw - mean) / sqrt(var + eps), clamped to +-clip: vine_normalize_obs'
 *   arithmetic) and the three ELU layers in one launch; writes x[row] = [mlp output (C3 = 64) | normalised obs (F_in) |
 *   zeros up to 32 columns] (rows ldx floats apart) -- the x block of the LSTM operand.  C1 = 256, C2 = 128, C3 = 64,
 *   F_in <= 32, n % 64 == 0, else VINE_ERR_UNSUPPORTED.  w1 [C1, 32] = the first layer's weight zero-padded to 32 columns
 *   (rows ldw1 >= 32 apart), w2 [C2, C1], w3 [C3, C2].
 * vine_lstm_step_f32: gates = [x | h] Wcat^T + bias, cell update as the epilogue (gate order i, f, g, o as torch.nn.LSTM):
 *   xh [N, K = 352] (rows ldx apart), w_tiled = vine_lstm_tile_weights_f32(Wcat [4H, K]), c_prev / c_out [N, H], h_out
 *   [N, H] (rows ldh apart), hp_next (nullable): a second copy of h (the h block of the NEXT step's operand, rows ldhp
 *   apart).  H = 256, K = 352, N % 512 == 0, else VINE_ERR_UNSUPPORTED. */
int vine_mlp3_elu_f32(int64_t n, float* x, int64_t ldx, const float* raw, int64_t F_in, const double* mean, const double* var,
                      float eps, float clip, const float* w1, int64_t ldw1, const float* b1, int64_t C1, const float* w2,
                      int64_t ldw2, const float* b2, int64_t C2, const float* w3, int64_t ldw3, const float* b3, int64_t C3,
                      float alpha, void* stream);

/* vine_mlp3_elu_f32 whose workgroup 0 first runs the finalise step of vine_rollout_post_defer (fin_meter == NULL: none):
 * meter / max_size / counter as in vine_rollout_post, fin_scratch / fin_blocks = that call's scratch rows. */
int vine_mlp3_elu_f32_fin(int64_t n, float* x, int64_t ldx, const float* raw, int64_t F_in, const double* mean,
                          const double* var, float eps, float clip, const float* w1, int64_t ldw1, const float* b1, int64_t C1,
                          const float* w2, int64_t ldw2, const float* b2, int64_t C2, const float* w3, int64_t ldw3,
                          const float* b3, int64_t C3, float alpha, float* fin_meter, float fin_max_size, int64_t* fin_counter,
                          const float* fin_scratch, int32_t fin_blocks, void* stream);
/* The same launch with EXACT products from bf16 pieces (the arithmetic of vine_lstm_step_f32_split, below): the four waves
 * of a workgroup share 16 rt rows and own a quarter of each layer's units; weights as vine_mlp3_tile_weights_split
 * leaves them (wt: 288 fragments x 1 KB of bfloat16 pieces), activations between the layers split once by their producer
 * and exchanged through LDS.  terms: piece pairs (9 = every bit of every product, 6 = without the three pairs below
 * 2^-24 of a product) in the low byte, row tiles per workgroup (1, 2, 4; 0 = chosen from n) in the second; bit 16: two
 * fp32 accumulators per tile, the hi x hi pair apart from the smaller pairs (see vine_lstm_step_f32_split).  C1 = 256,
 * C2 = 128, C3 = 64 fixed; F_in <= 32, n % (16 rt) == 0, ldx >= 96, else VINE_ERR_UNSUPPORTED.  fin_* as
 * vine_mlp3_elu_f32_fin.  Same output block as vine_mlp3_elu_f32.  (Replaces, on the rollout path, the fp32 network forward
 * of rl_games' play_steps: a2c_common.py's get_action_values -> model(...) under torch.no_grad, not autocast.) */
int vine_mlp3_elu_f32_split(int64_t n, float* x, int64_t ldx, const float* raw, int64_t F_in, const double* mean,
                            const double* var, float eps, float clip, const void* wt, const float* b1, const float* b2,
                            const float* b3, float alpha, int terms, float* fin_meter, float fin_max_size,
                            int64_t* fin_counter, const float* fin_scratch, int32_t fin_blocks, void* stream);
/* w1 [256, F_in] (rows ldw1 apart; columns F_in .. 31 are taken as zero), w2 [128, 256], w3 [64, 128], fp32 -> dst:
 * 288 * 512 bfloat16 (16-byte aligned), [layer][wave][k-block][unit tile][piece][lane][8]. */
int vine_mlp3_tile_weights_split(const float* w1, int64_t ldw1, int64_t F_in, const float* w2, int64_t ldw2, const float* w3,
                                 int64_t ldw3, void* dst, void* stream);
int vine_lstm_step_f32(int64_t N, int64_t H, int64_t K, const float* xh, int64_t ldx, const float* w_tiled, const float* bias,
                       const float* c_prev, float* h_out, int64_t ldh, float* c_out, float* hp_next, int64_t ldhp,
                       void* stream);
int vine_lstm_tile_weights_f32(int64_t H, int64_t K, const float* wcat, int64_t ldw, float* dst, void* stream);

/* vine_lstm_step_f32_split: the same step (same operands, same result layout) with every fp32 product formed EXACTLY on
 * the bf16 matrix cores: an fp32 number is the exact sum of three bfloat16 pieces (8 + 8 + 8 significand bits), a product
 * of two pieces is exact in the fp32 accumulator, and the 9 piece pairs of (w, x) sum to the exact 48-bit product -- so
 * `terms` = 9 v_mfma_f32_16x16x32_bf16 replace 8 v_mfma_f32_16x16x4_f32 per tile and 32 k at 16 / 9 of the fp32 matrix
 * rate, with fp32 accumulation as before (no operand is rounded; the reference's rollout GEMMs are fp32,
 * a2c_continuous / torch.nn.LSTM under no autocast).  `terms` = 6 leaves out the three pairs below 2^-24 of a product
 * (a measured variant).  w_split = vine_lstm_tile_weights_split(Wcat [4H, K]): 3 K 4H bfloat16 (6 bytes per weight).
 * H = 256, K = 352, N % 128 == 0, 16-byte aligned pointers, else VINE_ERR_UNSUPPORTED.  Bits 8-15 of `terms`: row tiles per wave
 * (tuning knob: 0 = 4 from 16384 rows on when N % 256 == 0, else 2).  Bit 16 (bits 8-15 then zero): the same arithmetic
 * with one gate per wave -- four waves share 64 rows of one 32-unit block, weight fragments straight from global memory,
 * operand pieces exchanged through LDS -- the faster form below 16384 rows.  Bit 17 (with bit 16): TWO fp32 accumulators
 * per tile -- the hi x hi pair of every k-block into one, the smaller pairs (each <= 2^-8 of it) into the other, added
 * once at the end: the roundings at the result's magnitude drop from one per instruction (66 / 99) to one per k-block (11),
 * the others happen at 2^-8 of it.  Measured against float64 (scripts/ubench/split_terms_error.py): <= 0.6x the error of
 * vine_lstm_step_f32 on every input, max and rms, with 6 or 9 pairs alike -- the form the rollout runs (terms = 6 | 3 << 16). */
int vine_lstm_step_f32_split(int64_t N, int64_t H, int64_t K, const float* xh, int64_t ldx, const void* w_split,
                             const float* bias, const float* c_prev, float* h_out, int64_t ldh, float* c_out, float* hp_next,
                             int64_t ldhp, int terms, void* stream);
int vine_lstm_tile_weights_split(int64_t H, int64_t K, const float* wcat, int64_t ldw, void* dst, void* stream);

/* Allocates the library's small per-device bookkeeping (the ticket words of the loss / Adam kernels' "last workgroup"
 * elections) for the CURRENT device.  It happens by itself on the first vine_ppo_loss / vine_ln_heads_loss /
 * vine_adam_step* call on a device; call this once beforehand when that first call would sit inside a stream capture
 * (an allocation is not allowed there).  The host mirror (learning/fused.py) calls it when it loads the library. */
int vine_ppo_runtime_init(void);

/* Rollout, policy head (row R1; rl_games play_steps_rnn / ModelA2CContinuousLogStd eval branch): from the LayerNorm
 * output y [N,H]: mu = y W_mu^T + b_mu, v = y w_v^T + b_v, sigma = exp(logstd), action = mu + sigma * eps
 * (eps ~ N(0,1) from Philox4x32-10 keyed by (seed, env, *counter)), neglogp, value un-normalised
 * (clamp(v, +-5) * value_std + value_mean when normalize_value).  Writes straight into the rollout buffers.
 * ln_gamma / ln_beta (both or neither; H == 256 only, else VINE_ERR_UNSUPPORTED): y is the RAW LSTM output and the
 * LayerNorm in front of the heads (vine_layernorm_forward's arithmetic, eps = ln_eps) is applied inside. */
int vine_policy_head(int64_t N, int32_t A, int64_t H, const float* y, const float* w_mu, const float* b_mu,
                     const float* w_v, const float* b_v, const float* logstd, const float* value_mean,
                     const float* value_std, int32_t normalize_value, uint64_t seed, const int64_t* counter,
                     float* mu_out, float* sigma_out, float* value_out, float* action_out, float* neglogp_out,
                     const float* ln_gamma, const float* ln_beta, float ln_eps, void* stream);

/* vine_policy_head reading the value normaliser's float64 running statistics itself (round 4): value un-normalised as
 * clamp(v, +-5) * sqrt(float(running_var) + value_eps) + float(running_mean) -- RunningMeanStd's own arithmetic; saves the
 * caller the four scalar launches that formed the two floats at the head of every rollout. */
int vine_policy_head_rms(int64_t N, int32_t A, int64_t H, const float* y, const float* w_mu, const float* b_mu,
                         const float* w_v, const float* b_v, const float* logstd, const double* running_mean,
                         const double* running_var, float value_eps, uint64_t seed, const int64_t* counter, float* mu_out,
                         float* sigma_out, float* value_out, float* action_out, float* neglogp_out, const float* ln_gamma,
                         const float* ln_beta, float ln_eps, void* stream);

/* Rollout, post-step bookkeeping in two launches (per-env pass + one-workgroup finalise):
 *   shaped = (rew + shift) * scale + gamma_bootstrap * value * time_out          (reward_shaper PY:58-59, value_bootstrap PY:56)
 *   dones_out = reset != 0;  cur_rewards += rew;  cur_lengths += 1
 *   finished episodes feed the two windowed means (rl_games AverageMeter, window max_size), then their
 *   accumulators and the LSTM state rows h,c [N,H] are zeroed;  *counter += 1.
 * meter[8] = {rew_mean, rew_size, len_mean, len_size, step_sum_rew, step_sum_len, step_count, 0}.
 * h_op (nullable): a second copy of h used as GEMM operand (rows h_op_stride elements apart, fp32 or bfloat16), zeroed
 * alongside h_state.  scratch: VINE_ROLLOUT_POST_SCRATCH_FLOATS floats (contents irrelevant) for the per-workgroup
 * episode sums -- fixed summation order, no atomics. */
#define VINE_ROLLOUT_POST_SCRATCH_FLOATS (1024 * 3)
int vine_rollout_post(int64_t N, int64_t H, const float* rew, const int64_t* reset, const uint8_t* timeouts,
                      const float* values, float reward_shift, float reward_scale, float gamma_bootstrap,
                      float* shaped_out, uint8_t* dones_out, float* cur_rewards, float* cur_lengths, float* h_state,
                      float* c_state, float* meter, float max_size, int64_t* counter, void* h_op, int64_t h_op_stride,
                      int32_t h_op_bf16, float* scratch, void* stream);

/* The same with the one-workgroup finalise DEFERRED (round 4): only the per-env pass runs; the caller hands `scratch` and
 * vine_rollout_post_blocks(N) (the rows it holds) to the next launch that can carry the fold as a side job of one workgroup
 * -- vine_mlp3_elu_f32_fin, the first kernel of the next rollout step -- which must run before anything reads `*counter`. */
int32_t vine_rollout_post_blocks(int64_t N);
/* The fold of vine_rollout_post_defer's per-workgroup rows as a launch of its own (one 256-thread workgroup): what
 * vine_rollout_post runs behind its per-env pass, for a caller whose next launch cannot carry the fold as a side job. */
int vine_rollout_finalize(float* meter, float max_size, int64_t* counter, const float* scratch, int32_t blocks, void* stream);
int vine_rollout_post_defer(int64_t N, int64_t H, const float* rew, const int64_t* reset, const uint8_t* timeouts,
                            const float* values, float reward_shift, float reward_scale, float gamma_bootstrap,
                            float* shaped_out, uint8_t* dones_out, float* cur_rewards, float* cur_lengths, float* h_state,
                            float* c_state, void* h_op, int64_t h_op_stride, int32_t h_op_bf16, float* scratch, void* stream);

/* Adam step on FLAT buffers (all parameters of the model live in one contiguous block, likewise gradients and
 * moments): torch.optim.Adam arithmetic (rl_games: Adam(lr, eps=1e-8), common_agent.py:80) in ONE launch instead of a
 * multi-tensor kernel over 17 small tensors.  `lr` and `step` are device scalars (the adaptive-KL schedule updates lr
 * on the device; `step` is incremented by the kernel).  g is pre-scaled by grad_scale (1/world after the all-reduce)
 * and zeroed after use.  bf16_shadow (nullable, n bfloat16): receives a bfloat16 copy of the updated parameters -- the
 * GEMM operands of the mixed-precision update -- so no separate cast pass over the weights is needed. */
int vine_adam_step(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq, const float* lr,
                   float* step, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                   void* bf16_shadow, void* stream);

/* The same step with rl_games' AdaptiveScheduler (`schedule_type: legacy`) folded in: every workgroup reads the OLD
 * learning rate, the last one to finish writes lr = max(lr/1.5, min_lr) if kl*kl_scale > 2*thr, min(lr*1.5, max_lr) if
 * kl*kl_scale < 0.5*thr -- i.e. exactly vine_adam_step followed by vine_adaptive_lr, in one launch (kl NULL: no
 * schedule).  The step counter is bumped the same way.  (The "last workgroup" election uses a ticket word per
 * (device, stream): launches on different streams do not interfere.) */
int vine_adam_step_sched(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* lr, float* step,
                         float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* bf16_shadow,
                         const float* kl, float kl_scale, float kl_threshold, float min_lr, float max_lr, void* stream);

/* vine_adam_step_sched with torch.amp.GradScaler restated on the device (rl_games: `self.scaler.step(self.optimizer);
 * self.scaler.update()` for `mixed_precision: True`, PY:53).  amp_state (nullable, device float[4]) = {loss scale, growth
 * tracker, growth interval, unused}: gradients are unscaled by 1 / scale.  found_inf (nullable, device scalar; with
 * several ranks it rides in the all-reduced gradient block, so every rank sees the sum): when non-zero the step is
 * SKIPPED -- parameters, moments, step counter and the 16-bit copies keep their values, the gradient block is cleared --
 * and the scale is halved; otherwise the tracker counts up and the scale doubles every `growth interval` good steps
 * (GradScaler defaults: 65536, x2 every 2000, x0.5 on overflow).  The flag is cleared by the launch.  The learning-rate
 * schedule runs either way. */
int vine_adam_step_amp(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* lr, float* step,
                       float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* lp16_shadow,
                       const float* kl, float kl_scale, float kl_threshold, float min_lr, float max_lr, float* amp_state,
                       float* found_inf, void* stream);

/* rl_games' AdaptiveScheduler on device scalars (`schedule_type: legacy`, PY:64-66):
 * kl > 2*thr -> lr = max(lr/1.5, min_lr); kl < 0.5*thr -> lr = min(lr*1.5, max_lr).  kl_scale = 1/world. */
int vine_adaptive_lr(float* lr, const float* kl, float kl_scale, float kl_threshold, float min_lr, float max_lr,
                     void* stream);

/* One ROLLOUT step in one launch (round 5): policy head -> vine_step -> rollout bookkeeping, for the PPO loop around the env
 * (rl_games play_steps_rnn; in-tree text common_agent.py:257-317).  What three launches did -- vine_policy_head (LayerNorm +
 * mu / value heads of the LSTM output rows, Gaussian sampling, neglogp, value un-normalisation), vine_step, vine_rollout_post
 * (reward shaping + time-out bootstrap, done flags, episode accumulators, LSTM-state rows of finished envs cleared) -- the
 * four-lanes-per-env step kernel does in its prologue and epilogue: the sampled action never leaves the registers, the
 * value is still there when the bootstrap needs it, and two of the five launches of a rollout step (and their ~1.5 us
 * boundaries) are gone.  Same formulas and the same Philox keys as the separate kernels (include/vine_ppo.h); the LayerNorm
 * is two-pass (mean, then centred moments) over 64 units per lane.  A == 2 actions, H == 256.
 *   y [N, 256] fp32: LSTM output rows;  hw [3][256], hc [3]: vine_rollout_head_prep's products gamma_u w_k[u] (k = mu_0, mu_1,
 *   value) and constants sum_u beta_u w_k[u] + b_k;  logstd [2];  value_mean / value_var: float64 scalars of the value
 *   normaliser (NULL: values are not un-normalised), value_eps its epsilon;  seed / counter: the head's Philox key and the
 *   rollout counter (device scalar, read only);  *_out: the step's rows of the rollout buffers.
 *   reward_shift / reward_scale / gamma_bootstrap, shaped_out [N], dones_out [N] u8, cur_rewards / cur_lengths [N] in/out,
 *   h_state / c_state [N, 256] (rows of finished envs cleared), h_op (nullable; fp32, row stride h_op_stride floats): the
 *   operand copy of h the next inference step reads, partial [vine_step_rollout_blocks(h)][3]: per-workgroup {sum of finished
 *   returns, sum of finished lengths, count} for vine_rollout_finalize / the fold that rides in the next MLP launch.
 * Returns VINE_ERR_UNSUPPORTED when the handle's configuration does not run the four-lane kernel. */
typedef struct VineRolloutArgs {
    const float* y; const float* hw; const float* hc; const float* logstd;
    const double* value_mean; const double* value_var;
    float ln_eps, value_eps;
    uint64_t seed; const int64_t* counter;
    float* mu_out; float* sigma_out; float* value_out; float* action_out; float* neglogp_out;
    float reward_shift, reward_scale, gamma_bootstrap; int32_t reserved;
    float* shaped_out; uint8_t* dones_out; float* cur_rewards; float* cur_lengths;
    float* h_state; float* c_state; float* h_op; int64_t h_op_stride;
    float* partial;
} VineRolloutArgs;
struct VineHandle;      /* include/vine.h */
int vine_step_rollout(struct VineHandle* h, const VineRolloutArgs* args, float* obs, float* rew, int64_t* reset, int64_t* progress,
                      uint8_t* timeouts, void* stream);
int32_t vine_step_rollout_blocks(struct VineHandle* h);
int32_t vine_step_rollout_args_size(void);      /* sizeof(VineRolloutArgs): checked against the ctypes mirror */
int vine_rollout_head_prep(const float* ln_gamma, const float* ln_beta, const float* w_mu, const float* b_mu, const float* w_v,
                           const float* b_v, float* hw, float* hc, void* stream);


#ifdef __cplusplus
}
#endif
#endif /* VINE_PPO_H */
