/* coskad_hip.h -- C ABI of the MI355X (gfx950) HIP library for COSKAD's STS-GCN hot path.
 *
 * The reference (aleflabo/COSKAD) has no FFI: the path is a chain of PyTorch ops inside
 * nn.Modules.  Each entry point below replaces the op chain of the reference lines it
 * cites; the Python host code in coskad_amd/ (torch.autograd.Functions / nn.Modules with
 * the reference's names) is the only caller.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - all tensors: device pointers, fp32, contiguous, layout [N, C, T, V] (T*V = "positions")
 *   - nothing is allocated, nothing synchronises; kernels are enqueued on `stream`
 *   - return 0 on success, negative COSKAD_ERR_* otherwise; text via coskad_last_error()
 *   - re-entrant; no global mutable state (last-error text is thread-local)
 */
#ifndef COSKAD_HIP_H
#define COSKAD_HIP_H
#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

int coskad_abi_version(void);
const char* coskad_last_error(void);

/* Timing probe for benchmarks: after coskad_probe_begin(kernel, Ci, Co) every launch of that tile kernel
 * (1 = layer_apply, 2 = bwd_data, 3 = bwd_reduce, 4 = fwd_moments, 5 = gcn_params; 6 = ALL launches of one
 * coskad_layer_bwd*_f32 call together; 7 = the fused eval-mode encoder kernel) with those channel counts is bracketed by
 * HIP events on its launch stream; coskad_probe_end() waits for them and returns the average duration. */
int coskad_probe_begin(int kernel, int Ci, int Co);
int coskad_probe_stride(int n);   /* time every n-th matching launch only (default 1) */
int coskad_probe_end(float* avg_ms, int* launches);

/* ---- forward ------------------------------------------------------------------------ */

/* ConvTemporalGraphical.forward (models/graph_layers/stsgcn.py:143-156) on rows = N*C rows
 * of T*V floats; adjoint != 0 applies the transposed operator (its backward w.r.t. X). */
int coskad_gcn_f32(const float* in, float* out, const float* A, const float* Tm, int rows, int T, int V,
                   int adjoint, hipStream_t stream);

/* BatchNorm2d statistics -> folded conv weights for one ST_GCNN_layer
 * (stsgcn.py:56-80: tcn = Conv1x1+BN, residual = Conv1x1+BN or Identity when Wr == NULL).
 * wfold: [2*Ci][CoP], bias: [CoP], CoP = Co rounded up to 16. */
int coskad_bn_fold_f32(const float* Wt, const float* bt, const float* gamma_t, const float* beta_t,
                       const float* mean_t, const float* var_t, const float* Wr, const float* br,
                       const float* gamma_r, const float* beta_r, const float* mean_r,
                       const float* var_r, float* wfold, float* bias, int Ci, int Co,
                       hipStream_t stream);

/* ST_GCNN_layer.forward (stsgcn.py:94-116) with folded BatchNorm:
 *   out = [PReLU_out]( Wz . gcn([PReLU_in] in) + Wx . [PReLU_in] in + b )
 * in_slope / out_slope: device pointers to the 1-element PReLU weight, or NULL to skip. */
int coskad_layer_apply_f32(const float* in, float* out, const float* A, const float* Tm,
                           const float* wfold, const float* bias, const float* in_slope,
                           const float* out_slope, int B, int Ci, int Co, int T, int V,
                           hipStream_t stream);

/* nn.PReLU() with one shared weight (stsgcn.py:82,110), elementwise, for API paths that must
 * materialise the post-activation tensor.  bwd: du = dout * PReLU'(u); dslope (+)= sum dout*u [u<0];
 * ws >= 1024 floats. */
int coskad_prelu_fwd_f32(const float* u, const float* slope, float* out, size_t n, hipStream_t stream);
int coskad_prelu_bwd_f32(const float* u, const float* dout, const float* slope, float* du, float* dslope,
                         float* ws, int accumulate, size_t n, hipStream_t stream);

/* Reconstruction head of the decoder models: x_rec = PReLU_slope(U) (the last decoder layer's activation, stsgcn.py:110),
 * loss[0] = F.mse_loss(x_rec, x) (euclidean_autoencoder.py:111, spherical_vae.py:90); dU (optional) = upstream * dloss/dU,
 * dslope (optional, needs dU) (+)= upstream * dloss/dslope, xrec (optional) = the reconstruction.  ws: >= 2048 floats. */
int coskad_rec_head_f32(const float* U, const float* x, const float* slope, float* xrec, float* dU, float* loss, float* dslope,
                        float upstream, float* ws, int accumulate, size_t n, hipStream_t stream);

/* `rev_btlnk` of the decoder models (models/sts/ae.py:223-227: nn.Linear(latent_dim -> hidden * T * V)) and its autograd as
 * streaming kernels over the one large tensor (csrc/rev_btlnk.hip); latent_dim L in {8, 16}, N % 4 == 0:
 *   fwd: H [B, N] = z [B, L] W^T + bias        (W [N, L])
 *   bwd: dz [B, L] (+)= dH W (dz_accumulate);  dW [N, L], db [N] (may be NULL) (+)= their batch sums (accumulate);
 *        ws: coskad_rev_btlnk_ws_floats(B, N, L) floats; two-stage fixed-order reductions (deterministic) */
size_t coskad_rev_btlnk_ws_floats(int B, int N, int L);
int coskad_rev_btlnk_fwd_f32(const float* z, const float* W, const float* bias, float* H, int B, int N, int L, hipStream_t stream);
int coskad_rev_btlnk_bwd_f32(const float* dH, const float* z, const float* W, float* dz, int dz_accumulate, float* dW, float* db,
                             int accumulate, float* ws, int B, int N, int L, hipStream_t stream);

/* ---- train-mode BatchNorm statistics -------------------------------------------------- */

/* Bytes of scratch `ws` that coskad_layer_train_stats_f32 needs for C_in = Ci. */
size_t coskad_train_stats_ws_bytes(int Ci);
/* Floats in the per-layer stat block saved for the backward pass. */
int coskad_stat_floats(int Ci, int Co);

/* Batch statistics of both BatchNorm2d of one ST_GCNN_layer in training mode
 * (stsgcn.py:65,76 via 106-108), from the moments of the conv inputs; writes the folded
 * weights for coskad_layer_apply_f32, the stat block for the backward pass, and updates
 * running_mean / running_var (unbiased) / num_batches_tracked in place (NULL to skip).
 * Wr == NULL: identity residual (stsgcn.py:79-80). */
int coskad_layer_train_stats_f32(const float* in, const float* A, const float* Tm, const float* in_slope,
                                 const float* Wt, const float* bt, const float* gamma_t,
                                 const float* beta_t, float* rmean_t, float* rvar_t, long long* nbt_t,
                                 const float* Wr, const float* br, const float* gamma_r,
                                 const float* beta_r, float* rmean_r, float* rvar_r, long long* nbt_r,
                                 float momentum, float* wfold, float* bias, float* stat, void* ws,
                                 size_t ws_bytes, int B, int Ci, int Co, int T, int V,
                                 hipStream_t stream);

/* Stored-Z variant of the training forward: the statistics pass also writes Z = gcn(PReLU(in)) [B,Ci,T,V]; the layer
 * is then a streaming GEMM over Z and `in` (no staging, no mixing recompute), and the backward reads Z as well. */
int coskad_layer_train_stats_z_f32(const float* in, const float* A, const float* Tm, const float* in_slope,
                                   const float* Wt, const float* bt, const float* gamma_t,
                                   const float* beta_t, float* rmean_t, float* rvar_t, long long* nbt_t,
                                   const float* Wr, const float* br, const float* gamma_r,
                                   const float* beta_r, float* rmean_r, float* rvar_r, long long* nbt_r,
                                   float momentum, float* wfold, float* bias, float* stat, void* ws,
                                   size_t ws_bytes, int B, int Ci, int Co, int T, int V,
                                   hipStream_t stream, float* Z);
int coskad_layer_apply_z_f32(const float* Z, const float* in, float* out, const float* A, const float* Tm,
                             const float* wfold, const float* bias, const float* in_slope, const float* out_slope,
                             int B, int Ci, int Co, int T, int V, hipStream_t stream);

/* ---- training forward with the NEXT layer's statistics fused into the apply kernel (csrc/fused_apply_next.hip) --------
 * Layer i's apply holds a clip's whole U_i on chip; for T = 12, V = 17, C_in in {2, 16, 32}, C_out in {16, 32} it also forms
 * X_{i+1} = PReLU_i(U_i), Z_{i+1} = gcn_{i+1}(X_{i+1}) (stsgcn.py:154-155 of the next layer) and the moment partials
 * [sum x x^T | sum x | sum z z^T | sum z] that layer i+1's BatchNorms need (stsgcn.py:65,76), so the statistics pass of
 * layer i+1 (coskad_layer_train_stats_z_f32's first kernel) and its re-read of U_i disappear.
 *   coskad_build_ftab_f32       : forward mixing tables (coskad_ftab_floats() floats each) of n <= 4 layers from their A / T
 *                                 (host arrays of n device pointers), one launch
 *   coskad_layer_apply_next_f32 : out = U_i [B,Co,T,V]; Z_next [B,Co,T,V]; partials [coskad_layer_apply_next_rows(B, Ci, Co)][2 (Co^2 + Co)]
 *   coskad_layer_train_fold_f32 : the rest of coskad_layer_train_stats_f32 for layer i+1 (fp64 sums of the partial rows, statistics,
 *                                 folded weights, stat block, running-stat update); ws >= coskad_train_stats_ws_bytes(Ci). */
int coskad_layer_apply_next_ok(int Ci, int Co, int T, int V);
int coskad_ftab_floats(void);
int coskad_layer_apply_next_rows(int B, int Ci, int Co);
int coskad_build_ftab_f32(const float* const* A, const float* const* Tm, float* const* tab, int n, int T, int V,
                          hipStream_t stream);
int coskad_layer_apply_next_f32(const float* Z, const float* in, float* out, const float* wfold, const float* bias,
                                const float* in_slope, const float* out_slope, const float* ftab_next, float* Z_next,
                                float* partials, size_t partials_bytes, int B, int Ci, int Co, int T, int V,
                                hipStream_t stream);
int coskad_layer_train_fold_f32(const float* partials, int rows, const float* Wt, const float* bt, const float* gamma_t,
                                const float* beta_t, float* rmean_t, float* rvar_t, long long* nbt_t,
                                const float* Wr, const float* br, const float* gamma_r,
                                const float* beta_r, float* rmean_r, float* rvar_r, long long* nbt_r,
                                float momentum, float* wfold, float* bias, float* stat, void* ws,
                                size_t ws_bytes, int B, int Ci, int Co, int T, int V, hipStream_t stream);

/* ---- fused eval-mode encoder (models/common/components.py:94-105 in one kernel) ------------------------------
 * Built for the reference's default geometry: n_frames 12, n_joints 17, channels 2-32-16-32-64.
 * coskad_gather_f32 : out[i] = idx[i] >= 0 ? src[idx[i]] : 0 -- builds the operand streams (coskad_amd/fused_plan.py
 *                     holds the index maps) from the concatenated A / T / folded-weight / bias / bottleneck tensors.
 * x     [B,2,12,17]; tab, wreg: operand streams; slopes[4]: the PReLU weights of the four layers;
 * out   [B][coskad_fused_encoder_out_floats()]: PReLU(last layer) in tile-major order, zero in the padding columns:
 *       the bottleneck (coskad_btlnk_fwd_f32 with slope NULL) reads it with the weight permuted the same way. */
int coskad_gather_f32(const float* src, const int* idx, float* out, size_t n, hipStream_t stream);
int coskad_fused_encoder_out_floats(void);
int coskad_fused_encoder_f32(const float* x, float* out, const float* tab, const float* wreg, const float* slopes, int B,
                             int T, int V, hipStream_t stream);

/* ---- strided batched fp32 GEMM (plain-GCN encoders: learnable_gcn.py:65-72, gcn.py:48-54; 1x1 convs of wide layers) ----
 * C[b][m][n] = act(sum_k A[b][m][k] B[b][k][n] + bias), every operand addressed by element strides (s?_b may be 0).
 * bias_mode: 0 none, 1 bias[m % bias_mod], 2 bias[n];  relu != 0: max(., 0).
 * reduce != 0: C receives partial sums [ceil(batch / chunk)][M][N] (contiguous) over the batches of each chunk -- sum them
 *              with coskad_gemm_sum_f32 (fp64, fixed order).  ktotal > 0: element (b, k) exists iff b*K + k < ktotal.
 * accum != 0 (reduce == 0): C += the product (after bias / ReLU). */
int coskad_gemm_f32(const float* A, const float* B, float* C, const float* bias, long long sa_b, long long sa_m,
                    long long sa_k, long long sb_b, long long sb_k, long long sb_n, long long sc_b, long long sc_m,
                    long long sc_n, int M, int N, int K, int batch, int bias_mode, int bias_mod, int relu, int reduce,
                    int chunk, long long ktotal, int accum, hipStream_t stream);
int coskad_gemm_sum_f32(const float* partials, int chunks, size_t E, float* out, int accumulate, hipStream_t stream);
/* g = dout * (out > 0) on [Nb, C, P]; part [slices][C]: per-slice channel sums of g (bias gradient partials) */
int coskad_relu_bwd_f32(const float* out, const float* dout, float* g, float* part, int Nb, int C, int P, int slices,
                        hipStream_t stream);
/* row softmax of an n x n matrix (nn.Softmax() on the 2-D learnable adjacency, learnable_gcn.py:36,66) and its backward */
int coskad_softmax_rows_f32(const float* x, float* y, int n, hipStream_t stream);
int coskad_softmax_rows_bwd_f32(const float* y, const float* dy, float* dx, int n, hipStream_t stream);

/* ---- 1x1 convolution in NCHW for wide layers (nn.Conv2d(C_in, C_out, 1) of stsgcn.py:57-63,71-75; csrc/conv1x1.hip) -----------
 * Out[b][m][p] (+)= sum_k A(m,k) In[b][k][p] (+ bias[m]) with In [batch][K][P], Out [batch][M][P] contiguous; A(m,k) = A[m sa_m + k sa_k],
 * sa_k == 1 (forward: W [M][K]) or sa_m == 1 (data gradient: W^T of W [K][M]).  Layout-specialised MFMA kernel (float4 loads, K tiles
 * double-buffered through registers, 16-byte stores) for P in {204, 300}, K % 16 == 0, M % 32 == 0 (coskad_conv1x1_ok); other shapes:
 * coskad_gemm_f32.  stats (optional): [coskad_conv1x1_stat_rows()][M][2] doubles receive the per-channel sum / sum of squares of Out --
 * the train-mode BatchNorm statistics without another pass over the tensor; coskad_bn2_stats_parts_f32 turns them into
 * stat [2C] = (mean, 1 / sqrt(var + eps)) and updates the running statistics like coskad_bn2_stats_f32. */
int coskad_conv1x1_ok(int M, int K, int P);
int coskad_conv1x1_stat_rows(int M, int K, int P, int batch);
int coskad_conv1x1_f32(const float* A, long long sa_m, long long sa_k, const float* In, float* Out, const float* bias, double* stats,
                       int M, int K, int P, int batch, int accumulate, hipStream_t stream);
int coskad_bn2_stats_parts_f32(const double* parts, int rows, float* stat, float* running_mean, float* running_var,
                               long long* num_batches_tracked, float momentum, float eps, double count, int C, hipStream_t stream);
/* weight gradient of the 1x1 convolution: partials [ceil(batch / chunk)][M][K] of dW[m][k] = sum_b sum_p G[b][m][p] X[b][k][p]
 * (G = gradient of the conv output [batch][M][P], X = conv input [batch][K][P]); P in {204, 300}, M a multiple of 32, K of 64
 * (coskad_conv1x1_wgrad_ok); sum the partials with coskad_gemm_sum_f32 (fp64, fixed order: deterministic) */
int coskad_conv1x1_wgrad_ok(int M, int K, int P);
int coskad_conv1x1_wgrad_f32(const float* G, const float* X, float* partials, int M, int K, int P, int batch, int chunk,
                             hipStream_t stream);

/* ---- BatchNorm2d + residual add + PReLU of an ST_GCNN layer on [Nb, C, P] tensors (stsgcn.py:56-80,106-110), for layers
 * beyond the LDS-resident tile kernels (their 1x1 convolutions are coskad_gemm_f32, their mixing coskad_gcn_f32).
 * stat [2C] = (mean, 1/sqrt(var + eps)); stat_r == NULL: identity residual.  ws: coskad_bn2_ws_bytes / _bwd_ws_bytes. */
size_t coskad_bn2_ws_bytes(int Nb, int C);
size_t coskad_bn2_bwd_ws_bytes(int Nb, int C);
int coskad_bn2_stats_f32(const float* x, float* stat, float* running_mean, float* running_var, long long* num_batches_tracked,
                         float momentum, float eps, int training, void* ws, size_t ws_bytes, int Nb, int C, int P,
                         hipStream_t stream);
int coskad_bn2_apply_prelu_f32(const float* Ct, const float* Cr, const float* stat_t, const float* gamma_t, const float* beta_t,
                               const float* stat_r, const float* gamma_r, const float* beta_r, const float* slope, float* out,
                               int Nb, int C, int P, hipStream_t stream, float drop_p, unsigned long long drop_seed);
int coskad_bn2_bwd_f32(const float* Ct, const float* Cr, const float* dOut, const float* stat_t, const float* gamma_t,
                       const float* beta_t, const float* stat_r, const float* gamma_r, const float* beta_r, const float* slope,
                       float* dCt, float* dCr, float* dgamma_t, float* dbeta_t, float* dgamma_r, float* dbeta_r, float* dslope,
                       int training, void* ws, size_t ws_bytes, int Nb, int C, int P, hipStream_t stream, float drop_p,
                       unsigned long long drop_seed);
/* Train-mode nn.Dropout(p) of the tcn branch (stsgcn.py:66, between BatchNorm and the residual add) in the two calls above:
 * drop_p in [0, 1) and a seed; element i of the [Nb, C, P] tensor is kept (and scaled by 1 / (1 - p)) iff a counter-based hash
 * of (seed, i) says so -- the backward recomputes the mask from the same seed, nothing is stored.  drop_p = 0: no dropout.
 * coskad_dropout_mask_f32 writes that mask (values 0 or 1 / (1 - p)) for tests / oracles. */
int coskad_dropout_mask_f32(float* out, size_t n, float drop_p, unsigned long long drop_seed, hipStream_t stream);

/* ---- `mlp` projector tail (models/common/components.py:209-226 behind its first Linear, which runs on the bottleneck
 * kernels): z = W2 . relu(BatchNorm1d(y1)) + b2 on y1 [B, H]; H, L <= 64.
 * forward : training != 0 -> batch statistics (biased variance normalises, running_mean / running_var (unbiased) and
 *           num_batches_tracked are updated in place; they may be NULL), else the running statistics.
 *           stat [0:2H] receives (mean, 1/sqrt(var + eps)) for the backward; the buffer holds coskad_mlp_head_ws_floats(B,H,L)
 *           floats (behind the 2H values: the per-block partial sums of the hidden, out <= 16 kernels).
 * backward: dy1 [B,H], dgamma, dbeta [H], dW2 [L,H], db2 [L] (may be NULL); red: coskad_mlp_head_ws_floats(B,H,L) floats of
 *           scratch; accumulate != 0 adds into the parameter gradients. */
size_t coskad_mlp_head_ws_floats(int B, int H, int L);   /* floats of `stat` (forward) and of `red` (backward), 8-byte aligned */
int coskad_mlp_head_fwd_f32(const float* y1, const float* gamma, const float* beta, float* running_mean,
                            float* running_var, long long* num_batches_tracked, float momentum, float eps, int training,
                            const float* W2, const float* b2, float* z, float* stat, int B, int H, int L,
                            hipStream_t stream);
int coskad_mlp_head_bwd_f32(const float* y1, const float* stat, const float* gamma, const float* beta, const float* W2,
                            const float* dz, float* dy1, float* dgamma, float* dbeta, float* dW2, float* db2, float* red,
                            int training, int accumulate, int B, int H, int L, hipStream_t stream);

/* ---- backward of one ST_GCNN_layer (autograd of stsgcn.py:94-116 in training mode) ------ */

size_t coskad_layer_bwd_ws_bytes(int B, int Ci, int Co, int T, int V);

/* 1 when one clip of a (Ci -> Co) ST_GCNN layer fits the LDS-resident tile kernels (forward, statistics, backward);
 * 0 otherwise (more than 64 channels, or 64 input channels on the 25-joint layout): callers then compose the layer
 * from coskad_gcn_f32 + GEMMs. */
int coskad_layer_fits(int Ci, int Co, int T, int V);

/* coskad_layer_bwd_z_f32 inside a chain of layers (reference: autograd walks stsgcn.py:94-116 layer by layer, last to first).
 * Stage 1 of a layer's backward (the batch reductions P = sum dU.Z^T, Q = sum dU.X^T, sdU) reads the dU the layer ABOVE has
 * just produced; where that layer's data kernel holds it on chip, it forms the reductions itself:
 *   stats_in, stats_in_rows                : this layer's chain buffer, filled by the call for the layer above (NULL: stage 1 runs
 *                                            here, as in coskad_layer_bwd_z_f32)
 *   below_in, below_Z [B,below_Ci,T,V]     : input of the layer below as stored (pre-activation; below_in_slope = its producer's
 *                                            PReLU weight, NULL for the raw network input) and its stored Z
 *   below_stats [coskad_layer_bwd_below_floats(B,Ci,Co,below_Ci,T,V) floats, 8-byte aligned] : ITS chain buffer (NULL: not formed):
 *       coskad_layer_bwd_below_rows(...) partial rows of 2 Ci below_Ci + Ci floats, then their fp64 sums (formed by this call's
 *       partial-sum launch: the layer below runs neither its batch reductions nor their summation)
 * coskad_layer_bwd_below_rows returns 0 when the (Ci -> Co) data kernel cannot form them (built at T = 12, V = 17: 32 -> 16 above
 * a 2-channel layer, 16 -> 32 above a 32-channel layer, 32 -> 64 above a 16-channel layer: the default stack). */
int coskad_layer_bwd_below_rows(int B, int Ci, int Co, int below_Ci, int T, int V);
size_t coskad_layer_bwd_below_floats(int B, int Ci, int Co, int below_Ci, int T, int V);
int coskad_layer_bwd_chain_f32(const float* in, const float* dU, const float* A, const float* Tm,
                               const float* in_slope, const float* stat, const float* Wt, const float* gamma_t,
                               const float* Wr, const float* gamma_r, float* dIn, float* dA, float* dT, float* dWt,
                               float* dbt, float* dgamma_t, float* dbeta_t, float* dWr, float* dbr,
                               float* dgamma_r, float* dbeta_r, float* dslope_in, void* ws, size_t ws_bytes,
                               int accumulate, int B, int Ci, int Co, int T, int V, hipStream_t stream, const float* Z,
                               const float* stats_in, int stats_in_rows, size_t stats_in_bytes, const float* below_in,
                               const float* below_Z, const float* below_in_slope, int below_Ci, float* below_stats,
                               size_t below_stats_bytes, double stats_count);

/* ---- SyncBN (optional: the reference trains with per-rank BatchNorm statistics, train_COSKAD.py:75-78) -------------
 * The batch reductions and the folds behind them as separate calls, so that a data-parallel caller can add the other ranks'
 * fp64 sums (one small all-reduce per BatchNorm boundary) in between.  Forward: coskad_layer_train_moments_f32 (a layer with its own
 * statistics pass) or coskad_layer_moment_sums_f32 (moment partials from coskad_layer_apply_next_f32) -> sums [2 (Ci^2 + Ci)]
 * doubles: [sum xx^T][sum x][sum zz^T][sum z] -> all-reduce -> coskad_layer_train_fold_sums_f32 with count = global clips x T x V.
 * Backward: coskad_layer_bwd_stats_f32 (stage 1 alone, into a chain buffer: rows, then sums at coskad_layer_bwd_sums_offset floats)
 * or the chain buffer the layer above filled -> all-reduce the sums in place -> coskad_layer_bwd_chain_f32 with stats_in and
 * stats_count = global clips x T x V (0: this batch). */
int coskad_layer_train_moments_f32(const float* in, const float* A, const float* Tm, const float* in_slope, float* Z, double* sums,
                                   void* ws, size_t ws_bytes, int B, int Ci, int T, int V, hipStream_t stream);
int coskad_layer_moment_sums_f32(const float* partials, int rows, int Ci, double* sums, hipStream_t stream);
int coskad_layer_train_fold_sums_f32(const double* sums, double count, const float* Wt, const float* bt, const float* gamma_t,
                                     const float* beta_t, float* rmean_t, float* rvar_t, long long* nbt_t,
                                     const float* Wr, const float* br, const float* gamma_r,
                                     const float* beta_r, float* rmean_r, float* rvar_r, long long* nbt_r,
                                     float momentum, float* wfold, float* bias, float* stat, int Ci, int Co, hipStream_t stream);
size_t coskad_layer_bwd_stats_floats(int B, int Ci, int Co, int T, int V);
size_t coskad_layer_bwd_sums_offset(int rows, int Ci, int Co);
int coskad_layer_bwd_stats_f32(const float* in, const float* dU, const float* A, const float* Tm, const float* in_slope,
                               int has_residual, float* stats_out, size_t stats_out_bytes, int* rows_out, void* ws, size_t ws_bytes,
                               int B, int Ci, int Co, int T, int V, hipStream_t stream, const float* Z);

/* in   : the layer's input as stored by the producer (pre-activation; in_slope = its PReLU weight,
 *        NULL when `in` is the raw network input)
 * dU   : gradient w.r.t. this layer's pre-activation output [B,Co,T,V]
 * stat : stat block written by coskad_layer_train_stats_f32 in the forward pass
 * dIn  : gradient w.r.t. `in` (already multiplied by the producer's PReLU derivative); NULL to skip
 * dslope_in : gradient of the producer's PReLU weight (1 float); NULL to skip
 * parameter gradients are written (accumulate == 0) or added (accumulate != 0).
 * Wr == NULL: identity residual. */
int coskad_layer_bwd_f32(const float* in, const float* dU, const float* A, const float* Tm,
                         const float* in_slope, const float* stat, const float* Wt, const float* gamma_t,
                         const float* Wr, const float* gamma_r, float* dIn, float* dA, float* dT, float* dWt,
                         float* dbt, float* dgamma_t, float* dbeta_t, float* dWr, float* dbr,
                         float* dgamma_r, float* dbeta_r, float* dslope_in, void* ws, size_t ws_bytes,
                         int accumulate, int B, int Ci, int Co, int T, int V, hipStream_t stream);

/* coskad_layer_bwd_f32 reading the stored Z = gcn(PReLU(in)) of coskad_layer_train_stats_z_f32 instead of recomputing
 * the mixing in its reduction and data kernels (Z == NULL: identical to coskad_layer_bwd_f32). */
int coskad_layer_bwd_z_f32(const float* in, const float* dU, const float* A, const float* Tm,
                           const float* in_slope, const float* stat, const float* Wt, const float* gamma_t,
                           const float* Wr, const float* gamma_r, float* dIn, float* dA, float* dT, float* dWt,
                           float* dbt, float* dgamma_t, float* dbeta_t, float* dWr, float* dbr,
                           float* dgamma_r, float* dbeta_r, float* dslope_in, void* ws, size_t ws_bytes,
                           int accumulate, int B, int Ci, int Co, int T, int V, hipStream_t stream, const float* Z);

/* The same backward split in two calls, so that stage 4 (dA, dT; independent of the rest of the chain) can be enqueued
 * on a second stream: coskad_layer_bwd_data_f32 = batch reductions + fold + data path, with the mixing-output gradient
 * dZ [B,Ci,T,V] written to the caller's buffer; coskad_layer_gcn_params_f32 = dA, dT from `in` and that dZ. */
int coskad_layer_bwd_data_f32(const float* in, const float* dU, const float* A, const float* Tm,
                              const float* in_slope, const float* stat, const float* Wt, const float* gamma_t,
                              const float* Wr, const float* gamma_r, float* dIn, float* dZ, float* dWt, float* dbt,
                              float* dgamma_t, float* dbeta_t, float* dWr, float* dbr, float* dgamma_r,
                              float* dbeta_r, float* dslope_in, void* ws, size_t ws_bytes, int accumulate, int B,
                              int Ci, int Co, int T, int V, hipStream_t stream, const float* Z /* stored gcn(in) or NULL */);
size_t coskad_layer_gcn_params_ws_bytes(int T, int V);
int coskad_layer_gcn_params_f32(const float* in, const float* in_slope, const float* dZ, const float* A,
                                const float* Tm, float* dA, float* dT, void* ws, size_t ws_bytes, int accumulate,
                                int B, int Ci, int T, int V, hipStream_t stream);

/* Parameter gradients of ConvTemporalGraphical alone (stsgcn.py:154-155), given its input x and
 * the gradient dZ of its output; rows = N*C. */
size_t coskad_gcn_bwd_params_ws_bytes(int T, int V);
int coskad_gcn_bwd_params_f32(const float* x, const float* dZ, const float* A, const float* Tm, float* dA,
                              float* dT, void* ws, size_t ws_bytes, int accumulate, int rows, int T, int V,
                              hipStream_t stream);

/* coskad_gcn_bwd_params_f32 AND the input gradient of ConvTemporalGraphical (models/graph_layers/stsgcn.py:143-156 under
 * autograd) in one pass over dZ: dX = gcn^T(dZ) (+ dX_add, optional, same shape: the gradient an identity residual carries
 * beside the mix) -- what coskad_gcn_f32(adjoint = 1) computes from a second read of dZ. */
int coskad_gcn_bwd_params_dx_f32(const float* x, const float* dZ, const float* A, const float* Tm, float* dA, float* dT,
                                 float* dX, const float* dX_add, void* ws, size_t ws_bytes, int accumulate, int rows, int T,
                                 int V, hipStream_t stream);

/* ---- bottleneck Linear (models/sts/ae.py:97-101,157) ---------------------------------- */

/* z[n][j] = bias[j] + sum_k W[j][k] * PReLU_slope(U[n][k]);  slope NULL: no activation. L <= 16. */
int coskad_btlnk_fwd_f32(const float* U, const float* W, const float* bias, const float* slope, float* z,
                         int B, int K, int L, hipStream_t stream);

/* The same as a split-K GEMM (blocks of 64 clips x 8 K slices, fixed-order sum of the partials) with a workspace of
 * coskad_btlnk_fwd_ws_bytes(B) bytes: the large-batch path; K must be a multiple of 16. */
size_t coskad_btlnk_fwd_ws_bytes(int B);
int coskad_btlnk_fwd_ws_f32(const float* U, const float* W, const float* bias, const float* slope, float* z, void* ws,
                            size_t ws_bytes, int B, int K, int L, hipStream_t stream);
size_t coskad_btlnk_bwd_ws_bytes(int B, int K, int L);
/* dU = (dz W) * PReLU'(U);  dW (+)= dz^T PReLU(U);  db (+)= sum_n dz;  dslope (+)= sum (dz W) U [U<0] */
int coskad_btlnk_bwd_f32(const float* U, const float* W, const float* dz, const float* slope, float* dU,
                         float* dW, float* db, float* dslope, void* ws, size_t ws_bytes, int accumulate,
                         int B, int K, int L, hipStream_t stream);

/* The same backward AND the batch reductions of the encoder's top ST_GCNN layer (models/graph_layers/stsgcn.py:94-116 under
 * autograd in training mode: P = sum dU Z^T, Q = sum dU PReLU(U_prev)^T, s = sum dU over clips and positions) from ONE pass
 * (csrc/btlnk_chain.hip: K tiled position-major, so the dU tile meets Z and U_prev of the same clips and positions on chip).
 * U is that layer's pre-activation [B, 64, TV] (K = 64 TV, TV % 4 == 0), below_in [B, below_Ci, TV] its input (pre-activation
 * of the layer below, below_in_slope its PReLU slope or NULL), below_Z [B, below_Ci, TV] its stored gcn output; below_Ci 16 / 32.
 * stats_out (8-byte aligned, coskad_btlnk_bwd_chain_floats(...) floats) receives the layer's backward chain buffer --
 * *stats_rows partial rows of 2 * 64 * below_Ci + 64 floats, then their fp64 sums -- which coskad_layer_bwd_chain_f32 takes
 * as `stats_in`.  ws: coskad_btlnk_bwd_chain_ws_bytes(B, K, L, TV) bytes.  Fixed-order sums: deterministic. */
int coskad_btlnk_bwd_chain_ok(int K, int TV, int below_Ci);
int coskad_btlnk_bwd_chain_rows(int B, int TV);
size_t coskad_btlnk_bwd_chain_floats(int B, int TV, int below_Ci);
size_t coskad_btlnk_bwd_chain_ws_bytes(int B, int K, int L, int TV);
int coskad_btlnk_bwd_chain_f32(const float* U, const float* W, const float* dz, const float* slope, float* dU, float* dW,
                               float* db, float* dslope, void* ws, size_t ws_bytes, int accumulate, int B, int K, int L,
                               const float* below_in, const float* below_Z, const float* below_in_slope, int below_Ci, int TV,
                               float* stats_out, size_t stats_out_bytes, int* stats_rows, hipStream_t stream);

/* ---- batch formation from the HBM-resident window table (callers' side of the path) ----- */

/* out[b, c, :] = M[t][c][0] x[s] + M[t][c][1] y[s] + M[t][c][2] with s = index[b] % N, t = index[b] / N, c < 2:
 * the item rule of utils/dataset.py:65-77 with the affine PoseTransform of utils/dataset_utils.py:272-310.
 * xy: [N, 2, TV], index: [B] int64, mats: [ntrans, 3, 3], out: [B, 2, TV]. */
int coskad_gather_transform_f32(const float* xy, const long long* index, const float* mats, float* out, int B, int N,
                                int ntrans, int TV, hipStream_t stream);

/* ---- one-class heads, centre statistics, regulariser, optimiser ------------------------ */

int coskad_head_slots(void);            /* floats in a stats / acc block (19) */
size_t coskad_head_ws_floats(int B);    /* scratch floats the head kernels need */

/* F.mse_loss(z, c) (euclidean_encoder_staticCenter.py:187, _dynamicCenter.py:116), its gradient,
 * the per-window score of utils/eval_utils.py:63-64, and the running-centre sums (:172-178).
 * stats: [0] loss, [1..L] sum_n z, [17] B, [18] sum_n |z_n|;  acc += raw sums. */
int coskad_mse_head_f32(const float* z, const float* c, float* dz, float* score, float* stats, float* acc,
                        float upstream, float* ws, int B, int L, hipStream_t stream);

/* Mahalanobis head: loss = mean_n sqrt((z_n-c)^T VI (z_n-c)) (utils/eval_utils.py:28-38, called at
 * euclidean_encoder_staticCenter.py:178-181), gradient, score (eval_utils.py:41-47), the MSE head's centre sums, and
 * gram[L*L] (+)= sum_n z_n z_n^T, from which batch_cov_mat_step / compute_inv_cov_mat (:40-46,133-142) follow. */
int coskad_mahalanobis_head_f32(const float* z, const float* c, const float* VI, float* dz, float* score,
                                float* stats, float* acc, float* gram, int gram_accumulate, float upstream,
                                float* ws, int B, int L, hipStream_t stream);

/* zh = project(expmap0(z)); loss = mean dist(c, zh) (hyperbolic_encoder.py:147,157 with the formulas
 * of utils/hyper_math.py:13-29,100-105,173-179,207-210,302-306), gradient w.r.t. z, score = dist,
 * stats: [0] loss, [1..L] sum gamma*zh, [17] sum (gamma-1), [18] sum |zh|;  c NULL: embed + sums only. */
int coskad_poincare_head_f32(const float* z, const float* c, float* dz, float* zh, float* score,
                             float* stats, float* acc, float upstream, float* ws, int B, int L,
                             hipStream_t stream);
int coskad_poincare_dist_f32(const float* zh, const float* c, float* score, int B, int L, hipStream_t stream);
/* out[n] = logmap0(y[n]) = y / |y| * artanh(|y|) for points on the unit Poincare ball (utils/hyper_math.py:367-370 at c = 1, with its
 * 1e-5 norm floor and Artanh's +-(1 - 1e-5) clamp, :18-24): the inverse of expmap0 that north_star names; the reference's wrappers
 * never call it, so it stands alone (no gradient entry). */
int coskad_poincare_logmap0_f32(const float* y, float* out, int B, int L, hipStream_t stream);

/* The statistics algebra of the folded first decoder layer (coskad_amd/lowrank.py: rev_btlnk, models/sts/ae.py:223-227, + the first
 * ST_GCNN layer of the decoder, models/graph_layers/stsgcn.py:106-110, on a rank-(latent + 1) input) for latent 8: X [2, 9, Co, TV] holds
 * the nine basis images of the two BatchNorm branches, G [9, 9] (fp64) the latents' Gram matrix sum_n [z, 1][z, 1]^T.
 *   fwd: batch statistics of both branches from G (fp64), running-statistics update (momentum, unbiased variance; conv biases shift
 *        the running means only), folded images in the rev_btlnk kernels' weight layout Mw [Co TV, 8], Mb [Co TV]; saved / xbar / xx:
 *        fp64 scratch of 6 Co / 18 Co / 162 Co doubles for the backward
 *   bwd: from dMw / dMb: dX [2, 9, Co, TV], dgamma [2][Co], dbeta [Co] (both BatchNorms'), dGc [Co][9][9] (its sum over Co is dG) */
int coskad_lowrank_fold_ok(int latent, int TV);
int coskad_lowrank_fold_fwd_f32(const float* X, const double* G, const float* gamma0, const float* beta0, const float* cbias0,
                                float* rmean0, float* rvar0, long long* nbt0, float momentum0, float eps0, const float* gamma1,
                                const float* beta1, const float* cbias1, float* rmean1, float* rvar1, long long* nbt1, float momentum1,
                                float eps1, double n_pos, float* Mw, float* Mb, double* saved, double* xbar, double* xx, int Co, int TV,
                                hipStream_t stream);
int coskad_lowrank_fold_bwd_f32(const float* X, const double* G, const float* dMw, const float* dMb, const double* saved,
                                const double* xbar, const double* xx, const float* gamma0, const float* gamma1, double n_pos, float* dX,
                                float* dgamma, float* dbeta, double* dGc, int Co, int TV, hipStream_t stream);

/* Narrow-output layers (C_out <= 4 behind 16 / 32 / 64 input channels: the decoder's last layer, models/common/components.py:143-179 ->
 * models/graph_layers/stsgcn.py:94-116) by commutation -- the mixing acts per channel, a 1x1 convolution per position, so
 * Wt gcn(X) = gcn(Wt X): both convolutions of the layer run first, as one streaming pass over the wide input, and mixing / BatchNorm /
 * PReLU see 2 C_out-channel tensors (csrc/last_layer.hip).
 *   fwd : out [B, J, TV] = W [J, Ci] . PReLU(U [B, Ci, TV])   (in_slope NULL: U is activated already); J = 2 C_out in {2, 4, 6, 8}
 *   bwd : dU [B, Ci, TV] = (W^T dOut) * PReLU'(U); partials [coskad_narrow_conv_rows(B, TV)][J Ci + 1]: per-workgroup sums of
 *         dW[j][c] = sum dOut_j PReLU(U)_c and, last column, of the producer's slope gradient -- add the rows with coskad_gemm_sum_f32 */
int coskad_narrow_conv_rows(int B, int TV);
int coskad_narrow_conv_fwd_f32(const float* U, const float* in_slope, const float* W, float* out, int B, int Ci, int J, int TV,
                               hipStream_t stream);
int coskad_narrow_conv_bwd_f32(const float* U, const float* in_slope, const float* W, const float* dOut, float* dU, float* partials,
                               size_t partials_floats, int B, int Ci, int J, int TV, hipStream_t stream);

/* coskad_layer_apply_z_f32 AND the next layer's statistics pass in one kernel on the 25-joint layout (what coskad_layer_apply_next_f32 is
 * at 17 joints; csrc/fused_apply_flat.hip): U = Wz.Z + Wx.PReLU_in(in) + b -> out; Z_next = gcn_next(PReLU_out(U)) [B, Co, T, V]; partials:
 * coskad_layer_apply_next_flat_rows(B) rows of [sum x x^T][sum x][sum z z^T][sum z] (2 (Co^2 + Co) floats) for
 * coskad_layer_train_fold_f32 (models/graph_layers/stsgcn.py:94-116 of layer i; 56-80 + the batch statistics of layer i + 1's
 * BatchNorms).  ok: 1 for 12 x 25, 16 / 32 -> 16 / 32 channels. */
int coskad_layer_apply_next_flat_ok(int Ci, int Co, int T, int V);
int coskad_layer_apply_next_flat_rows(int B);
int coskad_layer_apply_next_flat_f32(const float* Z, const float* in, float* out, const float* wfold, const float* bias,
                                     const float* in_slope, const float* out_slope, const float* A_next, const float* T_next,
                                     float* Z_next, float* partials, size_t partials_bytes, int B, int Ci, int Co, int T, int V,
                                     hipStream_t stream);

/* The FIRST two ST_GCNN layers of the encoder with folded BatchNorm (coskad_layer_apply_f32 twice: models/graph_layers/stsgcn.py:94-116
 * in eval mode, entries 0 and 1 of the nn.Sequential of models/common/components.py:94-105) in ONE pass -- x [B, 2, T, V] is the network
 * input, the first layer (2 -> 32) is formed per clip on the VALU inside the second layer's kernel and its output never reaches HBM
 * (csrc/eval_layer_bpc.hip).  ok: 1 when (n_frames, n_joints, C_in, C_mid, C_out) is built (12, 17 | 25, 2, 32, 16 | 32 | 64).
 * wfold1 [4, 32] / bias1, wfold2 [64, Co] / bias2 from coskad_bn_fold_f32; mid_slope: the first layer's PReLU weight; out_slope NULL:
 * pre-activation output. */
int coskad_layer_first_pair_ok(int T, int V, int Ci, int Cm, int Co);
int coskad_layer_first_pair_apply_f32(const float* x, float* out, const float* A1, const float* T1, const float* wfold1, const float* bias1,
                                      const float* A2, const float* T2, const float* wfold2, const float* bias2, const float* mid_slope,
                                      const float* out_slope, int B, int Cm, int Co, int T, int V, hipStream_t stream);

/* A (32 -> 16) ST_GCNN layer in training mode by the same commutation, on its own kernels (csrc/commute_layer.hip; replaces
 * models/graph_layers/stsgcn.py:94-116 and its autograd for such a layer on the 12 x 25 layout): Y = Wt X, R = Wr X first (X = PReLU(u_prev)),
 * Zy = gcn(Y); both BatchNorms become per-channel affine maps of Zy and R (batch statistics = row sums); the backward is one kernel per clip
 * behind two row-sum reductions.
 *   ok        : 1 when (n_frames, n_joints, C_in, C_out) is built (12, 25, 32, 16)
 *   ws_floats : floats of the scratch both entries need
 *   fwd       : writes YR [B, 32, TV] = [Y; R], Zy [B, 16, TV], U [B, 16, TV] (the layer's pre-activation output), stat [128]; updates the
 *               running statistics / num_batches_tracked of both BatchNorms as torch does (NULL: not tracked; momentum a number);
 *               A_next != NULL: the next layer's statistics pass in the same launches (Z_next = gcn_next(PReLU(U)), *rows_next rows of
 *               moment partials for coskad_layer_train_fold_f32; slope_out = this layer's PReLU weight)
 *   bwd       : dU [B, 16, TV] -> d_in [B, 32, TV] (PReLU mask of u_prev applied), dA [T, V, V], dT [V, T, T], dWt / dWr [16, 32],
 *               dgamma / dbeta [16] of both BatchNorms, dslope [1] (the producer's PReLU weight; NULL iff in_slope NULL): all overwritten;
 *               below_stats != NULL (a 2-channel layer in front, fed by the network input below_x, stored Z below_z): that layer's batch
 *               reductions too, as the chain buffer coskad_layer_bwd_chain_f32 takes (stats_in, coskad_commute_below_rows(B) rows,
 *               coskad_commute_below_floats(B) floats) */
int coskad_commute_ok(int T, int V, int Ci, int Co);
size_t coskad_commute_ws_floats(int B, int T, int V);
int coskad_commute_fwd_f32(const float* u_prev, const float* in_slope, const float* wt, const float* wr, const float* A, const float* Tm,
                           const float* gamma_t, const float* beta_t, const float* gamma_r, const float* beta_r, const float* bias_t,
                           const float* bias_r, float* rm_t, float* rv_t, float* rm_r, float* rv_r, long long* nbt_t, long long* nbt_r,
                           float momentum, float eps, float* YR, float* Zy, float* U, float* stat, float* ws, size_t ws_floats,
                           const float* A_next, const float* T_next, const float* slope_out, float* Z_next, float* partials_next,
                           int* rows_next, int B, int T, int V, hipStream_t stream);
int coskad_commute_bwd_f32(const float* u_prev, const float* in_slope, const float* wt, const float* wr, const float* A, const float* Tm,
                           const float* YR, const float* Zy, const float* stat, const float* dU, float* d_in, float* dA, float* dT,
                           float* dWt, float* dWr, float* dgamma_t, float* dbeta_t, float* dgamma_r, float* dbeta_r, float* dslope,
                           float* ws, size_t ws_floats, const float* below_x, const float* below_z, float* below_stats,
                           size_t below_stats_floats, int B, int T, int V, hipStream_t stream);
int coskad_commute_below_rows(int B);
size_t coskad_commute_below_floats(int B);

/* The spherical VAE's latent head between the raw outputs of fc_mean / fc_var and the decoder's input (models/sts/vae.py:79-91,
 * 104-118; loss terms of models/spherical_vae.py:86-94; PowerSpherical of the un-vendored `power_spherical` package, restated in
 * coskad_amd/models/sts/vae.py).  Rows are clips; mean_raw [B, L] and var_raw [B] are addressed with row strides ld_* (floats), so both
 * may be columns of ONE [B, L + 1] tensor.  L <= 16.
 *   prep  : mu = mean_raw / |mean_raw|, kappa = softplus(var_raw) + 1, concentration [B, 2] = ((L-1)/2 + kappa, (L-1)/2), total [B]
 *   (the caller draws x ~ Dirichlet(concentration) [B, 2] and eps ~ N(0, I) [B, L-1]; torch._dirichlet_grad gives the sampler's
 *    implicit reparameterisation gradient [B, 2] for the backward)
 *   sample: z [B, L] = Householder(e1 -> mu) applied to [2 x_0 - 1, sqrt(1 - t^2) eps / |eps|]; kl [B] = KL(PowerSpherical(mu, kappa) ||
 *           uniform on the sphere) per clip; inv_kappa [B] = 1 / kappa
 *   bwd   : d_mean_raw, d_var_raw from dz (gradient w.r.t. z) and the weights w_kl = d loss / d kl[n], w_exp = d loss / d inv_kappa[n] */
int coskad_ps_head_prep_f32(const float* mean_raw, int ld_mean, const float* var_raw, int ld_var, float* mu, float* kappa,
                            float* concentration, float* total, int B, int L, hipStream_t stream);
int coskad_ps_head_sample_f32(const float* x, const float* eps, const float* mu, const float* kappa, float* z, float* kl,
                              float* inv_kappa, int B, int L, hipStream_t stream);
int coskad_ps_head_bwd_f32(const float* dz, const float* x, const float* dirichlet_grad, const float* eps, const float* mu,
                           const float* kappa, const float* mean_raw, int ld_mean, const float* var_raw, int ld_var, float w_kl,
                           float w_exp, float* d_mean_raw, int ld_dmean, float* d_var_raw, int ld_dvar, int B, int L,
                           hipStream_t stream);

/* c = acc[1..L] / acc[17], then |c| < eps -> +-eps (staticCenter.py:118-121). */
int coskad_center_finalize_f32(const float* acc, float* c, float eps, int L, hipStream_t stream);
/* gyromidpoint from acc of coskad_poincare_head_f32 (hyperbolic_encoder.py:122,179). */
int coskad_midpoint_finalize_f32(const float* acc, float* c, int L, hipStream_t stream);

/* out[0] = scale * sum_i mask[i] p[i]^2  (utils/model_utils.py:90-105). ws: 256 floats. */
int coskad_sqnorm_f32(const float* p, const float* mask, size_t n, float scale, float* out, float* ws,
                      hipStream_t stream);
/* torch.optim.Adam step on flat buffers with g = grad*gscale + reg_coef*mask*p. */
int coskad_adam_f32(float* p, const float* g, float* m, float* v, const float* mask, size_t n, float lr,
                    float beta1, float beta2, float eps, int step, float gscale, float reg_coef,
                    hipStream_t stream);

/* coskad_adam_f32 with the caller's fp32 running products b1pow = beta1^t, b2pow = beta2^t instead of the step count: the
 * arithmetic of coskad_adam_dev_f32's device-side tick (an eager step and a hipGraph-captured one then agree bit for bit). */
int coskad_adam_pow_f32(float* p, const float* g, float* m, float* v, const float* mask, size_t n, float lr, float beta1,
                        float beta2, float eps, float b1pow, float b2pow, float gscale, float reg_coef, hipStream_t stream);

/* Adam with {lr, beta1^t, beta2^t} in device memory (hipGraph-replayable; init hyper = {lr, 1, 1}). */
int coskad_adam_dev_f32(float* p, const float* g, float* m, float* v, const float* mask, size_t n, float* hyper,
                        float beta1, float beta2, float eps, float gscale, float reg_coef, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* COSKAD_HIP_H */
