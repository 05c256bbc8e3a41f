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

#ifdef __cplusplus
}
#endif
#endif /* COSKAD_HIP_H */
