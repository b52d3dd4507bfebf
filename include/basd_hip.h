/*
 * basd_hip.h -- C-ABI of the MI355X (gfx950) BASD loss-path kernels.
 *
 * Every entry takes raw DEVICE pointers, explicit shapes and a hipStream_t
 * (passed as void*), never allocates / frees / synchronises the device, and
 * returns an int status (0 = ok).  basd_last_error() returns a thread-local
 * message for the last non-zero status.  No C++ exception crosses this
 * boundary, no torch type appears in any signature.
 *
 * The reference project is pure Python and has no FFI layer; the entries
 * below replace the library calls its loss path makes (citations are
 * file:line in the reference repository):
 *
 *   basd_token_gram        src/losses/layer_selector.py:72,135 (projection GEMM)
 *                          + :13 (X^T X) + :35 (column mean)
 *   basd_pchol_f64 +
 *   basd_jacobi_svd        torch.linalg.eigvalsh  layer_selector.py:16
 *                          torch.linalg.svd       layer_selector.py:36,92
 *                          torch.linalg.svdvals   layer_selector.py:99
 *                          matrix_norm(ord="nuc") src/losses/relational.py:48
 *   basd_mp_rank           layer_selector.py:17-20 (.median/.sum, on device, no .item())
 *   basd_angle_weights     layer_selector.py:100-108 (acos / spectral weighting / softmax over teacher layers)
 *   basd_angle_weights_bwd autograd of layer_selector.py:86-108 (student frame, angles, softmax, temperatures)
 *   basd_mix_tokens        layer_selector.py:110-112 (+ torch.stack :128-129 eliminated)
 *   basd_procrustes_prep   src/losses/relational.py:29-46, src/losses/combined.py:9-14
 *   basd_mix_grad_dots     autograd of layer_selector.py:111-112 w.r.t. the mixing weights
 *   basd_gemm_bf16         timm nn.Linear (+ nn.GELU) forward of the ViT blocks, teacher.py:212 / trainer.py:33
 *   basd_gemm_bf16_gelu_fwd, basd_gemm_bf16_gelu_bwd
 *                          timm Mlp (fc1 -> nn.GELU -> fc2) of the TRAINED student: forward and autograd (trainer.py:33,157)
 *   basd_gemm_bf16x3_f32   the teacher-token projection tokens.reshape(-1, D_t) @ proj_t.T (layer_selector.py:72 / :135)
 *                          at student widths > 256
 *   basd_transpose_bf16_table
 *                          the W^T operands autograd's nn.Linear backward forms per call (trainer.py:157)
 *   basd_procrustes_fwd    src/losses/relational.py:47-48 (cross-covariance, nuclear norm, U V^T) as one call
 *   basd_procrustes_bwd    autograd of relational.py:29-48 (svd_backward of the nuclear norm + centring / weighting)
 *   basd_ce_uwso           nn.CrossEntropyLoss(label_smoothing) (trainer.py:47) + UW-SO (src/losses/combined.py:57,78-85)
 *   basd_wgrad_bf16        autograd of the student's timm nn.Linear layers (trainer.py:157)
 *   basd_sf_adamw_step     schedulefree.AdamWScheduleFree.step  src/training/trainer.py:54-58,158
 *   basd_bgemm_f64, basd_trinv_f64
 *                          the products / triangular solves around the SVD of relational.py:47-48 and its backward
 *                          (torch.linalg.svd backward -> U V^T), formed in fp64
 *   basd_procrustes_bwd_rows
 *                          autograd of relational.py:29-46 (centring, sqrt-weights, traces)
 *   basd_attention_bwd_bf16
 *                          autograd through timm Attention.forward of the student (trainer.py:157)
 *   basd_attention_fwd_bf16, basd_attention_fwd_qmean_bf16, basd_cls_importance_bf16
 *                          timm Attention.forward of the frozen teacher (teacher.py:118 creates it) and the
 *                          attention capture hook src/models/teacher.py:27-39 + relational.py:22-27
 *   basd_layernorm_fwd_bf16 / _bwd_bf16, basd_add_layernorm_fwd_bf16
 *                          timm nn.LayerNorm of the ViT blocks (student: with backward; frozen teacher: fused with
 *                          the residual add in front of it)
 */
#ifndef BASD_HIP_H
#define BASD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BASD_OK 0
#define BASD_ERR_SHAPE 1        /* bad / unsupported shape                       */
#define BASD_ERR_LAUNCH 2       /* HIP launch error                              */
#define BASD_ERR_WORKSPACE 3    /* workspace too small                           */
#define BASD_ERR_DTYPE 4        /* unsupported dtype code                        */

/* Device-side health flags: kernels whose failure is data dependent OR them (atomically) into an optional
 * caller-owned int32 word instead of returning them -- no entry ever synchronises with the device.  The host
 * side reads the word when it chooses to (the Python binding: once per step, one step late, and raises a
 * torch.linalg.LinAlgError, which is what the reference's torch.linalg calls raise on such input). */
#define BASD_STATUS_NONCONVERGED 1   /* a Jacobi solve used all max_sweeps sweeps and was still rotating      */
#define BASD_STATUS_NONFINITE 2      /* NaN / Inf among the singular values (non-finite input)               */
#define BASD_STATUS_RANK0 4          /* Marchenko-Pastur rank 0: the reference divides by sum(sw) = 0 here   */
/* bits 8 .. 27: diagnostics OR-ed in with BASD_STATUS_NONCONVERGED (informational): bits 8-11 the Jacobi kernel variant
 * (1 LDS-resident, 2 / 3 odd-even with a double / single mailbox, 4 two matrices per workgroup, 5 block ordering),
 * bits 12-27 the index of the matrix in its launch (saturating at 65535) */

#define BASD_DTYPE_F32 0
#define BASD_DTYPE_BF16 1
#define BASD_DTYPE_F64 2

/* largest matrix the LDS-resident Jacobi accepts: cols * ld(rows) * 4 <= 160 KiB - scratch */
#define BASD_JACOBI_MAX_COLS 256
#define BASD_JACOBI_LDS_BYTES 163840

int basd_version(void);
const char* basd_last_error(void);

/* z = X P^T (fp32 MFMA), gram = z^T z and colsum = 1^T z accumulated in fp64.
 * x: [rows, d_in] row-major (dtype code), proj: [d_out, d_in] fp32 row-major,
 * gram: [d_out, d_out] fp64 (MUST be zeroed by the caller), colsum: [d_out] fp64 (zeroed).
 * Only the LOWER 16x16 tiles of gram are accumulated (fp64 atomics are the kernel's HBM write
 * traffic: 41 MB per call at d_out = 192); the caller mirrors them.
 * d_out <= 256 and d_out % 16 == 0; d_in % 32 == 0.
 * Strided token views: logical row r is token r % rows_per_batch of batch r / rows_per_batch and
 * lives at x + (r / rows_per_batch) * batch_stride + (r % rows_per_batch) * d_in (elements); a
 * contiguous matrix is rows_per_batch = rows, batch_stride = 0.  This is how the CLS-stripped view
 * out[:, 1:, :] of a block output is consumed without a copy. */
int basd_token_gram(const void* x, int x_dtype, int64_t rows, int d_in,
                    int rows_per_batch, int64_t batch_stride,
                    const float* proj, int d_out,
                    double* gram, double* colsum, void* stream);

/* bf16-input fast path of basd_token_gram: x [rows, d_in] bf16, proj_split [3, d_out, d_in] bf16 =
 * the three-term bf16 split of the fp32 projection (P = Ph + Pm + Pl, each term the bf16
 * rounding of the remainder).  Same outputs / zeroing contract.  d_out in {32, 64, 128, 192},
 * d_in % 32 == 0. */
int basd_token_gram_bf16x3(const void* x, int64_t rows, int d_in, int rows_per_batch, int64_t batch_stride,
                           const void* proj_split, int d_out, double* gram, double* colsum, void* stream);

/* Gram of a tall fp32 matrix z [rows, d] (row-major, 16-byte aligned, d % 16 == 0): gram [d, d] fp64 += z^T z (lower
 * 16 x 16 tiles only), colsum [d] fp64 += column sums; both MUST be zeroed by the caller (atomics).  Tile-centred: per
 * 64 rows the column means are subtracted, the centred tile runs on the fp32 matrix cores and the exact uncentred
 * statistics are restored with fp64 rank-1 terms (the arithmetic of the fused kernel's Gram phase).  The statistics of
 * src/losses/layer_selector.py:13,35 at student widths beyond 256 (BASELINE c4 / c5), after basd_gemm_bf16x3_f32. */
int basd_gram_f32_centred(const float* z, int64_t rows, int d, double* gram, double* colsum, void* stream);

/* Pivoted (diagonal pivoting) Cholesky of `batch` symmetric PSD fp64 matrices
 * a[b] (n x n).  Writes
 *   w0   [batch, n, ld] fp32: column k (contiguous, ld floats) = k-th Cholesky column,
 *        rows in ORIGINAL order (a[b] = W0 W0^T), columns >= rank zeroed;
 *   lwork[batch, n, n] fp64: the same columns in fp64 (column-major by step);
 *   piv  [batch, n] int32: pivot row chosen at step k (a permutation of 0..n-1);
 *   rank [batch] int32: number of pivots > tol * dmax, dmax = the largest diagonal entry of a[b], or
 *        dmax_ref[b] when dmax_ref (device fp64 [batch], nullable) is given: the panels of a BLOCKED
 *        factorisation of a wider matrix stop relative to the diagonal of the whole matrix, not of their
 *        own Schur complement.
 * Only the LOWER triangle of a[b] (row >= column) is read.  n <= 256. */
int basd_pchol_f64(const double* a, int batch, int n, double tol, const double* dmax_ref,
                   float* w0, int ld, double* lwork, int32_t* piv, int32_t* rank,
                   void* stream);
/* The same with a problem mask: skip (device int32 [batch], nullable): skip[b] != 0 leaves every output of matrix b
 * untouched (its workgroup returns at once).  For batched pipelines whose set of live problems is known on the DEVICE
 * only -- the block pairs of the wide students' eigensolver that contain an all-zero block (rank-masked
 * principal-angle problems, src/losses/layer_selector.py:84-92): no host sync, no re-packing of the batch. */
int basd_pchol_f64_masked(const double* a, int batch, int n, double tol, const double* dmax_ref,
                          float* w0, int ld, double* lwork, int32_t* piv, int32_t* rank,
                          const int32_t* skip, void* stream);

/* Inverse of the pivoted Cholesky factor left by basd_pchol_f64, fp64, packed triangle in LDS:
 * out[b] = L_p^-1 P  ([n, n] row-major; P = pivot permutation, L_p = P L lower triangular), so
 * out @ M == L_p^-1 (P M) for M with rows in ORIGINAL order.  Rows >= rank[b] are zero.  n <= 208 (blocked on the fp64
 * matrix cores up to 192; 193 .. 208: blocked leading 192 rows + border rows). */
int basd_trinv_f64(const double* lwork, const int32_t* piv, const int32_t* rank, int batch, int n,
                   double* out, void* stream);
/* skip[b] != 0: out[b] untouched (see basd_pchol_f64_masked) */
int basd_trinv_f64_masked(const double* lwork, const int32_t* piv, const int32_t* rank, int batch, int n,
                          double* out, const int32_t* skip, void* stream);

/* One-sided (Hestenes) Jacobi in LDS on `batch` column-major matrices
 * w[b]: n_cols columns of `ld` floats, first m_rows rows significant
 * (rows m_rows..ld-1 must be zero).  Columns are orthogonalised in place;
 * on return column c holds sigma_c * u_c.  If sort != 0 columns are permuted
 * so that norms descend.  sigma [batch, n_cols] receives the column norms of
 * the first `norm_rows` rows (norm_rows = m_rows normally; for a stacked
 * [A; I] input pass the row count of A so that the bottom block -- the
 * accumulated right singular vectors -- is excluded from the norm).
 * sweeps [batch] (optional, may be NULL) receives the sweeps used, NEGATED if the matrix was still rotating when
 * max_sweeps was reached (with max_sweeps = 1: "this visit still made a large rotation").
 * active (optional, device int32 [batch], may be NULL): matrix b has non-zero entries only in its
 * leading active[b] columns (and rows, if active_rows != 0); the sweeps then run over that
 * block only (rank-masked principal-angle problems, no host sync on the ranks).  active[b] < 0 skips matrix b
 * altogether (w, sigma untouched, sweeps[b] = 0; register-resident single-matrix kernels, i.e. batch < 512 or tall /
 * single-mailbox shapes): converged matrices of a block tournament cost a 5 us launch instead of a sweep.
 * active_rows == 2 declares `active` a pure MASK -- every entry is either < 0 (skip) or n_cols (solve completely) --
 * which the block-ordering kernel of the large batches honours too.
 * status (optional, device int32 word, may be NULL): BASD_STATUS_NONCONVERGED / BASD_STATUS_NONFINITE are OR-ed in.
 * Requires n_cols <= 256, ld % 4 == 0 and either n_cols * ld * 4 + 4096 <= 160 KiB (matrix resident in LDS) or the
 * register-resident forms: m_rows <= 224, or m_rows <= 384 with n_cols <= 192 (tall block pairs; with max_sweeps = 1
 * and sort = 0 this is one visit of a block-Jacobi tournament over wider matrices). */
int basd_jacobi_svd(float* w, int batch, int m_rows, int n_cols, int ld, int norm_rows,
                    float tol, int max_sweeps, int sort,
                    float* sigma, int32_t* sweeps, const int32_t* active, int active_rows,
                    int32_t* status, void* stream);

/* Marchenko-Pastur rank on device (no host sync).  evals [batch, n] (any order),
 * rows = M of the [M, D] token matrix, d = D; uses the min(M, D) largest eigenvalues,
 * LOWER median, lambda_+ = med * (1 + sqrt(D/M))^2, count > lambda_+, clamp to cap.  n <= 1024.
 * status (optional device int32 word): BASD_STATUS_RANK0 when a count is 0, BASD_STATUS_NONFINITE for a NaN spectrum. */
int basd_mp_rank(const float* evals, int batch, int n, int64_t rows, int d, int cap,
                 int32_t* ranks, int32_t* status, void* stream);

/* Data-dependent failure raised from host-orchestrated checks (the blocked eigensolver's orthogonality test): ORs
 * `bit` (a BASD_STATUS_* value) into the device health word if any of the n fp64 values is NOT <= tol -- a NaN raises it
 * too -- with an atomic OR (the word is shared with kernels of other streams).  Counterpart of the
 * torch._C._LinAlgError the reference's torch.linalg calls raise (layer_selector.py:16,36,92,99). */
int basd_flag_if_exceeds_f64(const double* values, int64_t n, double tol, int bit, int32_t* status, void* stream);

/* Principal-angle distances -> mixing weights of the Grassmannian selector, fused (reference
 * src/losses/layer_selector.py:100-108).  sigma [E, L, D] fp32: cosines of the principal angles between student
 * extraction point i and teacher layer j (descending; zero beyond the layer's rank), sw [L, D] fp32: the teacher's
 * singular values masked at the MP rank, log_temp [E].  Outputs (fp32): d2 [E, L] = sum_m sw theta^2 / sum_m sw with
 * theta = acos(min(sigma, 1 - eps)); pre [E, L] = -d2 / softplus(log_temp); weights [E, L] = softmax_j(pre);
 * coef [E, L, D] = (d d2 / d sigma) / sigma (divided by sigma^2 once more when unnormalised != 0: the caller's
 * singular vectors are then the Jacobi's sigma_m u_m columns) -- the diagonal of the backward seed Phi.  L <= 64. */
int basd_angle_weights(const float* sigma, const float* sw, const float* log_temp, int E, int L, int D,
                       int unnormalised, float* d2, float* pre, float* weights, float* coef, void* stream);

/* Backward of the selector weights (autograd of src/losses/layer_selector.py:86-108 w.r.t. the student tokens and the
 * temperatures).  g_w [E, L]: gradient w.r.t. the mixing weights; g_pre_out [E, L] (nullable): gradient w.r.t. the
 * pre-softmax values; weights, d2 [E, L], log_temp [E]: forward values; t_seed [E, L, D, D] fp32: the forward's seeds
 * A_full A_bar^T Phi (rows b >= k_j only); v_s [E, D, D] fp32 (rows = eigenvectors of the student's centred Gram),
 * lam_s [E, D] fp64 (its eigenvalues); proj_s [D, D_s] fp32.  Outputs: g_log_temp [E] fp32, w_tok [E, D_s, D_s] fp32 with
 * d loss / d s_i = (s_i - column mean) w_tok[i].  One reduction launch, four basd_bgemm_f64 products, one store, on a
 * workspace of basd_angle_weights_bwd_workspace_bytes(E, D, D_s) bytes.  L <= 64. */
int64_t basd_angle_weights_bwd_workspace_bytes(int E, int D, int D_s);
int basd_angle_weights_bwd(const float* g_w, const float* g_pre_out, const float* weights, const float* d2,
                           const float* log_temp, const float* t_seed, const float* v_s, const double* lam_s,
                           const float* proj_s, int E, int L, int D, int D_s, float* g_log_temp, float* w_tok,
                           void* workspace, int64_t workspace_bytes, void* stream);

/* mixed[i] = sum_j w[i, j] * x_j   (all E mixes from ONE pass over the teacher layers)
 * x_layers: HOST array of L device pointers to [elems] tensors (dtype code; the pointers are
 * passed to the kernel by value, no device-side table), w: [E, L] fp32,
 * out: [E, elems] fp32 (contiguous).  Each source layer is a batch-strided view: logical element e
 * is at layer + (e / per_batch) * batch_stride + e % per_batch (contiguous: per_batch = elems). */
int basd_mix_tokens(const void* const* x_layers, int x_dtype, int L, int E,
                    const float* w, int64_t elems, int64_t per_batch, int64_t batch_stride,
                    float* out, void* stream);

/* Procrustes prep for `batch` = B samples of one extraction point:
 * s [B, N_s, D_s] (dtype; sample b at s + b * s_batch_stride), t [B, N_t, D_t] fp32 (mixed teacher),
 * imp [B, N_t] fp32.
 * Resamples t and imp to N_s (2-tap linear, align_corners=False), normalises imp,
 * weighted-centres and sqrt-weights:
 *   s_w [B, N_s, D_s], t_w [B, N_s, D_t] fp32, a [B, N_s], tr [B, 2] = (tr_s, tr_t). */
int basd_procrustes_prep(const void* s, int s_dtype, int64_t s_batch_stride, const float* t, const float* imp,
                         int B, int N_s, int N_t, int D_s, int D_t,
                         float* s_w, float* t_w, float* a, float* tr, void* stream);

/* dots[i, j] = sum_e g[i, e] * x_j[e]  for all (i, j) from ONE pass over the teacher layers.
 * g: [E, elems] fp32, dots: [E, L] fp64 (MUST be zeroed by the caller; fp64 atomics keep the
 * result independent of the arrival order to ~1e-16). */
int basd_mix_grad_dots(const void* const* x_layers, int x_dtype, int L, int E,
                       const float* g, int64_t elems, int64_t per_batch, int64_t batch_stride,
                       double* dots, void* stream);

/* Batched GEMM with fp64 accumulation on the fp64 matrix cores: C[b] = op(A[b]) op(B[b]).
 * Row-major, explicit leading dimensions and batch strides (in elements); A/B dtype F32 or F64,
 * C dtype F32 or F64; trans_x != 0 means the stored matrix is the transpose of op(X).
 * symmetric != 0 (M == N, result known symmetric, e.g. X X^T): only the lower 64x64 tiles are
 * computed and mirrored on store.
 * Replaces the torch.bmm / matmul calls of the Procrustes core that must not round to fp32
 * (reference src/losses/relational.py:47 forms the cross-covariance in fp32). */
int basd_bgemm_f64(const void* a, int a_dtype, int64_t a_stride, int lda, int trans_a,
                   const void* b, int b_dtype, int64_t b_stride, int ldb, int trans_b,
                   void* c, int c_dtype, int64_t c_stride, int ldc,
                   int batch, int M, int N, int K, int symmetric, void* stream);
/* skip[b] != 0: c[b] untouched (see basd_pchol_f64_masked) */
int basd_bgemm_f64_masked(const void* a, int a_dtype, int64_t a_stride, int lda, int trans_a,
                          const void* b, int b_dtype, int64_t b_stride, int ldb, int trans_b,
                          void* c, int c_dtype, int64_t c_stride, int ldc,
                          int batch, int M, int N, int K, int symmetric, const int32_t* skip, void* stream);

/* ViT linear layer on the bf16 matrix cores with a fused epilogue (timm nn.Linear + nn.GELU of the blocks; reference
 * call sites src/models/teacher.py:212 and src/training/trainer.py:33; with w := W^T also the input gradient
 * dX = dY W of trainer.py:157):   y[m][n] = epi(sum_k x[m][k] w[n][k] + bias[n]),
 * x [M, K], w [N, K], y [M, N], bias [N] (nullable) all bf16 row-major, fp32 accumulation.
 * epilogue: 0 = none, 1 = + bias, 2 = exact-erf GELU(+ bias).  K % 64 == 0, N a multiple of 256, 192 or 128.
 * tile_run: how many output tiles a workgroup of the persistent kernel (large M, N % 256 == 0, K >= 192) multiplies
 * before it retires: 0 = its whole share (one workgroup per CU for the whole launch: fastest when the GEMM has the GPU
 * to itself), k >= 1 = at most k (the CUs come free every k tiles: for launches that share the GPU with another
 * stream; the pipelined BASD step uses 1 for the frozen teacher's launches and 0 for the student's own), k < 0 = fully
 * persistent on -k (8 .. 32) workgroups per XCD, i.e. the launch leaves the other CUs to another stream for its whole
 * duration (measured at -16 .. -28: equal to 1 within the noise; -12: the teacher branch becomes the critical path).  The
 * results do not depend on it. */
int basd_gemm_bf16(const void* x, const void* w, const void* bias, void* y, int64_t M, int N, int K,
                   int epilogue, int tile_run, void* stream);

/* Teacher-statistics projection z = X P^T at student widths > 256 (src/losses/layer_selector.py:72 / :135: the fixed fp32 projection of
 * the teacher tokens in front of the subspace SVD): x [M, K] bf16 row-major, w3 [ceil(N / 256) * 256, 3 K] bf16 = the
 * (hi | mid | lo) bf16 splits of P's rows side by side (zero rows behind N), y [M, N] fp32.  One persistent bf16-MFMA
 * GEMM of depth 3 K, fp32 accumulation over all three splits.  K % 64 == 0, K >= 128, N % 8 == 0, M > 1792. */
int basd_gemm_bf16x3_f32(const void* x, const void* w3, float* y, int64_t M, int N, int K, void* stream);

/* The trained student's MLP (timm Mlp: fc1 -> nn.GELU -> fc2, trainer.py:33 / :157) without a separate GELU pass:
 *   fwd:  pre[m][n] = bf16(sum_k x[m][k] w[n][k] + bias[n])   (saved for backward),   y = bf16(gelu(pre))
 *   bwd:  dpre[m][n] = bf16((sum_k dy[m][k] wt[n][k]) * gelu'(pre[m][n]))   with wt = fc2.weight^T  [N = hidden, K = out]
 * exact-erf GELU and its derivative Phi(x) + x phi(x), evaluated in fp32 on the bf16-rounded pre-activation (what
 * nn.GELU and its autograd see behind a bf16 nn.Linear).  Shapes and tile_run as basd_gemm_bf16. */
int basd_gemm_bf16_gelu_fwd(const void* x, const void* w, const void* bias, void* pre, void* y, int64_t M, int N,
                            int K, int tile_run, void* stream);
int basd_gemm_bf16_gelu_bwd(const void* dy, const void* wt, const void* pre, void* dpre, int64_t M, int N, int K,
                            int tile_run, void* stream);

/* ViT weight-gradient GEMM (backward of nn.Linear): dw[n][k] += sum_m dy[m][n] x[m][k],
 * db[n] += sum_m dy[m][n] (db may be NULL).  dy [M, N], x [M, K] bf16 row-major, dw [N, K] /
 * db [N] fp32, ACCUMULATED with atomics (caller zero-initialises or accumulates on purpose).
 * N % 64 == 0, K % 64 == 0. */
int basd_wgrad_bf16(const void* dy, const void* x, int64_t M, int N, int K, float* dw, float* db,
                    void* stream);

/* The same with a caller-provided workspace: shapes with N % 192 == 0, K % 192 == 0, M >= 4096 (every ViT student
 * width) run as 192 x 192 tiles whose per-slab partial sums go through the workspace and a reduction launch instead of
 * 256-way fp32 atomics (which execute at the memory side: 31 - 48 us per launch).  basd_wgrad_workspace_bytes returns
 * the bytes this shape needs (0: no workspace used; <= 37.75 MB); the workspace is scratch, 16-byte aligned, and may
 * be shared by all launches of a stream. */
int64_t basd_wgrad_workspace_bytes(int64_t M, int N, int K);
int basd_wgrad_bf16_ws(const void* dy, const void* x, int64_t M, int N, int K, float* dw, float* db,
                       void* workspace, int64_t workspace_bytes, void* stream);

/* LayerNorm of the ViT blocks (timm nn.LayerNorm, eps 1e-6): bf16 activations in/out, fp32 gamma/beta,
 * fp32 statistics.  fwd saves mean / rstd [rows]; bwd writes dx and ACCUMULATES dgamma / dbeta
 * (fp32 atomics; pass NULL for a frozen layer).  D % 8 == 0, D <= 2048. */
int basd_layernorm_fwd_bf16(const void* x, const float* gamma, const float* beta, int64_t rows, int D,
                            float eps, void* y, float* mean, float* rstd, void* stream);
/* Pre-norm residual step of a block: sum_out = bf16(residual + scale * x), y = LayerNorm(sum_out).
 * row_scale (nullable, fp32 [rows / rows_per_scale]): per-SAMPLE scale of the branch x = the stochastic-depth keep
 * mask divided by the keep probability (timm DropPath); NULL = 1.  mean / rstd may be NULL (frozen block). */
int basd_add_layernorm_fwd_bf16(const void* x, const void* residual, const float* gamma, const float* beta,
                                int64_t rows, int D, float eps, void* sum_out, void* y, float* mean,
                                float* rstd, const float* row_scale, int rows_per_scale, void* stream);
/* dx = LayerNorm backward (+ dres when given: the gradient that arrives through the residual connection of a pre-norm
 * block, so dx is the whole gradient of the block's residual stream); dbranch (nullable) = row_scale * dx, the
 * gradient of the branch input of basd_add_layernorm_fwd_bf16.  dgamma / dbeta are accumulated (nullable). */
int basd_layernorm_bwd_bf16(const void* dy, const void* x, const float* gamma, const float* mean,
                            const float* rstd, int64_t rows, int D, void* dx, float* dgamma, float* dbeta,
                            const void* dres, void* dbranch, const float* row_scale, int rows_per_scale,
                            void* stream);

/* Row epilogue of the Procrustes backward (reference src/losses/relational.py:22-45 differentiated):
 * p [rows, D] = (other side) G^T from the GEMM, w [rows, D] the weighted centred tokens, R = w - p,
 * a [rows] the normalised importance, gl [rows / rows_per_batch] the incoming gradient per sample ->
 *   out[row, :] = 2 gl sqrt(a[row]) R[row, :]   (fp32, may alias p, or bf16)
 *   rowdot[row] = 2 gl sum_d R[row, d] w[row, d].          D % 4 == 0. */
int basd_procrustes_bwd_rows(const float* p, const float* w, const float* a, const float* gl, int64_t rows,
                             int rows_per_batch, int D, void* out, int out_dtype, float* rowdot, void* stream);

/* Teacher attention tap (reference src/models/teacher.py:33-37 hook + src/losses/relational.py:22-27):
 * qkv [B, T, 3, H, hd] bf16 (the packed output of a block's qkv projection, CLS token first) ->
 * out[b, t-1] = mean_h softmax_t(bf16(q_cls . k_t) * scale), t = 1..T-1, fp32.  T <= 256, hd in {32, 64}. */
int basd_cls_importance_bf16(const void* qkv, int B, int T, int H, int hd, float scale, float* out,
                             void* stream);

/* Fused attention forward of a frozen block (inference): qkv [B, T, 3, H, hd] bf16 (packed projection) ->
 * out [B, T, H * hd] bf16 = softmax(Q K^T * scale) V per head (fp32 logits / probabilities, P rounded to bf16
 * for the second product, like the library flash kernel it replaces on the teacher: torch SDPA called at
 * timm Attention.forward).  importance (nullable): [B, H, T-1] fp32; receives per head the CLS-row softmax of
 * basd_cls_importance_bf16 divided by H (the tap is the sum over the H axis).  lse (nullable): [B, H, T] fp32,
 * log-sum-exp of the scaled logits per query (input of basd_attention_bwd_bf16).  hd == 64, T <= 272. */
int basd_attention_fwd_bf16(const void* qkv, int B, int T, int H, int hd, float scale, void* out,
                            float* importance, float* lse, void* stream);

/* The same forward for a teacher WITHOUT a CLS token (reference src/losses/relational.py:25-27: the tap is the
 * attention map averaged over heads and QUERIES): importance [B, H, T] fp32 receives per head
 * sum_q softmax(q . k * scale)[q][key] / (H T) -- the tap is the sum over the H axis.  Same shapes as above. */
int basd_attention_fwd_qmean_bf16(const void* qkv, int B, int T, int H, int hd, float scale, void* out,
                                  float* importance, void* stream);

/* Fused attention backward of a trained block (the student; reference src/training/trainer.py:157 through timm
 * Attention.forward): qkv [B, T, 3, H, hd] bf16, out / dout [B, T, H * hd] bf16 (forward output and its gradient),
 * lse [B, H, T] fp32 from basd_attention_fwd_bf16 -> dqkv [B, T, 3, H, hd] bf16, the gradient of the packed
 * projection.  P is recomputed (nothing T x T is stored); fp32 accumulation, P and dS rounded to bf16 for the second
 * products like a flash kernel.  hd == 64, T <= 224. */
int basd_attention_bwd_bf16(const void* qkv, const void* out, const void* dout, const float* lse, int B, int T, int H,
                            int hd, float scale, void* dqkv, void* stream);

/* Forward of the attention-weighted Procrustes term between basd_procrustes_prep and the loss value, as one chain of
 * launches on a caller-provided workspace (src/losses/relational.py:47-48: cross = s_w^T t_w, its nuclear norm, and
 * U V^T for the backward; cross is never formed).  s_w [batch, n, d_s], t_w [batch, n, d_t] fp32 (prep outputs);
 * tol: relative stop of the pivoted Cholesky factorisations (1e-13).  Outputs (fp32):
 *   nuc [batch]: nuclear norm of cross;   a_t [batch, n, n]:  s_w (U V^T) = a_t t_w;
 *   fac_s: n > d_s  ("feature side"): [batch, n, d_s] = t_w (U V^T)^T;
 *          n <= d_s ("token side"):   [batch, n, n]   = a_s with t_w (U V^T)^T = a_s s_w.
 * status: device health word (BASD_STATUS_* OR-ed in by the Jacobi solve; nullable).  min(n, d_s) <= 256.
 * basd_procrustes_workspace_bytes gives the workspace size (256-byte aligned scratch; 5 or 10 fp64 [batch, n, n]
 * matrices plus the fp32 Jacobi image). */
int64_t basd_procrustes_workspace_bytes(int batch, int n, int d_s, int d_t);
int basd_procrustes_fwd(const float* s_w, const float* t_w, int batch, int n, int d_s, int d_t, double tol,
                        float* nuc, float* fac_s, float* a_t, int32_t* status, void* workspace,
                        int64_t workspace_bytes, void* stream);

/* Backward of the same term (autograd of src/losses/relational.py:29-48: svd_backward of the nuclear norm = the polar
 * factor, then the centring / weighting), from the factors basd_procrustes_fwd saved.  Per side, with W the weighted
 * tokens [batch, n, d] and fac [batch, n, n] (a_t for the teacher side, a_s for the student side of the token-side form):
 *     out[b, i, :]  = 2 gl[b] sqrt(a[b, i]) (W - fac W)[b, i, :]        (fp32 or bf16)
 *     rowdot[b, i]  = 2 gl[b] <(W - fac W)[b, i, :], W[b, i, :]>
 * in one launch: the product runs on the bf16 matrix cores as a three-product split of both fp32 operands (relative
 * error 2^-16), residual / scaling / row dots in its epilogue.  4 <= n <= 256, n % 4 == 0, d % 16 == 0, 16-byte
 * aligned buffers. */
int basd_procrustes_bwd_side(const float* fac, const float* w, const float* a, const float* gl, int batch, int n, int d,
                             void* out, int out_dtype, float* rowdot, void* stream);

/* The whole backward: s_w [batch, n, d_s], t_w [batch, n, d_t], a [batch, n] (basd_procrustes_prep outputs), gl [batch]
 * (d loss / d value), fac_s / a_t (basd_procrustes_fwd outputs; fac_s is [batch, n, n] when n <= d_s, else
 * [batch, n, d_s]) -> g_s [batch, n, d_s] (g_s_dtype: BASD_DTYPE_F32 | BASD_DTYPE_BF16), g_t [batch, n, d_t] fp32
 * (gradients w.r.t. the RAW student tokens and the resampled teacher tokens), g_a [batch, n] (w.r.t. the normalised
 * importance).  workspace: >= basd_procrustes_bwd_workspace_bytes(batch, n) bytes (the two row-dot vectors). */
int64_t basd_procrustes_bwd_workspace_bytes(int batch, int n);
int basd_procrustes_bwd(const float* s_w, const float* t_w, const float* a, const float* gl, const float* fac_s,
                        const float* a_t, int batch, int n, int d_s, int d_t, void* g_s, int g_s_dtype, float* g_t,
                        float* g_a, void* workspace, int64_t workspace_bytes, void* stream);

/* Cross-entropy with label smoothing on soft targets [B, C] (MixUp / CutMix) or class indices [B] (exactly one of the
 * two non-NULL), its gradient, and the UW-SO combination with the Procrustes term (src/losses/combined.py:57,78-85;
 * nn.CrossEntropyLoss(label_smoothing) of trainer.py:47):  ce = mean_b -sum_c t'_bc log_softmax(z_b)_c,
 * w_ce = (1/ce) / (1/ce + 1/geo), w_geo = 1 - w_ce (detached, clamped at eps), total = w_ce ce + w_geo geo.
 * geo: DEVICE scalar (NULL: plain CE, w_ce = 1).  Outputs: dlogits [B, C] = d total / d logits, out4 = {total, ce,
 * w_ce, w_geo} (device; d total / d geo = w_geo); row_loss [B] is scratch.  All fp32. */
int basd_ce_uwso(const float* logits, const float* soft_targets, const int64_t* labels, int B, int C,
                 float smoothing, const float* geo, float* row_loss, float* dlogits, float* out4, void* stream);

/* Fused Schedule-Free AdamW step (schedulefree 1.4.1 AdamWScheduleFree, train mode;
 * reference src/training/trainer.py:54-58,158-159) over one flat fp32 buffer of n params:
 *   v = b2 v + (1-b2) g^2 ; gn = g / (sqrt(v / bias_correction2) + eps) + wd * y
 *   y = lerp(y, z, ckp1) + lr (b1 (1 - ckp1) - 1) gn ; z = z - lr gn
 * ckp1 / bias_correction2 / lr are the per-step scalars computed on the host. */
int basd_sf_adamw_step(float* y, const float* g, float* z, float* v, int64_t n, double lr,
                       double beta1, double beta2, double eps, double weight_decay, double ckp1,
                       double bias_correction2, void* stream);

/* y <- y + w (z - y): optimizer.train() / optimizer.eval() switch (trainer.py:180,184). */
int basd_lerp(float* y, const float* z, int64_t n, float w, void* stream);

/* Transposed bf16 images of n_entries weight matrices of the flat fp32 master buffer in one launch (the operands of
 * the input-gradient GEMMs dX = dY W of trainer.py:157; refreshed once per step after the optimizer update).
 * table: HOST array of n_entries x {src offset, dst offset, rows, cols} (int64, offsets in elements):
 * out[dst + c * rows + r] = bf16(master[src + r * cols + c]). */
int basd_transpose_bf16_table(const float* master, void* out, const int64_t* table, int n_entries, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BASD_HIP_H */
