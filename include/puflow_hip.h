/* puflow_hip.h - C ABI of libpuflow_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the native operators under the reference's discrete PU-Flow path.
 * Conventions (shaped like the reference's only native precedent, metric/emd/emd.cpp:14-31):
 *   - every pointer is a DEVICE pointer owned by the caller (contiguous, fp32 / int32);
 *   - the callee allocates nothing, keeps no global state and never synchronises: one call =
 *     stream-ordered enqueue on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return value: PF_OK (0) or a negative PF_ERR_* code; never throws, never prints;
 *   - scratch comes from the caller (`*_workspace_bytes` queries).
 * No torch types appear here; the Python side (puflow_amd/_lib.py) binds these with ctypes.
 */
#ifndef PUFLOW_HIP_H
#define PUFLOW_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define PF_OK 0
#define PF_ERR_NULL (-1)        /* a required pointer is NULL                         */
#define PF_ERR_SHAPE (-2)       /* shape precondition violated                        */
#define PF_ERR_UNSUPPORTED (-3) /* configuration not built (K, unit id, ...)          */
#define PF_ERR_LAUNCH (-4)      /* hipGetLastError() reported a launch failure        */
#define PF_ERR_WORKSPACE (-5)   /* caller workspace too small                         */

int pf_version(void);
const char* pf_error_string(int code);

/* Brute-force kNN. Replaces pytorch3d.ops.knn_points at modules/discrete/interpflow.py:104,:328
 * (and knn_cuda.KNN at modules/utils/patch.py:33,107 for K <= 32).
 * p1 [B,N,3] queries, p2 [B,M,3] references -> idx_out [B,N,K] int32 (index into p2's M),
 * dist_out [B,N,K] squared L2 (nullable).  Order: (distance asc, index asc); distance is the
 * unfused fp32 ((dx*dx)+(dy*dy))+(dz*dz).  K in {4,8,16,32}, K <= M. */
int pf_knn(const float* p1, const float* p2, int B, int N, int M, int K, int* idx_out, float* dist_out, void* stream);

/* Nearest neighbour (K=1): dist_out [B,N] squared L2, idx_out [B,N] (nullable); first minimum
 * wins ties.  Building block of chamfer (pytorch3d.loss.chamfer_distance, metric/loss.py:42;
 * kaolin chamfer, metric/loss.py:35; ChamferDistancePytorch chamfer_3DDist, modules/utils/patch.py:199-203). */
int pf_nn1(const float* p1, const float* p2, int B, int N, int M, float* dist_out, int* idx_out, void* stream);

/* Fused EdgeConv dense block + max-pool over K=16 neighbours, eval mode.
 * Replaces FeatureExtractUnit.forward (modules/discrete/interpflow.py:190-248).
 * cfg 7: units 2..5, cfg 8 / 9: units 0 / 1 - the PRODUCT arithmetic: split-fp16 with a natural-scale low half on
 * v_mfma_f32_16x16x32_f16 (fp32-class results, csrc/pf_mfma.h); wfrag = packing.ec4_weights / ec1n_w (unit 0's edge table
 * rides at the end of its image, input = xyz [B*N,3]); units 1..5 read the P|Q table [B*N, 2S] with the row scales of
 * packing.ec4_scales (csrc/edgeconv.hip edgeconv4_kernel / edgeconv1n_kernel);
 * cfg 0 / 1 / 2: unit 0 (input = xyz, `tab` = [96,8] folded edge table), unit 1, units 2..5 on the exact-fp32 kernel
 * (v_mfma_f32_16x16x4_f32, unscaled P|Q table): the in-library A/B reference the parity tests compare the product with;
 * cfg 3..6 (round-1 split-bf16 / scaled split-fp16 generations) are gone: PF_ERR_UNSUPPORTED.
 * idx [B*N,16] int32 (index inside the batch item); wfrag = fragment-packed growth weights;
 * out [B*N, odim]. */
int pf_edgeconv(int cfg, const float* pq_or_xyz, const float* tab, const int* idx, const float* wfrag, float* out,
                int B, int N, void* stream);

/* Same as pf_edgeconv with an explicit launch shape (points per wave / waves per workgroup).  The default build carries
 * only the shape pf_edgeconv ships per cfg (anything else: PF_ERR_UNSUPPORTED); a -DPF_TUNING_VARIANTS build adds the
 * alternatives and the timing-only ablation instantiations for tools/tune_ec4.py. */
int pf_edgeconv_tuned(int cfg, int variant, const float* pq_or_xyz, const float* tab, const int* idx,
                      const float* wfrag, float* out, int B, int N, void* stream);

/* EdgeConv unit `unit` (0..4, product arithmetic) AND the next unit's P|Q vectors: out [B*N, odim] as pf_edgeconv, pq_next
 * [B*N, rows] as pf_pq_gemm(unit, out, ...) - bit-identical to that pair.  fuse: -1 = one launch when B*N <= 16 384 (the GEMM
 * runs on each 16-point workgroup tile inside the EdgeConv kernel), otherwise the two kernels; 0 = never; 1 = always.
 * w: weight blob base, off[13]: POST_SLOTS offsets of unit `unit`. */
int pf_edgeconv_pq(int unit, const float* pq_or_xyz, const int* idx, const float* wfrag, float* out, const float* w,
                   const long long* off, float* pq_next, int B, int N, int fuse, void* stream);

/* Per-point stages after EdgeConv unit `unit` (0..5).  off[13] = float offsets into the blob `w` of
 * M1,b1,M2,H1,S2,bS2,T2,bT2,ST4,bST4,PQ,bPQ,scales (packing.POST_SLOTS; f16n fragment images + per-matrix 2^-sw).
 *
 * pf_pq_gemm (unit 0..4): next unit's per-point EdgeConv vectors  PQ' = Wpq h + bpq  -> pq_next [T, 2S']
 *   (the exact per-point fold of the edge feature [x_i, x_j, x_j - x_i], interpflow.py:229-232; packing.fold_edgeconv).
 * pf_cond (unit 0..5): FeatMergeUnit (interpflow.py:251-258), the injector conditioner nets (coupling.py:132-134,
 *   interpflow.py:22-43) and coupling1's c-part: c [T,cdim] (nullable), st [T,8], cp [T,64].
 * pf_post = both, one call per unit (pq_next may be NULL for unit 5 only). */
int pf_pq_gemm(int unit, const float* h, const float* w, const long long* off, float* pq_next, int T, void* stream);
int pf_cond(int unit, const float* h, const float* w, const long long* off, float* c, float* st, float* cp, int T,
            void* stream);
int pf_post(int unit, const float* h, const float* w, const long long* off, float* c, float* st, float* cp,
            float* pq_next, int T, void* stream);
/* pf_cond of all six units in ONE launch (the stages only feed the flow kernels, so they run after the EdgeConv chain):
 * h[6] device pointers to the units' EdgeConv outputs (HOST array of device pointers), c[6] likewise (or NULL),
 * st [6][T][8], cp [6][T][64], off[6*13].  st = cp = NULL with c given: the stage stops at the conditioning features (the
 * continuous model needs nothing else of it). */
int pf_cond_all(const float* const* h, const float* w, const long long* off, float* const* c, float* st, float* cp, int T,
                void* stream);

/* Flow forward f: all 6 FlowBlock.forward in one launch.  Replaces PointInterpFlow.f /
 * FlowBlock.forward (modules/discrete/interpflow.py:302-313,:66-74; normalize.py:30-37,
 * permutate.py:117-120, coupling.py:55-58,114-118,127-139).
 * xyz [T,3]; cp [6][T][64] and st [6][T][8] as written by pf_post (unit-major, contiguous);
 * w = 6 flow records of 5360 floats (layout: csrc/flow.hip header, packing.pack_flow).
 * z [T,3] out; ld_pt [T] out = per-point  -sum_blocks sum_ch s. */
int pf_flow_fwd(const float* xyz, const float* cp, const float* st, const float* w, float* z, float* ld_pt, int T,
                void* stream);

/* Flow inverse g (interpflow.py:315-321,:76-82; coupling.py:82-85,141-151; permutate.py:122-126;
 * normalize.py:39-43): u [T*R,3] (row = n*R + r, conditioning row n) -> x [T*R,3]. */
int pf_flow_inv(const float* u, const float* cp, const float* st, const float* w, float* x, int T, int R, void* stream);

/* Deterministic log-likelihood reduction (PointInterpFlow.log_prob interpflow.py:339-345,
 * probs.py:73-75,87-93):  ldj[b] = sum_n ld_pt + N*ld_const;  lpsum[b] = sum -0.5(z^2+log 2pi) + ldj[b];
 * logp[0] = -mean_b lpsum.  ldj, lpsum: [B] outputs; logp: 1 float. */
int pf_logp(const float* z, const float* ld_pt, float ld_const, int B, int N, float* ldj, float* lpsum, float* logp,
            void* stream);

/* pf_flow_fwd + pf_logp in ONE launch (the workgroup that finishes last reduces the wave tiles' sums per batch item, in a
 * fixed order): outputs as those two.  ws: pf_flow_fwd_logp_ws_floats(B, N) floats, ZERO before the first call and left
 * reusable by the kernel; one workspace per stream in flight.  N % 16 != 0 runs the two launches. */
long long pf_flow_fwd_logp_ws_floats(int B, int N);
int pf_flow_fwd_logp(const float* xyz, const float* cp, const float* st, const float* w, float* z, float* ld_pt,
                     float ld_const, int B, int N, float* ldj, float* lpsum, float* logp, float* ws, void* stream);

/* Interpolation module (interpflow.py:85-186), fused: kNN-8 context features -> weights ->
 * softmax_k -> weighted sum of neighbour latents.  idx16 [T,16] (first 8 columns used),
 * u_out [T*R,3] in the row order of g (row = n*R + r).  1 <= R <= 32 (r_max of WeightEstimationUnit,
 * interpflow.py:142; R <= 4 runs the 4-row fast path, larger ratios all 32 rows of the last weight conv).
 * off[15]: float offsets into w (csrc/interp.hip header, packing.INTERP_SLOTS); the blob does not depend on R. */
int pf_interp(const float* xyz, const float* z, const int* idx16, const float* w, const long long* off, float* u_out,
              int B, int N, int R, void* stream);

/* The same module split so that its heavy part does not wait for the latents: pf_interp_weights writes the softmax weights
 * aw [B*N][8][4] (a function of xyz and the neighbour lists only: it can run beside the feature extractor / flow f chain, e.g.
 * as a parallel branch of a captured graph), pf_flow_inv_interp forms u[n R + r] = sum_k aw[n][k][r] z[idx8[n][k]] inside the
 * flow-g kernel.  R <= 4.  The pair returns the bits of pf_interp followed by pf_flow_inv. */
int pf_interp_weights(const float* xyz, const int* idx16, const float* w, const long long* off, float* aw_out, int B, int N,
                      void* stream);
int pf_flow_inv_interp(const float* aw, const float* z, const int* idx16, const float* cp, const float* st, const float* w,
                       float* x, int B, int N, int R, void* stream);

/* Chamfer forward: dist1/idx1 [B,N] (x -> y), dist2/idx2 [B,M] (y -> x), squared L2, first-minimum ties.
 * per_sample [B] = mean_n dist1 + mean_m dist2 (nullable); mean_sum [2] = {mean_b, sum_b} of per_sample
 * (nullable).  Replaces pytorch3d chamfer_distance (metric/loss.py:42), kaolin chamfer (metric/loss.py:35)
 * and chamfer_3DDist (modules/utils/patch.py:199-203). */
int pf_chamfer_fwd(const float* x, const float* y, int B, int N, int M, float* dist1, int* idx1, float* dist2,
                   int* idx2, float* per_sample, float* mean_sum, void* stream);

/* Chamfer backward: g1 [B,N] = dL/d dist1, g2 [B,M] = dL/d dist2; ACCUMULATES into gx [B,N,3], gy [B,M,3]
 * (caller zero-fills).  d dist/d x_i = 2 (x_i - y_j), d dist/d y_j = -2 (x_i - y_j). */
int pf_chamfer_bwd(const float* x, const float* y, const int* idx1, const int* idx2, const float* g1, const float* g2,
                   float* gx, float* gy, int B, int N, int M, void* stream);
/* the same without float atomics (every point's incoming terms found by a scan of the other cloud's nearest-neighbour map and
 * added in index order): bit-reproducible, O(N M) per sample - the training step's debugging switch (PF_TRAIN_DETERMINISTIC) */
int pf_chamfer_bwd_det(const float* x, const float* y, const int* idx1, const int* idx2, const float* g1, const float* g2,
                       float* gx, float* gy, int B, int N, int M, void* stream);

/* Auction EMD forward.  Replaces emd.forward (metric/emd/emd.cpp:14-19 -> emd_cuda.cu:228-282).
 * Same ownership rule as the reference: the caller allocates outputs AND scratch (emd_module.py:43-56):
 * xyz1 (prediction), xyz2 (ground truth) [B,n,3]; dist [B,n] out; assignment [B,n] in/out (-1 = free);
 * price [B,n] in/out (0); assignment_inv [B,n] in/out (-1); bid, bid_increments, max_increments,
 * unass_idx, max_idx: [B,n] scratch.  The reference's unass_cnt / unass_cnt_sum / cnt_tmp (512-entry
 * batch prefix sums for its multi-kernel pipeline) have no counterpart: one workgroup owns one sample.
 * No n % 1024 or B <= 512 restriction.  Returns PF_OK / PF_ERR_* (the reference returns 1 / 0 / -1). */
int pf_emd_forward(const float* xyz1, const float* xyz2, float* dist, int* assignment, float* price,
                   int* assignment_inv, int* bid, float* bid_increments, float* max_increments, int* unass_idx,
                   int* max_idx, float eps, int iters, int B, int n, void* stream);

/* pf_emd_forward with the launch decision and the failure report made explicit.
 * groups: workgroups per sample.  0 = chosen here from the device: the multi-workgroup auction waits on grid barriers, so it
 * runs only when hipOccupancyMaxActiveBlocksPerMultiprocessor(kernel) x CU count says all B*G workgroups are resident at once
 * (G = largest power of two <= 16 that fits, >= 64 points per workgroup), otherwise one workgroup per sample; 1 = always one
 * workgroup per sample (no inter-workgroup waits: the setting for a GPU shared between processes); g > 1 = at most g.
 * status: nullable device word; += 1 per WORKGROUP whose barrier timed out (its slice of dist is NaN): any non-zero value is
 * a failure.  The caller reads it
 * at its next synchronisation point (puflow_amd.loss.check_emd_status raises). */
int pf_emd_forward_ex(const float* xyz1, const float* xyz2, float* dist, int* assignment, float* price,
                      int* assignment_inv, int* bid, float* bid_increments, float* max_increments, int* unass_idx,
                      int* max_idx, float eps, int iters, int B, int n, int groups, unsigned* status, void* stream);

/* Auction EMD backward.  Replaces emd.backward (emd.cpp:21-24 -> emd_cuda.cu:284-316):
 * gradxyz [B,n,3] += 2 graddist (xyz1 - xyz2[idx]); gradient wrt xyz2 is zero (emd_module.py:68-72). */
int pf_emd_backward(const float* xyz1, const float* xyz2, float* gradxyz, const float* graddist, const int* idx, int B,
                    int n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training-step building blocks (csrc/train_ops.hip): one kernel pair per eager op the reference's
 * train-mode forward/backward runs (modules/discrete/interpflow.py:203-258, train_pu1k.py:53-74),
 * on channels-last [rows, C] fp32 tensors.  Wired into autograd by puflow_amd/train_ops.py.
 * ------------------------------------------------------------------------------------------- */

/* C[M,N] = A(M,K) B(K,N) (+ bias[N]), generic element strides: A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn].
 * Serves every Conv2d(1x1) / nn.Linear forward, dX and dW.  ws: split-K slabs, pf_gemm_ws_floats(M,N,K) floats. */
long long pf_gemm_ws_floats(int M, int N, int K);
int pf_gemm(const float* A, long long sam, long long sak, const float* B, long long sbk, long long sbn, float* C,
            long long ldc, const float* bias, int M, int N, int K, float* ws, long long ws_floats, void* stream);
/* the split-K reduction of pf_gemm for callers that wrote their own slabs [nslab][M * N]: C [M, ldc] = their sum, fixed order */
int pf_gemm_reduce(const float* slabs, float* C, int M, int N, long long ldc, int nslab, void* stream);
/* pf_gemm with the matrix-pipe arithmetic chosen by the caller: 0 = f32 MFMA (what pf_gemm runs), 1 = the same products in the
 * same order on the round-1 kernel (A/B reference: bit-identical to 0), 2 = split-fp16 (operands
 * inside the fp16 range: forward GEMMs), 3 = split-bf16 (gradient operands); all fp32-class results. */
int pf_gemm_ex(int arith, const float* A, long long sam, long long sak, const float* B, long long sbk, long long sbn, float* C,
               long long ldc, const float* bias, int M, int N, int K, float* ws, long long ws_floats, void* stream);

/* BatchNorm2d(training) + LeakyReLU(slope) on x [R,C] (interpflow.py:205-206,216-217; eps 1e-5, momentum 0.1):
 * save [2][C] = batch mean, 1/sqrt(var+eps); running stats updated in place when non-NULL (unbiased variance).
 * ws: (2*pf_bn_chunks(R) + 2) * C floats. */
int pf_bn_chunks(long long R);
int pf_bn_lrelu_fwd(const float* x, long long R, int C, const float* gamma, const float* beta, float slope, float eps,
                    float momentum, float* run_mean, float* run_var, float* y, float* save, float* ws, void* stream);
int pf_bn_lrelu_bwd(const float* x, const float* dy, long long R, int C, const float* gamma, const float* beta, float slope,
                    const float* save, float* dx, float* dgamma, float* dbeta, float* ws, void* stream);
/* The same BatchNorm + LeakyReLU in stages, so that per-column sums can be all-reduced between them (SyncBN across
 * ranks; SURVEY 8e "decision to document" / f-3).  Sums are unscaled; the host divides by the global row count.
 *   pf_bn_colstat:     out[c] = sum_r x (mean_in NULL) or sum_r (x - mean_in)^2;   ws: 2*pf_bn_chunks(R)*C floats
 *   pf_bn_apply_stats: y from GIVEN mean / biased variance; save = [mean | invstd]; running stats updated (nullable)
 *   pf_bn_bwd_sums:    sums[2][C] = (sum dz, sum dz*xhat) of this rank's rows (= dbeta, dgamma of the local loss)
 *   pf_bn_bwd_apply:   dx from the global means[2][C] = all-reduced sums / global row count */
int pf_bn_colstat(const float* x, long long R, int C, const float* mean_in, float* out, float* ws, void* stream);
int pf_bn_apply_stats(const float* x, long long R, int C, const float* mean, const float* var_b, float unbias,
                      const float* gamma, const float* beta, float slope, float eps, float momentum, float* run_mean,
                      float* run_var, float* y, float* save, void* stream);
int pf_bn_bwd_sums(const float* x, const float* dy, long long R, int C, const float* gamma, const float* beta, float slope,
                   const float* save, float* sums, float* ws, void* stream);
int pf_bn_bwd_apply(const float* x, const float* dy, long long R, int C, const float* gamma, const float* beta, float slope,
                    const float* save, const float* means, float* dx, void* stream);
/* column sums of g [R,C] -> out [C] (bias gradients); ws: 2*pf_bn_chunks(R)*C floats */
int pf_colsum(const float* g, long long R, int C, float* out, float* ws, void* stream);

/* y = x > 0 ? x : slope*x (LeakyReLU; slope 0 = ReLU) and its backward from the OUTPUT sign */
int pf_act_fwd(const float* x, float slope, long long total, float* y, void* stream);
int pf_act_bwd(const float* y, const float* dy, float slope, long long total, float* dx, void* stream);

/* EdgeConv edge feature (interpflow.py:223-232): out [B*N*K, 3C] = [x_i, x_j, x_j - x_i]; backward ACCUMULATES into dx */
int pf_edge_feature_fwd(const float* x, const int* idx, int B, int N, int K, int C, float* out, void* stream);
int pf_edge_feature_bwd(const float* g, const int* idx, int B, int N, int K, int C, float* dx, void* stream);

/* max over the K neighbours (interpflow.py:245): y [T,K,C] -> out [T,C], arg [T,C]; backward dx [T,K,C] */
int pf_maxpool_k_fwd(const float* y, long long T, int K, int C, float* out, int* arg, void* stream);
int pf_maxpool_k_bwd(const float* dy, const int* arg, long long T, int K, int C, float* dx, void* stream);

/* backward of a neighbour-row gather x[idx] (interpflow.py:183): out [B*N,C] += g [B*N*K,C] (out zero-filled by caller) */
int pf_scatter_rows(const float* g, const int* idx, int B, int N, int K, int C, float* out, void* stream);
/* the same as a gather over the sorted transposed lists of idx (pf_knn_csr): out [T, C] = sum over the edges that point at a row,
 * in list order - no float atomics (PF_TRAIN_DETERMINISTIC) */
int pf_scatter_rows_det(const float* g, const int* csr_off, const int* csr_edge, long long T, int C, float* out, void* stream);
/* backward of repeat_interleave(c, R, dim=1) (interpflow.py:319): out [T,C] = sum_r g[T*R,C] */
int pf_group_sum(const float* g, long long T, int R, int C, float* out, void* stream);

/* interpolation tail (interpflow.py:180-185): softmax over K=8 of the first R channels of w [T,K,ldw], weighted sum of
 * the gathered latents zj [T,K,3] -> a [T,K,R], fz [T,3,R]; backward -> dw [T,K,ldw], dzj [T,K,3] */
int pf_softmax_wsum_fwd(const float* w, int ldw, const float* zj, int K, int R, long long T, float* a, float* fz,
                        void* stream);
int pf_softmax_wsum_bwd(const float* a, const float* zj, const float* dfz, int K, int R, int ldw, long long T, float* dw,
                        float* dzj, void* stream);

/* Flow-block elementwise algebra on [R,3] rows for the training path (csrc/train_ops.hip, second part), forward +
 * backward: ActNorm (normalize.py:30-43; inv = 1: (x-bias) exp(-logs)); additive coupling + channel reverse + injector
 * (coupling.py:55-58,114-118,132-137; permutate.py:75-80) and their inverse halves (coupling.py:82-85,141-151);
 * per-batch sums for the log-det and the Gaussian log-likelihood (probs.py:73-75,87-93).  td = width of the first
 * split (1 or 2).  Parameter-gradient rows (glogs_rows, gbias_rows [R,3]) are column-summed with pf_colsum. */
int pf_actnorm_fwd(const float* x, const float* logs, const float* bias, int inv, long long R, float* y, void* stream);
int pf_actnorm_bwd(const float* x, const float* dy, const float* logs, const float* bias, int inv, long long R, float* dx,
                   float* glogs_rows, float* gbias_rows, void* stream);
int pf_couple_inject_fwd(const float* y, const float* o, const float* s, const float* t, int td, long long R, float* out,
                         void* stream);
int pf_couple_inject_bwd(const float* out, const float* dout, const float* s, int td, long long R, float* dy, float* do_,
                         float* ds, float* dt, void* stream);
int pf_inject_inv_fwd(const float* u, const float* s, const float* t, long long R, float* v, void* stream);
int pf_inject_inv_bwd(const float* u, const float* s, const float* dv, long long R, float* du, float* ds, float* dt,
                      void* stream);
int pf_couple_add(const float* v, const float* o, int td, long long R, float* out, void* stream);
int pf_slice_tail(const float* g, int td, long long R, float* o, void* stream);
/* out[b] = sum_m f(x[b*M+m]); mode 0: f = x, mode 1: f = -0.5 (x^2 + log 2 pi); backward dx = g[b] f'(x) */
int pf_batch_sum_fwd(const float* x, int B, long long M, int mode, float* out, void* stream);
int pf_batch_sum_bwd(const float* x, const float* g, int B, long long M, int mode, float* dx, void* stream);
/* DistanceEncoder.distance_vec (interpflow.py:100-115): out [B*N*K,10] = [x_i, x_j, x_i - x_j, |x_i - x_j|] */
int pf_dist_feature(const float* xyz, const int* idx, int B, int N, int K, float* out, void* stream);

/* ---- one FeatureExtractUnit (EdgeConv dense block) of the training step, fused (csrc/train_fused.hip) ----
 * Replaces FeatureExtractUnit.forward in train() mode and its autograd backward (modules/discrete/interpflow.py:190-248):
 * edge feature -> [Conv2d 1x1 + BatchNorm2d(batch statistics) + LeakyReLU, dense concatenation] x nconv -> conv_out ->
 * max over the K neighbours (pooling = 1) or the per-edge output (pooling = 0, the interpolation's feat_conv).
 * T = B*N points, E = T*K edges (point-major rows), GT = growth*nconv, S = GT + odim.  growth in {8,16,32}, nconv <= 8,
 * odim a multiple of 16 <= 128, GT in {32, 64, 128}, E a multiple of 16, pooling requires K == 16.
 * The caller owns every buffer; those marked (kept) must survive from pf_ec_train_fwd to pf_ec_train_bwd. */
#define PF_TRAIN_STAT_DOUBLES 4097
typedef struct PfEcTrain {
    int B, N, K, C, growth, nconv, odim, pooling;
    float slope, eps, momentum;
    const float* x;                 /* [T, C] */
    const int* idx;                 /* [T, K] batch-local neighbour indices */
    const float* W[9];              /* conv t [growth, 3C + growth t] (t < nconv), then conv_out [odim, 3C + GT] */
    const float* bias[9];
    const float* gamma[8]; const float* beta[8];
    float* run_mean[8]; float* run_var[8];     /* nullable: running statistics, updated in place */
    float* Wpq;                     /* (kept) [2S, C] folded edge-feature weights */
    float* bpq;                     /* [2S] */
    float* PQ;                      /* (kept) [T, 2S] */
    float* Y;                       /* (kept) [E, GT] pre-BatchNorm outputs of the growth layers */
    float* aff;                     /* (kept) [4][GT] scale, shift, batch mean, 1/std */
    float* out;                     /* [T, odim] (pooling) or [E, odim] */
    unsigned char* arg;             /* (kept) [T, odim] argmax over K (pooling) */
    /* backward only */
    const float* dout;              /* gradient of `out` */
    float* dA;                      /* [E, GT] scratch */
    float* dPQ;                     /* [T, 2S] scratch */
    float* coef;                    /* [2][GT] scratch */
    float* dWpq;                    /* [2S, C] scratch */
    float* dx;                      /* [T, C], nullable */
    float* dW[9]; float* dbias[9]; float* dgamma[8]; float* dbeta[8];
    float* ws; long long ws_floats; /* >= pf_ec_train_ws_floats() */
    double* stat;                   /* PF_TRAIN_STAT_DOUBLES doubles (column-statistics accumulators): zeroed ONCE by the caller,
                                       every kernel that uses them leaves them zero again */
    const int* csr_off; const int* csr_edge;   /* backward, nullable: transposed neighbour lists (pf_knn_csr) - the neighbour
                                       scatter-add of dQ then runs as a gather without float atomics */
    int flags;                      /* PF_EC_PERSISTENT: the forward may run as ONE persistent launch with a grid barrier per
                                       BatchNorm layer (pooling units, K = 16, nconv = 4, every tile resident at once - decided by
                                       the library from the device's occupancy).  The caller sets it only when no other barrier
                                       kernel of the process can run beside this call (another stream, another captured graph in
                                       flight, another process on the device): two such kernels can starve each other */
    unsigned* sync;                 /* flags != 0: 4 words, zeroed ONCE by the caller; [0..2] are left zero by every launch, [3] is
                                       sticky: 1 = a grid barrier timed out (the unit's output is NaN) */
    /* SyncBN (BatchNorm statistics over all ranks, interpflow.py:93,96,205,216 with global-batch semantics): sync_sums != NULL
     * makes every BatchNorm layer leave its LOCAL column sums + row count in sync_sums (2 * 128 + 1 doubles) instead of
     * finishing; the library then calls sync_cb(sync_user, sync_sums, 257, stream) - which must all-reduce (SUM) those doubles
     * over the ranks, stream-ordered, and return 0 - and finishes the layer with the global sums in one small launch.  dgamma /
     * dbeta stay the local sums (as torch.nn.SyncBatchNorm: the gradient bucket's mean over ranks follows).  The persistent
     * launches are not used in this mode. */
    int (*sync_cb)(void* user, double* sums, int n, void* stream);
    void* sync_user;
    double* sync_sums;
    float* ws_dw; long long ws_dw_floats; /* backward with pf_train_set_dw_stream: a second workspace (>= pf_ec_train_ws_floats()) that
                                     * only the weight-gradient stream's kernels use; NULL = weight gradients on the calling stream */
    const float* dx_add;            /* backward, nullable: [B*N, C] added to dx in the epilogue of its GEMM - the part of x's gradient
                                     * that x's OTHER consumer (the unit's FeatMergeUnit, interpflow.py:251-258) has produced already,
                                     * instead of a separate add launch afterwards */
} PfEcTrain;
#define PF_EC_PERSISTENT 1
/* flags bit (PfEcTrain, PfBnMlpTrain): BatchNorm's batch statistics are accumulated as 64-bit fixed-point sums (quantum 2^-28)
 * with integer atomics instead of double atomics - exact, hence independent of the order in which workgroups arrive: two runs
 * of the same step give the same bits (the default's double sums round in arrival order once they need more than 53 bits, and
 * a 1e-7 difference in x flips discrete auction assignments and max-pool routes).  Values differ from the default's by the
 * quantisation (~4e-9 absolute per workgroup partial).  The persistent kernels are not used under it.  Debugging switch:
 * puflow_amd sets it from `net.deterministic` / `cfg.deterministic`. */
#define PF_TRAIN_DETERMINISTIC 2
/* flags bit (PfEcTrain): Wpq / bpq were filled by pf_ec_train_fold_batch since the parameters last changed - pf_ec_train_fwd
 * skips its own fold launch */
#define PF_EC_PREFOLDED 4
/* transposed neighbour lists of idx [B*N, K]: off [T+1], edge [T*K]; cnt: T ints (4-aligned size) of scratch */
/* Weight gradients beside the backward chain.  With a stream set here (per host thread; NULL = off, the default) the backward
 * entry points pf_ec_train_bwd, pf_mlp_train_bwd, pf_mlp_train_bwd_batch and pf_mlp_train_dw_batch enqueue their split-K
 * weight-gradient kernels and reductions on it, ordered by an event behind what their own stream holds at that point, and
 * return without waiting: the input gradient stays on the calling stream.  The caller (1) hands those calls workspaces that
 * nothing on another stream touches (PfEcTrain.ws_dw; PfMlpTrain.ws is weight-gradient scratch only), (2) keeps every buffer
 * of the call alive until (3) it has made the consumer of the weight gradients wait for the stream.  Inside a hipGraph
 * capture the stream must belong to the capture (it joins it through the event) and is a parallel branch of the graph. */
int pf_train_set_dw_stream(void* stream);
int pf_knn_csr(const int* idx, int B, int N, int K, int* off, int* edge, int* cnt, void* stream);
/* pf_knn_csr for idx [B*N, K] and for its first K2 <= K columns (as if those were stored contiguously: edge ids i K2 + k) from one
 * pass: off [T+1], edge [T*K], off2 [T+1], edge2 [T*K2]; cnt: 2 x ((T + 3) / 4 * 4) ints of scratch.  The training step uses both
 * (16 neighbours: feature units, 8: the interpolation unit, interpflow.py:85-151,300-306). */
int pf_knn_csr_pair(const int* idx, int B, int N, int K, int K2, int* off, int* edge, int* off2, int* edge2, int* cnt, void* stream);

/* sorts every list of pf_knn_csr (edge ids ascending; T = B*N lists): sums over a list then add in one order, run after run -
 * what PF_TRAIN_DETERMINISTIC's gather-form gradients need; the default mode does not call it */
int pf_knn_csr_sort(const int* off, int* edge, int T, void* stream);
long long pf_ec_train_ws_floats(const PfEcTrain* p);
int pf_ec_train_fwd(const PfEcTrain* p, void* stream);
/* The folded edge-feature weights Wpq / bpq of n <= 8 units (only C, growth, nconv, odim, B, N, K, W, bias, Wpq, bpq of each
 * descriptor are read) in ONE launch: they depend on parameters only (interpflow.py:190-248 applies the convolutions to
 * cat[x_i, x_j, x_j - x_i]; W_p = W_a - W_c, W_q = W_b + W_c), so a step folds all units before the first one runs. */
int pf_ec_train_fold_batch(const PfEcTrain* descs, int n, void* stream);
int pf_ec_train_bwd(const PfEcTrain* p, void* stream);

/* ---- BatchNorm MLP of the interpolation module in the training step, fused (csrc/train_fused.hip) ----
 * Replaces DistanceEncoder.mlp / WeightEstimationUnit.mlp (modules/discrete/interpflow.py:85-151: [Conv2d 1x1 +
 * BatchNorm2d(batch statistics) + LeakyReLU(slope)] x (nl - 1), then Conv2d 1x1) in train() mode with their autograd
 * backward.  Input = cat[xa [rows, kin0a], xb [rows, kin0b]] (never built; kin0b = 0: one input); widths multiples of 16
 * <= 128, kin0a, kin0b <= 128.  W[l]: [width[l], in_l], in_0 = kin0a + kin0b. */
typedef struct PfBnMlpTrain {
    int rows, nl, kin0a, kin0b;
    int width[3];
    float slope, eps, momentum;
    const float* xa; const float* xb;
    const float* W[3]; const float* b[3];
    const float* gamma[2]; const float* beta[2];
    float* run_mean[2]; float* run_var[2];      /* nullable */
    float* y[3];                    /* (kept) [rows, width[l]] pre-BatchNorm output of layer l; y[nl-1] is the result */
    float* aff[2];                  /* (kept) [4][width[l]] scale, shift, batch mean, 1/std */
    /* backward only */
    const float* dout;              /* [rows, width[nl-1]] */
    float* d[2];                    /* [rows, width[l]] scratch */
    float* coef[2];                 /* [2][width[l]] scratch */
    float* dxa; float* dxb;         /* nullable: [rows, kin0a], [rows, kin0b] */
    float* dW[3]; float* db[3]; float* dgamma[2]; float* dbeta[2];
    float* ws; long long ws_floats; /* >= pf_bnmlp_train_ws_floats() */
    double* stat;                   /* PF_TRAIN_STAT_DOUBLES doubles, zeroed once by the caller (see PfEcTrain) */
    int (*sync_cb)(void* user, double* sums, int n, void* stream);      /* SyncBN, as in PfEcTrain */
    void* sync_user;
    double* sync_sums;
    int flags;                      /* PF_TRAIN_DETERMINISTIC, PF_BNMLP_SUM_INPUTS */
} PfBnMlpTrain;
/* PfBnMlpTrain.flags: layer 0 is no product - its pre-BatchNorm output is xa + xb (kin0a = kin0b = width[0]; W[0], b[0], dW[0],
 * db[0], dxa, dxb unused: the gradient of BOTH inputs is d[0] after the backward call).  For a first layer whose weights were
 * folded into the last linear layers of the two producers (WeightEstimationUnit's conv 0 over cat[DistanceEncoder, EdgeConv],
 * modules/discrete/interpflow.py:134,144-146: no nonlinearity between them) - the same function, 2 x 128 x 128 MACs per row less. */
#define PF_BNMLP_SUM_INPUTS 4
long long pf_bnmlp_train_ws_floats(const PfBnMlpTrain* p);
int pf_bnmlp_train_fwd(const PfBnMlpTrain* p, void* stream);
int pf_bnmlp_train_bwd(const PfBnMlpTrain* p, void* stream);

/* ---- point-wise MLP of the training step (2 or 3 Linear layers, LeakyReLU / ReLU between them), fused (csrc/train_mlp.hip) ----
 * Replaces LinearA1D (modules/discrete/interpflow.py:22-43, the conditioner of the coupling / injector layers) and
 * FeatMergeUnit (interpflow.py:251-258) in train() mode together with their autograd backward.
 * Input of layer 0 = cat[y[row, :td], c[row / cdiv, :cc]] (never materialised): td <= 3, cc a multiple of 16 <= 128,
 * cdiv in {1,2,4,8,16} divides rows; hidden widths multiples of 16 <= 128; last width <= 128.
 * W[l]: [width[l], in_l] row-major, in_0 = td + cc, in_l = width[l-1]; b[l] nullable. */
typedef struct PfMlpTrain {
    int rows, nl, td, ldy, cc, cdiv;
    int width[3];
    float slope[2];                 /* activation after layer 0 (and 1): max(x, slope x) */
    const float* y;                 /* [rows, ldy], nullable when td == 0 */
    const float* c;                 /* [rows / cdiv, cc] */
    const float* W[3]; const float* b[3];
    float* h[2];                    /* (kept) [rows, width[l]] activations after layer l < nl - 1 */
    float* out;                     /* [rows, width[nl-1]] */
    /* backward only */
    const float* dout;
    float* dz[2];                   /* [rows, width[l]] scratch */
    float* dy;                      /* [rows, ldy] (columns >= td zeroed), nullable */
    float* dc;                      /* [rows / cdiv, cc], nullable */
    float* dW[3]; float* db[3];     /* db[l] nullable */
    float* ws; long long ws_floats; /* >= pf_mlp_train_ws_floats() */
    int chunk;                      /* rows per split-K chunk of the weight-gradient launch: 0 = default, else a multiple of 32 */
    int flags;                      /* PF_MLP_DW_DZSUM (pf_mlp_train_dw_batch only; 0 everywhere else) */
} PfMlpTrain;
/* PfMlpTrain.flags, pf_mlp_train_dw_batch with cdiv > 1 and cc a multiple of 16: `dc` is an INPUT - [rows / cdiv, width[0]], the sum
 * of dz[0] over the cdiv consecutive rows that share a conditioning row (pf_flowchain_bwd writes it: PfFlowChain.dz1s).  The
 * conditioning columns of layer 0's weight gradient are then accumulated over the rows / cdiv summed rows,
 *   sum_rows dz0[row]^T c[row / cdiv] = sum_points (sum_r dz0[point cdiv + r])^T c[point],
 * and only the td <= 3 coordinate columns over all rows: a third of the layer's matrix work at cdiv = 4, same value up to the
 * order of the additions. */
#define PF_MLP_DW_DZSUM 1
long long pf_mlp_train_ws_floats(const PfMlpTrain* p);
int pf_mlp_train_fwd(const PfMlpTrain* p, void* stream);
int pf_mlp_train_bwd(const PfMlpTrain* p, void* stream);
/* n <= 16 networks of the same depth in one launch per kernel (blockIdx.z = network), e.g. the scale / shift conditioners of
 * all flow blocks, which depend only on the conditioning features.  dev_descs: n * sizeof(PfMlpTrain) bytes of device
 * scratch. */
int pf_mlp_train_fwd_batch(const PfMlpTrain* descs, int n, void* dev_descs, void* stream);
int pf_mlp_train_bwd_batch(const PfMlpTrain* descs, int n, void* dev_descs, void* stream);

/* Weight gradients only (split-K launch + reduction) of n <= 16 networks whose dz / h / dout are already in memory
 * (written by pf_flowchain_bwd); same descriptors as pf_mlp_train_bwd_batch. */
int pf_mlp_train_dw_batch(const PfMlpTrain* descs, int n, void* dev_descs, void* stream);

/* ---- all flow blocks of one direction of the training step: two launches forward, four backward (csrc/train_flowchain.hip) ----
 * Replaces PointInterpFlow.f / .g over FlowBlock.forward / .inverse (modules/discrete/interpflow.py:46-82, 302-321) in train()
 * mode: ActNorm (normalize.py:28-54), the invertible 3x3 linear (permutate.py:117-124), the additive coupling with its LinearA1D
 * conditioner on cat[coords[:td], c] (coupling.py:55-58,114-118; interpflow.py:22-43; hidden width 64, 32 / 64 / 128
 * conditioning channels), the reverse permutation (permutate.py:77-80) and the conditional affine injector with s, t given per ORIGINAL
 * point (coupling.py:120-137).  inv = 0: x [T,3] -> z, ssum[i] = sum(s_i), ld[i] = (sum(logs_i) + log|det W_i|) n_ld;
 * inv = 1: u [T R, 3] -> x through the blocks in reverse order, conditioning rows shared by R in {1,2,4,8,16} consecutive rows.
 * Slabs (`pin`, `mid`, `o`, `h1`, `h2`, `dz1`, `dz2`, `dob`) are indexed by block and kept between forward and backward.
 * Backward: dout [rows,3] (+ dssum, dld [nb], nullable) -> dx (nullable), dc[i] [rows / R, cc[i]], ds[i], dt[i] [rows / R, 3] and
 * every parameter gradient.  part: pf_flowchain_part_floats() floats (both directions); counter: one zero word (left zero); ws:
 * pf_flowchain_ws_floats() floats; dev_descs: nb * sizeof(PfMlpTrain) bytes. */
#define PF_FLOWCHAIN_MAXB 8
typedef struct PfFlowChain {
    int nb, rows, R, inv;
    int td[PF_FLOWCHAIN_MAXB];       /* untouched leading coordinates of block i's coupling (1 or 2) */
    int cc[PF_FLOWCHAIN_MAXB];       /* conditioning channels of block i (32, 64 or 128) */
    float n_ld;                      /* inv = 0: points per batch item (factor of the log-determinant) */
    const float* x;                  /* [rows, 3] */
    const float* c[PF_FLOWCHAIN_MAXB];                                       /* [rows / R, cc[i]] */
    const float* s[PF_FLOWCHAIN_MAXB]; const float* t[PF_FLOWCHAIN_MAXB];    /* [rows / R, 3] */
    const float* logs[PF_FLOWCHAIN_MAXB]; const float* bias[PF_FLOWCHAIN_MAXB]; const float* W[PF_FLOWCHAIN_MAXB];
    const float* w0[PF_FLOWCHAIN_MAXB];                                      /* [64, td + cc] (no bias) */
    const float* w2[PF_FLOWCHAIN_MAXB]; const float* b2[PF_FLOWCHAIN_MAXB];  /* [64, 64], [64] */
    const float* w4[PF_FLOWCHAIN_MAXB]; const float* b4[PF_FLOWCHAIN_MAXB];  /* [3 - td, 64], [3 - td] */
    float* pin; float* mid;          /* [nb][rows, 3]: block input; y (inv = 0) / v (inv = 1) */
    float* o;                        /* [nb][rows, 2] coupling shift (inv = 1) */
    float* h1; float* h2;            /* [nb][rows, 64] */
    float* out;                      /* [rows, 3] */
    float* ssum; float* ld;          /* [nb] (inv = 0) */
    float* logp; int Bsz;            /* inv = 0, nullable: logp[0] = -(sum_rows log N(z) / Bsz + sum_i ld_i - sum_i ssum_i / Bsz), nb <= 7 */
    float* part; unsigned* counter;
    float* img;                      /* pf_flowchain_img_floats(): packed weights, written by the forward, read by the backward */
    /* backward only */
    const float* dout; const float* dssum; const float* dld;
    const float* dlogp;              /* gradient of logp[0]; replaces dssum / dld when given (dout then nullable) */
    float* dx;
    float* dc[PF_FLOWCHAIN_MAXB]; float* ds[PF_FLOWCHAIN_MAXB]; float* dt[PF_FLOWCHAIN_MAXB];
    float* dz1; float* dz2;          /* [nb][rows, 64] */
    float* dob;                      /* [nb] slabs of rows * 2 floats; slab i holds [rows, 3 - td_i] */
    float* dlogs[PF_FLOWCHAIN_MAXB]; float* dbias[PF_FLOWCHAIN_MAXB]; float* dW[PF_FLOWCHAIN_MAXB];
    float* dw0[PF_FLOWCHAIN_MAXB]; float* dw2[PF_FLOWCHAIN_MAXB]; float* db2[PF_FLOWCHAIN_MAXB];
    float* dw4[PF_FLOWCHAIN_MAXB]; float* db4[PF_FLOWCHAIN_MAXB];
    float* ws; long long ws_floats;
    void* dev_descs;
    float* dz1s;                     /* backward, nullable, R > 1: [nb][rows / R, 64] = dz1 summed over the R rows of a conditioning
                                      * row - with it the weight-gradient launch runs layer 0's conditioning columns over rows / R
                                      * summed rows (PF_MLP_DW_DZSUM) */
    int img_ready;                   /* forward: img already holds the packed weights of these parameters (the other direction's call
                                      * of the same forward packed them: the image does not depend on inv / R) - no pack launch */
} PfFlowChain;
long long pf_flowchain_ws_floats(const PfFlowChain* a);
long long pf_flowchain_part_floats(const PfFlowChain* a);
long long pf_flowchain_img_floats(const PfFlowChain* a);
int pf_flowchain_fwd(const PfFlowChain* a, void* stream);
int pf_flowchain_bwd(const PfFlowChain* a, void* stream);

/* ---- element-wise half of a flow block in the training step, fused per direction (csrc/train_flow.hip) ----
 * Replaces ActNorm / InvertibleConv1x1-style 3x3 linear / AffineCoupling / reverse permutation / AffineInjector of
 * modules/discrete/interpflow.py:46-82 (normalize.py:28-54, permutate.py:77-124, coupling.py:55-137) in train() mode.
 * partial: >= 256 * 15 floats of scratch; counter: one 32-bit word, zero between calls. */
int pf_flow_params_fwd(const float* W, const float* logs, float n, float* Winv, float* ld, void* stream);
int pf_flow_params_bwd(const float* Winv, const float* dWinv, const float* dld, float n, float* dW, float* dlogs, void* stream);
int pf_flow_affine_fwd(const float* x, const float* o, int td, const float* logs, const float* bias, const float* M, int inv,
                       long long R, float* y, void* stream);
int pf_flow_affine_bwd(const float* x, const float* o, int td, const float* logs, const float* bias, const float* M, int inv,
                       long long R, const float* dy, float* dx, float* dobuf, float* dlogs, float* dbias, float* dM,
                       float* partial, unsigned* counter, void* stream);
int pf_couple_inject2_fwd(const float* y, const float* o, const float* s, const float* t, int td, long long R, float* out,
                          float* ssum, float* partial, unsigned* counter, void* stream);
int pf_couple_inject2_bwd(const float* out, const float* dout, const float* dssum, const float* s, int td, long long R,
                          float* dy, float* dobuf, float* ds, float* dt, void* stream);
int pf_inject_inv2_fwd(const float* u, const float* s, const float* t, int Rr, long long R, float* v, void* stream);
int pf_inject_inv2_bwd(const float* u, const float* s, const float* dv, int Rr, long long R, float* du, float* ds, float* dt,
                       void* stream);

/* out = ((p0 + p1) + p2) + ... for n_terms in 2..8 tensors of n floats each (n % 4 == 0, 16-byte aligned): the gradient of a
 * tensor with several consumers in ONE launch instead of autograd's n_terms - 1 pairwise adds (train_ops.FanoutFn).  ptrs: host
 * array of device pointers.  `out` may be one of the operands. */
int pf_sum_n(const float* const* ptrs, int n_terms, float* out, long long n, void* stream);

/* n <= 8 contiguous device regions of 32-bit words (4-byte aligned), dst[j][0 .. words[j]) = src[j][...], in ONE launch: the
 * batch of a captured training step into the tensors its graph reads (train_graph.GraphedTrainStep).  src / dst / words: host
 * arrays. */
int pf_copy_n(const void* const* src, void* const* dst, const long long* words, int n, void* stream);

/* WeightEstimationUnit's first conv folded into its producers' last linear layers (interpflow.py:98, 134, 144-146, 219-221: no
 * nonlinearity between them): W0 = [W0a | W0b] [o, 2 o], W6 [o, k6], Wout [o, ko]  ->  W6f = W0a W6, b6f = W0a b6 + b0,
 * Wof = W0b Wout, bof = W0b bout; _bwd: the chain rule from the gradients of those four back to W0, b0, W6, b6, Wout, bout.
 * Tiny, fixed summation order (bit-reproducible), one launch each. */
int pf_fold_wu_fwd(const float* W0, const float* b0, const float* W6, const float* b6, const float* Wout, const float* bout,
                   int o, int k6, int ko, float* W6f, float* b6f, float* Wof, float* bof, void* stream);
int pf_fold_wu_bwd(const float* W0, const float* W6, const float* b6, const float* Wout, const float* bout, int o, int k6, int ko,
                   const float* dW6f, const float* db6f, const float* dWof, const float* dbof, float* dW0, float* db0,
                   float* dW6, float* db6, float* dWout, float* dbout, void* stream);

/* ---- fused glue of the training step (csrc/train_glue.hip): what were chains of one-element torch launches ----
 * pf_interp_wsum: interpolation of the latent (modules/discrete/interpflow.py:153-186, 312-318): softmax over the K = 8
 *   neighbours of the first R <= 8 weight channels of w [T, 8, ldw] and the weighted sum of the gathered latent rows
 *   z [B N, 3] (idx [T, 8] batch-local), written as the [T R, 3] rows flow g reads; a [T, 8, R] is kept for the backward,
 *   which returns dw [T, 8, ldw] and dz [B N, 3] (zero-filled, then scatter-added with float atomics).
 * pf_emd_init: inputs of the auction (metric/emd/emd_module.py:45-56): price = 0, assignment = assignment_inv = -1.
 * pf_pugan_loss: train_pugan.py:52-67, out[0] = w_logp logp + w_emd sum_b sum_n dist[b,n] / radius[b] + w_cd mean_b per[b],
 *   out[1..3] = the weighted EMD, logp and CD terms (per nullable: the EMD-only mix of train_pu1k.py:62-67); backward: the seeds graddist [B,N] (pf_emd_backward), g1 [B,N], g2 [B,M]
 *   (pf_chamfer_bwd), dlogp [1] and the zero-filled gx [B,N,3], gy [B,M,3] those kernels accumulate into. */
int pf_interp_wsum_fwd(const float* w, int ldw, const float* z, const int* idx, int N, int K, int R, long long T, float* a,
                       float* u, void* stream);
int pf_interp_wsum_bwd(const float* a, const float* z, const int* idx, const float* du, int N, int K, int R, int ldw, long long T,
                       float* dw, float* dz, void* stream);
/* csr_off / csr_edge non-NULL (pf_knn_csr of idx): dz as a gather over the sorted transposed lists instead of float atomics */
int pf_interp_wsum_bwd_det(const float* a, const float* z, const int* idx, const float* du, int N, int K, int R, int ldw, long long T,
                       float* dw, float* dz, const int* csr_off, const int* csr_edge,
                           void* stream);
int pf_emd_init(float* price, int* assign2, long long Bn, void* stream);
int pf_pugan_loss_fwd(const float* logp, const float* dist, const float* radius, const float* per, int B, int n, float w_logp,
                      float w_emd, float w_cd, float* out, void* stream);
int pf_pugan_loss_bwd(const float* g, const float* radius, int B, int N, int M, float w_logp, float w_emd, float w_cd,
                      float* graddist, float* g1, float* g2, float* dlogp, float* gx, float* gy, void* stream);
/* The prediction's gradient of the same loss in two launches instead of four (pf_pugan_loss_bwd + pf_chamfer_bwd + pf_emd_backward:
 * train_pugan.py:52-67, emd_cuda.cu:284-300, metric/loss.py:39-42): g [1] = d loss, x [B,n,3] prediction, y [B,n,3] ground truth,
 * assign [B,n] the auction's assignment, idx1 / idx2 [B,n] Chamfer's nearest neighbours (both NULL: no Chamfer term), radius [B]
 * nullable -> gx [B,n,3] (own terms stored, the second Chamfer direction added with float atomics), dlogp [1] = g w_logp.  No
 * gradient for y. */
int pf_pugan_grad(const float* g, const float* radius, const float* x, const float* y, const int* assign, const int* idx1,
                  const int* idx2, int B, int n, int m, float w_logp, float w_emd, float w_cd, float* gx, float* dlogp, void* stream);

/* ---- gradient clipping (L2 norm over all parameters) + Adam for the whole model in two launches (csrc/optim.hip) ----
 * Replaces torch.nn.utils.clip_grad_norm_ (Lightning gradient_clip_val, train_pu1k.py:149) + torch.optim.Adam.step
 * (train_pu1k.py:46).  flat_g / m / v: [numel] in the chunk table's flat layout; params: device array of parameter addresses;
 * chunks: device int32 [nchunks][4] = (tensor id, offset in the tensor, length, offset in the flat buffers), no chunk crosses
 * a tensor; lr, step: device scalars (step is incremented here, before use); partial: >= nchunks doubles; counter: one zero
 * word (left zero); coef: 4 floats - [0] clip coefficient, [1] gradient norm before clipping, [2] 1 when THIS update was skipped
 * because the norm is not finite (parameters, moments and the step counter untouched), [3] += 1 per skipped update (the caller
 * zeroes it). */
int pf_clip_adam(float* flat_g, float* m, float* v, float* const* params, const int* chunks, int nchunks, const float* lr,
                 float* step, float beta1, float beta2, float eps, float max_norm, double* partial, unsigned* counter,
                 float* coef, void* stream);
/* pf_clip_adam with the gradients where autograd left them: grads = device array of nparams gradient addresses in the chunk
 * table's tensor order (contiguous fp32, one per parameter; point a parameter without gradient at zeros) instead of one flat
 * buffer - no concatenation launch in front of the update.  The clipped gradients are written back through the table. */
int pf_clip_adam_ptrs(float* const* grads, float* m, float* v, float* const* params, const int* chunks, int nchunks, const float* lr,
                      float* step, float beta1, float beta2, float eps, float max_norm, double* partial, unsigned* counter,
                      float* coef, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Patch pipeline around the network (modules/utils/patch.py:35-214), csrc/patch_ops.hip
 * ------------------------------------------------------------------------------------------- */

/* Farthest point sampling.  Replaces pointnet2_ops furthest_point_sample (patch.py:102,156).
 * xyz [B,N,3] -> idx_out [B,npoint] int32; starts at index 0; first maximum wins ties; mind: [B,N] float scratch,
 * 8-byte aligned (running min-distances, or - clouds of >= 8192 points, <= 32 cooperating workgroups per cloud -
 * the candidate exchange ring; overwritten either way). */
int pf_fps(const float* xyz, int B, int N, int npoint, float* mind, int* idx_out, void* stream);
/* the same samples (bit for bit) with a layout hint: group > 0 = every `group` consecutive points are one spatial neighbourhood
 * (PatchHelper.merge_pc, modules/utils/patch.py:142-165: the candidates arrive patch after patch, 256 x (upratio + 1) each).  For
 * group = 768 / 1280 / 1536 a wave of the cooperative kernel then holds exactly one neighbourhood, whose bounding box lets it skip
 * the samples that cannot reach it.  group = 0 or any other value: pf_fps. */
int pf_fps_grouped(const float* xyz, int B, int N, int npoint, int group, float* mind, int* idx_out, void* stream);

/* Layout of pf_fps's scratch when the cooperative kernel runs (return value 1; 0 = single-workgroup kernel, no ring):
 * cloud b's ring starts at 64-bit word b * stride_words of `mind`; word `abort_word` of a ring is the cloud's status after
 * the launch: 0 = every step completed; 1 = its workgroups gave up waiting for each other (bounded spin); 2 = it never
 * finished (or never got its workgroups).  Non-zero = that cloud's idx_out row is invalid.  Word abort_word + 1 holds the
 * number of exchange rounds the cloud took (two-sample kernel: npoint / rounds samples per round; measurement only). */
int pf_fps_scratch_layout(int N, long long* stride_words, long long* abort_word);

/* Measurement aid (bench.py --mode pugan): `rounds` rounds of the cooperative FPS kernel's candidate exchange between G
 * workgroups (2..32) with no points to update - the latency floor of one round of pf_fps's cooperative kernel.  ring: >= 1032
 * 64-bit words of scratch; ring[1024] == 0 afterwards when every round completed.  No reference counterpart. */
int pf_fps_exchange_probe(int G, int rounds, unsigned long long* ring, void* stream);

/* PatchHelper.normalize_pc (modules/utils/patch.py:168-178): centroid = mean over the N points, x - centroid, divided by the
 * largest norm.  x, out [B,N,3] (out may alias x), centroid [B,3], fdist [B].  One workgroup per cloud with a fixed
 * summation order: a cloud's result does not depend on B. */
int pf_normalize_pc(const float* x, int B, int N, float* out, float* centroid, float* fdist, void* stream);

/* Text format of the CLI's output clouds (HOST memory, no GPU work): the bytes np.savetxt(path, cloud, fmt='%.6f') writes
 * (modules/discrete/upsample.py:57) - rows of c "%.6f" values separated by one blank, '\n' after every row.  pts [n,c]
 * float32; out must hold pf_format_xyz_bound(n, c) bytes.  Returns the bytes written or a negative PF_ERR_* code. */
long long pf_format_xyz_bound(long long n, int c);
long long pf_format_xyz(const float* pts, long long n, int c, char* out, long long cap);

/* Reader for the CLI's input clouds (HOST memory): the values np.loadtxt(path, dtype=np.float32) returns (upsample.py:42) for
 * whitespace-separated numeric text ('#' comments, blank lines skipped, double parse then rounded to float32).  text[len]
 * must be 0.  Returns the number of values written (rows x *ncols) or a negative PF_ERR_* code (UNSUPPORTED: a token that is
 * not a number - fall back to numpy; SHAPE: ragged rows; WORKSPACE: out too small). */
long long pf_parse_xyz(const char* text, long long len, float* out, long long cap, int* ncols);

/* K nearest references of every query for large K (patch extraction, K = 256).  Replaces knn_cuda.KNN
 * (patch.py:33,107).  ref [B,N,3], query [B,M,3], K <= N (any N; K <= 8192 when N > 16384: the references are then
 * streamed through LDS in chunks) -> idx_out [B,M,K] int32 ordered by (distance, index); dist_out [B,M,K] squared L2
 * (nullable). */
int pf_knn_large(const float* ref, const float* query, int B, int N, int M, int K, int* idx_out, float* dist_out,
                 void* stream);

/* ---------------------------------------------------------------------------------------------
 * Continuous (CNF) flow blocks (modules/continuous/), csrc/cnf.hip.  State rows are [y0 y1 y2 logp].
 * ------------------------------------------------------------------------------------------- */

/* One right-hand-side evaluation of a block's ODE, fused with the Runge-Kutta stage state:
 *   yi = y0 + h * sum_{j<ncoef} coef[j] * k[j]          (k: [7][rows][4] stage buffer; coef: host array)
 *   kout = sgn * ( f(t, yi), -e^T (df/dy) e )            (ODEfunc.forward odefunc.py:121-148, ODEnet :60-104,
 *                                                          ConcatSquashLinear diffeq_layers.py:72-86,
 *                                                          divergence_approx odefunc.py:9-31)
 * ctx [T,288]: per-point context terms (packing.pack_cnf_block), e [T,3] Hutchinson vector, row -> point = row / R,
 * rec: the block's weight record.  yout (nullable) receives yi.  sgn = -1 integrates backwards in time the way
 * torchdiffeq does (t' = -t, f' = -f). */
int pf_cnf_rhs(const float* y0, const float* k, const float* coef, int ncoef, float h, float t, float sgn,
               const float* ctx, const float* e, const float* rec, float* kout, float* yout, int rows, int R,
               void* stream);

/* One whole Dormand-Prince 5(4) step attempt per launch (the six stage evaluations fused, stage derivatives in
 * registers): y1 = y0 + h sum b_j k_j with k_1 = f0 (FSAL), f1 = k_7, ymid (nullable) = dense-output mid-point,
 * out[0] (double, device) = sum_i (err_i / (atol + rtol max(|y0_i|, |y1_i|)))^2 of the embedded error estimate.
 * t = start of the step in solver time; reverse != 0: torchdiffeq's decreasing-time convention (net time = -t, f negated).
 * ws: >= 1024 doubles.  Arithmetic of torchdiffeq's `_runge_kutta_step` / `_compute_error_ratio` called from cnf.py:97-113. */
int pf_cnf_step(const float* y0, const float* f0, float t, float h, int reverse, const float* ctx, const float* e,
                const float* rec, float* y1, float* f1, float* ymid, float rtol, float atol, int rows, int R, double* ws,
                double* out, void* stream);

/* ctx [T, 288] = c [T, cd] Hc^T + hb: the context's share of every gate / bias pre-activation of a flow block's three
 * ConcatSquash layers (modules/continuous/diffeq_layers.py:72-86), one split-fp16 GEMM (fp32-grade, bound by its 1 152-byte row
 * writes).  hc_image: Hc [288, cd] as the f16n fragment image of packing.pack_cnf_context, inv_scale: the inverse of the image's
 * power-of-two scale.  cd: 32, 64 or 128. */
int pf_cnf_context(const float* c, int cd, const float* hc_image, const float* hb, float inv_scale, float* ctx, int T, void* stream);

/* n_attempts dopri5 step attempts with the step-size controller ON THE DEVICE (no host read between attempts).
 * ctl: 16 doubles - [0] t [1] dt [2] t1 [3] n_tot (elements of the RMS norm) [4] cur (which of ya / yb, fa / fb is current)
 * [5] done [6] accepted [7] rejected [8] nfe [9] status (0 ok, 1 non-finite error norm, 2 dt underflow) [10] reverse.
 * The host initialises ctl and the current buffers, enqueues batches and reads `done` once per batch; `out` [rows,4] holds
 * the state at t1 when done with status 0.  ws: >= 1024 doubles.
 * flags: PF_CNF_SPLIT_GATES - the caller vouches that log2(e) x max|t-column of the three hyper_gate weights| x |t1 - t0| <= 100
 * for this record (packing.cnf_split_ok).  The gates' 2^(gt (t + alpha h) + gc) then factor, without overflow, into a part per
 * point and step and a part per channel and stage: 36 of an evaluation's 135 transcendental instructions less. */
#define PF_CNF_SPLIT_GATES 1
int pf_cnf_steps(double* ctl, float* ya, float* yb, float* fa, float* fb, const float* ctx, const float* e, const float* rec,
                 float* out, float rtol, float atol, int rows, int R, int n_attempts, double* ws, int flags, void* stream);

/* The start of that integration over [t0, t1] on the device, in two launches: the state rows y = (x, 0) from the points x
 * (row stride x_stride = 3 or 4 floats - a previous block's [rows,4] state is read in place), f0 = f(t0, y), ctl reset,
 * torchdiffeq's `_select_initial_step` into ctl[1].  n_tot: elements of the RMS norm; extra_d0 (DEVICE double, nullable)
 * x extra_scale: what the state rows outside y add to |y0 / scale|^2 (the context: pf_scaled_sumsq(c, NULL, c, ...)); no host
 * read anywhere in an integration's start.  ws: >= 3072 doubles.  ctl: ALL ZERO before its first use (every launch leaves its arrival word,
 * ctl[13], zero again). */
int pf_cnf_init(double* ctl, const float* x, int x_stride, float* y, float* f0, const float* ctx, const float* e,
                const float* rec, double t0, double t1, double n_tot, const double* extra_d0, double extra_scale, int reverse,
                float rtol, float atol, int rows, int R, double* ws, void* stream);

/* out[i] = sum_{j<n_terms} w[j] * ptrs[j][i]   (n_terms <= 8; ptrs / w are HOST arrays).  Runge-Kutta solution,
 * mid-point and dense-output combinations of torchdiffeq's dopri5 (cnf.py:97-113 call site). */
int pf_lincomb(const float* const* ptrs, const float* w, int n_terms, float* out, long long n, void* stream);

/* out[0] (double, device) = sum_i (v_i / (atol + rtol * max(|s0_i|, |s1_i|)))^2, v = a - b (b, s1 nullable), or
 * v = h * sum_j w[j] * k[j][i] when n_terms > 0 (the embedded error estimate).  ws: >= 256 doubles.  Deterministic. */
int pf_scaled_sumsq(const float* a, const float* b, const float* s0, const float* s1, const float* k, const float* w,
                    int n_terms, float h, float rtol, float atol, long long n, double* ws, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PUFLOW_HIP_H */
