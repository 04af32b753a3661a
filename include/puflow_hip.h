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
 * cfg 0: unit 0 (input = xyz [B*N,3], `tab` = [96,8] folded edge table);
 * cfg 1: unit 1, cfg 2: units 2..5 (input = PQ [B*N, 2S] per-point vectors, tab ignored).
 * idx [B*N,16] int32 (index inside the batch item); wfrag = fragment-packed growth weights;
 * out [B*N, odim]. */
int pf_edgeconv(int cfg, const float* pq_or_xyz, const float* tab, const int* idx, const float* wfrag, float* out,
                int B, int N, void* stream);

/* Same as pf_edgeconv with an explicit tuning variant (points per wave / waves per workgroup);
 * variant 0 is what pf_edgeconv ships.  Used by tools/tune_edgeconv.py for in-process A/B timing. */
int pf_edgeconv_tuned(int cfg, int variant, const float* pq_or_xyz, const float* tab, const int* idx,
                      const float* wfrag, float* out, int B, int N, void* stream);

/* Per-point stage after EdgeConv unit `unit` (0..5): FeatMergeUnit (interpflow.py:251-258), the
 * injector conditioner nets (coupling.py:132-134), coupling1's c-part and the next unit's PQ.
 * off[12] = float offsets into `w` of: M1,b1,M2,H1,S2,bS2,T2,bT2,ST4,bST4,PQ,bPQ.
 * c [T,cdim] (nullable), st [T,8], cp [T,64], pq_next [T,2S'] (NULL allowed for unit 5 only). */
int pf_post(int unit, const float* h, const float* w, const long long* off, float* c, float* st, float* cp,
            float* pq_next, int T, void* stream);

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

/* Interpolation module (interpflow.py:85-186), fused: kNN-8 context features -> weights ->
 * softmax_k -> weighted sum of neighbour latents.  idx16 [T,16] (first 8 columns used),
 * u_out [T*R,3] in the row order of g (row = n*R + r).  R must be 4.
 * off[13]: float offsets into w (csrc/interp.hip header, packing.INTERP_SLOTS). */
int pf_interp(const float* xyz, const float* z, const int* idx16, const float* w, const long long* off, float* u_out,
              int B, int N, int R, void* stream);

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

/* Auction EMD backward.  Replaces emd.backward (emd.cpp:21-24 -> emd_cuda.cu:284-316):
 * gradxyz [B,n,3] += 2 graddist (xyz1 - xyz2[idx]); gradient wrt xyz2 is zero (emd_module.py:68-72). */
int pf_emd_backward(const float* xyz1, const float* xyz2, float* gradxyz, const float* graddist, const int* idx, int B,
                    int n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PUFLOW_HIP_H */
