/* libgraphecho_hip.so -- C ABI of the MI355X (gfx950) GraphEcho hot path.
 *
 * The reference (xmed-lab/GraphEcho) has no native code: its "FFI" for this path is the set of ATen / cuDNN /
 * cuBLAS ops that models/*.py and utils/*.py dispatch.  Each entry point below replaces one of those op families
 * and cites the reference call sites it stands in for (paths relative to the reference tree).
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes; device pointers are fp32 contiguous NCHW / row-major unless noted; indices int64
 *   - `stream` is a hipStream_t (0 = default stream); the call only enqueues work: never blocks, never allocates
 *   - the caller owns every buffer including workspaces (sizes documented per call)
 *   - returns 0 on success, <0 on error (-1 bad argument, -2 launch failure, -3 unsupported);
 *     ge_last_error() returns the message for the calling thread
 *   - thread-safe with respect to distinct streams
 * The declarations are parsed by graphecho_amd/_lib.py to build the ctypes binding, so keep one declaration per
 * statement and only the scalar types used here.
 */
#ifndef GRAPHECHO_HIP_H
#define GRAPHECHO_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ---------------------------------------------------------------------------------------------- */
const char* ge_last_error(void);
/* name, as rocprofv3 --kernel-trace prints it, of the conv kernel instantiation the calling thread launched last
 * (bench.py keys its live HIP-event timings by it so they can be checked against the rocprof summary) */
const char* ge_last_conv_kernel(void);
/* Measurement hook (bench.py's roofline leg; no reference counterpart): while `event` (a hipEvent_t) is set for the
 * calling thread, ge_conv2d_wgrad / ge_conv2d_f16_wgrad record it on their stream between the weight-gradient kernel
 * and its split-K slab reduce.  Pass NULL to clear. */
void ge_set_wgrad_split_event(void* event);
int ge_abi_version(void);
int ge_device_count(void);

/* Weight AND bias gradient of a convolution in one pass (models/fpnseg.py:332-352 lateral / smooth layers, models/vig.py:395-401,
 * 480, 530-535: every Grapher / FFN conv carries a bias): db[Cout] (+)= sum over (b, y, x) of dy is the row sum of the operand
 * tile the weight-gradient kernel stages anyway, folded by the same slab reduce -- replaces a separate full read of dy
 * (ge_channel_sum).  ge_conv2d_wgrad_fuses_bias: 1 where the layer's weight-gradient kernel supports it (the general MFMA
 * kernel; not the 3x3 / s1 / p1 patch kernel, not the one-output-channel reduction).  Workspace: ge_conv2d_wgrad_workspace. */
int ge_conv2d_wgrad_fuses_bias(int B, int Cin, int Cout, int Hi, int Wi, int Ho, int Wo, int kh, int kw, int stride, int pad,
                               int groups);
int ge_conv2d_wgrad_bias(const float* x, const float* dy, float* dw, float* db, float* workspace, int B, int Cin, int Hi,
                         int Wi, int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, int accumulate,
                         void* stream);

/* ---- conv2d (nn.Conv2d: models/fpnseg.py:28-139,170,174,221,332-352,457-473; models/vig.py:395,401,480,530,535;
 *      models/TGCN.py:53,57,185).  fp32 MFMA implicit GEMM, NCHW, im2col-free. ------------------------------- */
/* OIHW weights -> K-major operand layout; transposed=0 for ge_conv2d_fwd, 1 for ge_conv2d_dgrad.
 * out holds Cout*Cin_g*kh*kw floats. */
int ge_conv2d_pack_weight(const float* w, float* out, int Cout, int Cin_g, int kh, int kw, int groups, int transposed, void* stream);
/* every conv weight of a model in ONE launch (after the optimizer step): table = device int64 [n][8] rows
 * (src_off, dst_off, total, groups, Cout/groups, Cin/groups, kh*kw, transposed), offsets in floats into w / out */
int ge_conv2d_pack_weights_batched(const float* w, float* out, const long long* table, int n, void* stream);
/* stats (nullable): [Cout][ge_conv2d_fwd_stat_parts()][3] = per-tile (count, mean, M2) of y, i.e. the BatchNorm batch
 * statistics fused into the conv epilogue (merge them with ge_bn_finalize) */
int ge_conv2d_fwd_stat_parts(int B, int Cin, int Cout, int Ho, int Wo, int kh, int kw, int groups);
int ge_conv2d_fwd(const float* x, const float* wp, const float* bias, float* y, float* stats, int B, int Cin, int Hi, int Wi, int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, int relu, void* stream);
/* addend (nullable): tensor of dx's shape added to the result (gradient arriving through a skip connection) */
int ge_conv2d_dgrad(const float* dy, const float* wp, const float* addend, float* dx, int B, int Cin, int Hi, int Wi, int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, void* stream);
/* split-K variants for layers whose tile grid cannot fill the 256 CUs (B*Ho*Wo of a few thousand): the reduction over
 * Cin*kh*kw (forward) / Cout*kh*kw (data gradient) is cut across workgroups, partial results go to `workspace` and one
 * pass adds them up.  *_workspace() returns the floats wanted, 0 = use the plain entry point.  No fused statistics /
 * activation; data gradient: stride 1 only */
long long ge_conv2d_fwd_workspace(int B, int Cin, int Cout, int Ho, int Wo, int kh, int kw, int groups);
int ge_conv2d_fwd_splitk(const float* x, const float* wp, const float* bias, float* y, int B, int Cin, int Hi, int Wi, int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, float* workspace, void* stream);
long long ge_conv2d_dgrad_workspace(int B, int Cin, int Hi, int Wi, int Cout, int kh, int kw, int stride, int groups);
int ge_conv2d_dgrad_splitk(const float* dy, const float* wp, const float* addend, float* dx, int B, int Cin, int Hi, int Wi, int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, float* workspace, void* stream);
long long ge_conv2d_wgrad_workspace(int B, int Cin, int Cout, int Ho, int Wo, int kh, int kw, int groups);
int ge_conv2d_wgrad(const float* x, const float* dy, float* dw, float* workspace, int B, int Cin, int Hi, int Wi, int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, int accumulate, void* stream);
/* Deferred slab reduce (launch-bound small batches): ge_conv2d_wgrad called with accumulate | 2 leaves its
 * ge_conv2d_wgrad_splits() K-split slabs in the workspace; up to 16 such calls are then reduced by ONE launch
 * (host arrays of slab / destination pointers, element counts, split counts, accumulate flags) -- bit-identical
 * to the per-call reduce (same sums, same order).  ge_conv2d_wgrad_splits() == 0: the layer cannot be deferred. */
int ge_conv2d_wgrad_splits(int B, int Cin, int Cout, int Hi, int Wi, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups);
int ge_slab_reduce_batched(const float* const* slabs, float* const* outs, const long long* ns, const long long* strides,
                           const int* splits, const int* accumulate, int count, void* stream);
/* out[c] (+)= sum_{b,hw} x[b][c][hw]  (conv bias gradient); partial: [B][C] workspace */
int ge_channel_sum(const float* x, float* out, float* partial, int B, int C, int HW, int accumulate, void* stream);

/* ---- strided batched GEMM (nn.Linear / torch.bmm / torch.mm: models/transformer.py:14,22,32-38,71;
 *      models/graph_matching.py:148-162,166,191-202,605; models/affinity_layer.py:20-30; models/TGCN.py:207-218).
 *      C[m*scm+n*scn] = alpha*sum_k A[m*sam+k*sak]*B[k*sbk+n*sbn] (+bias: 1 per-m, 2 per-n)(+C when accumulate)(relu) */
int ge_gemm(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, long long sam, long long sak, long long sbk, long long sbn, long long scm, long long scn, int batch, long long bsA, long long bsB, long long bsC, float alpha, int bias_mode, int relu, int accumulate, void* stream);

/* nn.Linear backward in one launch (models/transformer.py:14-38, models/graph_matching.py:148-162, models/TGCN.py node_dis_2):
 * C = alpha * op(A) op(B) as ge_gemm and asum[batch][M] (+)= sum_k op(A)[m][k] -- with A = dY^T the bias gradient -- from the A
 * chunks the kernel stages anyway.  Only for products the 32 x 32-tile kernel takes: ge_gemm_rowsum_ok. */
int ge_gemm_rowsum_ok(int M, int N, int K, int batch);
int ge_gemm_rowsum(const float* A, const float* B, float* C, int M, int N, int K, long long sam, long long sak, long long sbk,
                   long long sbn, long long scm, long long scn, int batch, long long bsA, long long bsB, long long bsC,
                   float alpha, int accumulate, float* asum, int asum_accumulate, void* stream);

/* ---- BatchNorm2d (models/fpnseg.py:34-139,183-187,222,240; models/vig.py:396,402,456,531,536; models/TGCN.py:54,186)
 *      partial: [C][ge_bn_num_partials(B,HW)][3] floats = (count, mean, M2) per slice; stats: [C][3]. */
int ge_bn_num_partials(int B, int HW);
int ge_bn_stats_partial(const float* x, float* partial, int B, int C, int HW, void* stream);
/* Merge NB moment triples per channel (element (c,i) at partial[c*stride_c + i*stride_b]); any output may be null.
 * running_* are updated in place with `momentum` (unbiased variance), as nn.BatchNorm2d does in train mode. */
int ge_bn_finalize(const float* partial, long long stride_c, long long stride_b, int NB, int C, float eps, float momentum, float* stats, float* mean, float* invstd, float* running_mean, float* running_var, void* stream);
/* y = (x-mean)*invstd*gamma+beta (+residual)(relu); gamma/beta/residual may be null */
int ge_bn_apply(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, const float* residual, float* y, int B, int C, int HW, int relu, void* stream);
/* sums[C][2] = (sum dy_m, sum dy_m*xhat); dy_m = dy*(out>0) when the saved output `out` is given, or -- with
 * out == null and recompute_relu != 0 -- dy*(fma(x,sc,sh)>0), the mask recomputed bit-identically from x, gamma, beta
 * (BN+ReLU without residual: the output is not re-read); partial: [C][nb][2] workspace;
 * dgamma/dbeta [C] (nullable) receive (or, with accumulate, are incremented by) the affine gradients */
int ge_bn_bwd_reduce(const float* dy, const float* x, const float* out, const float* mean, const float* invstd, const float* gamma, const float* beta, int recompute_relu, float* partial, float* sums, float* dgamma, float* dbeta, int accumulate, int B, int C, int HW, void* stream);
int ge_bn_bwd_apply(const float* dy, const float* x, const float* out, const float* mean, const float* invstd, const float* gamma, const float* beta, int recompute_relu, const float* sums, float inv_count, float* dx, float* dres, int B, int C, int HW, void* stream);

/* Big layers without SyncBN: reduce + apply in TWO launches (no finalize in between): the apply workgroups fold their channel's
 * per-slice sums themselves, in ge_bn_bwd_reduce's finalize order (identical bits), and leave dgamma / dbeta (+)=.
 * partial: C * ge_bn_num_partials(B, HW) * 2 floats. */
int ge_bn_bwd_two_launch_ok(int B, int HW);
int ge_bn_bwd_partials(const float* dy, const float* x, const float* out, const float* mean, const float* invstd,
                       const float* gamma, const float* beta, int recompute_relu, float* partial, int B, int C, int HW,
                       void* stream);
int ge_bn_bwd_apply_partials(const float* dy, const float* x, const float* out, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, int recompute_relu, const float* partial, float* dgamma,
                             float* dbeta, int accumulate, float inv_count, float* dx, float* dres, int B, int C, int HW,
                             void* stream);
/* small layers (ge_bn_channel_ok: B*HW <= 16384 per channel, HW % 4 == 0): the whole train-mode forward (moments from
 * conv-epilogue partials or from x, running statistics, apply + residual + ReLU) resp. the whole backward in ONE launch,
 * one workgroup per channel; same arguments as the pieces above.  Not for SyncBN */
int ge_bn_channel_ok(int B, int HW);
int ge_bn_fwd_channel(const float* x, const float* partial, long long stride_c, long long stride_b, int NB, const float* gamma, const float* beta, const float* residual, float* y, float* mean, float* invstd, float* running_mean, float* running_var, int B, int C, int HW, float eps, float momentum, int relu, void* stream);

/* Big layers without SyncBN whose moments come from the conv epilogue: ge_bn_finalize + ge_bn_apply in ONE launch (every
 * workgroup of a channel repeats the wave merge -- identical statistics -- and applies its slice of frames). */
int ge_bn_fwd_merge_apply_ok(int NB, int HW);
int ge_bn_fwd_merge_apply(const float* x, const float* partial, long long stride_c, long long stride_b, int NB,
                          const float* gamma, const float* beta, const float* residual, float* y, float* mean, float* invstd,
                          float* running_mean, float* running_var, int B, int C, int HW, float eps, float momentum, int relu,
                          void* stream);
/* first half of the backward alone, for SyncBN: sums [C][2] (and the local dgamma / dbeta) of a small layer in one launch */
int ge_bn_bwd_reduce_channel(const float* dy, const float* x, const float* out, const float* mean, const float* invstd, const float* gamma, const float* beta, int recompute_relu, float* sums, float* dgamma, float* dbeta, int accumulate, int B, int C, int HW, void* stream);
int ge_bn_bwd_channel(const float* dy, const float* x, const float* out, const float* mean, const float* invstd, const float* gamma, const float* beta, int recompute_relu, float* dgamma, float* dbeta, int accumulate, float inv_count, float* dx, float* dres, int B, int C, int HW, void* stream);
/* the S <= 16 independent passes concatenated along the batch (source / target / clip frames of one step, the time steps of a TGCN clip; models/fpnseg.py
 * BatchNorm2d call sites :139-214 seen once per pass by the reference) in ONE launch each way; seg: HOST array of
 * S x (first frame, frames, offset of the segment's triples in a channel's partials, triples); mean / invstd: [S][C] */
int ge_bn_fwd_channel_segs(const float* x, const float* partial, long long stride_c, long long stride_b, const int* seg, int S, const float* gamma, const float* beta, const float* residual, float* y, float* mean, float* invstd, float* running_mean, float* running_var, int C, int HW, float eps, float momentum, int relu, void* stream);
int ge_bn_bwd_channel_segs(const float* dy, const float* x, const float* out, const float* mean, const float* invstd, const float* gamma, const float* beta, int recompute_relu, float* dgamma, float* dbeta, int accumulate, const int* seg, int S, float* dx, float* dres, int C, int HW, void* stream);
/* the same segments under SyncBN (torch.nn.SyncBatchNorm as train_camus_echo.py:129-142 converts the reference's BatchNorm2d layers): the statistics
 * cross the ranks between the two halves of each direction, so a layer is finalize_segs -> all-gather -> fwd_channel_segs_sync forward and
 * bwd_reduce_channel_segs -> all-reduce -> bwd_apply_channel_segs backward, every segment in each launch.  stats [S][C][3]; gathered [world][S][C][3];
 * sums [S][C][2]; inv_count: HOST array of S floats = 1 / (frames * HW * world).  ge_bn_fwd_merge_apply_sync: a BIG layer's segment -- merge of its
 * channel's [world] gathered triples + apply in one launch (ge_bn_finalize + ge_bn_apply otherwise) */
int ge_bn_fwd_merge_apply_sync(const float* x, const float* gathered, long long stride_c, long long stride_b, int world, const float* gamma, const float* beta, const float* residual, float* y, float* mean, float* invstd, float* running_mean, float* running_var, int B, int C, int HW, float eps, float momentum, int relu, void* stream);
int ge_bn_finalize_segs(const float* partial, long long stride_c, long long stride_b, const int* seg, int S, int C, int HW, float* stats, void* stream);
int ge_bn_stats_channel_segs(const float* x, const int* seg, int S, int C, int HW, float* stats, void* stream);
int ge_bn_fwd_channel_segs_sync(const float* x, const float* gathered, int world, const int* seg, int S, const float* gamma, const float* beta, const float* residual, float* y, float* mean, float* invstd, float* running_mean, float* running_var, int C, int HW, float eps, float momentum, int relu, void* stream);
int ge_bn_bwd_reduce_channel_segs(const float* dy, const float* x, const float* out, const float* mean, const float* invstd, const float* gamma, const float* beta, int recompute_relu, float* sums, float* dgamma, float* dbeta, int accumulate, const int* seg, int S, int C, int HW, void* stream);
int ge_bn_bwd_apply_channel_segs(const float* dy, const float* x, const float* out, const float* mean, const float* invstd, const float* gamma, const float* beta, int recompute_relu, const float* sums, const float* inv_count, const int* seg, int S, float* dx, float* dres, int C, int HW, void* stream);

/* ---- GroupNorm (models/fpnseg.py:354-355,465) / LayerNorm (models/transformer.py:40; models/graph_matching.py:150,
 *      153,193-199; models/TGCN.py:209-215) ------------------------------------------------------------------ */
int ge_groupnorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* invstd, int B, int C, int HW, int G, float eps, int relu, void* stream);
/* dgamma_part/dbeta_part: [B][C] workspaces; dgamma/dbeta [C] may be null */
int ge_groupnorm_bwd(const float* dy, const float* x, const float* out, const float* gamma, const float* mean, const float* invstd, float* dx, float* dgamma_part, float* dbeta_part, float* dgamma, float* dbeta, int B, int C, int HW, int G, void* stream);
int ge_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* invstd, int R, int D, float eps, void* stream);
int ge_layernorm_bwd_blocks(int R);
/* dgamma_part/dbeta_part: [ge_layernorm_bwd_blocks(R)][D] workspaces (null when no affine) */
int ge_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* invstd, float* dx, float* dgamma_part, float* dbeta_part, float* dgamma, float* dbeta, int R, int D, void* stream);
int ge_colsum(const float* in, float* out, int R, int C, void* stream);
/* out[c] += sum_r in[r][c] */
int ge_colsum_accumulate(const float* in, float* out, int R, int C, void* stream);
/* the same for two sources of one shape in ONE launch: a GroupNorm / LayerNorm layer's d gamma and d beta partials
 * (models/fpnseg.py:354-355,465; models/transformer.py:40) */
int ge_colsum_accumulate2(const float* in0, float* out0, const float* in1, float* out1, int R, int C, void* stream);
/* whole-tensor (count, mean, M2) for nn.InstanceNorm2d(1) over the affinity matrix (models/graph_matching.py:177,574);
 * partial: [64][3] workspace */
int ge_tensor_moments(const float* x, float* partial, float* stats, long long n, void* stream);
int ge_strided_sum3(const float* partial, float* sums, int n, int nb, void* stream);

/* ---- bilinear resize, align_corners=True, with fused lateral add (FPN._upsample_add / _upsample,
 *      models/fpnseg.py:358-359,371-388); pooling (models/fpnseg.py:43-141,224; models/vig.py:200; models/TGCN.py:66,189);
 *      activations (ReLU throughout; GELU models/TGCN.py:55,187, models/vig.py:443-444) --------------------- */
int ge_upsample_bilinear_fwd(const float* x, const float* add, float* y, int B, int C, int Hi, int Wi, int Ho, int Wo, void* stream);
int ge_upsample_bilinear_bwd(const float* dy, float* dx, int B, int C, int Hi, int Wi, int Ho, int Wo, void* stream);
int ge_maxpool2d_fwd(const float* x, float* y, unsigned char* arg, int B, int C, int Hi, int Wi, int Ho, int Wo, int k, int s, int p, void* stream);
int ge_maxpool2d_bwd(const float* dy, const unsigned char* arg, float* dx, int B, int C, int Hi, int Wi, int Ho, int Wo, int k, int s, int p, void* stream);
int ge_avgpool2d_fwd(const float* x, float* y, int B, int C, int Hi, int Wi, int r, void* stream);
int ge_avgpool2d_bwd(const float* dy, float* dx, int B, int C, int Hi, int Wi, int r, void* stream);
int ge_plane_mean(const float* x, float* y, long long planes, int HW, void* stream);
/* act_layer (models/vig.py:433-450): mode 0 = relu (bwd ref = output), 1 = gelu/erf, 2 = leaky relu with `slope`,
 * 3 = hardswish (bwd ref = input for 1..3) */
int ge_act_fwd(const float* x, float* y, long long n, int mode, float slope, void* stream);
int ge_act_bwd(const float* dy, const float* ref, float* dx, long long n, int mode, float slope, void* stream);
/* PReLU with ONE learned slope (act 'prelu'): partial = ge_prelu_num_partials(n) floats of workspace, dslope 1 float */
int ge_prelu_num_partials(long long n);
int ge_prelu_fwd(const float* x, const float* slope, float* y, long long n, void* stream);
int ge_prelu_bwd(const float* dy, const float* x, const float* slope, float* dx, float* partial, float* dslope, long long n, void* stream);
/* reductions over the neighbour dimension of [rows][K] edge tensors (EdgeConv2d / GraphSAGE max, GINConv2d sum,
 * models/vig.py:108-160): max keeps the first arg-max (uint8) for the backward */
int ge_lastdim_max_fwd(const float* x, float* y, unsigned char* arg, long long rows, int K, void* stream);
int ge_lastdim_max_bwd(const float* dy, const unsigned char* arg, float* dx, long long rows, int K, void* stream);
int ge_lastdim_sum_fwd(const float* x, float* y, long long rows, int K, void* stream);
int ge_lastdim_sum_bwd(const float* dy, float* dx, long long rows, int K, void* stream);

/* ---- Grapher: dense k-NN graph + max-relative aggregation (models/vig.py:209-229 batched_index_select,
 *      232-274 *_pairwise_distance, 277-329 *_dense_knn_matrix, 332-381 DenseDilated*, 88-105 MRConv2d) ------- */
/* x [B][C][P] -> xn (L2-normalised over C when normalize!=0) and sq [B][P] = sum_c xn^2 */
int ge_knn_prepare(const float* x, float* xn, float* sq, int B, int C, int P, int normalize, void* stream);
/* edge_index int64 [2][B][N][ceil(K/dilation)]: [0] neighbour ids (nearest first, ties -> lowest id), [1] centre ids */
int ge_knn_topk(const float* xn, const float* sqx, const float* yn, const float* sqy, const float* relpos, long long* edge_index, int B, int C, int N, int M, int K, int dilation, void* stream);
/* the same graph from the RAW query tensor x (models/vig.py:372-378 F.normalize + :262-329): the kernel normalises its own 64
 * query rows in its prologue and writes them to xn (workspace of x's size; pinned arithmetic order: bit-identical to
 * ge_knn_prepare(x) + ge_knn_topk); yn / sqy: candidates from ge_knn_prepare with the same `normalize` */
int ge_knn_topk_fused(const float* x, float* xn, const float* yn, const float* sqy, const float* relpos, long long* edge_index, int B, int C, int N, int M, int K, int dilation, int normalize, void* stream);
/* out [B][2C][N] channel-interleaved (x_0, max_0, x_1, max_1, ...); argk uint8 [B][C][N].
 * centre_is_self != 0 asserts edge_index[1][b][n][k] == n (true for every graph ge_knn_topk builds) and selects the
 * LDS-tiled kernels; 0 keeps the general gather for arbitrary centre ids. */
int ge_mrconv_gather_fwd(const float* x, const float* y, const long long* edge, float* out, unsigned char* argk, int B, int C, int N, int M, int K, int centre_is_self, void* stream);
/* floats of workspace ge_mrconv_gather_bwd needs for these extents (may be 0) */
long long ge_mrconv_gather_bwd_workspace(int B, int C, int N, int M, int K, int centre_is_self);
/* dx and dy are overwritten; pass dy == dx for the self graph (y is x) */
int ge_mrconv_gather_bwd(const float* dout, const long long* edge, const unsigned char* argk, float* dx, float* dy, float* workspace, int B, int C, int N, int M, int K, int centre_is_self, void* stream);
/* The same backward as a DETERMINISTIC GATHER over inverse neighbour lists (centre-is-self graphs; bit-reproducible, no
 * floating-point atomics; replaces the scatter of vig.py:88-105's autograd for the graphs of :262-329):
 *   ge_mrconv_gather_bwd_det_ok(N, M, K, centre_is_self) == 0: take ge_mrconv_gather_bwd;
 *   ge_mr_inv_chunk(): nodes per chunk J of the lists;
 *   ge_mr_inv_build: edge int64 [2][B][N][K] -> inv uint32 [B][N*K] (node-in-chunk | k << 16, grouped per chunk by
 *   neighbour id in a fixed order; granules of 64 consecutive nodes are dealt to the ceil(N/J) chunks round-robin) and
 *   off int32 [B][ceil(N/J)][M+1] (segment starts inside each chunk);
 *   ge_mrconv_gather_bwd_det: dx [B][C][N], dy [B][C][M] overwritten; dy == dx for the self graph; workspace: floats per
 *   ge_mrconv_gather_bwd_det_workspace (partial sums of chunk splits, folded in split order; may be 0 -> null) */
int ge_mrconv_gather_bwd_det_ok(int N, int M, int K, int centre_is_self);
/* small graphs (<= 128 nodes by default: TGCN's 64-node graphs, the 8 x 8 pyramid level): the same deterministic backward
 * in ONE launch (every workgroup inverts the edge list itself, in LDS); ge_mrconv_gather_bwd_small_ok() == 1: take it,
 * no ge_mr_inv_build needed */
int ge_mrconv_gather_bwd_small_ok(int N, int M, int K, int centre_is_self);
int ge_mrconv_gather_bwd_small(const float* dout, const long long* edge, const unsigned char* argk, float* dx, float* dy, int B, int C, int N, int M, int K, void* stream);
int ge_mr_inv_chunk(void);
int ge_mr_inv_build(const long long* edge, unsigned* inv, int* off, int B, int N, int M, int K, void* stream);
long long ge_mrconv_gather_bwd_det_workspace(int B, int C, int N, int M, int K, int y_is_x);
int ge_mrconv_gather_bwd_det(const float* dout, const unsigned* inv, const int* off, const unsigned char* argk, float* dx, float* dy, float* workspace, int B, int C, int N, int M, int K, void* stream);

/* batched_index_select (vig.py:209-229): out [B][C][E] = src [B][C][M] gathered by idx [B][E] (int64), E = N*K edges;
 * backward overwrites dsrc with the scatter-add of dout (LDS accumulation per row, M <= 16384) */
int ge_edge_gather_fwd(const float* src, const long long* idx, float* out, int B, int C, int M, int E, void* stream);
int ge_edge_gather_bwd(const float* dout, const long long* idx, float* dsrc, int B, int C, int M, int E, void* stream);

/* ---- Sinkhorn: SinkhornDistance (utils/sinkhorn_distance.py:27-86) and GModule.sinkhorn_rpm
 *      (models/graph_matching.py:637-689, slack=True) ------------------------------------------------------- */
/* uh [B][T+1][P1], vh [B][T+1][P2], err [B][T] are kept for backward; nits [1] = iterations the reference runs */
int ge_sinkhorn_distance_fwd(const float* x, const float* y, float* Cm, float* pi, float* cost, int* nits, float* uh, float* vh, float* err, int B, int P1, int P2, int D, float eps, int max_iter, float thresh, void* stream);
/* the same forward in ONE launch (cost tile LDS-resident in both orientations, 16-lane-row sweeps, device-side stopping rule
 * across the batch through a two-int meeting point `sync` -- caller-owned, one pair per (device, stream), zeroed on the
 * stream by the call itself);
 * ge_sinkhorn_distance_fused_ok() == 0: take ge_sinkhorn_distance_fwd (tile beyond LDS, or B above a quarter of the
 * workgroups of this kernel the CURRENT device holds resident at once -- occupancy x CU count, queried, at most 128) */
int ge_sinkhorn_distance_fused_ok(int B, int P1, int P2);
int ge_sinkhorn_distance_fwd_fused(const float* x, const float* y, float* Cm, float* pi, float* cost, int* nits, float* uh, float* vh, float* err, int* sync, int B, int P1, int P2, int D, float eps, int max_iter, float thresh, void* stream);
int ge_sinkhorn_distance_bwd(const float* x, const float* y, const float* Cm, const float* uh, const float* vh, const int* nits, const float* g_cost, const float* g_pi, const float* g_C, float* dC, float* dx, float* dy, int B, int P1, int P2, int D, float eps, int max_iter, void* stream);
/* rho_hist [T][B][N1], gamma_hist [T+1][B][N2] are kept for backward */
int ge_sinkhorn_rpm_fwd(const float* A, float* X, float* rho_hist, float* gamma_hist, int B, int N1, int N2, int n_iters, void* stream);
/* the same forward in ONE launch of 16 co-operating workgroups (B = 1, N2 <= 512, 16 <= N1 <= 640: the training step's sizes);
   workspace: ge_sinkhorn_rpm_coop_workspace floats (0: not offered), caller-owned, one per (device, stream) in flight */
long long ge_sinkhorn_rpm_coop_workspace(int B, int N1, int N2);
int ge_sinkhorn_rpm_fwd_coop(const float* A, float* X, float* rho_hist, float* gamma_hist, float* workspace, int N1, int N2, int n_iters, void* stream);
int ge_sinkhorn_rpm_bwd_coop(const float* A, const float* gX, const float* rho_hist, const float* gamma_hist, float* gA, float* workspace, int N1, int N2, int n_iters, void* stream);
int ge_sinkhorn_rpm_bwd(const float* A, const float* gX, const float* rho_hist, const float* gamma_hist, float* gA, float* g_rho, float* g_gamma, int B, int N1, int N2, int n_iters, void* stream);
/* one-to-one matching loss on the log plan (GModule._forward_aff, models/graph_matching.py:577-590): M = exp(X), tp / fp focal terms;
   idx [N1] int32, rowpart [N1][4], loss [1], scal [3] feed the backward; g_loss / gM nullable */
int ge_match_o2o_fwd(const float* X, const float* lab1, const float* lab2, float* M, int* idx, float* rowpart, float* loss, float* scal, int N1, int N2, void* stream);
int ge_match_o2o_bwd(const float* M, const float* lab1, const float* lab2, const int* idx, const float* scal, const float* g_loss, const float* gM, float* gX, int N1, int N2, void* stream);

/* ---- Affinity MLP, algebraically fused (models/affinity_layer.py:52-73): M = b2 + w2 . relu(P_i + Q_j + b1) -- */
int ge_affinity_fwd(const float* P, const float* Q, const float* b1, const float* w2, const float* b2, float* M, int N1, int N2, int H, void* stream);
/* dw2_part: [ceil(N1/2)][H] workspace */
int ge_affinity_bwd(const float* P, const float* Q, const float* b1, const float* w2, const float* dM, float* dP, float* dQ, float* dw2_part, int N1, int N2, int H, void* stream);

/* ---- attention softmax (models/transformer.py:10-20): y = softmax(scale*x) over the last dim ---------------- */
int ge_softmax_fwd(const float* x, float* y, int R, int D, float scale, void* stream);
int ge_softmax_bwd(const float* dy, const float* p, float* dx, int R, int D, float scale, void* stream);

/* ---- single-head MultiHeadAttention block (models/transformer.py:28-78, version "v2", num_heads = 1: every attention of GModule and
 *      TGCN) as one entry point per direction: k/v/q projections, softmax(q k^T scale), 0 / 1 dropout keep masks (made by the caller, att_scale / out_scale = 1 / keep probability of each site), A v, final
 *      projection, residual, LayerNorm -- the kernels of the composed form, issued from one call.  Saved tensors and the backward's
 *      workspace (ge_mha1_bwd_workspace floats) are caller-owned; w_acc / b_acc: add the parameter gradients to what is there */
long long ge_mha1_bwd_workspace(int Nk, int Nq, int D);
int ge_mha1_fwd(const float* key, const float* value, const float* query, const float* Wk, const float* bk, const float* Wv, const float* bv, const float* Wq, const float* bq, const float* Wf, const float* bf, const float* gamma, const float* beta, const float* mask_att, const float* mask_out, float* k, float* v, float* q, float* P, float* A, float* ctx, float* z, float* mean, float* invstd, float* out, int Nk, int Nq, int D, float scale, float att_scale, float out_scale, float eps, void* stream);
int ge_mha1_bwd(const float* key, const float* value, const float* query, const float* Wk, const float* Wv, const float* Wq, const float* Wf, const float* gamma, const float* mask_att, const float* mask_out, const float* k, const float* v, const float* q, const float* P, const float* A, const float* ctx, const float* z, const float* mean, const float* invstd, const float* d_out, const float* d_att, float* dkey, float* dvalue, float* dquery, float* dWk, float* dbk, float* dWv, float* dbv, float* dWq, float* dbq, float* dWf, float* dbf, int w_acc, int b_acc, float* dgamma, float* dbeta, float* ws, int Nk, int Nq, int D, float scale, float att_scale, float out_scale, void* stream);

/* ---- segmentation losses (nn.BCEWithLogitsLoss train_camus_echo.py:124; DiceLoss utils/losses.py:24-95) ----- */
/* t may be null (constant target tconst); partial: 1024-float workspace */
int ge_bce_logits_fwd(const float* x, const float* t, float tconst, float* partial, float* loss, long long n, void* stream);
int ge_bce_logits_bwd(const float* x, const float* t, float tconst, const float* g, float* dx, long long n, void* stream);
int ge_dice_num_partials(int HW);
/* prob [B][C][HW] softmax over C; sums [B][C][3] = (sum p*t, sum p^2, sum t^2); partial [B][C][nblk][3] */
int ge_dice_fwd(const float* x, const float* t, float* prob, float* partial, float* sums, int B, int C, int HW, void* stream);
int ge_dice_bwd(const float* prob, const float* t, const float* ca, const float* cb, float* dx, int B, int C, int HW, void* stream);

/* ---- optimizers on flat fp32 buffers (torch.optim.Adam / SGD, train_camus_echo.py:425-435) ------------------- */
int ge_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);
int ge_sgd_step(float* p, const float* g, float* buf, long long n, float lr, float momentum, float weight_decay, int first_step, float grad_scale, void* stream);
/* the same steps with the per-parameter "received a gradient on some rank" decision read on the DEVICE (data parallelism:
 * `used` is the MAX-all-reduced flag tensor, find_unused_parameters semantics of train_camus_echo.py:129-142 without a
 * host read): seg_end[nseg] ascending exclusive end offsets of the parameters in the flat buffer, i0 = offset of p[0],
 * used[s] > 0 steps parameter s; started[s] > 0: parameter s already has a momentum buffer (torch.optim.SGD's first step) */
int ge_adam_step_masked(float* p, const float* g, float* m, float* v, long long n, long long i0, const int* seg_end, const float* used, int nseg, float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);
int ge_sgd_step_masked(float* p, const float* g, float* buf, long long n, long long i0, const int* seg_end, const float* used, const float* started, int nseg, float lr, float momentum, float weight_decay, float grad_scale, void* stream);
int ge_flags_max(float* a, const float* b, int n, void* stream);

/* ---- fp16-input MFMA conv path (BASELINE.json config 5: "fp16 MFMA conv path + fp32 Sinkhorn"): the same
 *      nn.Conv2d call sites as above; tensors stay fp32 in HBM, operands are rounded to fp16 into LDS, fp32 accumulate.
 *      Layers whose Cin/groups or Cout/groups is not a multiple of 32 stay on the fp32 entry points. ------------- */
int ge_conv2d_f16_supported(int Cin, int Cout, int groups);
/* out: Cout*Cin_g*kh*kw halves; transposed=0 forward operand Wp[g][tap][co][ci], 1 data-gradient Wp[g][tap][ci][co] */
int ge_conv2d_f16_pack_weight(const float* w, void* out, int Cout, int Cin_g, int kh, int kw, int groups, int transposed, void* stream);
int ge_conv2d_f16_fwd_stat_parts(int B, int Cout, int Ho, int Wo, int groups);
int ge_conv2d_f16_fwd(const float* x, const void* wp, const float* bias, float* y, float* stats, int B, int Cin, int Hi, int Wi, int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, int relu, void* stream);
long long ge_conv2d_f16_wgrad_workspace(int B, int Cin, int Cout, int Ho, int Wo, int kh, int kw, int groups);
int ge_conv2d_f16_wgrad(const float* x, const float* dy, float* dw, float* workspace, int B, int Cin, int Hi, int Wi, int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, int accumulate, void* stream);
int ge_conv2d_f16_dgrad(const float* dy, const void* wp, const float* addend, float* dx, int B, int Cin, int Hi, int Wi, int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, void* stream);


/* ---- mean(x^2) of a whole tensor (the auxiliary activation loss that trains the Graphers in the config-2 harness,
 *      DESIGN.md section 6); partial: ge_mean_square_blocks(n) floats; g: device scalar (gradient of the mean) ---- */
int ge_mean_square_blocks(long long n);
int ge_mean_square_fwd(const float* x, float* partial, float* out, long long n, void* stream);
int ge_mean_square_bwd(const float* x, const float* g, float* dx, long long n, void* stream);

/* ---- input formatting in front of FPN.forward and the validation metric behind it ------- */
/* frames: nearest resize (Hs x Ws x Ts -> S x S x To; torch/MONAI 'nearest': src = min(floor(dst*in/out), in-1)),
 * crop x crop window, / divisor (255.0), clip fold (C,H,W,T) -> (T,C,H,W): datasets/cardiac_uda.py:248-286,155;
 * datasets/camus.py:121-159,103; train_camus_echo.py:247-251.  src [N][C][Hs][Ws][Ts] uint8 or fp32;
 * dst [N*To][C][crop][crop] fp32; offsets: device int [N][2] crop origins (y,x) in the resized frame, or null -> (oy,ox) */
int ge_frames_prepare(const void* src, int src_is_float, float* dst, const int* offsets, int N, int C, int Hs, int Ws, int Ts, int S, int To, int crop, int oy, int ox, float divisor, void* stream);
/* label maps -> one-hot planes over a class-value list (datasets/cardiac_uda.py:128-151, datasets/camus.py:98-101),
 * same geometry.  labels [N][Hs][Ws][Ts] uint8; values device int [NC]; dst [N*To][NC][crop][crop] fp32 */
int ge_labels_onehot(const unsigned char* labels, float* dst, const int* offsets, const int* values, int N, int NC, int Hs, int Ws, int Ts, int S, int To, int crop, int oy, int ox, void* stream);
/* counts int64 [C][4] += (TP, FP, FN, TN) of (logit > 0) vs (mask != 0) per class (train_camus_echo.py:402-417) */
int ge_overlap_counts(const float* logits, const float* masks, long long* counts, int B, int C, int HW, void* stream);

/* ---- front end of GModule's graph construction (models/graph_matching.py) -------------------------------------- */
/* masks_to_boxes (graph_matching.py:702-740): masks [n][h][w] fp32 -> boxes [n][4] = (x1, y1, x2, y2) of the non-zero
 * pixels, (0, 0, w, h) for an all-zero mask */
int ge_mask_boxes(const float* masks, float* boxes, int n, int h, int w, void* stream);
/* compute_targets_for_locations (graph_matching.py:874-959): boxes [batch][num_class][4] -> labels
 * [batch][sum_l h_l*w_l] bytes, levels concatenated.  hws: HOST int [levels][3] = (h, w, stride) -- a location's
 * coordinate is index*stride + stride/2 --, ranges: HOST float [levels][2] = the level's (lo, hi) regression range */
int ge_fcos_labels(const float* boxes, unsigned char* labels, int batch, int num_class, int levels, const int* hws, const float* ranges, void* stream);
/* rows sampled from the NCHW pyramid levels (graph_matching.py:961-1013): out[n][:] = f_level[n][b][:][p],
 * (b, p) = divmod(index[n], hw_level); f0..f4 / hw0..hw4: up to five levels, unused ones null / 0 */
int ge_gather_nodes_fwd(const float* f0, const float* f1, const float* f2, const float* f3, const float* f4, int hw0, int hw1, int hw2, int hw3, int hw4, int channels, const long long* level, const long long* index, float* out, int n, void* stream);
/* its backward: scatter of dout [n][channels] into the pre-zeroed level gradients d0..d4 (null: level without gradient);
 * atomic != 0 when two rows may name the same location */
int ge_gather_nodes_bwd(const float* dout, const long long* level, const long long* index, float* d0, float* d1, float* d2, float* d3, float* d4, int hw0, int hw1, int hw2, int hw3, int hw4, int channels, int n, int atomic, void* stream);
/* momentum update of one seed bank (GModule.update_seed, models/graph_matching.py:532-567): class means of the kept rows, cosine
   similarity with the bank row, blend; cls [N] int32 (-1: row dropped by the clustering), has [nc] int32 */
int ge_seed_bank_update(float* bank, const float* nodes, const int* cls, const int* has, int nc, int N, int D, void* stream);

/* ---- fp16 ACTIVATION STORAGE (BASELINE.json config 5: "fp16 MFMA conv path"): the conv3x3 -> BatchNorm -> ReLU
 *      (-> 2x2 max-pool) stacks of the VGG16 backbone (models/fpnseg.py:18-166, built by train_cardiac_uda.py:73) with every
 *      activation and activation gradient stored as fp16 in the channel-blocked layout h[b][c/32][y][x][c%32]
 *      ("blocked": a pixel's 32 channels are 64 contiguous bytes).  Statistics, parameters, weight gradients fp32.
 *      Gradients inside a stack carry a caller-chosen loss scale; the kernels that leave the stack take its inverse. ---- */
int ge_h_conv3x3_supported(int B, int Cin, int Cout, int H, int W);
int ge_h_conv3x3_stat_parts(int B, int H, int W);
/* z = conv3x3/s1/p1(x) (+bias); wp = ge_conv2d_f16_pack_weight(w, .., transposed=0); stats (nullable):
 * [Cout][ge_h_conv3x3_stat_parts][3] (count, mean, M2) per 64 pixels, the layout ge_bn_finalize merges (nn.Conv2d + the
 * batch statistics of the nn.BatchNorm2d behind it, fpnseg.py:28-139) */
int ge_h_conv3x3_fwd(const void* x, const void* wp, const float* bias, void* z, float* stats, int B, int Cin, int Cout, int H, int W, void* stream);
/* dx = data gradient; wp = ge_conv2d_f16_pack_weight(w, .., transposed=1) */
int ge_h_conv3x3_dgrad(const void* dz, const void* wp, void* dx, int B, int Cin, int Cout, int H, int W, void* stream);
/* the same passes LEAVING the fp16 domain (any 3x3 / s1 / p1 conv whose neighbours are fp32 kernels: the FPN smoothing /
 * head convs fpnseg.py:340-352, the Discriminator towers :457-473, Bottleneck.conv2 :182-187): fp32 NCHW results; the data
 * gradient is multiplied by out_scale (1 / loss scale) and takes the optional skip-connection addend (fp32 NCHW) */
int ge_h_conv3x3_fwd_f32(const void* x, const void* wp, const float* bias, float* y, float* stats, int B, int Cin, int Cout, int H, int W, void* stream);
int ge_h_conv3x3_dgrad_f32(const void* dz, const void* wp, const float* addend, float* dx, float out_scale, const float* dev_scale, int B, int Cin, int Cout, int H, int W, void* stream);
long long ge_h_conv3x3_wgrad_workspace(int B, int Cin, int Cout, int H, int W);
/* dw[Cout][Cin][3][3] fp32 (+)= scale * weight gradient (scale = 1 / loss scale); workspace: ge_h_conv3x3_wgrad_workspace floats */
int ge_h_conv3x3_wgrad(const void* x, const void* dz, float* dw, float* workspace, int B, int Cin, int Cout, int H, int W, float scale, const float* dev_scale, int accumulate, void* stream);
/* fp32 NCHW <-> blocked fp16 (C % 32 == 0), values multiplied by scale: entry to / exit from a stack */
int ge_h_from_f32(const float* x, void* h, int B, int C, int HW, float scale, float* dev_scale, void* stream);
int ge_h_to_f32(const void* h, float* x, int B, int C, int HW, float scale, const float* dev_scale, void* stream);
/* h = saturate(x * scale + addend), summed in fp32: the gradient of a blocked tensor that is read inside the fp16 domain (addend)
 * AND, through ge_h_to_f32, outside it (x) -- an fp16 + fp16 add of two large gradients would overflow to inf */
int ge_h_from_f32_add(const float* x, const void* addend, void* h, int B, int C, int HW, float scale, float* dev_scale, void* stream);
/* nn.BatchNorm2d (+ nn.ReLU) on blocked fp16 tensors; mean / invstd from ge_bn_finalize over the conv's stats */
int ge_h_bn_apply(const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta, void* a, int B, int C, int HW, int relu, void* stream);
int ge_h_bn_slices(int HW);
/* partial: C * B * ge_h_bn_slices(HW) * 2 floats; sums [C][2] = (sum g, sum g xhat) * inv_scale, i.e. in true units (SyncBN
 * all-reduces them: the ranks' loss scales need not agree), dgamma / dbeta (nullable) (+)= the same; ge_h_bn_bwd_apply takes the
 * loss scale of da back in (scale, dev_scale) */
int ge_h_bn_bwd_reduce(const void* da, const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta, int relu, float* partial, float* sums, float* dgamma, float* dbeta, int accumulate, float inv_scale, const float* dev_scale, int B, int C, int HW, void* stream);
int ge_h_bn_bwd_apply(const void* da, const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta, int relu, const float* sums, float inv_count, float scale, const float* dev_scale, void* dz, int B, int C, int HW, void* stream);
/* out[C] (+)= inv_scale * sum over (b, y, x) of dz: bias gradient of the conv in front (partial as above) */
int ge_h_channel_sum(const void* dz, float* partial, float* out, int accumulate, float inv_scale, const float* dev_scale, int B, int C, int HW, void* stream);
/* nn.GroupNorm(C / 8, C) (+ nn.ReLU) on blocked fp16 tensors -- the discriminator towers' GroupNorm(32, 256), fpnseg.py:465: a group
 * is one 16-byte vector of a pixel.  mean / invstd [B][C / 8] from the conv epilogue's moments; backward: partial B * C *
 * ge_h_bn_slices(HW) * 2 floats, sums B * C / 8 * 2 floats; dgamma / dbeta (nullable) (+)= inv_scale (x dev_scale[1]) * the sums */
int ge_h_gn8_stats(const float* stats, float* mean, float* invstd, int B, int C, int HW, float eps, void* stream);
int ge_h_gn8_apply(const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta, void* a, int B, int C, int HW, int relu, void* stream);
int ge_h_gn8_bwd(const void* da, const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta, int relu, float* partial, float* sums, float* dgamma, float* dbeta, int accumulate, float inv_scale, const float* dev_scale, void* dz, int B, int C, int HW, void* stream);
/* nn.BatchNorm2d + nn.ReLU + nn.MaxPool2d(2, 2) of a stack's last layer in ONE pass each way: the full-resolution activation is never
 * written (forward: z -> pooled y; backward: the window's argmax is recomputed from z, dy = the POOLED gradient); partial:
 * C * B * ge_h_bn_slices(H * W / 4) * 2 floats; sums / scales as ge_h_bn_bwd_reduce / ge_h_bn_bwd_apply */
int ge_h_bn_relu_pool_fwd(const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta, void* y, int B, int C, int H, int W, void* stream);
int ge_h_bn_relu_pool_bwd_reduce(const void* dy, const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta, float* partial, float* sums, float* dgamma, float* dbeta, int accumulate, float inv_scale, const float* dev_scale, int B, int C, int H, int W, void* stream);
int ge_h_bn_relu_pool_bwd_apply(const void* dy, const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta, const float* sums, float inv_count, float scale, const float* dev_scale, void* dz, int B, int C, int H, int W, void* stream);
/* nn.MaxPool2d(2, 2) (fpnseg.py:44,65,92,118,139) on blocked fp16 */
int ge_h_maxpool2_fwd(const void* x, void* y, int B, int C, int H, int W, void* stream);
int ge_h_maxpool2_bwd(const void* x, const void* dy, void* dx, int B, int C, int H, int W, void* stream);
/* the stem -- nn.Conv2d(in_channels, 64, 3, padding=1) on the fp32 image (fpnseg.py:28; 1 or 3 channels) -- straight into the
 * blocked fp16 domain: plain fp32 FMAs (K = 9 * Cin), fp16 stores, moments as ge_h_conv3x3_fwd; and its weight gradient */
int ge_h_stem3x3_supported(int B, int Cin, int Cout, int H, int W);
int ge_h_stem3x3_fwd(const float* x, const float* w, const float* bias, void* z, float* stats, int B, int Cin, int Cout, int H, int W, void* stream);
long long ge_h_stem3x3_wgrad_workspace(int B, int Cin, int Cout, int H, int W);
int ge_h_stem3x3_wgrad(const float* x, const void* dz, float* dw, float* workspace, int B, int Cin, int Cout, int H, int W, float scale, const float* dev_scale, int accumulate, void* stream);
/* DEVICE-RESIDENT LOSS SCALE hs = {scale, 1/scale, bits of the largest |gradient| cast since the last update, unused} (4 floats):
 * every `dev_scale` argument above (nullable) multiplies by hs[0] (casts to fp16, which also record the magnitude) or hs[1]
 * (kernels leaving the fp16 domain) on top of the host-side factor.  ge_h_scale_update: once per step -- the power of two that
 * puts the recorded magnitude at `target`, clamped to [lo, hi]; nothing recorded: unchanged (torch.cuda.amp.GradScaler's job,
 * without a host read) */
int ge_h_scale_init(float* hs, float scale, void* stream);
int ge_h_scale_update(float* hs, float target, float lo, float hi, void* stream);
/* lane mapping of gfx950's ds_read_b64_tr_b16 as the weight-gradient kernel assumes it: out[64][4] (tests) */
int ge_h_probe_tr(float* out, void* stream);

/* ---- fp32 3x3 / stride 1 / pad 1 convolution as Winograd F(2x2, 3x3) (ge_wino.hip): the 3x3 layers of the reference's
 * nn.Conv2d calls (models/fpnseg.py:182-187 Bottleneck.conv2, :340-352 smoothing / head convs, :457-473 discriminator towers),
 * forward and data gradient.  A pass has C reduction channels and M output channels (forward: C = Cin, M = Cout; data gradient:
 * C = Cout, M = Cin).  Covered: C % 8 == 0, M % 64 == 0, (W % 32 == 0 and H % 4 == 0) or (W % 16 == 0 and H % 8 == 0).
 * Layers whose grid cannot fill the chip run split over the input channels (slabs in the caller's workspace, added in split
 * order); _supported / _splits say whether (and how) the layer is routed, _covered only checks the geometry. */
int ge_wino3x3_supported(int B, int C, int M, int H, int W);
int ge_wino3x3_covered(int B, int C, int M, int H, int W);
int ge_wino3x3_splits(int B, int C, int M, int H, int W);
long long ge_wino3x3_workspace(int B, int C, int M, int H, int W);
long long ge_wino3x3_weight_floats(int C, int M);
int ge_wino3x3_stat_parts(int B, int H, int W);
/* u = transformed filters: transposed = 0: w is [M][C][3][3] (forward); 1: w is [C][M][3][3], taps rotated (data gradient) */
int ge_wino3x3_pack_weight(const float* w, float* u, int M, int C, int transposed, void* stream);
/* every Winograd operand of a model in ONE launch (after the optimizer step): table = device int64 [n][5] rows
 * (offset of the OIHW weight in flat in floats, destination device pointer, M, C, transposed) */
int ge_wino3x3_pack_weights_batched(const float* flat, const long long* table, int n, void* stream);
/* y[B][M][H][W] = conv3x3(x[B][C][H][W]) (+ bias[M]) (+ addend[B][M][H][W]); stats (nullable, unsplit layers only):
 * [M][ge_wino3x3_stat_parts()][3] = (count, mean, M2) of y per workgroup and channel (merge with ge_bn_finalize);
 * workspace: ge_wino3x3_workspace() floats (may be null when that is 0) */
int ge_wino3x3_fwd(const float* x, const float* u, const float* bias, const float* addend, float* y, float* stats, float* workspace, int B, int C, int M, int H, int W, void* stream);

/* ---- the same layers' WEIGHT gradient as Winograd F(3x3, 2x2) (ge_wino_wgrad.hip; replaces ge_conv2d_wgrad where covered):
 * x [B][C][H][W], dy [B][M][H][W] -> dw [M][C][3][3].  Covered: C % 32 == 0, M % 64 == 0, W % 16 == 0, H even; _supported also asks
 * for enough tiles to fill the chip.  Split over the tiles into slabs in the caller's workspace (ge_wino3x3_wgrad_workspace floats),
 * reduced in split order.  accumulate bit 0: add to dw; bit 1: leave the slabs (stride M * C * 9) to ge_slab_reduce_batched. */
int ge_wino3x3_wgrad_supported(int B, int C, int M, int H, int W);
int ge_wino3x3_wgrad_covered(int B, int C, int M, int H, int W);
int ge_wino3x3_wgrad_splits(int B, int C, int M, int H, int W);
/* the whole routing decision of the Winograd weight gradient (csrc/ge_wino_plan.h): *splits, *ws_kernel (1: the warp-specialised
 * kernel); returns ge_wino3x3_wgrad_supported */
int ge_wino3x3_wgrad_plan(int B, int C, int M, int H, int W, int* splits, int* ws_kernel);
long long ge_wino3x3_wgrad_workspace(int B, int C, int M, int H, int W);
int ge_wino3x3_wgrad(const float* x, const float* dy, float* dw, float* workspace, int B, int C, int M, int H, int W, int accumulate, void* stream);
/* the same plus the bias gradient db[M] (+)= sum(dy) (replaces the ge_channel_sum pass; reference: autograd of nn.Conv2d(bias=True),
   fpnseg.py:332-352); the workspace of ge_wino3x3_wgrad_workspace already holds the bias slabs */
int ge_wino3x3_wgrad_bias(const float* x, const float* dy, float* dw, float* db, float* workspace, int B, int C, int M, int H, int W, int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GRAPHECHO_HIP_H */
