// k-reciprocal re-ranking (Zhong et al., CVPR 2017) on the device: torchreid/utils/re_ranking.py:30-95, the optional
// post-process of the reference's test() (train_vidreid_xent_htri.py:523-527, --re-rank). N = m + n samples.
//
//   rr_colmax / rr_build    joint matrix [[qq, qg], [qg^T, gg]], squared, every column scaled by its maximum, transposed
//                           (re_ranking.py:33-38): D (N x N fp32), built tile-wise through LDS so both sides coalesce
//   agrl_rank_topk          the argsort of :40 is only ever read through its first k1+1 columns: exact stable top-(k1+1)
//   rr_kreciprocal          per sample: k-reciprocal neighbours, 2/3-overlap expansion, sorted unique set, Gaussian
//                           weights normalised per row -> V (:43-64); one wavefront per sample, sets in LDS
//   rr_expand               local query expansion: mean of the k2 nearest samples' rows (:66-70), written TRANSPOSED
//   rr_jaccard              per query: sum over its non-zero columns c (ascending) of min(V[i][c], V[r][c]) for ALL r at
//                           once -- one coalesced row of V^T per column (:78-87; a zero contributes + 0.0, which changes
//                           nothing) -> jaccard -> final = (1 - lambda) jaccard + lambda D (:89-94)
// Dense fp32 matrices in HBM (4 x N^2 x 4 B = 3.2 GB at the MARS sizes); every sum runs in a fixed order.
#include "agrl_common.h"

namespace {

__device__ inline float joint_sq(const float* qq, const float* qg, const float* gg, int m, int n, int r, int c) {
    float v;
    if (r < m) v = c < m ? qq[(size_t)r * m + c] : qg[(size_t)r * n + (c - m)];
    else v = c < m ? qg[(size_t)c * n + (r - m)] : gg[(size_t)(r - m) * n + (c - m)];
    return v * v;
}

// column maxima of X = joint^2 (X >= 0: unsigned bit patterns order like the floats)
__global__ __launch_bounds__(256) void rr_colmax_kernel(const float* qq, const float* qg, const float* gg, int m, int n,
                                                        unsigned* __restrict__ colmax) {
    const int N = m + n;
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r0 = blockIdx.y * 64;
    if (c >= N) return;
    float mx = 0.f;
    for (int r = r0; r < min(r0 + 64, N); ++r) mx = fmaxf(mx, joint_sq(qq, qg, gg, m, n, r, c));
    atomicMax(colmax + c, __float_as_uint(mx));
}

// D[i][j] = X[j][i] / colmax[i]: 32 x 32 tiles through LDS (read X rows coalesced, write D rows coalesced)
__global__ __launch_bounds__(256) void rr_build_kernel(const float* qq, const float* qg, const float* gg, int m, int n,
                                                       const unsigned* __restrict__ colmax, float* __restrict__ D) {
    __shared__ float tile[32][33];
    const int N = m + n;
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;  // D tile rows i0.., cols j0..  <- X rows j0.., cols i0..
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int xr = j0 + k, xc = i0 + tx;
        tile[k][tx] = (xr < N && xc < N) ? joint_sq(qq, qg, gg, m, n, xr, xc) : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int i = i0 + k, j = j0 + tx;
        if (i < N && j < N) D[(size_t)i * N + j] = tile[tx][k] / __uint_as_float(colmax[i]);
    }
}

constexpr int RR_MAXSET = 1024;

// one wavefront per sample i
__global__ __launch_bounds__(64) void rr_kreciprocal_kernel(const float* __restrict__ D, const int* __restrict__ rank, int K, int N,
                                                            int k1p, int half, float* __restrict__ V) {
    __shared__ int s_base[64];
    __shared__ int s_set[RR_MAXSET];
    __shared__ float s_w[RR_MAXSET];
    __shared__ int s_cnt[4];
    const int i = blockIdx.x, lane = threadIdx.x;
    const int* ri = rank + (size_t)i * K;
    // base = { fwd[a] : i in rank[fwd[a]][:k1p] }, order of a kept
    bool mine = false;
    int cand = -1;
    if (lane < k1p) {
        cand = ri[lane];
        const int* rc = rank + (size_t)cand * K;
        for (int b = 0; b < k1p; ++b) mine = mine || rc[b] == i;
    }
    unsigned long long mask = __ballot(mine);
    const int nbase = __popcll(mask);
    if (mine) s_base[__popcll(mask & ((1ull << lane) - 1ull))] = cand;
    if (lane == 0) s_cnt[0] = nbase;
    __syncthreads();
    if (lane < nbase) s_set[lane] = s_base[lane];
    int nset = nbase;
    __syncthreads();
    // expansion: every member c of the base set contributes its own (half-size) reciprocal set when > 2/3 of it lies in base
    for (int ci = 0; ci < nbase; ++ci) {
        const int c = s_base[ci];
        const int* rcand = rank + (size_t)c * K;
        bool in_rc = false;
        int x = -1;
        if (lane < half) {
            x = rcand[lane];
            const int* rx = rank + (size_t)x * K;
            for (int b = 0; b < half; ++b) in_rc = in_rc || rx[b] == c;
        }
        const unsigned long long m2 = __ballot(in_rc);
        const int nrc = __popcll(m2);
        bool in_base = false;
        if (in_rc)
            for (int b = 0; b < nbase; ++b) in_base = in_base || s_base[b] == x;
        const int inter = __popcll(__ballot(in_base));
        // reference: len(intersect1d) > 2./3*len(candidate set), evaluated in double
        if ((double)inter > 2.0 / 3 * (double)nrc) {
            if (in_rc && nset + nrc <= RR_MAXSET) s_set[nset + __popcll(m2 & ((1ull << lane) - 1ull))] = x;
            nset += nrc;
        }
        __syncthreads();
    }
    if (nset > RR_MAXSET) nset = RR_MAXSET;  // cannot happen for k1 <= 30
    // sorted unique: a first occurrence lands at position = number of distinct smaller values
    __shared__ int s_sorted[RR_MAXSET];
    __shared__ unsigned char s_first[RR_MAXSET];
    __shared__ int s_nuniq;
    if (lane == 0) s_nuniq = 0;
    for (int e = lane; e < nset; e += 64) {
        const int v = s_set[e];
        bool first = true;
        for (int o = 0; o < e; ++o) first = first && s_set[o] != v;
        s_first[e] = first ? 1 : 0;
    }
    __syncthreads();
    for (int e = lane; e < nset; e += 64) {
        if (!s_first[e]) continue;
        const int v = s_set[e];
        int pos = 0;
        for (int o = 0; o < nset; ++o) pos += (s_first[o] && s_set[o] < v) ? 1 : 0;
        s_sorted[pos] = v;
        atomicAdd(&s_nuniq, 1);
    }
    __syncthreads();
    const int nu = s_nuniq;
    const float* Di = D + (size_t)i * N;
    for (int e = lane; e < nu; e += 64) s_w[e] = expf(-Di[s_sorted[e]]);
    __syncthreads();
    // np.sum of a short fp32 vector: numpy's pairwise kernel -- 8 running sums over blocks of 8, combined as
    // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), the tail added one by one; vectors longer than 128 split in two
    __shared__ float s_sum;
    if (lane == 0) {
        auto block_sum = [&](int lo, int cnt) {
            if (cnt < 8) {
                float r = 0.f;
                for (int t = 0; t < cnt; ++t) r += s_w[lo + t];
                return r;
            }
            float r[8];
            for (int t = 0; t < 8; ++t) r[t] = s_w[lo + t];
            int t = 8;
            for (; t < cnt - (cnt % 8); t += 8)
                for (int u = 0; u < 8; ++u) r[u] += s_w[lo + t + u];
            float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
            for (; t < cnt; ++t) res += s_w[lo + t];
            return res;
        };
        float total;
        if (nu <= 128) {
            total = block_sum(0, nu);
        } else {  // one level of the recursion covers nu <= 256; deeper levels for the (unreachable) larger sets
            auto rec2 = [&](int lo, int cnt) {
                int h = cnt / 2;
                h -= h % 8;
                return block_sum(lo, h) + block_sum(lo + h, cnt - h);
            };
            if (nu <= 256) total = rec2(0, nu);
            else {
                int h = nu / 2;
                h -= h % 8;
                total = rec2(0, h) + rec2(h, nu - h);
            }
        }
        s_sum = total;
    }
    __syncthreads();
    float* Vi = V + (size_t)i * N;
    for (int e = lane; e < nu; e += 64) Vi[s_sorted[e]] = s_w[e] / s_sum;
}

// VT[c][i] = mean over the k2 nearest samples a of V[rank[i][a]][c]  (k2 == 1: VT = V^T), 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void rr_expand_kernel(const float* __restrict__ V, const int* __restrict__ rank, int K, int N,
                                                        int k2, int m, float* __restrict__ V2, float* __restrict__ VT) {
    __shared__ float tile[32][33];
    const int i0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int i = i0 + k, c = c0 + tx;
        float v = 0.f;
        if (i < N && c < N) {
            if (k2 == 1) {
                v = V[(size_t)i * N + c];
            } else {
                const int* ri = rank + (size_t)i * K;
                for (int a = 0; a < k2; ++a) v += V[(size_t)ri[a] * N + c];   // row order, as add.reduce over axis 0
                v = v / (float)k2;
            }
            if (V2 && i < m) V2[(size_t)i * N + c] = v;   // only the query rows are read again (rr_jaccard)
        }
        tile[k][tx] = v;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, i = i0 + tx;
        if (c < N && i < N) VT[(size_t)c * N + i] = tile[tx][k];
    }
}

constexpr int RR_JT = 256;
constexpr int RR_JMAX = 64;   // rows per thread: N <= 16384

__global__ __launch_bounds__(RR_JT) void rr_jaccard_kernel(const float* __restrict__ V2, const float* __restrict__ VT,
                                                           const float* __restrict__ D, int m, int N, float c_jac, float c_org,
                                                           float* __restrict__ fin, int ldf, int* __restrict__ nzbuf,
                                                           float* __restrict__ nzval) {
    __shared__ int s_cnt;
    __shared__ int s_wcnt[RR_JT / 64];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* vi = V2 + (size_t)i * N;
    int* nz = nzbuf + (size_t)i * N;
    float* nv = nzval + (size_t)i * N;
    // ordered compaction of the non-zero columns of V2[i]
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int base = 0; base < N; base += RR_JT) {
        const int c = base + tid;
        const float v = c < N ? vi[c] : 0.f;
        const bool on = v != 0.f;
        const unsigned long long mk = __ballot(on);
        if (lane == 0) s_wcnt[wave] = __popcll(mk);
        __syncthreads();
        int off = s_cnt;
        for (int w = 0; w < wave; ++w) off += s_wcnt[w];
        if (on) {
            const int p = off + __popcll(mk & ((1ull << lane) - 1ull));
            nz[p] = c;
            nv[p] = v;
        }
        __syncthreads();
        if (tid == 0) s_cnt += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        __syncthreads();
    }
    const int nnz = s_cnt;
    __threadfence_block();
    float t[RR_JMAX];
#pragma unroll
    for (int k = 0; k < RR_JMAX; ++k) t[k] = 0.f;
    for (int e = 0; e < nnz; ++e) {
        const float vic = nv[e];
        const float* row = VT + (size_t)nz[e] * N;
#pragma unroll
        for (int k = 0; k < RR_JMAX; ++k) {
            const int r = tid + RR_JT * k;
            if (r < N) t[k] += fminf(vic, row[r]);
        }
    }
    const float* Di = D + (size_t)i * N;
#pragma unroll
    for (int k = 0; k < RR_JMAX; ++k) {
        const int r = tid + RR_JT * k;
        if (r >= m && r < N) {
            const float jac = 1.f - t[k] / (2.f - t[k]);
            fin[(size_t)i * ldf + (r - m)] = jac * c_jac + Di[r] * c_org;
        }
    }
}

}  // namespace

extern "C" size_t agrl_re_ranking_workspace(int m, int n, int k1) {
    const size_t N = (size_t)m + n;
    const size_t K = (size_t)k1 + 1;
    return 4 * N * N * sizeof(float) + N * sizeof(unsigned) + N * K * (sizeof(int) + sizeof(float)) +
           (size_t)m * N * (sizeof(int) + sizeof(float)) + 4096;
}

extern "C" int agrl_re_ranking(const float* q_g, const float* q_q, const float* g_g, int m, int n, int k1, int k2, double lambda_value,
                               float* final_dist, int ldf, void* workspace, size_t workspace_bytes, agrl_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AGRL_CHECK_ARG(q_g && q_q && g_g && final_dist && workspace, "agrl_re_ranking: null pointer");
    AGRL_CHECK_ARG(m > 0 && n > 0 && ldf >= n, "agrl_re_ranking: bad shape m=%d n=%d ldf=%d", m, n, ldf);
    const int N = m + n, K = k1 + 1;
    const int half = (int)nearbyint(k1 / 2.0) + 1;  // np.around: round half to even
    AGRL_CHECK_ARG(k1 >= 1 && k1 <= 30 && K <= N, "agrl_re_ranking: 1 <= k1 <= 30 and k1 < m + n (got %d)", k1);
    AGRL_CHECK_ARG(k2 >= 1 && k2 <= K, "agrl_re_ranking: 1 <= k2 <= k1 + 1 (got %d)", k2);
    AGRL_CHECK_ARG(N <= RR_JT * RR_JMAX, "agrl_re_ranking: m + n = %d exceeds %d", N, RR_JT * RR_JMAX);
    AGRL_CHECK_ARG(workspace_bytes >= agrl_re_ranking_workspace(m, n, k1), "agrl_re_ranking: workspace too small");
    unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
    const size_t NN = (size_t)N * N * sizeof(float);
    float* D = reinterpret_cast<float*>(w);
    float* V = reinterpret_cast<float*>(w + NN);
    float* V2 = reinterpret_cast<float*>(w + 2 * NN);
    float* VT = reinterpret_cast<float*>(w + 3 * NN);
    unsigned char* p = w + 4 * NN;
    unsigned* colmax = reinterpret_cast<unsigned*>(p); p += ((size_t)N * 4 + 255) / 256 * 256;
    int* rank = reinterpret_cast<int*>(p); p += ((size_t)N * K * 4 + 255) / 256 * 256;
    float* rval = reinterpret_cast<float*>(p); p += ((size_t)N * K * 4 + 255) / 256 * 256;
    int* nzbuf = reinterpret_cast<int*>(p); p += ((size_t)m * N * 4 + 255) / 256 * 256;
    float* nzval = reinterpret_cast<float*>(p);

    if (hipMemsetAsync(colmax, 0, (size_t)N * 4, stream) != hipSuccess || hipMemsetAsync(V, 0, NN, stream) != hipSuccess) {
        agrl_set_error("agrl_re_ranking: hipMemsetAsync failed");
        return 2;
    }
    hipLaunchKernelGGL(rr_colmax_kernel, dim3(cdiv(N, 256), cdiv(N, 64)), dim3(256), 0, stream, q_q, q_g, g_g, m, n, colmax);
    hipLaunchKernelGGL(rr_build_kernel, dim3(cdiv(N, 32), cdiv(N, 32)), dim3(256), 0, stream, q_q, q_g, g_g, m, n, colmax, D);
    AGRL_CHECK_LAUNCH("agrl_re_ranking(build)");
    int rc = agrl_rank_topk(D, N, N, N, K, 0, rank, rval, stream_);
    if (rc) return rc;
    hipLaunchKernelGGL(rr_kreciprocal_kernel, dim3(N), dim3(64), 0, stream, D, rank, K, N, K, half, V);
    hipLaunchKernelGGL(rr_expand_kernel, dim3(cdiv(N, 32), cdiv(N, 32)), dim3(256), 0, stream, V, rank, K, N, k2, m, V2, VT);
    const float c_jac = (float)(1.0 - lambda_value), c_org = (float)lambda_value;  // numpy rounds the Python scalars to fp32
    hipLaunchKernelGGL(rr_jaccard_kernel, dim3(m), dim3(RR_JT), 0, stream, V2, VT, D, m, N, c_jac, c_org, final_dist, ldf, nzbuf, nzval);
    AGRL_CHECK_LAUNCH("agrl_re_ranking");
    return 0;
}
