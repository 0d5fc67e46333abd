// Ranking side of the match step, on device:
//   agrl_rank_topk         np.argsort(distmat[k])[:max_rank]           torchreid/metrics/rank.py:170-172
//   agrl_rank_mars         evaluate_mars / Compute_AP per query         torchreid/metrics/rank.py:160-212
//   agrl_triplet_hard_mine batch-hard mining of TripletLoss.forward     torchreid/losses/hard_mine_triplet_loss.py:33-45
//
// Top-k is an exact radix select on the order-preserving 32-bit image of the fp32 distance followed by
// an index-ordered tie fill and a bitonic sort of the k winners on (key, index): the result equals a
// STABLE argsort truncated to k (ties -> lower gallery index), independent of launch geometry.
#include "agrl_common.h"

namespace {

__device__ inline uint32_t dist_key(float d) {
    uint32_t u = __float_as_uint(d);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;  // NaN sorts last (numpy convention)
    if (u == 0x80000000u) u = 0;                               // -0.0 == +0.0
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float key_dist(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
    return __uint_as_float(u);
}

// exclusive prefix sum of `flag` over the 256-thread block, plus the block total; s_w: 4+ ints of LDS
__device__ inline int block_excl_scan(int flag, int* s_w, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag != 0);
    const int within = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    int base = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int c = s_w[w];
        if (w < wave) base += c;
        total += c;
    }
    __syncthreads();
    return base + within;
}

// One row by radix select (reads the row 5 x: 4 histogram passes + the collect): any n, k <= 1024, any alignment.
// s_cand: kpad composites of LDS.
__device__ void radix_topk_row(const float* __restrict__ row, int n, int k, int kpad, int idx_offset,
                               int32_t* __restrict__ idx_row, float* __restrict__ val_row, unsigned long long* s_cand) {
    __shared__ int s_hist[256];
    __shared__ int s_w[4];
    __shared__ uint32_t s_prefix;
    __shared__ int s_kth, s_cnt;
    const int tid = threadIdx.x;

    if (tid == 0) {
        s_prefix = 0;
        s_kth = k;
        s_cnt = 0;
    }
    // ---- radix select of the k-th smallest key, 8 bits per pass, MSB first
    for (int shift = 24; shift >= 0; shift -= 8) {
        s_hist[tid] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix;
        const uint32_t himask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
        for (int j = tid; j < n; j += 256) {
            const uint32_t key = dist_key(row[j]);
            if ((key & himask) == prefix) atomicAdd(&s_hist[(key >> shift) & 0xff], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int kth = s_kth, cum = 0, bin = 0;
            for (; bin < 256; ++bin) {
                const int c = s_hist[bin];
                if (cum + c >= kth) break;
                cum += c;
            }
            s_prefix = prefix | ((uint32_t)bin << shift);
            s_kth = kth - cum;
        }
        __syncthreads();
    }
    const uint32_t T = s_prefix;  // k-th smallest key
    const int need_eq = s_kth;    // how many elements equal to T belong to the top-k (lowest indices)
    // ---- collect: every key < T, and the first need_eq keys == T in index order
    int eq_seen = 0;
    for (int base = 0; base < n; base += 256) {
        const int j = base + tid;
        uint32_t key = 0xffffffffu;
        bool lt = false, eq = false;
        if (j < n) {
            key = dist_key(row[j]);
            lt = key < T;
            eq = key == T;
        }
        int tot_eq;
        const int r_eq = block_excl_scan(eq ? 1 : 0, s_w, tot_eq);
        const bool take = lt || (eq && eq_seen + r_eq < need_eq);
        if (take) {
            const int pos = atomicAdd(&s_cnt, 1);
            s_cand[pos] = ((unsigned long long)key << 32) | (uint32_t)j;
        }
        eq_seen += tot_eq;
    }
    __syncthreads();
    for (int i = k + tid; i < kpad; i += 256) s_cand[i] = ~0ull;
    __syncthreads();
    // ---- bitonic sort of kpad composites (ascending)
    for (int size = 2; size <= kpad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (kpad >> 1); i += 256) {
                const int lo = ((i / stride) * (stride << 1)) + (i % stride);
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const unsigned long long a = s_cand[lo], b = s_cand[hi];
                if ((a > b) == up) {
                    s_cand[lo] = b;
                    s_cand[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < k; i += 256) {
        const unsigned long long c = s_cand[i];
        idx_row[i] = (int32_t)(uint32_t)(c & 0xffffffffull) + idx_offset;
        val_row[i] = key_dist((uint32_t)(c >> 32));
    }
}

__global__ __launch_bounds__(256) void rank_topk_kernel(const float* __restrict__ dist, int n, int ldd, int k,
                                                        int kpad, int idx_offset, int32_t* __restrict__ idx_out,
                                                        float* __restrict__ val_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_cand_dyn[];  // kpad composites
    radix_topk_row(dist + (size_t)blockIdx.x * ldd, n, k, kpad, idx_offset, idx_out + (size_t)blockIdx.x * k,
                   val_out + (size_t)blockIdx.x * k, s_cand_dyn);
}

// ---------------------------------------------------------------------------------------------------------------
// Bandwidth-bound top-k (k <= 128, n <= 1024 NV, 16-byte aligned rows): the row crosses HBM ONCE, as NV 16-byte loads
// per thread all in flight together, and stays in registers as order-preserving keys.
//   1. every thread's smallest key -> LDS; tau = the k-th smallest of those 256 minima. They are k distinct elements of
//      the row, so the row's k-th smallest key is <= tau: tau is an upper bound that needs no histogram;
//   2. candidates = every key <= tau (for i.i.d. data ~1.1 k of them; any data: >= k), compacted into LDS as
//      (key << 32 | index) composites;
//   3. each candidate's rank among the candidates by counting (composites are distinct: the index breaks ties towards the
//      lower gallery index); ranks < k are the answer, written in place.
// The result equals the radix kernel's bit for bit (a stable argsort truncated to k, NaN last). Rows whose candidate
// count exceeds TOPK_CAP (a row of mostly equal values) take the radix path from global memory, same kernel.
constexpr int TOPK_CAP = 1024;

template <int NV>
__global__ __launch_bounds__(256) void rank_topk_fast_kernel(const float* __restrict__ dist, int n, int ldd, int k, int kpad,
                                                             int idx_offset, int32_t* __restrict__ idx_out, float* __restrict__ val_out) {
    __shared__ __attribute__((aligned(16))) unsigned long long s_cand[TOPK_CAP];
    __shared__ __attribute__((aligned(16))) uint32_t s_min[256];
    __shared__ uint32_t s_tau;
    __shared__ int s_cnt;
    const int tid = threadIdx.x;
    const float* row = dist + (size_t)blockIdx.x * ldd;
    uint32_t key[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j0 = (i * 256 + tid) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j0 + 3 < n) {
            v = *reinterpret_cast<const float4*>(row + j0);
        } else {
            if (j0 < n) v.x = row[j0];
            if (j0 + 1 < n) v.y = row[j0 + 1];
            if (j0 + 2 < n) v.z = row[j0 + 2];
        }
        key[i][0] = j0 < n ? dist_key(v.x) : 0xffffffffu;
        key[i][1] = j0 + 1 < n ? dist_key(v.y) : 0xffffffffu;
        key[i][2] = j0 + 2 < n ? dist_key(v.z) : 0xffffffffu;
        key[i][3] = j0 + 3 < n ? dist_key(v.w) : 0xffffffffu;
    }
    uint32_t mine = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) mine = key[i][e] < mine ? key[i][e] : mine;
    s_min[tid] = mine;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    {   // rank of this thread's minimum among the 256 (ties -> lower thread id): the thread of rank k - 1 publishes tau
        int r = 0;
#pragma unroll 8
        for (int u4 = 0; u4 < 64; ++u4) {
            const uint4 o = *reinterpret_cast<const uint4*>(&s_min[u4 * 4]);
            const int u = u4 * 4;
            r += (o.x < mine || (o.x == mine && u < tid)) ? 1 : 0;
            r += (o.y < mine || (o.y == mine && u + 1 < tid)) ? 1 : 0;
            r += (o.z < mine || (o.z == mine && u + 2 < tid)) ? 1 : 0;
            r += (o.w < mine || (o.w == mine && u + 3 < tid)) ? 1 : 0;
        }
        if (r == k - 1) s_tau = mine;
    }
    __syncthreads();
    const uint32_t tau = s_tau;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = (i * 256 + tid) * 4 + e;
            if (j < n && key[i][e] <= tau) {
                const int pos = atomicAdd(&s_cnt, 1);
                if (pos < TOPK_CAP) s_cand[pos] = ((unsigned long long)key[i][e] << 32) | (uint32_t)j;
            }
        }
    __syncthreads();
    const int C = s_cnt;
    int32_t* idx_row = idx_out + (size_t)blockIdx.x * k;
    float* val_row = val_out + (size_t)blockIdx.x * k;
    if (C > TOPK_CAP) {   // block-uniform: a row dominated by ties at the threshold
        __syncthreads();
        radix_topk_row(row, n, k, kpad, idx_offset, idx_row, val_row, s_cand);
        return;
    }
    for (int i = tid; i < C; i += 256) {
        const unsigned long long c = s_cand[i];
        int r = 0;
        for (int j = 0; j < C; ++j) r += s_cand[j] < c ? 1 : 0;
        if (r < k) {
            idx_row[r] = (int32_t)(uint32_t)(c & 0xffffffffull) + idx_offset;
            val_row[r] = key_dist((uint32_t)(c >> 32));
        }
    }
}

// Full stable argsort of every row (the cuhk03 protocol walks the WHOLE ranking, torchreid/metrics/rank.py:45-47): one
// workgroup per row, (key << 32 | index) composites in LDS, bitonic sort. Composites are distinct, so the order is the stable
// one (ties -> lower gallery index, NaN last) whatever the sorting network does. n <= 16384 (128 KB of composites).
__global__ __launch_bounds__(1024) void rank_argsort_kernel(const float* __restrict__ dist, int n, int ldd, int npad,
                                                            int32_t* __restrict__ idx_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_c[];
    const int tid = threadIdx.x;
    const float* row = dist + (size_t)blockIdx.x * ldd;
    for (int i = tid; i < npad; i += 1024) s_c[i] = i < n ? (((unsigned long long)dist_key(row[i]) << 32) | (uint32_t)i) : ~0ull;
    __syncthreads();
    for (int size = 2; size <= npad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (npad >> 1); i += 1024) {
                const int lo = ((i / stride) * (stride << 1)) + (i % stride);
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const unsigned long long a = s_c[lo], b = s_c[hi];
                if ((a > b) == up) {
                    s_c[lo] = b;
                    s_c[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < n; i += 1024) idx_out[(size_t)blockIdx.x * n + i] = (int32_t)(uint32_t)(s_c[i] & 0xffffffffull);
}

// grid = m queries. Block-reduce ngood over the whole gallery, then lane 0 replays Compute_AP in fp64
// with the reference's operation order, so ap is bit-identical to the Python floats.
__global__ __launch_bounds__(256) void rank_mars_kernel(const int32_t* __restrict__ topk, const int32_t* __restrict__ q_pids,
                                                        const int32_t* __restrict__ q_camids,
                                                        const int32_t* __restrict__ g_pids,
                                                        const int32_t* __restrict__ g_camids, int n, int k,
                                                        double* __restrict__ ap_out, float* __restrict__ cmc_out) {
    __shared__ int s_part[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    const int qp = q_pids[q], qc = q_camids[q];
    int cnt = 0;
    for (int j = tid; j < n; j += 256) cnt += (g_pids[j] == qp && g_camids[j] != qc) ? 1 : 0;
    cnt = wave_sum_i(cnt);
    if ((tid & 63) == 0) s_part[tid >> 6] = cnt;
    __syncthreads();
    float* cmc = cmc_out + (size_t)q * k;
    for (int i = tid; i < k; i += 256) cmc[i] = 0.f;
    __syncthreads();
    if (tid != 0) return;
    const int ngood = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    const int32_t* index = topk + (size_t)q * k;
    double old_recall = 0.0, old_precision = 1.0, ap = 0.0;
    int intersect = 0, j = 0, good_now = 0, njunk = 0;
    int first_hit = -1;  // cmc[first_hit:] = 1
    for (int t = 0; t < k; ++t) {
        const int gi = index[t];
        const int gp = g_pids[gi], gc = g_camids[gi];
        const bool good = (gp == qp) && (gc != qc);
        const bool junk = (gp == -1) || ((gp == qp) && (gc == qc));
        int flag = 0;
        if (good) {
            const int from = t - njunk;
            if (first_hit < 0 || from < first_hit) first_hit = from;
            flag = 1;
            ++good_now;
        }
        if (junk) {
            ++njunk;
            continue;
        }
        if (flag) ++intersect;
        const double recall = (double)intersect / (double)ngood;
        const double precision = (double)intersect / (double)(j + 1);
        ap += (recall - old_recall) * (old_precision + precision) / 2;
        old_recall = recall;
        old_precision = precision;
        ++j;
        if (good_now == ngood) break;
    }
    if (first_hit >= 0)
        for (int i = first_hit; i < k; ++i) cmc[i] = 1.f;
    ap_out[q] = ap;
}

// grid = n anchors; each wavefront strides over the candidates j, lanes over the feature dim.
__global__ __launch_bounds__(256) void triplet_mine_kernel(const float* __restrict__ x, const int32_t* __restrict__ pids,
                                                           int n, int d, float* __restrict__ dist_ap,
                                                           float* __restrict__ dist_an, int32_t* __restrict__ idx_ap,
                                                           int32_t* __restrict__ idx_an) {
    __shared__ float s_ap[4], s_an[4];
    __shared__ int s_iap[4], s_ian[4];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xi = x + (size_t)i * d;
    float ni = 0.f;
    for (int c = lane; c < d; c += 64) ni = fmaf(xi[c], xi[c], ni);
    ni = wave_sum(ni);
    const int pi = pids[i];
    float best_ap = -INFINITY, best_an = INFINITY;
    int bi_ap = -1, bi_an = -1;
    for (int j = wave; j < n; j += 4) {
        const float* xj = x + (size_t)j * d;
        float dot = 0.f, nj = 0.f;
        for (int c = lane; c < d; c += 64) {
            const float a = xi[c], b = xj[c];
            dot = fmaf(a, b, dot);
            nj = fmaf(b, b, nj);
        }
        dot = wave_sum(dot);
        nj = wave_sum(nj);
        const float dd = sqrtf(fmaxf((ni + nj) - 2.f * dot, 1e-12f));
        if (pids[j] == pi) {
            if (dd > best_ap) { best_ap = dd; bi_ap = j; }
        } else {
            if (dd < best_an) { best_an = dd; bi_an = j; }
        }
    }
    if (lane == 0) {
        s_ap[wave] = best_ap; s_iap[wave] = bi_ap;
        s_an[wave] = best_an; s_ian[wave] = bi_an;
    }
    __syncthreads();
    if (tid == 0) {
        float ap = -INFINITY, an = INFINITY;
        int iap = -1, ian = -1;
        for (int w = 0; w < 4; ++w) {
            if (s_iap[w] >= 0 && (s_ap[w] > ap || (s_ap[w] == ap && s_iap[w] < iap))) { ap = s_ap[w]; iap = s_iap[w]; }
            if (s_ian[w] >= 0 && (s_an[w] < an || (s_an[w] == an && s_ian[w] < ian))) { an = s_an[w]; ian = s_ian[w]; }
        }
        dist_ap[i] = ap; dist_an[i] = an; idx_ap[i] = iap; idx_an[i] = ian;
    }
}


// Loss value and per-anchor pair coefficients from the mined distances (one workgroup; n is a train batch, <= a few hundred):
//   soft  : L_i = log(1 + exp(d_ap - d_an)),           dL_i/d(d_ap) = sigmoid(d_ap - d_an)
//   margin: L_i = max(0, d_ap - d_an + margin),        dL_i/d(d_ap) = [L_i > 0]           (MarginRankingLoss, y = 1)
// loss = mean_i L_i (summed in anchor order). d = sqrt(clamp(d2, 1e-12)): where the clamp is active the distance is a
// constant and passes no gradient. ca[i] = dL/d(d_ap) / (n d_ap), cn[i] = -dL/d(d_an) ... / (n d_an): the factors of
// (x_i - x_j) in the feature gradient. An anchor without a negative (idx_an < 0) makes the loss NaN.
__device__ inline void triplet_term(float ap, float an, int has_neg, float margin, int soft, float& L, float& g) {
    if (!has_neg) {
        L = __builtin_nanf("");
        g = 0.f;
    } else if (soft) {
        const float e = expf(ap - an);
        L = logf(1.f + e);
        g = e < INFINITY ? e / (1.f + e) : 1.f;
    } else {
        L = fmaxf(0.f, (ap - an) + margin);
        g = L > 0.f ? 1.f : 0.f;
    }
}

__global__ __launch_bounds__(256) void triplet_coeff_kernel(const float* __restrict__ dist_ap, const float* __restrict__ dist_an,
                                                            const int32_t* __restrict__ idx_an, int n, float margin, int soft,
                                                            float* __restrict__ loss, float* __restrict__ ca, float* __restrict__ cn) {
    const float tiny = sqrtf(1e-12f);
    for (int i = threadIdx.x; i < n; i += 256) {
        const float ap = dist_ap[i], an = dist_an[i];
        float L, g;
        triplet_term(ap, an, idx_an[i] >= 0, margin, soft, L, g);
        ca[i] = ap > tiny ? g / ((float)n * ap) : 0.f;
        cn[i] = (idx_an[i] >= 0 && an > tiny) ? -g / ((float)n * an) : 0.f;
    }
    if (threadIdx.x == 0) {  // anchor-ordered sum by one thread: n is a train batch; fixed order = deterministic
        float tot = 0.f;
        for (int i = 0; i < n; ++i) {
            float L, g;
            triplet_term(dist_ap[i], dist_an[i], idx_an[i] >= 0, margin, soft, L, g);
            tot += L;
        }
        loss[0] = tot / (float)n;
    }
}

// grad[r] = ca[r] (x_r - x_p(r)) + cn[r] (x_r - x_q(r)) - sum_{i: p(i) = r} ca[i] (x_i - x_r) - sum_{i: q(i) = r} cn[i] (x_i - x_r)
// grid = n rows, threads over the feature dim; a gather over the anchors (no atomics).
__global__ __launch_bounds__(256) void triplet_grad_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx_ap,
                                                           const int32_t* __restrict__ idx_an, const float* __restrict__ ca,
                                                           const float* __restrict__ cn, int n, int d, float* __restrict__ grad) {
    const int r = blockIdx.x;
    const float* xr = x + (size_t)r * d;
    const int pr = idx_ap[r], qr = idx_an[r];
    const float car = ca[r], cnr = cn[r];
    for (int c = threadIdx.x; c < d; c += 256) {
        const float v = xr[c];
        float g = 0.f;
        if (pr >= 0) g = fmaf(car, v - x[(size_t)pr * d + c], g);
        if (qr >= 0) g = fmaf(cnr, v - x[(size_t)qr * d + c], g);
        for (int i = 0; i < n; ++i) {
            const float xi = x[(size_t)i * d + c];
            if (idx_ap[i] == r) g = fmaf(-ca[i], xi - v, g);
            if (idx_an[i] == r) g = fmaf(-cn[i], xi - v, g);
        }
        grad[(size_t)r * d + c] = g;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// market1501 protocol (torchreid/metrics/rank.py:95-150; Cython twin rank_cylib/rank_cy.pyx:154-241) WITHOUT sorting
// the row. The reference argsorts all n distances of a query and walks the ranking; but AP and CMC only depend on the
// RANKS OF THE CORRECT MATCHES among the kept entries, and
//     rank(r) = 1 + #{ kept j : (d_j, j) < (d_r, r) }           (stable order: ties -> lower gallery index)
// is a counting pass. One workgroup per query: collect the (few) correct matches, count each one's rank with one sweep
// of the row per match (a wavefront per match, wave_sum), sort the <= MAXREL ranks, then
//     AP = sum_k k / rank_(k) / n_rel   (fp64, ascending k)      CMC[t] = 1 for t >= rank_(1) - 1.
// "kept" = not (same identity AND same camera as the query); a query without any kept match is invalid (valid = 0).
constexpr int MK_MAXREL = 4096;

__global__ __launch_bounds__(256) void rank_market1501_kernel(const float* __restrict__ dist, int n, int ldd,
                                                              const int32_t* __restrict__ q_pids, const int32_t* __restrict__ q_camids,
                                                              const int32_t* __restrict__ g_pids, const int32_t* __restrict__ g_camids,
                                                              int max_rank, double* __restrict__ ap, float* __restrict__ cmc,
                                                              int32_t* __restrict__ valid) {
    __shared__ int s_rel[MK_MAXREL];   // gallery indices of the correct matches, then their 1-based ranks
    __shared__ int s_nrel;
    const int q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = dist + (size_t)q * ldd;
    const int qp = q_pids[q], qc = q_camids[q];
    if (tid == 0) s_nrel = 0;
    __syncthreads();
    for (int j = tid; j < n; j += 256) {
        if (g_pids[j] == qp && g_camids[j] != qc) {
            const int slot = atomicAdd(&s_nrel, 1);
            if (slot < MK_MAXREL) s_rel[slot] = j;
        }
    }
    __syncthreads();
    const int nrel = s_nrel;
    float* crow = cmc + (size_t)q * max_rank;
    if (nrel == 0 || nrel > MK_MAXREL) {  // identity absent from the kept gallery (or more matches than the kernel holds)
        for (int t = tid; t < max_rank; t += 256) crow[t] = 0.f;
        if (tid == 0) {
            ap[q] = nan("");
            valid[q] = nrel == 0 ? 0 : -1;
        }
        return;
    }
    // rank of every correct match: a wavefront per match
    __shared__ int s_rank[MK_MAXREL];
    for (int r = wave; r < nrel; r += 4) {
        const int jr = s_rel[r];
        const float dr = row[jr];
        int cnt = 0;
        for (int j = lane; j < n; j += 64) {
            const bool kept = !(g_pids[j] == qp && g_camids[j] == qc);
            const float dj = row[j];
            cnt += (kept && (dj < dr || (dj == dr && j < jr))) ? 1 : 0;
        }
        cnt = wave_sum_i(cnt);
        if (lane == 0) s_rank[r] = cnt + 1;
    }
    __syncthreads();
    // sort the ranks ascending: rank order statistics by counting (nrel is small; ranks are distinct)
    for (int r = tid; r < nrel; r += 256) {
        const int mine = s_rank[r];
        int pos = 0;
        for (int o = 0; o < nrel; ++o) pos += s_rank[o] < mine ? 1 : 0;
        s_rel[pos] = mine;
    }
    __syncthreads();
    const int first = s_rel[0];
    for (int t = tid; t < max_rank; t += 256) crow[t] = t >= first - 1 ? 1.f : 0.f;
    if (tid == 0) {
        double acc = 0.0;
        for (int k = 0; k < nrel; ++k) acc += (double)(k + 1) / (double)s_rel[k];
        ap[q] = acc / (double)nrel;
        valid[q] = 1;
    }
}

}  // namespace

static int launch_topk(const float* dist, int m, int n, int ldd, int k, int idx_offset, int32_t* idx, float* val, hipStream_t st) {
    int kpad = 2;
    while (kpad < k) kpad <<= 1;
    const bool aligned = ((uintptr_t)dist % 16 == 0) && (ldd % 4 == 0);
    const bool force_radix = agrl_opts().topk_radix != 0;   // AGRL_TOPK_RADIX=1 (A/B switch): the 5-pass radix kernel for every row
    if (aligned && k <= 128 && n <= 32768 && !force_radix) {
        const int nv = (n + 1023) / 1024;
#define AGRL_TOPK_FAST(NV) hipLaunchKernelGGL(rank_topk_fast_kernel<NV>, dim3(m), dim3(256), 0, st, dist, n, ldd, k, kpad, idx_offset, idx, val)
        if (nv <= 1) AGRL_TOPK_FAST(1);
        else if (nv <= 2) AGRL_TOPK_FAST(2);
        else if (nv <= 4) AGRL_TOPK_FAST(4);
        else if (nv <= 8) AGRL_TOPK_FAST(8);
        else if (nv <= 12) AGRL_TOPK_FAST(12);
        else if (nv <= 16) AGRL_TOPK_FAST(16);
        else if (nv <= 24) AGRL_TOPK_FAST(24);
        else AGRL_TOPK_FAST(32);
#undef AGRL_TOPK_FAST
    } else {
        hipLaunchKernelGGL(rank_topk_kernel, dim3(m), dim3(256), (size_t)kpad * 8, st, dist, n, ldd, k, kpad, idx_offset, idx, val);
    }
    return 0;
}

extern "C" int agrl_rank_topk(const float* dist, int m, int n, int ldd, int k, int idx_offset, int32_t* idx, float* val,
                              agrl_stream_t stream) {
    AGRL_CHECK_ARG(dist && idx && val, "agrl_rank_topk: null pointer");
    AGRL_CHECK_ARG(m > 0 && n > 0 && ldd >= n, "agrl_rank_topk: bad shape m=%d n=%d ldd=%d", m, n, ldd);
    AGRL_CHECK_ARG(k > 0 && k <= n && k <= 1024, "agrl_rank_topk: need 0 < k <= min(n, 1024), got k=%d n=%d", k, n);
    launch_topk(dist, m, n, ldd, k, idx_offset, idx, val, (hipStream_t)stream);
    AGRL_CHECK_LAUNCH("agrl_rank_topk");
    return 0;
}

// Distance matrix + per-row top-k WITHOUT the m x n matrix (agrl_distmat_topk; reference
// metrics/distance.py:59-89 followed by metrics/rank.py:171-172): the queries are walked in row blocks whose distance rows
// (block x n fp32) live in a caller-provided workspace that is reused block after block -- small enough to stay in the
// 256 MB memory-side cache between the GEMM that writes it and the top-k that reads it -- so HBM sees the operands and the
// (m, k) lists, not 4 m n bytes out and back. Arithmetic and results are those of agrl_distmat + agrl_rank_topk, bit for bit.
extern "C" size_t agrl_distmat_topk_workspace(int m, int n) {
    if (m <= 0 || n <= 0) return 0;
    const int ldd = (n + 3) & ~3;
    int rows = (int)((size_t)(64u << 20) / ((size_t)ldd * 4));   // ~64 MB of distance rows per block (1280 rows at n = 12180) ...
    rows = rows / 128 * 128;
    if (rows < 256) rows = 256;                                  // ... at least two rows of GEMM tiles
    if (rows > m) rows = m;
    return (size_t)rows * ldd * 4;
}

extern "C" int agrl_distmat_topk(const void* q, const void* g, const float* qn, const float* gn, int m, int n, int D, int metric,
                                 int dtype, int k, int idx_offset, int32_t* idx, float* val, void* workspace, size_t workspace_bytes,
                                 void* gemm_workspace, size_t gemm_workspace_bytes, agrl_stream_t stream) {
    AGRL_CHECK_ARG(q && g && idx && val && workspace, "agrl_distmat_topk: null pointer");
    AGRL_CHECK_ARG(m > 0 && n > 0 && D > 0 && k > 0 && k <= n && k <= 1024, "agrl_distmat_topk: bad shape m=%d n=%d D=%d k=%d", m, n, D, k);
    const int ldd = (n + 3) & ~3;
    AGRL_CHECK_ARG((uintptr_t)workspace % 16 == 0 && workspace_bytes >= (size_t)ldd * 4,
                   "agrl_distmat_topk: workspace must be 16-byte aligned and hold at least one distance row (%zu bytes)", (size_t)ldd * 4);
    // Block size: the whole of m when it fits; else equal blocks of whole 128-row GEMM tiles. Every block goes through the same
    // kernel family (a tail of <= 64 rows would take the streaming form, whose k-order differs in the last bit): such a tail is
    // grown by taking 128 rows from the block before it.
    const int cap = (int)(workspace_bytes / ((size_t)ldd * 4) < (size_t)m ? workspace_bytes / ((size_t)ldd * 4) : (size_t)m);
    int rows = cap;
    if (cap < m && cap >= 128) {
        rows = cap / 128 * 128;
        const int nblk = (m + rows - 1) / rows;
        const int even = ((m + nblk - 1) / nblk + 127) / 128 * 128;
        if (even < rows) rows = even;
    }
    const size_t esz = dtype == AGRL_LP16 ? 2 : 4;
    for (int r0 = 0, mb = 0; r0 < m; r0 += mb) {
        mb = m - r0 < rows ? m - r0 : rows;
        const int rest = m - r0 - mb;
        if (rest > 0 && rest <= 64 && mb >= 256) mb -= 128;
        const int rc = agrl_distmat((const char*)q + (size_t)r0 * D * esz, g, qn ? qn + r0 : nullptr, gn, (float*)workspace, mb, n, D, ldd,
                                    metric, dtype, gemm_workspace, gemm_workspace_bytes, stream);
        if (rc != 0) return rc;
        launch_topk((const float*)workspace, mb, n, ldd, k, idx_offset, idx + (size_t)r0 * k, val + (size_t)r0 * k, (hipStream_t)stream);
    }
    AGRL_CHECK_LAUNCH("agrl_distmat_topk");
    return 0;
}

extern "C" int agrl_rank_argsort(const float* dist, int m, int n, int ldd, int32_t* idx, agrl_stream_t stream) {
    AGRL_CHECK_ARG(dist && idx && m > 0 && n > 0 && ldd >= n, "agrl_rank_argsort: bad arguments");
    AGRL_CHECK_ARG(n <= 16384, "agrl_rank_argsort: n=%d > 16384 (one row of composites must fit the LDS)", n);
    int npad = 2;
    while (npad < n) npad <<= 1;
    const size_t lds = (size_t)npad * 8;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)rank_argsort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        AGRL_CHECK_ARG(e == hipSuccess, "agrl_rank_argsort: cannot raise dynamic LDS: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(rank_argsort_kernel, dim3(m), dim3(1024), lds, (hipStream_t)stream, dist, n, ldd, npad, idx);
    AGRL_CHECK_LAUNCH("agrl_rank_argsort");
    return 0;
}

extern "C" int agrl_rank_mars(const int32_t* topk_idx, const int32_t* q_pids, const int32_t* q_camids,
                              const int32_t* g_pids, const int32_t* g_camids, int m, int n, int k, double* ap,
                              float* cmc, agrl_stream_t stream) {
    AGRL_CHECK_ARG(topk_idx && q_pids && q_camids && g_pids && g_camids && ap && cmc, "agrl_rank_mars: null pointer");
    AGRL_CHECK_ARG(m > 0 && n > 0 && k > 0, "agrl_rank_mars: bad shape");
    hipLaunchKernelGGL(rank_mars_kernel, dim3(m), dim3(256), 0, (hipStream_t)stream, topk_idx, q_pids, q_camids, g_pids,
                       g_camids, n, k, ap, cmc);
    AGRL_CHECK_LAUNCH("agrl_rank_mars");
    return 0;
}

extern "C" int agrl_triplet_hard_mine(const float* x, const int32_t* pids, int n, int d, float* dist_ap, float* dist_an,
                                      int32_t* idx_ap, int32_t* idx_an, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && pids && dist_ap && dist_an && idx_ap && idx_an, "agrl_triplet_hard_mine: null pointer");
    AGRL_CHECK_ARG(n > 0 && d > 0, "agrl_triplet_hard_mine: bad shape");
    hipLaunchKernelGGL(triplet_mine_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, x, pids, n, d, dist_ap, dist_an,
                       idx_ap, idx_an);
    AGRL_CHECK_LAUNCH("agrl_triplet_hard_mine");
    return 0;
}

extern "C" int agrl_triplet_loss(const float* x, const int32_t* pids, int n, int d, float margin, int soft, float* loss, float* grad,
                                 float* dist_ap, float* dist_an, int32_t* idx_ap, int32_t* idx_an, float* coeff, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && pids && loss && grad && dist_ap && dist_an && idx_ap && idx_an && coeff, "agrl_triplet_loss: null pointer");
    AGRL_CHECK_ARG(n > 0 && d > 0, "agrl_triplet_loss: bad shape");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(triplet_mine_kernel, dim3(n), dim3(256), 0, st, x, pids, n, d, dist_ap, dist_an, idx_ap, idx_an);
    hipLaunchKernelGGL(triplet_coeff_kernel, dim3(1), dim3(256), 0, st, dist_ap, dist_an, idx_an, n, margin, soft, loss, coeff, coeff + n);
    hipLaunchKernelGGL(triplet_grad_kernel, dim3(n), dim3(256), 0, st, x, idx_ap, idx_an, coeff, coeff + n, n, d, grad);
    AGRL_CHECK_LAUNCH("agrl_triplet_loss");
    return 0;
}

extern "C" int agrl_rank_market1501(const float* dist, int m, int n, int ldd, const int32_t* q_pids, const int32_t* q_camids,
                                    const int32_t* g_pids, const int32_t* g_camids, int max_rank, double* ap, float* cmc,
                                    int32_t* valid, agrl_stream_t stream) {
    AGRL_CHECK_ARG(dist && q_pids && q_camids && g_pids && g_camids && ap && cmc && valid, "agrl_rank_market1501: null pointer");
    AGRL_CHECK_ARG(m > 0 && n > 0 && ldd >= n && max_rank > 0 && max_rank <= n, "agrl_rank_market1501: bad shape m=%d n=%d ldd=%d max_rank=%d",
                   m, n, ldd, max_rank);
    hipLaunchKernelGGL(rank_market1501_kernel, dim3(m), dim3(256), 0, (hipStream_t)stream, dist, n, ldd, q_pids, q_camids, g_pids,
                       g_camids, max_rank, ap, cmc, valid);
    AGRL_CHECK_LAUNCH("agrl_rank_market1501");
    return 0;
}
