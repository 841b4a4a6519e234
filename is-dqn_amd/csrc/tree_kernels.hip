// Float64 sum-tree on the device: bit-exact restatement of the reference's numpy SumTree
// (slimdqn/sample_collection/sum_tree.py).  HBM-resident node array (2^depth - 1 doubles;
// 16.8 MB at capacity 1e6, i.e. Infinity-Cache resident), integer/f64 latency-bound work.
//
// Why the set kernel looks the way it does: the reference never recomputes a parent from
// its children; it adds each leaf's delta to every ancestor with np.add.at, in ascending
// leaf order (sum_tree.py:33-47).  Float64 addition does not associate, so the value of a
// shared ancestor depends on that order.  A node's final value depends only on its old
// value and on the ordered list of deltas below it, so all levels are independent: one
// wave per level, one lane per run of equal ancestors, serial adds inside the run.
#include "common.h"

namespace isdqn {

constexpr int TREE_SET_THREADS = 1024;

__device__ __forceinline__ int ancestor(int node, int level) { return (int)(((unsigned)node + 1u) >> level) - 1; }

__global__ __launch_bounds__(TREE_SET_THREADS) void tree_set_kernel(double* __restrict__ nodes, int depth,
                                                                    const int* __restrict__ indices,
                                                                    const double* __restrict__ values, int n,
                                                                    double* max_prio, uint32_t* status) {
    __shared__ __attribute__((aligned(16))) int s_node[ISDQN_TREE_MAX_BATCH];
    __shared__ double s_delta[ISDQN_TREE_MAX_BATCH];
    __shared__ int u_node[ISDQN_TREE_MAX_BATCH];
    __shared__ double u_delta[ISDQN_TREE_MAX_BATCH];
    __shared__ unsigned char s_first[ISDQN_TREE_MAX_BATCH];
    __shared__ double s_wmax[TREE_SET_THREADS / 64];
    __shared__ int s_m;

    const int t = threadIdx.x;
    const int first_leaf = (1 << (depth - 1)) - 1;
    const int n_nodes = (1 << depth) - 1;

    // phase 0: validate, delta = value - leaf (sum_tree.py:30-34)
    int bad = 0;
    double vmax = 0.0;
    for (int i = t; i < n; i += TREE_SET_THREADS) {
        double v = values[i];
        int node = first_leaf + indices[i];
        if (!(v >= 0.0)) bad |= 1;
        if (node < first_leaf || node >= n_nodes) {
            bad |= 2;
            node = first_leaf;
        }
        s_node[i] = node;
        s_delta[i] = v - nodes[node];
        vmax = v > vmax ? v : vmax;
    }
    if (t == 0) s_m = 0;
    int any_bad = __syncthreads_or(bad);
    if (any_bad) {  // the reference asserts before touching the tree
        if (t == 0) atomicOr(status, ISDQN_STATUS_NEGATIVE_VALUE);
        return;
    }
    if (max_prio != nullptr) {
        for (int off = 32; off > 0; off >>= 1) {
            double o = __shfl_xor(vmax, off);
            vmax = o > vmax ? o : vmax;
        }
        if ((t & 63) == 0) s_wmax[t >> 6] = vmax;
        __syncthreads();
        if (t == 0) {
            double m = *max_prio;
            for (int w = 0; w < TREE_SET_THREADS / 64; ++w) m = s_wmax[w] > m ? s_wmax[w] : m;
            *max_prio = m;
        }
    }

    // phase 1: np.unique(node, return_index=True): first occurrence of each node, nodes ascending
    for (int i = t; i < n; i += TREE_SET_THREADS) {
        int node = s_node[i];
        int first = 1;
        for (int j = 0; j < i; ++j) first &= (s_node[j] != node);
        s_first[i] = (unsigned char)first;
    }
    __syncthreads();
    // rank of a first occurrence = number of first occurrences with a smaller node.  Later occurrences get the key
    // INT_MAX so that one array and a branch-free compare suffice: with `s_first[j] && s_node[j] < node` the loop is a
    // branch per element and its LDS reads cannot be pipelined (26 of the kernel's 44 us at n = 256).
    int my_node[ISDQN_TREE_MAX_BATCH / TREE_SET_THREADS];
#pragma unroll
    for (int k = 0; k < ISDQN_TREE_MAX_BATCH / TREE_SET_THREADS; ++k) {
        const int i = t + k * TREE_SET_THREADS;
        my_node[k] = i < n ? s_node[i] : 0;
    }
    __syncthreads();  // everybody has its node in a register before the keys overwrite s_node
#pragma unroll
    for (int k = 0; k < ISDQN_TREE_MAX_BATCH / TREE_SET_THREADS; ++k) {
        const int i = t + k * TREE_SET_THREADS;
        if (i < n && !s_first[i]) s_node[i] = 0x7fffffff;
    }
    for (int i = n + t; i < ((n + 3) & ~3); i += TREE_SET_THREADS) s_node[i] = 0x7fffffff;  // pad to a multiple of 4
    __syncthreads();
    const int n4 = (n + 3) >> 2;
#pragma unroll
    for (int k = 0; k < ISDQN_TREE_MAX_BATCH / TREE_SET_THREADS; ++k) {
        const int i = t + k * TREE_SET_THREADS;
        if (i >= n || !s_first[i]) continue;
        const int node = my_node[k];
        int rank = 0;
#pragma unroll 4
        for (int j4 = 0; j4 < n4; ++j4) {
            const int4 q = reinterpret_cast<const int4*>(s_node)[j4];  // same address in every lane: LDS broadcast
            rank += (q.x < node) + (q.y < node) + (q.z < node) + (q.w < node);
        }
        u_node[rank] = node;
        u_delta[rank] = s_delta[i];
        atomicAdd(&s_m, 1);
    }
    __syncthreads();
    const int m = s_m;

    // phase 2: every level independently; runs of equal ancestors added serially in leaf order
    const int wave = t >> 6, lane = t & 63;
    for (int level = wave; level < depth; level += TREE_SET_THREADS / 64) {
        for (int base = 0; base < m; base += 64) {
            int i = base + lane;
            if (i >= m) continue;
            int nd = ancestor(u_node[i], level);
            int prev = i > 0 ? ancestor(u_node[i - 1], level) : -1;
            if (nd != prev) {
                // the run of leaves below `nd` is u_node in [first_below, last_below]: find its end by binary
                // search (u_node is sorted), then add the deltas strictly left to right (np.add.at order)
                const long long limit = ((long long)(nd + 2) << level) - 1;  // first node index NOT below nd
                int lo = i + 1, hi = m;
                while (lo < hi) {
                    int mid = (lo + hi) >> 1;
                    if ((long long)u_node[mid] < limit) lo = mid + 1; else hi = mid;
                }
                const int end = lo;
                double acc = nodes[nd];
                int j = i;
                for (; j + 4 <= end; j += 4) {
                    const double d0 = u_delta[j], d1 = u_delta[j + 1], d2 = u_delta[j + 2], d3 = u_delta[j + 3];
                    acc = acc + d0;
                    acc = acc + d1;
                    acc = acc + d2;
                    acc = acc + d3;
                }
                for (; j < end; ++j) acc = acc + u_delta[j];
                nodes[nd] = acc;
            }
        }
    }
}

// samplers.py:89-103: set([index, last], [get(last), 0.0]) or set(index, 0.0); one lane per level.
__global__ __launch_bounds__(64) void tree_swap_remove_kernel(double* __restrict__ nodes, int depth, int index,
                                                              int last_index, uint32_t* status) {
    const int first_leaf = (1 << (depth - 1)) - 1;
    const int na = first_leaf + index, nb = first_leaf + last_index;
    double leaf_a = nodes[na], leaf_b = nodes[nb];
    double da, db;
    if (index == last_index) {
        da = 0.0 - leaf_a;
        db = 0.0;
    } else {
        da = leaf_b - leaf_a;  // new value of `index` is get(last_index)
        db = 0.0 - leaf_b;
    }
    // ascending node order
    int n0 = na, n1 = nb;
    double d0 = da, d1 = db;
    if (index != last_index && nb < na) {
        n0 = nb;
        n1 = na;
        d0 = db;
        d1 = da;
    }
    __syncthreads();  // all lanes have read the leaves before any level-0 store
    for (int level = threadIdx.x; level < depth; level += 64) {
        int a0 = ancestor(n0, level);
        double acc = nodes[a0] + d0;
        if (index != last_index) {
            int a1 = ancestor(n1, level);
            if (a1 == a0) {
                acc = acc + d1;
            } else {
                nodes[a1] = nodes[a1] + d1;
            }
        }
        nodes[a0] = acc;
    }
    (void)status;
}

__global__ __launch_bounds__(256) void tree_query_kernel(const double* __restrict__ nodes, int depth,
                                                         const double* __restrict__ targets, int n, int unit,
                                                         int* __restrict__ out, uint32_t* status) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double root = nodes[0];
    double tgt = targets[i];
    if (unit) tgt = 0.0 + root * tgt;  // numpy random_uniform: low + range * next_double
    if (!(tgt >= 0.0 && tgt < root)) atomicOr(status, root == 0.0 ? ISDQN_STATUS_EMPTY_TREE : ISDQN_STATUS_TARGET_RANGE);
    int node = 0;
    for (int level = 0; level < depth - 1; ++level) {
        int left = 2 * node + 1;
        double ls = nodes[left];
        bool go_left = tgt < ls;
        node = go_left ? left : left + 1;
        tgt = go_left ? tgt : tgt - ls;
    }
    out[i] = node - ((1 << (depth - 1)) - 1);
}

}  // namespace isdqn

using namespace isdqn;

extern "C" int isdqn_tree_layout(int64_t capacity, int32_t* depth, int64_t* first_leaf_offset, int64_t* n_nodes) {
    ISDQN_REQUIRE(capacity > 0, ISDQN_ERR_CAPACITY, "Capacity to sum tree must be positive.");
    ISDQN_REQUIRE(capacity <= (1ll << 29), ISDQN_ERR_CAPACITY, "capacity above 2^29 leaves is not supported");
    int d = 0;  // ceil(log2(capacity)) + 1
    while ((1ll << d) < capacity) ++d;
    d += 1;
    if (depth) *depth = d;
    if (first_leaf_offset) *first_leaf_offset = (1ll << (d - 1)) - 1;
    if (n_nodes) *n_nodes = (1ll << d) - 1;
    return ISDQN_OK;
}

extern "C" int isdqn_tree_set(double* nodes, int32_t depth, const int32_t* indices, const double* values, int32_t n,
                              double* max_recorded_priority, uint32_t* dev_status, void* stream) {
    ISDQN_REQUIRE(nodes && indices && values && dev_status, ISDQN_ERR_ARG, "null pointer");
    ISDQN_REQUIRE(depth >= 1 && depth <= 30, ISDQN_ERR_ARG, "bad depth");
    ISDQN_REQUIRE(n >= 0 && n <= ISDQN_TREE_MAX_BATCH, ISDQN_ERR_SHAPE, "batch larger than ISDQN_TREE_MAX_BATCH");
    if (n == 0) return ISDQN_OK;
    hipLaunchKernelGGL(tree_set_kernel, dim3(1), dim3(TREE_SET_THREADS), 0, (hipStream_t)stream, nodes, depth, indices,
                       values, n, max_recorded_priority, dev_status);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}

extern "C" int isdqn_tree_swap_remove(double* nodes, int32_t depth, int32_t index, int32_t last_index,
                                      uint32_t* dev_status, void* stream) {
    ISDQN_REQUIRE(nodes && dev_status, ISDQN_ERR_ARG, "null pointer");
    ISDQN_REQUIRE(depth >= 1 && depth <= 30, ISDQN_ERR_ARG, "bad depth");
    const int64_t leaves = 1ll << (depth - 1);
    ISDQN_REQUIRE(index >= 0 && last_index >= 0 && index < leaves && last_index < leaves, ISDQN_ERR_ARG,
                  "leaf index out of range");
    hipLaunchKernelGGL(tree_swap_remove_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, nodes, depth, index,
                       last_index, dev_status);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}

extern "C" int isdqn_tree_query(const double* nodes, int32_t depth, const double* targets, int32_t n,
                                int32_t targets_are_unit, int32_t* out_indices, uint32_t* dev_status, void* stream) {
    ISDQN_REQUIRE(nodes && targets && out_indices && dev_status, ISDQN_ERR_ARG, "null pointer");
    ISDQN_REQUIRE(depth >= 1 && depth <= 30, ISDQN_ERR_ARG, "bad depth");
    ISDQN_REQUIRE(n >= 0, ISDQN_ERR_SHAPE, "negative batch");
    if (n == 0) return ISDQN_OK;
    hipLaunchKernelGGL(tree_query_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, nodes, depth,
                       targets, n, targets_are_unit, out_indices, dev_status);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}
