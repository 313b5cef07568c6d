// LDS counting sort shared by the gather-form scatters (mesh corner slots by vertex, Chamfer
// nearest-neighbour indices by target point): slots are grouped by key with LDS integer atomics,
// and consumers walk a bucket in ascending slot order, so float sums over a bucket have a fixed
// order although the sort itself does not.
#ifndef SV3D_BUCKET_SORT_H
#define SV3D_BUCKET_SORT_H

#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

namespace sv3d_dev {

constexpr int kSortThreads = 1024;

// Counting sort of `nslots` slots by key fb[slot] (clamped to [0, V)): on return
// bucket[start[v] .. fill[v]) holds the slots with key v, in arbitrary order.  All kSortThreads
// threads of the workgroup call it; fill / start hold V ints each, bucket nslots ints,
// s_wave kSortThreads / 64 ints.
__device__ inline void build_buckets(const int32_t *__restrict__ fb, int V, int nslots, int *fill,
                                     int *start, int *bucket, int *s_wave)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int v = tid; v < V; v += kSortThreads) fill[v] = 0;
    __syncthreads();
    for (int i = tid; i < nslots; i += kSortThreads) atomicAdd(&fill[min(max(fb[i], 0), V - 1)], 1);
    __syncthreads();
    // exclusive scan of the counts; thread t owns entries [t*per, (t+1)*per)
    const int per = (V + kSortThreads - 1) / kSortThreads;
    const int lo = min(tid * per, V), hi = min(lo + per, V);
    int local = 0;
    for (int v = lo; v < hi; ++v) local += fill[v];
    int incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int n = __shfl_up(incl, d);
        if (lane >= d) incl += n;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int run = incl - local;
    for (int w = 0; w < wave; ++w) run += s_wave[w];
    for (int v = lo; v < hi; ++v) {
        const int c = fill[v];
        start[v] = run;
        fill[v] = run;
        run += c;
    }
    __syncthreads();
    for (int i = tid; i < nslots; i += kSortThreads) bucket[atomicAdd(&fill[min(max(fb[i], 0), V - 1)], 1)] = i;
    __syncthreads();
}

// Next slot of bucket [s0, s1) above `last`, so that a bucket is walked in ascending slot order
// whatever order the sort left it in.
__device__ inline int next_slot(const int *bucket, int s0, int s1, int last)
{
    int m = INT_MAX;
    for (int j = s0; j < s1; ++j) {
        const int s = bucket[j];
        m = (s > last && s < m) ? s : m;
    }
    return m;
}

// Calls fn(slot) for the slots of bucket [s0, s1) -- those with key `key` -- in ascending slot
// order.  Short buckets are walked by repeated minimum search (k^2 steps for k slots); a bucket so
// long that k^2 exceeds the number of slots is walked by scanning the keys of all slots instead,
// which bounds a thread's work by nslots whatever the keys are (a vertex shared by every face,
// a cloud collapsed onto one point).
template <typename Fn>
__device__ inline void for_each_slot_ascending(const int *bucket, int s0, int s1, const int32_t *__restrict__ fb,
                                               int key, int V, int nslots, Fn fn)
{
    const int k = s1 - s0;
    if ((long)k * k <= (long)nslots) {
        int last = -1;
        for (int i = s0; i < s1; ++i) {
            last = next_slot(bucket, s0, s1, last);
            fn(last);
        }
    } else {
        for (int i = 0; i < nslots; ++i)
            if (min(max(fb[i], 0), V - 1) == key) fn(i);
    }
}

}  // namespace sv3d_dev
#endif
