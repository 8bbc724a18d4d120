// The three CUB block primitives the reference uses, with CUB's documented
// semantics, on top of dgemu's __syncthreads (TEST INFRASTRUCTURE ONLY, see
// dgemu.h).  Call sites: GpuANSStatistics.cuh:239-241,336-341,
// GpuANSDecode.cuh:439-445, BatchPrefixSum.cuh:51-56,80-83, GpuChecksum.cuh:85-88.
//   BlockRadixSort<K, THREADS, ITEMS>::SortDescending(keys): blocked arrangement in
//     and out (thread t owns ranks [t*ITEMS, (t+1)*ITEMS)), keys in descending order.
//   BlockScan<T, THREADS>::ExclusiveSum: exclusive prefix sum with initial value 0 over
//     the threads (and, for arrays, over each thread's consecutive items: blocked
//     arrangement); optional block-wide aggregate.
//   BlockReduce<T, THREADS>::Reduce(v, op): the reduction over all threads, valid in
//     thread 0 (returned to every thread here).
#pragma once
#include <algorithm>
#include <functional>
#include "../dgemu.h"

namespace cub {

template <typename KeyT, int BLOCK_THREADS, int ITEMS_PER_THREAD>
class BlockRadixSort {
 public:
  struct TempStorage {
    KeyT keys[BLOCK_THREADS * ITEMS_PER_THREAD];
  };
  explicit BlockRadixSort(TempStorage& t) : t_(t) {}
  void SortDescending(KeyT (&keys)[ITEMS_PER_THREAD]) {
    const int tid = (int)dgemu::cur->linear;
    for (int i = 0; i < ITEMS_PER_THREAD; ++i) t_.keys[tid * ITEMS_PER_THREAD + i] = keys[i];
    __syncthreads();
    if (tid == 0) std::stable_sort(t_.keys, t_.keys + BLOCK_THREADS * ITEMS_PER_THREAD, std::greater<KeyT>());
    __syncthreads();
    for (int i = 0; i < ITEMS_PER_THREAD; ++i) keys[i] = t_.keys[tid * ITEMS_PER_THREAD + i];
    __syncthreads();
  }

 private:
  TempStorage& t_;
};

template <typename T, int BLOCK_THREADS>
class BlockScan {
 public:
  struct TempStorage {
    T v[BLOCK_THREADS];
  };
  explicit BlockScan(TempStorage& t) : t_(t) {}
  void ExclusiveSum(T input, T& output) {
    T agg;
    ExclusiveSum(input, output, agg);
  }
  void ExclusiveSum(T input, T& output, T& block_aggregate) {
    const int tid = (int)dgemu::cur->linear;
    t_.v[tid] = input;
    __syncthreads();
    T before = 0, all = 0;
    for (int i = 0; i < BLOCK_THREADS; ++i) {
      if (i < tid) before += t_.v[i];
      all += t_.v[i];
    }
    __syncthreads();
    output = before;
    block_aggregate = all;
  }
  template <int N>
  void ExclusiveSum(T (&input)[N], T (&output)[N]) {
    T agg;
    ExclusiveSum(input, output, agg);
  }
  template <int N>
  void ExclusiveSum(T (&input)[N], T (&output)[N], T& block_aggregate) {
    T mine = 0;
    for (int i = 0; i < N; ++i) mine += input[i];
    T base;
    ExclusiveSum(mine, base, block_aggregate);
    for (int i = 0; i < N; ++i) {
      const T in = input[i];  // input and output may alias
      output[i] = base;
      base += in;
    }
  }

 private:
  TempStorage& t_;
};

template <typename T, int BLOCK_THREADS>
class BlockReduce {
 public:
  struct TempStorage {
    T v[BLOCK_THREADS];
  };
  explicit BlockReduce(TempStorage& t) : t_(t) {}
  template <typename Op>
  T Reduce(T input, Op op) {
    const int tid = (int)dgemu::cur->linear;
    t_.v[tid] = input;
    __syncthreads();
    T r = t_.v[0];
    for (int i = 1; i < BLOCK_THREADS; ++i) r = op(r, t_.v[i]);
    __syncthreads();
    return r;
  }
  T Sum(T input) {
    return Reduce(input, [](T a, T b) { return a + b; });
  }

 private:
  TempStorage& t_;
};

}  // namespace cub
