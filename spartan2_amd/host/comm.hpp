// The exchange layer of the multi-GPU drivers: one process per GPU, and ONE primitive — an all-gather of a small fixed-size
// record per rank — because nothing this path exchanges is an elementwise reduction RCCL knows: field sums are modular, group sums are elliptic-curve
// additions. Every rank gathers everyone's record and combines locally in rank order (exact arithmetic: all ranks obtain identical values and run
// the deterministic transcript redundantly, so nothing is ever broadcast).
//   * RCCL backend (production: xGMI between the GPUs of a node): ncclAllGather on a stream of its own. Small records (the 64 - 96 bytes of a
//     sum-check round: the latency case) take the copy-free form: the host writes the record straight into the device send buffer through the PCIe
//     BAR (fine-grained device memory, as the challenge mailbox does), ncclAllGather runs on it, and a one-block kernel behind it on the same stream
//     publishes the gathered records into a self-validating slot in mapped host memory that the host polls — no hipMemcpyAsync pair, no stream
//     synchronise. Larger records keep pinned-host staging copies. librccl is resolved with dlopen at first use, so the one copy already in the
//     process (torch ships its own) is shared.
//   * callback backend (tests: gloo ranks that share one GPU; RCCL refuses two ranks on one device).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstring>
#include <string>
#include <vector>

#include "host_common.hpp"

namespace spartan2 {

// gathered records -> mapped host memory, then the tag (sequence number + check word) the host polls for; bytes is a multiple of 8
static __global__ void __launch_bounds__(256) k_comm_publish(const unsigned long long* __restrict__ src, size_t words, unsigned long long* __restrict__ dst, unsigned seq) {
  __shared__ unsigned long long part[256];
  unsigned long long acc = 0;
  for (size_t i = threadIdx.x; i < words; i += blockDim.x) {
    const unsigned long long v = src[i];
    dst[2 + i] = v;
    acc += v * (2 * i + 1);
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    __hip_atomic_store(dst + 1, part[0] + seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dst, (unsigned long long)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

typedef int (*ssc_allgather_fn)(void* user, const void* send, size_t bytes_per_rank, void* recv);

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  static RcclApi& get() {
    static RcclApi api = [] {
      RcclApi a;
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        a.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (a.handle) break;
      }
      if (!a.handle) return a;
      a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.handle, "ncclGetUniqueId");
      a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.handle, "ncclCommInitRank");
      a.AllGather = (decltype(a.AllGather))dlsym(a.handle, "ncclAllGather");
      a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.handle, "ncclCommDestroy");
      a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.handle, "ncclGetErrorString");
      return a;
    }();
    return api;
  }
  bool ok() const { return handle && GetUniqueId && CommInitRank && AllGather && CommDestroy && GetErrorString; }
};

struct Comm {
  int rank = 0, world = 1;
  // callback backend
  ssc_allgather_fn fn = nullptr;
  void* user = nullptr;
  // RCCL backend
  ncclComm_t nc = nullptr;
  int device = 0;
  hipStream_t st = nullptr;
  void *d_send = nullptr, *d_recv = nullptr, *h_send = nullptr, *h_recv = nullptr;
  size_t cap = 0;  // bytes per rank the staging buffers hold
  uint64_t calls = 0, bytes_moved = 0;
  // small-record path (<= SMALL_MAX bytes per rank)
  static constexpr size_t SMALL_MAX = 16384;  // (hipMemcpyAsync of a few KiB is the slowest size class of the staging path: 94 - 108 us at 4 KiB)
  void* d_small_send = nullptr;           // fine-grained device memory; h_small_send = the same bytes through the BAR (nullptr: no large BAR)
  volatile uint8_t* h_small_send = nullptr;
  void* d_small_recv = nullptr;           // world * SMALL_MAX
  unsigned long long* h_slot = nullptr;   // mapped pinned: [seq, check, words...]
  unsigned long long* d_slot = nullptr;
  unsigned small_seq = 0;
  bool small_ready = false, small_failed = false;
  uint64_t small_calls = 0;

  ~Comm() {
    if (nc) RcclApi::get().CommDestroy(nc);
    if (d_small_send) (void)hipFree(d_small_send);
    if (d_small_recv) (void)hipFree(d_small_recv);
    if (h_slot) (void)hipHostFree(h_slot);
    if (d_send) (void)hipFree(d_send);
    if (d_recv) (void)hipFree(d_recv);
    if (h_send) (void)hipHostFree(h_send);
    if (h_recv) (void)hipHostFree(h_recv);
    if (st) (void)hipStreamDestroy(st);
  }
  void hipck(hipError_t e, const char* what) {
    if (e != hipSuccess) throw Error(SP_ERR_NO_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
  }
  void reserve(size_t bytes) {
    if (bytes <= cap) return;
    size_t want = bytes < 4096 ? 4096 : bytes + bytes / 2;
    cap = 0;  // until all four allocations below have succeeded the buffers are unusable (a throw in between must not leave a stale capacity)
    if (d_send) (void)hipFree(d_send);
    if (d_recv) (void)hipFree(d_recv);
    if (h_send) (void)hipHostFree(h_send);
    if (h_recv) (void)hipHostFree(h_recv);
    d_send = d_recv = h_send = h_recv = nullptr;
    hipck(hipMalloc(&d_send, want), "comm: hipMalloc");
    hipck(hipMalloc(&d_recv, want * world), "comm: hipMalloc");
    hipck(hipHostMalloc(&h_send, want), "comm: hipHostMalloc");
    hipck(hipHostMalloc(&h_recv, want * world), "comm: hipHostMalloc");
    cap = want;
  }
  bool small_setup() {
    if (small_ready) return true;
    if (small_failed) return false;
    int large_bar = 0;
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device) != hipSuccess || !large_bar ||
        hipExtMallocWithFlags(&d_small_send, SMALL_MAX, hipDeviceMallocFinegrained) != hipSuccess) {
      small_failed = true;
      return false;
    }
    h_small_send = reinterpret_cast<volatile uint8_t*>(d_small_send);
    const size_t slot_bytes = 16 + SMALL_MAX * (size_t)world;
    if (hipMalloc(&d_small_recv, SMALL_MAX * (size_t)world) != hipSuccess || hipHostMalloc((void**)&h_slot, slot_bytes, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&d_slot, h_slot, 0) != hipSuccess) {
      small_failed = true;
      return false;
    }
    memset(h_slot, 0, slot_bytes);
    (void)hipMemset(d_small_send, 0, SMALL_MAX);
    (void)hipDeviceSynchronize();
    small_ready = true;
    return true;
  }
  // recv[r * bytes .. (r + 1) * bytes) = rank r's `send`
  void allgather(const void* send, size_t bytes, void* recv) {
    ++calls;
    bytes_moved += bytes * world;
    if (fn) {
      int rc = fn(user, send, bytes, recv);
      if (rc) throw Error(SP_ERR_INTERNAL, "comm: the all-gather callback failed");
      return;
    }
    if (!nc) {  // a world of one without a backend
      if (world != 1) throw Error(SP_ERR_INTERNAL, "comm: no backend");
      memcpy(recv, send, bytes);
      return;
    }
    hipck(hipSetDevice(device), "comm: hipSetDevice");
    if (bytes <= SMALL_MAX && bytes % 8 == 0 && small_setup()) {
      // record -> device send buffer through the BAR (write-combining: the fence pushes it out ahead of the launch's doorbell)
      for (size_t i = 0; i < bytes; i += 8) *reinterpret_cast<volatile uint64_t*>(h_small_send + i) = *reinterpret_cast<const uint64_t*>((const uint8_t*)send + i);
      __builtin_ia32_sfence();
      ncclResult_t r = RcclApi::get().AllGather(d_small_send, d_small_recv, bytes, ncclUint8, nc, st);
      if (r != ncclSuccess) throw Error(SP_ERR_INTERNAL, std::string("ncclAllGather: ") + RcclApi::get().GetErrorString(r));
      if (++small_seq == 0) ++small_seq;
      const size_t words = bytes * (size_t)world / 8;
      hipLaunchKernelGGL(k_comm_publish, dim3(1), dim3(256), 0, st, reinterpret_cast<const unsigned long long*>(d_small_recv), words, d_slot, small_seq);
      ++small_calls;
      volatile unsigned long long* slot = h_slot;
      bool synced = false;
      for (long spins = 0;; ++spins) {
        if (slot[0] == small_seq) {
          std::atomic_thread_fence(std::memory_order_acquire);
          unsigned long long chk = small_seq;
          for (size_t i = 0; i < words; ++i) chk += slot[2 + i] * (2 * i + 1);
          if (slot[1] == chk && slot[0] == small_seq) break;
        }
        if (spins > 2000000) {  // seconds: a peer is late, or a profiler serialises the stream
          if (synced) throw Error(SP_ERR_INTERNAL, "comm: the gathered record did not arrive");
          hipck(hipStreamSynchronize(st), "comm: synchronize");
          synced = true;
          spins = 0;
        }
        sp_relax();
      }
      for (size_t i = 0; i < words; ++i) reinterpret_cast<uint64_t*>(recv)[i] = slot[2 + i];
      return;
    }
    reserve(bytes);
    memcpy(h_send, send, bytes);
    hipck(hipMemcpyAsync(d_send, h_send, bytes, hipMemcpyHostToDevice, st), "comm: H2D");
    ncclResult_t r = RcclApi::get().AllGather(d_send, d_recv, bytes, ncclUint8, nc, st);
    if (r != ncclSuccess) throw Error(SP_ERR_INTERNAL, std::string("ncclAllGather: ") + RcclApi::get().GetErrorString(r));
    hipck(hipMemcpyAsync(h_recv, d_recv, bytes * world, hipMemcpyDeviceToHost, st), "comm: D2H");
    hipck(hipStreamSynchronize(st), "comm: synchronize");
    memcpy(recv, h_recv, bytes * world);
  }
  // The same for buffers that live in device memory (the layer hand-off of the sharded NIFS rounds: tens of MiB per rank): RCCL gathers straight
  // from / into them over xGMI; the callback backend stages through the host. The caller has synchronised the stream that produced d_src, and
  // the call returns with d_dst complete. d_dst must not overlap d_src.
  void allgather_device(const void* d_src, size_t bytes, void* d_dst) {
    ++calls;
    bytes_moved += bytes * world;
    if (fn) {
      std::vector<uint8_t> hs(bytes), hr(bytes * world);
      hipck(hipMemcpy(hs.data(), d_src, bytes, hipMemcpyDeviceToHost), "comm: D2H");
      int rc = fn(user, hs.data(), bytes, hr.data());
      if (rc) throw Error(SP_ERR_INTERNAL, "comm: the all-gather callback failed");
      hipck(hipMemcpy(d_dst, hr.data(), bytes * world, hipMemcpyHostToDevice), "comm: H2D");
      return;
    }
    if (!nc) {
      if (world != 1) throw Error(SP_ERR_INTERNAL, "comm: no backend");
      hipck(hipMemcpy(d_dst, d_src, bytes, hipMemcpyDeviceToDevice), "comm: D2D");
      return;
    }
    hipck(hipSetDevice(device), "comm: hipSetDevice");
    ncclResult_t r = RcclApi::get().AllGather(d_src, d_dst, bytes, ncclUint8, nc, st);
    if (r != ncclSuccess) throw Error(SP_ERR_INTERNAL, std::string("ncclAllGather: ") + RcclApi::get().GetErrorString(r));
    hipck(hipStreamSynchronize(st), "comm: synchronize");
  }
  // field sum over ranks of `count` elements, in rank order, in place (the reduce step of a slice-sharded sum-check round)
  void field_sum(fe_t* vals, size_t count) {
    if (world == 1) return;
    std::vector<fe_t> all(count * world);
    allgather(vals, count * sizeof(fe_t), all.data());
    for (size_t i = 0; i < count; ++i) {
      fe_t acc = all[i];
      for (int r = 1; r < world; ++r) acc = fe_add<S>(acc, all[(size_t)r * count + i]);
      vals[i] = acc;
    }
  }
};

}  // namespace spartan2
