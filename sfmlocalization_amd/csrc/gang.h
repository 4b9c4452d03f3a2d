// Gang launches: several contexts of one map take one query each through the same chain of kernels, and every kernel
// of the chain is launched ONCE for all of them (gridDim.z = member), the members' argument lists side by side in the
// kernel arguments.  The arithmetic of a member is untouched -- a kernel's body is the same device function whether it
// is launched alone or in a gang -- only the number of launches and of busy streams changes: a rank of an 8-rank run has
// 1/8 of the scan per query but every query's launches, and a dependent launch costs the more the more hardware queues
// are busy (Concurrency; 6: a rank of 8 goes from 7.5 k to 15.0 k queries/s).
//
// How: between sfmloc_gang_begin and sfmloc_gang_end the members' launchers RECORD their launches (sfm_launch,
// sfmloc_internal.h) instead of issuing them; sfmloc_gang_end (gang_flush, capi.hip) walks the members' lists in step
// and issues, on the gang's stream (the first member's), per position ONE gang kernel for the members whose record is
// the same kernel on the same grid -- as many of them as the kernel arguments hold, the rest in further launches -- and a
// plain launch for a record without a partner.  Any other use of a member's stream while recording (a copy, an event)
// first issues what has been recorded, so the order on the stream is always the order of the calls.
// The five hot kernels keep their text in *.body.inc files included by the kernel and by its Body::run: compiled as an
// inlined call the scan's loop is unrolled and scheduled before the inlining and runs 20 % slower.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <new>
#include <type_traits>
#include <utility>
#include <vector>

namespace sfmloc {

constexpr int kGangMembers = 32;      // contexts per session at most
constexpr int kGangArgBytes = 2032;   // one member's argument list at most (the AKAZE level tables are 1 KB)
constexpr int kGangKernargBytes = 3968;  // the members' lists of one launch together (the kernarg segment holds 4 KB)

// a trivially copyable tuple (std::tuple is neither that nor usable in a kernel signature)
template <size_t I, class T>
struct TupLeaf {
  T v;
};
template <class Seq, class... Ts>
struct TupImpl;
template <size_t... Is, class... Ts>
struct TupImpl<std::index_sequence<Is...>, Ts...> : TupLeaf<Is, Ts>... {};
template <class... Ts>
using Tup = TupImpl<std::index_sequence_for<Ts...>, Ts...>;
template <size_t I, class T>
__host__ __device__ __forceinline__ const T &tup_get(const TupLeaf<I, T> &l) {
  return l.v;
}

// members one launch of a kernel with these arguments can carry: as many as fit the kernel arguments
template <class... Ts>
constexpr int gang_cap() {
  return (int)(kGangKernargBytes / sizeof(Tup<Ts...>)) < kGangMembers ? (int)(kGangKernargBytes / sizeof(Tup<Ts...>)) : kGangMembers;
}
template <int Cap, class... Ts>
struct GangArgs {
  Tup<Ts...> a[Cap];
};

template <class Body, class... Ts, size_t... Is>
__device__ __forceinline__ void gang_call(const Tup<Ts...> &t, std::index_sequence<Is...>) {
  Body::run(tup_get<Is>(t)...);
}

// Body: struct { static constexpr int kGangThreads; __device__ static void run(Ts...); } -- the kernel's body; optionally
// static constexpr int kGangMinWaves: waves per SIMD the gang form must leave room for (the second launch bound of the
// kernel the body belongs to, so that a session's launch occupies the registers the single launch does)
template <class Body, class = void>
struct GangMinWaves {
  static constexpr int value = 1;
};
template <class Body>
struct GangMinWaves<Body, std::void_t<decltype(Body::kGangMinWaves)>> {
  static constexpr int value = Body::kGangMinWaves;
};
template <class Body, int Cap, class... Ts>
__global__ __launch_bounds__(Body::kGangThreads, GangMinWaves<Body>::value) void k_gang(GangArgs<Cap, Ts...> g) {
  gang_call<Body, Ts...>(g.a[blockIdx.z], std::index_sequence_for<Ts...>{});
}

struct GangRec {
  const void *key;  // the gang kernel: same key = same body and argument types
  const void *single;
  void (*launch_one)(const GangRec &, hipStream_t);
  void (*launch_many)(GangRec *const *, int, hipStream_t);
  dim3 grid, block;
  uint32_t shmem;
  int cap;  // members one launch of this kernel carries
  alignas(16) unsigned char args[kGangArgBytes];
};

struct GangMember;
struct GangState {
  hipStream_t stream = nullptr;  // the first member's own stream
  hipEvent_t done = nullptr;
  std::vector<GangMember *> members;
  uint64_t launches = 0, gang_launches = 0;  // issued by flushes (all kinds / with more than one member)
};

int gang_flush(GangState *g);  // capi.hip

// A context's stream.  Reading it (every hipXxx(..., c->stream) of the launchers) gives the stream work has to be
// queued on NOW: the context's own, or -- while the context records for a gang -- the gang's, after everything
// recorded so far has been issued.
struct CtxStream {
  hipStream_t own = nullptr;
  GangState *gang = nullptr;  // non-null while recording
  bool dirty = false;         // work was queued on `own` since the host last waited for it
  operator hipStream_t() {
    if (gang) {
      gang_flush(gang);
      return gang->stream;
    }
    dirty = true;
    return own;
  }
  // the stream for work that depends on NOTHING this member has recorded in the session so far (an upload into a buffer
  // the recorded kernels do not touch before it): queued at once, ahead of the recorded launches, without issuing them
  hipStream_t unordered() {
    if (gang) return gang->stream;
    dirty = true;
    return own;
  }
};

// what takes part in gang sessions: a context (sfmloc_internal.h) or an AKAZE extractor (akaze.hip)
struct GangMember {
  CtxStream stream;  // reads as the stream to queue on now (the member's own, or its gang's while recording)
  std::vector<GangRec> gang_recs;  // launches recorded for the gang session in progress
  size_t gang_head = 0;
  bool gang_oom = false;            // a record could not be stored (host memory): the session's flush returns ENOMEM
  bool ever_ganged = false;         // has taken part in a session (a stream created for it LATER must wait for that work)
  GangState *gang_owned = nullptr;  // this member has led a gang: its state (stream = this member's own)
  hipEvent_t gang_ev = nullptr;     // orders the gang's stream after this member's own earlier work
};

// capi.hip.  gang_open: the members (the first leads: its stream carries the session) start recording; the gang's stream
// first waits for whatever a member still has queued on a stream of its own.  gang_close: everything recorded is issued
// and the members' own streams continue after it.  gang_member_free: the events / state a member may own.
int gang_open(GangMember *const *members, int n);
int gang_close(GangMember *lead);
void gang_member_free(GangMember *m);

template <class Body, class... Ts>
struct GangLaunch {
  using T = Tup<Ts...>;
  static_assert(sizeof(T) <= kGangArgBytes, "argument list too long for a gang record");
  static constexpr int kCap = gang_cap<Ts...>();
  static_assert(kCap >= 2, "argument list too long for a gang launch");
  static const void *key() { return reinterpret_cast<const void *>(&k_gang<Body, kCap, Ts...>); }
  template <size_t... Is>
  static void one_impl(const GangRec &r, hipStream_t s, std::index_sequence<Is...>) {
    const T &t = *reinterpret_cast<const T *>(r.args);
    auto k = reinterpret_cast<void (*)(Ts...)>(const_cast<void *>(r.single));
    hipLaunchKernelGGL(k, r.grid, r.block, r.shmem, s, tup_get<Is>(t)...);
  }
  static void one(const GangRec &r, hipStream_t s) { one_impl(r, s, std::index_sequence_for<Ts...>{}); }
  static void many(GangRec *const *rs, int n, hipStream_t s) {
    const GangRec &r = *rs[0];
    static std::atomic<uint32_t> lds_allowed{48 * 1024};  // (dynamic LDS beyond the default needs the attribute)
    if (r.shmem > lds_allowed.load(std::memory_order_relaxed)) {
      if (hipFuncSetAttribute(key(), hipFuncAttributeMaxDynamicSharedMemorySize, (int)r.shmem) != hipSuccess)
        return;  // (the error stays for the caller's hipGetLastError)
      lds_allowed.store(r.shmem, std::memory_order_relaxed);
    }
    GangArgs<kCap, Ts...> g;
    for (int i = 0; i < n; ++i) g.a[i] = *reinterpret_cast<const T *>(rs[i]->args);
    for (int i = n; i < kCap; ++i) g.a[i] = g.a[0];
    hipLaunchKernelGGL((k_gang<Body, kCap, Ts...>), dim3(r.grid.x, r.grid.y, n), r.block, r.shmem, s, g);
  }
};

}  // namespace sfmloc
