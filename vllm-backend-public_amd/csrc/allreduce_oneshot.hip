// One-shot all-reduce (sum) for the small messages of a tensor-parallel decode step
// ([tokens, hidden] 16-bit partial sums after o_proj / down_proj: 256 KiB - 1 MiB).
//
// Every rank owns a device buffer that its peers map through HIP IPC handles:
//   slots [2 parities][world ranks][max_bytes]   data, written by the peers (and by the owner)
//   flags [world ranks][kMaxBlocks]              last round whose data a peer's workgroup delivered
// A call is one kernel of nblocks workgroups on every rank.  Workgroup b of rank r
//   1. copies chunk b of its input into slot[parity][r] of EVERY rank's buffer (all links of a
//      fully connected node carry data at once -- a ring pushes the message through one link
//      per hop),
//   2. waits until its stores have been acknowledged (the buffers are uncached, the stores
//      system-scope: no fence, see below) and writes the round number to flag[r][b] of every rank,
//   3. waits until flag[p][b] of its own buffer has reached the round for every p,
//   4. sums chunk b of the world slots in RANK ORDER in fp32 and writes the rounded result
//      over its input -- every rank adds the same values in the same order, so the results
//      are bit-identical across ranks (the lock-step schedulers of the ranks rely on that).
// Synchronisation is per chunk: no grid-wide barrier, no host round trip, one launch.  The
// round number lives in device memory (read by every workgroup at its start, advanced by the
// last workgroup of the call to finish), so a call captured into a HIP graph advances it on
// every replay.  Slots alternate between two parities: a rank can run
// at most one call ahead of its slowest peer (it needs that peer's flags of the round), so data
// of round n+1 never lands in a slot a peer still reads for round n.
// The buffers are allocated uncached (fine-grained): data written by a peer over xGMI is read
// from memory, not from a stale line of the reader's L2.
// Every wait is bounded (kSpinLimitTicks of the constant 100 MHz clock): a peer that never
// arrives makes the kernel give up, raise the communicator's error word and leave its input
// untouched -- the grid always drains.
#include "common.h"

#include <string.h>

namespace mi355 {

constexpr int kArMaxWorld = 8;   // (ar_load_all / ar_wait_all name eight registers)
constexpr int kArMaxBlocks = 256;
constexpr int kArThreads = 256;
// 10 s of the 100 MHz wall clock: long against any skew between ranks that are alive (a rank that captures a
// graph or times GEMM forms for a new batch size arrives seconds late; ranks sharing one card in a rehearsal
// have been seen 2 s apart), short against a job that would otherwise hang for good
constexpr long long kSpinLimitTicks = 1000000000LL;

struct ArPeers {
  unsigned char* buf[kArMaxWorld];   // every rank's buffer as mapped into THIS process
};

struct ArComm {
  int rank, world, device;
  int64_t max_bytes;
  unsigned char* local;              // this rank's buffer (hipExtMallocWithFlags, uncached)
  unsigned char* peer[kArMaxWorld];  // peer[rank] == local
  bool opened[kArMaxWorld];
  unsigned* rounds;                  // device words: [0] last completed round, [1] finish ticket, [2] error
  unsigned* error;                   // = rounds + 2: != 0 after a wait timed out
  bool connected;
};

__host__ __device__ inline int64_t ar_flags_offset(int world, int64_t max_bytes) {
  return 2 * static_cast<int64_t>(world) * max_bytes;
}
inline int64_t ar_buffer_bytes(int world, int64_t max_bytes) {
  return ar_flags_offset(world, max_bytes) + static_cast<int64_t>(world) * kArMaxBlocks * sizeof(unsigned);
}

// Accesses to the exchange buffers.  The buffers are fine-grained (uncached) device memory and the
// peers map them as such: no cache level holds their lines, so a store is where its readers look
// once it has been acknowledged (the wave's vmcnt has drained), and a load finds what was last
// stored.  Every access carries system scope (sc0 sc1) on top, so that this does not rest on the
// mapping alone.  What the kernel does NOT do is fence: a system-scope release or acquire fence
// writes back / invalidates the whole L2 -- with the dirty fp32 slabs of the GEMM that ran just
// before it, several microseconds per call, 64-160 calls per step -- for data that is not in it.
typedef unsigned int ar_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void ar_store_sys(void* p, uint4 v) {
  const ar_u32x4 r = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ ar_u32x4 ar_load_sys(const void* p) {   // the result may be used only behind ar_wait_loads
  ar_u32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(r) : "v"(p) : "memory");
  return r;
}
__device__ __forceinline__ void ar_wait_loads(ar_u32x4& r) {       // ties the value to the wait: no use can move above it
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(r)::"memory");
}
// The loads of one vector from every possible rank's slot and the wait that covers them, in straight-line
// code: no branch lies between a load and its wait, so the compiler never has a reason to copy a destination
// register before the data is in it (a copy made at a join of branches would carry the register's OLD value).
// A rank that does not take part (p >= world, or p == skip) is asked for one fixed address; its value is unused.
__device__ __forceinline__ void ar_load_all(ar_u32x4 (&in)[8], const unsigned char* slots, int64_t slot_bytes, int64_t byte_off,
                                            int world, int skip) {
#pragma unroll
  for (int p = 0; p < 8; ++p)
    in[p] = ar_load_sys((p < world && p != skip) ? slots + p * slot_bytes + byte_off : slots);
}
__device__ __forceinline__ void ar_wait_all(ar_u32x4 (&in)[8]) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(in[0]), "+v"(in[1]), "+v"(in[2]), "+v"(in[3]), "+v"(in[4]), "+v"(in[5]), "+v"(in[6]), "+v"(in[7])::"memory");
}

constexpr int kArVecPerThread = 2;   // 16-byte vectors a thread holds in flight per pass

// The residual add + RMSNorm that follows the all-reduce of an o_proj / down_proj result, done by the
// workgroups that have just summed it (NV > 0: a workgroup owns ONE row of the [rows][hidden] message,
// a thread NV 16-byte vectors of it): residual = round(round(sum) + residual), out = round(residual *
// inv_rms) * weight -- the arithmetic, the thread-to-vector map and the order of the sum of squares of
// rms_norm_slabs_kernel, so the fused call is bit-identical to the all-reduce followed by
// mi355_fused_add_rms_norm; the summed projection itself is not written anywhere (nobody reads it).
struct ArNorm {
  uint16_t* residual;       // [rows][res_stride] in/out
  const uint16_t* weight;   // [hidden]
  uint16_t* out;            // [rows][out_stride]
  int64_t res_stride, out_stride;
  int hidden;
  float eps;
};

// `slabs` != nullptr: this rank's input is not `data` but round(slabs[0] + slabs[1] + ...), fp32
// partial sums [num_slabs][count] of a projection whose K was split over workgroups, added in slab
// order and rounded once -- the 16-bit tensor a separate slab-summing launch would have formed
// (mi355_sum_slabs), formed here on the way into the peers' buffers; `data` is then output only.
template <typename T, int NV = 0>
__global__ __launch_bounds__(kArThreads) void allreduce_oneshot_kernel(
    uint16_t* __restrict__ data, int64_t count, ArPeers peers, int rank, int world, int64_t max_bytes,
    unsigned* __restrict__ rounds, unsigned* __restrict__ error, const float* __restrict__ slabs, int num_slabs,
    ArNorm nm) {
  const int b = blockIdx.x, nb = gridDim.x, tid = threadIdx.x;
  // the call's round: one more than the last completed call's (nobody advances the word before
  // every workgroup of this call has finished, hence read it)
  const unsigned round = __hip_atomic_load(rounds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
  const int parity = round & 1;
  const int64_t nvec = count >> 3;                    // 16-byte vectors (count % 8 == 0)
  const int64_t per = NV > 0 ? (nm.hidden >> 3) : (nvec + nb - 1) / nb;   // NV > 0: chunk b is row b
  const int64_t v0 = b * per, v1 = min(nvec, v0 + per);
  const int64_t slot_bytes = max_bytes;
  const int64_t flags_off = ar_flags_offset(world, max_bytes);
  const uint4* src = reinterpret_cast<const uint4*>(data);

  // 1. deliver this rank's chunk to every rank (own buffer included: the sum below reads all slots alike;
  //    NV > 0: the own chunk stays in registers instead -- no store to, flag in or load from the own buffer)
  const int64_t slot_off = (static_cast<int64_t>(parity) * world + rank) * slot_bytes;
  uint4 own[NV > 0 ? NV : 1];
#pragma unroll
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) own[i] = make_uint4(0, 0, 0, 0);
  for (int64_t v = v0 + tid; v < v1; v += kArThreads) {
    uint4 mine_v;
    if (slabs != nullptr) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int g = 0; g < num_slabs; ++g) {
        const float4* sp = reinterpret_cast<const float4*>(slabs + g * count + 8 * v);
        const float4 a = sp[0], b4 = sp[1];
        acc[0] += a.x, acc[1] += a.y, acc[2] += a.z, acc[3] += a.w;
        acc[4] += b4.x, acc[5] += b4.y, acc[6] += b4.z, acc[7] += b4.w;
      }
      mine_v = pack8<T>(acc);
    } else {
      mine_v = src[v];
    }
    if (NV > 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (v == v0 + tid + i * kArThreads) own[i] = mine_v;
    }
    for (int p = 0; p < world; ++p)
      if (NV == 0 || p != rank) ar_store_sys(peers.buf[p] + slot_off + v * 16, mine_v);
  }
  // the norm's own operands are asked for here: they arrive while the flags are waited for
  uint4 nres[NV > 0 ? NV : 1], nw[NV > 0 ? NV : 1];
  if (NV > 0) {
    const uint4* rr = reinterpret_cast<const uint4*>(nm.residual + static_cast<int64_t>(b) * nm.res_stride);
    const uint4* wv = reinterpret_cast<const uint4*>(nm.weight);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = min(tid + i * kArThreads, static_cast<int>(per) - 1);
      nres[i] = MI355_LD(uint4, nm.residual, rr + v);
      nw[i] = MI355_LD(uint4, nm.weight, wv + v);
    }
  }
  // 2. stores first, then the flag: every storing wave waits until its stores have been acknowledged,
  //    the workgroup meets, then one lane per peer raises this chunk's flag there
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid < world && (NV == 0 || tid != rank)) {
    unsigned* flag = reinterpret_cast<unsigned*>(peers.buf[tid] + flags_off) + rank * kArMaxBlocks + b;
    __hip_atomic_store(flag, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // 3. wait for chunk b of every rank (bounded)
  __shared__ int ok_s;
  if (tid == 0) ok_s = 1;
  __syncthreads();
  if (tid < world && (NV == 0 || tid != rank)) {
    const unsigned* flag = reinterpret_cast<const unsigned*>(peers.buf[rank] + flags_off) + tid * kArMaxBlocks + b;
    const long long t0 = wall_clock64();
    bool ok = true;
    // rounds are compared modulo 2^32: "reached" = not more than half the range behind
    while (static_cast<int>(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - round) < 0) {
      if (wall_clock64() - t0 > kSpinLimitTicks) {
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    if (!ok) {
      ok_s = 0;
      atomicExch(error, 1u + static_cast<unsigned>(tid));     // which peer did not arrive
    }
  }
  __syncthreads();                                    // the data loads below come after every flag was seen
  const bool ok_all = ok_s != 0;
  if (NV > 0) {
    // 4'. the row's sum (rank order, fp32, one rounding -- as below), then the residual add and the norm
    __shared__ float part[kArThreads / 64];
    float f[NV > 0 ? NV : 1][8];
    float ss = 0.f;
    if (ok_all) {
      const unsigned char* mine = peers.buf[rank] + static_cast<int64_t>(parity) * world * slot_bytes;
      ar_u32x4 in[NV > 0 ? NV : 1][kArMaxWorld];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int64_t v = v0 + min(tid + i * kArThreads, static_cast<int>(per) - 1);
        ar_load_all(in[i], mine, slot_bytes, v * 16, world, rank);
      }
      uint4* rr = reinterpret_cast<uint4*>(nm.residual + static_cast<int64_t>(b) * nm.res_stride);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int v = tid + i * kArThreads;
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
        ar_wait_all(in[i]);
#pragma unroll
        for (int p = 0; p < kArMaxWorld; ++p) {
          if (p < world) {
            float g[8];
            const bool mine_p = p == rank;            // (a thread beyond the row holds nothing: its values go nowhere)
            unpack8<T>(mine_p ? own[i] : make_uint4(in[i][p].x, in[i][p].y, in[i][p].z, in[i][p].w), g);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] += g[q];
          }
        }
        float r[8];
        unpack8<T>(pack8<T>(acc), f[i]);               // the 16-bit tensor the plain all-reduce would have left
        unpack8<T>(nres[i], r);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[i][j] = round_f<T>(f[i][j] + r[j]);
        if (v < per) {
          MI355_ST(uint4, nm.residual, rr + v, pack8<T>(f[i]));
#pragma unroll
          for (int j = 0; j < 8; ++j) ss += f[i][j] * f[i][j];
        }
      }
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) part[tid >> 6] = ss;
    __syncthreads();
    if (ok_all) {
      float total = 0.f;
#pragma unroll
      for (int i = 0; i < kArThreads / 64; ++i) total += part[i];
      const float inv = 1.0f / sqrtf(total / static_cast<float>(nm.hidden) + nm.eps);
      uint4* orow = reinterpret_cast<uint4*>(nm.out + static_cast<int64_t>(b) * nm.out_stride);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int v = tid + i * kArThreads;
        float w[8], y[8];
        unpack8<T>(nw[i], w);
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = round_f<T>(f[i][j] * inv) * w[j];
        if (v < per) MI355_ST(uint4, nm.out, orow + v, pack8<T>(y));
      }
    }
  } else if (ok_all) {
    // 4. sum in rank order, fp32, one rounding; the loads of a pass (vectors x ranks) are all in flight together
    const unsigned char* mine = peers.buf[rank] + static_cast<int64_t>(parity) * world * slot_bytes;
    uint4* out = reinterpret_cast<uint4*>(data);
    for (int64_t vb = v0 + tid; vb < v1; vb += kArVecPerThread * kArThreads) {
      ar_u32x4 in[kArVecPerThread][kArMaxWorld];
#pragma unroll
      for (int j = 0; j < kArVecPerThread; ++j) {
        const int64_t v = min(vb + j * kArThreads, v1 - 1);        // clamped: a surplus load re-reads the last vector
        ar_load_all(in[j], mine, slot_bytes, v * 16, world, -1);
      }
#pragma unroll
      for (int j = 0; j < kArVecPerThread; ++j) {
        const int64_t v = vb + j * kArThreads;
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
        ar_wait_all(in[j]);
#pragma unroll
        for (int p = 0; p < kArMaxWorld; ++p) {
          if (p < world) {
            float f[8];
            unpack8<T>(make_uint4(in[j][p].x, in[j][p].y, in[j][p].z, in[j][p].w), f);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] += f[q];
          }
        }
        if (v < v1) out[v] = pack8<T>(acc);
      }
    }
  }   // a timed-out wait leaves the input as it was; the host sees the error word
  // the last workgroup to finish spends the round (a failed call too: the peers' flags stay comparable)
  __syncthreads();
  if (tid == 0) {
    if (__hip_atomic_fetch_add(rounds + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == static_cast<unsigned>(nb - 1)) {
      __hip_atomic_store(rounds + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(rounds, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// The same all-reduce + residual add + norm in TWO hops, for messages whose bytes, not the latency of a hop, set
// the time: a rank's links carry (world - 1) copies of the whole message in the one-shot form, 2 (world - 1) /
// world copies here.  The rows are dealt to the ranks in contiguous slices; workgroup b of every rank handles
// row b:
//   1. a rank that does not own row b stores its row into ITS slot of the owner's buffer and raises flag
//      [rank][b] there;
//   2. the owner waits for those world - 1 flags, sums the world rows in RANK ORDER in fp32 (its own from
//      registers), rounds once -- the value the one-shot form computes on every rank -- and stores the 16-bit
//      row into ITS slot of every peer's buffer, then raises flag [owner][b] there;
//   3. the others wait for that one flag and read the row from their own buffer; every rank then adds the
//      residual and normalises, as in the one-shot form (bit-identical results: the same sum, the same order).
// Slots and flags are the one-shot kernel's, used disjointly: row b of slot p at rank q holds p's partial row if
// q owns b, p's finished row if p owns it, and nothing otherwise; rounds, parities and bounded waits as above (an
// owner that gives up raises no flag, so its peers give up after their own wait).
template <int NV>
__device__ __forceinline__ void ar_wait_n(ar_u32x4 (&r)[NV]) {
  static_assert(NV == 2 || NV == 4, "the waits below name two or four registers");
  if constexpr (NV == 2)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1])::"memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3])::"memory");
}

template <typename T, int NV>
__global__ __launch_bounds__(kArThreads) void allreduce_twoshot_norm_kernel(
    const uint16_t* __restrict__ data, int64_t count, ArPeers peers, int rank, int world, int64_t max_bytes,
    unsigned* __restrict__ rounds, unsigned* __restrict__ error, const float* __restrict__ slabs, int num_slabs,
    ArNorm nm) {
  const int b = blockIdx.x, nb = gridDim.x, tid = threadIdx.x;
  const unsigned round = __hip_atomic_load(rounds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
  const int parity = round & 1;
  const int per = nm.hidden >> 3;                      // 16-byte vectors of a row
  const int64_t v0 = static_cast<int64_t>(b) * per;
  const int rows_per = (nb + world - 1) / world;
  const int owner = b / rows_per;                      // < world (b < nb <= rows_per * world); the same on every rank
  const bool own_row = owner == rank;
  const int64_t slot_bytes = max_bytes;
  const int64_t flags_off = ar_flags_offset(world, max_bytes);
  const int64_t parity_off = static_cast<int64_t>(parity) * world * slot_bytes;
  const uint4* src = reinterpret_cast<const uint4*>(data);
  unsigned char* local = peers.buf[rank];

  // 1. this rank's row: into registers, and to the owner unless that is this rank.  A thread beyond the row
  //    repeats the row's last vector (same value, same place).
  uint4 own[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int64_t v = v0 + min(tid + i * kArThreads, per - 1);
    if (slabs != nullptr) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int g = 0; g < num_slabs; ++g) {
        const float4* sp = reinterpret_cast<const float4*>(slabs + g * count + 8 * v);
        const float4 a = sp[0], b4 = sp[1];
        acc[0] += a.x, acc[1] += a.y, acc[2] += a.z, acc[3] += a.w;
        acc[4] += b4.x, acc[5] += b4.y, acc[6] += b4.z, acc[7] += b4.w;
      }
      own[i] = pack8<T>(acc);
    } else {
      own[i] = src[v];
    }
    if (!own_row) ar_store_sys(peers.buf[owner] + parity_off + rank * slot_bytes + v * 16, own[i]);
  }
  uint4 nres[NV], nw[NV];
  {
    const uint4* rr = reinterpret_cast<const uint4*>(nm.residual + static_cast<int64_t>(b) * nm.res_stride);
    const uint4* wv = reinterpret_cast<const uint4*>(nm.weight);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = min(tid + i * kArThreads, per - 1);
      nres[i] = MI355_LD(uint4, nm.residual, rr + v);
      nw[i] = MI355_LD(uint4, nm.weight, wv + v);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (!own_row && tid == 0) {
    unsigned* flag = reinterpret_cast<unsigned*>(peers.buf[owner] + flags_off) + rank * kArMaxBlocks + b;
    __hip_atomic_store(flag, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __shared__ int ok_s;
  if (tid == 0) ok_s = 1;
  __syncthreads();
  // 2. / 3.  which lanes wait for whom: the owner's lanes p != rank for the rows of the peers, one lane of
  //    everybody else for the owner's finished row
  const bool waits = own_row ? (tid < world && tid != rank) : tid == 0;
  if (waits) {
    const int from = own_row ? tid : owner;
    const unsigned* flag = reinterpret_cast<const unsigned*>(local + flags_off) + from * kArMaxBlocks + b;
    const long long t0 = wall_clock64();
    bool ok = true;
    while (static_cast<int>(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - round) < 0) {
      if (wall_clock64() - t0 > kSpinLimitTicks) {
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    if (!ok) {
      ok_s = 0;
      atomicExch(error, 1u + static_cast<unsigned>(from));
    }
  }
  __syncthreads();
  const bool ok_all = ok_s != 0;
  uint4 red[NV];                                       // the finished 16-bit row, this thread's vectors
#pragma unroll
  for (int i = 0; i < NV; ++i) red[i] = make_uint4(0, 0, 0, 0);
  if (own_row) {                                       // (workgroup-uniform)
    if (ok_all) {
      ar_u32x4 in[NV][kArMaxWorld];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int64_t v = v0 + min(tid + i * kArThreads, per - 1);
        ar_load_all(in[i], local + parity_off, slot_bytes, v * 16, world, rank);
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int64_t v = v0 + min(tid + i * kArThreads, per - 1);
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
        ar_wait_all(in[i]);
#pragma unroll
        for (int p = 0; p < kArMaxWorld; ++p) {
          if (p < world) {
            float g[8];
            unpack8<T>(p == rank ? own[i] : make_uint4(in[i][p].x, in[i][p].y, in[i][p].z, in[i][p].w), g);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] += g[q];
          }
        }
        red[i] = pack8<T>(acc);
        for (int p = 0; p < world; ++p)
          if (p != rank) ar_store_sys(peers.buf[p] + parity_off + rank * slot_bytes + v * 16, red[i]);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (ok_all && tid < world && tid != rank) {
      unsigned* flag = reinterpret_cast<unsigned*>(peers.buf[tid] + flags_off) + rank * kArMaxBlocks + b;
      __hip_atomic_store(flag, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  } else if (ok_all) {
    ar_u32x4 in[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
      in[i] = ar_load_sys(local + parity_off + owner * slot_bytes + (v0 + min(tid + i * kArThreads, per - 1)) * 16);
    ar_wait_n<NV>(in);
#pragma unroll
    for (int i = 0; i < NV; ++i) red[i] = make_uint4(in[i].x, in[i].y, in[i].z, in[i].w);
  }
  // residual add + norm of the finished row: the arithmetic, thread-to-vector map and summation order of the
  // one-shot form (and of rms_norm_slabs_kernel)
  __shared__ float part[kArThreads / 64];
  float f[NV][8];
  float ss = 0.f;
  if (ok_all) {
    uint4* rr = reinterpret_cast<uint4*>(nm.residual + static_cast<int64_t>(b) * nm.res_stride);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + i * kArThreads;
      float r[8];
      unpack8<T>(red[i], f[i]);
      unpack8<T>(nres[i], r);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[i][j] = round_f<T>(f[i][j] + r[j]);
      if (v < per) {
        MI355_ST(uint4, nm.residual, rr + v, pack8<T>(f[i]));
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += f[i][j] * f[i][j];
      }
    }
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) part[tid >> 6] = ss;
  __syncthreads();
  if (ok_all) {
    float total = 0.f;
#pragma unroll
    for (int i = 0; i < kArThreads / 64; ++i) total += part[i];
    const float inv = 1.0f / sqrtf(total / static_cast<float>(nm.hidden) + nm.eps);
    uint4* orow = reinterpret_cast<uint4*>(nm.out + static_cast<int64_t>(b) * nm.out_stride);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + i * kArThreads;
      float w[8], y[8];
      unpack8<T>(nw[i], w);
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] = round_f<T>(f[i][j] * inv) * w[j];
      if (v < per) MI355_ST(uint4, nm.out, orow + v, pack8<T>(y));
    }
  }
  __syncthreads();
  if (tid == 0) {
    if (__hip_atomic_fetch_add(rounds + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == static_cast<unsigned>(nb - 1)) {
      __hip_atomic_store(rounds + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(rounds, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Max-reduce of one 64-bit key per row (the greedy pick of a vocabulary-sharded step: every rank holds the
// (logit, index) key of its slice's best token, mi355_argmax_shard_key) and the token ids of the winners, in
// ONE workgroup: thread t delivers key t to the peers' slots, the flags of chunk 0 say when theirs are here,
// the max is taken over the ranks (signed 64-bit, as the process group's MAX does) and the id -- the low word,
// stored complemented so that the lower index wins a tie -- is written to out_ids (and out_ids2).  Same
// rounds, parities and bounded waits as the sum kernel; the own key stays in a register.
typedef unsigned int ar_u32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(kArThreads) void allreduce_oneshot_max_keys_kernel(
    const int64_t* __restrict__ keys, int64_t* __restrict__ out_ids, int64_t* __restrict__ out_ids2, int rows,
    ArPeers peers, int rank, int world, int64_t max_bytes, unsigned* __restrict__ rounds, unsigned* __restrict__ error) {
  const int tid = threadIdx.x;
  const unsigned round = __hip_atomic_load(rounds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
  const int parity = round & 1;
  const int64_t flags_off = ar_flags_offset(world, max_bytes);
  const int64_t slot_off = (static_cast<int64_t>(parity) * world + rank) * max_bytes;
  const int t = min(tid, rows - 1);                    // surplus threads repeat the last row (same value, same place)
  const int64_t mine = MI355_LD(int64_t, keys, keys + t);
  const ar_u32x2 mv = {static_cast<unsigned>(static_cast<unsigned long long>(mine) & 0xFFFFFFFFull),
                       static_cast<unsigned>(static_cast<unsigned long long>(mine) >> 32)};
  for (int p = 0; p < world; ++p)
    if (p != rank) {
      void* dst = peers.buf[p] + slot_off + t * 8;
      asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(dst), "v"(mv) : "memory");
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid < world && tid != rank) {
    unsigned* flag = reinterpret_cast<unsigned*>(peers.buf[tid] + flags_off) + rank * kArMaxBlocks;
    __hip_atomic_store(flag, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __shared__ int ok_s;
  if (tid == 0) ok_s = 1;
  __syncthreads();
  if (tid < world && tid != rank) {
    const unsigned* flag = reinterpret_cast<const unsigned*>(peers.buf[rank] + flags_off) + tid * kArMaxBlocks;
    const long long t0 = wall_clock64();
    bool ok = true;
    while (static_cast<int>(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - round) < 0) {
      if (wall_clock64() - t0 > kSpinLimitTicks) {
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    if (!ok) {
      ok_s = 0;
      atomicExch(error, 1u + static_cast<unsigned>(tid));
    }
  }
  __syncthreads();
  if (ok_s != 0) {                                     // a timed-out wait leaves the outputs as they were
    const unsigned char* slots = peers.buf[rank] + static_cast<int64_t>(parity) * world * max_bytes;
    // straight-line: a load for every possible rank (a slot that does not take part: one fixed address, its value
    // unused), one wait, selects -- no branch between a load and the wait that covers it, so the compiler has no
    // reason to copy a destination register before its data has arrived
    ar_u32x2 in[kArMaxWorld];
#pragma unroll
    for (int p = 0; p < kArMaxWorld; ++p) {
      const bool peer = p < world && p != rank;
      const void* src = peer ? static_cast<const void*>(slots + p * max_bytes + t * 8) : static_cast<const void*>(slots);
      asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(in[p]) : "v"(src) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(in[0]), "+v"(in[1]), "+v"(in[2]), "+v"(in[3]), "+v"(in[4]), "+v"(in[5]), "+v"(in[6]), "+v"(in[7])::"memory");
    static_assert(kArMaxWorld == 8, "the wait above names eight registers");
    int64_t best = mine;
#pragma unroll
    for (int p = 0; p < kArMaxWorld; ++p) {
      const bool peer = p < world && p != rank;
      const int64_t k = static_cast<int64_t>((static_cast<unsigned long long>(in[p].y) << 32) | in[p].x);
      best = (peer && k > best) ? k : best;
    }
    if (tid < rows) {
      const int64_t idx = static_cast<int64_t>(0xFFFFFFFFu - static_cast<uint32_t>(static_cast<unsigned long long>(best) & 0xFFFFFFFFull));
      MI355_ST(int64_t, out_ids, out_ids + tid, idx);
      if (out_ids2 != nullptr) MI355_ST(int64_t, out_ids2, out_ids2 + tid, idx);
    }
  }
  __syncthreads();
  if (tid == 0) __hip_atomic_store(rounds, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // one workgroup: it is the last
}

}  // namespace mi355

extern "C" int64_t mi355_allreduce_oneshot_handle_bytes(void) { return static_cast<int64_t>(sizeof(hipIpcMemHandle_t)); }

extern "C" int32_t mi355_allreduce_oneshot_init(void** comm_out, int32_t rank, int32_t world, int64_t max_bytes,
                                                void* handle_out) {
  using namespace mi355;
  const char* name = "mi355_allreduce_oneshot_init";
  MI355_REQUIRE(comm_out && handle_out, MI355_ERR_INVALID, "%s: null pointer", name);
  MI355_REQUIRE(world >= 1 && world <= kArMaxWorld && rank >= 0 && rank < world, MI355_ERR_INVALID,
                "%s: rank %d / world %d out of range (world <= %d)", name, rank, world, kArMaxWorld);
  MI355_REQUIRE(max_bytes >= 16 && max_bytes % 16 == 0 && max_bytes <= (16 << 20), MI355_ERR_INVALID,
                "%s: max_bytes %lld must be a multiple of 16 in [16, 16 MiB]", name, (long long)max_bytes);
  ArComm* c = new ArComm();
  memset(c, 0, sizeof(*c));
  c->rank = rank, c->world = world, c->max_bytes = max_bytes;
  hipError_t e = hipGetDevice(&c->device);
  const int64_t bytes = ar_buffer_bytes(world, max_bytes);
  if (e == hipSuccess) e = hipExtMallocWithFlags(reinterpret_cast<void**>(&c->local), bytes, hipDeviceMallocUncached);
  if (e == hipSuccess) e = hipMemset(c->local, 0, bytes);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->rounds), 4 * sizeof(unsigned));
  if (e == hipSuccess) e = hipMemset(c->rounds, 0, 4 * sizeof(unsigned));
  if (e == hipSuccess) e = hipIpcGetMemHandle(static_cast<hipIpcMemHandle_t*>(handle_out), c->local);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    if (c->local) (void)hipFree(c->local);
    if (c->rounds) (void)hipFree(c->rounds);
    delete c;
    return fail(MI355_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e));
  }
  c->error = c->rounds + 2;
  c->peer[rank] = c->local;
  *comm_out = c;
  return MI355_OK;
}

extern "C" int32_t mi355_allreduce_oneshot_connect(void* comm, const void* all_handles) {
  using namespace mi355;
  const char* name = "mi355_allreduce_oneshot_connect";
  MI355_REQUIRE(comm && all_handles, MI355_ERR_INVALID, "%s: null pointer", name);
  ArComm* c = static_cast<ArComm*>(comm);
  MI355_REQUIRE(!c->connected, MI355_ERR_INVALID, "%s: already connected", name);
  const auto* handles = static_cast<const hipIpcMemHandle_t*>(all_handles);
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank) continue;
    void* ptr = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&ptr, handles[p], hipIpcMemLazyEnablePeerAccess);
    MI355_REQUIRE(e == hipSuccess && ptr, MI355_ERR_LAUNCH, "%s: cannot map the buffer of rank %d: %s", name, p,
                  hipGetErrorString(e));
    c->peer[p] = static_cast<unsigned char*>(ptr);
    c->opened[p] = true;
  }
  c->connected = true;
  return MI355_OK;
}

static int32_t allreduce_oneshot_launch(const char* name, void* comm, void* data, const float* slabs, int32_t num_slabs,
                                        int64_t count, int32_t dtype, void* stream, const mi355::ArNorm* norm = nullptr,
                                        int32_t rows = 0, bool two_shot = false) {
  using namespace mi355;
  MI355_REQUIRE(comm && (data || (norm && slabs)), MI355_ERR_INVALID, "%s: null pointer", name);
  ArComm* c = static_cast<ArComm*>(comm);
  MI355_REQUIRE(c->connected, MI355_ERR_INVALID, "%s: not connected (exchange the handles first)", name);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED, "%s: dtype %d not supported (bf16/fp16 only)",
                name, dtype);
  MI355_REQUIRE(count >= 0 && count % 8 == 0 && count * 2 <= c->max_bytes && (data == nullptr || aligned16(data)), MI355_ERR_INVALID,
                "%s: count %lld must be a multiple of 8 with count * 2 <= max_bytes %lld, data 16-byte aligned", name,
                (long long)count, (long long)c->max_bytes);
  MI355_REQUIRE(slabs == nullptr || (num_slabs >= 1 && num_slabs <= 8 && aligned16(slabs)), MI355_ERR_INVALID,
                "%s: num_slabs %d outside [1, 8] or slabs not 16-byte aligned", name, num_slabs);
  if (count == 0) return MI355_OK;
  ArPeers peers;
  for (int p = 0; p < kArMaxWorld; ++p) peers.buf[p] = c->peer[p < c->world ? p : c->rank];
  // one workgroup per 8 KB (two 16-byte vectors a thread: one pass of loads), the SAME count on every
  // rank (it only depends on `count`)
  int nb = static_cast<int>((count * 2 + 8191) / 8192);
  nb = nb < 1 ? 1 : nb > kArMaxBlocks ? kArMaxBlocks : nb;
  hipStream_t s = static_cast<hipStream_t>(stream);
#define MI355_AR_LAUNCH(T_, NV_, NB_, NM_)                                                                               \
  hipLaunchKernelGGL((allreduce_oneshot_kernel<T_, NV_>), dim3(NB_), dim3(kArThreads), 0, s, static_cast<uint16_t*>(data), \
                     count, peers, c->rank, c->world, c->max_bytes, c->rounds, c->error, slabs, num_slabs, NM_)
#define MI355_AR2_LAUNCH(T_, NV_)                                                                                          \
  hipLaunchKernelGGL((allreduce_twoshot_norm_kernel<T_, NV_>), dim3(rows), dim3(kArThreads), 0, s,                         \
                     static_cast<const uint16_t*>(data), count, peers, c->rank, c->world, c->max_bytes, c->rounds, c->error, \
                     slabs, num_slabs, *norm)
  if (norm != nullptr && two_shot) {
    const bool wide = (norm->hidden >> 3) > 2 * kArThreads;
    if (dtype == MI355_BF16) {
      if (wide) MI355_AR2_LAUNCH(BF16, 4); else MI355_AR2_LAUNCH(BF16, 2);
    } else {
      if (wide) MI355_AR2_LAUNCH(F16, 4); else MI355_AR2_LAUNCH(F16, 2);
    }
  } else if (norm != nullptr) {      // one workgroup per row (the same count on every rank: it is the row count)
    const bool wide = (norm->hidden >> 3) > 2 * kArThreads;
    if (dtype == MI355_BF16) {
      if (wide) MI355_AR_LAUNCH(BF16, 4, rows, *norm); else MI355_AR_LAUNCH(BF16, 2, rows, *norm);
    } else {
      if (wide) MI355_AR_LAUNCH(F16, 4, rows, *norm); else MI355_AR_LAUNCH(F16, 2, rows, *norm);
    }
  } else if (dtype == MI355_BF16) {
    MI355_AR_LAUNCH(BF16, 0, nb, ArNorm{});
  } else {
    MI355_AR_LAUNCH(F16, 0, nb, ArNorm{});
  }
#undef MI355_AR_LAUNCH
#undef MI355_AR2_LAUNCH
  return check_launch(name);
}

extern "C" int32_t mi355_allreduce_oneshot_run(void* comm, void* data, int64_t count, int32_t dtype, void* stream) {
  return allreduce_oneshot_launch("mi355_allreduce_oneshot_run", comm, data, nullptr, 0, count, dtype, stream);
}

extern "C" int32_t mi355_allreduce_oneshot_run_slabs(void* comm, void* out, const float* slabs, int32_t num_slabs,
                                                     int64_t count, int32_t dtype, void* stream) {
  using namespace mi355;
  const char* name = "mi355_allreduce_oneshot_run_slabs";
  MI355_REQUIRE(slabs != nullptr, MI355_ERR_INVALID, "%s: null pointer", name);
  return allreduce_oneshot_launch(name, comm, out, slabs, num_slabs, count, dtype, stream);
}

static int32_t allreduce_add_norm_entry(const char* name, bool two_shot, void* comm, const void* data, const float* slabs, int32_t num_slabs,
                                                        void* residual, int64_t residual_stride, const void* norm_weight,
                                                        void* out, int64_t out_stride, int32_t rows, int32_t hidden,
                                                        float eps, int32_t dtype, void* stream) {
  using namespace mi355;
  MI355_REQUIRE(residual && norm_weight && out && (data || slabs), MI355_ERR_INVALID, "%s: null pointer", name);
  MI355_REQUIRE(rows >= 0 && rows <= kArMaxBlocks, MI355_ERR_UNSUPPORTED, "%s: rows %d outside [0, %d]", name, rows,
                kArMaxBlocks);
  MI355_REQUIRE(hidden > 0 && hidden % 8 == 0 && (hidden >> 3) <= 4 * kArThreads, MI355_ERR_UNSUPPORTED,
                "%s: hidden %d must be a multiple of 8, at most %d", name, hidden, 32 * kArThreads);
  MI355_REQUIRE(residual_stride >= hidden && residual_stride % 8 == 0 && out_stride >= hidden && out_stride % 8 == 0 &&
                    aligned16(residual) && aligned16(norm_weight) && aligned16(out) && eps >= 0.f,
                MI355_ERR_INVALID, "%s: residual / norm_weight / out must be 16-byte aligned, strides >= hidden and multiples of 8",
                name);
  if (rows == 0) return MI355_OK;
  const ArNorm nm{static_cast<uint16_t*>(residual), static_cast<const uint16_t*>(norm_weight), static_cast<uint16_t*>(out),
                  residual_stride, out_stride, hidden, eps};
  return allreduce_oneshot_launch(name, comm, const_cast<void*>(data), slabs, slabs ? num_slabs : 0,
                                  static_cast<int64_t>(rows) * hidden, dtype, stream, &nm, rows, two_shot);
}

extern "C" int32_t mi355_allreduce_oneshot_run_add_norm(void* comm, const void* data, const float* slabs, int32_t num_slabs,
                                                        void* residual, int64_t residual_stride, const void* norm_weight,
                                                        void* out, int64_t out_stride, int32_t rows, int32_t hidden,
                                                        float eps, int32_t dtype, void* stream) {
  return allreduce_add_norm_entry("mi355_allreduce_oneshot_run_add_norm", false, comm, data, slabs, num_slabs, residual,
                                  residual_stride, norm_weight, out, out_stride, rows, hidden, eps, dtype, stream);
}

extern "C" int32_t mi355_allreduce_twoshot_run_add_norm(void* comm, const void* data, const float* slabs, int32_t num_slabs,
                                                        void* residual, int64_t residual_stride, const void* norm_weight,
                                                        void* out, int64_t out_stride, int32_t rows, int32_t hidden,
                                                        float eps, int32_t dtype, void* stream) {
  return allreduce_add_norm_entry("mi355_allreduce_twoshot_run_add_norm", true, comm, data, slabs, num_slabs, residual,
                                  residual_stride, norm_weight, out, out_stride, rows, hidden, eps, dtype, stream);
}

extern "C" int32_t mi355_allreduce_oneshot_run_max_keys(void* comm, const int64_t* keys, int64_t* out_ids,
                                                        int64_t* out_ids2, int32_t rows, void* stream) {
  using namespace mi355;
  const char* name = "mi355_allreduce_oneshot_run_max_keys";
  MI355_REQUIRE(comm && keys && out_ids, MI355_ERR_INVALID, "%s: null pointer", name);
  ArComm* c = static_cast<ArComm*>(comm);
  MI355_REQUIRE(c->connected, MI355_ERR_INVALID, "%s: not connected (exchange the handles first)", name);
  MI355_REQUIRE(rows >= 0 && rows <= kArThreads && static_cast<int64_t>(rows) * 8 <= c->max_bytes, MI355_ERR_UNSUPPORTED,
                "%s: rows %d outside [0, %d]", name, rows, kArThreads);
  MI355_REQUIRE((reinterpret_cast<uintptr_t>(keys) & 7) == 0 && (reinterpret_cast<uintptr_t>(out_ids) & 7) == 0 &&
                    (reinterpret_cast<uintptr_t>(out_ids2) & 7) == 0,
                MI355_ERR_INVALID, "%s: pointers must be 8-byte aligned", name);
  if (rows == 0) return MI355_OK;
  ArPeers peers;
  for (int p = 0; p < kArMaxWorld; ++p) peers.buf[p] = c->peer[p < c->world ? p : c->rank];
  hipLaunchKernelGGL(allreduce_oneshot_max_keys_kernel, dim3(1), dim3(kArThreads), 0, static_cast<hipStream_t>(stream), keys,
                     out_ids, out_ids2, rows, peers, c->rank, c->world, c->max_bytes, c->rounds, c->error);
  return check_launch(name);
}

extern "C" int32_t mi355_allreduce_oneshot_status_async(void* comm, uint32_t* host_word, void* stream) {
  using namespace mi355;
  const char* name = "mi355_allreduce_oneshot_status_async";
  MI355_REQUIRE(comm && host_word, MI355_ERR_INVALID, "%s: null pointer", name);
  ArComm* c = static_cast<ArComm*>(comm);
  const hipError_t e = hipMemcpyAsync(host_word, c->error, sizeof(uint32_t), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream));
  MI355_REQUIRE(e == hipSuccess, MI355_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e));
  return MI355_OK;
}

extern "C" int32_t mi355_allreduce_oneshot_status(void* comm) {
  using namespace mi355;
  const char* name = "mi355_allreduce_oneshot_status";
  MI355_REQUIRE(comm, MI355_ERR_INVALID, "%s: null pointer", name);
  ArComm* c = static_cast<ArComm*>(comm);
  unsigned err = 0;
  const hipError_t e = hipMemcpy(&err, c->error, sizeof(err), hipMemcpyDeviceToHost);   // waits for the work queued before it
  MI355_REQUIRE(e == hipSuccess, MI355_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e));
  MI355_REQUIRE(err == 0, MI355_ERR_LAUNCH, "%s: rank %d waited 10 s for rank %u, which never delivered its chunk", name, c->rank,
                err - 1);
  return MI355_OK;
}

extern "C" int32_t mi355_allreduce_oneshot_destroy(void* comm) {
  using namespace mi355;
  if (!comm) return MI355_OK;
  ArComm* c = static_cast<ArComm*>(comm);
  (void)hipDeviceSynchronize();
  for (int p = 0; p < c->world; ++p)
    if (c->opened[p]) (void)hipIpcCloseMemHandle(c->peer[p]);
  if (c->local) (void)hipFree(c->local);
  if (c->rounds) (void)hipFree(c->rounds);
  delete c;
  return MI355_OK;
}
