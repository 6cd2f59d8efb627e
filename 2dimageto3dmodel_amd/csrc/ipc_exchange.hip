// Small-message all-reduce over peer-mapped device memory: the SyncBN statistics exchange of the data-parallel GAN path without a
// collective-library call (replaces the master / slave pipes of code/sync_batchnorm/batchnorm.py:110-131 and
// code/sync_batchnorm/comm.py, and -- behind a switch -- the 56 RCCL all-reduces of <= 4097 floats a training cycle issues).
//
// Every rank owns ONE fine-grained device region (hipExtMallocWithFlags: coherent across devices inside a kernel) that it exports
// with hipIpcGetMemHandle; the peers map it (hipIpcOpenMemHandle, xGMI on a multi-GPU node).  A message is ONE launch of one
// workgroup on the compute stream, no host round trip:
//   1. wait until every peer has finished READING the message that last used this slot (their `done` counter);
//   2. copy the local vector into the own region's slot, release-store the slot's flag = sequence number (system scope);
//   3. acquire-spin on every peer's flag for the same sequence number, then read their slots;
//   4. add the W vectors in RANK ORDER -- every rank performs the same additions in the same order: bit-equal results on all ranks,
//      run-to-run deterministic;
//   5. release-store `done` = sequence number.
// Messages are matched by CHANNEL, not by global order: a channel is one call site (a SyncBN layer's forward, or its backward), with its
// own sequence counter, flags and two slots in every region.  Consecutive messages of one call site are ordered by data dependence on
// every rank; messages of DIFFERENT call sites may execute in different orders on different ranks (the generator's mesh head runs on a
// second stream, gan_ops.Fork) -- a global sequence number paired the wrong vectors there (found by the two-process test: a training
// cycle's bits changed).  The counters live in the region (the kernel increments them): a launch is replayable from a hipGraph.
// EVERY spin is bounded by a wall-clock timeout (default 2 s): a missing peer raises a bit in the status word and the kernel leaves --
// a wrong result that is reported, never a hung GPU.
#include <cstring>

#include "common.h"

namespace m355 {

constexpr int kIpcSlots = 2;          // per channel: a rank can be one message ahead of the slowest reader
constexpr int kIpcMaxFloats = 4104;   // 2 * 2048 channels + the pixel count, rounded up
constexpr int kIpcChannels = M355_IPC_CHANNELS;

struct IpcChannel {
    unsigned seq;                    // messages this rank has sent on the channel (written by the owner's kernel only)
    unsigned done;                   // highest sequence number whose peer slots this rank has finished reading
    unsigned flag[kIpcSlots];        // sequence number of the message that is complete in the slot
    unsigned pad_[12];
    float data[kIpcSlots][kIpcMaxFloats];
};
struct IpcRegion {
    IpcChannel ch[kIpcChannels];
};

struct IpcPeers {
    IpcRegion *r[M355_IPC_MAX_RANKS];
};

__device__ __forceinline__ unsigned ld_acquire_sys(const unsigned *p)
{
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_release_sys(unsigned *p, unsigned v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// spin until *p >= want (sequence numbers only grow; wrap-around after 2^32 messages is not handled: 2^32 x 20 us = a day of nothing
// but messages); false on timeout
__device__ __forceinline__ bool spin_until(const unsigned *p, unsigned want, long long t_end)
{
    for (;;) {
        if ((int)(ld_acquire_sys(p) - want) >= 0) return true;
        if (wall_clock64() > t_end) return false;
        __builtin_amdgcn_s_sleep(2);
    }
}

__global__ __launch_bounds__(256) void k_ipc_allreduce(float *__restrict__ inout, int n, IpcPeers peers, int rank, int world, int channel,
                                                       unsigned *__restrict__ status, long long timeout_ticks)
{
    __shared__ unsigned seq_s;
    __shared__ int bad_s;
    const int tid = threadIdx.x;
    IpcChannel *mine = &peers.r[rank]->ch[channel];
    if (tid == 0) {
        seq_s = mine->seq + 1;
        bad_s = 0;
    }
    __syncthreads();
    const unsigned seq = seq_s;
    const int slot = seq % kIpcSlots;
    const long long t_end = wall_clock64() + timeout_ticks;
    // 1. the slot's previous message (seq - NSLOT) has been read by everyone
    if (tid < world && tid != rank && seq > (unsigned)kIpcSlots) {
        if (!spin_until(&peers.r[tid]->ch[channel].done, seq - kIpcSlots, t_end)) bad_s = 1;
    }
    __syncthreads();
    // 2. publish
    for (int i = tid; i < n; i += 256) mine->data[slot][i] = inout[i];
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        mine->seq = seq;
        st_release_sys(&mine->flag[slot], seq);
    }
    // 3. the peers' messages
    if (tid < world && tid != rank) {
        if (!spin_until(&peers.r[tid]->ch[channel].flag[slot], seq, t_end)) bad_s = 2;
    }
    __syncthreads();
    __threadfence_system();
    // 4. rank-ordered sum (the own contribution from `inout`: same bits as the published copy)
    for (int i = tid; i < n; i += 256) {
        float s = 0.0f;
        for (int p = 0; p < world; ++p) {
            const float v = p == rank ? inout[i] : __builtin_nontemporal_load(&peers.r[p]->ch[channel].data[slot][i]);
            s = p == 0 ? v : s + v;
        }
        inout[i] = s;
    }
    __syncthreads();
    // 5. done reading
    if (tid == 0) {
        st_release_sys(&mine->done, seq);
        if (bad_s) atomicOr(status, (unsigned)bad_s);
    }
}

}  // namespace m355

using namespace m355;

extern "C" size_t m355_ipc_region_bytes(void) { return sizeof(IpcRegion); }
extern "C" int m355_ipc_max_floats(void) { return kIpcMaxFloats; }
extern "C" int m355_ipc_channels(void) { return kIpcChannels; }

/* one region per rank: fine-grained device memory, zeroed, + its 64-byte IPC handle (exchange it with the peers through any channel) */
extern "C" int m355_ipc_alloc(void **region, void *handle64)
{
    M355_REQUIRE(region && handle64, "ipc_alloc: null pointer");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the ABI passes IPC handles as 64 bytes");
    void *p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, sizeof(IpcRegion), hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        set_error("ipc_alloc: hipExtMallocWithFlags(finegrained): %s", hipGetErrorString(e));
        return M355_ERR_LAUNCH;
    }
    e = hipMemset(p, 0, sizeof(IpcRegion));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t *>(handle64), p);
    if (e != hipSuccess) {
        set_error("ipc_alloc: %s", hipGetErrorString(e));
        (void)hipFree(p);
        return M355_ERR_LAUNCH;
    }
    *region = p;
    return M355_OK;
}

extern "C" int m355_ipc_open(const void *handle64, void **region)
{
    M355_REQUIRE(region && handle64, "ipc_open: null pointer");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    void *p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
        set_error("ipc_open: hipIpcOpenMemHandle: %s", hipGetErrorString(e));
        return M355_ERR_LAUNCH;
    }
    *region = p;
    return M355_OK;
}

extern "C" int m355_ipc_close(void *region)
{
    if (!region) return M355_OK;
    hipError_t e = hipIpcCloseMemHandle(region);
    if (e != hipSuccess) {
        set_error("ipc_close: %s", hipGetErrorString(e));
        return M355_ERR_LAUNCH;
    }
    return M355_OK;
}

extern "C" int m355_ipc_free(void *region)
{
    if (!region) return M355_OK;
    hipError_t e = hipFree(region);
    if (e != hipSuccess) {
        set_error("ipc_free: %s", hipGetErrorString(e));
        return M355_ERR_LAUNCH;
    }
    return M355_OK;
}

/* inout[n] (device, fp32) <- sum over the ranks' vectors, added in rank order.  regions[world]: every rank's region as mapped into
 * THIS process (regions[rank] = the own one), host array.  status: one device word, bit 0 / 1 raised when a wait timed out (a peer
 * did not show up within timeout_ms; the result is then wrong -- check it where the host synchronises anyway).  channel: the call site
 * (0 .. m355_ipc_channels() - 1); messages of one channel are matched in order, different channels are independent.  n <= m355_ipc_max_floats(),
 * world <= M355_IPC_MAX_RANKS.  Asynchronous on `stream`; every rank must issue the same sequence of calls. */
extern "C" int m355_ipc_allreduce(float *inout, int n, void *const *regions, int rank, int world, int channel, unsigned *status,
                                  int timeout_ms, void *stream)
{
    M355_REQUIRE(inout && regions && status && n > 0 && n <= kIpcMaxFloats, "ipc_allreduce: bad vector (n <= %d)", kIpcMaxFloats);
    M355_REQUIRE(channel >= 0 && channel < kIpcChannels, "ipc_allreduce: channel %d outside [0, %d)", channel, kIpcChannels);
    M355_REQUIRE(world >= 1 && world <= M355_IPC_MAX_RANKS && rank >= 0 && rank < world, "ipc_allreduce: bad rank / world");
    IpcPeers peers{};
    for (int p = 0; p < world; ++p) {
        M355_REQUIRE(regions[p], "ipc_allreduce: region %d is not mapped", p);
        peers.r[p] = static_cast<IpcRegion *>(regions[p]);
    }
    const long long ticks = (long long)(timeout_ms > 0 ? timeout_ms : 2000) * 100000LL;   // wall_clock64: 100 MHz
    hipLaunchKernelGGL(k_ipc_allreduce, dim3(1), dim3(256), 0, (hipStream_t)stream, inout, n, peers, rank, world, channel, status, ticks);
    return check_launch("ipc_allreduce");
}
