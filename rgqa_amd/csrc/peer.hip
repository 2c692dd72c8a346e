// Peer-to-peer exchange over hipIpc buffers: the hand-written fallback for RCCL's all-to-all / all-gather in the data-parallel gradient exchange
// (replaces what nn.DataParallel's gather / reduce does at lxrt/entry.py:102-103 when the library's collectives do not drive
// all seven xGMI links of a GPU - bench.py's `dp_wire` probe decides).
//
// Every rank (one process per GPU) owns ONE staging buffer, allocated here as fine-grained device memory (coherent for peers: a rank's kernel may read
// what another rank's finished kernel wrote without an L2 write-back protocol of its own), exported as a hipIpc handle and mapped by every other
// rank.  Data moves by PULL: rgqa_peer_pull copies the same byte range out of every rank's staging buffer into a local destination with ONE launch
// whose workgroups are dealt over the peers, so a rank reads from all of its peers - all seven links on the xGMI mesh - at once:
//   reduce-scatter   every rank stages its payload (part r for rank r); barrier; rank r pulls part r from every rank and adds the W parts in rank order
//                    (rgqa_sum_parts: deterministic);
//   all-gather       every rank stages the range it owns; barrier; every rank pulls every range.
// Ordering between ranks is NOT done here and never by a spinning kernel: the caller brackets staging and pulls with a stream-ordered barrier (one
// tiny RCCL all-reduce on the exchange's stream; rgqa_amd/parallel.py PeerShardedExchange), so a rank that is late delays its peers' streams, it cannot
// wedge a GPU.
#include <vector>
#include <string.h>
#include "kernels.h"
#include "../../include/rgqa.h"

struct rgqa_peer_comm {
    int rank = 0, world = 1, device = 0;
    void* stage = nullptr; size_t bytes = 0;
    bool fine = false;
    hipIpcMemHandle_t handle;         // of `stage`, taken at creation
    std::vector<void*> peer;          // [world]: this process's mapping of rank r's staging buffer (peer[rank] = stage)
    std::vector<bool> opened;
};

#define PEER_MAX 16
struct PeerPtrs { const unsigned char* p[PEER_MAX]; };

// workgroup b serves peer (rank + 1 + b % world) % world - at any moment the ranks read from different peers - chunk b / world of that peer's range
__global__ __launch_bounds__(256) void peer_pull_kernel(PeerPtrs src, int rank, int world, size_t src_off, size_t bytes, unsigned char* __restrict__ dst, size_t dst_stride) {
    const int r = (rank + 1 + (int)(blockIdx.x % world)) % world;
    const size_t nb = gridDim.x / world, jb = blockIdx.x / world;
    const uint4* s = reinterpret_cast<const uint4*>(src.p[r] + src_off);
    uint4* d = reinterpret_cast<uint4*>(dst + (size_t)r * dst_stride);
    const size_t nv = bytes >> 4;
    // four 16-byte loads in flight per lane: a remote read is a fabric round trip
    const size_t stride = nb * 256;
    size_t i = jb * 256 + threadIdx.x;
    for (; i + 3 * stride < nv; i += 4 * stride) {
        const uint4 a = s[i], b = s[i + stride], c = s[i + 2 * stride], e = s[i + 3 * stride];
        d[i] = a; d[i + stride] = b; d[i + 2 * stride] = c; d[i + 3 * stride] = e;
    }
    for (; i < nv; i += stride) d[i] = s[i];
}

extern "C" {

int rgqa_peer_comm_create(int rank, int world, size_t stage_bytes, rgqa_peer_comm** out) {
    RGQA_REQUIRE(out != nullptr && world >= 1 && world <= PEER_MAX && rank >= 0 && rank < world && stage_bytes > 0, "peer_comm_create: rank %d / world %d / %zu bytes", rank, world, stage_bytes);
    rgqa_peer_comm* c = new rgqa_peer_comm();
    c->rank = rank; c->world = world; c->bytes = (stage_bytes + 255) & ~(size_t)255;
    if (rgqa_check_hip(hipGetDevice(&c->device), "hipGetDevice")) { delete c; return RGQA_ERR_HIP; }
    // fine-grained device memory: coherent for the peers that map it; plain hipMalloc memory if this runtime refuses the flag (then correct by the
    // caller's barriers alone as long as a kernel boundary lies between the writer and the reader, which is how the exchange uses it)
    if (hipExtMallocWithFlags(&c->stage, c->bytes, hipDeviceMallocFinegrained) == hipSuccess) {
        c->fine = true;
        if (hipIpcGetMemHandle(&c->handle, c->stage) != hipSuccess) {      // this runtime does not export fine-grained memory: plain device memory instead
            (void)hipGetLastError();
            (void)hipFree(c->stage);
            c->stage = nullptr; c->fine = false;
        }
    } else (void)hipGetLastError();
    if (c->stage == nullptr) {
        if (rgqa_check_hip(hipMalloc(&c->stage, c->bytes), "hipMalloc(staging buffer)")) { delete c; return RGQA_ERR_HIP; }
        if (rgqa_check_hip(hipIpcGetMemHandle(&c->handle, c->stage), "hipIpcGetMemHandle(staging buffer)")) { (void)hipFree(c->stage); delete c; return RGQA_ERR_HIP; }
    }
    c->peer.assign(world, nullptr); c->opened.assign(world, false);
    c->peer[rank] = c->stage;
    *out = c;
    return RGQA_OK;
}

int rgqa_peer_comm_stage(const rgqa_peer_comm* c, void** ptr, size_t* bytes) {
    RGQA_REQUIRE(c != nullptr && ptr != nullptr, "peer_comm_stage: null argument");
    *ptr = c->stage;
    if (bytes) *bytes = c->bytes;
    return RGQA_OK;
}

int rgqa_peer_comm_export(rgqa_peer_comm* c, void* handle) {
    RGQA_REQUIRE(c != nullptr && handle != nullptr, "peer_comm_export: null argument");
    static_assert(sizeof(hipIpcMemHandle_t) <= RGQA_PEER_HANDLE_BYTES, "handle size");
    memset(handle, 0, RGQA_PEER_HANDLE_BYTES);
    memcpy(handle, &c->handle, sizeof c->handle);
    return RGQA_OK;
}

int rgqa_peer_comm_connect(rgqa_peer_comm* c, const void* handles) {
    RGQA_REQUIRE(c != nullptr && handles != nullptr, "peer_comm_connect: null argument");
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank || c->opened[r]) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const unsigned char*)handles + (size_t)r * RGQA_PEER_HANDLE_BYTES, sizeof h);
        void* p = nullptr;
        RGQA_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        c->peer[r] = p; c->opened[r] = true;
    }
    return RGQA_OK;
}

void rgqa_peer_comm_destroy(rgqa_peer_comm* c) {
    if (c == nullptr) return;
    for (int r = 0; r < c->world; ++r)
        if (c->opened[r] && c->peer[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
    if (c->stage) (void)hipFree(c->stage);
    delete c;
}

int rgqa_peer_pull(rgqa_peer_comm* c, size_t src_off_bytes, size_t bytes, void* dst, size_t dst_stride_bytes, void* stream) {
    RGQA_REQUIRE(c != nullptr && dst != nullptr, "peer_pull: null argument");
    RGQA_REQUIRE((src_off_bytes % 16) == 0 && (bytes % 16) == 0 && (dst_stride_bytes % 16) == 0 && (((uintptr_t)dst) % 16) == 0, "peer_pull: offsets and sizes must be multiples of 16 bytes");
    RGQA_REQUIRE(src_off_bytes + bytes <= c->bytes, "peer_pull: [%zu, %zu) exceeds the %zu-byte staging buffers", src_off_bytes, src_off_bytes + bytes, c->bytes);
    if (bytes == 0) return RGQA_OK;
    PeerPtrs pp;
    for (int r = 0; r < PEER_MAX; ++r) pp.p[r] = nullptr;
    for (int r = 0; r < c->world; ++r) {
        RGQA_REQUIRE(c->peer[r] != nullptr, "peer_pull: rank %d's staging buffer is not mapped (peer_comm_connect)", r);
        pp.p[r] = reinterpret_cast<const unsigned char*>(c->peer[r]);
    }
    size_t per = (bytes / 16 + 1023) / 1024;                 // >= 4 x 16 bytes per lane
    if (per < 1) per = 1;
    if (per > 256) per = 256;                                  // <= 256 workgroups per peer: with 7 peers the chip is covered several times over
    hipLaunchKernelGGL(peer_pull_kernel, dim3((unsigned)(per * c->world)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pp, c->rank, c->world, src_off_bytes, bytes,
                       reinterpret_cast<unsigned char*>(dst), dst_stride_bytes);
    RGQA_LAUNCH_CHECK("peer_pull_kernel");
    return RGQA_OK;
}

int rgqa_peer_comm_is_fine_grained(const rgqa_peer_comm* c) { return (c != nullptr && c->fine) ? 1 : 0; }

}  // extern "C"
