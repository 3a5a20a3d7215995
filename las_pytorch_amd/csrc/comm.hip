// Data-parallel gradient exchange behind the C ABI: one RCCL communicator per device, ONE in-place all-reduce of the flat
// fp32 gradient per step (include/las_hip.h, las_comm_* / las_allreduce_f32).  Replaces nn.DataParallel's per-step
// parameter broadcast + output gather + gradient reduce, reference train.py:76-78.
//
// librccl is opened with dlopen on first use: liblas_hip.so has no LINK-time dependency on it (<rccl/rccl.h> is needed at build
// time for the types only), single-GPU users never load it, and inside a PyTorch process the already-loaded RCCL is reused
// (same SONAME).  xGMI is point-to-point, so the ring
// all-reduce of the 39.9 MB (P) gradient is per-link bound; one large message per step is the shape that suits it.
#include "../../include/las_hip.h"
#include "las_common.h"
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

using namespace las;

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

constexpr int MAX_DEVICES = 64;
std::mutex g_mu;                       // guards g_rccl and g_comm (per-device handles are the only global state)
Rccl g_rccl;
ncclComm_t g_comm[MAX_DEVICES] = {};
int g_world[MAX_DEVICES] = {};

int load_rccl() {                      // caller holds g_mu
    if (g_rccl.handle) return LAS_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) return fail(LAS_ERR_UNSUPPORTED, "cannot open librccl: %s", dlerror());
    Rccl r;
    r.handle = h;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy || !r.GetErrorString)
        return fail(LAS_ERR_UNSUPPORTED, "librccl lacks an expected symbol%s", "");
    g_rccl = r;
    return LAS_OK;
}

int nccl_fail(const char* what, ncclResult_t rc) {
    snprintf(g_err, sizeof(g_err), "%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return LAS_ERR_HIP;
}

}  // namespace

extern "C" {

int las_comm_uid(void* uid_out128) {
    LAS_REQUIRE(uid_out128 != nullptr, "uid buffer");
    static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id size");
    std::lock_guard<std::mutex> lk(g_mu);
    LAS_TRY(load_rccl());
    ncclUniqueId id;
    const ncclResult_t rc = g_rccl.GetUniqueId(&id);
    if (rc != ncclSuccess) return nccl_fail("ncclGetUniqueId", rc);
    memcpy(uid_out128, &id, sizeof(id));
    return LAS_OK;
}

int las_comm_init(int rank, int world, const void* uid128) {
    LAS_REQUIRE(uid128 != nullptr && world >= 1 && rank >= 0 && rank < world, "communicator arguments");
    int dev = 0;
    LAS_HIP_CHECK(hipGetDevice(&dev));
    LAS_REQUIRE(dev < MAX_DEVICES, "device index");
    ncclUniqueId id;
    memcpy(&id, uid128, sizeof(id));
    // the lock only guards the handle table: ncclCommInitRank blocks until every rank has arrived, and in a one-thread-per-device
    // process the peer thread must be able to enter this function meanwhile — never hold g_mu across a collective call
    ncclComm_t old = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        LAS_TRY(load_rccl());
        old = g_comm[dev]; g_comm[dev] = nullptr;
    }
    if (old) g_rccl.CommDestroy(old);
    ncclComm_t c = nullptr;
    const ncclResult_t rc = g_rccl.CommInitRank(&c, world, id, rank);
    if (rc != ncclSuccess) return nccl_fail("ncclCommInitRank", rc);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_comm[dev] = c;
        g_world[dev] = world;
    }
    return LAS_OK;
}

int las_allreduce_f32(float* buf, size_t count, int average, void* stream) {
    LAS_REQUIRE(buf != nullptr && count > 0, "all-reduce buffer");
    int dev = 0;
    LAS_HIP_CHECK(hipGetDevice(&dev));
    LAS_REQUIRE(dev < MAX_DEVICES, "device index");
    ncclComm_t c;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        c = g_comm[dev];
    }
    LAS_REQUIRE(c != nullptr, "las_comm_init has not been called on this device");
    const ncclResult_t rc = g_rccl.AllReduce(buf, buf, count, ncclFloat32, average ? ncclAvg : ncclSum, c, (hipStream_t)stream);
    if (rc != ncclSuccess) return nccl_fail("ncclAllReduce", rc);
    return LAS_OK;
}

int las_comm_destroy(void) {
    int dev = 0;
    LAS_HIP_CHECK(hipGetDevice(&dev));
    LAS_REQUIRE(dev < MAX_DEVICES, "device index");
    ncclComm_t c = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        c = g_comm[dev]; g_comm[dev] = nullptr;
    }
    if (c) {
        const ncclResult_t rc = g_rccl.CommDestroy(c);
        if (rc != ncclSuccess) return nccl_fail("ncclCommDestroy", rc);
    }
    return LAS_OK;
}

}  // extern "C"
