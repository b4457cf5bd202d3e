// zr_comm.cpp — the frame's one collective for C++ hosts: ncclReduce of the double3 accumulator over RCCL / xGMI.
// librccl.so is resolved lazily with dlopen so that libzr_hip.so itself has no RCCL dependency (single-GPU hosts and
// the Python host, which reduces through torch.distributed, never load it).
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/zr_capi.h"

extern "C" int zr_internal_fail(int code, const char* msg);   // zr_host.cpp: sets zr_last_error()
extern "C" int zr_internal_device(const zr_ctx*);

struct Id128 { unsigned char b[ZR_COMM_ID_BYTES]; };  // ncclUniqueId: 128 opaque bytes, passed by value
namespace {
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*Reduce)(const void*, void*, size_t, int, int, int, void*, void*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
bool load_rccl() {
    if (g_rccl.lib) return true;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) { g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (g_rccl.lib) break; }
    if (!g_rccl.lib) return false;
    g_rccl.GetUniqueId = (int (*)(void*))dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, Id128, int))dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.Reduce = (int (*)(const void*, void*, size_t, int, int, int, void*, void*))dlsym(g_rccl.lib, "ncclReduce");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(g_rccl.lib, "ncclGetErrorString");
    return g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.Reduce && g_rccl.CommDestroy;
}
int nccl_fail(const char* what, int rc) {
    std::string m = std::string(what) + " failed: " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
    return zr_internal_fail(ZR_E_DEVICE, m.c_str());
}
}  // namespace

struct zr_comm { void* comm = nullptr; int device = 0; int nranks = 1, rank = 0; };

extern "C" {

int zr_comm_unique_id(unsigned char id[ZR_COMM_ID_BYTES]) {
    if (!id) return zr_internal_fail(ZR_E_INVALID, "null id");
    if (!load_rccl()) return zr_internal_fail(ZR_E_DEVICE, "librccl.so could not be loaded");
    int rc = g_rccl.GetUniqueId(id);
    return rc == 0 ? ZR_OK : nccl_fail("ncclGetUniqueId", rc);
}

zr_comm* zr_comm_create(zr_ctx* ctx, int nranks, int rank, const unsigned char id[ZR_COMM_ID_BYTES]) {
    if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) { zr_internal_fail(ZR_E_INVALID, "bad communicator arguments"); return nullptr; }
    if (!load_rccl()) { zr_internal_fail(ZR_E_DEVICE, "librccl.so could not be loaded"); return nullptr; }
    zr_comm* c = new zr_comm();
    c->device = zr_internal_device(ctx); c->nranks = nranks; c->rank = rank;
    if (hipSetDevice(c->device) != hipSuccess) { zr_internal_fail(ZR_E_DEVICE, "hipSetDevice failed"); delete c; return nullptr; }
    Id128 uid; std::memcpy(uid.b, id, ZR_COMM_ID_BYTES);
    int rc = g_rccl.CommInitRank(&c->comm, nranks, uid, rank);
    if (rc != 0) { nccl_fail("ncclCommInitRank", rc); delete c; return nullptr; }
    return c;
}

int zr_comm_reduce_frame(zr_comm* c, void* d_frame, size_t n_doubles, int root, void* hip_stream) {
    if (!c || !d_frame) return zr_internal_fail(ZR_E_INVALID, "null argument");
    if (root < 0 || root >= c->nranks) return zr_internal_fail(ZR_E_INVALID, "root out of range");
    if (hipSetDevice(c->device) != hipSuccess) return zr_internal_fail(ZR_E_DEVICE, "hipSetDevice failed");
    const int ncclDouble = 8, ncclSum = 0;
    int rc = g_rccl.Reduce(d_frame, d_frame, n_doubles, ncclDouble, ncclSum, root, c->comm, hip_stream);
    return rc == 0 ? ZR_OK : nccl_fail("ncclReduce", rc);
}

void zr_comm_destroy(zr_comm* c) {
    if (!c) return;
    if (c->comm && g_rccl.CommDestroy) { (void)hipSetDevice(c->device); (void)g_rccl.CommDestroy(c->comm); }
    delete c;
}

}  // extern "C"
