// zr_comm.cpp — the frame's one collective for C++ hosts, over RCCL / xGMI: every rank packs the pixels of the tiles it owns
// (1/nranks of the double3 frame) and SENDS them to the root, which receives the nranks - 1 shares (one grouped
// ncclSend / ncclRecv exchange: a gather — only the root needs the frame, so each share crosses one xGMI link once and nothing is
// delivered to ranks that would drop it, as round 2's all-gather did) and scatters them into its frame (zr_comm_gather_frame);
// the whole-frame ncclReduce of round 1 stays available (zr_comm_reduce_frame).
// librccl.so is resolved lazily with dlopen so that libzr_hip.so itself has no RCCL dependency (single-GPU hosts and
// the Python host, which exchanges through torch.distributed, never load it).
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

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
    int (*Send)(const void*, size_t, int, int, void*, void*) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
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
    g_rccl.Send = (int (*)(const void*, size_t, int, int, void*, void*))dlsym(g_rccl.lib, "ncclSend");
    g_rccl.Recv = (int (*)(void*, size_t, int, int, void*, void*))dlsym(g_rccl.lib, "ncclRecv");
    g_rccl.GroupStart = (int (*)())dlsym(g_rccl.lib, "ncclGroupStart");
    g_rccl.GroupEnd = (int (*)())dlsym(g_rccl.lib, "ncclGroupEnd");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(g_rccl.lib, "ncclGetErrorString");
    return g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.Reduce && g_rccl.Send && g_rccl.Recv && g_rccl.GroupStart && g_rccl.GroupEnd && g_rccl.CommDestroy;
}
int nccl_fail(const char* what, int rc) {
    std::string m = std::string(what) + " failed: " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
    return zr_internal_fail(ZR_E_DEVICE, m.c_str());
}
}  // namespace

// device side of the exchange (zr_post.hip): frame pixels <-> a rank's packed tile list
namespace zr {
hipError_t launch_pack_tiles(const double* frame, const uint32_t* idx, size_t n, double* packed, hipStream_t stream);
hipError_t launch_unpack_tiles(double* frame, const uint32_t* idx, size_t n, const double* packed, hipStream_t stream);
}

struct zr_comm {
    void* comm = nullptr; int device = 0; int nranks = 1, rank = 0;
    // packed-tile exchange: per rank the flat pixel indices (y * W + x, ascending) of the tiles it owns, cached per frame geometry
    int W = 0, H = 0, tile = 0, skew = 0, root_built = 0;
    std::vector<size_t> first, count;     // rank r owns idx[first[r] .. first[r] + count[r])
    size_t share = 0;                     // max over ranks of count[r]: the stride of the root's receive slots
    uint32_t* d_idx = nullptr; double* d_packed = nullptr; double* d_all = nullptr;
    void release() { if (d_idx) (void)hipFree(d_idx); if (d_packed) (void)hipFree(d_packed); if (d_all) (void)hipFree(d_all); d_idx = nullptr; d_packed = d_all = nullptr; }
};

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

int zr_comm_gather_frame(zr_comm* c, void* d_frame, int W, int H, const zr_region* region, int root, void* hip_stream) {
    if (!c || !d_frame) return zr_internal_fail(ZR_E_INVALID, "null argument");
    if (W < 1 || H < 1 || (size_t)W * H > 0xFFFFFFFFull) return zr_internal_fail(ZR_E_INVALID, "bad frame size");
    if (root < 0 || root >= c->nranks) return zr_internal_fail(ZR_E_INVALID, "root out of range");
    // The exchange moves tile t of the whole frame from rank t % nranks: the region this rank rendered with must be exactly that
    // partition, or the root's pixels would be overwritten by pixels nobody rendered.
    int ts = 32, skew = 0;
    if (region) {
        skew = region->tile_skew > 0 ? region->tile_skew : 0;
        if (region->tile_size > 0) ts = region->tile_size;
        const bool whole = (region->w <= 0 || region->h <= 0) || (region->x0 == 0 && region->y0 == 0 && region->w == W && region->h == H);
        const int mod = region->tile_mod > 1 ? region->tile_mod : 1, rem = region->tile_mod > 1 ? region->tile_rem : 0;
        if (!whole || mod != c->nranks || rem != c->rank)
            return zr_internal_fail(ZR_E_INVALID, "zr_comm_gather_frame: the region must be the whole frame with tile_mod = nranks and tile_rem = rank (the partition the exchange assumes)");
    } else if (c->nranks != 1) return zr_internal_fail(ZR_E_INVALID, "zr_comm_gather_frame: pass the zr_region this rank rendered with");
    if (ts > 1024) return zr_internal_fail(ZR_E_INVALID, "bad tile size");
    if (hipSetDevice(c->device) != hipSuccess) return zr_internal_fail(ZR_E_DEVICE, "hipSetDevice failed");
    hipStream_t st = (hipStream_t)hip_stream;
    if (W != c->W || H != c->H || ts != c->tile || skew != c->skew) {   // (re)build the ownership lists: tile t = (y / ts) * tiles_x + x / ts belongs to rank t % nranks
        c->release();
        const int tiles_x = (W + ts - 1) / ts;
        std::vector<std::vector<uint32_t>> own((size_t)c->nranks);
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const int tx = x / ts, ty = y / ts;
                const int part = skew > 0 ? (int)(((long long)tx + (long long)skew * ty) % c->nranks) : (ty * tiles_x + tx) % c->nranks;
                own[(size_t)part].push_back((uint32_t)((size_t)y * W + x));
            }
        c->first.assign((size_t)c->nranks, 0); c->count.assign((size_t)c->nranks, 0); c->share = 0;
        std::vector<uint32_t> flat; flat.reserve((size_t)W * H);
        for (int r = 0; r < c->nranks; r++) {
            c->first[r] = flat.size(); c->count[r] = own[r].size();
            flat.insert(flat.end(), own[r].begin(), own[r].end());
            if (own[r].size() > c->share) c->share = own[r].size();
        }
        const size_t all = c->rank == root ? (size_t)c->nranks : 1;   // only the root receives
        if (hipMalloc((void**)&c->d_idx, std::max<size_t>(flat.size(), 1) * sizeof(uint32_t)) != hipSuccess ||
            hipMalloc((void**)&c->d_packed, std::max<size_t>(c->share, 1) * 3 * sizeof(double)) != hipSuccess ||
            hipMalloc((void**)&c->d_all, std::max<size_t>(c->share, 1) * 3 * sizeof(double) * all) != hipSuccess ||
            hipMemcpyAsync(c->d_idx, flat.data(), flat.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) {
            c->release(); c->W = c->H = c->tile = 0;
            return zr_internal_fail(ZR_E_DEVICE, "zr_comm_gather_frame: out of device memory");
        }
        c->W = W; c->H = H; c->tile = ts; c->skew = skew; c->root_built = root;
    }
    if (c->root_built != root) { c->W = c->H = c->tile = 0; return zr_internal_fail(ZR_E_INVALID, "zr_comm_gather_frame: the root changed between calls on one communicator (destroy and recreate it)"); }
    const int ncclDouble = 8;
    int rc;
    if (c->share != 0 && c->nranks == 1 && std::getenv("ZR_COMM_SELF_EXCHANGE")) {
        // Rehearsal of the exchange on a one-GPU box: the whole path a non-root rank and the root take together — pack -> ncclSend / ncclRecv
        // (to itself, inside one group) -> unpack — with the frame poisoned in between, so that what comes back can only have come through RCCL
        hipError_t e = zr::launch_pack_tiles((const double*)d_frame, c->d_idx + c->first[0], c->count[0], c->d_packed, st);
        if (e != hipSuccess) return zr_internal_fail(ZR_E_DEVICE, hipGetErrorString(e));
        if (hipMemsetAsync(d_frame, 0xFF, (size_t)W * H * 3 * sizeof(double), st) != hipSuccess) return zr_internal_fail(ZR_E_DEVICE, "hipMemsetAsync failed");
        if ((rc = g_rccl.GroupStart()) != 0) return nccl_fail("ncclGroupStart", rc);
        if ((rc = g_rccl.Send(c->d_packed, c->count[0] * 3, ncclDouble, 0, c->comm, hip_stream)) != 0) { (void)g_rccl.GroupEnd(); return nccl_fail("ncclSend", rc); }
        if ((rc = g_rccl.Recv(c->d_all, c->count[0] * 3, ncclDouble, 0, c->comm, hip_stream)) != 0) { (void)g_rccl.GroupEnd(); return nccl_fail("ncclRecv", rc); }
        if ((rc = g_rccl.GroupEnd()) != 0) return nccl_fail("ncclGroupEnd", rc);
        e = zr::launch_unpack_tiles((double*)d_frame, c->d_idx + c->first[0], c->count[0], c->d_all, st);
        if (e != hipSuccess) return zr_internal_fail(ZR_E_DEVICE, hipGetErrorString(e));
        return ZR_OK;
    }
    if (c->share == 0 || c->nranks == 1) return ZR_OK;
    if (c->rank != root) {
        hipError_t e = zr::launch_pack_tiles((const double*)d_frame, c->d_idx + c->first[c->rank], c->count[c->rank], c->d_packed, st);
        if (e != hipSuccess) return zr_internal_fail(ZR_E_DEVICE, hipGetErrorString(e));
        if (c->count[c->rank] == 0) return ZR_OK;
        if ((rc = g_rccl.Send(c->d_packed, c->count[c->rank] * 3, ncclDouble, root, c->comm, hip_stream)) != 0) return nccl_fail("ncclSend", rc);
        return ZR_OK;
    }
    if ((rc = g_rccl.GroupStart()) != 0) return nccl_fail("ncclGroupStart", rc);
    for (int r = 0; r < c->nranks; r++) {
        if (r == c->rank || c->count[r] == 0) continue;
        if ((rc = g_rccl.Recv(c->d_all + (size_t)r * c->share * 3, c->count[r] * 3, ncclDouble, r, c->comm, hip_stream)) != 0) { (void)g_rccl.GroupEnd(); return nccl_fail("ncclRecv", rc); }
    }
    if ((rc = g_rccl.GroupEnd()) != 0) return nccl_fail("ncclGroupEnd", rc);
    for (int r = 0; r < c->nranks; r++) {
        if (r == c->rank || c->count[r] == 0) continue;
        hipError_t e = zr::launch_unpack_tiles((double*)d_frame, c->d_idx + c->first[r], c->count[r], c->d_all + (size_t)r * c->share * 3, st);
        if (e != hipSuccess) return zr_internal_fail(ZR_E_DEVICE, hipGetErrorString(e));
    }
    return ZR_OK;
}

void zr_comm_destroy(zr_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    c->release();
    if (c->comm && g_rccl.CommDestroy) { (void)hipSetDevice(c->device); (void)g_rccl.CommDestroy(c->comm); }
    delete c;
}

}  // extern "C"
