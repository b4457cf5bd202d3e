// zr_host.cpp — context life cycle and error reporting of the C ABI (include/zr_capi.h).  The scene side lives in zr_commit.cpp, the render side in
// zr_render.cpp; zr_host_internal.h holds what they share.
#include "zr_host_internal.h"

namespace zr_host {
namespace { thread_local std::string g_err; }
int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
const char* last_error() { return g_err.c_str(); }
double env_double(const char* name, double dflt) {
    const char* v = std::getenv(name);
    return v && *v ? std::atof(v) : dflt;
}
}  // namespace zr_host

extern "C" {

int zr_abi_version(void) { return ZR_ABI_VERSION; }
// helpers for zr_comm.cpp
int zr_internal_fail(int code, const char* msg) { return fail(code, "%s", msg); }
int zr_internal_device(const zr_ctx* c) { return c ? c->device : 0; }
const char* zr_last_error(void) { return zr_host::last_error(); }

zr_ctx* zr_create(int device_ordinal) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { fail(ZR_E_DEVICE, "no HIP device available (%s): this library has no CPU path", hipGetErrorString(e)); return nullptr; }
    if (device_ordinal < 0 || device_ordinal >= n) { fail(ZR_E_INVALID, "device ordinal %d out of range (0..%d)", device_ordinal, n - 1); return nullptr; }
    if ((e = hipSetDevice(device_ordinal)) != hipSuccess) { fail(ZR_E_DEVICE, "hipSetDevice: %s", hipGetErrorString(e)); return nullptr; }
    zr_ctx* c = new zr_ctx();
    c->device = device_ordinal;
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        fail(ZR_E_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e)); delete c; return nullptr;
    }
    for (int k = 1; k < ST_MAX_POOLS; k++)
        if ((e = hipStreamCreateWithFlags(&c->sub[k], hipStreamNonBlocking)) != hipSuccess) {
            fail(ZR_E_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e)); delete c; return nullptr;
        }
    if (c->d_ctr.alloc(48) != ZR_OK)   /* 16 counter words + 32 for development builds (ZR_WAVE_PROFILE: per-phase lane histograms) */ { delete c; return nullptr; }
    c->variant = (int)env_double("ZR_KERNEL", 2);
    if (c->variant != 2) c->variant = 0;   // (variant 1, the round-1 wave-scheduler megakernel, is retired: 2-6 x slower and nothing depended on it)
    c->log_kind = (int)env_double("ZR_TIMELOG_KIND", 1);
    if (c->variant == 2) {
        c->st_blocks = zr::stream_extend_blocks();
        int over = (int)env_double("ZR_ST_BLOCKS", 0);
        if (over > 0) c->st_blocks = over;
        // up to 128 Mi slots = 26.8 GB of path state (of 288 GB), cut into two sub-pools of 64 Mi for worlds whose lean kernels run side by side (render_stream):
        // a launch that carries twice the rays walks the tree 9 % cheaper per ray (round 2: cfg3 368 ms with 64 Mi in one pool against 392 with 32 Mi, nothing more
        // from 96 or 128 Mi in ONE pool; round 4, two sub-pools: 351 / 340 / 341 ms with 64 / 96 / 128 Mi, profiles/r4_experiments_ab.txt).  A frame uses
        // min(this, units / 8) slots, so small frames and a rank's share of a sharded frame stay small
        c->st_slots = (uint32_t)env_double("ZR_STREAM_SLOTS", 128.0 * 1024 * 1024);
        // dynamic work units of shard s are handed out only by SHADE blocks with blockIdx % 64 == s (zr_stream.hip): a pool needs at
        // least 64 SHADE blocks (64 x 256 slots) or the units of the unserved shards would never be rendered
        c->st_slots = std::max<uint32_t>(64u * 256u, c->st_slots / 256 * 256);
        c->st_pools = (int)env_double("ZR_STREAM_POOLS", -1);
        if (env_double("ZR_STREAM_OVERLAP", -1) == 0) c->st_pools = 1;
        if (c->d_ctl.alloc((ST_MAX_POOLS + 1) * zr::stream_ctl_words()) != ZR_OK ||
            hipEventCreateWithFlags(&c->st_event, hipEventDisableTiming) != hipSuccess ||
            hipHostMalloc((void**)&c->h_active, (ST_MAX_POOLS + 1) * zr::stream_ctl_words() * sizeof(unsigned int), 0) != hipSuccess) { fail(ZR_E_DEVICE, "variant-2 buffers: out of memory"); delete c; return nullptr; }
    }
    return c;
}

void zr_destroy(zr_ctx* c) {
    if (!c) return;
    c->drain_trash();
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); }
    for (hipEvent_t e : c->pool) (void)hipEventDestroy(e);
    for (auto& p : c->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    c->d_ctr.release(); c->d_out.release(); c->d_tiles.release();
    c->d_pool.release(); c->d_pixels.release(); c->d_partial.release(); c->d_kend.release(); c->d_cls.release(); c->d_cpart.release(); c->d_ctl.release(); c->d_st_overflow.release();
    if (c->h_active) (void)hipHostFree(c->h_active);
    if (c->st_event) (void)hipEventDestroy(c->st_event);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    for (int k = 1; k < ST_MAX_POOLS; k++) if (c->sub[k]) (void)hipStreamDestroy(c->sub[k]);
    delete c;
}

}  // extern "C"
