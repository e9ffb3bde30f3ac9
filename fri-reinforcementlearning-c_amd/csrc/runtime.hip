// runtime.hip -- device discovery, error reporting and argument checks of libfrirl_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "device_common.h"

namespace frirl_host {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

static Options g_opts;
static std::once_flag g_opts_once;
struct OptName { const char *name, *env; int Options::*field; int def; };
static const OptName k_opts[] = {
    {"no_uidx", "FRIRL_HIP_NO_UIDX", &Options::no_uidx, 0},
    {"rd_unroll", "FRIRL_HIP_RD_UNROLL", &Options::rd_unroll, 0},
    {"rd_chunk", "FRIRL_HIP_RD_CHUNK", &Options::rd_chunk, 0},
    {"rd_nt", "FRIRL_HIP_RD_NT", &Options::rd_nt, -1},
    {"rd_persist", "FRIRL_HIP_RD_PERSIST", &Options::rd_persist, -1},
    {"rd_order", "FRIRL_HIP_RD_ORDER", &Options::rd_order, 0},
    {"step_wave", "FRIRL_HIP_STEP_WAVE", &Options::step_wave, -1},
    {"step_track", "FRIRL_HIP_STEP_TRACK", &Options::step_track, -1},
    {"lanes_slices", "FRIRL_HIP_LANES_SLICES", &Options::lanes_slices, 0},
    {"lanes_wpe", "FRIRL_HIP_LANES_WPE", &Options::lanes_wpe, 0},
    {"rollout_group", "FRIRL_HIP_ROLLOUT_GROUP", &Options::rollout_group, 0},
    {"rollout_slices", "FRIRL_HIP_ROLLOUT_SLICES", &Options::rollout_slices, 0},
    {"rollout_resident", "FRIRL_HIP_ROLLOUT_RESIDENT", &Options::rollout_resident, -1},
    {"rollout_cap", "FRIRL_HIP_ROLLOUT_CAP", &Options::rollout_cap, 0},
    {"rollout_pair", "FRIRL_HIP_ROLLOUT_PAIR", &Options::rollout_pair, -1},
    {"rollout_wps", "FRIRL_HIP_ROLLOUT_WPS", &Options::rollout_wps, 0},
    {"learn_slices", "FRIRL_HIP_LEARN_SLICES", &Options::learn_slices, 0},
    {"learn_alone", "FRIRL_HIP_LEARN_ALONE", &Options::learn_alone, 0},
    {"learn_persistent", "FRIRL_HIP_LEARN_PERSISTENT", &Options::learn_persistent, -1},
    {"multi_loopback", "FRIRL_HIP_MULTI_LOOPBACK", &Options::multi_loopback, 0},
    {"no_many", "FRIRL_HIP_NO_MANY", &Options::no_many, 0},
    {"mirror_sync", "FRIRL_HIP_MIRROR_SYNC", &Options::mirror_sync, 0},
};
static void opts_init()
{
    for (const OptName &o : k_opts) {
        const char *e = getenv(o.env);
        g_opts.*(o.field) = e ? atoi(e) : o.def;
    }
}
const Options &opts()
{
    std::call_once(g_opts_once, opts_init);
    return g_opts;
}
static const OptName *find_opt(const char *name)
{
    if (name) for (const OptName &o : k_opts) if (!strcmp(o.name, name)) return &o;
    return nullptr;
}

int check_device()
{
    static thread_local int cached = 1;   // 1 = unknown
    if (cached != 1) return cached;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no usable HIP device (%s): the FRIRL hot path has no CPU fallback", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        (void)hipGetLastError();
        return FRIRL_HIP_ENODEV;          // not cached: a device may appear (tests without GPU stay ENODEV)
    }
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
        set_error("hipGetDeviceProperties failed");
        return FRIRL_HIP_ENODEV;
    }
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library is built for gfx950 (MI355X) only", dev, p.gcnArchName);
        return FRIRL_HIP_ENODEV;
    }
    cached = FRIRL_HIP_OK;
    return cached;
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return FRIRL_HIP_ELAUNCH;
    }
    return FRIRL_HIP_OK;
}

int check_tables(const frirl_hip_tables *t)
{
    if (!t || !t->u || !t->ve) { set_error("tables: NULL pointer"); return FRIRL_HIP_EINVAL; }
    if (t->nant < 1 || t->nant > FRIRL_HIP_MAX_NANT) { set_error("tables: nant=%d outside 1..%d", t->nant, FRIRL_HIP_MAX_NANT); return FRIRL_HIP_EINVAL; }
    if (t->U < 2) { set_error("tables: U=%d < 2", t->U); return FRIRL_HIP_EINVAL; }
    return FRIRL_HIP_OK;
}

int check_rulebases(const frirl_hip_tables *t, const frirl_hip_rulebases *b)
{
    int rc = check_tables(t);
    if (rc) return rc;
    if (!b || !b->rb || !b->nrules) { set_error("rulebases: NULL pointer"); return FRIRL_HIP_EINVAL; }
    if (b->E < 1) { set_error("rulebases: E=%d < 1", b->E); return FRIRL_HIP_EINVAL; }
    if (b->maxR < 2 || (b->maxR & 1)) { set_error("rulebases: maxR=%d must be even and >= 2 (16-byte vector loads)", b->maxR); return FRIRL_HIP_EINVAL; }
    if (reinterpret_cast<uintptr_t>(b->rb) & 15) { set_error("rulebases: rb must be 16-byte aligned"); return FRIRL_HIP_EINVAL; }
    return FRIRL_HIP_OK;
}

}  // namespace frirl_host

extern "C" {

const char *frirl_hip_version(void) { return "frirl-hip 0.1 (gfx950)"; }

const char *frirl_hip_last_error(void) { return frirl_host::g_err; }

int frirl_hip_set_option(const char *name, int value)
{
    const frirl_host::OptName *o = frirl_host::find_opt(name);
    if (!o) { frirl_host::set_error("frirl_hip_set_option: unknown option '%s'", name ? name : "(null)"); return FRIRL_HIP_EINVAL; }
    (void)frirl_host::opts();
    frirl_host::g_opts.*(o->field) = value;
    return FRIRL_HIP_OK;
}

int frirl_hip_get_option(const char *name, int *value)
{
    const frirl_host::OptName *o = frirl_host::find_opt(name);
    if (!o || !value) { frirl_host::set_error("frirl_hip_get_option: unknown option '%s'", name ? name : "(null)"); return FRIRL_HIP_EINVAL; }
    *value = frirl_host::opts().*(o->field);
    return FRIRL_HIP_OK;
}

int frirl_hip_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); frirl_host::set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return FRIRL_HIP_ENODEV; }
    return n;
}

int frirl_hip_device_info(int device, char *name, int name_len, int32_t *cus, int64_t *hbm_bytes)
{
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) { (void)hipGetLastError(); frirl_host::set_error("hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e)); return FRIRL_HIP_ENODEV; }
    if (name && name_len > 0) snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
    if (cus) *cus = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    return FRIRL_HIP_OK;
}

}  // extern "C"
