// gsvc_amd/csrc/api.cpp — error slot and version of libgsvc_hip.so.
#include "common.h"

#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace gsvc {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct ProfRec {
    std::string name;
    hipEvent_t a, b;
};
static bool g_prof_on = false;
static bool g_bwd_probe = false;
static bool g_deterministic = false;
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_pool;

bool profile_enabled() { return g_prof_on; }
bool bwd_probe_enabled() { return g_bwd_probe; }
bool deterministic() { return g_deterministic; }

static hipEvent_t get_event()
{
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

void profile_begin(const char *name, hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r{name, get_event(), get_event()};
    (void)hipEventRecord(r.a, s);
    g_prof.push_back(r);
}

void profile_end(hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof.empty()) (void)hipEventRecord(g_prof.back().b, s);
}

}  // namespace gsvc

extern "C" int gsvc_set_deterministic(int on)
{
    gsvc::g_deterministic = on != 0;
    return GSVC_OK;
}

extern "C" int gsvc_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(gsvc::g_prof_mu);
    gsvc::g_prof_on = (on & 1) != 0;
    gsvc::g_bwd_probe = (on & 2) != 0;
    for (auto &r : gsvc::g_prof) { gsvc::g_pool.push_back(r.a); gsvc::g_pool.push_back(r.b); }
    gsvc::g_prof.clear();
    return GSVC_OK;
}

extern "C" int gsvc_profile_collect(char *names, int32_t *launches, float *total_ms, int max_kernels)
{
    std::lock_guard<std::mutex> lk(gsvc::g_prof_mu);
    std::map<std::string, std::pair<int, double>> agg;
    std::vector<std::string> order;
    for (auto &r : gsvc::g_prof) {
        if (hipEventSynchronize(r.b) != hipSuccess) { gsvc::set_error("profile_collect: event sync failed"); return GSVC_E_LAUNCH; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        if (!agg.count(r.name)) order.push_back(r.name);
        auto &e = agg[r.name];
        e.first += 1;
        e.second += ms;
        gsvc::g_pool.push_back(r.a);
        gsvc::g_pool.push_back(r.b);
    }
    gsvc::g_prof.clear();
    int n = 0;
    for (auto &name : order) {
        if (n >= max_kernels) break;
        std::snprintf(names + 64 * n, 64, "%s", name.c_str());
        launches[n] = agg[name].first;
        total_ms[n] = (float)agg[name].second;
        n++;
    }
    return n;
}

extern "C" const char *gsvc_last_error(void) { return gsvc::g_err; }

extern "C" const char *gsvc_version(void) { return "gsvc_hip 0.1.0 gfx950"; }
