// devmem.hip -- allocation layer of libsdvgn with two debugging modes (see devmem.hpp).
#include "devmem.hpp"

#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace sdvgn {
namespace gmem {
namespace {

struct Rec {
    int kind;            // 1 = fenced device mapping, 2 = banded device buffer, 3 = fenced pinned host buffer
    void* base;          // start of the reservation / hipMalloc block / mmap block
    size_t mapped;       // bytes with storage behind them
    size_t reserved;     // bytes of address space (kind 1, 3)
    size_t bytes;        // what the caller asked for
    hipMemGenericAllocationHandle_t handle;
    const char* tag;
    void* res = nullptr; // kind 1: start of the address reservation (front guard)
};
std::mutex g_mu;
std::unordered_map<void*, Rec> g_recs;
unsigned long long g_violations = 0;
constexpr size_t kBand = 4096;
constexpr unsigned char kPoison = 0xA5;

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
int env_int(const char* name, int dflt) { const char* s = getenv(name); return s ? atoi(s) : dflt; }
bool log_on() { static const bool on = env_int("SDVGN_GUARD_LOG", 0) != 0; return on; }
size_t guard_align() { static const size_t a = (size_t)std::max(1, env_int("SDVGN_GUARD_ALIGN", 16)); return a; }

void log_alloc(const char* what, const Rec& r, void* user) {
    if (!log_on()) return;
    fprintf(stderr, "[sdvgn guard] %s %-40s [%p, %p) %zu bytes; storage [%p, %p), fence behind\n", what, r.tag, user, (char*)user + r.bytes, r.bytes,
            r.base, (char*)r.base + r.mapped);
}

hipError_t fenced_device_alloc(void** p, size_t bytes, size_t align, const char* tag) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemAllocationProp prop;
    std::memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if ((e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum)) != hipSuccess) return e;
    if (gran == 0) gran = 1 << 21;
    const size_t a = std::max(align, guard_align());
    // [front guard | mapping | back guard]: both guards stay reserved and unmapped.  SDVGN_GUARD_PAD_MB (default 64) is their size -- an
    // overrun must not be able to jump over the fence into the next buffer's mapping; SDVGN_GUARD_SIDE=front puts the buffer at the START of
    // its mapping instead (reads / writes BEFORE the start of a buffer fault), default back (past the end)
    static const size_t pad_mb = (size_t)std::max(0, env_int("SDVGN_GUARD_PAD_MB", 64));
    static const bool front = getenv("SDVGN_GUARD_SIDE") && !strcmp(getenv("SDVGN_GUARD_SIDE"), "front");
    const size_t guard = std::max(gran, round_up(pad_mb << 20, gran));
    const size_t need = round_up(bytes ? bytes : 1, a), mapped = round_up(need, gran), reserved = guard + mapped + guard;
    void* res = nullptr;
    if ((e = hipMemAddressReserve(&res, reserved, gran, nullptr, 0)) != hipSuccess) return e;
    void* va = (char*)res + guard;
    hipMemGenericAllocationHandle_t h;
    if ((e = hipMemCreate(&h, mapped, &prop, 0)) != hipSuccess) { hipMemAddressFree(res, reserved); return e; }
    if ((e = hipMemMap(va, mapped, 0, h, 0)) != hipSuccess) { hipMemRelease(h); hipMemAddressFree(res, reserved); return e; }
    hipMemAccessDesc desc;
    std::memset(&desc, 0, sizeof(desc));
    desc.location.type = hipMemLocationTypeDevice;
    desc.location.id = dev;
    desc.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemSetAccess(va, mapped, &desc, 1)) != hipSuccess) { hipMemUnmap(va, mapped); hipMemRelease(h); hipMemAddressFree(res, reserved); return e; }
    void* user = front ? va : (void*)((char*)va + (mapped - need));
    Rec r{1, va, mapped, reserved, bytes, h, tag};
    r.res = res;
    { std::lock_guard<std::mutex> lk(g_mu); g_recs[user] = r; }
    log_alloc("device", r, user);
    *p = user;
    return hipSuccess;
}

hipError_t banded_device_alloc(void** p, size_t bytes, const char* tag) {
    char* base = nullptr;
    const size_t body = round_up(bytes ? bytes : 1, 256);
    hipError_t e = hipMalloc((void**)&base, body + 2 * kBand);
    if (e != hipSuccess) return e;
    if ((e = hipMemset(base, kPoison, kBand)) != hipSuccess || (e = hipMemset(base + kBand + bytes, kPoison, body - bytes + kBand)) != hipSuccess) { hipFree(base); return e; }
    hipDeviceSynchronize();
    void* user = base + kBand;
    Rec r{2, base, body + 2 * kBand, 0, bytes, {}, tag};
    { std::lock_guard<std::mutex> lk(g_mu); g_recs[user] = r; }
    log_alloc("device(banded)", r, user);
    *p = user;
    return hipSuccess;
}

void check_bands(const Rec& r, void* user) {
    const size_t tail = r.mapped - kBand - r.bytes;
    std::vector<unsigned char> head(kBand), back(tail);
    hipDeviceSynchronize();
    if (hipMemcpy(head.data(), r.base, kBand, hipMemcpyDeviceToHost) != hipSuccess) return;
    if (hipMemcpy(back.data(), (char*)user + r.bytes, tail, hipMemcpyDeviceToHost) != hipSuccess) return;
    long first_head = -1, first_back = -1;
    for (size_t i = 0; i < kBand; ++i) if (head[i] != kPoison) { first_head = (long)i; break; }
    for (size_t i = 0; i < tail; ++i) if (back[i] != kPoison) { first_back = (long)i; break; }
    if (first_head >= 0 || first_back >= 0) {
        std::lock_guard<std::mutex> lk(g_mu);
        ++g_violations;
        fprintf(stderr, "[sdvgn guard] VIOLATION at %s (%zu bytes): %s%s (head byte %ld from the start of the band, tail byte %ld past the end)\n", r.tag, r.bytes,
                first_head >= 0 ? "store BEFORE the buffer " : "", first_back >= 0 ? "store PAST the end of the buffer" : "", first_head, first_back);
    }
}

}  // namespace

int guard_mode() { static const int m = env_int("SDVGN_GUARD", 0); return m; }
unsigned long long guard_violations() { std::lock_guard<std::mutex> lk(g_mu); return g_violations; }

// SDVGN_ALLOC_FILL=<0..255>: every new device buffer is filled with that byte (any mode) -- 255 turns every float that is read before
// it was written into a NaN, so a result that depends on uninitialised memory shows in the first test that computes with it
static int alloc_fill() { static const int f = env_int("SDVGN_ALLOC_FILL", -1); return f; }
// SDVGN_FREE_POISON=1 (any mode): a buffer is filled with 0xFF (every float a NaN) before it is given back, so a kernel that still reads it
// AFTER the free computes NaNs instead of plausible stale numbers.  Mode 0 keeps a size table for that (only when the switch is on).
static bool free_poison() { static const bool on = env_int("SDVGN_FREE_POISON", 0) != 0; return on; }
static bool guard_nofree() { static const bool on = env_int("SDVGN_GUARD_NOFREE", 0) != 0; return on; }
// SDVGN_GUARD_QUARANTINE=1 (mode 1): a freed buffer's pages are unmapped and released but its ADDRESS RANGE is never handed back
// (hipMemAddressFree is skipped), so no later allocation -- the library's, torch's, the runtime's -- can land there: a stale pointer
// faults at its first use, in the kernel that uses it, on every run.  (Without it a freed range is typically re-used by the very next
// reservation of the same size and the stale access silently reads or writes the new owner's data.)
static bool guard_quarantine() { static const bool on = env_int("SDVGN_GUARD_QUARANTINE", 0) != 0; return on; }
std::unordered_map<void*, size_t> g_plain_sizes;   // mode 0 + SDVGN_FREE_POISON only

hipError_t dmalloc_impl(void** p, size_t bytes, size_t align, const char* tag) {
    const int m = guard_mode();
    hipError_t e;
    if (m == 1) e = fenced_device_alloc(p, bytes, align, tag);
    else if (m == 2) e = banded_device_alloc(p, bytes, tag);
    else {
        e = hipMalloc(p, bytes ? bytes : 1);
        if (e == hipSuccess && free_poison()) { std::lock_guard<std::mutex> lk(g_mu); g_plain_sizes[*p] = bytes ? bytes : 1; }
    }
    if (e == hipSuccess && alloc_fill() >= 0) {
        void* from = *p; size_t n = bytes ? bytes : 1;
        if (m == 1) {   // the slack in front of an end-aligned buffer too: a read BEFORE the start of a buffer then shows like an uninitialised one
            std::lock_guard<std::mutex> lk(g_mu);
            const Rec& r = g_recs[*p];
            from = r.base; n = r.mapped;
        }
        e = hipMemset(from, alloc_fill() & 255, n);
        hipDeviceSynchronize();
    }
    return e;
}

hipError_t dfree(void* p) {
    if (!p) return hipSuccess;
    if (guard_mode() == 0) {
        if (free_poison()) {
            size_t n = 0;
            { std::lock_guard<std::mutex> lk(g_mu); auto it = g_plain_sizes.find(p); if (it != g_plain_sizes.end()) { n = it->second; g_plain_sizes.erase(it); } }
            if (n) { hipDeviceSynchronize(); hipMemset(p, 0xFF, n); hipDeviceSynchronize(); }
        }
        return hipFree(p);
    }
    Rec r;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_recs.find(p);
        if (it == g_recs.end()) return hipFree(p);
        r = it->second;
        g_recs.erase(it);
    }
    if (log_on()) fprintf(stderr, "[sdvgn guard] free   %-40s [%p, %p) %zu bytes\n", r.tag, p, (char*)p + r.bytes, r.bytes);
    if (r.kind == 2) { check_bands(r, p); if (free_poison()) { hipMemset(p, 0xFF, r.bytes); hipDeviceSynchronize(); } return hipFree(r.base); }
    hipDeviceSynchronize();
    if (free_poison()) { hipMemset(r.base, 0xFF, r.mapped); hipDeviceSynchronize(); }
    if (guard_nofree()) return hipSuccess;   // experiment: never give a fenced mapping back (no reuse of its address range or pages)
    hipError_t e = hipMemUnmap(r.base, r.mapped);
    hipMemRelease(r.handle);
    if (!guard_quarantine()) hipMemAddressFree(r.res, r.reserved);
    return e;
}

static bool host_fence_on() { static const bool on = guard_mode() == 1 && env_int("SDVGN_GUARD_HOST", 1) != 0; return on; }
hipError_t hmalloc_impl(void** p, size_t bytes, const char* tag) {
    if (!host_fence_on()) return hipHostMalloc(p, bytes ? bytes : 1);
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    const size_t need = round_up(bytes ? bytes : 1, std::max<size_t>(16, guard_align())), mapped = round_up(need, page), reserved = mapped + page;
    void* base = mmap(nullptr, reserved, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (base == MAP_FAILED) return hipErrorOutOfMemory;
    mprotect((char*)base + mapped, page, PROT_NONE);
    std::memset(base, 0, mapped);
    hipError_t e = hipHostRegister(base, mapped, hipHostRegisterDefault);
    if (e != hipSuccess) { munmap(base, reserved); return e; }
    void* user = (char*)base + (mapped - need);
    Rec r{3, base, mapped, reserved, bytes, {}, tag};
    { std::lock_guard<std::mutex> lk(g_mu); g_recs[user] = r; }
    log_alloc("pinned host", r, user);
    *p = user;
    return hipSuccess;
}

hipError_t hfree(void* p) {
    if (!p) return hipSuccess;
    if (!host_fence_on()) return hipHostFree(p);
    Rec r;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_recs.find(p);
        if (it == g_recs.end()) return hipHostFree(p);
        r = it->second;
        g_recs.erase(it);
    }
    hipDeviceSynchronize();
    hipError_t e = hipHostUnregister(r.base);
    munmap(r.base, r.reserved);
    return e;
}

}  // namespace gmem
}  // namespace sdvgn

__global__ void k_debug_peek(const double* p, double* out) { *out = *p; }
// the practical HBM peak of this box: float4 copy kernels (MI355X_MICROARCH.md: 6.29 TB/s measured of the 8 TB/s spec), three shapes
typedef float copy_f4 __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) k_debug_copy4(const copy_f4* __restrict__ src, copy_f4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        copy_f4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) v[k] = NT ? __builtin_nontemporal_load(&src[i + k * stride]) : src[i + k * stride];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) { if (NT) __builtin_nontemporal_store(v[k], &dst[i + k * stride]); else dst[i + k * stride] = v[k]; }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
extern "C" {
// tools/probe_fence.py: one 8-byte device load from an arbitrary address (a stray one ends the process with the runtime's memory access fault)
int sdvgn_debug_peek(const void* p, double* value_out) {
    double* pin = nullptr;
    if (hipHostMalloc((void**)&pin, 8) != hipSuccess) return -1;
    *pin = 0;
    k_debug_peek<<<1, 1>>>((const double*)p, pin);
    const hipError_t e = hipDeviceSynchronize();
    if (value_out) *value_out = *pin;
    hipHostFree(pin);
    return e == hipSuccess ? 0 : -(int)e;
}
// bench: bytes read + bytes written per second (GB/s) of a float4 copy kernel over `bytes` bytes each way, `reps` launches timed with HIP events
double sdvgn_debug_copy_rate(size_t bytes, int reps) {
    copy_f4 *a = nullptr, *b = nullptr;
    const size_t n = bytes / 16;
    if (hipMalloc((void**)&a, n * 16) != hipSuccess || hipMalloc((void**)&b, n * 16) != hipSuccess) { if (a) hipFree(a); return -1.0; }
    hipMemset(a, 1, n * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double best = -1.0;
    for (int variant = 0; variant < 4; ++variant) {
        const unsigned grid = variant == 3 ? 256 * 8 : 256 * 32;              // workgroups per CU x 256 CUs, grid-stride
        auto launch = [&]() {
            if (variant == 0) k_debug_copy4<1, false><<<grid, 256>>>(a, b, n);
            else if (variant == 1) k_debug_copy4<4, false><<<grid, 256>>>(a, b, n);
            else if (variant == 2) k_debug_copy4<4, true><<<grid, 256>>>(a, b, n);
            else k_debug_copy4<8, false><<<grid, 256>>>(a, b, n);
        };
        launch();
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms > 0) best = std::max(best, 2.0 * (double)(n * 16) * reps / (ms * 1e-3) / 1e9);
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(a); hipFree(b);
    return best;
}
// test rigs: caller-owned device buffers from the same fenced / banded / poisoned allocator (sdv-loam_amd/parallel.py, SDVGN_FENCE_EXTERNAL=1),
// so that the buffers a caller hands to sdvgn_ef_set_external_buffers / _set_collective_buffer sit behind the same instruments as the library's own
void* sdvgn_debug_dmalloc(size_t bytes) { void* p = nullptr; return sdvgn::gmem::dmalloc_impl(&p, bytes, 16, "caller-owned (sdvgn_debug_dmalloc)") == hipSuccess ? p : nullptr; }
int sdvgn_debug_dfree(void* p) { return (int)sdvgn::gmem::dfree(p); }
int sdvgn_debug_guard_mode(void) { return sdvgn::gmem::guard_mode(); }
unsigned long long sdvgn_debug_guard_violations(void) { return sdvgn::gmem::guard_violations(); }
}
