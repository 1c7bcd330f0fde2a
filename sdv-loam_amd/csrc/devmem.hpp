// devmem.hpp -- every device / pinned-host allocation of libsdvgn goes through these four functions.
//
// Default (SDVGN_GUARD unset or 0): plain hipMalloc / hipHostMalloc / hipFree / hipHostFree, nothing else.
//
// SDVGN_GUARD=1  "electric fence": a device allocation becomes its own virtual-memory mapping (hipMemAddressReserve + hipMemCreate +
//                hipMemMap) with the buffer placed so that it ENDS at the end of the mapping; the address range behind it stays reserved
//                and unmapped.  A kernel that loads or stores even one element past the end of a buffer takes a memory access fault at
//                once, on every run, instead of silently reading a neighbouring allocation (and faulting one run in N when the
//                neighbour happens to be the end of a mapping).  Pinned host buffers are mmap'ed + hipHostRegister'ed the same way
//                (end-aligned, PROT_NONE page behind).  SDVGN_GUARD_ALIGN (default 16) is the alignment kept for the start of a buffer:
//                over-reads shorter than that go unnoticed, so a second run with SDVGN_GUARD_ALIGN=4 closes the gap for 4-byte planes.
//                SDVGN_GUARD_HOST=0 keeps the pinned host buffers on hipHostMalloc (only device buffers fenced).
// SDVGN_GUARD=2  poisoned guard bands: every device buffer is over-allocated by 4 KiB on both sides, the bands are filled with 0xA5 and
//                checked when the buffer is freed (sdvgn_*_destroy) -- catches stores BEFORE the start / past the end without changing
//                the address-space layout.  Violations are counted (sdvgn_debug_guard_violations) and printed to stderr.
// SDVGN_ALLOC_FILL=<byte>  (any mode) fills every new device buffer with that byte: 255 makes every float read before it was written a NaN.
// SDVGN_GUARD_LOG=1  prints every allocation (tag, address range) to stderr, so that the address of a reported fault can be matched to
//                the buffer it lies behind.
//
// This is the instrument profiles/r04_fault_hunt.txt was produced with.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace sdvgn {
namespace gmem {

hipError_t dmalloc_impl(void** p, size_t bytes, size_t align, const char* tag);
hipError_t dfree(void* p);
hipError_t hmalloc_impl(void** p, size_t bytes, const char* tag);
hipError_t hfree(void* p);
int guard_mode();                       // 0 / 1 / 2, read once from SDVGN_GUARD
unsigned long long guard_violations();  // mode 2: bands found damaged so far (process-wide)

#define SDVGN_MEM_STR2(x) #x
#define SDVGN_MEM_STR(x) SDVGN_MEM_STR2(x)
#define SDVGN_MEM_TAG __FILE__ ":" SDVGN_MEM_STR(__LINE__)

template <typename T>
struct align_of_pointee { static constexpr size_t value = alignof(T); };
template <>
struct align_of_pointee<void> { static constexpr size_t value = 16; };

template <typename T>
inline hipError_t dmalloc_tagged(T** p, size_t bytes, const char* tag) {
    return dmalloc_impl((void**)p, bytes, align_of_pointee<T>::value, tag);
}
template <typename T>
inline hipError_t hmalloc_tagged(T** p, size_t bytes, const char* tag) {
    return hmalloc_impl((void**)p, bytes, tag);
}

}  // namespace gmem
}  // namespace sdvgn

// call sites read like the HIP calls they replace
#define SDVGN_DMALLOC(p, bytes) ::sdvgn::gmem::dmalloc_tagged((p), (bytes), SDVGN_MEM_TAG)
#define SDVGN_HMALLOC(p, bytes) ::sdvgn::gmem::hmalloc_tagged((p), (bytes), SDVGN_MEM_TAG)
#define SDVGN_DFREE(p) ::sdvgn::gmem::dfree((void*)(p))
#define SDVGN_HFREE(p) ::sdvgn::gmem::hfree((void*)(p))
