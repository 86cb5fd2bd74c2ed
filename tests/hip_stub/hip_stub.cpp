// hip_stub.cpp -- TEST INFRASTRUCTURE, never shipped: a synchronous in-memory stand-in for the HIP runtime calls and
// the kernel launchers that the HOST side of the library uses (kifs_api.cpp, kifs_schedule.cpp, kifs_shards.cpp,
// kifs_multi.cpp, kifs_host.cpp), so that those 2 500 lines of slot / stream / event / buffer bookkeeping can run under
// AddressSanitizer and UndefinedBehaviorSanitizer on a box without a GPU (the GPU pool does not
// offer sanitizers).  `make -C kifs_raymarching_amd/csrc asan` links it with the sanitised host objects into
// build/kifs_host_asan; tests/test_host_sanitizers.py runs that.
//
// What it models: "device memory" is malloc'ed host memory, registered so that hipPointerGetAttributes can tell it from
// host pointers (and a hipFree of something never allocated, or twice, is an error); every asynchronous call executes at
// once -- host call order is one valid serialisation of stream order, because a stream can only wait for an event that
// has already been recorded.  The render "kernel" writes a pattern that depends on the view's camera and the pixel's
// FRAME coordinates only, through the launch's own tile table, stripe table, pitch and in-place flag -- so a gathered
// multi-device frame must equal the frame one context renders, exactly as on the GPU -- and checks that the tile table
// is a permutation.  The sparse pack / unpack / fill / stripe kernels are restated on the CPU from their documented
// formats (kifs_support_kernels.hip:115-125,184-187,211).  stub_fail_in(n): the n-th HIP call or launch from now on
// fails once (hipErrorOutOfMemory / hipErrorLaunchFailure), for the error paths.
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <set>
#include <vector>

#include "../../kifs_raymarching_amd/csrc/kifs_internal.hpp"

struct ihipStream_t { int device; };
struct ihipEvent_t { int device; bool recorded; };

namespace {

int g_devices = [] { const char* e = std::getenv("KIFS_STUB_DEVICES"); return e ? std::atoi(e) : 4; }();
int g_current = 0;
hipError_t g_last = hipSuccess;
long g_fail_in = -1;  // countdown to an injected failure
std::map<uintptr_t, size_t> g_device_mem, g_host_mem;
std::set<ihipStream_t*> g_streams;
std::set<ihipEvent_t*> g_events;
long g_calls = 0, g_launches = 0;

hipError_t fail(hipError_t e) { g_last = e; return e; }

bool injected() {
    ++g_calls;
    if (g_fail_in > 0 && --g_fail_in == 0) { g_fail_in = -1; return true; }
    return false;
}

bool inside(const std::map<uintptr_t, size_t>& m, const void* p, size_t bytes = 1) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    auto it = m.upper_bound(a);
    if (it == m.begin()) return false;
    --it;
    return a >= it->first && a + bytes <= it->first + it->second;
}

void need_device(const void* p, size_t bytes, const char* what) {
    if (bytes && !inside(g_device_mem, p, bytes)) {
        std::fprintf(stderr, "hip_stub: %s touches %zu bytes at %p outside every device allocation\n", what, bytes, p);
        std::abort();
    }
}

}  // namespace

extern "C" {

// ---- test hooks
void stub_fail_in(long n) { g_fail_in = n; }
long stub_calls() { return g_calls; }
long stub_launches() { return g_launches; }
size_t stub_live_device_allocations() { return g_device_mem.size(); }
size_t stub_live_streams_and_events() { return g_streams.size() + g_events.size() + g_host_mem.size(); }

// ---- devices
hipError_t hipGetDeviceCount(int* n) { *n = g_devices; return g_devices > 0 ? hipSuccess : fail(hipErrorNoDevice); }
hipError_t hipGetDevice(int* d) { *d = g_current; return hipSuccess; }
hipError_t hipSetDevice(int d) {
    if (d < 0 || d >= g_devices) return fail(hipErrorInvalidDevice);
    g_current = d;
    return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipDeviceCanAccessPeer(int* can, int a, int b) { *can = a != b; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
hipError_t hipGetLastError(void) { hipError_t e = g_last; g_last = hipSuccess; return e; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "stub error"; }

// ---- memory
hipError_t hipMalloc(void** p, size_t bytes) {
    if (injected()) return fail(hipErrorOutOfMemory);
    *p = std::malloc(bytes ? bytes : 1);
    if (!*p) return fail(hipErrorOutOfMemory);
    std::memset(*p, 0xCD, bytes);  // fresh device memory holds garbage: nobody may rely on zeros
    g_device_mem[reinterpret_cast<uintptr_t>(*p)] = bytes ? bytes : 1;
    return hipSuccess;
}
hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    auto it = g_device_mem.find(reinterpret_cast<uintptr_t>(p));
    if (it == g_device_mem.end()) {
        std::fprintf(stderr, "hip_stub: hipFree(%p): not a live device allocation\n", p);
        std::abort();
    }
    g_device_mem.erase(it);
    std::free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) {
    if (injected()) return fail(hipErrorOutOfMemory);
    *p = std::malloc(bytes ? bytes : 1);
    g_host_mem[reinterpret_cast<uintptr_t>(*p)] = bytes ? bytes : 1;
    return hipSuccess;
}
hipError_t hipHostFree(void* p) {
    if (!p) return hipSuccess;
    if (!g_host_mem.erase(reinterpret_cast<uintptr_t>(p))) std::abort();
    std::free(p);
    return hipSuccess;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
    std::memset(a, 0, sizeof *a);
    if (inside(g_device_mem, p)) { a->type = hipMemoryTypeDevice; return hipSuccess; }
    if (inside(g_host_mem, p)) { a->type = hipMemoryTypeHost; return hipSuccess; }
    return fail(hipErrorInvalidValue);  // plain host memory: what the real runtime says too
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) {
    if (injected()) return fail(hipErrorLaunchFailure);
    return hipMemcpy(d, s, n, k);
}
hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t) {
    if (injected()) return fail(hipErrorLaunchFailure);
    need_device(d, n, "hipMemcpyPeerAsync(dst)");
    need_device(s, n, "hipMemcpyPeerAsync(src)");
    if (n) std::memmove(d, s, n);
    return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
    for (size_t y = 0; y < h; ++y) std::memmove(static_cast<char*>(d) + y * dp, static_cast<const char*>(s) + y * sp, w);
    return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) { need_device(d, n, "hipMemset"); std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }

// ---- streams and events (everything has already happened by the time anybody asks)
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
    if (injected()) return fail(hipErrorOutOfMemory);
    *s = new ihipStream_t{g_current};
    g_streams.insert(*s);
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
    if (!g_streams.erase(s)) std::abort();
    delete s;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) { if (s && !g_streams.count(s)) std::abort(); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    if ((s && !g_streams.count(s)) || !g_events.count(e)) std::abort();  // a destroyed stream or event
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) {
    if (injected()) return fail(hipErrorOutOfMemory);
    *e = new ihipEvent_t{g_current, false};
    g_events.insert(*e);
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e) {
    if (!g_events.erase(e)) std::abort();
    delete e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    if (injected()) return fail(hipErrorLaunchFailure);
    if (!g_events.count(e) || (s && !g_streams.count(s))) std::abort();
    e->recorded = true;
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) { if (!g_events.count(e)) std::abort(); return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t e) { if (!g_events.count(e)) std::abort(); return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    if (!g_events.count(a) || !g_events.count(b)) std::abort();
    if (!a->recorded || !b->recorded) return fail(hipErrorInvalidHandle);
    *ms = 0.125f;
    return hipSuccess;
}

}  // extern "C"

// ---- the launchers of kifs_internal.hpp, on the CPU ------------------------------------------------------------------
namespace kifs {

namespace {
constexpr int TW = 32, TH = 8;

uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// The stand-in for a pixel's colour: a function of the camera and the pixel's frame coordinates.  A quarter of the
// 32 x 8 tiles hold something, the rest is background (so the sparse gather has both kinds).
uint32_t stub_pixel(const BatchView& v, int x, int y, uint32_t background) {
    const uint32_t key = bits(v.origin.x) * 2654435761u ^ bits(v.origin.y) * 40503u ^ bits(v.origin.z) * 69069u ^ bits(v.m1.x);
    if ((((unsigned(x) >> 5) + 3u * (unsigned(y) >> 3) + (key >> 7)) & 3u) != 0u) return background;
    uint32_t c = (unsigned(x) * 73856093u ^ unsigned(y) * 19349663u ^ key) | 0xff000000u;
    return c == background ? c ^ 1u : c;
}
}  // namespace

hipError_t launch_render(const BatchParams& B, uint32_t, uint32_t, hipStream_t) {
    ++g_launches;
    if (injected()) return fail(hipErrorLaunchFailure);
    const FrameParams& P = B.frame;
    const int tiles_x = (P.width + TW - 1) / TW;
    // the tile table must name every tile of the launch exactly once
    std::vector<char> seen;
    for (int f = 0; f < B.count; ++f) {
        const BatchView& v = B.table ? B.table[f] : B.view[f];
        uint32_t* out = v.out;
        seen.assign(size_t(P.tile_count), 0);
        need_device(P.tile_order, size_t(P.tile_count) * 4, "render: tile order");
        for (uint32_t i = 0; i < P.tile_count; ++i) {
            const uint32_t tx = P.tile_order[i] & 0xffffu, tj = P.tile_order[i] >> 16;
            if (int(tx) >= tiles_x) { std::fprintf(stderr, "hip_stub: tile column %u outside the frame\n", tx); std::abort(); }
            const size_t flat = size_t(tj) * size_t(tiles_x) + tx;
            if (flat >= seen.size() || seen[flat]) { std::fprintf(stderr, "hip_stub: the tile table is not a permutation\n"); std::abort(); }
            seen[flat] = 1;
            int base;
            if (P.stripe_rows) {
                need_device(P.stripe_rows + tj, 4, "render: stripe table");
                base = int(P.stripe_rows[tj]);
            } else {
                base = P.y0 + TH * int(tj);
            }
            for (int r = 0; r < TH; ++r) {
                const int y = base + r;
                if (y >= P.y1) break;
                const size_t row = P.out_frame_rows ? size_t(y) : (P.stripe_rows ? size_t(TH) * tj + size_t(r) : size_t(y - P.y0));
                for (int x = int(tx) * TW; x < int(tx) * TW + TW && x < P.width; ++x) {
                    uint32_t* dst = out + row * P.pitch_words + x;
                    need_device(dst, 4, "render: a pixel");
                    *dst = stub_pixel(v, x, y, P.background_rgba);
                }
            }
        }
    }
    if (P.tile_cost) {  // the feedback's per-tile costs: something the sort can chew on
        need_device(P.tile_cost, size_t(P.tile_count) * 4, "render: tile costs");
        for (uint32_t i = 0; i < P.tile_count; ++i) P.tile_cost[i] = (i * 2654435761u) >> 20;
    }
    return hipSuccess;
}

hipError_t launch_tile_order(uint32_t* cost, uint32_t* order, uint32_t tile_count, uint32_t tiles_x, uint32_t shift, hipStream_t) {
    ++g_launches;
    need_device(cost, size_t(tile_count) * 4, "tile_order: costs");
    need_device(order, size_t(tile_count) * 4, "tile_order: order");
    std::vector<std::pair<uint32_t, uint32_t>> keyed(tile_count);
    for (uint32_t i = 0; i < tile_count; ++i) keyed[i] = {~(cost[i] >> shift), i};  // descending cost, stable
    std::sort(keyed.begin(), keyed.end());
    for (uint32_t i = 0; i < tile_count; ++i) {
        const uint32_t t = keyed[i].second;
        order[i] = (t % tiles_x) | ((t / tiles_x) << 16);
        cost[t] = 0;
    }
    return hipSuccess;
}

hipError_t launch_unpack_stripes(uint8_t* dst, size_t dst_pitch, size_t dst_frame_stride, const uint8_t* src, size_t src_pitch,
                                 size_t src_shard_stride, const uint32_t* stripe_rows, int n_stripes, int count, int width,
                                 int height, hipStream_t) {
    ++g_launches;
    if (injected()) return fail(hipErrorLaunchFailure);
    for (int f = 0; f < count; ++f)
        for (int s = 0; s < n_stripes; ++s)
            for (int r = 0; r < TH && int(stripe_rows[s]) + r < height; ++r) {
                uint8_t* d = dst + size_t(f) * dst_frame_stride + (size_t(stripe_rows[s]) + r) * dst_pitch;
                const uint8_t* q = src + size_t(f) * src_shard_stride + (size_t(TH) * s + r) * src_pitch;
                need_device(d, size_t(width) * 4, "unpack_stripes: frame row");
                need_device(q, size_t(width) * 4, "unpack_stripes: shard row");
                std::memcpy(d, q, size_t(width) * 4);
            }
    return hipSuccess;
}

hipError_t launch_pack_sparse(const uint8_t* src, size_t src_pitch, size_t src_shard_stride, const uint32_t* stripe_rows, int n_stripes,
                              int count, int width, int height, uint32_t background, uint32_t* records, uint32_t* n_records, hipStream_t) {
    ++g_launches;
    if (injected()) return fail(hipErrorLaunchFailure);
    const int tiles_x = (width + TW - 1) / TW;
    need_device(n_records, 4, "pack_sparse: count");
    if (*n_records != 0) { std::fprintf(stderr, "hip_stub: pack_sparse: the record count was not cleared\n"); std::abort(); }
    uint32_t n = 0;
    for (int f = 0; f < count; ++f)
        for (int s = 0; s < n_stripes; ++s)
            for (int c = 0; c < tiles_x; ++c) {
                uint32_t px[TH * TW];
                bool any = false;
                for (int r = 0; r < TH; ++r)
                    for (int x = 0; x < TW; ++x) {
                        uint32_t v = background;
                        if (int(stripe_rows[s]) + r < height && c * TW + x < width) {
                            const uint8_t* q = src + size_t(f) * src_shard_stride + (size_t(TH) * s + r) * src_pitch + size_t(c * TW + x) * 4;
                            need_device(q, 4, "pack_sparse: shard pixel");
                            std::memcpy(&v, q, 4);
                        }
                        px[r * TW + x] = v;
                        any = any || v != background;
                    }
                if (!any) continue;
                uint32_t* rec = records + size_t(n) * SPARSE_RECORD_WORDS_HOST;
                need_device(rec, size_t(SPARSE_RECORD_WORDS_HOST) * 4, "pack_sparse: record");
                rec[0] = uint32_t((f * n_stripes + s) * tiles_x + c);
                rec[1] = rec[2] = rec[3] = 0;
                std::memcpy(rec + 4, px, sizeof px);
                ++n;
            }
    *n_records = n;
    return hipSuccess;
}

hipError_t launch_unpack_sparse(uint8_t* dst, size_t dst_pitch, size_t dst_frame_stride, const uint32_t* records, uint32_t n_records,
                                const uint32_t* stripe_rows, int n_stripes, int count, int width, int height, int erase,
                                uint32_t background, hipStream_t) {
    ++g_launches;
    if (injected()) return fail(hipErrorLaunchFailure);
    const int tiles_x = (width + TW - 1) / TW;
    for (uint32_t i = 0; i < n_records; ++i) {
        const uint32_t* rec = records + size_t(i) * SPARSE_RECORD_WORDS_HOST;
        need_device(rec, size_t(SPARSE_RECORD_WORDS_HOST) * 4, "unpack_sparse: record");
        const uint32_t id = rec[0];
        if (id >= uint32_t(count) * uint32_t(n_stripes) * uint32_t(tiles_x)) continue;
        const int f = int(id / uint32_t(n_stripes * tiles_x)), rest = int(id % uint32_t(n_stripes * tiles_x));
        const int s = rest / tiles_x, c = rest % tiles_x;
        for (int r = 0; r < TH && int(stripe_rows[s]) + r < height; ++r)
            for (int x = 0; x < TW && c * TW + x < width; ++x) {
                uint8_t* d = dst + size_t(f) * dst_frame_stride + (size_t(stripe_rows[s]) + r) * dst_pitch + size_t(c * TW + x) * 4;
                need_device(d, 4, "unpack_sparse: frame pixel");
                const uint32_t v = erase ? background : rec[4 + r * TW + x];
                std::memcpy(d, &v, 4);
            }
    }
    return hipSuccess;
}

hipError_t launch_fill_stripes(uint8_t* dst, size_t dst_pitch, size_t dst_frame_stride, const uint32_t* stripe_rows, int n_stripes,
                               int count, int width, int height, uint32_t background, hipStream_t) {
    ++g_launches;
    if (injected()) return fail(hipErrorLaunchFailure);
    for (int f = 0; f < count; ++f)
        for (int s = 0; s < n_stripes; ++s)
            for (int r = 0; r < TH && int(stripe_rows[s]) + r < height; ++r) {
                uint8_t* d = dst + size_t(f) * dst_frame_stride + (size_t(stripe_rows[s]) + r) * dst_pitch;
                need_device(d, size_t(width) * 4, "fill_stripes: frame row");
                for (int x = 0; x < width; ++x) std::memcpy(d + size_t(x) * 4, &background, 4);
            }
    return hipSuccess;
}

hipError_t launch_eval_points(const FrameParams&, uint32_t, uint32_t, const float*, int n, float* sdf, float* nrm, hipStream_t) {
    ++g_launches;
    if (sdf) std::memset(sdf, 0, size_t(n) * 4);
    if (nrm) std::memset(nrm, 0, size_t(n) * 12);
    return hipSuccess;
}

hipError_t launch_eval_math(int, const float* in, float, const float*, float* out, int n, hipStream_t) {
    ++g_launches;
    std::memcpy(out, in, size_t(n) * 4);
    return hipSuccess;
}

}  // namespace kifs
