// host_driver.cpp -- TEST INFRASTRUCTURE: drives the C ABI of include/kifs_hip.h against tests/hip_stub/hip_stub.cpp
// under AddressSanitizer + UndefinedBehaviorSanitizer (make -C kifs_raymarching_amd/csrc asan; tests/
// test_host_sanitizers.py).  The scenarios are the ones tests/test_gpu_multi_batch.py, test_gpu_shards.py and
// test_gpu_api.py run on the GPU -- here what is checked is the HOST side: every frame gathered from four "devices"
// must equal the frame one context renders (the stub's pixels depend on camera and frame coordinates only), nothing may
// touch memory it does not own, every failure that is injected must leave the object usable, and after the last
// destroy the stub must hold no allocation, stream or event.  Prints "host_driver: N checks ok".
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/kifs_hip.h"

extern "C" {
void stub_fail_in(long n);
long stub_calls();
size_t stub_live_device_allocations();
size_t stub_live_streams_and_events();
}

static long g_checks = 0;
#define CHECK(cond)                                                                        \
    do {                                                                                   \
        ++g_checks;                                                                        \
        if (!(cond)) {                                                                     \
            std::fprintf(stderr, "host_driver: %s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #cond); \
            std::exit(1);                                                                  \
        }                                                                                  \
    } while (0)

namespace {

struct Scene {
    KifsScreenUniform screen;
    KifsOptionsUniform options;
    int w, h;
};

Scene scene(int w, int h, uint8_t bg = 0) {
    Scene s{};
    s.w = w;
    s.h = h;
    CHECK(kifs_host_screen(uint32_t(w), uint32_t(h), &s.screen) == KIFS_OK);
    KifsGuiData gui;
    kifs_host_gui_default(&gui);
    gui.fractal_group = 1;  // Julia
    gui.background_color[0] = bg;
    gui.background_color[1] = uint8_t(bg / 2);
    CHECK(kifs_host_options(&gui, &s.options) == KIFS_OK);
    return s;
}

std::vector<KifsCameraUniform> cameras(int n, int first) {
    std::vector<KifsCameraUniform> out(static_cast<size_t>(n), KifsCameraUniform{});
    for (int i = 0; i < n; ++i) {
        KifsCameraData c;
        kifs_host_camera_default(&c);
        c.origin_distance = 3.0f + 0.25f * float((first + i) % 7);
        c.phi = 0.37f * float(first + i);
        c.theta = 0.2f * float((first + i) % 5) - 0.4f;
        CHECK(kifs_host_camera(&c, &out[size_t(i)]) == KIFS_OK);
    }
    return out;
}

uint8_t* dev_alloc(size_t bytes) {
    void* p = nullptr;
    CHECK(hipMalloc(&p, bytes) == hipSuccess);
    return static_cast<uint8_t*>(p);
}

// The frames one context renders, one kifs_render each, into host memory.
std::vector<uint8_t> single_frames(const Scene& s, const std::vector<KifsCameraUniform>& cams, int encode = 1) {
    int st = 0;
    kifs_ctx* c = kifs_create(0, &st);
    CHECK(c && st == KIFS_OK);
    CHECK(kifs_set_screen(c, &s.screen) == KIFS_OK && kifs_set_options(c, &s.options) == KIFS_OK);
    const size_t fb = size_t(s.w) * s.h * 4;
    std::vector<uint8_t> out(fb * cams.size());
    for (size_t i = 0; i < cams.size(); ++i) {
        CHECK(kifs_set_camera(c, &cams[i]) == KIFS_OK);
        CHECK(kifs_render(c, out.data() + i * fb, size_t(s.w) * 4, 0, s.h, encode) == KIFS_OK);
    }
    kifs_destroy(c);
    return out;
}

bool equal_dev(const uint8_t* dev, const std::vector<uint8_t>& want) { return std::memcmp(dev, want.data(), want.size()) == 0; }

kifs_multi* multi(const std::vector<int>& devs, const Scene& s, int gather, int transport = KIFS_TRANSPORT_COPY) {
    int st = 0;
    kifs_multi* m = kifs_multi_create(devs.data(), int(devs.size()), &st);
    CHECK(m && st == KIFS_OK);
    CHECK(kifs_multi_set_screen(m, &s.screen) == KIFS_OK && kifs_multi_set_options(m, &s.options) == KIFS_OK);
    auto cam = cameras(1, 0);
    CHECK(kifs_multi_set_camera(m, &cam[0]) == KIFS_OK);
    CHECK(kifs_multi_set_gather(m, gather, transport) == KIFS_OK);
    return m;
}

// ---- scenarios -----------------------------------------------------------------------------------------------

void null_and_argument_checks() {
    CHECK(kifs_abi_version() == KIFS_ABI_VERSION);
    CHECK(kifs_set_screen(nullptr, nullptr) == KIFS_ERR_BAD_ARG && kifs_render(nullptr, nullptr, 0, 0, 0, 0) == KIFS_ERR_BAD_ARG);
    CHECK(kifs_order_after(nullptr, nullptr, nullptr) == KIFS_ERR_BAD_ARG && kifs_multi_order_after(nullptr, nullptr) == KIFS_ERR_BAD_ARG);
    CHECK(kifs_multi_wait_all(nullptr) == KIFS_ERR_BAD_ARG && kifs_multi_stats(nullptr, nullptr, 0) == KIFS_ERR_BAD_ARG);
    int st = 0;
    CHECK(kifs_create(99, &st) == nullptr && st == KIFS_ERR_NO_DEVICE);
    CHECK(kifs_multi_create(nullptr, 2, &st) == nullptr && st == KIFS_ERR_BAD_ARG);
    int bad[2] = {0, 77};
    CHECK(kifs_multi_create(bad, 2, &st) == nullptr && st == KIFS_ERR_NO_DEVICE);
    kifs_destroy(nullptr);
    kifs_multi_destroy(nullptr);
    kifs_ctx* c = kifs_create(1, &st);
    CHECK(c && st == KIFS_OK);
    uint8_t px[64];
    CHECK(kifs_render(c, px, 16, 0, 1, 1) == KIFS_ERR_UNCONFIGURED);
    Scene s = scene(4, 4);
    KifsScreenUniform zero = s.screen;
    zero.width = 0.0f;
    CHECK(kifs_set_screen(c, &zero) == KIFS_ERR_BAD_SIZE);
    CHECK(kifs_set_screen(c, &s.screen) == KIFS_OK && kifs_set_options(c, &s.options) == KIFS_OK);
    KifsOptionsUniform o = s.options;
    o.fractal_group_id = 9;
    CHECK(kifs_set_options(c, &o) == KIFS_ERR_BAD_ARG && kifs_set_iters(c, -1, 0, 0) == KIFS_ERR_BAD_ARG);
    auto cam = cameras(1, 0);
    CHECK(kifs_set_camera(c, &cam[0]) == KIFS_OK);
    CHECK(kifs_render(c, px, 16, 0, 4, 1) == KIFS_OK && kifs_render(c, px, 16, 0, 5, 1) == KIFS_ERR_BAD_ARG);
    CHECK(kifs_render(c, px, 12, 0, 4, 1) == KIFS_ERR_BAD_SIZE && kifs_render(c, px, 16, 0, 4, 2) == KIFS_ERR_BAD_ARG);
    CHECK(kifs_render(c, px, 16, 2, 2, 1) == KIFS_OK);
    CHECK(kifs_order_after(c, nullptr, nullptr) == KIFS_OK);
    int y0, y1, n, rows;
    CHECK(kifs_band_range(1080, 3, 8, &y0, &y1) == KIFS_OK && y0 == 405 && y1 == 540);
    CHECK(kifs_band_range(10, 8, 8, &y0, &y1) == KIFS_ERR_BAD_ARG);
    int stripes[200];
    int wts[3] = {3, 1, 1};
    CHECK(kifs_shard_stripes(1080, 3, wts, 0, stripes, 200, &n, &rows) == KIFS_OK && n == 81 && rows == 648);
    CHECK(kifs_shard_stripes(1080, 3, wts, 0, stripes, 10, &n, &rows) == KIFS_ERR_BAD_ARG);
    kifs_destroy(c);
}

// One context: bands, batches (inline and through the view table), shards packed and in place, the sparse round trip.
void single_context(int w, int h) {
    Scene s = scene(w, h, 40);
    const size_t fb = size_t(w) * h * 4;
    auto cams = cameras(70, 3);
    auto want = single_frames(s, cams);
    int st = 0;
    kifs_ctx* c = kifs_create(2, &st);
    CHECK(c != nullptr);
    CHECK(kifs_set_screen(c, &s.screen) == KIFS_OK && kifs_set_options(c, &s.options) == KIFS_OK && kifs_set_camera(c, &cams[0]) == KIFS_OK);
    CHECK(kifs_set_profiling(c, 1) == KIFS_OK);
    // a frame in three bands with a padded pitch, into device memory
    const size_t pitch = size_t(w) * 4 + 32;
    uint8_t* d = dev_alloc(pitch * h);
    const int cuts[4] = {0, h / 3, h / 3 + 5 < h ? h / 3 + 5 : h, h};
    for (int b = 0; b < 3; ++b)
        CHECK(kifs_render_async(c, nullptr, d + size_t(cuts[b]) * pitch, pitch, cuts[b], cuts[b + 1], 1) == KIFS_OK);
    CHECK(kifs_synchronize(c) == KIFS_OK);
    for (int y = 0; y < h; ++y) CHECK(std::memcmp(d + y * pitch, want.data() + size_t(y) * w * 4, size_t(w) * 4) == 0);
    CHECK(hipFree(d) == hipSuccess);
    // batches of 5 and of 70 (beyond the kernel argument: the device view table, four times round its ring)
    uint8_t* frames = dev_alloc(fb * 70);
    std::vector<uint8_t*> outs(70);
    for (int i = 0; i < 70; ++i) outs[size_t(i)] = frames + fb * i;
    CHECK(kifs_render_batch_async(c, nullptr, 5, cams.data(), outs.data(), size_t(w) * 4, 0, h, 1) == KIFS_OK);
    CHECK(std::memcmp(frames, want.data(), fb * 5) == 0);
    for (int round = 0; round < 6; ++round) {
        std::memset(frames, 0xEE, fb * 70);
        CHECK(kifs_render_batch_async(c, nullptr, 70, cams.data(), outs.data(), size_t(w) * 4, 0, h, 1) == KIFS_OK);
        CHECK(equal_dev(frames, want));
    }
    CHECK(kifs_render_batch_async(c, nullptr, 0, cams.data(), outs.data(), size_t(w) * 4, 0, h, 1) == KIFS_ERR_BAD_ARG);
    CHECK(kifs_render_batch_async(c, nullptr, 513, cams.data(), outs.data(), size_t(w) * 4, 0, h, 1) == KIFS_ERR_BAD_ARG);
    int launches = 0;
    double mean = 0, lo = 0, hi = 0;
    CHECK(kifs_profile_read(c, &launches, &mean, &lo, &hi) == KIFS_OK && launches >= 7);
    // three ranks' shards of 6 frames: packed + unpack, in place, and the sparse form; every partition's union = the frames
    const int B = 6, world = 3;
    const int all = (h + 7) / 8;
    uint8_t* gathered = dev_alloc(fb * B);
    for (int mode = 0; mode < 3; ++mode) {  // 0 dense packed, 1 in place, 2 sparse
        std::memset(gathered, 0x11, fb * B);
        std::vector<int> peer_stripes;
        for (int r = 0; r < world; ++r) {
            std::vector<int> st_(static_cast<size_t>(all), 0);
            int n = 0, rows = 0;
            CHECK(kifs_shard_stripes(h, world, nullptr, r, st_.data(), all, &n, &rows) == KIFS_OK);
            st_.resize(size_t(n));
            if (n == 0) continue;
            if (mode == 1) {
                std::vector<uint8_t*> fo(B);
                for (int i = 0; i < B; ++i) fo[size_t(i)] = gathered + fb * i;
                CHECK(kifs_render_shard_async(c, nullptr, B, cams.data(), fo.data(), size_t(w) * 4, st_.data(), n, 1, 1) == KIFS_OK);
                continue;
            }
            const size_t sb = size_t(rows) * w * 4;
            uint8_t* shard = dev_alloc(sb * B);
            std::vector<uint8_t*> so(B);
            for (int i = 0; i < B; ++i) so[size_t(i)] = shard + sb * i;
            CHECK(kifs_render_shard_async(c, nullptr, B, cams.data(), so.data(), size_t(w) * 4, st_.data(), n, 0, 1) == KIFS_OK);
            if (mode == 0) {
                CHECK(kifs_unpack_shard_async(c, nullptr, B, gathered, size_t(w) * 4, fb, shard, size_t(w) * 4, sb, st_.data(), n) == KIFS_OK);
            } else {
                const size_t cap = size_t(B) * n * ((w + 31) / 32);
                uint8_t* rec = dev_alloc(cap * KIFS_SPARSE_RECORD_BYTES);
                uint32_t* cnt = reinterpret_cast<uint32_t*>(dev_alloc(4));
                void* pinned = nullptr;
                CHECK(hipHostMalloc(&pinned, 4, 0) == hipSuccess);
                CHECK(kifs_pack_sparse_async(c, nullptr, B, shard, size_t(w) * 4, sb, st_.data(), n, 1, rec, cap, cnt,
                                             static_cast<uint32_t*>(pinned)) == KIFS_OK);
                const uint32_t nrec = *static_cast<uint32_t*>(pinned);
                CHECK(nrec == *cnt && nrec <= cap && nrec > 0 && nrec < cap);  // (some tiles hold something, some do not)
                CHECK(kifs_pack_sparse_async(c, nullptr, B, shard, size_t(w) * 4, sb, st_.data(), n, 1, rec, cap - 1, cnt, nullptr) == KIFS_ERR_BAD_SIZE);
                CHECK(kifs_fill_shard_async(c, nullptr, B, gathered, size_t(w) * 4, fb, st_.data(), n, 1) == KIFS_OK);
                CHECK(kifs_unpack_sparse_async(c, nullptr, B, gathered, size_t(w) * 4, fb, rec, nrec, st_.data(), n) == KIFS_OK);
                // erase + scatter again: the same frames
                CHECK(kifs_erase_sparse_async(c, nullptr, B, gathered, size_t(w) * 4, fb, rec, nrec, st_.data(), n, 1) == KIFS_OK);
                CHECK(kifs_unpack_sparse_async(c, nullptr, B, gathered, size_t(w) * 4, fb, rec, nrec, st_.data(), n) == KIFS_OK);
                CHECK(hipFree(rec) == hipSuccess && hipFree(cnt) == hipSuccess && hipHostFree(pinned) == hipSuccess);
            }
            CHECK(hipFree(shard) == hipSuccess);
        }
        CHECK(std::memcmp(gathered, want.data(), fb * B) == 0);
    }
    // stripe lists the library must refuse
    {
        int desc[2] = {1, 0}, beyond[1] = {all}, neg[1] = {-1};
        uint8_t* o1[1] = {gathered};
        CHECK(kifs_render_shard_async(c, nullptr, 1, cams.data(), o1, size_t(w) * 4, desc, 2, 1, 1) == KIFS_ERR_BAD_ARG);
        CHECK(kifs_render_shard_async(c, nullptr, 1, cams.data(), o1, size_t(w) * 4, beyond, 1, 1, 1) == KIFS_ERR_BAD_ARG);
        CHECK(kifs_render_shard_async(c, nullptr, 1, cams.data(), o1, size_t(w) * 4, neg, 1, 1, 1) == KIFS_ERR_BAD_ARG);
    }
    CHECK(hipFree(gathered) == hipSuccess && hipFree(frames) == hipSuccess);
    kifs_destroy(c);
}

// kifs_multi: batches on four devices, both gathers, against single frames; the lone-frame form; weights.
void multi_batches(int w, int h, int gather) {
    Scene s = scene(w, h, 90);
    const size_t fb = size_t(w) * h * 4;
    const int B = 9;
    auto cams = cameras(B, 11);
    auto want = single_frames(s, cams);
    kifs_multi* m = multi({0, 1, 2, 3}, s, gather);
    uint8_t* frames = dev_alloc(fb * B);
    std::memset(frames, 0x5A, fb * B);
    CHECK(kifs_multi_render_batch(m, B, cams.data(), frames, size_t(w) * 4, fb, 1) == KIFS_OK);
    CHECK(equal_dev(frames, want));
    KifsMultiStats st{};
    CHECK(kifs_multi_stats(m, &st, 0) == KIFS_OK && st.steps == 1 && st.transport == KIFS_TRANSPORT_COPY && st.gather == gather);
    CHECK(st.bytes_received > 0);
    if (gather == KIFS_GATHER_SPARSE) CHECK(st.records_received > 0 && st.records_received < st.tiles_covered);
    // the lone-frame form: device destination, host destination
    CHECK(kifs_multi_set_camera(m, &cams[4]) == KIFS_OK);
    std::memset(frames, 0, fb);
    CHECK(kifs_multi_render(m, frames, size_t(w) * 4, 1) == KIFS_OK);
    CHECK(std::memcmp(frames, want.data() + 4 * fb, fb) == 0);
    std::vector<uint8_t> host(fb);
    CHECK(kifs_multi_render(m, host.data(), size_t(w) * 4, 1) == KIFS_OK);
    CHECK(std::memcmp(host.data(), want.data() + 4 * fb, fb) == 0);
    for (int i = 0; i < 4; ++i) CHECK(kifs_multi_shard_ms(m, i) >= 0.0);
    // other shares, a device with no share at all
    int wts[4] = {2, 0, 5, 1};
    CHECK(kifs_multi_set_weights(m, wts) == KIFS_OK);
    int dev = 0, n = 0, rows = 0, total = 0;
    for (int i = 0; i < 4; ++i) {
        CHECK(kifs_multi_shard(m, i, &dev, &n, &rows) == KIFS_OK && dev == i);
        total += rows;
        if (i == 1) CHECK(n == 0 && rows == 0);
    }
    CHECK(total == h);
    std::memset(frames, 0xA5, fb * B);
    uint64_t step = 0;
    CHECK(kifs_multi_render_batch_async(m, B, cams.data(), frames, size_t(w) * 4, fb, 1, 0, &step) == KIFS_OK);
    CHECK(kifs_multi_order_after(m, nullptr) == KIFS_OK);
    CHECK(kifs_multi_wait(m, step) == KIFS_OK && equal_dev(frames, want));
    int zero[4] = {0, 0, 0, 0}, neg[4] = {1, -1, 1, 1};
    CHECK(kifs_multi_set_weights(m, zero) == KIFS_ERR_BAD_ARG && kifs_multi_set_weights(m, neg) == KIFS_ERR_BAD_ARG);
    CHECK(kifs_multi_set_weights(m, nullptr) == KIFS_OK);
    // the root alone, and a root with no share
    int root_only[4] = {1, 0, 0, 0}, no_root[4] = {0, 1, 1, 1};
    for (int* wt : {root_only, no_root}) {
        CHECK(kifs_multi_set_weights(m, wt) == KIFS_OK);
        std::memset(frames, 0x3C, fb * B);
        CHECK(kifs_multi_render_batch(m, B, cams.data(), frames, size_t(w) * 4, fb, 1) == KIFS_OK && equal_dev(frames, want));
    }
    // argument checks of the batched form
    CHECK(kifs_multi_render_batch_async(m, 0, cams.data(), frames, size_t(w) * 4, fb, 1, 0, &step) == KIFS_ERR_BAD_ARG);
    CHECK(kifs_multi_render_batch_async(m, B, cams.data(), frames, size_t(w) * 4 - 4, fb, 1, 0, &step) == KIFS_ERR_BAD_SIZE);
    CHECK(kifs_multi_render_batch_async(m, B, cams.data(), frames, size_t(w) * 4, fb - 4, 1, 0, &step) == KIFS_ERR_BAD_SIZE);
    CHECK(kifs_multi_render_batch_async(m, B, cams.data(), host.data(), size_t(w) * 4, fb, 1, 0, &step) == KIFS_ERR_BAD_ARG);
    CHECK(kifs_multi_wait(m, 1000) == KIFS_ERR_BAD_ARG && kifs_multi_stream_wait(m, 0, nullptr) == KIFS_ERR_BAD_ARG);
    CHECK(hipFree(frames) == hipSuccess);
    kifs_multi_destroy(m);
}

// Pipelined steps: two buffers handed back untouched; ONE buffer for every step; a lone render in between; overlapping
// buffers; a change of options / partition / gather under the untouched flag; destroy with steps in flight.
void multi_pipeline(int w, int h) {
    Scene s = scene(w, h, 0);
    const size_t fb = size_t(w) * h * 4;
    const int B = 5;
    kifs_multi* m = multi({0, 1, 2}, s, KIFS_GATHER_SPARSE);
    uint8_t* buf[2] = {dev_alloc(fb * B), dev_alloc(fb * B)};
    uint64_t steps[9];
    std::vector<std::vector<uint8_t>> wants;
    for (int k = 0; k < 9; ++k) wants.push_back(single_frames(s, cameras(B, 10 * k)));
    for (int k = 0; k < 9; ++k) {
        if (k >= 2) {
            CHECK(kifs_multi_wait(m, steps[k - 2]) == KIFS_OK);
            CHECK(equal_dev(buf[k % 2], wants[size_t(k - 2)]));
        }
        auto cams = cameras(B, 10 * k);
        CHECK(kifs_multi_render_batch_async(m, B, cams.data(), buf[k % 2], size_t(w) * 4, fb, 1, k >= 2 ? KIFS_MULTI_FRAMES_UNTOUCHED : 0,
                                            &steps[k]) == KIFS_OK && steps[k] == uint64_t(k));
    }
    CHECK(kifs_multi_wait_all(m) == KIFS_OK && equal_dev(buf[1], wants[7]) && equal_dev(buf[0], wants[8]));
    KifsMultiStats st{};
    CHECK(kifs_multi_stats(m, &st, 1) == KIFS_OK && st.steps == 9);
    // ONE buffer for every step, the flag always set: step k - 1's tiles must not survive under step k's background
    for (int k = 0; k < 6; ++k) {
        auto cams = cameras(B, 100 + 13 * k);
        uint64_t sn = 0;
        CHECK(kifs_multi_render_batch_async(m, B, cams.data(), buf[0], size_t(w) * 4, fb, 1, KIFS_MULTI_FRAMES_UNTOUCHED, &sn) == KIFS_OK);
        CHECK(kifs_multi_wait(m, sn) == KIFS_OK && equal_dev(buf[0], single_frames(s, cams)));
        if (k == 3) {  // a lone frame into the middle of the buffer between two steps
            auto lone = cameras(1, 999);
            CHECK(kifs_multi_set_camera(m, &lone[0]) == KIFS_OK && kifs_multi_render(m, buf[0] + 2 * fb, size_t(w) * 4, 1) == KIFS_OK);
        }
    }
    // overlapping buffers: frames 0..3 and 1..4 of one allocation, alternating
    for (int k = 0; k < 4; ++k) {
        auto cams = cameras(B - 1, 300 + 7 * k);
        uint8_t* view = buf[1] + (k % 2) * fb;
        uint64_t sn = 0;
        CHECK(kifs_multi_render_batch_async(m, B - 1, cams.data(), view, size_t(w) * 4, fb, 1, KIFS_MULTI_FRAMES_UNTOUCHED, &sn) == KIFS_OK);
        CHECK(kifs_multi_wait(m, sn) == KIFS_OK && equal_dev(view, single_frames(s, cams)));
    }
    // another background colour, another partition, the dense gather and back -- the flag stays set throughout
    Scene red = scene(w, h, 200);
    auto cams = cameras(B, 5);
    CHECK(kifs_multi_set_options(m, &red.options) == KIFS_OK);
    for (int k = 0; k < 3; ++k) {
        uint64_t sn = 0;
        CHECK(kifs_multi_render_batch_async(m, B, cams.data(), buf[k % 2], size_t(w) * 4, fb, 1, KIFS_MULTI_FRAMES_UNTOUCHED, &sn) == KIFS_OK);
    }
    CHECK(kifs_multi_wait_all(m) == KIFS_OK);
    auto want_red = single_frames(red, cams);
    CHECK(equal_dev(buf[0], want_red) && equal_dev(buf[1], want_red));
    int wts[3] = {1, 3, 2};
    CHECK(kifs_multi_set_weights(m, wts) == KIFS_OK);
    for (int k = 0; k < 2; ++k) {
        uint64_t sn = 0;
        CHECK(kifs_multi_render_batch_async(m, B, cams.data(), buf[k], size_t(w) * 4, fb, 1, KIFS_MULTI_FRAMES_UNTOUCHED, &sn) == KIFS_OK);
    }
    CHECK(kifs_multi_set_gather(m, KIFS_GATHER_DENSE, KIFS_TRANSPORT_COPY) == KIFS_OK);  // (drains the two steps)
    CHECK(equal_dev(buf[0], want_red) && equal_dev(buf[1], want_red));
    for (int k = 0; k < 3; ++k) {
        uint64_t sn = 0;
        std::memset(buf[k % 2], 0x77, fb * B);
        CHECK(kifs_multi_render_batch_async(m, B, cams.data(), buf[k % 2], size_t(w) * 4, fb, 1, KIFS_MULTI_FRAMES_UNTOUCHED, &sn) == KIFS_OK);
        CHECK(kifs_multi_wait(m, sn) == KIFS_OK && equal_dev(buf[k % 2], want_red));
    }
    CHECK(kifs_multi_set_gather(m, KIFS_GATHER_SPARSE, KIFS_TRANSPORT_AUTO) == KIFS_OK);
    CHECK(kifs_multi_set_gather(m, 7, 0) == KIFS_ERR_BAD_ARG && kifs_multi_set_gather(m, 0, 7) == KIFS_ERR_BAD_ARG);
    // a device listed twice cannot have RCCL: explicit RCCL is an error there, AUTO means peer copies
    {
        int st_ = 0;
        int twice[3] = {0, 1, 1};
        kifs_multi* m2 = kifs_multi_create(twice, 3, &st_);
        CHECK(m2 != nullptr);
        CHECK(kifs_multi_set_screen(m2, &s.screen) == KIFS_OK && kifs_multi_set_options(m2, &s.options) == KIFS_OK);
        CHECK(kifs_multi_set_gather(m2, KIFS_GATHER_SPARSE, KIFS_TRANSPORT_RCCL) == KIFS_ERR_COMM);
        CHECK(kifs_multi_set_gather(m2, KIFS_GATHER_SPARSE, KIFS_TRANSPORT_AUTO) == KIFS_OK);
        auto c2 = cameras(B, 40);
        CHECK(kifs_multi_render_batch(m2, B, c2.data(), buf[0], size_t(w) * 4, fb, 1) == KIFS_OK && equal_dev(buf[0], single_frames(s, c2)));
        CHECK(kifs_multi_stats(m2, &st, 0) == KIFS_OK && st.transport == KIFS_TRANSPORT_COPY);
        kifs_multi_destroy(m2);
    }
    // destroy with two steps in flight
    CHECK(kifs_multi_set_gather(m, KIFS_GATHER_SPARSE, KIFS_TRANSPORT_COPY) == KIFS_OK);
    for (int k = 0; k < 2; ++k) {
        uint64_t sn = 0;
        CHECK(kifs_multi_render_batch_async(m, B, cams.data(), buf[k], size_t(w) * 4, fb, 1, 0, &sn) == KIFS_OK);
    }
    kifs_multi_destroy(m);
    CHECK(equal_dev(buf[0], want_red) && equal_dev(buf[1], want_red));  // (destroy completes what is in flight)
    CHECK(hipFree(buf[0]) == hipSuccess && hipFree(buf[1]) == hipSuccess);
}

// Every HIP call and launch of a submit / wait cycle fails once, in turn: the call reports an error (or absorbs it), the
// object stays usable, the next steps are exact, and nothing leaks.
void injected_failures(int w, int h) {
    Scene s = scene(w, h, 10);
    const size_t fb = size_t(w) * h * 4;
    const int B = 4;
    auto cams = cameras(B, 21);
    auto want = single_frames(s, cams);
    uint8_t* frames = dev_alloc(fb * B);
    int failed_calls = 0;
    for (int gather : {KIFS_GATHER_SPARSE, KIFS_GATHER_DENSE}) {
        for (long n = 1; n < 400; ++n) {
            kifs_multi* m = multi({0, 1, 2}, s, gather);
            uint64_t s0 = 0, s1 = 0;
            // a clean first step, so that the failure lands in a steady-state submit (buffers exist) for small n as well
            CHECK(kifs_multi_render_batch_async(m, B, cams.data(), frames, size_t(w) * 4, fb, 1, 0, &s0) == KIFS_OK);
            const long before = stub_calls();
            stub_fail_in(n);
            int st = kifs_multi_render_batch_async(m, B, cams.data(), frames, size_t(w) * 4, fb, 1, 0, &s1);
            int sw = st == KIFS_OK ? kifs_multi_wait(m, s1) : KIFS_OK;
            const bool reached = stub_calls() - before >= n;
            stub_fail_in(-1);
            if (st != KIFS_OK || sw != KIFS_OK) ++failed_calls;
            CHECK(st == KIFS_OK || st == KIFS_ERR_RUNTIME || st == KIFS_ERR_COMM);
            CHECK(sw == KIFS_OK || sw == KIFS_ERR_RUNTIME || sw == KIFS_ERR_COMM);
            // whatever happened: the object renders exact frames afterwards (a failed wait may need a second one to drain)
            (void)kifs_multi_wait_all(m);
            std::memset(frames, 0x42, fb * B);
            uint64_t s2 = 0;
            CHECK(kifs_multi_render_batch_async(m, B, cams.data(), frames, size_t(w) * 4, fb, 1, 0, &s2) == KIFS_OK);
            CHECK(kifs_multi_wait(m, s2) == KIFS_OK && equal_dev(frames, want));
            kifs_multi_destroy(m);
            if (!reached) break;  // the cycle has fewer than n calls: every one has had its turn
        }
    }
    CHECK(failed_calls > 10);
    // a context whose creation fails half way
    for (long n = 1; n < 12; ++n) {
        stub_fail_in(n);
        int st = 0;
        kifs_ctx* c = kifs_create(0, &st);
        stub_fail_in(-1);
        if (c) kifs_destroy(c);
        else CHECK(st == KIFS_ERR_DEVICE_INIT);
    }
    CHECK(hipFree(frames) == hipSuccess);
}

}  // namespace

int main() {
    null_and_argument_checks();
    single_context(333, 211);  // neither dimension a multiple of the 32 x 8 tile
    single_context(64, 8);
    for (int gather : {KIFS_GATHER_SPARSE, KIFS_GATHER_DENSE}) {
        multi_batches(333, 211, gather);
        multi_batches(96, 20, gather);
    }
    multi_pipeline(320, 200);
    multi_pipeline(70, 37);
    injected_failures(100, 50);
    CHECK(stub_live_device_allocations() == 0);
    CHECK(stub_live_streams_and_events() == 0);
    std::printf("host_driver: %ld checks ok\n", g_checks);
    return 0;
}
