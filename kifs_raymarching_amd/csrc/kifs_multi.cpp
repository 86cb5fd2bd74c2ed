// kifs_multi.cpp -- one process driving several devices (the reference's host is one process and one
// thread, application.rs:37-48): a kifs_multi owns one context per listed device.
//
//   kifs_multi_render              one frame: stripe shards, dense, synchronous (the latency form)
//   kifs_multi_render_batch_async  a step of up to 512 frames: ONE launch per device, the other devices' rows
//                                  packed into sparse records (or sent whole), gathered on the root by RCCL
//                                  grouped send/recv (or peer copies), two steps in flight
//
// How a step flows (S = 2 slots; device 0 is the root; every device has its context's render stream and a
// comm stream, the root also a gather stream):
//
//   submit(k)   host: complete step k - 2 (the slot's previous step: its payload buffers are free again)
//               root gather stream : background under the other devices' rows of the step's frames (fill, or
//                                    erase under the slot's previous records)
//               every render stream: render shard  [pack + record count to pinned memory] -> event packed[i]
//                                    (the root: in place into the frames -> event rendered)
//               then flush(k - 1)
//   flush(j)    host: wait packed[i] of step j, read the record counts (pinned memory)       -- by now every device
//               RCCL: GroupStart; device i comm stream: Send(records, root) ...; root gather   is rendering step j + 1
//                     stream: Recv(i) ...; GroupEnd          (COPY: hipMemcpyPeerAsync on the gather stream)
//               root gather stream : scatter the records (or unpack the stripes) -> event gathered
//   wait(j)     flush(j) if still pending; host waits for rendered and gathered of step j
//
// Nothing on a GPU waits for the host, and the host blocks only in flush (on work enqueued a step earlier).
// Host code only; kernels live in kifs_kernels.hip / kifs_support_kernels.hip.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <vector>

#include <dlfcn.h>

#include "kifs_comm.hpp"
#include "kifs_context.hpp"

using namespace kifs::host;

// ---- RCCL, opened on first use -------------------------------------------------------------------------
namespace kifs {
namespace host {

const RcclApi* rccl() {
    static const RcclApi* api = []() -> const RcclApi* {
        const bool verbose = std::getenv("KIFS_DEBUG") != nullptr;
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) {
            if (verbose) std::fprintf(stderr, "kifs: dlopen(librccl.so.1) failed: %s\n", dlerror());
            return nullptr;
        }
        static RcclApi a;
        bool ok = true;
        auto sym = [&](auto& fn, const char* name) {
            fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(h, name));
            if (!fn) {
                ok = false;
                if (verbose) std::fprintf(stderr, "kifs: librccl lacks %s\n", name);
            }
        };
        sym(a.GetVersion, "ncclGetVersion");
        sym(a.CommInitAll, "ncclCommInitAll");
        sym(a.CommDestroy, "ncclCommDestroy");
        sym(a.GroupStart, "ncclGroupStart");
        sym(a.GroupEnd, "ncclGroupEnd");
        sym(a.Send, "ncclSend");
        sym(a.Recv, "ncclRecv");
        sym(a.GetErrorString, "ncclGetErrorString");
        if (!ok) return nullptr;
        int v = 0;
        if (a.GetVersion(&v) == ncclSuccess) a.version = v;
        return &a;
    }();
    return api;
}

bool nccl_ok(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return true;
    static const bool verbose = std::getenv("KIFS_DEBUG") != nullptr;
    if (verbose) {
        const RcclApi* a = rccl();
        std::fprintf(stderr, "kifs: %s failed: %s\n", what, a ? a->GetErrorString(r) : "?");
    }
    return false;
}

}  // namespace host
}  // namespace kifs

// ---- the object ------------------------------------------------------------------------------------------
namespace {

constexpr int SLOTS = 2;  // steps in flight

// What one device contributes to one step (device 0, the root, uses only `stripes`, `rows`, ev0 / ev1).
struct PeerPart {
    std::vector<int> stripes;      // the device's shard when the step was submitted
    int rows = 0;
    uint8_t* shard = nullptr;      // on the device: `count` packed shards, rows x W x 4 each
    size_t shard_bytes = 0;
    uint8_t* records = nullptr;    // on the device: the sparse payload
    size_t records_bytes = 0;
    uint32_t* d_count = nullptr;   // on the device: number of records
    uint32_t* h_count = nullptr;   // pinned host copy
    uint8_t* recv = nullptr;       // on the root: the payload as received
    size_t recv_bytes = 0;
    hipEvent_t packed = nullptr;   // device's render stream: payload (and count) ready
    uint32_t n_records = 0;        // as flushed (sparse)
    size_t payload_bytes = 0;      // as flushed
};

struct StepSlot {
    bool used = false;             // holds a step that has not been completed by a wait
    bool flushed = false;
    uint64_t step = 0;
    int count = 0, encode = 0, gather = 0, width = 0, height = 0;
    uint8_t* frames = nullptr;
    size_t pitch = 0, stride = 0;
    uint32_t background = 0;
    bool background_known = false; // `frames` hold the background outside the records of `part` (sparse, flushed)
    bool overwritten = false;      // something else wrote into `frames` since this step was submitted
    std::vector<PeerPart> part;
    std::vector<int> peer_stripes; // every stripe that is not the root's, ascending
    hipEvent_t rendered = nullptr; // root render stream: the root's own rows are in the frames
    hipEvent_t gathered = nullptr; // root gather stream: everybody else's are
};

}  // namespace

struct kifs_multi {
    std::vector<kifs_ctx*> ctx;
    std::vector<int> dev;
    std::vector<int> weight;               // share of each device (kifs_shard_stripes weights)
    std::vector<std::vector<int>> stripes; // the shard of each device for the current frame height
    std::vector<int> rows;
    int stripes_height = -1;
    std::vector<uint8_t*> shard;           // kifs_multi_render: per-device packed shard buffer (non-root)
    std::vector<size_t> shard_bytes;
    std::vector<uint8_t*> recv;            // kifs_multi_render: the same shards after the peer copy (on the root)
    std::vector<size_t> recv_bytes;
    std::vector<hipEvent_t> ev0, ev1;      // kernel start/stop on each device's stream (latest launch)
    std::vector<double> shard_ms;
    uint8_t* root_frame = nullptr;         // staging frame on the root when the destination is host memory
    size_t root_frame_bytes = 0;
    // ---- batched steps
    int gather = KIFS_GATHER_SPARSE;
    int transport_wanted = KIFS_TRANSPORT_AUTO;
    int transport = KIFS_TRANSPORT_AUTO;   // decided at the first step (or by kifs_multi_set_gather)
    std::vector<hipStream_t> comm_stream;  // per device; [0] is the root's gather stream
    std::vector<ncclComm_t> comm;          // per device, RCCL transport only
    StepSlot slot[SLOTS];
    uint64_t next_step = 0;
    bool have_pending = false;             // a submitted step whose transfers are not posted yet
    uint64_t pending = 0;
    std::vector<uint8_t*> outs_scratch;    // destination pointers of one device's launch
    KifsMultiStats stats{};
};

namespace {

bool all_distinct(const std::vector<int>& v) {
    for (size_t i = 0; i < v.size(); ++i)
        for (size_t j = i + 1; j < v.size(); ++j)
            if (v[i] == v[j]) return false;
    return true;
}

// (Re)deal the frame's stripes to the devices.
int multi_partition(kifs_multi* m, int h) {
    if (m->stripes_height == h) return KIFS_OK;
    const int n = int(m->ctx.size());
    const int all = (h + KIFS_STRIPE_ROWS - 1) / KIFS_STRIPE_ROWS;
    for (int i = 0; i < n; ++i) {
        m->stripes[size_t(i)].assign(size_t(all), 0);
        int count = 0, rows = 0;
        int st = kifs_shard_stripes(h, n, m->weight.data(), i, m->stripes[size_t(i)].data(), all, &count, &rows);
        if (st != KIFS_OK) return st;
        m->stripes[size_t(i)].resize(size_t(count));
        m->rows[size_t(i)] = rows;
    }
    m->stripes_height = h;
    return KIFS_OK;
}

// Transport of the batched steps: decided once, communicators created on demand.
int ensure_transport(kifs_multi* m) {
    if (m->transport != KIFS_TRANSPORT_AUTO) return KIFS_OK;
    const int n = int(m->dev.size());
    int want = m->transport_wanted;
    const bool automatic = want == KIFS_TRANSPORT_AUTO;
    if (automatic) want = (n >= 2 && all_distinct(m->dev)) ? KIFS_TRANSPORT_RCCL : KIFS_TRANSPORT_COPY;
    if (want == KIFS_TRANSPORT_RCCL) {
        // An explicit KIFS_TRANSPORT_RCCL that cannot be had is an error; AUTO falls back to peer copies (a node
        // without a loadable librccl, or whose ncclCommInitAll fails, still gathers) and says so under KIFS_DEBUG
        // and in stats.transport.
        auto give_up = [&](const char* why) {
            if (!automatic) return int(KIFS_ERR_COMM);
            if (std::getenv("KIFS_DEBUG")) std::fprintf(stderr, "kifs: %s; KIFS_TRANSPORT_AUTO falls back to peer copies\n", why);
            want = KIFS_TRANSPORT_COPY;
            return int(KIFS_OK);
        };
        int st = KIFS_OK;
        const RcclApi* a = nullptr;
        if (!all_distinct(m->dev)) st = give_up("a device is listed twice (ncclCommInitAll refuses that)");
        else if (!(a = rccl())) st = give_up("librccl could not be opened");
        else {
            m->comm.assign(size_t(n), nullptr);
            if (!nccl_ok(a->CommInitAll(m->comm.data(), n, m->dev.data()), "ncclCommInitAll")) {
                m->comm.clear();
                (void)hipGetLastError();
                st = give_up("ncclCommInitAll failed");
            } else {
                m->stats.rccl_version = a->version;
                m->stats.comm_ranks = n;
            }
        }
        if (st != KIFS_OK) return st;
    }
    m->transport = want;
    m->stats.transport = want;
    return KIFS_OK;
}

int ensure_streams(kifs_multi* m) {
    if (!m->comm_stream.empty()) return KIFS_OK;
    const size_t n = m->dev.size();
    m->comm_stream.assign(n, nullptr);
    for (size_t i = 0; i < n; ++i) {
        DeviceGuard g(m->dev[i]);
        if (!g.ok || !hip_ok(hipStreamCreateWithFlags(&m->comm_stream[i], hipStreamNonBlocking), "comm stream"))
            return KIFS_ERR_RUNTIME;
    }
    return KIFS_OK;
}

bool make_event(hipEvent_t& ev) {
    return ev || hip_ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate(multi)");
}

// One group of point-to-point transfers: device i's `src[i]` (bytes[i] > 0) -> the root's `dst[i]`.  `ready[i]` is
// an event after which src[i] may be read.  The receives (or the copies) are enqueued on the root's gather stream,
// so whatever follows there sees the data -- and once that has completed, every src[i] has been read.
int transfer_to_root(kifs_multi* m, const std::vector<const uint8_t*>& src, const std::vector<uint8_t*>& dst,
                     const std::vector<size_t>& bytes, const std::vector<hipEvent_t>& ready, bool include_root) {
    const int n = int(m->dev.size());
    hipStream_t gstream = m->comm_stream[0];
    const int first = include_root ? 0 : 1;
    if (m->transport == KIFS_TRANSPORT_RCCL) {
        const RcclApi* a = rccl();
        if (!a || m->comm.size() != size_t(n)) return KIFS_ERR_COMM;
        for (int i = first; i < n; ++i) {
            if (!bytes[size_t(i)] || !ready[size_t(i)]) continue;
            DeviceGuard g(m->dev[size_t(i)]);
            hipStream_t s = i == 0 ? gstream : m->comm_stream[size_t(i)];
            if (!hip_ok(hipStreamWaitEvent(s, ready[size_t(i)], 0), "wait(payload ready)")) return KIFS_ERR_RUNTIME;
        }
        if (!nccl_ok(a->GroupStart(), "ncclGroupStart")) return KIFS_ERR_COMM;
        bool ok = true;
        for (int i = first; i < n && ok; ++i) {
            if (!bytes[size_t(i)]) continue;
            hipStream_t s = i == 0 ? gstream : m->comm_stream[size_t(i)];
            ok = nccl_ok(a->Send(src[size_t(i)], bytes[size_t(i)], ncclUint8, 0, m->comm[size_t(i)], s), "ncclSend");
        }
        for (int i = first; i < n && ok; ++i) {
            if (!bytes[size_t(i)]) continue;
            ok = nccl_ok(a->Recv(dst[size_t(i)], bytes[size_t(i)], ncclUint8, i, m->comm[0], gstream), "ncclRecv");
        }
        const bool ended = nccl_ok(a->GroupEnd(), "ncclGroupEnd");
        return ok && ended ? KIFS_OK : KIFS_ERR_COMM;
    }
    // COPY: the root pulls every payload with the copy engines, one peer copy each, on its gather stream
    DeviceGuard g(m->dev[0]);
    for (int i = first; i < n; ++i) {
        if (!bytes[size_t(i)]) continue;
        if (ready[size_t(i)] && !hip_ok(hipStreamWaitEvent(gstream, ready[size_t(i)], 0), "wait(payload ready)"))
            return KIFS_ERR_RUNTIME;
        const hipError_t e = m->dev[size_t(i)] == m->dev[0]
                                 ? hipMemcpyAsync(dst[size_t(i)], src[size_t(i)], bytes[size_t(i)], hipMemcpyDeviceToDevice, gstream)
                                 : hipMemcpyPeerAsync(dst[size_t(i)], m->dev[0], src[size_t(i)], m->dev[size_t(i)],
                                                      bytes[size_t(i)], gstream);
        if (!hip_ok(e, "peer copy of a payload")) return KIFS_ERR_COMM;
    }
    return KIFS_OK;
}

// Posts the transfers of the step in `sl` and the root's scatter of what arrives.
int flush_slot(kifs_multi* m, StepSlot& sl) {
    if (!sl.used || sl.flushed) return KIFS_OK;
    const int n = int(m->dev.size());
    const size_t row_bytes = size_t(sl.width) * 4;
    const bool sparse = sl.gather == KIFS_GATHER_SPARSE;
    std::vector<const uint8_t*> src(size_t(n), nullptr);
    std::vector<uint8_t*> dst(size_t(n), nullptr);
    std::vector<size_t> bytes(size_t(n), 0);
    std::vector<hipEvent_t> ready(size_t(n), nullptr);
    uint64_t records = 0, tiles = 0, payload = 0;
    for (int i = 1; i < n; ++i) {
        PeerPart& p = sl.part[size_t(i)];
        p.n_records = 0;
        p.payload_bytes = 0;
        if (p.stripes.empty()) continue;
        if (sparse) {
            // the count was copied to pinned memory before `packed` was recorded
            if (!hip_ok(hipEventSynchronize(p.packed), "wait(packed)")) return KIFS_ERR_RUNTIME;
            const size_t capacity = size_t(sl.count) * p.stripes.size() * size_t((sl.width + kifs::TILE_W - 1) / kifs::TILE_W);
            if (*p.h_count > capacity) return KIFS_ERR_RUNTIME;
            p.n_records = *p.h_count;
            p.payload_bytes = size_t(p.n_records) * KIFS_SPARSE_RECORD_BYTES;
            src[size_t(i)] = p.records;
            records += p.n_records;
            tiles += capacity;
        } else {
            p.payload_bytes = size_t(sl.count) * size_t(p.rows) * row_bytes;
            src[size_t(i)] = p.shard;
        }
        dst[size_t(i)] = p.recv;
        bytes[size_t(i)] = p.payload_bytes;
        ready[size_t(i)] = p.packed;
        payload += p.payload_bytes;
    }
    int st = transfer_to_root(m, src, dst, bytes, ready, false);
    if (st != KIFS_OK) return st;
    m->stats.records_received += records;  // (counted once the transfers are posted: a failed flush that is
    m->stats.tiles_covered += tiles;       // tried again must not count twice)
    m->stats.bytes_received += payload;
    {   // the root moves what arrived to its rows of the frames
        DeviceGuard g(m->dev[0]);
        kifs_ctx* root = m->ctx[0];
        hipStream_t gstream = m->comm_stream[0];
        for (int i = 1; i < n; ++i) {
            PeerPart& p = sl.part[size_t(i)];
            if (!p.payload_bytes) continue;
            const RowTable* rows = row_table(root, p.stripes.data(), int(p.stripes.size()), sl.height);
            if (!rows) return KIFS_ERR_RUNTIME;
            const hipError_t e =
                sparse ? kifs::launch_unpack_sparse(sl.frames, sl.pitch, sl.stride, reinterpret_cast<const uint32_t*>(p.recv),
                                                    p.n_records, rows->d_rows, int(p.stripes.size()), sl.count, sl.width,
                                                    sl.height, 0, 0u, gstream)
                       : kifs::launch_unpack_stripes(sl.frames, sl.pitch, sl.stride, p.recv, row_bytes, size_t(p.rows) * row_bytes,
                                                     rows->d_rows, int(p.stripes.size()), sl.count, sl.width, sl.height, gstream);
            if (!hip_ok(e, "scatter of a received payload")) return KIFS_ERR_RUNTIME;
        }
        if (!hip_ok(hipEventRecord(sl.gathered, gstream), "record(gathered)")) return KIFS_ERR_RUNTIME;
    }
    sl.flushed = true;
    sl.background_known = sparse && !sl.overwritten;
    if (m->have_pending && m->pending == sl.step) m->have_pending = false;
    return KIFS_OK;
}

// Host-side completion of a slot's step (flushes it first if need be).
int complete_slot(kifs_multi* m, StepSlot& sl) {
    if (!sl.used) return KIFS_OK;
    int st = flush_slot(m, sl);
    if (st != KIFS_OK) return st;
    if (!hip_ok(hipEventSynchronize(sl.rendered), "wait(rendered)") || !hip_ok(hipEventSynchronize(sl.gathered), "wait(gathered)"))
        return KIFS_ERR_RUNTIME;
    sl.used = false;
    m->stats.steps += 1;
    return KIFS_OK;
}

int drain(kifs_multi* m) {
    // oldest first: a step's erase may depend on the step before it in the gather stream
    StepSlot* order[SLOTS];
    for (int s = 0; s < SLOTS; ++s) order[s] = &m->slot[s];
    std::sort(order, order + SLOTS, [](const StepSlot* a, const StepSlot* b) { return a->step < b->step; });
    for (StepSlot* sl : order) {
        int st = complete_slot(m, *sl);
        if (st != KIFS_OK) return st;
    }
    return KIFS_OK;
}

// [frames, end) of what a step writes
inline const uint8_t* frames_end(const uint8_t* frames, int count, size_t pitch, size_t stride, int width, int height) {
    return frames + size_t(count - 1) * stride + size_t(height - 1) * pitch + size_t(width) * 4;
}

// Something is about to write [lo, hi) on the root: a slot whose frames overlap that range no longer knows that they
// hold the background outside its records -- its next submission must fill, whatever KIFS_MULTI_FRAMES_UNTOUCHED
// says (the other slot rendering into the same buffer, a lone kifs_multi_render into it, buffers that overlap).
void forget_background(kifs_multi* m, const StepSlot* except, const uint8_t* lo, const uint8_t* hi) {
    for (StepSlot& o : m->slot) {
        if (&o == except || !o.frames || o.count < 1) continue;
        const uint8_t* olo = o.frames;
        const uint8_t* ohi = frames_end(o.frames, o.count, o.pitch, o.stride, o.width, o.height);
        if (lo < ohi && olo < hi) {
            o.background_known = false;
            o.overwritten = true;  // (also for a flush of that slot's step that is still to come)
        }
    }
}

void free_slot(kifs_multi* m, StepSlot& sl) {
    for (size_t i = 0; i < sl.part.size(); ++i) {
        PeerPart& p = sl.part[i];
        {
            DeviceGuard g(m->dev[i]);
            if (p.shard) (void)hipFree(p.shard);
            if (p.records) (void)hipFree(p.records);
            if (p.d_count) (void)hipFree(p.d_count);
            if (p.h_count) (void)hipHostFree(p.h_count);
            if (p.packed) (void)hipEventDestroy(p.packed);
        }
        if (p.recv) {
            DeviceGuard g(m->dev[0]);
            (void)hipFree(p.recv);
        }
    }
    if (!m->dev.empty()) {
        DeviceGuard g(m->dev[0]);
        if (sl.rendered) (void)hipEventDestroy(sl.rendered);
        if (sl.gathered) (void)hipEventDestroy(sl.gathered);
    }
    sl = StepSlot();
}

}  // namespace

extern "C" {

void kifs_multi_destroy(kifs_multi* m) {
    if (!m) return;
    (void)drain(m);
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        if (!m->ctx[i]) continue;
        DeviceGuard g(m->dev[i]);
        (void)hipStreamSynchronize(m->ctx[i]->stream);
        if (i < m->comm_stream.size() && m->comm_stream[i]) (void)hipStreamSynchronize(m->comm_stream[i]);
    }
    if (!m->comm.empty()) {
        const RcclApi* a = rccl();
        for (ncclComm_t c : m->comm)
            if (a && c) (void)a->CommDestroy(c);
    }
    for (StepSlot& sl : m->slot) free_slot(m, sl);
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        if (!m->ctx[i]) continue;
        DeviceGuard g(m->dev[i]);
        if (i < m->comm_stream.size() && m->comm_stream[i]) (void)hipStreamDestroy(m->comm_stream[i]);
        if (i < m->shard.size() && m->shard[i]) (void)hipFree(m->shard[i]);
        if (i < m->ev0.size() && m->ev0[i]) (void)hipEventDestroy(m->ev0[i]);
        if (i < m->ev1.size() && m->ev1[i]) (void)hipEventDestroy(m->ev1[i]);
    }
    if (!m->dev.empty()) {
        DeviceGuard g(m->dev[0]);
        for (uint8_t* r : m->recv)
            if (r) (void)hipFree(r);
        if (m->root_frame) (void)hipFree(m->root_frame);
    }
    for (kifs_ctx* c : m->ctx) kifs_destroy(c);
    delete m;
}

kifs_multi* kifs_multi_create(const int* devices, int n, int* status) {
    auto fail = [&](int st, kifs_multi* m) -> kifs_multi* {
        if (status) *status = st;
        kifs_multi_destroy(m);
        return nullptr;
    };
    if (!devices || n <= 0 || n > 64) return fail(KIFS_ERR_BAD_ARG, nullptr);
    kifs_multi* m = new (std::nothrow) kifs_multi();
    if (!m) return fail(KIFS_ERR_DEVICE_INIT, nullptr);
    const size_t N = size_t(n);
    m->weight.assign(N, 1);
    m->stripes.assign(N, {});
    m->rows.assign(N, 0);
    m->shard.assign(N, nullptr);
    m->shard_bytes.assign(N, 0);
    m->recv.assign(N, nullptr);
    m->recv_bytes.assign(N, 0);
    m->ev0.assign(N, nullptr);
    m->ev1.assign(N, nullptr);
    m->shard_ms.assign(N, -1.0);
    m->stats.gather = m->gather;
    for (int i = 0; i < n; ++i) {
        int st = KIFS_OK;
        kifs_ctx* c = kifs_create(devices[i], &st);
        if (!c) return fail(st, m);
        m->ctx.push_back(c);
        m->dev.push_back(devices[i]);
        DeviceGuard g(devices[i]);
        if (hipEventCreate(&m->ev0[size_t(i)]) != hipSuccess || hipEventCreate(&m->ev1[size_t(i)]) != hipSuccess)
            return fail(KIFS_ERR_DEVICE_INIT, m);
        if (devices[i] != devices[0]) {  // direct xGMI access both ways; failure only means staged copies
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[i], devices[0]) == hipSuccess && can)
                (void)hipDeviceEnablePeerAccess(devices[0], 0);
            (void)hipGetLastError();
        }
    }
    if (status) *status = KIFS_OK;
    return m;
}

// Uniform changes apply to the steps submitted afterwards; steps in flight are completed first (their buffers and
// row partition belong to the old settings).
#define KIFS_MULTI_FORWARD(call)                 \
    if (!m) return KIFS_ERR_BAD_ARG;             \
    {                                            \
        int dst_ = drain(m);                     \
        if (dst_ != KIFS_OK) return dst_;        \
    }                                            \
    for (kifs_ctx* c : m->ctx) {                 \
        int st = (call);                         \
        if (st != KIFS_OK) return st;            \
    }                                            \
    return KIFS_OK;

int kifs_multi_set_screen(kifs_multi* m, const KifsScreenUniform* s) { KIFS_MULTI_FORWARD(kifs_set_screen(c, s)) }
int kifs_multi_set_camera(kifs_multi* m, const KifsCameraUniform* cam) { KIFS_MULTI_FORWARD(kifs_set_camera(c, cam)) }
int kifs_multi_set_options(kifs_multi* m, const KifsOptionsUniform* o) { KIFS_MULTI_FORWARD(kifs_set_options(c, o)) }
int kifs_multi_set_iters(kifs_multi* m, int a, int b, int f) { KIFS_MULTI_FORWARD(kifs_set_iters(c, a, b, f)) }
int kifs_multi_set_extensions(kifs_multi* m, const KifsExtensions* e) { KIFS_MULTI_FORWARD(kifs_set_extensions(c, e)) }

int kifs_multi_set_weights(kifs_multi* m, const int* weights) {
    if (!m) return KIFS_ERR_BAD_ARG;
    long long total = 0;
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        const int w = weights ? weights[i] : 1;
        if (w < 0 || w > (1 << 20)) return KIFS_ERR_BAD_ARG;
        total += w;
    }
    if (total <= 0) return KIFS_ERR_BAD_ARG;
    int st = drain(m);
    if (st != KIFS_OK) return st;
    for (size_t i = 0; i < m->ctx.size(); ++i) m->weight[i] = weights ? weights[i] : 1;
    m->stripes_height = -1;
    return KIFS_OK;
}

int kifs_multi_shard(kifs_multi* m, int i, int* device, int* n_stripes, int* rows) {
    if (!m || i < 0 || size_t(i) >= m->ctx.size()) return KIFS_ERR_BAD_ARG;
    int w, h;
    if (!m->ctx[0]->have_screen) return KIFS_ERR_UNCONFIGURED;
    int st = frame_dims(m->ctx[0], &w, &h);
    if (st != KIFS_OK) return st;
    st = multi_partition(m, h);
    if (st != KIFS_OK) return st;
    if (device) *device = m->dev[size_t(i)];
    if (n_stripes) *n_stripes = int(m->stripes[size_t(i)].size());
    if (rows) *rows = m->rows[size_t(i)];
    return KIFS_OK;
}

double kifs_multi_shard_ms(kifs_multi* m, int i) {
    if (!m || i < 0 || size_t(i) >= m->shard_ms.size()) return -1.0;
    // batched steps leave their launches' event pairs behind: resolved here, once they have completed
    if (m->shard_ms[size_t(i)] < 0.0 && m->ev0[size_t(i)] && m->ev1[size_t(i)]) {
        DeviceGuard g(m->dev[size_t(i)]);
        float ms = 0.0f;
        if (hipEventQuery(m->ev1[size_t(i)]) == hipSuccess && hipEventElapsedTime(&ms, m->ev0[size_t(i)], m->ev1[size_t(i)]) == hipSuccess)
            m->shard_ms[size_t(i)] = double(ms);
        (void)hipGetLastError();
    }
    return m->shard_ms[size_t(i)];
}

int kifs_multi_render(kifs_multi* m, uint8_t* out, size_t pitch, int encode) {
    if (!m || !out) return KIFS_ERR_BAD_ARG;
    kifs_ctx* root = m->ctx[0];
    if (!root->have_screen || !root->have_camera || !root->have_options) return KIFS_ERR_UNCONFIGURED;
    int w, h;
    int st = frame_dims(root, &w, &h);
    if (st != KIFS_OK) return st;
    const size_t row_bytes = size_t(w) * 4;
    if (pitch < row_bytes || (pitch & 3u)) return KIFS_ERR_BAD_SIZE;
    st = drain(m);  // batched steps in flight use the same contexts and streams
    if (st != KIFS_OK) return st;
    st = multi_partition(m, h);
    if (st != KIFS_OK) return st;
    const int n = int(m->ctx.size());
    forget_background(m, nullptr, out, out + size_t(h - 1) * pitch + row_bytes);  // (a host pointer overlaps nothing)
    // the frame the shards are collected into: the caller's buffer if it is root-device memory
    uint8_t* frame = out;
    size_t fpitch = pitch;
    bool host_dst;
    {
        DeviceGuard g(m->dev[0]);
        host_dst = !is_device_pointer(out);
        if (host_dst) {
            if (!grow(m->root_frame, m->root_frame_bytes, row_bytes * size_t(h), "hipMalloc(multi frame)")) return KIFS_ERR_RUNTIME;
            frame = m->root_frame;
            fpitch = row_bytes;
        }
    }
    // 1. every device renders its shard: the root straight into the frame, the others into a packed buffer
    for (int i = 0; i < n; ++i) {
        kifs_ctx* c = m->ctx[size_t(i)];
        DeviceGuard g(m->dev[size_t(i)]);
        const std::vector<int>& stripes = m->stripes[size_t(i)];
        uint8_t* dst = frame;
        size_t dpitch = fpitch;
        if (i != 0) {
            if (!grow(m->shard[size_t(i)], m->shard_bytes[size_t(i)], row_bytes * size_t(m->rows[size_t(i)]), "hipMalloc(shard)"))
                return KIFS_ERR_RUNTIME;
            dst = m->shard[size_t(i)];
            dpitch = row_bytes;
        }
        if (hipEventRecord(m->ev0[size_t(i)], c->stream) != hipSuccess) return KIFS_ERR_RUNTIME;
        if (!stripes.empty()) {
            st = enqueue_batch(c, c->stream, 1, nullptr, &dst, dpitch, 0, h, encode, stripes.data(), int(stripes.size()),
                               i == 0 ? 1 : 0);
            if (st != KIFS_OK) return st;
        }
        if (hipEventRecord(m->ev1[size_t(i)], c->stream) != hipSuccess) return KIFS_ERR_RUNTIME;
    }
    // 2. the root pulls each finished shard over xGMI and moves its stripes to their frame rows
    {
        DeviceGuard g(m->dev[0]);
        for (int i = 1; i < n; ++i) {
            const std::vector<int>& stripes = m->stripes[size_t(i)];
            if (stripes.empty()) continue;
            const size_t bytes = row_bytes * size_t(m->rows[size_t(i)]);
            if (!grow(m->recv[size_t(i)], m->recv_bytes[size_t(i)], bytes, "hipMalloc(received shard)")) return KIFS_ERR_RUNTIME;
            if (hipStreamWaitEvent(root->stream, m->ev1[size_t(i)], 0) != hipSuccess) return KIFS_ERR_RUNTIME;
            if (!hip_ok(hipMemcpyPeerAsync(m->recv[size_t(i)], m->dev[0], m->shard[size_t(i)], m->dev[size_t(i)], bytes,
                                           root->stream), "peer copy of a shard"))
                return KIFS_ERR_COMM;
            const RowTable* rows = row_table(root, stripes.data(), int(stripes.size()), h);
            if (!rows) return KIFS_ERR_RUNTIME;
            if (!hip_ok(kifs::launch_unpack_stripes(frame, fpitch, 0, m->recv[size_t(i)], row_bytes, 0, rows->d_rows,
                                                    int(stripes.size()), 1, w, h, root->stream), "unpack_stripes_kernel launch"))
                return KIFS_ERR_RUNTIME;
        }
        if (host_dst &&
            !hip_ok(hipMemcpy2DAsync(out, pitch, frame, fpitch, row_bytes, size_t(h), hipMemcpyDeviceToHost,
                                     root->stream), "frame to host"))
            return KIFS_ERR_RUNTIME;
        if (!hip_ok(hipStreamSynchronize(root->stream), "multi sync")) return KIFS_ERR_RUNTIME;
    }
    for (int i = 0; i < n; ++i) {
        DeviceGuard g(m->dev[size_t(i)]);
        if (hipStreamSynchronize(m->ctx[size_t(i)]->stream) != hipSuccess) return KIFS_ERR_RUNTIME;
        float ms = 0.0f;
        m->shard_ms[size_t(i)] =
            hipEventElapsedTime(&ms, m->ev0[size_t(i)], m->ev1[size_t(i)]) == hipSuccess ? double(ms) : -1.0;
    }
    return KIFS_OK;
}

// ---- batched steps ---------------------------------------------------------------------------------------

int kifs_multi_set_gather(kifs_multi* m, int gather, int transport) {
    if (!m || (gather != KIFS_GATHER_SPARSE && gather != KIFS_GATHER_DENSE) ||
        (transport != KIFS_TRANSPORT_AUTO && transport != KIFS_TRANSPORT_RCCL && transport != KIFS_TRANSPORT_COPY))
        return KIFS_ERR_BAD_ARG;
    int st = drain(m);
    if (st != KIFS_OK) return st;
    m->gather = gather;
    m->stats.gather = gather;
    if (transport != m->transport_wanted || (transport != KIFS_TRANSPORT_AUTO && transport != m->transport)) {
        // a different transport: communicators go, the next step (or the lines below) decides anew
        if (!m->comm.empty()) {
            const RcclApi* a = rccl();
            for (ncclComm_t c : m->comm)
                if (a && c) (void)a->CommDestroy(c);
            m->comm.clear();
            m->stats.comm_ranks = 0;
        }
        m->transport_wanted = transport;
        m->transport = KIFS_TRANSPORT_AUTO;
        m->stats.transport = KIFS_TRANSPORT_AUTO;
    }
    for (StepSlot& sl : m->slot) sl.background_known = false;
    if (transport == KIFS_TRANSPORT_AUTO) return KIFS_OK;
    st = ensure_streams(m);  // an explicit transport is set up now, so that its failure is reported here
    return st != KIFS_OK ? st : ensure_transport(m);
}

int kifs_multi_render_batch_async(kifs_multi* m, int count, const KifsCameraUniform* cameras, uint8_t* dev_frames,
                                  size_t frame_pitch, size_t frame_stride, int encode, int flags, uint64_t* step_out) {
    if (!m || !cameras || !dev_frames || count < 1 || count > KIFS_MAX_BATCH) return KIFS_ERR_BAD_ARG;
    if (encode != KIFS_ENCODE_UNORM && encode != KIFS_ENCODE_SRGB) return KIFS_ERR_BAD_ARG;
    kifs_ctx* root = m->ctx[0];
    if (!root->have_screen || !root->have_options) return KIFS_ERR_UNCONFIGURED;
    int w, h;
    int st = frame_dims(root, &w, &h);
    if (st != KIFS_OK) return st;
    const size_t row_bytes = size_t(w) * 4;
    if (frame_pitch < row_bytes || ((frame_pitch | frame_stride) & 3u) || (reinterpret_cast<uintptr_t>(dev_frames) & 3u) ||
        (count > 1 && frame_stride < frame_pitch * size_t(h - 1) + row_bytes))
        return KIFS_ERR_BAD_SIZE;
    {
        DeviceGuard g(m->dev[0]);
        if (!g.ok) return KIFS_ERR_RUNTIME;
        if (!is_device_pointer(dev_frames)) return KIFS_ERR_BAD_ARG;
    }
    if ((st = ensure_streams(m)) != KIFS_OK || (st = ensure_transport(m)) != KIFS_OK || (st = multi_partition(m, h)) != KIFS_OK)
        return st;
    const int n = int(m->ctx.size());
    const uint64_t step = m->next_step;
    StepSlot& sl = m->slot[step % SLOTS];
    // the slot's previous step (k - 2) ends here: its consumer has had it since the wait, or never asked
    if ((st = complete_slot(m, sl)) != KIFS_OK) return st;
    const bool sparse = m->gather == KIFS_GATHER_SPARSE;
    const float* bc = root->options.background_color;
    const uint32_t background = background_pixel(root, kifs::V3{bc[0], bc[1], bc[2]}, encode);
    // may the frames be assumed to hold the background everywhere but under the slot's last records?
    bool same_parts = sl.part.size() == size_t(n);
    for (int i = 0; same_parts && i < n; ++i) same_parts = sl.part[size_t(i)].stripes == m->stripes[size_t(i)];
    const bool erase_only = sparse && (flags & KIFS_MULTI_FRAMES_UNTOUCHED) && sl.background_known && same_parts &&
                            sl.frames == dev_frames && sl.pitch == frame_pitch && sl.stride == frame_stride &&
                            sl.count == count && sl.encode == encode && sl.background == background &&
                            sl.width == w && sl.height == h && sl.gather == KIFS_GATHER_SPARSE;
    // (erase_only looked at this slot's own record of the buffer; the OTHER slot's record of any buffer this step
    // writes into is void from here on)
    forget_background(m, &sl, dev_frames, frames_end(dev_frames, count, frame_pitch, frame_stride, w, h));
    if (sl.part.size() != size_t(n)) sl.part.resize(size_t(n));
    {
        DeviceGuard g(m->dev[0]);
        if (!make_event(sl.rendered) || !make_event(sl.gathered)) return KIFS_ERR_RUNTIME;
    }
    // ---- the root's gather stream: background under everybody else's rows
    hipStream_t gstream = m->comm_stream[0];
    if (sparse && n > 1) {
        DeviceGuard g(m->dev[0]);
        if (erase_only) {
            for (int i = 1; i < n; ++i) {
                PeerPart& p = sl.part[size_t(i)];
                if (!p.n_records) continue;
                const RowTable* rows = row_table(root, p.stripes.data(), int(p.stripes.size()), h);
                if (!rows) return KIFS_ERR_RUNTIME;
                if (!hip_ok(kifs::launch_unpack_sparse(dev_frames, frame_pitch, frame_stride, reinterpret_cast<const uint32_t*>(p.recv),
                                                       p.n_records, rows->d_rows, int(p.stripes.size()), count, w, h, 1, background,
                                                       gstream), "erase of the previous records"))
                    return KIFS_ERR_RUNTIME;
            }
        } else {
            sl.peer_stripes.clear();
            for (int i = 1; i < n; ++i) sl.peer_stripes.insert(sl.peer_stripes.end(), m->stripes[size_t(i)].begin(), m->stripes[size_t(i)].end());
            std::sort(sl.peer_stripes.begin(), sl.peer_stripes.end());
            if (!sl.peer_stripes.empty()) {
                const RowTable* rows = row_table(root, sl.peer_stripes.data(), int(sl.peer_stripes.size()), h);
                if (!rows) return KIFS_ERR_RUNTIME;
                if (!hip_ok(kifs::launch_fill_stripes(dev_frames, frame_pitch, frame_stride, rows->d_rows, int(sl.peer_stripes.size()),
                                                      count, w, h, background, gstream), "fill under the other devices' rows"))
                    return KIFS_ERR_RUNTIME;
            }
        }
    }
    sl.used = true;
    sl.flushed = false;
    sl.background_known = false;
    sl.overwritten = false;
    sl.step = step;
    sl.count = count; sl.encode = encode; sl.gather = m->gather; sl.width = w; sl.height = h;
    sl.frames = dev_frames; sl.pitch = frame_pitch; sl.stride = frame_stride; sl.background = background;
    // ---- every device: one launch for its shard of all the step's frames, then its payload
    const size_t tiles_x = size_t((w + kifs::TILE_W - 1) / kifs::TILE_W);
    m->outs_scratch.resize(size_t(count));
    auto launch_all = [&]() -> int {
        for (int i = 0; i < n; ++i) {
            kifs_ctx* c = m->ctx[size_t(i)];
            PeerPart& p = sl.part[size_t(i)];
            p.stripes = m->stripes[size_t(i)];
            p.rows = m->rows[size_t(i)];
            p.n_records = 0;
            p.payload_bytes = 0;
            DeviceGuard g(m->dev[size_t(i)]);
            if (!g.ok) return KIFS_ERR_RUNTIME;
            m->shard_ms[size_t(i)] = -1.0;
            if (p.stripes.empty()) {
                if (i == 0 && !hip_ok(hipEventRecord(sl.rendered, c->stream), "record(rendered)")) return KIFS_ERR_RUNTIME;
                continue;
            }
            const size_t shard_stride = size_t(p.rows) * row_bytes;
            if (i == 0) {
                for (int f = 0; f < count; ++f) m->outs_scratch[size_t(f)] = dev_frames + size_t(f) * frame_stride;
            } else {
                const size_t need = shard_stride * size_t(count);
                const size_t capacity = size_t(count) * p.stripes.size() * tiles_x;
                if (!make_event(p.packed)) return KIFS_ERR_RUNTIME;
                // buffers that grow are replaced while nothing reads them: the slot's previous step is complete
                if (!grow(p.shard, p.shard_bytes, need, "hipMalloc(step shards)")) return KIFS_ERR_RUNTIME;
                if (sparse) {
                    if (!grow(p.records, p.records_bytes, capacity * KIFS_SPARSE_RECORD_BYTES, "hipMalloc(step records)")) return KIFS_ERR_RUNTIME;
                    if (!p.d_count && !hip_ok(hipMalloc(reinterpret_cast<void**>(&p.d_count), sizeof(uint32_t)), "hipMalloc(record count)"))
                        return KIFS_ERR_RUNTIME;
                    if (!p.h_count && !hip_ok(hipHostMalloc(reinterpret_cast<void**>(&p.h_count), sizeof(uint32_t), hipHostMallocDefault),
                                              "hipHostMalloc(record count)"))
                        return KIFS_ERR_RUNTIME;
                }
                {
                    DeviceGuard gr(m->dev[0]);
                    if (!grow(p.recv, p.recv_bytes, sparse ? capacity * KIFS_SPARSE_RECORD_BYTES : need, "hipMalloc(step receive)"))
                        return KIFS_ERR_RUNTIME;
                }
                // (the payload of the slot's previous step has left these buffers: that step was completed above, and
                // its completion includes the root's scatter, which follows the transfer in the gather stream)
                for (int f = 0; f < count; ++f) m->outs_scratch[size_t(f)] = p.shard + size_t(f) * shard_stride;
            }
            if (!hip_ok(hipEventRecord(m->ev0[size_t(i)], c->stream), "record(launch start)")) return KIFS_ERR_RUNTIME;
            st = enqueue_batch(c, c->stream, count, cameras, m->outs_scratch.data(), i == 0 ? frame_pitch : row_bytes, 0, h, encode,
                               p.stripes.data(), int(p.stripes.size()), i == 0 ? 1 : 0);
            if (st != KIFS_OK) return st;
            if (!hip_ok(hipEventRecord(m->ev1[size_t(i)], c->stream), "record(launch stop)")) return KIFS_ERR_RUNTIME;
            if (i == 0) {
                if (!hip_ok(hipEventRecord(sl.rendered, c->stream), "record(rendered)")) return KIFS_ERR_RUNTIME;
                continue;
            }
            if (sparse) {
                const RowTable* rows = row_table(c, p.stripes.data(), int(p.stripes.size()), h);
                if (!rows) return KIFS_ERR_RUNTIME;
                if (!hip_ok(hipMemsetAsync(p.d_count, 0, sizeof(uint32_t), c->stream), "memset(record count)") ||
                    !hip_ok(kifs::launch_pack_sparse(p.shard, row_bytes, shard_stride, rows->d_rows, int(p.stripes.size()), count, w, h,
                                                     background, reinterpret_cast<uint32_t*>(p.records), p.d_count, c->stream),
                            "pack_sparse_kernel launch") ||
                    !hip_ok(hipMemcpyAsync(p.h_count, p.d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream), "copy(record count)"))
                    return KIFS_ERR_RUNTIME;
            }
            if (!hip_ok(hipEventRecord(p.packed, c->stream), "record(packed)")) return KIFS_ERR_RUNTIME;
        }
        return KIFS_OK;
    };
    st = launch_all();
    if (st == KIFS_OK && n == 1) {
        DeviceGuard g(m->dev[0]);
        if (hip_ok(hipEventRecord(sl.gathered, gstream), "record(gathered)")) sl.flushed = true;
        else st = KIFS_ERR_RUNTIME;
    }
    if (st != KIFS_OK) {
        // A step that was only partly enqueued is no step: whatever was launched is waited for (it writes into the
        // caller's frames and the slot's buffers), the slot is free again, the step number is not used up -- a later
        // submit, wait or destroy must not flush record counts and events of a launch that never happened.
        for (int i = 0; i < n; ++i) {
            DeviceGuard g(m->dev[size_t(i)]);
            (void)hipStreamSynchronize(m->ctx[size_t(i)]->stream);
        }
        {
            DeviceGuard g(m->dev[0]);
            (void)hipStreamSynchronize(gstream);
        }
        (void)hipGetLastError();
        sl.used = false;
        sl.flushed = false;
        sl.background_known = false;
        return st;
    }
    m->next_step = step + 1;
    if (step_out) *step_out = step;
    // ---- the step before this one: its senders have had a whole submission to pack
    if (m->have_pending) {
        StepSlot& prev = m->slot[m->pending % SLOTS];
        if (prev.used && prev.step == m->pending && (st = flush_slot(m, prev)) != KIFS_OK) return st;
        m->have_pending = false;
    }
    if (!sl.flushed) {
        m->have_pending = true;
        m->pending = step;
    }
    return KIFS_OK;
}

int kifs_multi_wait(kifs_multi* m, uint64_t step) {
    if (!m) return KIFS_ERR_BAD_ARG;
    if (step >= m->next_step) return KIFS_ERR_BAD_ARG;
    StepSlot& sl = m->slot[step % SLOTS];
    if (!sl.used || sl.step != step) return KIFS_OK;  // completed earlier (by a later submission or a wait)
    // an older step still unflushed goes first: the gather stream runs them in order
    StepSlot& other = m->slot[(step + 1) % SLOTS];
    if (other.used && other.step < step) {
        int st = flush_slot(m, other);
        if (st != KIFS_OK) return st;
    }
    return complete_slot(m, sl);
}

int kifs_multi_wait_all(kifs_multi* m) { return m ? drain(m) : KIFS_ERR_BAD_ARG; }

int kifs_multi_stream_wait(kifs_multi* m, uint64_t step, void* hip_stream) {
    if (!m || !hip_stream || step >= m->next_step) return KIFS_ERR_BAD_ARG;
    StepSlot& sl = m->slot[step % SLOTS];
    if (!sl.used || sl.step != step) return KIFS_OK;
    StepSlot& other = m->slot[(step + 1) % SLOTS];
    int st = KIFS_OK;
    if (other.used && other.step < step && (st = flush_slot(m, other)) != KIFS_OK) return st;
    if ((st = flush_slot(m, sl)) != KIFS_OK) return st;
    DeviceGuard g(m->dev[0]);
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    return hip_ok(hipStreamWaitEvent(s, sl.rendered, 0), "stream wait(rendered)") &&
                   hip_ok(hipStreamWaitEvent(s, sl.gathered, 0), "stream wait(gathered)")
               ? KIFS_OK : KIFS_ERR_RUNTIME;
}

int kifs_multi_order_after(kifs_multi* m, void* producer_stream) {
    if (!m) return KIFS_ERR_BAD_ARG;
    int st = ensure_streams(m);
    if (st != KIFS_OK) return st;
    // the root's render stream first (kifs_order_after records the context's ordering event on the producer's
    // stream), then the same event for the gather stream
    if ((st = kifs_order_after(m->ctx[0], nullptr, producer_stream)) != KIFS_OK) return st;
    if (m->ctx[0]->stream == static_cast<hipStream_t>(producer_stream)) return KIFS_OK;
    DeviceGuard g(m->dev[0]);
    return hip_ok(hipStreamWaitEvent(m->comm_stream[0], m->ctx[0]->ev_order, 0), "wait(multi order_after)") ? KIFS_OK
                                                                                                          : KIFS_ERR_RUNTIME;
}

int kifs_multi_render_batch(kifs_multi* m, int count, const KifsCameraUniform* cameras, uint8_t* dev_frames,
                            size_t frame_pitch, size_t frame_stride, int encode) {
    uint64_t step = 0;
    int st = kifs_multi_render_batch_async(m, count, cameras, dev_frames, frame_pitch, frame_stride, encode, 0, &step);
    return st != KIFS_OK ? st : kifs_multi_wait(m, step);
}

int kifs_multi_stats(kifs_multi* m, KifsMultiStats* out, int reset) {
    if (!m || !out) return KIFS_ERR_BAD_ARG;
    *out = m->stats;
    if (reset) {
        m->stats.steps = 0;
        m->stats.records_received = 0;
        m->stats.tiles_covered = 0;
        m->stats.bytes_received = 0;
    }
    return KIFS_OK;
}

int kifs_multi_comm_selftest(kifs_multi* m, size_t bytes) {
    if (!m || bytes == 0 || bytes > (size_t(1) << 30)) return KIFS_ERR_BAD_ARG;
    int st = drain(m);
    if (st != KIFS_OK || (st = ensure_streams(m)) != KIFS_OK || (st = ensure_transport(m)) != KIFS_OK) return st;
    const int n = int(m->dev.size());
    const bool self = n == 1;  // one device: the root sends to itself and receives from itself in one group
    std::vector<uint8_t*> src(size_t(n), nullptr), dst(size_t(n), nullptr);
    std::vector<const uint8_t*> csrc(size_t(n), nullptr);
    std::vector<size_t> sizes(size_t(n), 0);
    std::vector<hipEvent_t> none(size_t(n), nullptr);
    std::vector<uint8_t> pattern(bytes), back(bytes);
    int rc = KIFS_OK;
    for (int i = self ? 0 : 1; i < n && rc == KIFS_OK; ++i) {
        for (size_t k = 0; k < bytes; ++k) pattern[k] = uint8_t((k * 131u + size_t(i) * 29u + 7u) & 255u);
        {
            DeviceGuard g(m->dev[size_t(i)]);
            if (!hip_ok(hipMalloc(reinterpret_cast<void**>(&src[size_t(i)]), bytes), "hipMalloc(selftest)") ||
                !hip_ok(hipMemcpy(src[size_t(i)], pattern.data(), bytes, hipMemcpyHostToDevice), "hipMemcpy(selftest)"))
                rc = KIFS_ERR_RUNTIME;
        }
        DeviceGuard g(m->dev[0]);
        if (rc == KIFS_OK && (!hip_ok(hipMalloc(reinterpret_cast<void**>(&dst[size_t(i)]), bytes), "hipMalloc(selftest)") ||
                              !hip_ok(hipMemset(dst[size_t(i)], 0, bytes), "hipMemset(selftest)")))
            rc = KIFS_ERR_RUNTIME;
        csrc[size_t(i)] = src[size_t(i)];
        sizes[size_t(i)] = bytes;
    }
    if (rc == KIFS_OK) rc = transfer_to_root(m, csrc, dst, sizes, none, self);
    if (rc == KIFS_OK) {
        for (int i = 0; i < n; ++i) {
            DeviceGuard g(m->dev[size_t(i)]);
            if (!hip_ok(hipStreamSynchronize(m->comm_stream[size_t(i)]), "sync(selftest)")) rc = KIFS_ERR_COMM;
        }
    }
    for (int i = self ? 0 : 1; i < n && rc == KIFS_OK; ++i) {
        DeviceGuard g(m->dev[0]);
        if (!hip_ok(hipMemcpy(back.data(), dst[size_t(i)], bytes, hipMemcpyDeviceToHost), "hipMemcpy(selftest back)")) {
            rc = KIFS_ERR_RUNTIME;
            break;
        }
        for (size_t k = 0; k < bytes; ++k)
            if (back[k] != uint8_t((k * 131u + size_t(i) * 29u + 7u) & 255u)) {
                rc = KIFS_ERR_COMM;
                break;
            }
    }
    for (int i = 0; i < n; ++i) {
        if (src[size_t(i)]) {
            DeviceGuard g(m->dev[size_t(i)]);
            (void)hipFree(src[size_t(i)]);
        }
        if (dst[size_t(i)]) {
            DeviceGuard g(m->dev[0]);
            (void)hipFree(dst[size_t(i)]);
        }
    }
    return rc;
}

}  // extern "C"
