// splat_multi.hip -- multi-GPU layer of the C ABI (include/splat_hip.h, "Multi-GPU"): one frame as disjoint
// tile-row slabs, one per GPU, gathered to the root over xGMI with a grouped ncclSend / ncclRecv.
// Host code only; the kernels are the single-GPU ones (splat_set_slab restricts a context to its rows).
//
//   (A) splat_comm_*   one process per GPU: a communicator attached to an ordinary context
//   (B) splat_multi_*  one process: one host thread + context per device, ncclCommInitAll
//
// RCCL is loaded with dlopen on first use (a single-GPU caller never pays for it, and a process that
// already carries a librccl.so.1 -- PyTorch's -- shares it).
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "splat_internal.h"

using namespace splat;

namespace {

// ------------------------------------------------------------------------------------------ RCCL, lazily
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {std::getenv("SPLAT_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) {
            if (!nm || !*nm) continue;
            r.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            const char* de = dlerror();        // (once: the call clears the error state)
            r.error = std::string("cannot load RCCL: ") + (de ? de : "librccl.so.1 not found");
            return;
        }
#define SPLAT_SYM(field, name)                                                                   \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name));                         \
    if (!r.field && r.error.empty()) r.error = std::string("RCCL symbol missing: ") + name;
        SPLAT_SYM(GetUniqueId, "ncclGetUniqueId") SPLAT_SYM(CommInitRank, "ncclCommInitRank")
        SPLAT_SYM(CommInitAll, "ncclCommInitAll") SPLAT_SYM(CommDestroy, "ncclCommDestroy") SPLAT_SYM(CommAbort, "ncclCommAbort")
        SPLAT_SYM(GroupStart, "ncclGroupStart") SPLAT_SYM(GroupEnd, "ncclGroupEnd") SPLAT_SYM(Send, "ncclSend")
        SPLAT_SYM(Recv, "ncclRecv") SPLAT_SYM(GetErrorString, "ncclGetErrorString")
#undef SPLAT_SYM
    });
    return &r;
}

thread_local std::string g_multi_create_error;

inline void slab_px(const int32_t* slab, int h, int* a, int* b) {
    *a = std::min(slab[0] * TILE, h);
    *b = std::min(slab[1] * TILE, h);
}

}  // namespace

// ------------------------------------------------------------------------------------------ per-context state
namespace splat {
struct CommState {
    ncclComm_t comm = nullptr;
    int n_ranks = 0, rank = -1;
    std::vector<int32_t> slabs;      // n_ranks x {row0, row1}
    std::atomic<bool> aborted{false}; // ncclCommAbort has run on `comm` (a peer could not take part in a gather): it
                                      // is gone, later gathers fail instead of blocking
    bool loopback = false;            // test hook (splat_comm_loopback): see splat_comm_gather
    std::mutex mu;                    // held by a gather from its `aborted` check to the end of its enqueue (see abort_all_comms)
};
void comm_release(CommState* s) {
    if (!s) return;
    if (s->comm && !s->aborted.load() && rccl()->CommDestroy) (void)rccl()->CommDestroy(s->comm);
    delete s;
}
}  // namespace splat

namespace {
int nccl_fail(splat_ctx* c, const char* what, ncclResult_t e) {
    std::string msg = std::string(what) + ": " + (rccl()->GetErrorString ? rccl()->GetErrorString(e) : "RCCL error");
    return ctx_fail(c, SPLAT_ERR_HIP, msg.c_str());
}

// The gather of one frame on `stream`: peers send their slab's pixel rows (contiguous in a row-major image)
// straight out of their image, the root receives straight into its own.  Ragged sizes, no staging.
int gather_rows(splat_ctx* c, CommState* st, uint32_t* img, int w, int h, int root, hipStream_t stream) {
    Rccl* R = rccl();
    ncclResult_t e = R->GroupStart();
    if (e != ncclSuccess) return nccl_fail(c, "ncclGroupStart", e);
    if (st->rank == root) {
        for (int r = 0; r < st->n_ranks && e == ncclSuccess; ++r) {
            if (r == root) continue;
            int a, b;
            slab_px(&st->slabs[2 * r], h, &a, &b);
            if (b > a) e = R->Recv(img + (size_t)a * w, (size_t)(b - a) * w, ncclUint32, r, st->comm, stream);
        }
    } else {
        int a, b;
        slab_px(&st->slabs[2 * st->rank], h, &a, &b);
        if (b > a) e = R->Send(img + (size_t)a * w, (size_t)(b - a) * w, ncclUint32, root, st->comm, stream);
    }
    ncclResult_t e2 = R->GroupEnd();
    if (e != ncclSuccess) return nccl_fail(c, "ncclSend/ncclRecv", e);
    if (e2 != ncclSuccess) return nccl_fail(c, "ncclGroupEnd", e2);
    return SPLAT_OK;
}

// Test hook: rank 0 of a one-rank communicator sends its slab rows to ITSELF through RCCL: the rows travel out of the
// image into a scratch buffer (ncclSend / ncclRecv, same group, same stream), the image rows are then overwritten
// with a marker and restored from the scratch copy -- so the caller can tell from the pixels that they went through
// the wire and arrived intact, and a profile of the run shows RCCL's send/recv kernel.
int loopback_rows(splat_ctx* c, CommState* st, uint32_t* img, int w, int h, hipStream_t stream) {
    Rccl* R = rccl();
    int a, b;
    slab_px(&st->slabs[0], h, &a, &b);
    if (b <= a) return SPLAT_OK;
    const size_t count = (size_t)(b - a) * w;
    uint32_t* scratch = nullptr;
    if (hipMalloc(&scratch, count * 4) != hipSuccess) return ctx_fail(c, SPLAT_ERR_HIP, "hipMalloc(loopback scratch)");
    int rc = SPLAT_OK;
    ncclResult_t e = R->GroupStart();
    if (e != ncclSuccess) rc = nccl_fail(c, "ncclGroupStart", e);
    if (rc == SPLAT_OK) {
        e = R->Send(img + (size_t)a * w, count, ncclUint32, 0, st->comm, stream);
        ncclResult_t e1 = R->Recv(scratch, count, ncclUint32, 0, st->comm, stream);
        ncclResult_t e2 = R->GroupEnd();
        if (e != ncclSuccess) rc = nccl_fail(c, "ncclSend (loopback)", e);
        else if (e1 != ncclSuccess) rc = nccl_fail(c, "ncclRecv (loopback)", e1);
        else if (e2 != ncclSuccess) rc = nccl_fail(c, "ncclGroupEnd", e2);
    }
    if (rc == SPLAT_OK) {
        hipError_t he = hipMemsetAsync(img + (size_t)a * w, 0x5a, count * 4, stream);             // marker: the rows are gone ...
        if (he == hipSuccess) he = hipMemcpyAsync(img + (size_t)a * w, scratch, count * 4, hipMemcpyDeviceToDevice, stream);   // ... and back, from what RCCL delivered
        if (he == hipSuccess) he = hipStreamSynchronize(stream);
        if (he != hipSuccess) rc = ctx_fail(c, SPLAT_ERR_HIP, hipGetErrorString(he));
    } else {
        (void)hipStreamSynchronize(stream);
    }
    (void)hipFree(scratch);
    return rc;
}
}  // namespace

extern "C" {

// Linear partition (minimise the heaviest slab) by bisection on the bottleneck; ranks get at least one row while
// rows last.  The same arithmetic as splat_amd/dist.py:slab_partition_balanced, which the tests hold it against.
int splat_slab_partition(const uint64_t* row_loads, int32_t n_rows, int32_t n_ranks, double row_overhead, int32_t* out) {
    if (n_rows < 0 || n_ranks <= 0 || !out) return SPLAT_ERR_INVALID;
    const int n = n_rows, k = n_ranks;
    if (!row_loads) {                       // equal split, earlier ranks take the extra row
        int base = n / k, extra = n % k, r = 0;
        for (int i = 0; i < k; ++i) { int cnt = base + (i < extra ? 1 : 0); out[2 * i] = r; out[2 * i + 1] = r + cnt; r += cnt; }
        return SPLAT_OK;
    }
    if (k >= n) {
        for (int i = 0; i < k; ++i) { out[2 * i] = std::min(i, n); out[2 * i + 1] = std::min(i + 1, n); }
        return SPLAT_OK;
    }
    std::vector<double> loads(n);
    double lo = 0.0, hi = 0.0;
    for (int i = 0; i < n; ++i) { loads[i] = (double)row_loads[i] + row_overhead; lo = std::max(lo, loads[i]); hi += loads[i]; }
    auto cuts = [&](double limit, std::vector<std::pair<int, int>>& s) {
        s.clear();
        double acc = 0.0;
        int start = 0;
        for (int i = 0; i < n; ++i) {
            if (acc + loads[i] > limit && i > start) { s.emplace_back(start, i); start = i; acc = 0.0; }
            acc += loads[i];
        }
        s.emplace_back(start, n);
    };
    std::vector<std::pair<int, int>> s;
    for (int it = 0; it < 60; ++it) {
        const double mid = 0.5 * (lo + hi);
        cuts(mid, s);
        if ((int)s.size() <= k) hi = mid; else lo = mid;
    }
    cuts(hi, s);
    auto sum = [&](int a, int b) { double t = 0; for (int i = a; i < b; ++i) t += loads[i]; return t; };
    while ((int)s.size() < k) {             // fewer slabs than ranks: split the heaviest splittable one
        int best = -1;
        for (int j = 0; j < (int)s.size(); ++j) {
            if (s[j].second - s[j].first < 2) continue;
            if (best < 0 || sum(s[j].first, s[j].second) > sum(s[best].first, s[best].second)) best = j;
        }
        if (best < 0) break;
        const int a = s[best].first, b = s[best].second;
        const double half = 0.5 * sum(a, b);
        double acc = 0.0;
        int cut = a + 1;
        for (int i = a; i < b - 1; ++i) { acc += loads[i]; cut = i + 1; if (acc >= half) break; }
        s[best] = {a, cut};
        s.insert(s.begin() + best + 1, {cut, b});
    }
    while ((int)s.size() < k) s.emplace_back(n, n);
    for (int i = 0; i < k; ++i) { out[2 * i] = s[i].first; out[2 * i + 1] = s[i].second; }
    return SPLAT_OK;
}

int splat_comm_unique_id(uint8_t id[SPLAT_UNIQUE_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == SPLAT_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!id) return SPLAT_ERR_INVALID;
    Rccl* R = rccl();
    if (!R->error.empty()) { g_multi_create_error = R->error; return SPLAT_ERR_HIP; }
    ncclUniqueId u;
    ncclResult_t e = R->GetUniqueId(&u);
    if (e != ncclSuccess) { g_multi_create_error = std::string("ncclGetUniqueId: ") + R->GetErrorString(e); return SPLAT_ERR_HIP; }
    std::memcpy(id, &u, sizeof u);
    return SPLAT_OK;
}

int splat_comm_init_rank(splat_ctx* c, const uint8_t id[SPLAT_UNIQUE_ID_BYTES], int32_t n_ranks, int32_t rank) {
    if (!c) return SPLAT_ERR_INVALID;
    if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return ctx_fail(c, SPLAT_ERR_INVALID, "bad communicator arguments");
    Rccl* R = rccl();
    if (!R->error.empty()) return ctx_fail(c, SPLAT_ERR_HIP, R->error.c_str());
    splat_comm_destroy(c);
    if (hipSetDevice(ctx_device(c)) != hipSuccess) return ctx_fail(c, SPLAT_ERR_HIP, "hipSetDevice");
    std::unique_ptr<CommState> st(new CommState());
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    ncclResult_t e = R->CommInitRank(&st->comm, n_ranks, u, rank);
    if (e != ncclSuccess) return nccl_fail(c, "ncclCommInitRank", e);
    st->n_ranks = n_ranks; st->rank = rank;
    *ctx_comm_slot(c) = st.release();
    return SPLAT_OK;
}

int splat_comm_set_slabs(splat_ctx* c, const int32_t* slabs) {
    if (!c) return SPLAT_ERR_INVALID;
    CommState* st = *ctx_comm_slot(c);
    if (!st) return ctx_fail(c, SPLAT_ERR_INVALID, "no communicator on this context");
    if (!slabs) return ctx_fail(c, SPLAT_ERR_INVALID, "slabs is NULL");
    for (int r = 0; r < st->n_ranks; ++r)
        if (slabs[2 * r] < 0 || slabs[2 * r + 1] < slabs[2 * r] || (r && slabs[2 * r] < slabs[2 * r - 1]))
            return ctx_fail(c, SPLAT_ERR_INVALID, "slabs must be ordered, disjoint tile-row ranges");
    st->slabs.assign(slabs, slabs + 2 * st->n_ranks);
    return splat_set_slab(c, slabs[2 * st->rank], slabs[2 * st->rank + 1]);
}

int splat_comm_gather(splat_ctx* c, void* d_argb, int32_t w, int32_t h, int32_t root) {
    if (!c) return SPLAT_ERR_INVALID;
    CommState* st = *ctx_comm_slot(c);
    if (!st || st->slabs.empty()) return ctx_fail(c, SPLAT_ERR_INVALID, "no communicator / partition on this context");
    if (!d_argb || w < 1 || h < 1 || root < 0 || root >= st->n_ranks) return ctx_fail(c, SPLAT_ERR_INVALID, "bad gather arguments");
    std::lock_guard<std::mutex> in_gather(st->mu);       // check-then-use of the communicator is one step for abort_all_comms
    if (st->aborted.load()) return ctx_fail(c, SPLAT_ERR_HIP, "the communicator was aborted after a rank failed; create it again");
    if (hipSetDevice(ctx_device(c)) != hipSuccess) return ctx_fail(c, SPLAT_ERR_HIP, "hipSetDevice");
    // A single rank has nobody to exchange rows with ...
    if (st->n_ranks == 1 && !st->loopback) return SPLAT_OK;
    // ... unless the loopback hook is on: the rank then moves its own slab rows through RCCL to itself (ncclSend +
    // ncclRecv to the own rank inside one group, which NCCL/RCCL permit), so that the very code path of the
    // multi-rank gather -- group, send, receive, the RCCL kernel on the context's stream -- runs on a one-GPU box.
    // (behind the most recent frame, on the lane its compositor runs on: the next frame to ANOTHER image composites beside
    // this gather instead of behind it -- splat_set_frame_overlap)
    int rc = (st->n_ranks == 1) ? loopback_rows(c, st, (uint32_t*)d_argb, w, h, frame_stream(c))
                                : gather_rows(c, st, (uint32_t*)d_argb, w, h, root, frame_stream(c));
    // "this frame has ended" now includes its gather: a later frame to the image on the other lane, a download / upload
    // on the context's stream and the reuse of the frame's slot all wait for the rows to have left / landed
    const int rt = frame_tail(c);
    return rc != SPLAT_OK ? rc : rt;
}

int splat_comm_loopback(splat_ctx* c, int32_t on) {
    if (!c) return SPLAT_ERR_INVALID;
    CommState* st = *ctx_comm_slot(c);
    if (!st) return ctx_fail(c, SPLAT_ERR_INVALID, "no communicator on this context");
    if (st->n_ranks != 1) return ctx_fail(c, SPLAT_ERR_INVALID, "the loopback hook is for single-rank communicators");
    st->loopback = on != 0;
    return SPLAT_OK;
}

void splat_comm_destroy(splat_ctx* c) {
    if (!c) return;
    CommState** slot = ctx_comm_slot(c);
    if (*slot) {
        (void)hipSetDevice(ctx_device(c));
        (void)hipStreamSynchronize(ctx_stream(c));
        if (frame_stream(c) != ctx_stream(c)) (void)hipStreamSynchronize(frame_stream(c));     // (a gather on the second compositor lane)
        comm_release(*slot);
        *slot = nullptr;
    }
}

}  // extern "C"

// ------------------------------------------------------------------------------------------ (B) one process
// One worker thread per rank, each bound to its device for life (HIP's current device is per thread) and fed
// from a FIFO of commands: the caller's thread only posts, so consecutive frames queue up on every device
// (cross-frame overlap inside each context stays what it is on one GPU) and nobody shares a context.
namespace {
enum CmdKind { CMD_UPLOAD, CMD_FRAME, CMD_RENDER_HOST, CMD_SYNC, CMD_LOADS, CMD_SET_SLABS, CMD_ALLOC, CMD_OVERLAP, CMD_STOP };
struct Cmd {
    CmdKind kind = CMD_SYNC;
    splat_camera cam{};
    // upload
    uint64_t n = 0;
    const float *pos4 = nullptr, *cov3d = nullptr, *opacity = nullptr, *sh = nullptr;
    // host render
    uint32_t* host = nullptr;
    bool want_stats = false;
    // loads
    uint64_t* row_pairs = nullptr;
    int32_t n_rows = 0;
};

struct Worker {
    int rank = 0, device = 0;
    splat_ctx* ctx = nullptr;
    // this rank's w x h image (device).  With frame overlap 2 (splat_multi_set_frame_overlap) there are two, used in turn
    // by the frames of splat_multi_render_frame -- every rank counts the same frames, so frame N is in image N & 1
    // everywhere -- and the compositors of consecutive frames share the chip (splat_set_frame_overlap, splat_api.hip)
    uint32_t* imgs[2] = {nullptr, nullptr};
    size_t img_px = 0;
    uint64_t n_frames = 0;            // frames of splat_multi_render_frame so far
    std::atomic<int> cur{0};          // the image that holds the most recent frame
    hipEvent_t ev_rows = nullptr;     // peer transport: this rank's rows have landed in the root's image
    hipEvent_t ev_root = nullptr;     // peer transport: the root has finished with the previous frame's image (unused so far)
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Cmd> q;
    uint64_t posted = 0, done = 0;
    int rc = SPLAT_OK;                // first error since the last sync
    std::string err;
    splat_stats stats{};
};
}  // namespace

struct splat_multi {
    splat_config cfg{};
    std::vector<std::unique_ptr<Worker>> w;
    std::vector<int32_t> slabs;
    bool peer = false;                // rows travel as device copies + events instead of RCCL
    int overlap = 1;                  // splat_multi_set_frame_overlap
    std::mutex abort_mu;
    int img_w = 0, img_h = 0;
    std::string err;
    uint64_t n_scene = 0;
};

namespace {

int mfail(splat_multi* m, int code, const std::string& msg) {
    if (m) m->err = msg; else g_multi_create_error = msg;
    return code;
}

// A rank that cannot take part in a gather (no image to send from / receive into) must not leave its peers blocked
// in the grouped ncclSend / ncclRecv they have already enqueued: every communicator of the group is aborted (this
// process owns them all), the kernels waiting on them end, and later gathers fail with a message instead of hanging.
void abort_all_comms(splat_multi* m);

void worker_main(splat_multi* m, Worker* me) {
    (void)hipSetDevice(me->device);
    auto note = [&](int rc, const std::string& msg) {
        std::lock_guard<std::mutex> g(me->mu);
        if (me->rc == SPLAT_OK) { me->rc = rc; me->err = "rank " + std::to_string(me->rank) + ": " + msg; }
    };
    auto check = [&](int rc, const char* what) {
        if (rc != SPLAT_OK) note(rc, std::string(what) + ": " + splat_last_error(me->ctx));
        return rc == SPLAT_OK;
    };
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e != hipSuccess) note(SPLAT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
        return e == hipSuccess;
    };
    auto ensure_image = [&](int w, int h) -> bool {
        const size_t px = (size_t)w * h;
        const int want = m->overlap >= 2 ? 2 : 1;
        if (px <= me->img_px && me->imgs[want - 1] != nullptr) return true;
        (void)splat_sync(me->ctx);
        const size_t npx = std::max(px, me->img_px);
        for (int k = 0; k < 2; ++k) { if (me->imgs[k]) (void)hipFree(me->imgs[k]); me->imgs[k] = nullptr; }
        me->img_px = 0;
        for (int k = 0; k < want; ++k) {
            hipError_t e = hipMalloc(&me->imgs[k], npx * 4);
            if (e != hipSuccess) { note(SPLAT_ERR_HIP, std::string("hipMalloc(image): ") + hipGetErrorString(e)); return false; }
        }
        me->img_px = npx;
        return true;
    };
    // this rank's rows -> the root's image (stream order)
    auto gather = [&](int w, int h, int k, bool rendered) {
        Worker* root = m->w[0].get();
        hipStream_t st = frame_stream(me->ctx);              // behind the frame just enqueued, on its compositor's lane
        uint32_t* const mine = me->imgs[k];
        uint32_t* const roots = root->imgs[k];
        if (!m->peer) { (void)check(splat_comm_gather(me->ctx, mine, w, h, 0), "splat_comm_gather"); return; }
        if (me->rank == 0) return;
        int a, b;
        slab_px(&m->slabs[2 * me->rank], h, &a, &b);
        // (a render that failed ships nothing: the copy has no peer waiting for it -- unlike the grouped send / recv above
        // -- and its rows are stale or half cleared; the root keeps what it had, the sync reports the error)
        if (b > a && rendered) {
            const size_t off = (size_t)a * w, bytes = (size_t)(b - a) * w * 4;
            hipError_t e = (root->device == me->device)
                               ? hipMemcpyAsync(roots + off, mine + off, bytes, hipMemcpyDeviceToDevice, st)
                               : hipMemcpyPeerAsync(roots + off, root->device, mine + off, me->device, bytes, st);
            if (e != hipSuccess) note(SPLAT_ERR_HIP, std::string("row copy to the root: ") + hipGetErrorString(e));
        }
        (void)hipEventRecord(me->ev_rows, st);
        (void)frame_tail(me->ctx);
    };
    // One command.  Every path through it ends in the bookkeeping below (done++): nothing in here may leave the
    // thread -- drain(), sync_all_ranks and splat_multi_destroy wait for done == posted.
    auto run = [&](const Cmd& c) {
        hipStream_t st = me->ctx ? (hipStream_t)splat_stream(me->ctx) : nullptr;
        switch (c.kind) {
            case CMD_STOP: break;
            case CMD_UPLOAD:
                (void)check(splat_upload_scene(me->ctx, c.n, c.pos4, c.cov3d, c.opacity, c.sh), "splat_upload_scene");
                break;
            case CMD_SET_SLABS:
                if (m->peer) (void)check(splat_set_slab(me->ctx, m->slabs[2 * me->rank], m->slabs[2 * me->rank + 1]), "splat_set_slab");
                else (void)check(splat_comm_set_slabs(me->ctx, m->slabs.data()), "splat_comm_set_slabs");
                break;
            case CMD_ALLOC:      // images exist on every rank before any rank's gather addresses the root's
                (void)ensure_image((int)c.cam.w, (int)c.cam.h);
                break;
            case CMD_OVERLAP:
                (void)check(splat_set_frame_overlap(me->ctx, m->overlap), "splat_set_frame_overlap");
                break;
            case CMD_LOADS:
                (void)check(splat_tile_row_loads(me->ctx, &c.cam, c.row_pairs, c.n_rows), "splat_tile_row_loads");
                break;
            case CMD_FRAME: {
                const int w = (int)c.cam.w, h = (int)c.cam.h;
                if (!ensure_image(w, h)) { abort_all_comms(m); break; }      // (CMD_ALLOC made the images: not expected)
                // color.clear(0) + render_to_buffer of src/main.rs:73-74, on this rank's rows only (the clear is fused
                // into the compositor).  A render that failed still takes part in the gather -- the peers have
                // enqueued their halves of it -- with whatever its rows hold; the error is reported by the sync.
                const int k = m->overlap >= 2 ? (int)(me->n_frames & 1ull) : 0;
                me->n_frames++;
                const bool ok = check(splat_render_frame_device(me->ctx, &c.cam, me->imgs[k], 0, nullptr), "splat_render_frame_device");
                gather(w, h, k, ok);
                me->cur.store(k);
                break;
            }
            case CMD_RENDER_HOST: {
                const int w = (int)c.cam.w, h = (int)c.cam.h;
                if (!ensure_image(w, h)) { abort_all_comms(m); break; }
                int a, b;
                slab_px(&m->slabs[2 * me->rank], h, &a, &b);
                const size_t off = (size_t)a * w, bytes = (size_t)std::max(0, b - a) * w * 4;
                // the slab's rows of the caller's in/out image; the render is synchronous so that a frame that
                // outgrew its storage is redone here (splat_render_device retries its own frame)
                // (the synchronous in/out frame always uses image 0: every rank agrees without counting)
                // (frames in flight on the other lane: the copy below is on the context's stream; a loss found while
                // waiting stays pending for the caller's next splat_multi_sync, as before)
                if (m->overlap >= 2) (void)check(ctx_quiesce(me->ctx), "waiting for the frames in flight");
                if (bytes) (void)hip_ok(hipMemcpyAsync(me->imgs[0] + off, c.host + off, bytes, hipMemcpyHostToDevice, st), "hipMemcpyAsync(slab rows in)");
                std::memset(&me->stats, 0, sizeof me->stats);
                const bool ok = check(splat_render_device(me->ctx, &c.cam, me->imgs[0], 1, c.want_stats ? &me->stats : nullptr), "splat_render_device");
                gather(w, h, 0, ok);     // (also after a failure: see CMD_FRAME)
                me->cur.store(0);
                break;
            }
            case CMD_SYNC: {
                int rc = splat_sync(me->ctx);
                if (rc != SPLAT_OK) note(rc, std::string("splat_sync: ") + splat_last_error(me->ctx));
                if (m->peer && me->rank != 0 && me->ev_rows) (void)hipEventSynchronize(me->ev_rows);
                break;
            }
        }
    };
    for (;;) {
        Cmd c;
        {
            std::unique_lock<std::mutex> lk(me->mu);
            me->cv.wait(lk, [&] { return !me->q.empty(); });
            c = me->q.front();
            me->q.pop_front();
        }
        run(c);
        {
            std::lock_guard<std::mutex> g(me->mu);
            me->done++;
        }
        me->cv.notify_all();
        if (c.kind == CMD_STOP) return;
    }
}

void abort_all_comms(splat_multi* m) {
    if (m->peer) return;
    std::lock_guard<std::mutex> g(m->abort_mu);
    for (auto& w : m->w) {
        CommState* st = *ctx_comm_slot(w->ctx);
        if (!st || !st->comm || st->aborted.exchange(true)) continue;      // (from here on no gather STARTS on this communicator)
        if (!rccl()->CommAbort) continue;
        // A gather that is past its `aborted` check holds st->mu until its enqueue has returned.  If none is in flight the
        // communicator is aborted under the lock: no thread can be between the check and the use.  If one is, it may be
        // blocked INSIDE RCCL waiting for this very rank (connection set-up of a grouped send / recv) -- waiting for its
        // lock would deadlock -- and aborting a communicator under a call in progress on another thread is what
        // ncclCommAbort is for.
        std::unique_lock<std::mutex> lk(st->mu, std::try_to_lock);
        (void)rccl()->CommAbort(st->comm);
    }
}

void post(Worker* w, const Cmd& c) {
    {
        std::lock_guard<std::mutex> g(w->mu);
        w->q.push_back(c);
        w->posted++;
    }
    w->cv.notify_all();
}
void post_all(splat_multi* m, const Cmd& c) { for (auto& w : m->w) post(w.get(), c); }

// wait until every worker has drained its queue; collects the first error
int drain(splat_multi* m) {
    int rc = SPLAT_OK;
    for (auto& w : m->w) {
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [&] { return w->done == w->posted; });
        if (w->rc != SPLAT_OK && rc == SPLAT_OK) { rc = w->rc; m->err = w->err; }
        w->rc = SPLAT_OK;
    }
    return rc;
}

int sync_all_ranks(splat_multi* m) {
    Cmd c; c.kind = CMD_SYNC;
    post_all(m, c);
    return drain(m);
}

}  // namespace

extern "C" {

const char* splat_multi_last_error(const splat_multi* m) { return m ? m->err.c_str() : g_multi_create_error.c_str(); }

int splat_multi_create(const splat_config* cfg, const int32_t* devices, int32_t n_devices, splat_multi** out) {
    if (!out) return mfail(nullptr, SPLAT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) return mfail(nullptr, SPLAT_ERR_INVALID, "need 1..64 devices");
    std::unique_ptr<splat_multi> m(new splat_multi());
    if (cfg) m->cfg = *cfg; else splat_default_config(&m->cfg);
    bool dup = false;
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j) dup = dup || devices[i] == devices[j];
    const char* tr = std::getenv("SPLAT_MULTI_TRANSPORT");
    m->peer = dup || (tr && std::strcmp(tr, "peer") == 0);
    // contexts first (on the caller's thread: creation is not the hot path), then the communicators, then the threads
    for (int i = 0; i < n_devices; ++i) {
        std::unique_ptr<Worker> w(new Worker());
        w->rank = i; w->device = devices[i];
        splat_config c = m->cfg;
        c.device = devices[i];
        if (splat_create(&c, &w->ctx) != SPLAT_OK) {
            std::string msg = std::string("splat_create(device ") + std::to_string(devices[i]) + "): " + splat_last_error(nullptr);
            for (auto& x : m->w) splat_destroy(x->ctx);
            return mfail(nullptr, SPLAT_ERR_HIP, msg);
        }
        (void)hipSetDevice(devices[i]);
        (void)hipEventCreateWithFlags(&w->ev_rows, hipEventDisableTiming);
        m->w.push_back(std::move(w));
    }
    auto bail = [&](int code, const std::string& msg) {
        for (auto& x : m->w) { if (x->ev_rows) { (void)hipSetDevice(x->device); (void)hipEventDestroy(x->ev_rows); } splat_destroy(x->ctx); }
        return mfail(nullptr, code, msg);
    };
    if (!m->peer) {
        Rccl* R = rccl();
        if (!R->error.empty()) return bail(SPLAT_ERR_HIP, R->error);
        std::vector<ncclComm_t> comms(n_devices);
        ncclResult_t e = R->CommInitAll(comms.data(), n_devices, devices);
        if (e != ncclSuccess) return bail(SPLAT_ERR_HIP, std::string("ncclCommInitAll: ") + R->GetErrorString(e));
        for (int i = 0; i < n_devices; ++i) {
            CommState* st = new CommState();
            st->comm = comms[i]; st->n_ranks = n_devices; st->rank = i;
            *ctx_comm_slot(m->w[i]->ctx) = st;
        }
    } else if (!dup) {
        // distinct devices, copy transport: the peers write into the root's memory
        for (int i = 1; i < n_devices; ++i) {
            (void)hipSetDevice(devices[i]);
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, devices[i], devices[0]);
            if (can) (void)hipDeviceEnablePeerAccess(devices[0], 0);
        }
    }
    m->slabs.resize(2 * (size_t)n_devices);
    splat_multi* mp = m.get();
    for (auto& w : m->w) w->th = std::thread(worker_main, mp, w.get());
    *out = m.release();
    return SPLAT_OK;        // (the partition is made at the first frame / splat_multi_balance: it needs the target height)
}

void splat_multi_destroy(splat_multi* m) {
    if (!m) return;
    (void)sync_all_ranks(m);
    Cmd c; c.kind = CMD_STOP;
    post_all(m, c);
    for (auto& w : m->w) if (w->th.joinable()) w->th.join();
    for (auto& w : m->w) {
        (void)hipSetDevice(w->device);
        splat_destroy(w->ctx);        // releases the communicator too
        for (int k = 0; k < 2; ++k) if (w->imgs[k]) (void)hipFree(w->imgs[k]);
        if (w->ev_rows) (void)hipEventDestroy(w->ev_rows);
    }
    delete m;
}

int splat_multi_upload_scene(splat_multi* m, uint64_t n, const float* pos4, const float* cov3d, const float* opacity, const float* sh) {
    if (!m) return SPLAT_ERR_INVALID;
    int rc = drain(m);
    if (rc != SPLAT_OK) return rc;
    Cmd c; c.kind = CMD_UPLOAD; c.n = n; c.pos4 = pos4; c.cov3d = cov3d; c.opacity = opacity; c.sh = sh;
    post_all(m, c);                   // every rank uploads its replica at the same time
    rc = drain(m);
    if (rc == SPLAT_OK) m->n_scene = n;
    return rc;
}

int splat_multi_balance(splat_multi* m, const splat_camera* cam) {
    if (!m) return SPLAT_ERR_INVALID;
    int rc = sync_all_ranks(m);
    if (rc != SPLAT_OK) return rc;
    const int k = (int)m->w.size();
    // cam == NULL: equal slabs for the target size of the last partition (a camera is needed once: its height fixes
    // the number of tile rows)
    splat_camera eq{};
    const bool equal = cam == nullptr;
    if (equal) {
        if (m->img_h < 1) return mfail(m, SPLAT_ERR_INVALID, "camera is NULL and no frame size is known yet (its height fixes the tile rows)");
        eq.w = (float)m->img_w; eq.h = (float)m->img_h;
        cam = &eq;
    }
    const int n_rows = ((int)cam->h + TILE - 1) / TILE;
    if (n_rows < 1) return mfail(m, SPLAT_ERR_INVALID, "bad camera height");
    std::vector<uint64_t> loads((size_t)n_rows, 0);
    bool have_loads = false;
    if (m->n_scene && k > 1 && !equal) {
        Cmd c; c.kind = CMD_LOADS; c.cam = *cam; c.row_pairs = loads.data(); c.n_rows = n_rows;
        post(m->w[0].get(), c);
        rc = drain(m);
        if (rc != SPLAT_OK) return rc;
        have_loads = true;
    }
    // row_overhead: a tile row costs something even when it is empty (scan, launch of its tiles); 2000 pairs' worth -- and
    // 100 000 when the compositors of consecutive frames share the chip (frame overlap 2): a slab's frame then costs
    // its work rather than its densest tile's latency (tools/slab_try.py auto:8:<overhead>)
    rc = splat_slab_partition(have_loads ? loads.data() : nullptr, n_rows, k, m->overlap >= 2 ? 100000.0 : 2000.0, m->slabs.data());
    if (rc != SPLAT_OK) return mfail(m, rc, "splat_slab_partition");
    m->img_w = (int)cam->w; m->img_h = (int)cam->h;
    Cmd c; c.kind = CMD_SET_SLABS;
    post_all(m, c);
    c.kind = CMD_ALLOC; c.cam = *cam;
    post_all(m, c);
    rc = drain(m);
    if (rc != SPLAT_OK) { m->img_w = 0; m->img_h = 0; }      // (the next frame partitions -- and allocates -- again)
    return rc;
}

int splat_multi_get_slabs(const splat_multi* m, int32_t* out) {
    if (!m || !out) return SPLAT_ERR_INVALID;
    std::copy(m->slabs.begin(), m->slabs.end(), out);
    return SPLAT_OK;
}

static int need_partition(splat_multi* m, const splat_camera* cam) {
    if (!cam) return mfail(m, SPLAT_ERR_INVALID, "camera is NULL");
    if (m->img_w == (int)cam->w && m->img_h == (int)cam->h) return SPLAT_OK;
    return splat_multi_balance(m, cam);          // first frame at this target size: partition for it
}

int splat_multi_render_frame(splat_multi* m, const splat_camera* cam) {
    if (!m) return SPLAT_ERR_INVALID;
    int rc = need_partition(m, cam);
    if (rc != SPLAT_OK) return rc;
    Cmd c; c.kind = CMD_FRAME; c.cam = *cam;
    post_all(m, c);
    return SPLAT_OK;
}

int splat_multi_sync(splat_multi* m) { return m ? sync_all_ranks(m) : SPLAT_ERR_INVALID; }

void* splat_multi_image(splat_multi* m) { return (m && !m->w.empty()) ? m->w[0]->imgs[m->w[0]->cur.load()] : nullptr; }

int splat_multi_set_frame_overlap(splat_multi* m, int32_t n) {
    if (!m) return SPLAT_ERR_INVALID;
    if (n < 1 || n > 2) return mfail(m, SPLAT_ERR_INVALID, "frame overlap is 1 or 2");
    int rc = sync_all_ranks(m);
    if (rc != SPLAT_OK) return rc;
    if (n == m->overlap) return SPLAT_OK;
    m->overlap = n;
    Cmd c; c.kind = CMD_OVERLAP;
    post_all(m, c);
    rc = drain(m);
    if (rc != SPLAT_OK) return rc;
    m->img_w = 0; m->img_h = 0;          // the next frame partitions again (the balance weighs rows differently) and allocates the second image
    return SPLAT_OK;
}

splat_ctx* splat_multi_ctx(splat_multi* m, int32_t rank) {
    return (m && rank >= 0 && rank < (int)m->w.size()) ? m->w[rank]->ctx : nullptr;
}

int splat_multi_download(splat_multi* m, uint32_t* out, int32_t w, int32_t h) {
    if (!m || !out) return SPLAT_ERR_INVALID;
    int rc = sync_all_ranks(m);
    if (rc != SPLAT_OK) return rc;
    Worker* root = m->w[0].get();
    uint32_t* const img = root->imgs[root->cur.load()];
    if (!img || (size_t)w * h > root->img_px) return mfail(m, SPLAT_ERR_INVALID, "no frame of that size has been rendered");
    (void)hipSetDevice(root->device);
    hipError_t e = hipMemcpy(out, img, (size_t)w * h * 4, hipMemcpyDeviceToHost);
    return e == hipSuccess ? SPLAT_OK : mfail(m, SPLAT_ERR_HIP, std::string("hipMemcpy: ") + hipGetErrorString(e));
}

int splat_multi_render(splat_multi* m, const splat_camera* cam, uint32_t* argb, splat_stats* stats) {
    if (!m) return SPLAT_ERR_INVALID;
    if (!argb) return mfail(m, SPLAT_ERR_INVALID, "argb is NULL");
    int rc = need_partition(m, cam);
    if (rc != SPLAT_OK) return rc;
    Cmd c; c.kind = CMD_RENDER_HOST; c.cam = *cam; c.host = argb; c.want_stats = stats != nullptr;
    post_all(m, c);
    rc = sync_all_ranks(m);
    if (rc != SPLAT_OK) return rc;
    const int w = (int)cam->w, h = (int)cam->h;
    rc = splat_multi_download(m, argb, w, h);
    if (rc != SPLAT_OK) return rc;
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        for (auto& wk : m->w) {
            const splat_stats& s = wk->stats;
            stats->n_gaussians = s.n_gaussians;
            stats->n_visible += s.n_visible; stats->n_singular = std::max(stats->n_singular, s.n_singular);
            stats->n_pairs += s.n_pairs; stats->max_tile_len = std::max(stats->max_tile_len, s.max_tile_len);
            stats->bytes_algorithmic += s.bytes_algorithmic; stats->flops_algorithmic += s.flops_algorithmic;
            stats->n_fallback += s.n_fallback; stats->n_sort_fallback += s.n_sort_fallback;
            stats->n_iter_scan += s.n_iter_scan; stats->n_iter_blend += s.n_iter_blend;
            stats->n_blocks_culled += s.n_blocks_culled;
            stats->n_near_tiles += s.n_near_tiles; stats->n_near_fallback += s.n_near_fallback;
            stats->ms_preprocess = std::max(stats->ms_preprocess, s.ms_preprocess); stats->ms_scan = std::max(stats->ms_scan, s.ms_scan);
            stats->ms_emit = std::max(stats->ms_emit, s.ms_emit); stats->ms_sort = std::max(stats->ms_sort, s.ms_sort);
            stats->ms_composite = std::max(stats->ms_composite, s.ms_composite); stats->ms_total = std::max(stats->ms_total, s.ms_total);
        }
    }
    return SPLAT_OK;
}

}  // extern "C"
