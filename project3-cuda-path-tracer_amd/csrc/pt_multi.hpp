// pt_multi.hpp -- the exported C-ABI of libptmi355.so (include/ptmi355.h) and, behind it, the frame tiled over
// several GPUs of one node INSIDE the library (included by ptmi355.hip only).
//
// The reference renders on one device chosen by the host (cudaGLSetGLDevice(0), preview.cpp:107) and keeps its
// renderer state in file-static globals (pathtrace.cu:70-75).  Here a session owns one CONTEXT per device
// (pt_h_session.hpp: Renderer); with one device the exported functions run the context on the caller's thread, exactly
// as before.  With several (pt_scene_desc::devices / num_devices, or PTMI355_DEVICES in the environment -- so the
// reference's host needs no change at all):
//
//   * context k renders tile k of K: the rows y with (y / strip_rows) % K == k (interleaved strips: border rows die at
//     bounce 0, centre rows live longest), with the GLOBAL pixelIndex as RNG key -- every radiance value is the one
//     the single-device run computes;
//   * every context has its own host thread (HIP's current device is per-thread state; the launches of the K devices
//     are issued concurrently, not one device after the other), its own launch stream and an exchange stream;
//   * after every pt_trace / batch the tiles' running sums travel to device 0: each context packs its rows
//     (k_pack_tile, tile_pixels * 12 B), the packed tiles move over the root's xGMI links -- RCCL: one
//     ncclGroupStart ... ncclRecv x (K-1) on the root / ncclSend on every peer ... ncclGroupEnd on a communicator
//     from ncclCommInitAll (one process, no launcher); or, where RCCL cannot be used (two contexts on ONE device:
//     tests on a 1-GPU box; PTMI355_XCHG=peer), hipMemcpyPeerAsync -- and the root unpacks them into the frame, which
//     is context 0's accumulation buffer (k_unpack_tile).  Copies only: the assembled frame is bit-identical to the
//     single-device image.  This replaces finalGather + the per-iteration device-to-host copy of
//     pathtrace.cu:380-390 as the point where the image comes together;
//   * two staging slots per context: the exchange of call i runs on the exchange streams while call i+1 traces
//     (pt_trace_batch_async); pt_trace / pt_trace_batch return with state.image filled from the assembled frame.
//
// librccl is opened with dlopen on first use (single-device sessions never load it; in a process that already
// holds a copy -- PyTorch's -- the same copy is reused through its SONAME).
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>      // types and prototypes only: nothing links against librccl

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

namespace {

// packed[3 j + c] = image[3 pixel(j) + c] for the tile's local pixels j (rows of W * 3 contiguous floats)
__global__ __launch_bounds__(BLOCK) void k_pack_tile(const float *__restrict__ image, TileMap map, float *__restrict__ packed) {
    const uint32_t e = blockIdx.x * BLOCK + threadIdx.x;
    if (e >= (uint32_t)map.tile_pixels * 3u) return;
    const uint32_t j = e / 3u, c = e - 3u * j;
    packed[e] = image[(size_t)local_to_pixel(map, (int)j) * 3 + c];
}
__global__ __launch_bounds__(BLOCK) void k_unpack_tile(float *__restrict__ frame, TileMap map, const float *__restrict__ packed) {
    const uint32_t e = blockIdx.x * BLOCK + threadIdx.x;
    if (e >= (uint32_t)map.tile_pixels * 3u) return;
    const uint32_t j = e / 3u, c = e - 3u * j;
    frame[(size_t)local_to_pixel(map, (int)j) * 3 + c] = packed[e];
}

// ---- librccl, bound at run time ------------------------------------------------------------------------------
struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string why;
    bool load() {
        if (handle) return true;
        const char *names[] = {getenv("PTMI355_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
            why = dlerror();
        }
        if (!handle) return false;
        bool ok = true;
        auto sym = [&](const char *name) { void *p = dlsym(handle, name); if (!p) { ok = false; why = std::string("missing symbol ") + name; } return p; };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        GetVersion = (decltype(GetVersion))sym("ncclGetVersion");
        if (!ok) { dlclose(handle); handle = nullptr; }
        return ok;
    }
} g_rccl;

#define NCCLCHK(expr)                                                                              \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess)                                                                     \
            return fail(PT_ERR_DEVICE, "RCCL error (%s:%d): %s: %s", "pt_multi.hpp", __LINE__, #expr, \
                        g_rccl.GetErrorString(r_));                                                \
    } while (0)

// staging slots per context: call i uses slot i % XSLOTS, so the tracing runs up to XSLOTS exchanges ahead of the slowest
// one (round 3 had two: at one exchange per iteration the LATENCY of an exchange -- pack, RCCL's own bookkeeping copies
// and clears, its kernel, the unpack, ~0.3 ms end to end -- then paced the iterations, not its throughput)
constexpr int XSLOTS = 4;

// ---- one host thread per device context ------------------------------------------------------------------------
struct Worker {
    Renderer ctx;
    char err[ERR_BYTES] = "";
    int index = 0, device = 0;
    std::thread th;
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    std::deque<std::function<int()>> jobs;      // FIFO: an asynchronous batch is posted and not waited for (multi_enqueue)
    std::atomic<uint64_t> posted{0}, finished{0};
    bool quit = false;
    int rc = PT_OK;                              // of the last job run
    std::atomic<int> sticky{PT_OK};              // first failure of a job (its message stays in err): reported by the next wait
    // exchange state of this context (device `device`)
    hipStream_t xs = nullptr;                   // exchange stream
    float *pack[XSLOTS] = {};                    // packed tile rows, one per slot
    hipEvent_t ev_packed[XSLOTS] = {};           // slot s packed (launch stream)
    hipEvent_t ev_sent[XSLOTS] = {};             // slot s has left this device (exchange stream)
    bool sent_once[XSLOTS] = {};
    float *stage[XSLOTS] = {};                   // ON THE ROOT DEVICE: where this context's rows land
    size_t floats = 0;                           // tile_pixels * 3
    TileMap map{};
    float *host_dev = nullptr; const float *host_dev_of = nullptr;   // this device's address of Group::dhost

    void loop() {
        t_ctx = &ctx;
        t_err = err;
        (void)hipSetDevice(device);
        uint64_t seen = 0;
        for (;;) {
            // a batch follows a batch within microseconds: look for the next job for a moment before sleeping
            for (int spin = 0; spin < 4000 && posted.load(std::memory_order_acquire) == seen; ++spin) __builtin_ia32_pause();
            std::function<int()> f;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_job.wait(lk, [&] { return !jobs.empty() || quit; });
                if (jobs.empty()) return;                                   // quit, nothing left to run
                f = std::move(jobs.front());
                jobs.pop_front();
            }
            const int r = f();
            rc = r;
            if (r < 0) { int none = PT_OK; sticky.compare_exchange_strong(none, r); }
            finished.store(++seen, std::memory_order_release);
            { std::lock_guard<std::mutex> lk(m); }
            cv_done.notify_all();
        }
    }
    uint64_t post(std::function<int()> f) {
        uint64_t seq;
        { std::lock_guard<std::mutex> lk(m); jobs.push_back(std::move(f)); seq = posted.fetch_add(1, std::memory_order_release) + 1; }
        cv_job.notify_one();
        return seq;
    }
    // job number `seq` (and every job before it) has run; a failure of any job so far is returned, and forgotten if `clear`
    int wait_seq(uint64_t seq, bool clear) {
        for (int spin = 0; spin < 4000 && finished.load(std::memory_order_acquire) < seq; ++spin) __builtin_ia32_pause();
        if (finished.load(std::memory_order_acquire) < seq) {
            std::unique_lock<std::mutex> lk(m);
            cv_done.wait(lk, [&] { return finished.load(std::memory_order_acquire) >= seq; });
        }
        const int e = clear ? sticky.exchange(PT_OK) : sticky.load();
        return e;
    }
    int wait() {                                   // everything posted has run: the last job's result, or an earlier failure
        const int e = wait_seq(posted.load(std::memory_order_acquire), true);
        return e < 0 ? e : rc;
    }
};

// The exchange of a call is a dozen host calls -- a grouped RCCL send / recv costs the issuing thread ~80 us -- and
// nothing in it needs the caller.  With asynchronous batches (pt_trace_batch_async) it is issued from this thread while
// the caller already enqueues the next batch's launches: at one exchange per iteration the host, not the device, set
// the pace.  Jobs run in order; the caller posts up to XSLOTS - 1 exchanges ahead (a call's packing waits on events the
// exchange XSLOTS calls earlier records) and waits for all of them before anything else touches the exchange streams.
struct Exchanger {
    char err[ERR_BYTES] = "";
    std::thread th;
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    std::deque<std::function<int()>> jobs;      // FIFO: the caller runs up to XSLOTS - 1 exchanges ahead (multi_enqueue)
    std::atomic<uint64_t> posted{0}, finished{0};
    bool quit = false;
    std::atomic<int> sticky{PT_OK};              // first failure (its message stays in err): reported by the next wait
    void loop() {
        t_err = err;
        uint64_t seen = 0;
        for (;;) {
            for (int spin = 0; spin < 8000 && posted.load(std::memory_order_acquire) == seen; ++spin) __builtin_ia32_pause();
            std::function<int()> f;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_job.wait(lk, [&] { return !jobs.empty() || quit; });
                if (jobs.empty()) return;
                f = std::move(jobs.front());
                jobs.pop_front();
            }
            const int r = f();
            if (r < 0) { int none = PT_OK; sticky.compare_exchange_strong(none, r); }
            finished.store(++seen, std::memory_order_release);
            { std::lock_guard<std::mutex> lk(m); }
            cv_done.notify_all();
        }
    }
    uint64_t post(std::function<int()> f) {
        uint64_t seq;
        { std::lock_guard<std::mutex> lk(m); jobs.push_back(std::move(f)); seq = posted.fetch_add(1, std::memory_order_release) + 1; }
        cv_job.notify_one();
        return seq;
    }
    // exchange number `seq` (and every one before it) has been ENQUEUED (not executed); a failure so far is returned
    int wait_seq(uint64_t seq, bool clear) {
        for (int spin = 0; spin < 8000 && finished.load(std::memory_order_acquire) < seq; ++spin) __builtin_ia32_pause();
        if (finished.load(std::memory_order_acquire) < seq) {
            std::unique_lock<std::mutex> lk(m);
            cv_done.wait(lk, [&] { return finished.load(std::memory_order_acquire) >= seq; });
        }
        return clear ? sticky.exchange(PT_OK) : sticky.load();
    }
    int wait() { return wait_seq(posted.load(std::memory_order_acquire), true); }
};

struct Group {
    bool live = false;
    int K = 0;
    std::vector<std::unique_ptr<Worker>> w;
    int W = 0, H = 0, npix = 0;
    bool use_rccl = false;
    bool self_exchange = false;                 // K == 1 over RCCL (rehearsal: every RCCL call with a communicator of one)
    std::vector<ncclComm_t> comms;
    hipEvent_t ev_frame[XSLOTS] = {};           // slot s unpacked into the frame (root's exchange stream)
    bool frame_once[XSLOTS] = {};
    uint64_t exchanges = 0;                     // exchanges enqueued (by whichever thread issues them)
    uint64_t calls = 0;                         // calls made (caller's thread): call i uses staging slot i % XSLOTS
    uint64_t xseq[XSLOTS] = {};                 // the exchange-thread job that last used slot s (0: none outstanding)
    // pathtrace() with a host image and no PBO: every context's launch writes its own tile's pixels into the caller's
    // page-locked image (registered ONCE, portable: every device maps it) -- no pack, no exchange, no frame-to-host copy
    bool direct_ok = false, direct_enabled = true;   // every context can (asked at pt_init) / PTMI355_MULTI_DIRECT
    float *dhost = nullptr; size_t dhost_bytes = 0;  // the registered host image
    bool frame_stale = false;  /* (set by the caller's thread, cleared by it or by the exchange thread -- always through __atomic_load_n / __atomic_store_n, relaxed: a late clear only costs one redundant exchange) */              // the device frame lacks the peers' rows of such calls: the next exchange brings them
    std::unique_ptr<Exchanger> x;               // asynchronous batches hand their exchange to this thread
    float *frame = nullptr;                     // where the tiles are assembled: context 0's accumulation buffer -- except in the
                                                // one-context RCCL rehearsal, where it is a buffer of its own (self_frame)
    float *self_frame = nullptr;
    pt_camera last_cam{};
    int last_depth = -1;
    std::string transport;
} G;

// run f(worker) on every context's thread, concurrently; first failure wins (its message goes to the caller's buffer)
int on_all(const std::function<int(Worker &)> &f) {
    for (auto &wp : G.w) { Worker *w = wp.get(); w->post([w, &f] { return f(*w); }); }
    int rc = PT_OK;
    for (auto &wp : G.w) {
        const int r = wp->wait();
        if (r < 0 && rc == PT_OK) { rc = r; memcpy(t_err, wp->err, ERR_BYTES); }
    }
    return rc;
}
int on_one(int k, const std::function<int(Worker &)> &f) {
    Worker *w = G.w[(size_t)k].get();
    w->post([w, &f] { return f(*w); });
    const int r = w->wait();
    if (r < 0) memcpy(t_err, w->err, ERR_BYTES);
    return r;
}

struct DeviceGuard {            // the exchange is issued from the caller's thread: leave its current device as it was
    int saved = -1;
    DeviceGuard() { if (hipGetDevice(&saved) != hipSuccess) saved = -1; }
    ~DeviceGuard() { if (saved >= 0) (void)hipSetDevice(saved); }
};

// worker side: after this call's launches, pack the tile into slot s (once the slot's previous content has left)
int worker_pack(Worker &w, int s) {
    if (w.index == 0 && !G.self_exchange) {                 // the root's rows are already in the frame
        HIPCHK(hipEventRecord(w.ev_packed[s], R.stream));
        return PT_OK;
    }
    if (w.sent_once[s]) HIPCHK(hipStreamWaitEvent(R.stream, w.ev_sent[s], 0));
    hipLaunchKernelGGL(k_pack_tile, dim3((unsigned)((w.floats + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, R.stream, R.image, R.map, w.pack[s]);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(w.ev_packed[s], R.stream));
    return PT_OK;
}

// caller's thread: tiles -> root -> frame, on the exchange streams (nothing here waits on the host)
int enqueue_exchange(int s) {
    Worker &root = *G.w[0];
    if (G.K == 1 && !G.self_exchange) { G.exchanges++; return PT_OK; }     // one context, nothing to move: its buffer is the frame
    __atomic_store_n(&G.frame_stale, false, __ATOMIC_RELAXED);      // every exchange carries the running sums as they are: the peers' rows of direct calls come along
    DeviceGuard guard;
    if (G.use_rccl) {
        for (auto &wp : G.w) {
            HIPCHK(hipSetDevice(wp->device));
            HIPCHK(hipStreamWaitEvent(wp->xs, wp->ev_packed[s], 0));
        }
        // the receives are ordered behind the unpack that last read the slot's staging buffers: same stream
        NCCLCHK(g_rccl.GroupStart());
        for (int k = G.self_exchange ? 0 : 1; k < G.K; ++k) {
            Worker &p = *G.w[(size_t)k];
            NCCLCHK(g_rccl.Recv(p.stage[s], p.floats, ncclFloat, k, G.comms[0], root.xs));
            NCCLCHK(g_rccl.Send(p.pack[s], p.floats, ncclFloat, 0, G.comms[(size_t)k], p.xs));
        }
        NCCLCHK(g_rccl.GroupEnd());
        for (int k = G.self_exchange ? 0 : 1; k < G.K; ++k) {
            Worker &p = *G.w[(size_t)k];
            HIPCHK(hipSetDevice(p.device));
            HIPCHK(hipEventRecord(p.ev_sent[s], p.xs));
            p.sent_once[s] = true;
        }
        HIPCHK(hipSetDevice(root.device));
    } else {
        for (int k = 1; k < G.K; ++k) {
            Worker &p = *G.w[(size_t)k];
            HIPCHK(hipSetDevice(p.device));
            HIPCHK(hipStreamWaitEvent(p.xs, p.ev_packed[s], 0));
            if (G.frame_once[s]) HIPCHK(hipStreamWaitEvent(p.xs, G.ev_frame[s], 0));      // the root has unpacked the slot's last content
            if (p.device == root.device)
                HIPCHK(hipMemcpyAsync(p.stage[s], p.pack[s], p.floats * 4, hipMemcpyDeviceToDevice, p.xs));
            else
                HIPCHK(hipMemcpyPeerAsync(p.stage[s], root.device, p.pack[s], p.device, p.floats * 4, p.xs));
            HIPCHK(hipEventRecord(p.ev_sent[s], p.xs));
            p.sent_once[s] = true;
        }
        HIPCHK(hipSetDevice(root.device));
        HIPCHK(hipStreamWaitEvent(root.xs, root.ev_packed[s], 0));
        for (int k = 1; k < G.K; ++k) HIPCHK(hipStreamWaitEvent(root.xs, G.w[(size_t)k]->ev_sent[s], 0));
    }
    for (int k = G.self_exchange ? 0 : 1; k < G.K; ++k) {
        Worker &p = *G.w[(size_t)k];
        hipLaunchKernelGGL(k_unpack_tile, dim3((unsigned)((p.floats + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, root.xs, G.frame, p.map,
                           p.stage[s]);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(G.ev_frame[s], root.xs));
    G.frame_once[s] = true;
    G.exchanges++;
    return PT_OK;
}

int exchange_settled(void);

// The frame on device 0 after calls that wrote the host image directly (multi_trace): one exchange of the running sums
// as they are brings every peer's rows; every exchange does, so `frame_stale` falls with the next one of any kind.
int multi_refresh(void) {
    if (!__atomic_load_n(&G.frame_stale, __ATOMIC_RELAXED)) return PT_OK;
    __atomic_store_n(&G.frame_stale, false, __ATOMIC_RELAXED);
    if (G.K == 1) return PT_OK;
    const int s = (int)(G.calls % XSLOTS);
    int rc = exchange_settled();
    if (rc) return rc;
    rc = on_all([&](Worker &w) -> int { return worker_pack(w, s); });
    if (rc) return rc;
    G.calls++;
    return enqueue_exchange(s);
}

// the launches of one call on every device, the packing of the tiles, then the exchange
// `overlap`: the caller does not wait for this call (pt_trace_batch_async): consecutive batches may overlap on each
// device's lanes (pt_h_enqueue.hpp: enqueue_batch_direct); their gathers stay on the launch stream, which the packing waits on
// the exchange thread has enqueued everything it was given (an error of its own is the caller's now)
int exchange_settled(void) {
    if (!G.x) return PT_OK;
    const int r = G.x->wait();
    for (int s = 0; s < XSLOTS; ++s) G.xseq[s] = 0;
    if (r < 0) memcpy(t_err, G.x->err, ERR_BYTES);
    return r;
}

// PTMI355_XCHG_STATS=1: host time per asynchronous call, by thread, printed by pt_free (where the pace of one exchange per
// iteration is set when the device is not the limit)
struct XStats {
    std::atomic<uint64_t> calls{0}, main_ns{0}, worker_ns{0}, xwait_ns{0}, xissue_ns{0};
    bool on = false;
} g_xstats;
static inline uint64_t now_ns(void) { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int multi_enqueue(int iter0, int count, bool overlap = false) {
    const int s = (int)(G.calls % XSLOTS);
    if (G.x && overlap) {
        // The caller does not wait for this batch: it does not wait for the contexts' threads to have ENQUEUED it either.
        // Each worker gets the job (launches + packing of slot s) and the exchange thread the exchange, which first waits
        // until every worker has run that job (the events it makes the exchange streams wait for are recorded by then).
        // The caller runs up to XSLOTS - 1 calls ahead of both (a slot is reused when its last exchange has been issued); a
        // failure of a posted job is reported by a later call or by pt_synchronize.  (With the caller one call ahead
        // every call was a chain of three thread wake-ups -- caller -> worker -> exchange thread -> caller: 0.100 ms per
        // iteration with NOTHING to exchange against 0.081 without the threads, profiles/r04/x_*.json.)
        int rc = PT_OK;
        const uint64_t t_main = g_xstats.on ? now_ns() : 0;
        // the slot's last exchange has been issued: its events are what this call's packing waits for
        if (G.xseq[s]) {
            rc = G.x->wait_seq(G.xseq[s], false);
            if (rc < 0) { (void)G.x->wait(); memcpy(t_err, G.x->err, ERR_BYTES); return rc; }
        }
        std::vector<uint64_t> seq((size_t)G.K);
        for (auto &wp : G.w) {
            Worker *w = wp.get();
            if (w->posted.load(std::memory_order_acquire) - w->finished.load(std::memory_order_acquire) >= (uint64_t)XSLOTS) {
                rc = w->wait_seq(w->posted.load(std::memory_order_acquire) - 1, false);
                if (rc < 0) { (void)w->wait_seq(0, true); memcpy(t_err, w->err, ERR_BYTES); return rc; }
            }
            seq[(size_t)w->index] = w->post([w, iter0, count, s] {
                const uint64_t t0 = g_xstats.on ? now_ns() : 0;
                R.in_step = false;
                R.ov_ok = true;
                int r = enqueue_batch(iter0, count);
                R.ov_ok = false;
                if (!r) r = worker_pack(*w, s);
                if (g_xstats.on && w->index == 0) g_xstats.worker_ns += now_ns() - t0;
                return r;
            });
        }
        G.calls++;
        G.xseq[s] = G.x->post([s, seq] {
            const uint64_t t0 = g_xstats.on ? now_ns() : 0;
            for (auto &wp : G.w) {
                const int r = wp->wait_seq(seq[(size_t)wp->index], false);
                if (r < 0) { memcpy(t_err, wp->err, ERR_BYTES); return r; }
            }
            const uint64_t t1 = g_xstats.on ? now_ns() : 0;
            const int r = enqueue_exchange(s);
            if (g_xstats.on) { g_xstats.xwait_ns += t1 - t0; g_xstats.xissue_ns += now_ns() - t1; }
            return r;
        });
        if (g_xstats.on) { g_xstats.main_ns += now_ns() - t_main; g_xstats.calls++; }
        return PT_OK;
    }
    // the slot protocol: whatever the exchange thread still holds (up to XSLOTS asynchronous calls) is issued before this
    // call's packing re-records the slot's events
    int rc = exchange_settled();
    if (rc) return rc;
    rc = on_all([&](Worker &w) -> int {
        R.in_step = false;
        R.ov_ok = overlap;
        const int r = enqueue_batch(iter0, count);
        R.ov_ok = false;
        if (r) return r;
        return worker_pack(w, s);
    });
    if (rc) return rc;
    G.calls++;
    rc = exchange_settled();
    if (rc) return rc;
    return enqueue_exchange(s);
}

int multi_refresh(void);

int multi_sync(void) {
    int rc = multi_refresh();
    if (rc) return rc;
    rc = exchange_settled();
    if (rc) return rc;
    rc = on_all([&](Worker &w) -> int {
        HIPCHK(hipStreamSynchronize(R.stream));
        HIPCHK(hipStreamSynchronize(w.xs));
        return PT_OK;
    });
    return rc;
}

void multi_free(void) {
    if (!G.live && G.w.empty()) return;
    if (G.dhost) { (void)hipHostUnregister(G.dhost); G.dhost = nullptr; G.dhost_bytes = 0; }
    if (g_xstats.on && g_xstats.calls.load()) {
        const double n = (double)g_xstats.calls.load();
        fprintf(stderr, "[ptmi355] %llu asynchronous calls: caller %.1f us, worker 0 %.1f us, exchange thread waits %.1f us + issues %.1f us per call (%s)\n",
                (unsigned long long)g_xstats.calls.load(), g_xstats.main_ns.load() / n / 1e3, g_xstats.worker_ns.load() / n / 1e3,
                g_xstats.xwait_ns.load() / n / 1e3, g_xstats.xissue_ns.load() / n / 1e3, G.transport.c_str());
        g_xstats.calls = 0; g_xstats.main_ns = 0; g_xstats.worker_ns = 0; g_xstats.xwait_ns = 0; g_xstats.xissue_ns = 0;
    }
    if (G.x) {
        (void)G.x->wait();
        { std::lock_guard<std::mutex> lk(G.x->m); G.x->quit = true; }
        G.x->cv_job.notify_one();
        if (G.x->th.joinable()) G.x->th.join();
        G.x.reset();
    }
    for (auto &wp : G.w) {
        Worker *w = wp.get();
        if (!w->th.joinable()) continue;
        w->post([w] {
            if (R.stream) (void)hipStreamSynchronize(R.stream);
            if (w->xs) (void)hipStreamSynchronize(w->xs);
            return PT_OK;
        });
        (void)w->wait();
    }
    if (!G.comms.empty() && g_rccl.handle)
        for (ncclComm_t c : G.comms) if (c) (void)g_rccl.CommDestroy(c);
    G.comms.clear();
    {
        DeviceGuard guard;
        if (!G.w.empty()) {
            (void)hipSetDevice(G.w[0]->device);
            for (auto &wp : G.w)
                for (int s = 0; s < XSLOTS; ++s) if (wp->stage[s]) (void)hipFree(wp->stage[s]);
            for (int s = 0; s < XSLOTS; ++s) if (G.ev_frame[s]) (void)hipEventDestroy(G.ev_frame[s]);
            if (G.self_frame) (void)hipFree(G.self_frame);
        }
    }
    for (auto &wp : G.w) {
        Worker *w = wp.get();
        if (!w->th.joinable()) continue;
        w->post([w] {
            for (int s = 0; s < XSLOTS; ++s) {
                if (w->pack[s]) (void)hipFree(w->pack[s]);
                if (w->ev_packed[s]) (void)hipEventDestroy(w->ev_packed[s]);
                if (w->ev_sent[s]) (void)hipEventDestroy(w->ev_sent[s]);
            }
            if (w->xs) (void)hipStreamDestroy(w->xs);
            one::pt_free();
            return PT_OK;
        });
        (void)w->wait();
        { std::lock_guard<std::mutex> lk(w->m); w->quit = true; }
        w->cv_job.notify_one();
        w->th.join();
    }
    G = Group{};
}

// "0,2,3" / "all" / "4" (= the first four) -> device ordinals
bool parse_devices(const char *text, int ndev, std::vector<int> &out) {
    out.clear();
    std::string t(text ? text : "");
    if (t.empty()) return false;
    if (t == "all") { for (int k = 0; k < ndev; ++k) out.push_back(k); return true; }
    size_t pos = 0;
    while (pos <= t.size()) {
        size_t end = t.find(',', pos);
        if (end == std::string::npos) end = t.size();
        const std::string tok = t.substr(pos, end - pos);
        if (tok.empty() || tok.find_first_not_of("0123456789") != std::string::npos) return false;
        out.push_back(atoi(tok.c_str()));
        pos = end + 1;
    }
    return !out.empty();
}

int multi_init(const pt_scene_desc *d, const std::vector<int> &devs) {
    const int K = (int)devs.size();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(PT_ERR_DEVICE, "pt_init: no HIP device (this library has no CPU fallback)");
    for (int k = 0; k < K; ++k)
        if (devs[(size_t)k] < 0 || devs[(size_t)k] >= ndev) return fail(PT_ERR_INVALID, "pt_init: device %d of %d", devs[(size_t)k], ndev);
    if (d->tile_count > 1)
        return fail(PT_ERR_INVALID, "pt_init: a session over several devices tiles the frame itself (tile_count must be 0 or 1)");
    bool distinct = true;
    for (int a = 0; a < K; ++a) for (int b = 0; b < a; ++b) if (devs[(size_t)a] == devs[(size_t)b]) distinct = false;
    // transport: RCCL whenever it can carry the exchange (every context on its own device), peer copies otherwise
    const char *x = getenv("PTMI355_XCHG");
    bool want_rccl = distinct && K > 1;
    if (x && !strcmp(x, "peer")) want_rccl = false;
    if (x && !strcmp(x, "rccl")) {
        if (!distinct) return fail(PT_ERR_INVALID, "pt_init: PTMI355_XCHG=rccl needs every context on its own device (RCCL refuses duplicates)");
        want_rccl = true;
    }
    if (want_rccl && !g_rccl.load()) {
        if (x && !strcmp(x, "rccl")) return fail(PT_ERR_DEVICE, "pt_init: librccl could not be loaded: %s", g_rccl.why.c_str());
        want_rccl = false;                                    // auto: fall back to peer copies
    }
    G = Group{};
    G.K = K; G.use_rccl = want_rccl; G.self_exchange = want_rccl && K == 1;
    G.W = d->camera.resolution[0]; G.H = d->camera.resolution[1]; G.npix = G.W * G.H;
    G.transport = want_rccl ? "rccl" : (K > 1 ? "peer" : "none");
    const int strip = d->strip_rows > 0 ? d->strip_rows : 8;
    for (int k = 0; k < K; ++k) {
        G.w.emplace_back(new Worker());
        Worker *w = G.w.back().get();
        w->index = k; w->device = devs[(size_t)k];
        w->th = std::thread([w] { w->loop(); });
    }
    G.live = true;
    pt_scene_desc base = *d;
    base.devices = nullptr; base.num_devices = 0;
    base.flags &= ~(uint32_t)PT_ASYNC_IMAGE;                  // the frame is assembled before it is copied out: synchronous
    int rc = on_all([&](Worker &w) -> int {
        pt_scene_desc mine = base;
        mine.device = w.device;
        mine.tile_index = w.index; mine.tile_count = G.K; mine.strip_rows = strip;
        if (w.index != 0) { mine.device_image = nullptr; mine.stream = nullptr; }
        const int r = one::pt_init(&mine);
        if (r) return r;
        w.map = R.map;
        w.floats = (size_t)R.map.tile_pixels * 3;
        {   // the exchange stream outranks the launch streams: its kernels (the pack's successor RCCL kernel, the unpack) are
            // a few workgroups each and must not queue behind the persistent grids of the batches in flight, whose
            // workgroups otherwise take every slot a retiring workgroup frees
            int lo = 0, hi = 0;
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
            if (pt_experiment("PTMI355_XCHG_PRIO") && atoi(pt_experiment("PTMI355_XCHG_PRIO")) == 0) hi = lo = 0;
            if (hipStreamCreateWithPriority(&w.xs, hipStreamNonBlocking, hi) != hipSuccess) {
                (void)hipGetLastError();
                HIPCHK(hipStreamCreateWithFlags(&w.xs, hipStreamNonBlocking));
            }
        }
        for (int s = 0; s < XSLOTS; ++s) {
            if (w.index != 0 || G.self_exchange) HIPCHK(hipMalloc(&w.pack[s], w.floats * 4));
            HIPCHK(hipEventCreateWithFlags(&w.ev_packed[s], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&w.ev_sent[s], hipEventDisableTiming));
        }
        return PT_OK;
    });
    if (rc) return rc;
    {
        DeviceGuard guard;
        Worker &root = *G.w[0];
        HIPCHK(hipSetDevice(root.device));
        for (int k = G.self_exchange ? 0 : 1; k < K; ++k)
            for (int s = 0; s < XSLOTS; ++s) HIPCHK(hipMalloc(&G.w[(size_t)k]->stage[s], G.w[(size_t)k]->floats * 4));
        for (int s = 0; s < XSLOTS; ++s) HIPCHK(hipEventCreateWithFlags(&G.ev_frame[s], hipEventDisableTiming));
        // The frame is context 0's accumulation buffer: the peers' rows are unpacked beside the rows it sums itself.  In the
        // one-context rehearsal the "peer" is the root, and unpacking its rows over themselves would make every gather
        // wait for the previous exchange (a dependency real peers do not have: round 3 measured the rehearsal at 13.5
        // Grays/s for that reason alone): there the tiles are assembled in a buffer of their own, so the rehearsal costs
        // what an exchange between devices costs -- pack, grouped send / recv, unpack -- and nothing else.
        G.frame = root.ctx.image;
        if (G.self_exchange) {
            HIPCHK(hipMalloc(&G.self_frame, (size_t)G.npix * 12));
            HIPCHK(hipMemset(G.self_frame, 0, (size_t)G.npix * 12));
            G.frame = G.self_frame;
        }
        if (!G.use_rccl)
            for (int k = 1; k < K; ++k) {
                const int dk = G.w[(size_t)k]->device;
                if (dk == root.device) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, dk, root.device) == hipSuccess && can) {
                    HIPCHK(hipSetDevice(dk));
                    const hipError_t e = hipDeviceEnablePeerAccess(root.device, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                    (void)hipGetLastError();
                }
            }
        if (G.use_rccl) {
            G.comms.assign((size_t)K, nullptr);
            NCCLCHK(g_rccl.CommInitAll(G.comms.data(), K, devs.data()));
        }
    }
    {   // The exchange of an asynchronous batch is issued from a thread of its own (Exchanger), one call behind the tracing:
        // a grouped RCCL send / recv costs the issuing thread ~80 us, and at one exchange per iteration (the north star's
        // cadence: 0.1 ms of tracing per iteration at 800x800) the host set the pace, not the device.
        // PTMI355_XCHG_THREAD=0 issues it from the caller's thread as rounds 1-3 did.
        bool on = true;
        if (const char *e = pt_experiment("PTMI355_XCHG_THREAD")) on = atoi(e) != 0;
        if (on) {
            G.x.reset(new Exchanger());
            Exchanger *x = G.x.get();
            x->th = std::thread([x] { x->loop(); });
        }
    }
    {   // can every context take pathtrace() with a host image as one self-gathering launch?
        std::vector<int> can((size_t)K, 0);
        (void)on_all([&](Worker &w) -> int { can[(size_t)w.index] = one::whole_host_possible() ? 1 : 0; return PT_OK; });
        G.direct_ok = true;
        for (int v : can) G.direct_ok = G.direct_ok && v;
        if (const char *e = pt_experiment("PTMI355_MULTI_DIRECT")) G.direct_enabled = atoi(e) != 0;
    }
    G.last_cam = d->camera; G.last_depth = d->trace_depth;
    g_xstats.on = pt_experiment("PTMI355_XCHG_STATS") && atoi(pt_experiment("PTMI355_XCHG_STATS")) != 0;
    t_err[0] = 0;
    return PT_OK;
}

}  // namespace

// ===========================================================================
// C-ABI (include/ptmi355.h)
// ===========================================================================
extern "C" {

const char *pt_last_error(void) { return g_err; }
const char *pt_version(void) { return one::pt_version(); }

void pt_free(void) {
    if (G.live || !G.w.empty()) { multi_free(); return; }
    one::pt_free();
}

int pt_init(const pt_scene_desc *d) {
    if (!d) return fail(PT_ERR_INVALID, "pt_init: null descriptor");
    pt_free();
    std::vector<int> devs;
    if (d->num_devices < 0 || (d->num_devices > 0 && !d->devices)) return fail(PT_ERR_INVALID, "pt_init: bad device list");
    if (d->num_devices > 0) devs.assign(d->devices, d->devices + d->num_devices);
    else if (const char *e = getenv("PTMI355_DEVICES")) {
        // the reference's host knows nothing about devices (cudaGLSetGLDevice(0), preview.cpp:107): the environment
        // spreads its frame over several without touching it.  Sessions that are themselves one tile of K processes
        // (tile_count > 1: bench.py under torch.distributed, ptbench --tile) keep their one device.
        int ndev = 0;
        if (*e && d->tile_count <= 1) {
            if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
                return fail(PT_ERR_DEVICE, "pt_init: no HIP device (this library has no CPU fallback)");
            if (!parse_devices(e, ndev, devs)) return fail(PT_ERR_INVALID, "pt_init: PTMI355_DEVICES='%s' is not a device list", e);
        }
    }
    const char *x = getenv("PTMI355_XCHG");
    if (devs.size() > 1 || (devs.size() == 1 && x && !strcmp(x, "rccl"))) {
        const int rc = multi_init(d, devs);
        if (rc != PT_OK) {
            char keep[ERR_BYTES];
            memcpy(keep, t_err, sizeof keep);
            multi_free();
            memcpy(t_err, keep, sizeof keep);
        }
        return rc;
    }
    if (devs.size() == 1) { pt_scene_desc one_dev = *d; one_dev.device = devs[0]; return one::pt_init(&one_dev); }
    return one::pt_init(d);
}

int pt_num_devices(void) { return G.live ? G.K : (g_single.live ? 1 : 0); }
const char *pt_exchange_transport(void) { return G.live ? G.transport.c_str() : "none"; }

int pt_set_camera(const pt_camera *camera, int trace_depth) {
    if (!G.live) return one::pt_set_camera(camera, trace_depth);
    if (!camera) return fail(PT_ERR_INVALID, "pt_set_camera: null camera");
    // the shim forwards the camera on every pathtrace() (pathtrace.cu:285-286): nothing to tell the devices when it
    // has not changed
    if (memcmp(&G.last_cam, camera, sizeof G.last_cam) == 0 && trace_depth == G.last_depth) return PT_OK;
    const int rc = on_all([&](Worker &) -> int { return one::pt_set_camera(camera, trace_depth); });
    if (rc == PT_OK) { G.last_cam = *camera; G.last_depth = trace_depth; }
    else G.last_depth = -1;
    return rc;
}

int pt_set_lens(float lens_radius, float focal_distance) {
    if (!G.live) return one::pt_set_lens(lens_radius, focal_distance);
    return on_all([&](Worker &) -> int { return one::pt_set_lens(lens_radius, focal_distance); });
}

int pt_synchronize(void) {
    if (!G.live) return one::pt_synchronize();
    return multi_sync();
}

int pt_trace_batch_async(int iter0, int count) {
    if (!G.live) return one::pt_trace_batch_async(iter0, count);
    return multi_enqueue(iter0, count, true);
}

// the calls that hand the image back: launches + exchange enqueued on every device, then the frame -> host copy on the
// root's exchange stream, then one synchronisation per context (which also folds its statistics)
// the caller's host image, page-locked once for every device (PT_PIN_IMAGE: it outlives the session)
static bool multi_pin(float *host, size_t bytes) {
    if (G.dhost == host && G.dhost_bytes >= bytes) return true;
    if (G.dhost) { (void)hipHostUnregister(G.dhost); G.dhost = nullptr; G.dhost_bytes = 0; }
    for (auto &wp : G.w) { wp->host_dev = nullptr; wp->host_dev_of = nullptr; }
    if (bytes < ((size_t)1 << 20)) return false;
    if (hipHostRegister(host, bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) { (void)hipGetLastError(); return false; }
    G.dhost = host; G.dhost_bytes = bytes;
    return true;
}

static int multi_trace(uint8_t *pbo_rgba, int iter0, int count, float *host_image_sum) {
    // pathtrace() with the host image and no PBO, one iteration: no exchange at all -- every context's launch writes its
    // own tile's pixels (those whose sum changed) into the caller's image while it traces (pt_trace_mapped)
    if (host_image_sum && !pbo_rgba && count == 1 && G.direct_ok && G.direct_enabled && !G.self_exchange &&
        (G.w[0]->ctx.flags & PT_PIN_IMAGE) && multi_pin(host_image_sum, (size_t)G.npix * 12)) {
        int rc = exchange_settled();
        if (rc) return rc;
        rc = on_all([&](Worker &w) -> int {
            if (w.host_dev_of != G.dhost) {
                void *dp = nullptr;
                HIPCHK(hipHostGetDevicePointer(&dp, G.dhost, 0));
                w.host_dev = (float *)dp; w.host_dev_of = G.dhost;
            }
            return one::pt_trace_mapped(iter0, w.host_dev);
        });
        __atomic_store_n(&G.frame_stale, true, __ATOMIC_RELAXED);
        return rc;
    }
    int rc = multi_enqueue(iter0, count);
    if (rc) return rc;
    Worker &root = *G.w[0];
    {
        DeviceGuard guard;
        HIPCHK(hipSetDevice(root.device));
        if (pbo_rgba) {
            hipLaunchKernelGGL(k_tonemap, dim3((G.npix + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, root.xs, pbo_rgba, G.frame, G.npix, iter0);
            HIPCHK(hipGetLastError());
        }
        if (host_image_sum)
            HIPCHK(hipMemcpyAsync(host_image_sum, G.frame, (size_t)G.npix * 12, hipMemcpyDeviceToHost, root.xs));
    }
    return on_all([&](Worker &w) -> int {
        const int r = collect_stats();
        if (r) return r;
        HIPCHK(hipStreamSynchronize(w.xs));
        return PT_OK;
    });
}

int pt_trace_batch(int iter0, int count, float *host_image_sum) {
    if (!G.live) return one::pt_trace_batch(iter0, count, host_image_sum);
    return multi_trace(nullptr, iter0, count, host_image_sum);
}

int pt_trace(uint8_t *pbo_rgba, int frame, int iter, float *host_image_sum) {
    if (!G.live) return one::pt_trace(pbo_rgba, frame, iter, host_image_sum);
    return multi_trace(pbo_rgba, iter, 1, host_image_sum);
}

#define PT_SINGLE_ONLY(name)                                                                                   \
    if (G.live) return fail(PT_ERR_INVALID, name ": the stepping interface drives one device; this session has %d", G.K)

int pt_trace_begin(int iter0, int count) { PT_SINGLE_ONLY("pt_trace_begin"); return one::pt_trace_begin(iter0, count); }
int pt_trace_bounce(int depth, int *n_live_after) { PT_SINGLE_ONLY("pt_trace_bounce"); return one::pt_trace_bounce(depth, n_live_after); }
int pt_trace_end(void) { PT_SINGLE_ONLY("pt_trace_end"); return one::pt_trace_end(); }
int pt_export_paths(pt_path_segment *host_paths, int capacity, int *n_live) {
    PT_SINGLE_ONLY("pt_export_paths");
    return one::pt_export_paths(host_paths, capacity, n_live);
}
int pt_export_intersections(pt_shadeable_intersection *host_isects, uint8_t *host_outside, int capacity) {
    PT_SINGLE_ONLY("pt_export_intersections");
    return one::pt_export_intersections(host_isects, host_outside, capacity);
}
int pt_intersect_once(const pt_path_segment *host_paths, int n, pt_shadeable_intersection *host_isects, uint8_t *host_outside) {
    if (!G.live) return one::pt_intersect_once(host_paths, n, host_isects, host_outside);
    return on_one(0, [&](Worker &) -> int { return one::pt_intersect_once(host_paths, n, host_isects, host_outside); });
}

int pt_get_image(float *host_image_sum) {
    if (!G.live) return one::pt_get_image(host_image_sum);
    if (!host_image_sum) return fail(PT_ERR_INVALID, "pt_get_image: null buffer");
    const int rc = multi_sync();
    if (rc) return rc;
    if (G.self_frame)                                          // the one-context rehearsal assembles the frame beside the sum
        return on_one(0, [&](Worker &) -> int { HIPCHK(hipMemcpy(host_image_sum, G.self_frame, (size_t)G.npix * 12, hipMemcpyDeviceToHost)); return PT_OK; });
    return on_one(0, [&](Worker &) -> int { return one::pt_get_image(host_image_sum); });      // context 0's buffer is the frame
}

int pt_tonemap(uint8_t *host_rgba, int iter) {
    if (!G.live) return one::pt_tonemap(host_rgba, iter);
    const int rc = multi_sync();
    if (rc) return rc;
    return on_one(0, [&](Worker &) -> int { return one::pt_tonemap(host_rgba, iter); });
}

int pt_clear_image(void) {
    if (!G.live) return one::pt_clear_image();
    const int rc = multi_sync();
    if (rc) return rc;
    return on_all([&](Worker &) -> int {
        if (G.self_frame) HIPCHK(hipMemset(G.self_frame, 0, (size_t)G.npix * 12));
        return one::pt_clear_image();
    });
}

// every context gets the whole frame: it adds to (and packs) its own rows only, and context 0's buffer IS the frame
int pt_set_image(const float *host_image_sum) {
    if (!G.live) return one::pt_set_image(host_image_sum);
    if (!host_image_sum) return fail(PT_ERR_INVALID, "pt_set_image: null buffer");
    const int rc = multi_sync();
    if (rc) return rc;
    return on_all([&](Worker &) -> int {
        if (G.self_frame) HIPCHK(hipMemcpy(G.self_frame, host_image_sum, (size_t)G.npix * 12, hipMemcpyHostToDevice));
        return one::pt_set_image(host_image_sum);
    });
}

float *pt_device_image(void) {
    if (!G.live) return one::pt_device_image();
    (void)multi_refresh();      // after calls that wrote the host image directly: the peers' rows travel now (stream-ordered on devices[0]'s exchange stream; pt_synchronize before reading)
    return G.frame;
}

long long pt_total_rays(void) {
    if (!G.live) return one::pt_total_rays();
    std::vector<long long> r((size_t)G.K, 0);
    const int rc = on_all([&](Worker &w) -> int { r[(size_t)w.index] = one::pt_total_rays(); return r[(size_t)w.index] < 0 ? (int)r[(size_t)w.index] : PT_OK; });
    if (rc) return rc;
    long long sum = 0;
    for (long long v : r) sum += v;
    return sum;
}

int pt_get_counters(int64_t *rays, int64_t *first_bounce_rays, int64_t *iterations) {
    if (!G.live) return one::pt_get_counters(rays, first_bounce_rays, iterations);
    std::vector<int64_t> a((size_t)G.K, 0), b((size_t)G.K, 0), c((size_t)G.K, 0);
    const int rc = on_all([&](Worker &w) -> int { return one::pt_get_counters(&a[(size_t)w.index], &b[(size_t)w.index], &c[(size_t)w.index]); });
    if (rc) return rc;
    int64_t sa = 0, sb = 0;
    for (int k = 0; k < G.K; ++k) { sa += a[(size_t)k]; sb += b[(size_t)k]; }
    if (rays) *rays = sa;
    if (first_bounce_rays) *first_bounce_rays = sb;
    if (iterations) *iterations = c[0];                        // every context traces every iteration (of its tile)
    return PT_OK;
}

int pt_get_stats(pt_stats *stats) {
    if (!G.live) return one::pt_get_stats(stats);
    if (!stats) return fail(PT_ERR_INVALID, "pt_get_stats: null");
    pt_stats sum{};
    for (auto &wp : G.w) {                                     // filled by collect_stats on the workers; they are idle now
        const pt_stats &s = wp->ctx.stats;
        sum.bounces = std::max(sum.bounces, s.bounces);
        sum.rays += s.rays; sum.total_rays += s.total_rays;
        for (int dd = 0; dd < 64; ++dd) sum.live[dd] += s.live[dd];
        sum.total_iterations = s.total_iterations;
    }
    *stats = sum;
    return PT_OK;
}

int pt_set_profiling(int enable) {
    if (!G.live) return one::pt_set_profiling(enable);
    return on_all([&](Worker &) -> int { return one::pt_set_profiling(enable); });
}

int pt_get_profile(pt_profile *out) {
    if (!G.live) return one::pt_get_profile(out);
    if (!out) return fail(PT_ERR_INVALID, "pt_get_profile: null");
    std::vector<pt_profile> p((size_t)G.K);
    const int rc = on_all([&](Worker &w) -> int { return one::pt_get_profile(&p[(size_t)w.index]); });
    if (rc) return rc;
    pt_profile sum{};
    for (const pt_profile &q : p)
        for (int s = 0; s < PT_STAGE_COUNT; ++s) { sum.ms[s] += q.ms[s]; sum.launches[s] += q.launches[s]; }
    *out = sum;                                                // summed over the devices: ms / launches is still the mean launch
    return PT_OK;
}

int pt_get_bvh_info(pt_bvh_info *out) {
    if (!G.live) return one::pt_get_bvh_info(out);
    return on_one(0, [&](Worker &) -> int { return one::pt_get_bvh_info(out); });
}

int pt_bvh_build(const pt_triangle *triangles, int count, float *nodes, int node_capacity, int32_t *order, float *grid) {
    return one::pt_bvh_build(triangles, count, nodes, node_capacity, order, grid);
}
int pt_tri_bounds(const pt_triangle *triangles, int count, float origin_bound, float *bounds) {
    return one::pt_tri_bounds(triangles, count, origin_bound, bounds);
}
int pt_tri_records(const pt_triangle *triangles, int count, float origin_bound, uint16_t *records, float frame[4]) {
    return one::pt_tri_records(triangles, count, origin_bound, records, frame);
}
int pt_cull_boxes(const pt_geom *geoms, int count, const float *eye, float *boxes, float *origin_bound, float *reject) {
    return one::pt_cull_boxes(geoms, count, eye, boxes, origin_bound, reject);
}
int pt_probe_rng(const uint32_t *seeds, int n, int draws, uint32_t *state, float *u) { return one::pt_probe_rng(seeds, n, draws, state, u); }
int pt_probe_sincos(const float *x, uint32_t first_bits, uint32_t n, float *s, float *c, uint64_t sum[2]) {
    return one::pt_probe_sincos(x, first_bits, n, s, c, sum);
}
int pt_probe_hemisphere(const float *normals, const uint32_t *seeds, int n, float *dirs) { return one::pt_probe_hemisphere(normals, seeds, n, dirs); }
int pt_probe_sqrt(uint32_t first_bits, uint32_t n, uint64_t mismatch[2]) { return one::pt_probe_sqrt(first_bits, n, mismatch); }
int pt_probe_clock(int microseconds, double *ghz) { return one::pt_probe_clock(microseconds, ghz); }

}  // extern "C"
