// nemo-server-amd -- multi-stream streaming ASR server speaking the reference's wire protocol
// (reference src/server-protocol.h:24-41: 9-byte frame = u8 opcode, u32le stream id, u32le payload length;
// flow STREAM_START/STARTED, PUSH/ACK, SET_LANG/LANG_SET, TEXT, STREAM_END/ENDED, ERROR).
//
// Reader threads only move bytes into one FIFO, as in the reference (src/nemo-server.cpp:288-358).  The worker
// differs: the reference processes ONE data event of ONE stream at a time (src/nemo-server.cpp:230-239);
// here the worker is a batch former (below): it drains what is queued, keeps each session's audio until it holds whole
// chunks, and issues ONE nemo_stream_process_batch per right_context in which every stream completes the same number of
// chunks -- the shape the MI355X engine replays as a hipGraph.  One worker thread owns each engine; with --devices there is
// one engine + FIFO + worker ("lane") per GPU and stream s is served by lane s mod count, no cross-GPU traffic.
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <sys/un.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include <algorithm>
#include <chrono>
#include <cstdint>

#include "nemo_amd.h"
#include "nemotron_asr_amd.h"
#include "server_protocol.h"

namespace {

using namespace nasr_proto;
constexpr size_t kMaxQueuedBytes = 64u << 20;   // back-pressure, as the reference (64 MiB)

struct Conn {
    int fd;
    std::mutex wmtx;
    std::atomic<bool> closed{false};
    explicit Conn(int f) : fd(f) {}
    ~Conn() { ::close(fd); }
    void send(uint8_t op, uint32_t id, const void *payload, uint32_t len) {
        if (closed.load()) return;
        uint8_t h[kHeader] = {op, (uint8_t)id, (uint8_t)(id >> 8), (uint8_t)(id >> 16), (uint8_t)(id >> 24),
                              (uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
        std::lock_guard<std::mutex> lk(wmtx);
        if (::send(fd, h, kHeader, MSG_NOSIGNAL) != (ssize_t)kHeader) { closed = true; return; }
        size_t off = 0;
        while (off < len) {
            ssize_t k = ::send(fd, (const char *)payload + off, len - off, MSG_NOSIGNAL);
            if (k <= 0) { closed = true; return; }
            off += (size_t)k;
        }
    }
    void send_str(uint8_t op, uint32_t id, const std::string &s) { send(op, id, s.data(), (uint32_t)s.size()); }
};

enum class Ev { CREATE, DATA, LANG, END, CLOSE };
struct Event {
    Ev type;
    uint32_t id = 0;
    std::shared_ptr<Conn> conn;
    int right_context = 0;
    std::string text;                 // language code
    std::vector<int16_t> pcm;
};

// One lane per GPU: its own engine (weights replicated), FIFO and worker thread.  Session id s lives on lane
// s mod n_lanes for its whole life (streams are independent, no cross-GPU traffic).
struct Lane {
    nemo_context *model = nullptr;
    std::mutex mtx;
    std::condition_variable cv, space_cv;
    std::deque<Event> queue;
    size_t queued_bytes = 0;
    std::thread worker;
};
std::vector<std::unique_ptr<Lane>> g_lanes;
std::atomic<bool> g_stop{false};
std::atomic<uint32_t> g_next_id{1};
int g_default_rc = 0;
int g_pipeline = 0;

void enqueue(Event &&ev) {
    Lane &ln = *g_lanes[ev.id % g_lanes.size()];
    std::unique_lock<std::mutex> lk(ln.mtx);
    const size_t bytes = ev.pcm.size() * sizeof(int16_t);
    ln.space_cv.wait(lk, [&] { return g_stop || ln.queued_bytes + bytes <= kMaxQueuedBytes || ln.queued_bytes == 0; });
    ln.queued_bytes += bytes;
    ln.queue.push_back(std::move(ev));
    ln.cv.notify_one();
}

// ---- the batch former --------------------------------------------------------------------------------------------------
// The engine's fast path is a hipGraph replay (or, with --pipeline, graphs of consecutive calls side by side), and a call is
// graph-eligible only if every stream in it completes the SAME number of chunks (csrc/nasr_abi.hip: try_graph_step); anything
// else runs eagerly, chunk by chunk, after draining the pipeline.  Round 3's worker handed the engine whatever had arrived --
// ragged pushes, sessions one chunk apart -- and delivered a fifth of the engine's throughput (profiles/r3_server_load_64_streams.json).
// Now a session is only ever handed WHOLE chunks: how many chunks a session has completed is a pure function of the samples it
// has been given (256 zeros of left padding, 512-sample frames at hop 160, 9 zero frames of pre-encode cache, 8 T new frames per chunk:
// reference src/preprocessor.cpp:220-221, :320-328, src/nemo-stream.h:65-81), so the former knows for every session how many
// samples complete its next g chunks.  Per right_context, the sessions that hold at least one whole chunk form ONE call in which
// each completes G = the smallest number available among them (capped by the engine's row budget); the rest of their audio waits
// for the next call.  A forming window holds the call back for at most kFormingWindowUs after the first chunk became complete, or
// until every live session of that right_context has one -- live streams that push on the same clock land in the same call.
constexpr int kFormingWindowUs = 200;

struct Session {
    std::shared_ptr<Conn> conn;
    nemo_stream_context *sctx = nullptr;
    std::vector<int16_t> pending;                // audio received and not yet handed to the engine, from `head` on
    size_t head = 0;
    int64_t handed = 0;                          // samples the engine has been given since the stream began
    int T = 1;                                   // 1 + right_context
    size_t avail() const { return pending.size() - head; }
    int64_t samples_for_chunks(int64_t k) const { return nasr_proto::samples_for_chunks(k, T); }
    int64_t chunks_after(int64_t samples) const { return nasr_proto::chunks_after(samples, T); }
    int whole_chunks_pending() const { return (int)(chunks_after(handed + (int64_t)avail()) - chunks_after(handed)); }
    void consume(size_t n) {
        head += n; handed += (int64_t)n;
        if (head == pending.size()) { pending.clear(); head = 0; }
        else if (head > (1u << 20)) { pending.erase(pending.begin(), pending.begin() + (long)head); head = 0; }
    }
};

// Every batch size a lane can meet is a step shape of its own (hipGraphs per (streams, lookahead, chunks)), and capturing one takes
// ~20 ms -- the p99 of a live load whose calls carry 57..64 streams was those captures (profiles/r4_server_load.md).  --prewarm R[,R..]
// runs one chunk of silence through B = max_streams .. 1 scratch sessions per listed right_context before the socket opens, so the
// shapes of the streaming cadence (G = 1) exist when the first client connects; the graph cache is sized to keep them all.
void prewarm(nemo_context *model, int right_context) {
    const int N = model->max_streams, T = 1 + right_context;
    nemo_cache_config cfg = nemo_cache_config::default_config();
    cfg.att_right_context = right_context;
    std::vector<nemo_stream_context *> sc;
    for (int i = 0; i < N; i++) {
        nemo_stream_context *s = nemo_stream_init(model, &cfg);
        if (!s) break;
        sc.push_back(s);
    }
    const size_t first = (size_t)nasr_proto::samples_for_chunks(1, T), shift = (size_t)1280 * (size_t)T;      // the first chunk, one more chunk
    const std::vector<int16_t> silence(first, 0);
    const auto t0 = std::chrono::steady_clock::now();
    for (int B = (int)sc.size(); B >= 1; B--) {          // the first call gives every session its first chunk, each later one another chunk to B of them
        std::vector<const int16_t *> pcm((size_t)B, silence.data());
        std::vector<int> ns((size_t)B, (int)(B == (int)sc.size() ? first : shift));
        std::vector<std::string> out((size_t)B);
        if (!nemo_stream_process_batch(sc.data(), B, pcm.data(), ns.data(), out.data())) break;
    }
    if (!sc.empty()) { std::vector<std::string> out(sc.size()); nemo_stream_collect_batch(sc.data(), (int)sc.size(), out.data()); }
    for (nemo_stream_context *s : sc) nemo_stream_free(s);
    fprintf(stderr, "prewarm: right_context %d, %zu step shapes in %.2f s\n", right_context, sc.size(),
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
}

void worker_loop(Lane *lane) {
    nemo_context *model = lane->model;
    std::map<uint32_t, Session> sessions;
    uint64_t n_batches = 0, n_batched_streams = 0, n_partial = 0, n_finalize = 0;
    double t_apply = 0.0, t_call = 0.0, t_wait = 0.0;      // where the worker's wall time goes: events into sessions / inside engine calls / waiting for events (printed at exit)
    auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    std::vector<uint32_t> ending;               // sessions whose STREAM_END has been read
    std::map<int, uint64_t> b_hist;               // streams per engine call -> calls (printed at exit: tests/server_load.py reads it)
    std::set<uint32_t> in_flight;               // --pipeline: sessions whose last steps may still be on the GPU
    const int row_budget = std::max(model->workspace_rows, std::max(model->max_streams * 14, 256));      // rows one launch sequence may carry (csrc/nasr_engine.hip: w_rows, nasr_engine_create_ex)
    auto drain = [&]() {                         // complete the steps in flight and send their text (FIFO empty, END, LANG)
        if (in_flight.empty()) return;
        std::map<int, std::vector<uint32_t>> by_T;         // one collect per right_context: an engine call takes streams of one lookahead
        for (uint32_t id : in_flight) {
            auto it = sessions.find(id);
            if (it != sessions.end()) by_T[it->second.T].push_back(id);
        }
        in_flight.clear();
        for (auto &g : by_T) {
            const std::vector<uint32_t> &ids = g.second;
            std::vector<nemo_stream_context *> sc;
            for (uint32_t id : ids) sc.push_back(sessions[id].sctx);
            std::vector<std::string> out(sc.size());
            const bool ok = nemo_stream_collect_batch(sc.data(), (int)sc.size(), out.data());
            for (size_t b = 0; b < sc.size(); b++) {
                Session &s = sessions[ids[b]];
                if (!ok) s.conn->send_str(OP_ERROR, ids[b], "engine collect failed");
                else if (!out[b].empty()) s.conn->send_str(OP_TEXT, ids[b], out[b]);
            }
        }
    };
    // one engine call: session ids[b] is handed n[b] samples of its pending audio
    auto call = [&](const std::vector<uint32_t> &ids, const std::vector<size_t> &n) {
        const int B = (int)ids.size();
        std::vector<nemo_stream_context *> sc((size_t)B);
        std::vector<const int16_t *> pcm((size_t)B);
        std::vector<int> ns((size_t)B);
        std::vector<std::string> out((size_t)B);
        for (int b = 0; b < B; b++) {
            Session &s = sessions[ids[(size_t)b]];
            sc[(size_t)b] = s.sctx; pcm[(size_t)b] = s.pending.data() + s.head; ns[(size_t)b] = (int)n[(size_t)b];
        }
        const double tc0 = now_s();
        const bool ok = nemo_stream_process_batch(sc.data(), B, pcm.data(), ns.data(), out.data());
        t_call += now_s() - tc0;
        n_batches++; n_batched_streams += (uint64_t)B; b_hist[B]++;
        for (int b = 0; b < B; b++) {
            Session &s = sessions[ids[(size_t)b]];
            s.consume(n[(size_t)b]);
            if (g_pipeline > 0) in_flight.insert(ids[(size_t)b]);
            if (!ok) s.conn->send_str(OP_ERROR, ids[(size_t)b], "engine step failed");
            else if (!out[(size_t)b].empty()) s.conn->send_str(OP_TEXT, ids[(size_t)b], out[(size_t)b]);
        }
    };
    // whole chunks: per right_context one call in which every session completes the same number of chunks.  Returns true if a
    // session still holds a whole chunk afterwards (the caller comes back without waiting).
    auto form_calls = [&]() {
        bool more = false;
        std::map<int, std::vector<uint32_t>> groups;
        for (auto &kv : sessions)
            if (kv.second.whole_chunks_pending() > 0) groups[kv.second.T].push_back(kv.first);
        for (auto &g : groups) {
            const int T = g.first;
            std::vector<uint32_t> ids = g.second;
            if ((int)ids.size() * T > row_budget) { ids.resize((size_t)(row_budget / T)); more = true; }
            // chunks per session in this call and who takes part: nasr_proto::pick_call (server_protocol.h; tests/test_sanitizers.py checks the rule on the host)
            int G = 1;
            {
                std::vector<int> pend(ids.size());
                for (size_t b = 0; b < ids.size(); b++) pend[b] = sessions[ids[b]].whole_chunks_pending();
                std::unique_ptr<bool[]> take(new bool[ids.size()]);
                G = nasr_proto::pick_call(pend.data(), (int)ids.size(), T, row_budget, model->max_streams, take.get());
                std::vector<uint32_t> keep;
                for (size_t b = 0; b < ids.size(); b++) {
                    if (take[b]) keep.push_back(ids[b]);
                    else more = true;
                }
                ids.swap(keep);
            }
            std::vector<size_t> n(ids.size());
            for (size_t b = 0; b < ids.size(); b++) {
                Session &s = sessions[ids[b]];
                n[b] = (size_t)(s.samples_for_chunks(s.chunks_after(s.handed) + G) - s.handed);
                more = more || s.whole_chunks_pending() > G;
            }
            call(ids, n);
        }
        return more;
    };
    // what is left of one session (less than a chunk; END, LANG): handed as it is -- an eager step, once per stream
    auto flush_session = [&](uint32_t id) {
        Session &s = sessions[id];
        while (s.whole_chunks_pending() > 0) form_calls();
        if (s.avail() == 0) return;
        n_partial++;
        call({id}, {s.avail()});
    };
    auto all_ready = [&]() {                    // every live session holds a whole chunk (per right_context that has one ready)
        std::set<int> ready_T;
        for (auto &kv : sessions) if (kv.second.whole_chunks_pending() > 0) ready_T.insert(kv.second.T);
        for (auto &kv : sessions) if (ready_T.count(kv.second.T) && kv.second.whole_chunks_pending() == 0) return false;
        return true;
    };
    auto any_ready = [&]() { for (auto &kv : sessions) if (kv.second.whole_chunks_pending() > 0) return true; return false; };
    auto apply = [&](Event &ev) {
        auto it = sessions.find(ev.id);
        switch (ev.type) {
        case Ev::CREATE: {
            nemo_cache_config cfg = nemo_cache_config::default_config();
            cfg.att_right_context = ev.right_context;
            Session s;
            s.conn = ev.conn;
            s.T = 1 + ev.right_context;
            s.sctx = nemo_stream_init(model, &cfg);
            if (!s.sctx) { ev.conn->send_str(OP_ERROR, ev.id, "failed to init stream"); break; }
            if (!ev.text.empty() && ev.text != "auto") nemo_stream_set_language(s.sctx, ev.text.c_str());
            sessions[ev.id] = std::move(s);
        } break;
        case Ev::DATA:
            if (it != sessions.end()) it->second.pending.insert(it->second.pending.end(), ev.pcm.begin(), ev.pcm.end());
            break;                           // stale data of a closed session is dropped, as in the reference
        case Ev::LANG:
            if (it == sessions.end()) break;
            flush_session(ev.id);            // audio queued before the switch uses the old language
            drain();
            if (nemo_stream_set_language(it->second.sctx, ev.text.c_str())) {
                char buf[160];
                const int n = snprintf(buf, sizeof(buf), "{\"id\":%u,\"lang\":\"%s\",\"index\":%d}", ev.id, ev.text.c_str(), it->second.sctx->prompt_index);
                it->second.conn->send(OP_LANG_SET, ev.id, buf, (uint32_t)n);
            } else it->second.conn->send_str(OP_ERROR, ev.id, "unknown or unsupported language: " + ev.text);
            break;
        case Ev::END:
            if (it != sessions.end()) ending.push_back(ev.id);     // ended together below: sessions that stop at the same time share their tail calls
            break;
        case Ev::CLOSE:
            if (it == sessions.end()) break;
            ending.erase(std::remove(ending.begin(), ending.end(), ev.id), ending.end());
            in_flight.erase(ev.id);
            nemo_stream_free(it->second.sctx);
            sessions.erase(it);
            break;
        }
    };
    // STREAM_END of every session in `ending`: the whole chunks they still hold go through the usual calls, what is left of each
    // (less than a chunk) in ONE call, then ONE tail flush for all of them (nasr_engine_finalize takes B streams)
    auto end_sessions = [&]() {
        if (ending.empty()) return;
        std::vector<uint32_t> ids;
        for (uint32_t id : ending) if (sessions.count(id) && std::find(ids.begin(), ids.end(), id) == ids.end()) ids.push_back(id);
        ending.clear();
        if (ids.empty()) return;
        for (bool again = true; again;) {
            again = false;
            for (uint32_t id : ids) again = again || sessions[id].whole_chunks_pending() > 0;
            if (again) form_calls();
        }
        std::map<int, std::vector<uint32_t>> by_T;         // the tail calls per right_context
        for (uint32_t id : ids) by_T[sessions[id].T].push_back(id);
        for (auto &g : by_T) {
            const std::vector<uint32_t> &gi = g.second;
            std::vector<uint32_t> tail_ids;
            std::vector<size_t> tail_n;
            for (uint32_t id : gi) if (sessions[id].avail() > 0) { tail_ids.push_back(id); tail_n.push_back(sessions[id].avail()); }
            if (!tail_ids.empty()) { n_partial++; call(tail_ids, tail_n); }
            std::vector<nemo_stream_context *> sc;
            for (uint32_t id : gi) { sc.push_back(sessions[id].sctx); in_flight.erase(id); }      // finalize completes the steps in flight itself
            std::vector<std::string> out(gi.size());
            const bool ok = nemo_stream_finalize_batch(sc.data(), (int)sc.size(), out.data());
            n_finalize++;
            for (size_t b = 0; b < gi.size(); b++) {
                Session &s = sessions[gi[b]];
                if (!ok) s.conn->send_str(OP_ERROR, gi[b], "engine finalize failed");
                s.conn->send_str(OP_ENDED, gi[b], out[b]);
                nemo_stream_free(s.sctx);
                sessions.erase(gi[b]);
            }
        }
    };
    auto take = [&](std::deque<Event> &batch) {          // everything that is queued, without waiting
        std::lock_guard<std::mutex> lk(lane->mtx);
        if (lane->queue.empty()) return false;
        batch.swap(lane->queue);
        lane->queued_bytes = 0;
        lane->space_cv.notify_all();
        return true;
    };
    bool backlog = false;                       // a session still holds a whole chunk: no waiting
    for (;;) {
        std::deque<Event> batch;
        const double tw0 = now_s();
        if (backlog) take(batch);
        else {
            std::unique_lock<std::mutex> lk(lane->mtx);
            lane->cv.wait(lk, [&] { return g_stop || !lane->queue.empty(); });
            if (g_stop && lane->queue.empty()) break;
            batch.swap(lane->queue);
            lane->queued_bytes = 0;
            lane->space_cv.notify_all();
        }
        const double ta0 = now_s();
        t_wait += ta0 - tw0;
        for (Event &ev : batch) apply(ev);
        t_apply += now_s() - ta0;
        // the forming window: the first whole chunk is in; wait for the sessions that push on the same clock (with a backlog too: a
        // call that leaves out the sessions whose next frame is a few hundred microseconds away is another, smaller step shape)
        if (any_ready() && !all_ready()) {
            // under a backlog the GPU still has the previous calls to work off (--pipeline keeps four in flight): waiting longer for the
            // sessions whose next frames are on the wire costs nothing and keeps the calls at the full batch size (one step shape)
            // (round 6: a 4 ms window that closes only when every session holds a full call's worth of chunks was measured too: 21.0-23.3 k RTFx against 22.1-22.5 k, nothing)
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(backlog ? 5 * kFormingWindowUs : kFormingWindowUs);
            for (;;) {
                std::deque<Event> more;
                {
                    std::unique_lock<std::mutex> lk(lane->mtx);
                    if (!lane->cv.wait_until(lk, deadline, [&] { return g_stop || !lane->queue.empty(); })) break;
                    if (lane->queue.empty()) break;
                    more.swap(lane->queue);
                    lane->queued_bytes = 0;
                    lane->space_cv.notify_all();
                }
                { const double ta1 = now_s(); for (Event &ev : more) apply(ev); t_apply += now_s() - ta1; }
                if (all_ready() || std::chrono::steady_clock::now() >= deadline) break;
            }
        }
        end_sessions();
        backlog = form_calls();
        if (g_pipeline > 0 && !backlog) {        // nothing else to do: do not sit on finished text
            bool idle;
            { std::lock_guard<std::mutex> lk(lane->mtx); idle = lane->queue.empty(); }
            if (idle) drain();
        }
    }
    for (auto &kv : sessions) nemo_stream_free(kv.second.sctx);
    fprintf(stderr, "worker: %llu engine calls, %.2f streams per call, %llu partial-chunk calls, %llu tail flushes\n", (unsigned long long)n_batches,
            n_batches ? (double)n_batched_streams / (double)n_batches : 0.0, (unsigned long long)n_partial, (unsigned long long)n_finalize);
    fprintf(stderr, "worker: wall time by activity: %.3f s inside engine calls, %.3f s moving received events into the sessions, %.3f s waiting for events\n", t_call, t_apply, t_wait);
    std::string h = "worker: B histogram";
    for (auto &kv : b_hist) h += " " + std::to_string(kv.first) + ":" + std::to_string(kv.second);
    fprintf(stderr, "%s\n", h.c_str());
    std::string c = "worker: engine counters";       // which path the calls took (tests/server_load.py reads this line)
    for (const char *name : {"graph_replays", "pipelined_steps", "grouped_steps", "eager_steps", "graph_shapes", "graph_evictions"}) {
        int64_t v = 0;
        if (nasr_engine_get_counter(model->engine, name, &v) == 0) c += std::string(" ") + name + ":" + std::to_string((long long)v);
    }
    fprintf(stderr, "%s\n", c.c_str());
}

bool recv_full(int fd, uint8_t *buf, size_t n) {
    size_t off = 0;
    while (off < n) {
        const ssize_t k = ::recv(fd, buf + off, n - off, 0);
        if (k <= 0) return false;
        off += (size_t)k;
    }
    return true;
}

void reader_loop(int fd) {
    auto conn = std::make_shared<Conn>(fd);
    std::vector<uint32_t> mine;
    for (;;) {
        uint8_t h[kHeader];
        if (!recv_full(fd, h, kHeader)) break;
        uint8_t op;
        uint32_t id, len;
        decode_header(h, op, id, len);
        if (len > kMaxPayload) { conn->send_str(OP_ERROR, id, "payload too large"); break; }
        std::vector<uint8_t> payload(len);
        if (len && !recv_full(fd, payload.data(), len)) break;
        Event ev;
        ev.conn = conn;
        switch (op) {
        case OP_STREAM_START: {
            const uint32_t nid = g_next_id.fetch_add(1);
            const std::string cfg(payload.begin(), payload.end());
            ev.type = Ev::CREATE; ev.id = nid; ev.right_context = g_default_rc;
            json_str(cfg, "lang", ev.text);
            json_int(cfg, "right_context", ev.right_context);
            if (!valid_right_context(ev.right_context)) {      // reference src/nemo-stream.h:15-20: the four latency modes
                conn->send_str(OP_ERROR, 0, "right_context must be 0, 1, 6 or 13");
                break;
            }
            char buf[48];
            const int n = snprintf(buf, sizeof(buf), "{\"id\":%u}", nid);
            conn->send(OP_STARTED, nid, buf, (uint32_t)n);
            mine.push_back(nid);
            enqueue(std::move(ev));
        } break;
        case OP_PUSH: {
            ev.type = Ev::DATA; ev.id = id;
            ev.pcm.resize(len / 2);
            if (len >= 2) memcpy(ev.pcm.data(), payload.data(), (len / 2) * 2);
            char buf[64];
            const int n = snprintf(buf, sizeof(buf), "{\"queued_samples\":%u}", len / 2);
            enqueue(std::move(ev));
            conn->send(OP_ACK, id, buf, (uint32_t)n);
        } break;
        case OP_SET_LANG:
            ev.type = Ev::LANG; ev.id = id; ev.text.assign(payload.begin(), payload.end());
            enqueue(std::move(ev));
            break;
        case OP_STREAM_END:
            ev.type = Ev::END; ev.id = id;
            enqueue(std::move(ev));
            break;
        default:
            conn->send_str(OP_ERROR, id, "unknown opcode");
        }
    }
    conn->closed = true;
    for (uint32_t sid : mine) { Event ev; ev.type = Ev::CLOSE; ev.id = sid; ev.conn = conn; enqueue(std::move(ev)); }
}

int g_listen_fd = -1;
void wake_lanes() { for (auto &ln : g_lanes) { ln->cv.notify_all(); ln->space_cv.notify_all(); } }
void on_signal(int) { g_stop = true; if (g_listen_fd >= 0) ::shutdown(g_listen_fd, SHUT_RDWR); wake_lanes(); }

}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "Usage: %s <model.gguf> [--tcp host:port | --unix path] [--right-context R] [--device N | --devices N,M,...] [--f32] [--max-streams N] [--backlog-chunks K] [--pipeline E] [--prewarm R[,R...] | --no-prewarm] [--cpu | --cuda]\n"
                        "  --prewarm: capture the step graphs of every batch size 1..max-streams for these right_context values before listening (~20 ms each);\n"
                        "             with --pipeline the default right_context is prewarmed unless --no-prewarm\n"
                        "  --tcp: default 127.0.0.1:8300 (the reference's port, src/nemo-server.cpp:411; the reference binds every interface when no host is given, this server binds loopback unless told otherwise: --tcp 0.0.0.0:8300)\n"
                        "  --pipeline E: consecutive engine calls overlap on the GPU (E = 0..4; 4 pieces on 4 hardware queues is the throughput optimum, 0 the lowest latency); a stream's text arrives\n"
                        "                E calls later while the FIFO is busy and at once when it runs empty\n"
                        "  --devices: one engine + worker per listed GPU; stream s is served by entry s mod count\n"
                        "  --backlog-chunks K: an engine call may carry up to K whole chunks of every stream (default 4; 1 = one chunk per stream and call)\n", argv[0]);
        return 1;
    }
    std::string tcp = "127.0.0.1:8300", unix_path;
    std::vector<int> devices{0};
    int dtype = 1, max_streams = 64, backlog_chunks = 4;
    std::vector<int> prewarm_rc;
    bool no_prewarm = false;
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--tcp" && i + 1 < argc) tcp = argv[++i];
        else if (a == "--unix" && i + 1 < argc) unix_path = argv[++i];
        else if (a == "--right-context" && i + 1 < argc) g_default_rc = atoi(argv[++i]);
        else if (a == "--device" && i + 1 < argc) devices.assign(1, atoi(argv[++i]));
        else if (a == "--devices" && i + 1 < argc) {
            devices.clear();
            for (const char *p = argv[++i]; *p;) {
                devices.push_back(atoi(p));
                while (*p && *p != ',') p++;
                if (*p == ',') p++;
            }
            if (devices.empty()) { fprintf(stderr, "--devices needs a comma-separated list\n"); return 1; }
        }
        else if (a == "--max-streams" && i + 1 < argc) max_streams = atoi(argv[++i]);
        else if (a == "--backlog-chunks" && i + 1 < argc) backlog_chunks = std::max(1, std::min(16, atoi(argv[++i])));
        else if (a == "--f32") dtype = 0;
        else if (a == "--cpu" || a == "--cuda" || a == "--metal") {
            // the reference server's backend switches (src/nemo-server.cpp:405-406): accepted so that its invocations run unchanged
            fprintf(stderr, "note: %s has no effect, this build runs on the MI355X engine (HIP)\n", a.c_str());
        }
        else if (a == "--pipeline" && i + 1 < argc) g_pipeline = atoi(argv[++i]);
        else if (a == "--no-prewarm") no_prewarm = true;
        else if (a == "--prewarm" && i + 1 < argc) {
            for (const char *p = argv[++i]; *p;) {
                prewarm_rc.push_back(atoi(p));
                while (*p && *p != ',') p++;
                if (*p == ',') p++;
            }
        }
        else { fprintf(stderr, "Unknown flag: %s\n", a.c_str()); return 1; }
    }
    if (prewarm_rc.empty() && !no_prewarm && g_pipeline > 0) prewarm_rc.push_back(g_default_rc);     // a throughput server: the default lookahead's shapes are ready when the socket opens
    for (int dev : devices) {
        std::unique_ptr<Lane> ln(new Lane());
        // room for backlog_chunks whole chunks of EVERY stream at the longest lookahead in one engine call: sessions that hold several chunks (a file, a client
        // that fell behind) are worked off in GEMMs of that many times the rows (64 streams x R = 13: 896 -> 3 584 rows)
        ln->model = nemo_init_with_rows(argv[1], dev, dtype, max_streams, max_streams * 14 * backlog_chunks);
        if (!ln->model) { fprintf(stderr, "Failed to load ASR model on device %d\n", dev); return 1; }
        if (g_pipeline > 0 && !nemo_set_pipeline(ln->model, g_pipeline)) return 1;
        // one step shape per batch size and right_context in use, and as many again for the multi-chunk shapes of a backlog (chunk counts are
        // powers of two): none is evicted in steady state
        if (nasr_engine_set_option(ln->model->engine, "graph_cache", std::max(16, 2 * (max_streams + 8) * std::max(1, (int)prewarm_rc.size()))) < 0) return 1;
        for (int rc : prewarm_rc) prewarm(ln->model, rc);
        g_lanes.push_back(std::move(ln));
    }
    int fd;
    if (!unix_path.empty()) {
        fd = ::socket(AF_UNIX, SOCK_STREAM, 0);
        sockaddr_un sa{};
        sa.sun_family = AF_UNIX;
        strncpy(sa.sun_path, unix_path.c_str(), sizeof(sa.sun_path) - 1);
        ::unlink(unix_path.c_str());
        if (fd < 0 || ::bind(fd, (sockaddr *)&sa, sizeof(sa)) != 0) { perror("bind"); return 1; }
    } else {
        const size_t c = tcp.rfind(':');
        const std::string host = c == std::string::npos ? "127.0.0.1" : tcp.substr(0, c);
        const int port = atoi(tcp.c_str() + (c == std::string::npos ? 0 : c + 1));
        fd = ::socket(AF_INET, SOCK_STREAM, 0);
        int one = 1;
        setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        sockaddr_in sa{};
        sa.sin_family = AF_INET;
        sa.sin_port = htons((uint16_t)port);
        inet_pton(AF_INET, host.c_str(), &sa.sin_addr);
        if (fd < 0 || ::bind(fd, (sockaddr *)&sa, sizeof(sa)) != 0) { perror("bind"); return 1; }
    }
    if (::listen(fd, 64) != 0) { perror("listen"); return 1; }
    g_listen_fd = fd;
    signal(SIGINT, on_signal);
    signal(SIGTERM, on_signal);
    fprintf(stderr, "listening on %s (default right_context %d, %zu GPU lane(s), max %d streams each)\n",
            unix_path.empty() ? tcp.c_str() : unix_path.c_str(), g_default_rc, g_lanes.size(), max_streams);
    for (auto &ln : g_lanes) ln->worker = std::thread(worker_loop, ln.get());
    while (!g_stop) {
        const int cfd = ::accept(fd, nullptr, nullptr);
        if (cfd < 0) { if (g_stop) break; continue; }
        int one = 1;
        setsockopt(cfd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
        // replies are sent from the worker thread: a client that stops reading must not stall every other session behind a full socket
        // buffer -- after 5 s without progress the send fails and the connection is marked closed (its sessions end with the reader)
        timeval snd{5, 0};
        setsockopt(cfd, SOL_SOCKET, SO_SNDTIMEO, &snd, sizeof(snd));
        std::thread(reader_loop, cfd).detach();
    }
    g_stop = true;
    wake_lanes();
    for (auto &ln : g_lanes) { ln->worker.join(); nemo_free(ln->model); }
    if (!unix_path.empty()) ::unlink(unix_path.c_str());
    return 0;
}
