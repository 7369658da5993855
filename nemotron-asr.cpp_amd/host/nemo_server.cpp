// nemo-server-amd -- multi-stream streaming ASR server speaking the reference's wire protocol
// (reference src/server-protocol.h:24-41: 9-byte frame = u8 opcode, u32le stream id, u32le payload length;
// flow STREAM_START/STARTED, PUSH/ACK, SET_LANG/LANG_SET, TEXT, STREAM_END/ENDED, ERROR).
//
// Reader threads only move bytes into one FIFO, as in the reference (src/nemo-server.cpp:288-358).  The worker
// differs: the reference processes ONE data event of ONE stream at a time (src/nemo-server.cpp:230-239);
// here the worker drains everything that is queued, concatenates each session's pending audio, groups the
// sessions by right_context and issues ONE nemo_stream_process_batch per group -- the batch former that the
// MI355X engine needs (B streams per launch sequence).  One worker thread owns each engine; with --devices there is
// one engine + FIFO + worker ("lane") per GPU and stream s is served by lane s mod count, no cross-GPU traffic.
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
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

#include "nemo_amd.h"

namespace {

enum : uint8_t { OP_STREAM_START = 0x01, OP_PUSH = 0x02, OP_STREAM_END = 0x03, OP_SET_LANG = 0x04,
                 OP_STARTED = 0x81, OP_ACK = 0x82, OP_TEXT = 0x83, OP_ENDED = 0x84, OP_LANG_SET = 0x85, OP_ERROR = 0x8F };
constexpr size_t kHeader = 9;
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

// minimal JSON field extraction for {"lang":"xx","right_context":N}
bool json_str(const std::string &j, const char *key, std::string &out) {
    const std::string k = std::string("\"") + key + "\"";
    size_t p = j.find(k);
    if (p == std::string::npos) return false;
    p = j.find(':', p + k.size());
    if (p == std::string::npos) return false;
    p = j.find('"', p);
    if (p == std::string::npos) return false;
    const size_t e = j.find('"', p + 1);
    if (e == std::string::npos) return false;
    out = j.substr(p + 1, e - p - 1);
    return true;
}
bool json_int(const std::string &j, const char *key, int &out) {
    const std::string k = std::string("\"") + key + "\"";
    size_t p = j.find(k);
    if (p == std::string::npos) return false;
    p = j.find(':', p + k.size());
    if (p == std::string::npos) return false;
    out = atoi(j.c_str() + p + 1);
    return true;
}

struct Session {
    std::shared_ptr<Conn> conn;
    nemo_stream_context *sctx = nullptr;
    std::vector<int16_t> pending;
};

void worker_loop(Lane *lane) {
    nemo_context *model = lane->model;
    std::map<uint32_t, Session> sessions;
    uint64_t n_batches = 0, n_batched_streams = 0;
    std::map<int, uint64_t> b_hist;               // streams per engine call -> calls (printed at exit: tests/server_load.py reads it)
    std::set<uint32_t> in_flight;               // --pipeline: sessions whose last steps may still be on the GPU
    auto drain = [&]() {                         // complete the steps in flight and send their text (FIFO empty, END, LANG)
        if (in_flight.empty()) return;
        std::vector<uint32_t> ids;
        std::vector<nemo_stream_context *> sc;
        for (uint32_t id : in_flight) {
            auto it = sessions.find(id);
            if (it != sessions.end()) { ids.push_back(id); sc.push_back(it->second.sctx); }
        }
        in_flight.clear();
        if (sc.empty()) return;
        std::vector<std::string> out(sc.size());
        const bool ok = nemo_stream_collect_batch(sc.data(), (int)sc.size(), out.data());
        for (size_t b = 0; b < sc.size(); b++) {
            Session &s = sessions[ids[b]];
            if (!ok) s.conn->send_str(OP_ERROR, ids[b], "engine collect failed");
            else if (!out[b].empty()) s.conn->send_str(OP_TEXT, ids[b], out[b]);
        }
    };
    auto flush = [&]() {
        // group sessions with pending audio by right_context; one engine call per group
        std::map<int, std::vector<uint32_t>> groups;
        for (auto &kv : sessions)
            if (!kv.second.pending.empty()) groups[kv.second.sctx->config.att_right_context].push_back(kv.first);
        for (auto &g : groups) {
            const int B = (int)g.second.size();
            std::vector<nemo_stream_context *> sc((size_t)B);
            std::vector<const int16_t *> pcm((size_t)B);
            std::vector<int> ns((size_t)B);
            std::vector<std::string> out((size_t)B);
            for (int b = 0; b < B; b++) {
                Session &s = sessions[g.second[(size_t)b]];
                sc[(size_t)b] = s.sctx; pcm[(size_t)b] = s.pending.data(); ns[(size_t)b] = (int)s.pending.size();
            }
            const bool ok = nemo_stream_process_batch(sc.data(), B, pcm.data(), ns.data(), out.data());
            n_batches++; n_batched_streams += (uint64_t)B; b_hist[B]++;
            for (int b = 0; b < B; b++) {
                Session &s = sessions[g.second[(size_t)b]];
                s.pending.clear();
                if (g_pipeline > 0) in_flight.insert(g.second[(size_t)b]);
                if (!ok) s.conn->send_str(OP_ERROR, g.second[(size_t)b], "engine step failed");
                else if (!out[(size_t)b].empty()) s.conn->send_str(OP_TEXT, g.second[(size_t)b], out[(size_t)b]);
            }
        }
    };
    for (;;) {
        std::deque<Event> batch;
        {
            std::unique_lock<std::mutex> lk(lane->mtx);
            lane->cv.wait(lk, [&] { return g_stop || !lane->queue.empty(); });
            if (g_stop && lane->queue.empty()) break;
            batch.swap(lane->queue);             // take EVERYTHING that is queued: this is the batch former
            lane->queued_bytes = 0;
            lane->space_cv.notify_all();
        }
        for (Event &ev : batch) {
            auto it = sessions.find(ev.id);
            switch (ev.type) {
            case Ev::CREATE: {
                nemo_cache_config cfg = nemo_cache_config::default_config();
                cfg.att_right_context = ev.right_context;
                Session s;
                s.conn = ev.conn;
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
                flush();                         // audio queued before the switch uses the old language
                drain();
                if (nemo_stream_set_language(it->second.sctx, ev.text.c_str())) {
                    char buf[160];
                    const int n = snprintf(buf, sizeof(buf), "{\"id\":%u,\"lang\":\"%s\",\"index\":%d}", ev.id, ev.text.c_str(), it->second.sctx->prompt_index);
                    it->second.conn->send(OP_LANG_SET, ev.id, buf, (uint32_t)n);
                } else it->second.conn->send_str(OP_ERROR, ev.id, "unknown or unsupported language: " + ev.text);
                break;
            case Ev::END:
                if (it == sessions.end()) break;
                flush();
                in_flight.erase(ev.id);           // finalize completes the steps in flight itself
                it->second.conn->send_str(OP_ENDED, ev.id, nemo_stream_finalize(it->second.sctx));
                nemo_stream_free(it->second.sctx);
                sessions.erase(it);
                break;
            case Ev::CLOSE:
                if (it == sessions.end()) break;
                it->second.pending.clear();
                in_flight.erase(ev.id);
                nemo_stream_free(it->second.sctx);
                sessions.erase(it);
                break;
            }
        }
        flush();
        if (g_pipeline > 0) {                    // nothing else queued: do not sit on finished text
            bool idle;
            { std::lock_guard<std::mutex> lk(lane->mtx); idle = lane->queue.empty(); }
            if (idle) drain();
        }
    }
    for (auto &kv : sessions) nemo_stream_free(kv.second.sctx);
    fprintf(stderr, "worker: %llu engine calls, %.2f streams per call\n", (unsigned long long)n_batches,
            n_batches ? (double)n_batched_streams / (double)n_batches : 0.0);
    std::string h = "worker: B histogram";
    for (auto &kv : b_hist) h += " " + std::to_string(kv.first) + ":" + std::to_string(kv.second);
    fprintf(stderr, "%s\n", h.c_str());
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
        const uint8_t op = h[0];
        const uint32_t id = (uint32_t)h[1] | ((uint32_t)h[2] << 8) | ((uint32_t)h[3] << 16) | ((uint32_t)h[4] << 24);
        const uint32_t len = (uint32_t)h[5] | ((uint32_t)h[6] << 8) | ((uint32_t)h[7] << 16) | ((uint32_t)h[8] << 24);
        if (len > (256u << 20)) { conn->send_str(OP_ERROR, id, "payload too large"); break; }
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
        fprintf(stderr, "Usage: %s <model.gguf> [--tcp host:port | --unix path] [--right-context R] [--device N | --devices N,M,...] [--f32] [--max-streams N] [--pipeline E]\n"
                        "  --pipeline E: consecutive engine calls overlap on the GPU (E = 0..4; 4 pieces on 4 hardware queues is the throughput optimum, 0 the lowest latency); a stream's text arrives\n"
                        "                E calls later while the FIFO is busy and at once when it runs empty\n"
                        "  --devices: one engine + worker per listed GPU; stream s is served by entry s mod count\n", argv[0]);
        return 1;
    }
    std::string tcp = "127.0.0.1:8765", unix_path;
    std::vector<int> devices{0};
    int dtype = 1, max_streams = 64;
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
        else if (a == "--f32") dtype = 0;
        else if (a == "--pipeline" && i + 1 < argc) g_pipeline = atoi(argv[++i]);
        else { fprintf(stderr, "Unknown flag: %s\n", a.c_str()); return 1; }
    }
    for (int dev : devices) {
        std::unique_ptr<Lane> ln(new Lane());
        ln->model = nemo_init_with_device(argv[1], dev, dtype, max_streams);
        if (!ln->model) { fprintf(stderr, "Failed to load ASR model on device %d\n", dev); return 1; }
        if (g_pipeline > 0 && !nemo_set_pipeline(ln->model, g_pipeline)) return 1;
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
        std::thread(reader_loop, cfd).detach();
    }
    g_stop = true;
    wake_lanes();
    for (auto &ln : g_lanes) { ln->worker.join(); nemo_free(ln->model); }
    if (!unix_path.empty()) ::unlink(unix_path.c_str());
    return 0;
}
