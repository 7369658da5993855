// nemo-load-amd -- load generator for nemo-server-amd (and for the reference's nemo-server: same wire protocol,
// reference src/server-protocol.h:24-41).  Round 3's load client was one Python process and was itself the bottleneck of the
// burst figures (profiles/r3_server_load_64_streams.json); this one is native: per connection one sender and one receiver thread,
// the PCM of every stream read from a file up front.
//
//   nemo-load-amd --unix PATH | --tcp HOST:PORT  --pcm-dir DIR --streams N [--conns C] [--right-context R]
//                 [--mode burst|realtime] [--push-chunks K] [--out report.json]
//
// DIR/stream_%04d.s16 = raw s16le mono 16 kHz audio of stream i.  Every stream pushes K x 1280 (1 + R) samples per push
// (default K = 1: the streaming cadence); realtime = one push per stream every K x 80 (1 + R) ms, burst = back to back.
// The report (JSON on stdout or --out) holds, per stream: the transcript, the wall time of every push (seconds since the common
// start), every text arrival as [time, cumulative text length in bytes], the ENDED time; plus wall time from the first push to
// the last ENDED.  tests/server_load.py judges transcripts and latencies from it.
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

enum : uint8_t { OP_START = 0x01, OP_PUSH = 0x02, OP_END = 0x03, OP_STARTED = 0x81, OP_ACK = 0x82, OP_TEXT = 0x83, OP_ENDED = 0x84, OP_ERROR = 0x8F };

struct StreamState {
    std::vector<int16_t> pcm;
    uint32_t sid = 0;
    std::vector<double> send_times;
    std::vector<std::pair<double, size_t>> arrivals;
    std::string text, error;
    double ended = -1;
};

using Clock = std::chrono::steady_clock;
Clock::time_point g_t0;
double now_s() { return std::chrono::duration<double>(Clock::now() - g_t0).count(); }

bool send_all(int fd, const void *p, size_t n) {
    size_t off = 0;
    while (off < n) {
        const ssize_t k = ::send(fd, (const char *)p + off, n - off, MSG_NOSIGNAL);
        if (k <= 0) return false;
        off += (size_t)k;
    }
    return true;
}
bool recv_all(int fd, void *p, size_t n) {
    size_t off = 0;
    while (off < n) {
        const ssize_t k = ::recv(fd, (char *)p + off, n - off, 0);
        if (k <= 0) return false;
        off += (size_t)k;
    }
    return true;
}
bool send_frame(int fd, uint8_t op, uint32_t sid, const void *payload, uint32_t len) {
    // header and payload in ONE buffer and one send: a frame is never split between two syscalls of different threads
    std::vector<uint8_t> buf(9 + (size_t)len);
    buf[0] = op;
    memcpy(&buf[1], &sid, 4);
    memcpy(&buf[5], &len, 4);
    if (len) memcpy(&buf[9], payload, len);
    return send_all(fd, buf.data(), buf.size());
}
bool recv_frame(int fd, uint8_t &op, uint32_t &sid, std::string &payload) {
    uint8_t h[9];
    if (!recv_all(fd, h, 9)) return false;
    op = h[0];
    uint32_t len;
    memcpy(&sid, h + 1, 4);
    memcpy(&len, h + 5, 4);
    payload.resize(len);
    return len == 0 || recv_all(fd, &payload[0], len);
}

int connect_to(const std::string &unix_path, const std::string &tcp) {
    if (!unix_path.empty()) {
        const int fd = ::socket(AF_UNIX, SOCK_STREAM, 0);
        sockaddr_un sa{};
        sa.sun_family = AF_UNIX;
        strncpy(sa.sun_path, unix_path.c_str(), sizeof(sa.sun_path) - 1);
        if (fd < 0 || ::connect(fd, (sockaddr *)&sa, sizeof(sa)) != 0) { perror("connect"); return -1; }
        return fd;
    }
    const size_t c = tcp.rfind(':');
    const std::string host = c == std::string::npos || c == 0 ? "127.0.0.1" : tcp.substr(0, c);
    const int port = atoi(tcp.c_str() + (c == std::string::npos ? 0 : c + 1));
    const int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    sockaddr_in sa{};
    sa.sin_family = AF_INET;
    sa.sin_port = htons((uint16_t)port);
    inet_pton(AF_INET, host.c_str(), &sa.sin_addr);
    if (fd < 0 || ::connect(fd, (sockaddr *)&sa, sizeof(sa)) != 0) { perror("connect"); return -1; }
    int one = 1;
    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
    return fd;
}

std::string json_escape(const std::string &s) {
    std::string o;
    for (unsigned char ch : s) {
        if (ch == '"' || ch == '\\') { o += '\\'; o += (char)ch; }
        else if (ch < 0x20) { char b[8]; snprintf(b, sizeof(b), "\\u%04x", ch); o += b; }
        else o += (char)ch;
    }
    return o;
}

}  // namespace

int main(int argc, char **argv) {
    std::string unix_path, tcp, pcm_dir, out_path, mode = "burst";
    int n_streams = 1, n_conns = 8, R = 0, push_chunks = 1;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() { return i + 1 < argc ? std::string(argv[++i]) : std::string(); };
        if (a == "--unix") unix_path = next();
        else if (a == "--tcp") tcp = next();
        else if (a == "--pcm-dir") pcm_dir = next();
        else if (a == "--streams") n_streams = atoi(next().c_str());
        else if (a == "--conns") n_conns = atoi(next().c_str());
        else if (a == "--right-context") R = atoi(next().c_str());
        else if (a == "--mode") mode = next();
        else if (a == "--push-chunks") push_chunks = atoi(next().c_str());
        else if (a == "--out") out_path = next();
        else { fprintf(stderr, "unknown flag %s\n", a.c_str()); return 2; }
    }
    if ((unix_path.empty() && tcp.empty()) || pcm_dir.empty() || n_streams < 1 || n_conns < 1 || push_chunks < 1 || (mode != "burst" && mode != "realtime")) {
        fprintf(stderr, "usage: %s --unix PATH | --tcp HOST:PORT --pcm-dir DIR --streams N [--conns C] [--right-context R] [--mode burst|realtime] [--push-chunks K] [--out report.json]\n", argv[0]);
        return 2;
    }
    n_conns = std::min(n_conns, n_streams);
    std::vector<StreamState> streams((size_t)n_streams);
    for (int i = 0; i < n_streams; i++) {
        char name[64];
        snprintf(name, sizeof(name), "/stream_%04d.s16", i);
        FILE *f = fopen((pcm_dir + name).c_str(), "rb");
        if (!f) { fprintf(stderr, "cannot open %s%s\n", pcm_dir.c_str(), name); return 1; }
        fseek(f, 0, SEEK_END);
        const long bytes = ftell(f);
        fseek(f, 0, SEEK_SET);
        streams[(size_t)i].pcm.resize((size_t)bytes / 2);
        if (bytes >= 2 && fread(streams[(size_t)i].pcm.data(), 2, (size_t)bytes / 2, f) != (size_t)bytes / 2) { fclose(f); return 1; }
        fclose(f);
    }
    const size_t n_push = (size_t)1280 * (size_t)(1 + R) * (size_t)push_chunks;
    std::vector<int> fds((size_t)n_conns);
    std::vector<std::vector<int>> by_conn((size_t)n_conns);
    for (int i = 0; i < n_streams; i++) by_conn[(size_t)(i % n_conns)].push_back(i);
    std::map<uint32_t, int> by_sid;
    for (int c = 0; c < n_conns; c++) {
        fds[(size_t)c] = connect_to(unix_path, tcp);
        if (fds[(size_t)c] < 0) return 1;
        for (int i : by_conn[(size_t)c]) {
            char cfg[96];
            const int n = snprintf(cfg, sizeof(cfg), "{\"lang\":\"auto\",\"right_context\":%d}", R);
            uint8_t op;
            uint32_t sid;
            std::string payload;
            if (!send_frame(fds[(size_t)c], OP_START, 0, cfg, (uint32_t)n) || !recv_frame(fds[(size_t)c], op, sid, payload) || op != OP_STARTED) {
                fprintf(stderr, "STREAM_START failed on connection %d\n", c);
                return 1;
            }
            streams[(size_t)i].sid = sid;
            by_sid[sid] = i;
        }
    }
    std::atomic<int> failures{0};
    g_t0 = Clock::now();
    const double t_start = 0.05;
    auto receiver = [&](int c) {
        int left = (int)by_conn[(size_t)c].size();
        while (left > 0) {
            uint8_t op;
            uint32_t sid;
            std::string payload;
            if (!recv_frame(fds[(size_t)c], op, sid, payload)) { failures++; return; }
            if (op == OP_ACK) continue;
            const double t = now_s();
            auto it = by_sid.find(sid);
            if (it == by_sid.end()) continue;
            StreamState &st = streams[(size_t)it->second];
            if (op == OP_TEXT || op == OP_ENDED) {
                st.text += payload;
                st.arrivals.emplace_back(t, st.text.size());
                if (op == OP_ENDED) { st.ended = t; left--; }
            } else {
                st.error = payload;
                failures++;
                left--;
            }
        }
    };
    auto sender = [&](int c) {
        const std::vector<int> &mine = by_conn[(size_t)c];
        size_t n_total = 0;
        for (int i : mine) n_total = std::max(n_total, streams[(size_t)i].pcm.size());
        const double period = (double)n_push / 16000.0;
        while (now_s() < t_start) std::this_thread::sleep_for(std::chrono::microseconds(200));
        for (size_t k = 0; k * n_push < n_total; k++) {
            if (mode == "realtime") {
                const double due = t_start + (double)k * period;
                const double d = due - now_s();
                if (d > 0) std::this_thread::sleep_for(std::chrono::duration<double>(d));
            }
            for (int i : mine) {
                StreamState &st = streams[(size_t)i];
                if (k * n_push >= st.pcm.size()) continue;
                const size_t n = std::min(n_push, st.pcm.size() - k * n_push);
                st.send_times.push_back(now_s());
                if (!send_frame(fds[(size_t)c], OP_PUSH, st.sid, st.pcm.data() + k * n_push, (uint32_t)(n * 2))) { failures++; return; }
            }
        }
        for (int i : mine)
            if (!send_frame(fds[(size_t)c], OP_END, streams[(size_t)i].sid, nullptr, 0)) { failures++; return; }
    };
    std::vector<std::thread> th;
    for (int c = 0; c < n_conns; c++) th.emplace_back(receiver, c);
    for (int c = 0; c < n_conns; c++) th.emplace_back(sender, c);
    for (auto &t : th) t.join();
    for (int fd : fds) ::close(fd);
    double t_end = 0, audio = 0;
    for (auto &st : streams) { t_end = std::max(t_end, st.ended); audio += (double)st.pcm.size() / 16000.0; }
    FILE *o = out_path.empty() ? stdout : fopen(out_path.c_str(), "w");
    if (!o) { perror("open report"); return 1; }
    fprintf(o, "{\"mode\":\"%s\",\"streams\":%d,\"conns\":%d,\"right_context\":%d,\"push_samples\":%zu,\"audio_seconds\":%.3f,\"wall_seconds\":%.6f,\"failures\":%d,\"per_stream\":[",
            mode.c_str(), n_streams, n_conns, R, n_push, audio, t_end - t_start, failures.load());
    for (size_t i = 0; i < streams.size(); i++) {
        const StreamState &st = streams[i];
        fprintf(o, "%s{\"sid\":%u,\"text\":\"%s\",\"error\":\"%s\",\"ended\":%.6f,\"send_times\":[", i ? "," : "", st.sid, json_escape(st.text).c_str(), json_escape(st.error).c_str(), st.ended);
        for (size_t k = 0; k < st.send_times.size(); k++) fprintf(o, "%s%.6f", k ? "," : "", st.send_times[k]);
        fprintf(o, "],\"arrivals\":[");
        for (size_t k = 0; k < st.arrivals.size(); k++) fprintf(o, "%s[%.6f,%zu]", k ? "," : "", st.arrivals[k].first, st.arrivals[k].second);
        fprintf(o, "]}");
    }
    fprintf(o, "]}\n");
    if (o != stdout) fclose(o);
    return failures.load() ? 1 : 0;
}
