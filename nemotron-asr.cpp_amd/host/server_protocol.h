// server_protocol.h -- the parts of the server that are pure functions of bytes: the reference's wire format
// (src/server-protocol.h:24-41: 9-byte frame = u8 opcode, u32le stream id, u32le payload length), the two JSON fields of a
// STREAM_START payload, and the chunk arithmetic of the batch former.  Shared by nemo_server.cpp and the sanitizer / fuzz harness
// (host/fuzz_harness.cpp, tests/test_sanitizers.py), which feeds them mutated frames without a GPU.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <string>

namespace nasr_proto {

enum : uint8_t { OP_STREAM_START = 0x01, OP_PUSH = 0x02, OP_STREAM_END = 0x03, OP_SET_LANG = 0x04,
                 OP_STARTED = 0x81, OP_ACK = 0x82, OP_TEXT = 0x83, OP_ENDED = 0x84, OP_LANG_SET = 0x85, OP_ERROR = 0x8F };
constexpr size_t kHeader = 9;
constexpr uint32_t kMaxPayload = 256u << 20;

inline void decode_header(const uint8_t *h, uint8_t &op, uint32_t &id, uint32_t &len) {
    op = h[0];
    id = (uint32_t)h[1] | ((uint32_t)h[2] << 8) | ((uint32_t)h[3] << 16) | ((uint32_t)h[4] << 24);
    len = (uint32_t)h[5] | ((uint32_t)h[6] << 8) | ((uint32_t)h[7] << 16) | ((uint32_t)h[8] << 24);
}
inline bool valid_right_context(int rc) { return rc == 0 || rc == 1 || rc == 6 || rc == 13; }

// minimal JSON field extraction for {"lang":"xx","right_context":N}
inline bool json_str(const std::string &j, const char *key, std::string &out) {
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
inline bool json_int(const std::string &j, const char *key, int &out) {
    const std::string k = std::string("\"") + key + "\"";
    size_t p = j.find(k);
    if (p == std::string::npos) return false;
    p = j.find(':', p + k.size());
    if (p == std::string::npos) return false;
    // like the reference's json_get_int (src/nemo-server.cpp:172-188): blanks, an optional '-', at least one digit -- anything else
    // (null, a string, a bare word) leaves `out` alone and returns false, so the server keeps its default; the value is clamped
    // instead of overflowing (round-4 advisor: atoi returned 0 for "13" / null and overrode --right-context)
    const char *c = j.c_str() + p + 1;
    while (*c == ' ' || *c == '\t') c++;
    bool neg = false;
    if (*c == '-') { neg = true; c++; }
    if (*c < '0' || *c > '9') return false;
    long long v = 0;
    for (; *c >= '0' && *c <= '9'; c++) { v = v * 10 + (*c - '0'); if (v > 2147483647LL) v = 2147483647LL; }
    out = (int)(neg ? -v : v);
    return true;
}


// ---- chunk arithmetic (reference src/preprocessor.cpp:220-221, :320-328, src/nemo-stream.h:65-81, src/nemo-stream.cpp:73-74): the audio
// buffer starts with 256 zeros of left padding, frames are 512 samples at hop 160, the mel buffer starts with the 9 literal-zero frames
// of the pre-encode cache, a chunk takes 9 + 8 T mel frames and advances by 8 T (T = 1 + right_context) -- so chunk k is complete once
// 8 T k frames exist.  Checked against the oracle's stream manager in tests/test_sanitizers.py. -------------------------------------
// samples that must have been handed for k chunks to be complete
inline int64_t samples_for_chunks(int64_t k, int T) { return k <= 0 ? 0 : 160 * (k * 8 * T - 1) + 256; }
// chunks complete once `samples` have been handed
inline int64_t chunks_after(int64_t samples, int T) {
    const int64_t frames = samples + 256 < 512 ? 0 : (samples + 256 - 512) / 160 + 1;
    return frames / (8 * T);
}

// The batch former's choice for ONE engine call (host/nemo_server.cpp, worker_loop::form_calls): `pending[i]` = whole chunks session i of one lookahead group holds (> 0),
// T = 1 + right_context rows per chunk, row_budget = rows one launch sequence may carry (nasr_engine_create_ex: workspace_rows), max_streams = the server's session capacity.
// Returns G, the chunks every chosen session completes in the call, and marks the chosen sessions in `take` (first come, first served, at most row_budget / (G T) of them):
//   * G is a power of two (every (streams, chunks) pair is a step shape with hipGraphs of its own: a backlog is worked off in 8 + 4 + 2 + 1, not in 13 sizes);
//   * G T <= 248 rows per session (a call's samples stay below the engine's MAX_PUSH of 256 encoder frames) and G <= row_budget / (max_streams T): never more chunks per
//     session than leave room for EVERY session of a full server (round 6: without this cap the rule below picked 16 chunks x 16 streams);
//   * among those, the G that carries the most rows; sessions holding fewer than G chunks are left for the next call instead of pulling the call down to their count; ties -> larger G.
inline int pick_call(const int *pending, int n, int T, int row_budget, int max_streams, bool *take) {
    for (int i = 0; i < n; i++) take[i] = false;
    if (n <= 0 || T <= 0) return 0;
    int gmax = 248 / T;
    const int per_session = row_budget / ((max_streams > 0 ? max_streams : 1) * T);
    if (per_session < gmax) gmax = per_session;
    if (gmax < 1) gmax = 1;
    int G = 1;
    long best_rows = 0;
    for (int g = 1; g <= gmax; g <<= 1) {
        int n_g = 0;
        for (int i = 0; i < n; i++) n_g += pending[i] >= g;
        const int room = row_budget / (g * T);
        if (n_g > room) n_g = room;
        if (n_g > 0 && (long)n_g * g >= best_rows) { best_rows = (long)n_g * g; G = g; }
    }
    int room = row_budget / (G * T), taken = 0;
    if (room < 1) room = 1;                    // a single session always fits: the engine cuts an over-long push into pieces itself
    for (int i = 0; i < n && taken < room; i++)
        if (pending[i] >= G) { take[i] = true; taken++; }
    return G;
}

}  // namespace nasr_proto
