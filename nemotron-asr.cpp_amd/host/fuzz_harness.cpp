// fuzz_harness -- CPU-only harness for the sanitizer build (make -C host san -> bin/host_fuzz_san, built with
// -fsanitize=address,undefined): feeds the byte-level parsers of the host side whatever files it is given.  No GPU, no engine.
//   host_fuzz_san gguf   FILE...   every file through the GGUF reader: open, every KV looked at, every tensor's bytes summed
//   host_fuzz_san frames FILE...   every file as a byte stream of wire frames (server_protocol.h): headers decoded, payload bounds
//                                  checked as the server's reader does, STREAM_START payloads through the JSON helpers, PUSH sizes
//                                  through the batch former's chunk arithmetic
// Prints one line per file ("ok ..." or "rejected: ...") and exits 0: a malformed input must be REJECTED, never crash -- what the test
// (tests/test_sanitizers.py) asserts is that the sanitizers stay silent over a corpus of mutated files.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <memory>
#include <vector>

#include "gguf_reader.h"
#include "server_protocol.h"

static int run_gguf(const char *path) {
    nasr_host::GgufFile f;
    std::string err;
    if (!f.open(path, err)) { printf("rejected: %s: %s\n", path, err.c_str()); return 0; }
    unsigned long long sum = 0;
    uint32_t v = 0;
    for (const char *k : {"nemo.n_layers", "nemo.d_model", "nemo.vocab_size", "nemo.num_prompts", "general.alignment"}) sum += f.get_u32(k, v) ? v : 0;
    if (const nasr_host::GgufValue *vv = f.find("tokenizer.ggml.tokens")) for (const std::string &s : vv->arr_s) sum += s.size();
    if (const nasr_host::GgufValue *vv = f.find("tokenizer.vocab")) sum += vv->s.size();
    for (const nasr_host::GgufTensor &t : f.tensors()) {
        sum += (unsigned long long)t.ne[0] + (unsigned long long)t.n_dims;
        if (t.data && t.nbytes) { sum += t.data[0]; sum += t.data[t.nbytes - 1]; sum += t.data[t.nbytes / 2]; }     // first / middle / last byte: inside the mapping?
    }
    printf("ok: %s: %zu kv, %zu tensors, checksum %llu\n", path, f.n_kv(), f.tensors().size(), sum);
    return 0;
}

static int run_frames(const char *path) {
    using namespace nasr_proto;
    std::ifstream in(path, std::ios::binary);
    std::vector<uint8_t> b((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    size_t off = 0, n_frames = 0, rejected = 0;
    long long chunks = 0, handed = 0;
    int T = 1;
    while (off + kHeader <= b.size()) {
        uint8_t op;
        uint32_t id, len;
        decode_header(b.data() + off, op, id, len);
        off += kHeader;
        if (len > kMaxPayload) { rejected++; break; }                     // the server answers ERROR and closes
        if (len > b.size() - off) { rejected++; break; }                  // short read: connection closed
        const uint8_t *payload = b.data() + off;
        off += len;
        n_frames++;
        switch (op) {
        case OP_STREAM_START: {
            const std::string cfg(payload, payload + len);
            std::string lang;
            int rc = 0;
            json_str(cfg, "lang", lang);
            json_int(cfg, "right_context", rc);
            if (!valid_right_context(rc)) { rejected++; break; }
            T = 1 + rc;
            handed = 0;
        } break;
        case OP_PUSH: {
            handed += len / 2;
            chunks = chunks_after(handed, T);
            if (samples_for_chunks(chunks, T) > handed) { printf("BUG: chunk arithmetic\n"); return 1; }
            if (samples_for_chunks(chunks + 1, T) <= handed) { printf("BUG: chunk arithmetic\n"); return 1; }
        } break;
        default: break;
        }
        (void)id;
    }
    printf("ok: %s: %zu frames, %zu rejected, %lld chunks\n", path, n_frames, rejected, chunks);
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 4 && !strcmp(argv[1], "chunks")) {      // chunks T S...: chunks complete after S samples, and the samples that complete them (host logic test)
        const int T = atoi(argv[2]);
        for (int i = 3; i < argc; i++) {
            const long long S = atoll(argv[i]), k = nasr_proto::chunks_after(S, T);
            printf("%lld %lld %lld\n", S, k, (long long)nasr_proto::samples_for_chunks(k, T));
        }
        return 0;
    }
    if (argc >= 4 && !strcmp(argv[1], "jsonint")) {     // jsonint DEFAULT JSON...: what the server's right_context would be after json_int (host logic test)
        for (int i = 3; i < argc; i++) {
            int v = atoi(argv[2]);
            const bool found = nasr_proto::json_int(argv[i], "right_context", v);
            printf("%d %d\n", found ? 1 : 0, v);
        }
        return 0;
    }
    if (argc >= 6 && !strcmp(argv[1], "pickcall")) {    // pickcall T ROW_BUDGET MAX_STREAMS PENDING...: the batch former's choice for one call -> "G take0 take1 ..." (host logic test)
        const int T = atoi(argv[2]), budget = atoi(argv[3]), cap = atoi(argv[4]), n = argc - 5;
        std::vector<int> pend((size_t)n);
        for (int i = 0; i < n; i++) pend[(size_t)i] = atoi(argv[5 + i]);
        std::unique_ptr<bool[]> take(new bool[(size_t)n]);
        const int G = nasr_proto::pick_call(pend.data(), n, T, budget, cap, take.get());
        printf("%d", G);
        for (int i = 0; i < n; i++) printf(" %d", take[(size_t)i] ? 1 : 0);
        printf("\n");
        return 0;
    }
    if (argc < 3) { fprintf(stderr, "usage: %s gguf|frames FILE... | chunks T SAMPLES... | jsonint DEFAULT JSON... | pickcall T ROW_BUDGET MAX_STREAMS PENDING...\n", argv[0]); return 2; }
    int rc = 0;
    for (int i = 2; i < argc; i++) rc |= !strcmp(argv[1], "gguf") ? run_gguf(argv[i]) : run_frames(argv[i]);
    return rc;
}
