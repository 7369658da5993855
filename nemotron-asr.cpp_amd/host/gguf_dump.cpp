// gguf_dump -- prints the header of a GGUF file as JSON (CPU-side conformance check of gguf_reader).
#include <cstdio>
#include <string>

#include "gguf_reader.h"

static void json_str(const std::string &s) {
    putchar('"');
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') { putchar('\\'); putchar(c); }
        else if (c < 0x20) printf("\\u%04x", c);
        else putchar(c);                       // UTF-8 passes through
    }
    putchar('"');
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s model.gguf [--full]\n", argv[0]); return 1; }
    const bool full = argc > 2 && std::string(argv[2]) == "--full";
    nasr_host::GgufFile g;
    std::string err;
    if (!g.open(argv[1], err)) { fprintf(stderr, "error: %s\n", err.c_str()); return 2; }
    printf("{\"version\": %u, \"n_kv\": %zu, \"data_start\": %llu, \"kv\": {", g.version(), g.n_kv(), (unsigned long long)g.data_start());
    const char *keys[] = {"nemo.n_mels", "nemo.d_model", "nemo.n_heads", "nemo.d_head", "nemo.d_ff", "nemo.n_layers", "nemo.vocab_size",
                          "nemo.decoder_dim", "nemo.joint_dim", "nemo.subsampling_factor", "nemo.att_left_context", "nemo.num_prompts", "nemo.kernel_size"};
    bool first = true;
    for (const char *k : keys) {
        uint32_t v;
        if (g.get_u32(k, v)) { printf("%s\"%s\": %u", first ? "" : ", ", k, v); first = false; }
    }
    const nasr_host::GgufValue *vl = g.find("tokenizer.vocab_list");
    printf("}, \"vocab_list\": %zu, ", vl ? vl->arr_s.size() : 0);
    if (full) {      // everything the model loader reads besides the hyper-parameters (host/nemo_amd.cpp, src/nemo-ggml.cpp:149-182)
        printf("\"name\": ");
        const nasr_host::GgufValue *nm = g.find("general.name");
        json_str(nm ? nm->s : "");
        printf(", \"vocab\": [");
        for (size_t i = 0; vl && i < vl->arr_s.size(); i++) { if (i) printf(", "); json_str(vl->arr_s[i]); }
        printf("], \"legacy_vocab\": ");
        const nasr_host::GgufValue *vb = g.find("tokenizer.vocab");
        if (vb && vb->type == 8) {
            printf("[");
            for (size_t i = 0; (i + 1) * 8 <= vb->s.size(); i++) {
                const char *rec = vb->s.data() + i * 8;
                size_t n = 0;
                while (n < 8 && rec[n]) n++;
                if (i) printf(", ");
                json_str(std::string(rec, n));
            }
            printf("]");
        } else printf("null");
        const nasr_host::GgufValue *pl = g.find("nemo.prompt_langs"), *pi = g.find("nemo.prompt_ids");
        printf(", \"prompt_langs\": [");
        for (size_t i = 0; pl && i < pl->arr_s.size(); i++) { if (i) printf(", "); json_str(pl->arr_s[i]); }
        printf("], \"prompt_ids\": [");
        for (size_t i = 0; pi && i < pi->arr_i.size(); i++) printf("%s%lld", i ? ", " : "", (long long)pi->arr_i[i]);
        printf("], ");
    }
    printf("\"tensors\": [");
    first = true;
    for (const auto &t : g.tensors()) {
        uint64_t sum = 0;
        for (uint64_t i = 0; i < t.nbytes; i++) sum = sum * 1099511628211ull + t.data[i];
        printf("%s{\"name\": \"%s\", \"type\": %d, \"ne\": [%lld, %lld, %lld, %lld], \"n_dims\": %d, \"offset\": %llu, \"nbytes\": %llu, \"hash\": \"%016llx\"}",
               first ? "" : ", ", t.name.c_str(), t.type, (long long)t.ne[0], (long long)t.ne[1], (long long)t.ne[2], (long long)t.ne[3],
               t.n_dims, (unsigned long long)t.offset, (unsigned long long)t.nbytes, (unsigned long long)sum);
        first = false;
    }
    printf("]}\n");
    return 0;
}
