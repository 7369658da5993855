// gguf_dump -- prints the header of a GGUF file as JSON (CPU-side conformance check of gguf_reader).
#include <cstdio>
#include <string>

#include "gguf_reader.h"

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s model.gguf\n", argv[0]); return 1; }
    nasr_host::GgufFile g;
    std::string err;
    if (!g.open(argv[1], err)) { fprintf(stderr, "error: %s\n", err.c_str()); return 2; }
    printf("{\"version\": %u, \"n_kv\": %zu, \"data_start\": %llu, \"kv\": {", g.version(), g.n_kv(), (unsigned long long)g.data_start());
    const char *keys[] = {"nemo.n_mels", "nemo.d_model", "nemo.n_heads", "nemo.d_head", "nemo.d_ff", "nemo.n_layers", "nemo.vocab_size",
                          "nemo.decoder_dim", "nemo.joint_dim", "nemo.subsampling_factor", "nemo.att_left_context", "nemo.num_prompts"};
    bool first = true;
    for (const char *k : keys) {
        uint32_t v;
        if (g.get_u32(k, v)) { printf("%s\"%s\": %u", first ? "" : ", ", k, v); first = false; }
    }
    const nasr_host::GgufValue *vl = g.find("tokenizer.vocab_list");
    printf("}, \"vocab_list\": %zu, \"tensors\": [", vl ? vl->arr_s.size() : 0);
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
