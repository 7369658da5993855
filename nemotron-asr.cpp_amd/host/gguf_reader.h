// gguf_reader.h -- ggml-free GGUF v3 reader (replaces gguf_init_from_file + the gguf_get_* calls of
// reference src/nemo-ggml.cpp:99-182, :214-283; file layout: reference docs/TENSOR_FORMAT.md and
// scripts/convert_to_gguf.py:491-540).  Host-only, no GPU dependency.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace nasr_host {

struct GgufTensor {
    std::string name;
    int32_t type = 0;          // ggml type id: 0 F32, 1 F16, 2 Q4_0, 8 Q8_0
    int32_t n_dims = 0;
    int64_t ne[4] = {1, 1, 1, 1};   // ggml order, ne[0] fastest
    uint64_t offset = 0;       // relative to data_start
    uint64_t nbytes = 0;
    const uint8_t *data = nullptr;
};

struct GgufValue {
    int32_t type = -1;         // GGUF metadata value type
    uint64_t u = 0;            // integer / bool payload
    double f = 0;              // float payload
    std::string s;             // string payload
    int32_t arr_type = -1;
    std::vector<std::string> arr_s;
    std::vector<int64_t> arr_i;
};

class GgufFile {
public:
    ~GgufFile();
    bool open(const std::string &path, std::string &err);
    const GgufValue *find(const std::string &key) const;
    bool get_u32(const std::string &key, uint32_t &out) const;
    const std::vector<GgufTensor> &tensors() const { return tensors_; }
    const GgufTensor *tensor(const std::string &name) const;
    uint64_t data_start() const { return data_start_; }
    uint32_t version() const { return version_; }
    size_t n_kv() const { return kv_.size(); }

private:
    std::map<std::string, GgufValue> kv_;
    std::vector<GgufTensor> tensors_;
    std::map<std::string, size_t> by_name_;
    uint8_t *map_ = nullptr;
    size_t map_size_ = 0;
    uint64_t data_start_ = 0;
    uint32_t version_ = 0;
};

uint64_t ggml_type_nbytes(int32_t type, int64_t numel);   // 0 if unsupported

}  // namespace nasr_host
