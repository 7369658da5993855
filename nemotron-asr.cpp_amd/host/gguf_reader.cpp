// gguf_reader.cpp -- see gguf_reader.h
#include "gguf_reader.h"

#include <cstdint>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace nasr_host {

uint64_t ggml_type_nbytes(int32_t type, int64_t n) {
    switch (type) {
    case 0: return (uint64_t)n * 4;                 // F32
    case 1: return (uint64_t)n * 2;                 // F16
    case 2: return n % 32 ? 0 : (uint64_t)n / 32 * 18;   // Q4_0
    case 8: return n % 32 ? 0 : (uint64_t)n / 32 * 34;   // Q8_0
    }
    return 0;
}

GgufFile::~GgufFile() {
    if (map_) munmap(map_, map_size_);
}

namespace {
struct Cursor {
    const uint8_t *p, *end;
    bool ok = true;
    template <typename T> T get() {
        T v{};
        if (p + sizeof(T) > end) { ok = false; return v; }
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    std::string str() {
        uint64_t n = get<uint64_t>();
        if (!ok || n > (uint64_t)(end - p)) { ok = false; return {}; }
        std::string s((const char *)p, (size_t)n);
        p += n;
        return s;
    }
};

bool read_scalar(Cursor &c, int32_t t, GgufValue &v) {
    switch (t) {
    case 0: v.u = c.get<uint8_t>(); break;
    case 1: v.u = (uint64_t)(int64_t)c.get<int8_t>(); break;
    case 2: v.u = c.get<uint16_t>(); break;
    case 3: v.u = (uint64_t)(int64_t)c.get<int16_t>(); break;
    case 4: v.u = c.get<uint32_t>(); break;
    case 5: v.u = (uint64_t)(int64_t)c.get<int32_t>(); break;
    case 6: v.f = c.get<float>(); break;
    case 7: v.u = c.get<uint8_t>(); break;
    case 10: v.u = c.get<uint64_t>(); break;
    case 11: v.u = (uint64_t)c.get<int64_t>(); break;
    case 12: v.f = c.get<double>(); break;
    default: return false;
    }
    return c.ok;
}
}  // namespace

bool GgufFile::open(const std::string &path, std::string &err) {
    int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) { err = "cannot open " + path; return false; }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 24) { ::close(fd); err = "file too small"; return false; }
    map_size_ = (size_t)st.st_size;
    map_ = (uint8_t *)mmap(nullptr, map_size_, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (map_ == MAP_FAILED) { map_ = nullptr; err = "mmap failed"; return false; }
    Cursor c{map_, map_ + map_size_};
    if (memcmp(c.p, "GGUF", 4) != 0) { err = "not a GGUF file"; return false; }
    c.p += 4;
    version_ = c.get<uint32_t>();
    if (version_ < 2 || version_ > 3) { err = "unsupported GGUF version " + std::to_string(version_); return false; }
    const int64_t n_tensors = c.get<int64_t>(), n_kv = c.get<int64_t>();
    if (!c.ok || n_tensors < 0 || n_kv < 0 || n_tensors > (1 << 20) || n_kv > (1 << 20)) { err = "bad header counts"; return false; }
    uint32_t alignment = 32;
    for (int64_t i = 0; i < n_kv; i++) {
        std::string key = c.str();
        GgufValue v;
        v.type = c.get<int32_t>();
        if (!c.ok) { err = "truncated metadata"; return false; }
        if (v.type == 8) v.s = c.str();
        else if (v.type == 9) {
            v.arr_type = c.get<int32_t>();
            const uint64_t n = c.get<uint64_t>();
            if (!c.ok || n > (1u << 24)) { err = "bad array in key " + key; return false; }
            for (uint64_t k = 0; k < n; k++) {
                if (v.arr_type == 8) v.arr_s.push_back(c.str());
                else {
                    GgufValue e;
                    if (!read_scalar(c, v.arr_type, e)) { err = "bad array element type in key " + key; return false; }
                    v.arr_i.push_back(v.arr_type == 6 || v.arr_type == 12 ? (int64_t)e.f : (int64_t)e.u);
                }
            }
        } else if (!read_scalar(c, v.type, v)) { err = "bad value type for key " + key; return false; }
        if (!c.ok) { err = "truncated metadata at key " + key; return false; }
        if (key == "general.alignment" && v.type == 4) alignment = (uint32_t)v.u;
        kv_[key] = std::move(v);
    }
    tensors_.resize((size_t)n_tensors);
    for (auto &t : tensors_) {
        t.name = c.str();
        t.n_dims = (int32_t)c.get<uint32_t>();
        if (!c.ok || t.n_dims < 0 || t.n_dims > 4) { err = "bad tensor info"; return false; }
        int64_t numel = 1;
        for (int d = 0; d < t.n_dims; d++) {
            t.ne[d] = c.get<int64_t>();
            // every extent positive, product checked against overflow (a corrupt header must not wrap numel)
            if (!c.ok || t.ne[d] <= 0 || numel > (int64_t)(INT64_MAX / 64) / t.ne[d]) { err = "bad tensor shape: " + t.name; return false; }
            numel *= t.ne[d];
        }
        t.type = c.get<int32_t>();
        t.offset = c.get<uint64_t>();
        t.nbytes = ggml_type_nbytes(t.type, numel);
        if (!c.ok || t.nbytes == 0) { err = "unsupported tensor type/shape: " + t.name; return false; }
    }
    if (alignment == 0 || (alignment & (alignment - 1)) != 0) { err = "general.alignment must be a power of two"; return false; }
    data_start_ = ((uint64_t)(c.p - map_) + alignment - 1) / alignment * alignment;
    if (data_start_ > map_size_) { err = "tensor data section starts past the end of the file"; return false; }
    const uint64_t data_size = map_size_ - data_start_;
    for (size_t i = 0; i < tensors_.size(); i++) {
        auto &t = tensors_[i];
        // written so that a huge offset cannot wrap the sum
        if (t.offset > data_size || t.nbytes > data_size - t.offset) { err = "tensor data out of file bounds: " + t.name; return false; }
        t.data = map_ + data_start_ + t.offset;
        by_name_[t.name] = i;
    }
    return true;
}

const GgufValue *GgufFile::find(const std::string &key) const {
    auto it = kv_.find(key);
    return it == kv_.end() ? nullptr : &it->second;
}
bool GgufFile::get_u32(const std::string &key, uint32_t &out) const {
    const GgufValue *v = find(key);
    if (!v || (v->type != 4 && v->type != 5)) return false;
    out = (uint32_t)v->u;
    return true;
}
const GgufTensor *GgufFile::tensor(const std::string &name) const {
    auto it = by_name_.find(name);
    return it == by_name_.end() ? nullptr : &tensors_[it->second];
}

}  // namespace nasr_host
